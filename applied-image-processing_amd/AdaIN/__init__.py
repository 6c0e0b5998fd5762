"""Drop-in for the reference package ``Style_3DGS.AdaIN`` (reference Style_3DGS/AdaIN/__init__.py:1)."""
from .test import adain_inference, get_style_embeddings  # noqa: F401
