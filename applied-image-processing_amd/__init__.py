"""MI355X-native AdaIN style-transfer inference path (gfx950 HIP kernels behind a C ABI).

Drop-in for the reference's ``Style_3DGS/AdaIN`` Python surface
(reference: Style_3DGS/AdaIN/__init__.py:1 re-exports ``adain_inference`` and
``get_style_embeddings`` from test.py).  Import is lazy so that ``arch`` / ``synth`` can be used
without torch or a GPU.
"""

__all__ = ["adain_inference", "get_style_embeddings"]


def __getattr__(name):
    if name in __all__:
        from .AdaIN import test as _t

        return getattr(_t, name)
    raise AttributeError(name)
