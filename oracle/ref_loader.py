"""ORACLE — TEST INFRASTRUCTURE ONLY.  Imports the reference's own, unmodified
``function.py`` / ``net.py`` / ``test.py`` from /root/reference (read-only, exists only in the
build container — never on the GPU box) so golden vectors can be generated from the real code.

Recipe: no bytecode is written into the reference tree; ``torchvision``
and ``cv2`` (absent from the image, imported at the top of test.py:8-10) are replaced by empty
stub modules; a synthetic package ``refadain`` whose ``__path__`` is the reference AdaIN directory
skips both ``__init__.py`` files (they chain-import the CUDA-only 3DGS stack).
"""
import importlib
import os
import sys
import types

REF_DIR = "/root/reference/Style_3DGS/AdaIN"


def available():
    return os.path.isfile(os.path.join(REF_DIR, "function.py"))


def load():
    """Returns (function, net, test) reference modules."""
    if not available():
        raise RuntimeError("reference tree not present (expected only in the build container)")
    sys.dont_write_bytecode = True
    for name in ("torchvision", "torchvision.transforms", "torchvision.utils", "cv2"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision"].utils = sys.modules["torchvision.utils"]
    if not hasattr(sys.modules["torchvision.utils"], "save_image"):
        sys.modules["torchvision.utils"].save_image = lambda *a, **k: None
    if "refadain" not in sys.modules:
        pkg = types.ModuleType("refadain")
        pkg.__path__ = [REF_DIR]
        sys.modules["refadain"] = pkg
    fn = importlib.import_module("refadain.function")
    net = importlib.import_module("refadain.net")
    test = importlib.import_module("refadain.test")
    return fn, net, test


def load_localized():
    """The reference's unmodified Style_3DGS/localized_style_transfer.py (host-side Reinhard / PCA / CDF colour transfer, :22-168).
    Its module-level imports that cannot load here are replaced by empty stub modules, exactly as for test.py above:
    ``torchvision`` (models, transforms.functional) and the package ``Style_3DGS.AdaIN`` (whose real __init__ chain-imports the
    CUDA-only 3DGS stack).  ``sklearn`` and ``matplotlib`` are installed and import normally."""
    import importlib.util

    path = "/root/reference/Style_3DGS/localized_style_transfer.py"
    if not os.path.isfile(path):
        raise RuntimeError("reference tree not present (expected only in the build container)")
    sys.dont_write_bytecode = True
    tv = sys.modules.setdefault("torchvision", types.ModuleType("torchvision"))
    for sub in ("models", "transforms", "transforms.functional"):
        name = "torchvision." + sub
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    tv.models = sys.modules["torchvision.models"]
    tv.transforms = sys.modules["torchvision.transforms"]
    tv.transforms.functional = sys.modules["torchvision.transforms.functional"]
    if "Style_3DGS" not in sys.modules:
        pkg = types.ModuleType("Style_3DGS")
        pkg.__path__ = []
        sys.modules["Style_3DGS"] = pkg
    if "Style_3DGS.AdaIN" not in sys.modules:
        sub = types.ModuleType("Style_3DGS.AdaIN")
        sub.adain_inference = None
        sys.modules["Style_3DGS.AdaIN"] = sub
    spec = importlib.util.spec_from_file_location("reflocalized", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod
