import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def weights():
    """Seed-0 synthetic weights as torch CPU state_dicts (reference key layout)."""
    import applied_image_processing_amd.synth as synth

    return synth.to_torch(synth.vgg_state_dict(0, full=False)), synth.to_torch(synth.decoder_state_dict(0))


@pytest.fixture(scope="session")
def weights_tl():
    """The TRAINED-LIKE weight set (synth.trained_like_state_dicts: Caffe-style conv0, non-zero-mean / zero-sum filters, channels
    normalised to a post-ReLU mean near 1) as torch CPU state_dicts; the encoder with all 17 convs (a strict load_state_dict works)."""
    import applied_image_processing_amd.synth as synth

    vgg, dec = synth.trained_like_state_dicts(0)
    return synth.to_torch(vgg), synth.to_torch(dec)


def golden(name):
    import numpy as np

    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture
def diag_lib():
    """Runs the test against the DIAGNOSTIC build of the library (libadain_hip_diag.so: the product sources with -DADAIN_DIAG -
    environment tuning switches and the stamp kernels tools/ use) and switches back to the product library afterwards."""
    import applied_image_processing_amd.runtime as rt

    if not os.path.exists(rt.DIAG_LIB_PATH):
        import __graft_entry__ as g

        g.build()
    product = os.path.join(os.path.dirname(rt.DIAG_LIB_PATH), "libadain_hip.so")
    rt.use_library(rt.DIAG_LIB_PATH)
    try:
        yield rt
    finally:
        rt.use_library(product)
