"""Imported first by the probe scripts: points the package at the DIAGNOSTIC build of the C-ABI library
(``python applied-image-processing_amd/build.py --diag`` -> libadain_hip_diag.so: environment tuning switches, stamp and
timing-only kernel variants, ``adain_debug_set_conv_stamp_buffer``).  The product library has none of these."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
_lib = os.path.join(ROOT, "applied-image-processing_amd", "libadain_hip_diag.so")
if not os.path.exists(_lib):
    raise SystemExit(f"{_lib} is missing: build it with `python applied-image-processing_amd/build.py --diag`")
import applied_image_processing_amd.runtime as _rt  # noqa: E402

_rt.use_library(_lib)
