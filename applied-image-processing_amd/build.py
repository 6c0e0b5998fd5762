"""Builds the gfx950 shared library in-tree: ``applied-image-processing_amd/libadain_hip.so``.

hipcc cross-compiles without a GPU.  Objects are rebuilt only when a source or header is newer.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
INC = os.path.join(os.path.dirname(PKG), "include")
LIB = os.path.join(PKG, "libadain_hip.so")
DIAG_LIB = os.path.join(PKG, "libadain_hip_diag.so")     # -DADAIN_DIAG: env tuning switches, stamp / timing-only kernels (tools/ only)
# Both libraries are these sources; the diagnostic one adds -DADAIN_DIAG (environment tuning switches, stamp / timing-only variants of
# the F(4,3) x F(2,3) kernel).  The direct implicit-GEMM and F(2x2,3x3) families of rounds 1-2 were retired in round 6 (git history).
SOURCES = ["conv_edge.hip", "conv_wino4.hip", "stats.hip", "pixel.hip", "resample.hip", "api.hip"]
DIAG_SOURCES = []
# -fvisibility=hidden: the shared library exports the C ABI of include/adain_hip.h (ADAIN_API) and nothing else
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]
# The MFMA kernels carry their fp32 vector-ALU work (input transform, epilogues) next to the matrix instructions, where
# v_pk_add_f32 / v_pk_fma_f32 issue far slower than the plain forms (MI355X_MICROARCH.md, "price of one filler beside
# MFMAs"): keep hipcc from packing f32 pairs in those files.  Measured on the Winograd kernel: +7 %.
NO_PACKED_F32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
EXTRA_FLAGS = {"conv_wino4.hip": NO_PACKED_F32}


def _hipcc():
    c = os.environ.get("HIPCC")
    if c:
        return c
    return "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else "hipcc"


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, diag=False):
    """Builds the product library; ``diag=True`` builds ``libadain_hip_diag.so`` instead (the same sources with -DADAIN_DIAG:
    environment tuning switches, stamp / timing-only variants used by tools/).  A process loads it explicitly
    with ``runtime.use_library(runtime.DIAG_LIB_PATH)`` - nothing in the environment selects it."""
    objdir = os.path.join(PKG, "build_diag" if diag else "build")
    os.makedirs(objdir, exist_ok=True)
    lib = DIAG_LIB if diag else LIB
    flags = FLAGS + (["-DADAIN_DIAG"] if diag else []) + os.environ.get("ADAIN_EXTRA_HIPCC_FLAGS", "").split()
    headers = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "device_utils.h"), os.path.join(INC, "adain_hip.h"),
               os.path.join(INC, "adain_hip_diag.h")]
    hipcc = _hipcc()
    jobs = []
    objs = []
    for src in SOURCES + (DIAG_SOURCES if diag else []):
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _newer(o, [s] + headers):
            jobs.append([hipcc] + flags + EXTRA_FLAGS.get(src, []) + ["-I", INC, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr)
        return r.stderr

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _newer(lib, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    return lib


if __name__ == "__main__":
    if "--all" in sys.argv:
        print(build(force="--force" in sys.argv, verbose=True))
        print(build(force="--force" in sys.argv, verbose=True, diag=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True, diag="--diag" in sys.argv))
