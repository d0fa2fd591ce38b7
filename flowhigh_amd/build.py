"""Build libflowhigh_hip.so (gfx950) in-tree with hipcc.  `python -m flowhigh_amd.build`."""
import os
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "lib" / "libflowhigh_hip.so"
SOURCES = ["api_common.hip", "conv_mfma.hip", "conv_wino.hip", "conv_wino54.hip", "conv_wino54_bf.hip", "amp_fused.hip", "act1d.hip", "gemm_mfma.hip", "flow_ops.hip",
           "attention.hip", "frontend.hip", "fft.hip"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
# per-source extra flags.  The bf16 x 6 kernels keep everything beside their MFMAs one result per lane: the SLP vectoriser
# would re-pack it into v_pk_*_f32, which stall a bf16 MFMA (conv_wino54_kernel.h)
EXTRA_FLAGS = {"conv_wino54_bf.hip": ["-fno-slp-vectorize"]}
HEADERS = ["fh_common.h", "conv_wino54_kernel.h"]


def _deps():
    return [CSRC / h for h in HEADERS] + [PKG.parent / "include" / "flowhigh_hip.h"]


def needs_build():
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    return any(d.stat().st_mtime > t for d in [CSRC / s for s in SOURCES] + _deps())


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    LIB.parent.mkdir(parents=True, exist_ok=True)
    objdir = PKG / "build"
    objdir.mkdir(exist_ok=True)
    hdr_t = max(d.stat().st_mtime for d in _deps())
    procs = []
    for s in SOURCES:
        obj = objdir / (s + ".o")
        src = CSRC / s
        if not force and obj.exists() and obj.stat().st_mtime > max(src.stat().st_mtime, hdr_t):
            continue
        cmd = [hipcc, *FLAGS, *EXTRA_FLAGS.get(s, []), "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {s}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB)] + [str(objdir / (s + ".o")) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
