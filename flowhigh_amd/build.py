"""Build libflowhigh_hip.so (gfx950) in-tree with hipcc.  `python -m flowhigh_amd.build`."""
import os
import re
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "lib" / "libflowhigh_hip.so"
SOURCES = ["api_common.hip", "conv_mfma.hip", "conv_wino.hip", "conv_wino54.hip", "conv_wino54_bf.hip", "amp_fused.hip", "narrow_bf.hip", "act1d.hip", "gemm_mfma.hip", "gemm_bf.hip", "flow_ops.hip",
           "attention.hip", "frontend.hip", "fft.hip"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
# per-source extra flags.  The bf16 x 6 kernels keep everything beside their MFMAs one result per lane: the SLP vectoriser
# would re-pack it into v_pk_*_f32, which stall a bf16 MFMA (conv_wino54_kernel.h)
# (conv_wino.hip as a whole: its bf16 x 6 instantiations need it for the same reason, and without it the 128 x 256 one spills)
EXTRA_FLAGS = {"conv_wino54_bf.hip": ["-fno-slp-vectorize"], "conv_wino.hip": ["-fno-slp-vectorize"]}
# every kernel's resources are read from the compiler's remarks: a kernel that spills more than a few registers fails the build.
# The conv kernels fill every register they are given; a variant that spills a hundred (it happened three times in round 6: the
# scheduler hoisting the next tile column's work until the file is full) runs its K loop through scratch.  (Round 6 also found
# such variants computing garbage: an inline-asm prefetch wrote a register the compiler had meanwhile given to something else --
# fixed at the source, conv_wino.hip: prefetch_a; the limit here is about speed.)
RESOURCE_FLAGS = ["-Rpass-analysis=kernel-resource-usage"]
MAX_SCRATCH_BYTES = 16          # per lane: up to 4 spilled registers (prologue / epilogue values) are tolerated and reported
HEADERS = ["fh_common.h", "conv_wino54_kernel.h"]


def _deps():
    return [CSRC / h for h in HEADERS] + [PKG.parent / "include" / "flowhigh_hip.h"]


def needs_build():
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    return any(d.stat().st_mtime > t for d in [CSRC / s for s in SOURCES] + _deps())


def parse_resource_remarks(stderr_text):
    """{kernel name: {field: int}} from hipcc's -Rpass-analysis=kernel-resource-usage remarks."""
    import re
    kernels, cur = {}, None
    for ln in stderr_text.splitlines():
        m = re.search(r"remark: (?:[^:]*: )?Function Name: (\S+)", ln)
        if m:
            cur = kernels.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark: (?:[^:]*: )?\s*([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", ln)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return kernels


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
        cmd = [hipcc, *FLAGS, *EXTRA_FLAGS.get(s, []), *RESOURCE_FLAGS, "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((s, obj, subprocess.Popen(cmd, stderr=subprocess.PIPE, text=True)))
    spilled = []
    for s, obj, p in procs:
        err = p.communicate()[1]
        # the resource remarks are read, everything else the compiler said is passed on
        kernels = parse_resource_remarks(err)
        rest, in_remark = [], False
        lines = err.splitlines()
        for n, ln in enumerate(lines):               # (a remark = its header line + the source line and caret clang prints under it)
            if ln.startswith("In file included from") and n + 1 < len(lines) and "remark:" in lines[n + 1]:
                continue
            if "remark:" in ln:
                in_remark = True
                continue
            if in_remark and (re.match(r"\s*\d*\s*\|", ln) or ln.startswith("In file included from") or not ln.strip()):
                continue
            in_remark = False
            rest.append(ln)
        rest = [ln for ln in rest if not re.match(r"\d+ warnings? generated", ln) or any("warning:" in r_ for r_ in rest)]
        if verbose and any(ln.strip() for ln in rest):
            print("\n".join(rest), file=sys.stderr, flush=True)
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}")
        for k, r in kernels.items():
            if verbose and 0 < r.get("ScratchSize", 0) <= MAX_SCRATCH_BYTES:
                print(f"note: {s}: {k} spills {r.get('VGPRs Spill', '?')} registers ({r['ScratchSize']} bytes of scratch per lane)", flush=True)
        bad = [(k, r) for k, r in kernels.items() if r.get("ScratchSize", 0) > MAX_SCRATCH_BYTES]
        if bad:
            obj.unlink(missing_ok=True)
            spilled += [f"{s}: {k} needs {r['ScratchSize']} bytes of scratch per lane ({r.get('VGPRs Spill', '?')} VGPRs spilled)" for k, r in bad]
    if spilled:
        raise RuntimeError(f"kernels that need more than {MAX_SCRATCH_BYTES} bytes of scratch per lane are not built:\n  " + "\n  ".join(spilled))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB)] + [str(objdir / (s + ".o")) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
