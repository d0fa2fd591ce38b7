"""Model construction time at SYNTH-CFG: from_local on the reference's three checkpoint files, the one-time
conversion, and from_local through the weight blob.  GPU box:  python tools/load_time.py [dir=/tmp/fh_load_time]"""
import os
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from flowhigh_amd import FlowHighSR, convert, synth    # noqa: E402

d = Path(sys.argv[1] if len(sys.argv) > 1 else "/tmp/fh_load_time")
if not (d / "FLowHigh_basic_400k.pt").exists():
    synth.write_checkpoint_dir(d, synth.SYNTH_CFG, seed=0)
torch.zeros(1, device="cuda")                      # HIP runtime up before anything is timed
os.environ["FH_ACT_BLOCKS"] = "0"                  # (the occupancy calibration is ~0.6 s of launches: not load time)


def timed(label):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m = FlowHighSR.from_local(d, "cuda")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{label}: {dt:.2f} s")
    return m


os.environ["FH_BLOB"] = "0"
timed("from_local, checkpoint files (torch.load x 2 + packing + ~440 uploads), first call")
timed("from_local, checkpoint files, second call (files in the page cache)")
t0 = time.perf_counter()
r = convert.convert(d)
print(f"python -m flowhigh_amd.convert: {time.perf_counter() - t0:.2f} s ({r['bytes'] / 2 ** 20:.0f} MiB, {r['tensors']} tensors; "
      f"read {r['read_s']:.1f} s, pack {r['pack_s']:.1f} s, write {r['write_s']:.1f} s)")
del os.environ["FH_BLOB"]
timed("from_local, weight blob (map + digests of the 3 source files + one H2D copy), first call")
timed("from_local, weight blob, second call")
os.environ["FH_BLOB_VERIFY"] = "0"
timed("from_local, weight blob, FH_BLOB_VERIFY=0")
