"""`python -m flowhigh_amd.convert <ckpt_dir> [<blob>]`: pack the reference's checkpoint files once into the flat weight blob
that `FlowHighSR.from_local` maps and uploads with one copy (flowhigh_amd/weights.py).

Runs on the CPU (no GPU, no HIP library): the packers are plain torch.  The blob is tied to the content of the three source
files and to the layout switches in force (FH_WINO, FH_WINO54, FH_AMP, FH_CONV_BF16X6 ...): from_local ignores a blob that was
made from other files or under other switches and reads the checkpoints instead.
"""
import sys
import time
from pathlib import Path

from . import weights
from .flow import FlowNet
from .flowhighsr import CKPT_FILES, read_checkpoints, weights_bf16x6
from .vocoder import Vocoder


def build_store(sd, cfg, device="cpu", record=True):
    """Run both constructors through one recording store (the order of the keys is the file order)."""
    store = weights.WeightStore(device, record=record)
    FlowNet(sd, device, store=store)
    Vocoder(cfg, sd, device, store=store)
    return store


def convert(ckpt_dir, blob=None):
    ckpt_dir = Path(ckpt_dir)
    blob = Path(blob) if blob else ckpt_dir / weights.BLOB_NAME
    t0 = time.time()
    sd, cfg = read_checkpoints(ckpt_dir)
    t1 = time.time()
    store = build_store(sd, cfg)
    t2 = time.time()
    nbytes = store.save(blob, cfg, weights.format_tag(weights_bf16x6()), {f: weights.file_digest(ckpt_dir / f) for f in CKPT_FILES})
    t3 = time.time()
    return dict(blob=str(blob), bytes=nbytes, tensors=len(store.items), read_s=t1 - t0, pack_s=t2 - t1, write_s=t3 - t2)


if __name__ == "__main__":
    if len(sys.argv) < 2:
        sys.exit(__doc__)
    r = convert(*sys.argv[1:3])
    print(f"{r['blob']}: {r['tensors']} tensors, {r['bytes'] / 2 ** 20:.1f} MiB "
          f"(checkpoints read in {r['read_s']:.1f} s, packed in {r['pack_s']:.1f} s, written in {r['write_s']:.1f} s)")
