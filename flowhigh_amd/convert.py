"""`python -m flowhigh_amd.convert <ckpt_dir> [<blob>] [--form=winograd|bf16x6|direct] [--probe] [--verify]`: pack the reference's checkpoint files once into the flat weight blob
that `FlowHighSR.from_local` maps and uploads with one copy (flowhigh_amd/weights.py).

Runs on the CPU (no GPU, no HIP library): the packers are plain torch.  The blob is tied to the content of the three source
files, to the conv form it was packed for (--form, else FH_CONV_FORM / the default form; --probe: the form the load-time probe of
conv_form='auto' chooses on this GPU box) and to the packers' source: from_local ignores a blob that was made from other files,
for another form or by another version, and reads the checkpoints instead.  --verify re-reads a blob against its sources.
"""
import sys
import time
from pathlib import Path

from . import weights
from .flow import FlowNet
from .flowhighsr import CKPT_FILES, read_checkpoints, weights_conv_form
from .vocoder import Vocoder


def build_store(sd, cfg, device="cpu", record=True, conv_form=None):
    """Run both constructors through one recording store (the order of the keys is the file order)."""
    store = weights.WeightStore(device, record=record)
    from .planner import resolve_conv_form, use_gemm_bf16x6
    FlowNet(sd, device, store=store, bf=use_gemm_bf16x6(resolve_conv_form(conv_form)[0]))
    Vocoder(cfg, sd, device, store=store, conv_form=conv_form)
    return store


def convert(ckpt_dir, blob=None, conv_form=None, probe=False):
    """conv_form: None = what the environment asks for ('auto' = the default form).  probe (needs the GPU): run the load-time
    probe of conv_form='auto' first (FLowHigh.probe_conv_form) and pack the form it chooses."""
    ckpt_dir = Path(ckpt_dir)
    blob = Path(blob) if blob else ckpt_dir / weights.BLOB_NAME
    t0 = time.time()
    sd, cfg = read_checkpoints(ckpt_dir)
    t1 = time.time()
    form = weights_conv_form() if conv_form in (None, "auto") else conv_form
    if probe:
        from .flowhighsr import FLowHigh
        form = FLowHigh(sd, cfg, "cuda", conv_form="auto").conv_form
    store = build_store(sd, cfg, conv_form=form)
    t2 = time.time()
    nbytes = store.save(blob, cfg, weights.format_tag(form), {f: weights.file_digest(ckpt_dir / f) for f in CKPT_FILES})
    t3 = time.time()
    return dict(blob=str(blob), bytes=nbytes, tensors=len(store.items), read_s=t1 - t0, pack_s=t2 - t1, write_s=t3 - t2, form=form)


def verify(ckpt_dir, blob=None):
    """Re-read a blob against its sources: the checkpoint files' digests, the tensor bytes' digest, and every tensor bit for bit
    against a fresh packing of the checkpoints in the blob's own conv form.  -> list of problems (empty: the blob is good)."""
    import json
    import numpy as np
    import struct
    ckpt_dir = Path(ckpt_dir)
    blob = Path(blob) if blob else ckpt_dir / weights.BLOB_NAME
    problems = []
    with open(blob, "rb") as f:
        if f.read(len(weights.MAGIC)) != weights.MAGIC:
            return ["not a weight blob"]
        (hlen,) = struct.unpack("<Q", f.read(8))
        header = json.loads(f.read(hlen).decode())
    srcs = {f: weights.file_digest(ckpt_dir / f) for f in CKPT_FILES}
    if header["sources"] != srcs:
        problems.append("made from other checkpoint files")
    if header.get("data_digest") and weights.file_digest_region(blob, header["data_offset"]) != header["data_digest"]:
        problems.append("tensor bytes do not match the header's digest")
    form = json.loads(header["format"]).get("form")
    if header["format"] != weights.format_tag(form):
        problems.append("made by another version of the packers (format tag differs): tensors not compared")
        return problems
    sd, cfg = read_checkpoints(ckpt_dir)
    fresh = build_store(sd, cfg, conv_form=form)
    mm = np.memmap(blob, dtype=np.uint8, mode="r", offset=header["data_offset"])
    if set(fresh.items) != set(header["tensors"]):
        problems.append(f"tensor keys differ: {sorted(set(fresh.items) ^ set(header['tensors']))[:6]} ...")
    for key, (t, _host) in fresh.items.items():
        e = header["tensors"].get(key)
        if e is not None and bytes(mm[e["offset"]:e["offset"] + e["nbytes"]]) != t.numpy().tobytes():
            problems.append(f"{key}: bytes differ from a fresh packing")
    return problems


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    flags = [a for a in sys.argv[1:] if a.startswith("--")]
    if not args:
        sys.exit(__doc__)
    if "--verify" in flags:
        bad = verify(*args[:2])
        print("\n".join(bad) if bad else "blob verified: sources, tensor digest and every tensor match a fresh packing")
        sys.exit(1 if bad else 0)
    form = next((f.split("=", 1)[1] for f in flags if f.startswith("--form=")), None)
    r = convert(*args[:2], conv_form=form, probe="--probe" in flags)
    print(f"{r['blob']}: conv form {r['form']}, {r['tensors']} tensors, {r['bytes'] / 2 ** 20:.1f} MiB "
          f"(checkpoints read in {r['read_s']:.1f} s, packed in {r['pack_s']:.1f} s, written in {r['write_s']:.1f} s)")
