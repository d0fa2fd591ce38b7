"""Packed device weights as ONE flat file: `python -m flowhigh_amd.convert <ckpt_dir>` writes every tensor the kernels read
(weight-norm folded, channel-padded, Winograd-transformed in float64, fragment layouts of include/flowhigh_hip.h) once;
`FlowHighSR.from_local` then maps the file and uploads it with one host-to-device copy instead of unpickling 613 MB of
checkpoints and redoing the transforms of 118 M parameters in every process.

Replaces the load half of /root/reference/src/flowhigh/flowhighsr.py:110-149 (from_local: torch.load x 2 + load_state_dict +
.cuda()) and models/bigvgan/init_vocoder.py:8-23 (remove_weight_norm at load).

File: b"FHBLOB1\\n" | u64 header bytes | header (JSON, utf-8) | zero padding to 4096 | tensor bytes, each 256-byte aligned.
Header: format tag (every switch the packed layouts depend on), digests of the source files, the vocoder JSON, and per tensor
dtype / shape / offset / whether it stays on the host (filter taps and the like).
"""
import hashlib
import json
import os
import struct
from pathlib import Path

import numpy as np
import torch

MAGIC = b"FHBLOB1\n"
BLOB_NAME = "flowhigh_amd_gfx950.blob"
_ALIGN = 256
_DTYPES = {"float32": torch.float32, "float64": torch.float64, "int16": torch.int16, "int32": torch.int32}


def format_tag(form="winograd"):
    """Everything that decides WHICH layouts the constructors pack: a blob made under other decisions is not used.
    form: the RESOLVED conv form ('winograd' | 'bf16x6' | 'direct', planner.resolve_conv_form); the remaining layout switches as
    normalised booleans / numbers (unset and "1" are the same setting), and a fingerprint of the packers' and the planner's source
    (tile choices, thresholds and fragment layouts live there: a code change must not silently reuse a key with another layout)."""
    from . import hip
    pkg = Path(__file__).resolve().parent
    fp = hashlib.blake2b(digest_size=8)
    for name in ("packing.py", "planner.py", "vocoder.py", "flow.py"):
        fp.update((pkg / name).read_bytes())
    on = lambda k: os.environ.get(k, "1") != "0"
    sw = dict(wino54=on("FH_WINO54"), wino54_h16=on("FH_WINO54_H16"), amp=on("FH_AMP"))
    return json.dumps(dict(abi=hip.ABI_VERSION, layout=4, form=str(form), switches=sw, packers=fp.hexdigest()), sort_keys=True)


def file_digest(path):
    """Content digest of a source file (xxh3 when the module is there: ~5 GB/s; else blake2b)."""
    try:
        import xxhash
        h = xxhash.xxh3_128()
    except ImportError:
        h = hashlib.blake2b(digest_size=16)
    with open(path, "rb") as f:
        while True:
            b = f.read(1 << 24)
            if not b:
                break
            h.update(b)
    return h.hexdigest()


def file_digest_region(path, offset):
    """Digest of a file from `offset` to its end (the tensor bytes of a blob: header['data_digest'], checked with FH_BLOB_VERIFY=2
    and by `python -m flowhigh_amd.convert --verify`)."""
    h = hashlib.blake2b(digest_size=16)
    with open(path, "rb") as f:
        f.seek(offset)
        while True:
            b = f.read(1 << 24)
            if not b:
                break
            h.update(b)
    return h.hexdigest()


class WeightStore:
    """Where the constructors get their device tensors from.
    Build mode (default): `dev(key, fn)` calls fn() -> CPU tensor and uploads it (and keeps the CPU tensor when recording, for
    save()).  Blob mode (WeightStore.open): returns the view of the uploaded file that belongs to `key`; fn is not called, the
    checkpoint is not needed.  `host(key, fn)`: the same for small tensors that stay on the host."""

    def __init__(self, device, record=False):
        self.device = torch.device(device)
        self.record = record
        self.items = {}             # key -> (cpu tensor, host flag), insertion order = file order
        self._blob = None
        self.cfg = None
        self.form = None            # (blob mode) the conv form the blob was packed for

    # ---- build mode / common -----------------------------------------------------------------------------------------
    def dev(self, key, fn):
        if self._blob is not None:
            return self._view(key, False)
        t = fn().contiguous()
        self._remember(key, t, False)
        return t.to(self.device)

    def host(self, key, fn):
        if self._blob is not None:
            return self._view(key, True)
        t = fn().contiguous()
        self._remember(key, t, True)
        return t

    def _remember(self, key, t, host):
        if key in self.items:
            raise KeyError(f"weight key {key!r} used twice")
        self.items[key] = (t if self.record else None, host)

    # ---- blob mode -----------------------------------------------------------------------------------------------------
    def _view(self, key, host):
        e = self._blob["tensors"].get(key)
        if e is None or bool(e["host"]) != host:
            raise KeyError(f"{key!r} is not in the weight blob (made by another version of the package? re-run flowhigh_amd.convert)")
        src = self._blob["host"] if host else self._blob["device"]
        off = e["offset"] - (0 if host else self._blob["dev_base"])
        t = src[off:off + e["nbytes"]].view(_DTYPES[e["dtype"]]).view(e["shape"])
        return t.clone() if host else t

    @classmethod
    def open(cls, path, device, expect_format=None, sources=None):
        """Map `path` and upload its device part; None (with the reason in .why on the class) if the file is not a blob of
        this format or was made from other source files."""
        path = Path(path)
        cls.why = None
        try:
            with open(path, "rb") as f:
                if f.read(len(MAGIC)) != MAGIC:
                    cls.why = "not a weight blob"
                    return None
                (hlen,) = struct.unpack("<Q", f.read(8))
                header = json.loads(f.read(hlen).decode())
                need = ("format", "sources", "cfg", "tensors", "host_offset", "data_offset")
                if not isinstance(header, dict) or any(k not in header for k in need):
                    cls.why = "header incomplete"
                    return None
        except (OSError, ValueError, struct.error) as e:
            cls.why = str(e) or type(e).__name__
            return None
        if expect_format is not None and header["format"] != expect_format:
            cls.why = "made under other layout switches"
            return None
        if sources is not None and header["sources"] != sources:
            cls.why = "made from other checkpoint files"
            return None
        data0 = header["data_offset"]
        # a truncated or partly copied file must not open: every tensor has to lie inside it
        try:
            end = max([e["offset"] + e["nbytes"] for e in header["tensors"].values()], default=0)
            size = path.stat().st_size
        except (OSError, KeyError, TypeError) as e:
            cls.why = f"tensor index unreadable ({type(e).__name__})"
            return None
        if size < data0 + end:
            cls.why = f"truncated: {size} bytes, the tensor index needs {data0 + end}"
            return None
        if os.environ.get("FH_BLOB_VERIFY", "1") == "2" and header.get("data_digest"):
            if file_digest_region(path, data0) != header["data_digest"]:
                cls.why = "tensor bytes do not match the header's digest"
                return None
        mm = np.memmap(path, dtype=np.uint8, mode="c", offset=data0)         # (copy-on-write: torch wants a writable array)
        self = cls(device)
        self.cfg = header["cfg"]
        try:
            self.form = json.loads(header["format"]).get("form")      # the conv form the tensors were packed for
        except (ValueError, AttributeError):
            self.form = None
        split = header["host_offset"]                    # [0, split): device tensors, [split, end): host tensors
        whole = torch.from_numpy(np.asarray(mm))         # zero-copy view of the mapping
        dev_part = whole[:split]
        if self.device.type == "cuda":
            dev_t = torch.empty(split, dtype=torch.uint8, device=self.device)
            dev_t.copy_(dev_part)                        # ONE host-to-device copy of the whole file
        else:
            dev_t = dev_part.clone()
        self._blob = dict(tensors=header["tensors"], device=dev_t, dev_base=0, host=whole[split:].clone().contiguous(), mm=mm)
        # host offsets are stored relative to the file's data region: rebase them on the host slice
        for e in self._blob["tensors"].values():
            if e["host"]:
                e["offset"] -= split
        return self

    def save(self, path, cfg, fmt, sources):
        """Write the recorded tensors (build mode with record=True): device tensors first, then the host ones."""
        if not self.record:
            raise RuntimeError("WeightStore.save needs record=True")
        tensors, off, chunks = {}, 0, []
        order = [k for k, (_, h) in self.items.items() if not h] + [k for k, (_, h) in self.items.items() if h]
        host_offset = None
        for key in order:
            t, host = self.items[key]
            if host and host_offset is None:
                host_offset = off
            raw = t.numpy().tobytes()
            name = str(t.dtype).replace("torch.", "")
            if name not in _DTYPES:
                raise TypeError(f"{key}: dtype {t.dtype} is not storable")
            tensors[key] = dict(dtype=name, shape=list(t.shape), offset=off, nbytes=len(raw), host=host)
            pad = -len(raw) % _ALIGN
            chunks.append(raw + bytes(pad))
            off += len(raw) + pad
        if host_offset is None:
            host_offset = off
        dg = hashlib.blake2b(digest_size=16)
        for c in chunks:
            dg.update(c)
        header = dict(format=fmt, sources=sources, cfg=cfg, tensors=tensors, host_offset=host_offset, data_offset=0,
                      data_digest=dg.hexdigest())
        for _ in range(2):                                # (data_offset depends on the header's own length)
            h = json.dumps(header).encode()
            header["data_offset"] = -(-(len(MAGIC) + 8 + len(h) + 32) // 4096) * 4096
        h = json.dumps(header).encode()
        tmp = Path(str(path) + ".tmp")
        with open(tmp, "wb") as f:
            f.write(MAGIC + struct.pack("<Q", len(h)) + h)
            f.write(bytes(header["data_offset"] - f.tell()))
            for c in chunks:
                f.write(c)
        os.replace(tmp, path)
        return off
