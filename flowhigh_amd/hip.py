"""ctypes binding of libflowhigh_hip.so (the C ABI declared in include/flowhigh_hip.h).

There is NO fallback: if the library is missing or a call fails this module raises.  Device
pointers are passed as integers (`tensor.data_ptr()`), the stream as
`torch.cuda.current_stream().cuda_stream`.
"""
import contextlib
import ctypes as C
import functools
import inspect
import os
from pathlib import Path

import torch

# FH_LIB_PATH: load another build of the same ABI (used for A/B experiments only)
LIB_PATH = Path(os.environ.get("FH_LIB_PATH", Path(__file__).resolve().parent / "lib" / "libflowhigh_hip.so"))

CONV_MAX_TAPS = 16
CONV_MAX_SEG = 3
CONV_MAX_HALO = 64

ABI_VERSION = 5            # FH_ABI_VERSION of include/flowhigh_hip.h
EPI_LINEAR, EPI_GEGLU, EPI_MAG, EPI_LOGCLAMP = 0, 1, 2, 3


class ConvSeg(C.Structure):
    _fields_ = [("x", C.c_void_p), ("w", C.c_void_p), ("cin", C.c_int32), ("ntaps", C.c_int32),
                ("off_min", C.c_int32), ("off_max", C.c_int32), ("tap_off", C.c_int32 * CONV_MAX_TAPS)]


class ConvGroup(C.Structure):
    _fields_ = [("seg", ConvSeg * CONV_MAX_SEG), ("bias", C.c_void_p), ("res", C.c_void_p * CONV_MAX_SEG),
                ("out", C.c_void_p), ("nseg", C.c_int32), ("nres", C.c_int32), ("cout", C.c_int32),
                ("cout_pad", C.c_int32), ("lin", C.c_int32), ("lout", C.c_int32), ("n_len", C.c_int32),
                ("out_stride", C.c_int32), ("out_phase", C.c_int32), ("scale", C.c_float)]


class WinoSeg(C.Structure):
    _fields_ = [("x", C.c_void_p), ("u", C.c_void_p), ("cin", C.c_int32), ("ngrp", C.c_int32),
                ("center", C.c_int32), ("xlen", C.c_int32)]


class WinoGroup(C.Structure):
    _fields_ = [("seg", WinoSeg * CONV_MAX_SEG), ("bias", C.c_void_p), ("res", C.c_void_p * CONV_MAX_SEG),
                ("out", C.c_void_p), ("nseg", C.c_int32), ("nres", C.c_int32), ("cout", C.c_int32),
                ("cout_pad", C.c_int32), ("len", C.c_int32), ("scale", C.c_float), ("out_stride", C.c_int32),
                ("out_phase", C.c_int32), ("out_len", C.c_int32), ("pad_", C.c_int32)]


class ActGroup(C.Structure):
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("alpha", C.c_void_p), ("inv_beta", C.c_void_p),
                ("up_taps", C.c_float * 12), ("down_taps", C.c_float * 12), ("len", C.c_int32), ("tile_base", C.c_int32)]


class AmpSeg(C.Structure):
    _fields_ = [("x", C.c_void_p), ("u", C.c_void_p), ("ngrp", C.c_int32), ("center", C.c_int32)]


class AmpGroup(C.Structure):
    _fields_ = [("seg", AmpSeg * CONV_MAX_SEG), ("bias", C.c_void_p), ("res", C.c_void_p * CONV_MAX_SEG),
                ("out", C.c_void_p), ("nseg", C.c_int32), ("nres", C.c_int32), ("len", C.c_int32),
                ("pad0_", C.c_int32), ("scale", C.c_float), ("pad1_", C.c_int32)]


class SumJob(C.Structure):
    _fields_ = [("src", C.c_void_p * 12), ("out", C.c_void_p), ("n", C.c_int64), ("n_src", C.c_int32), ("scale", C.c_float)]


class HipError(RuntimeError):
    pass


_P, _I, _F = C.c_void_p, C.c_int, C.c_float
_SIGS = {
    "fh_abi_version": [],
    "fh_sizeof_conv_group": [],
    "fh_sizeof_act_group": [],
    "fh_conv_tile_m": [_I],
    "fh_conv_tile_n": [_I],
    "fh_conv_grouped_f32": [_P, _I, _I, _I, _I, _I, _I, _P],
    "fh_conv_transpose_fused_f32": [_P, _I, _I, _I, _I, _I, _I, _P],
    "fh_sizeof_wino_group": [],
    "fh_wino_tile_m": [_I],
    "fh_phase_len": [_I, _I],
    "fh_conv_wino_f32": [_P, _I, _I, _I, _I, _I, _I, _I, _P],
    "fh_wino_tile_n": [_I],
    "fh_wino_run_len": [_I],
    "fh_conv_wino_ragged_f32": [_P, _I, _I, _I, _I, _I, _I, _P, _I, _P],
    "fh_wino54_tile_m": [_I],
    "fh_wino54_tile_n": [],
    "fh_conv_wino54_f32": [_P, _I, _I, _I, _I, _I, _I, _I, _P],
    "fh_wino54_n_tiles": [_I, _I, _I],
    "fh_wino54_run_len": [_I],
    "fh_conv_wino54_ragged_f32": [_P, _I, _I, _I, _I, _I, _I, _P, _I, _P],
    "fh_mean_f32": [_P, _P, _P, _P, C.c_longlong, _F, _P],
    "fh_sum_f32": [_P, C.c_int, _P, C.c_longlong, _F, _P],
    "fh_conv_post_tanh_f32": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "fh_act1d_grouped_f32": [_P, _I, _I, _I, _I, _P],
    "fh_act1d_grouped_pm_f32": [_P, _I, _I, _I, _I, _I, _I, _P],
    "fh_act_tile_len": [],
    "fh_act_set_blocks_per_cu": [_I],
    "fh_act_get_blocks_per_cu": [],
    "fh_act1d_ragged_f32": [_P, _I, _I, _I, _I, C.c_longlong, _I, _P],
    "fh_sizeof_amp_group": [],
    "fh_amp_tile_len": [_I],
    "fh_amp_max_channels": [],
    "fh_sizeof_amp_tile": [],
    "fh_amp_actconv_f32": [_P, _I, _P, _I, _I, _I, _I, _I, _P],
    "fh_narrow_tile_len": [],
    "fh_narrow_conv_bf16x6_f32": [_P, _I, _P, _I, _I, _I, _I, _P],
    "fh_sizeof_sum_job": [],
    "fh_sum_multi_f32": [_P, _I, C.c_longlong, _P],
    "fh_attention_seg_f32": [_P, _P, _P, _I, _I, _I, _F, _P],
    "fh_dwconv_gelu_res_seg_f32": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "fh_qknorm_rope_seg_f32": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "fh_gemm_f32": [_P, _I, _P, _P, _P, _I, _P, _I, _I, _I, _I, _F, _I, _P],
    "fh_gemm_bf16x6_f32": [_P, _I, _P, _P, _P, _I, _P, _I, _I, _I, _I, _F, _I, _P],
    "fh_gemv_f32": [_P, _P, _P, _P, _I, _I, _I, _P],
    "fh_time_fourier_f32": [_P, _F, _P, _I, _P],
    "fh_dwconv_gelu_res_f32": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "fh_rmsnorm_f32": [_P, _P, _P, _P, _I, _I, _P],
    "fh_qknorm_rope_f32": [_P, _P, _P, _P, _P, _I, _I, _I, _P],
    "fh_attention_f32": [_P, _P, _I, _I, _I, _F, _P],
    "fh_rfft2048_f32": [_P, _P, _P, _I, _I, _P],
    "fh_irfft2048_f32": [_P, _P, _P, _I, _P],
    "fh_frame_f32": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "fh_spec_energy_f32": [_P, _P, _I, _I, _P],
    "fh_cutoff_index_f32": [_P, _P, _I, _I, _F, _P],
    "fh_mel_energy_f32": [_P, _P, _I, _I, _I, _P],
    "fh_mel_splice_f32": [_P, _P, _P, _P, _I, _I, _I, _P],
    "fh_mel_energy_seg_f32": [_P, _P, _P, _I, _I, _P],
    "fh_mel_splice_seg_f32": [_P, _P, _P, _P, _P, _I, _I, _I, _P],
    "fh_axpby_f32": [_P, _F, _P, _F, _P, C.c_longlong, _P],
    "fh_spec_splice_f32": [_P, _P, _P, _P, _I, _I, _P],
    "fh_istft_ola_f32": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "fh_peak_scale_f32": [_P, _P, _I, _I, _F, _P],
    "fh_peak_abs_f32": [_P, _P, _I, _I, _P],
    "fh_resample_poly_f32": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
}
EXPORTS = sorted(_SIGS) + ["fh_last_error"]

_lib = None


def lib():
    """Load (once) and return the ctypes library; raises HipError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise HipError(f"{LIB_PATH} is missing: build it with `python -m flowhigh_amd.build` "
                       "(there is no CPU fallback)")
    L = C.CDLL(str(LIB_PATH))
    for name, args in _SIGS.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = C.c_int
    L.fh_last_error.argtypes = []
    L.fh_last_error.restype = C.c_char_p
    if L.fh_abi_version() != ABI_VERSION:
        raise HipError("libflowhigh_hip.so ABI version mismatch")
    if L.fh_sizeof_conv_group() != C.sizeof(ConvGroup) or L.fh_sizeof_act_group() != C.sizeof(ActGroup) \
            or L.fh_sizeof_wino_group() != C.sizeof(WinoGroup) or L.fh_sizeof_sum_job() != C.sizeof(SumJob) \
            or L.fh_sizeof_amp_group() != C.sizeof(AmpGroup):
        raise HipError("descriptor struct layout mismatch between hip.py and flowhigh_hip.h")
    _lib = L
    return L


def _tensor_bytes(obj, seen):
    """Device bytes held by the tensors reachable from a workspace / plan (containers and plain objects)."""
    if isinstance(obj, torch.Tensor):
        st = obj.untyped_storage()
        if st.data_ptr() in seen:
            return 0
        seen.add(st.data_ptr())
        return st.nbytes()
    if isinstance(obj, (str, bytes, int, float, bool, type(None), type)) or ("id", id(obj)) in seen:
        return 0
    seen.add(("id", id(obj)))                   # (cycles, shared sub-objects)
    if isinstance(obj, dict):
        return sum(_tensor_bytes(v, seen) for v in obj.values())
    if isinstance(obj, (list, tuple, set)):
        return sum(_tensor_bytes(v, seen) for v in obj)
    if hasattr(obj, "__dict__"):
        return sum(_tensor_bytes(v, seen) for v in vars(obj).values())
    return 0


class ShapeCache(dict):
    """Per-shape workspaces / launch plans, bounded by bytes and by count: a service sees many clip lengths
    (a 1 s clip holds ~65 MB of vocoder workspace, a 10 s clip 645 MB, a batch of 32 of them 20 GB) and a
    rebuilt plan costs 10-20 ms.  Evicts least recently used entries; an eviction waits for the device first
    (batches of different shapes may be in flight on several streams, generate_many), then the memory returns
    to torch's caching allocator.  FH_CACHE_GB (default 24) bounds the bytes of EACH cache."""

    def __init__(self, max_entries=64, max_bytes=None):
        super().__init__()
        self.max_entries = max_entries
        self.max_bytes = int(float(os.environ.get("FH_CACHE_GB", "24")) * 2 ** 30) if max_bytes is None else max_bytes
        self._bytes = {}

    def __getitem__(self, key):
        v = super().pop(key)
        super().__setitem__(key, v)            # most recently used last
        return v

    def put(self, key, value, nbytes):
        """Insert with a known size (entries that only reference buffers owned by other entries)."""
        self._insert(key, value, int(nbytes))

    def __setitem__(self, key, value):
        self._insert(key, value, _tensor_bytes(value, set()))

    def _insert(self, key, value, nbytes):
        if key in self:
            super().pop(key)
        self._bytes[key] = nbytes
        total = sum(self._bytes[k] for k in self) + self._bytes[key]
        synced = False
        while len(self) and (len(self) >= self.max_entries or total > self.max_bytes):
            if not synced and torch.cuda.is_available():
                torch.cuda.synchronize()
                synced = True
            old = next(iter(self))
            total -= self._bytes.pop(old)
            super().pop(old)
        super().__setitem__(key, value)


def check(rc, what=""):
    if rc != 0:
        raise HipError(f"{what}: rc={rc}: {lib().fh_last_error().decode()}")


def stream(device=None):
    """Raw hipStream_t of torch's current stream on `device` (default: the current device; inside an
    @on_device method that IS the model's device)."""
    return torch.cuda.current_stream(device).cuda_stream


def norm_device(device):
    """torch.device with an explicit ordinal: 'cuda' means the device that is current NOW (at construction), so
    that a model keeps launching there whatever the caller makes current later (the reference's
    `from_local(ckpt_dir, device)`, flowhighsr.py:110-137)."""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None and torch.cuda.is_available():
        device = torch.device("cuda", torch.cuda.current_device())
    return device


def device_guard(device):
    """Context manager making `device` the current HIP device (no-op for the CPU-device objects the host-logic
    tests build: they plan launches but never enqueue one)."""
    device = torch.device(device)
    return torch.cuda.device(device) if device.type == "cuda" else contextlib.nullcontext()


def on_device(fn):
    """Method decorator: run with `self.device` as the current HIP device, so that every allocation, every
    `hip.stream()` and every kernel launch inside goes to the model's device and its current stream, not to
    whatever device the calling thread has current.  Generator methods are guarded per resumption."""
    if inspect.isgeneratorfunction(fn):
        @functools.wraps(fn)
        def gen_wrapper(self, *args, **kwargs):
            gen = fn(self, *args, **kwargs)
            while True:
                with device_guard(self.device):
                    try:
                        item = next(gen)
                    except StopIteration:
                        return
                yield item
        return gen_wrapper

    @functools.wraps(fn)
    def wrapper(self, *args, **kwargs):
        with device_guard(self.device):
            return fn(self, *args, **kwargs)
    return wrapper


def ptr(t):
    return 0 if t is None else t.data_ptr()


def to_device_struct_array(structs, device, tail_bytes=0):
    """ctypes Structure list -> device uint8 tensor holding the packed array (+ tail_bytes zeroed bytes behind it)."""
    if not structs:
        raise ValueError("empty descriptor list")
    arr = (type(structs[0]) * len(structs))(*structs)
    host = torch.frombuffer(bytearray(bytes(arr) + bytes(tail_bytes)), dtype=torch.uint8)
    return host.to(device)


# ---- thin typed wrappers (all enqueue-only) --------------------------------------------------
def _f32c(t, name):
    if t is None:
        return
    if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
        raise HipError(f"{name}: expected a contiguous float32 CUDA tensor, got {t.dtype} "
                       f"contiguous={t.is_contiguous()} device={t.device}")


def gemm(A, W, C_out, M, N, K, *, bias=None, R=None, alpha=1.0, epilogue=EPI_LINEAR, lda=None,
         ldc=None, ldr=None, bf=False):
    """C[M, N or N/2] = epi(A[M,K] @ W[N,K]^T).  W must have rows padded to a multiple of 128.
    bf: the bf16 x 6 form (fh_gemm_bf16x6_f32): W = packing.pack_gemm_bf_weight of such a matrix (6 bytes per weight)."""
    for t, n in ((A, "A"), (W, "W"), (C_out, "C"), (bias, "bias"), (R, "R")):
        _f32c(t, n)
    if bf:
        n_pad = -(-N // 128) * 128
        if W.numel() * 4 != n_pad * K * 6:
            raise HipError(f"gemm: packed bf16 x 6 W of {W.numel() * 4} bytes does not fit [{n_pad}, {K}] x 6 bytes")
    elif W.shape[0] % 128 or W.shape[1] != K:
        raise HipError(f"gemm: W must be [n_pad % 128 == 0, K], got {tuple(W.shape)} K={K}")
    lda = lda if lda is not None else A.shape[-1]
    out_w = N // 2 if epilogue in (EPI_GEGLU, EPI_MAG) else N
    ldc = ldc if ldc is not None else C_out.shape[-1]
    ldr = ldr if ldr is not None else (R.shape[-1] if R is not None else 0)
    if ldc < out_w:
        raise HipError("gemm: output row too short")
    fn = lib().fh_gemm_bf16x6_f32 if bf else lib().fh_gemm_f32
    check(fn(ptr(A), lda, ptr(W), ptr(bias), ptr(R), ldr, ptr(C_out), ldc, M, N, K,
             float(alpha), epilogue, stream()), "fh_gemm_bf16x6_f32" if bf else "fh_gemm_f32")
    return C_out
