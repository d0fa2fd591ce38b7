"""Launch planning of the BigVGAN generator: which kernel and tile shape a conv position runs on (per STAGE, never per
length or batch), the launch-time model that ranks tile shapes, descriptor builders, and `_PlanBuilder`, which turns one
[batch, num_mels, n_frames] shape into the ordered list of launches `runtime.launch_step` enqueues.
(Split out of vocoder.py in round 5; `flowhigh_amd.vocoder` re-exports every name.)

Mirrors /root/reference/src/flowhigh/models/bigvgan/models.py:124-194 (BigVGAN.forward), AMPBlock1 :21-78, AMPBlock2 :81-121.
"""
import ctypes as C
import functools
import os

import torch

from . import hip
from .packing import phase_len

# preference order among equal padded heights: tiles that keep 3 blocks per CU resident first
_TILE_PREF = [(0, 128), (6, 96), (3, 64), (1, 192), (2, 96), (4, 32)]

# Residual-stack convs run as Winograd F(4,3) from this many channels on, and at the narrower widths where it measured
# faster (tools/wino_bench.py, B = 1; the dilated ones work on phase-major tensors)
WINO_MIN_C = 192
WINO_NARROW = (96, 48)
# transposed convs run as Winograd phase groups from this many INPUT channels on (366 -> 207 us for 1536 -> 768
# channels at B = 1, a wash at 768 -> 384, slower below: one-tap-group blocks pay the per-chunk slab cost every step)
WINO_UPS_MIN_CIN = 768
# ... in the bf16 x 6 form from 192 on: the F(4,3) bf16 x 6 phase groups take 157 us against the direct fp32 kernel's 215 at 384 -> 192
# channels, 97 against 105 at 192 -> 96, 77 against 72 at 96 -> 48 (tools/exp/ups_wino_threshold.py, B = 1, 10 s clip)
WINO_UPS_MIN_CIN_BF = 192


def wino_ups_min_cin(form):
    return WINO_UPS_MIN_CIN_BF if form == "bf16x6" else WINO_UPS_MIN_CIN


# ---- which arithmetic form the vocoder's convs run in (round 6: a constructor keyword of the drop-in class) ---------------------
#   'winograd'  Winograd F(5,4) / F(4,3) on the fp32 matrix instructions (v_mfma_f32_32x32x2_f32): the default of rounds 3-5
#   'bf16x6'    the same Winograd contractions on the BF16 matrix cores with every fp32 operand split EXACTLY into three bf16
#               pieces and six piece-pair MFMAs per k-block (fp32 in / out / accumulate, fp32-grade products: the dropped pairs are
#               <= 2^-24 |a b|); the narrow stages as a direct conv in the same arithmetic (narrow_bf.hip), the transformer's linears too
#               (gemm_bf.hip); the last two upsamplers stay on the fp32 matrix instructions
#   'direct'    no Winograd anywhere: the direct implicit-GEMM kernel (fp32 MFMA), ~20 % closer to a float64 run where the
#               weights' gain is high (profiles/r05_regime_sweep.txt) and ~2 x slower
#   'auto'      DEFAULT_CONV_FORM, unless a load-time probe through the loaded weights (Vocoder.probe_conv_form: the default
#               form against the direct form on a 20-frame random mel) differs by more than PROBE_LIMIT: then 'direct'
# Keyword > environment (FH_CONV_FORM; the older switches FH_WINO=0 -> direct, FH_CONV_BF16X6=1 / 0 -> bf16x6 / winograd) > 'auto'.
CONV_FORMS = ("auto", "winograd", "bf16x6", "direct")
DEFAULT_CONV_FORM = "bf16x6"
PROBE_LIMIT = 3e-5


def resolve_conv_form(conv_form=None, bf16x6=None):
    """-> (form, from_auto): one of 'winograd' / 'bf16x6' / 'direct', and whether it came from 'auto' (a probe may still change
    it).  conv_form: the keyword (None: ask the environment); bf16x6: the older boolean keyword (True / False / None)."""
    form = conv_form
    if form is None and bf16x6 is not None:
        form = "bf16x6" if bf16x6 else "winograd"
    if form is None:
        form = os.environ.get("FH_CONV_FORM", "").strip().lower() or None
    if form is None and os.environ.get("FH_WINO", "1") == "0":
        form = "direct"
    if form is None and os.environ.get("FH_CONV_BF16X6", "") in ("0", "1"):
        form = "bf16x6" if os.environ["FH_CONV_BF16X6"] == "1" else "winograd"
    if form is None:
        form = "auto"
    if form not in CONV_FORMS:
        raise ValueError(f"conv_form must be one of {CONV_FORMS}, got {form!r}")
    return (DEFAULT_CONV_FORM, True) if form == "auto" else (form, False)


def _wino_on(form):
    """Winograd kernels allowed?  form None: the environment (FH_WINO=0 / FH_CONV_FORM=direct switch them off)."""
    return (resolve_conv_form()[0] if form is None else form) != "direct"


def use_wino(c, d, form=None):
    """Residual-stack convs [c -> c, dilation d] that run as Winograd F(4,3) (conv_wino.hip) instead of the direct
    implicit GEMM: c a multiple of 16 and >= WINO_MIN_C or one of WINO_NARROW, at every dilation (a dilated conv works
    on phase-major tensors written / read by the neighbouring activation launches: contiguous runs instead of stride-d
    access).  form 'direct' (FH_WINO=0) switches the path off (direct kernel everywhere)."""
    return _wino_on(form) and c % 16 == 0 and (c >= WINO_MIN_C or c in WINO_NARROW)


def wino_conv_ok(c, d, k, form=None):
    """use_wino for a k-tap conv: kernels longer than 4 tap groups run on the direct kernel."""
    return use_wino(c, d, form) and k <= WINO_MAX_K


def pick_tile_cfg(cout):
    best = None
    for cfg, bm in _TILE_PREF:
        cost = -(-cout // bm) * bm
        if best is None or cost < best[0]:
            best = (cost, cfg, bm)
    return best[1], best[2], best[0]


def make_conv_seg(x, w, cin, offs):
    s = hip.ConvSeg()
    s.x, s.w, s.cin, s.ntaps = hip.ptr(x), hip.ptr(w), cin, len(offs)
    if len(offs) > hip.CONV_MAX_TAPS or max(offs) - min(offs) > hip.CONV_MAX_HALO:
        raise NotImplementedError(f"tap list {offs} exceeds kernel limits")
    s.off_min, s.off_max = min(offs), max(offs)
    for i, o in enumerate(offs):
        s.tap_off[i] = o
    return s

def make_conv_group(segs, bias, res, out, cout, cpad, lin, lout, n_len, stride=1, phase=0, scale=1.0):
    g = hip.ConvGroup()
    for i, s in enumerate(segs):
        g.seg[i] = s
    g.nseg, g.nres = len(segs), len(res)
    g.bias = hip.ptr(bias)
    for i, r in enumerate(res):
        g.res[i] = hip.ptr(r)
    g.out = hip.ptr(out)
    if cout * lout * 4 >= 2 ** 31 or lin * 4 >= 2 ** 31:
        raise NotImplementedError("per-clip tensor exceeds the 2 GiB range of a buffer descriptor")
    g.cout, g.cout_pad, g.lin, g.lout, g.n_len = cout, cpad, lin, lout, n_len
    g.out_stride, g.out_phase, g.scale = stride, phase, scale
    return g

def make_act_group(x, y, p):
    g = hip.ActGroup()
    g.x, g.y, g.alpha, g.inv_beta = hip.ptr(x), hip.ptr(y), hip.ptr(p["alpha"]), hip.ptr(p["inv_beta"])
    for i in range(12):
        g.up_taps[i] = p["up"][i]
        g.down_taps[i] = p["down"][i]
    return g



# Winograd tiles (tile_cfg -> rows x outputs) and their measured block time on one CU: _WINO_COST[cfg] = (a, b),
# a us per K step (16 input channels x one tap group), b us of prologue + epilogue (tools/wino_cfg_sweep.py)
# | WINO_F54: the F(5,4) kernel (conv_wino54.hip: 8-wave blocks of 128 / 96 / 64 co x 320 outputs; K steps = 16 input
# channels x one group of FOUR taps; constants from tools/wino54_bench.py fit)
WINO_F54 = 256            # flag in a plan's tile id: the launch runs fh_conv_wino54_f32 with tile_cfg = id & 15
_WINO_TILES = {0: (64, 512), 1: (96, 256), 4: (64, 256), 5: (32, 256), 6: (128, 256),
               WINO_F54 | 0: (128, 320), WINO_F54 | 1: (96, 320), WINO_F54 | 2: (64, 320), WINO_F54 | 3: (48, 320)}
_WINO_COST = {0: (2.98, 20.0), 1: (2.21, 16.0), 4: (1.564, 14.7), 5: (0.917, 17.0), 6: (2.75, 20.0),
              # (tools/wino54_cost_fit.py: 4.4 / 3.1-3.6 / 2.5 us per step = 0.78 / 0.71-0.83 / 0.68 of the matrix-pipe time; b: ~10 us
              # reproduces both closing-conv forms of the C = 96 stage: 3 groups, 1 125 blocks: 232 us; fused, 375 blocks: 308 us)
              # Chosen among five constant sets by the measured total of the 43 conv launches of a 10 s clip (13.25 ms; the others
              # 13.35-13.41): the launch model is a ranking device, not a clock.
              # (| 3: the 48-row block, three 16-row MFMA tiles: 3/4 of the 64-row block's matrix instructions)
              WINO_F54 | 0: (4.4, 12.0), WINO_F54 | 1: (3.3, 10.0), WINO_F54 | 2: (2.52, 9.0), WINO_F54 | 3: (1.95, 9.0)}
# 128-row tiles halve the LDS reads and transform instructions per MFMA (one B fragment feeds 4 MFMAs) but
# double the weight bytes a block streams: beyond this panel size (6 x 128 rows x K, bytes) a chip full of
# such blocks thrashes the 4 MB L2s (C = 768: 857 us against 732 us with 64 x 512 tiles)
_WINO_WIDE_PANEL_MAX = 6 * 2 ** 20
_WINO_TILES_OFF = set()            # (A/B experiments: tools add tile ids here before the first plan)
_WINO_RUN = 8                      # W_RUN of conv_wino.hip


# K-loop time per step of the bf16 x 6 form relative to the fp32 form, per tile shape (tools/wino_cost_fit.py:
# 2.47 / 1.72 / 1.34 / 1.08 / 2.43 us against 3.28 / 2.27 / 1.86 / 1.14 / 3.22): the matrix pipe needs 0.375 of the
# cycles, but the B split (36 VALU instructions per 8 values), 1.5 x the weight bytes and the unchanged slab / transform
# work keep the loop issue-bound (ablations: no weight loads -10 %, no split -10 %, neither and no slab -35 %)
_WINO_BF_SPEED = {0: 0.75, 1: 0.76, 4: 0.72, 5: 0.94, 6: 0.75,
                  # F(5,4) bf16 x 6 blocks (round 6; 96- and 64-row blocks: the 128-row one spills): tools/wino54_cost_fit.py ... bf
                  # (round 6, tools/wino54_cost_fit.py bf: 2.20 / 1.80-1.93 us per K step against 2.87-3.13 / 2.44-2.58)
                  WINO_F54 | 1: 0.72, WINO_F54 | 2: 0.74}
# F(5,4) tiles that have a bf16 x 6 form the planner may pick
_WINO54_BF_TILES = (WINO_F54 | 1, WINO_F54 | 2)


def wino_n_tiles(cfg, length, dil, pm):
    """Output tiles per (group, batch, co tile) panel of a Winograd launch (the kernels' launchers compute the same): per phase in
    general; the F(5,4) kernel tiles phase-major rows as one sequence of 5-output tile slots (conv_wino54.hip: v_tile_slots)."""
    bm, bt = _WINO_TILES[cfg & (15 | WINO_F54)]
    if cfg & WINO_F54 and pm:
        slots = ((-(-length // dil) + 4) // 5 + 3 + 3) & ~3
        return -(-dil * slots // 64)
    return -(-(-(-length // dil)) // bt) * dil


def wino_launch_cost(ksteps, batch, wpad, length, dil, cfg, cus_per_xcd=32, bf=False):
    """Estimated duration (us) of one fh_conv_wino_f32 launch: the kernel's block -> (panel, tile) map replayed
    on 8 XCDs x 32 CUs with in-order dispatch per XCD (block i goes to XCD i % 8).  ksteps: K steps
    (sum over segments of cin / 16 x tap groups) of each group, launch order.  Blocks of a launch differ up to
    4 x in length (k = 11 / 7 / 3) and a 10 s clip is only 1-6 blocks per CU, so the block count per tile
    shape, not the per-tile efficiency, decides between the tile shapes (measured +-10 % at batch 1)."""
    import heapq
    bm, bt = _WINO_TILES[cfg]
    a, b = _WINO_COST[cfg]
    if bf:
        a *= _WINO_BF_SPEED[cfg]
    n_tiles = wino_n_tiles(cfg, length, dil, dil > 1)       # (dilated Winograd launches of the model are phase-major)
    cot = wpad // bm
    panel_w = [a * k + b for k in ksteps for _ in range(batch * cot)]
    run_len = -(-n_tiles // -(-n_tiles // _WINO_RUN))
    rpp = -(-n_tiles // run_len)
    real = len(panel_w) * n_tiles
    load = 1.12 if real > 200 else 1.0 + 0.12 * real / 200        # blocks run ~12 % slower on a full chip
    if cfg == 6 and real > 8 * cus_per_xcd and max(ksteps) * 16 * 6 * bm * 4 > _WINO_WIDE_PANEL_MAX:
        load *= 1.4
    if cfg == WINO_F54 and real > 8 * cus_per_xcd and max(ksteps) * 16 * 8 * bm * 4 > _WINO_WIDE_PANEL_MAX:
        load *= 1.15          # (C = 768, dilation 3: 625-635 us with 324 128-row blocks against 580 us with 864 96-row half-depth ones)
    if real > 16384:                                                  # many blocks per CU: throughput bound
        return load * sum(panel_w) * n_tiles / (8 * cus_per_xcd)
    total_runs = len(panel_w) * rpp
    end = 0.0
    for x in range(8):
        cu = [0.0] * cus_per_xcd
        for run in range(x, total_runs, 8):
            w = panel_w[run // rpp] * load
            first = (run % rpp) * run_len
            for _ in range(min(run_len, n_tiles - first)):
                heapq.heapreplace(cu, cu[0] + w)
        end = max(end, max(cu))
    return end


def choose_wino_cfg(ksteps, batch, wpad, length, dil, default=None, bf=False):
    return _choose_wino_cfg(tuple(ksteps), batch, wpad, length, dil, default, bool(bf))


@functools.lru_cache(maxsize=4096)
def _choose_wino_cfg(ksteps, batch, wpad, length, dil, default, bf=False):
    """Tile shape with the smallest estimated launch time among those the packed weights (cout_pad) allow;
    the default shape stays unless another one is estimated at least 3 % faster (the model is good to a few
    per cent; at large batch every shape is within that and the default has the best steady state)."""
    fam = WINO_F54 if (default is not None and default & WINO_F54) else 0             # tiles of the default's kernel only
    # (the 48-row F(5,4) block sums a chunk's channels in another order than the 32-row-tile blocks: it is the shape of the
    # weights it alone divides, never an alternative to the 96-row block)
    cands = [cfg for cfg, (bm, _) in _WINO_TILES.items()
             if wpad % bm == 0 and cfg not in _WINO_TILES_OFF and (cfg & WINO_F54) == fam
             and not (cfg == WINO_F54 | 3 and wpad % 96 == 0)
             and not (bf and fam and cfg not in _WINO54_BF_TILES)]
    cost = {cfg: wino_launch_cost(ksteps, batch, wpad, length, dil, cfg, bf=bf) for cfg in cands}
    best = min(cands, key=lambda cfg: cost[cfg])
    if default in cost and cost[best] > 0.97 * cost[default]:
        best = default
    return best, cost[best]
WINO_BM = 64


def pick_wino_tile(c):
    """Default (tile_cfg, cout_pad) of the Winograd kernel: 96-row tiles where they divide c and 64-row tiles do
    not (the launch plan may pick another shape that divides cout_pad: choose_wino_cfg)."""
    if c % 64 and c % 96 == 0:
        return 1, c
    return 0, -(-c // WINO_BM) * WINO_BM


def pick_wino54_tile(c, bf=False):
    """(plan tile id, cout_pad) of the F(5,4) kernel: 128-row blocks where they divide c, else 96, else 48 (three 16-row MFMA
    tiles), else 64 (padded).  bf (the bf16 x 6 form): 96-row blocks where they divide c, else 64 (padded)."""
    if bf:
        return (WINO_F54 | 1, c) if c % 96 == 0 else (WINO_F54 | 2, -(-c // 64) * 64)
    if c % 128 == 0:
        return WINO_F54 | 0, c
    if c % 96 == 0:
        return WINO_F54 | 1, c
    if c % 48 == 0:
        return WINO_F54 | 3, c
    return WINO_F54 | 2, -(-c // 64) * 64


# Residual-stack convs run in the F(5,4) form (conv_wino54.hip) from this many channels on: 20 % fewer matrix
# instructions and fewer vector / LDS instructions per MFMA than F(4,3) (tools/wino54_bench.py: x 1.15-1.3 per launch at
# 192-768 channels, x 1.1-1.2 at 96).  FH_WINO54=0 switches it off (every Winograd conv in the F(4,3) form).  The choice
# is per STAGE (channel count), never per length or batch: a clip gets the same bits alone, batched, ragged, chunked.
WINO54_MIN_C = 96


def use_wino54(c, form=None):
    """... and at the odd multiples of 48 channels below it (C = 48: the 48-row block has no padding rows, the F(4,3) kernel's
    64-row tile a quarter; FH_WINO54_H16=0: not there)."""
    if os.environ.get("FH_WINO54", "1") == "0" or not use_wino(c, 1, form):
        return False
    return c >= WINO54_MIN_C or (c % 48 == 0 and os.environ.get("FH_WINO54_H16", "1") != "0")


def use_amp(c, ks, dils, form=None):
    """Residual-stack convs of a stage that run on the narrow-stage kernel (amp_fused.hip, conv-only form: the activation stays
    a launch of its own): at most 48 channels (a multiple of 8), odd kernels of at most 11 taps, dilations of at most 6.
    A property of the STAGE (channel count and checkpoint configuration), never of the length or the batch.  FH_AMP=0 switches
    it off (those stages then run as in round 4: F(5,4) 48-row blocks / the direct kernel)."""
    if os.environ.get("FH_AMP", "1") == "0" or not _wino_on(form):
        return False
    cmax = max((k - 1) // 2 for k in ks)
    return (c % 8 == 0 and 8 <= c <= 48 and all(k % 2 == 1 and k <= 11 for k in ks)
            and all(1 <= d <= AMP_MAX_D for dl in dils for d in dl)
            and all(cmax - (k - 1) // 2 + 4 * -(-k // 4) + 3 <= 16 for k in ks))


def ups_fused_ok(st, enabled=True):
    """A stage's direct-kernel ConvTranspose1d runs with all its output phases in one block (conv_mfma.hip, PH = 2): stride 2
    with an even k - u, 16-channel chunks, one of the tile shapes that have the form, and an even number of K steps per phase
    (the kernel's two weight register sets alternate per step).  A per-stage property; FH_UPS_FUSE=0 (enabled=False): one group per
    phase with strided stores, as until round 4 (same bits).  (The stride-3 form of round 5 -- 275 us against 139 us for the
    384 -> 192 stage at batch 1, profiles/r05_upsampler_phases.txt -- left the library in round 6.)"""
    if not enabled:
        return False
    return (st["u"] == 2 and st["extra"] == 0 and st["up_ck"] == 16 and st["tile_cfg"] in (3, 4, 6)
            and len(st["up_phases"]) <= hip.CONV_MAX_SEG
            and all((st["cin"] // 16 * len(ph["offs"])) % 2 == 0 for ph in st["up_phases"]))


def plan_switches():
    """The environment switches that shape a launch PLAN, read ONCE when a model is built (Vocoder.sw) and used for every plan
    of that model: a model's launches (and, for FH_WINO_SPLITK, its bits) do not change when the environment does while it
    lives.  (The switches that decide which kernel a stage's weights are packed for -- FH_WINO, FH_WINO54*, FH_AMP,
    FH_CONV_BF16X6 -- are read at construction as well, by use_wino / use_wino54 / use_amp / use_bf16x6.)"""
    return dict(splitk=os.environ.get("FH_WINO_SPLITK", "1") != "0", ups_fuse=os.environ.get("FH_UPS_FUSE", "1") != "0")


def use_gemm_bf16x6(form=None):
    """The transformer's linears of a bf16 x 6 model run in the bf16 x 6 form too (gemm_bf.hip; flow.FlowNet(bf=)).  (The A/B
    against the fp32 GEMM inside a bf16 x 6 model -- 12.89 against 13.02 ms per step, configs[4] 632 against 593 x real time --
    was run with an environment switch that left with the measurement; conv_form='winograd' is the fp32 path.)"""
    return (resolve_conv_form()[0] if form is None else form) == "bf16x6"


def amp_tile_len(d):
    """Outputs per block and row of the narrow-stage kernel at dilation d (fh_amp_tile_len)."""
    return 5 * d * 4 * (16 // d)


AMP_MAX_D = 6             # F_MAX_D of amp_fused.hip
AMP_DIRECT = 4            # plan flag of a narrow-stage launch: the direct bf16 x 6 form (fh_narrow_conv_bf16x6_f32, narrow_bf.hip)
NARROW_TILE = 256         # fh_narrow_tile_len(): outputs per block and row of that form, any dilation


def use_amp_bf16x6(form=None):
    """Narrow stages of a bf16 x 6 model run the direct bf16 x 6 kernel (narrow_bf.hip) instead of the fp32-MFMA Winograd one
    (amp_fused.hip, the narrow-stage kernel of conv_form='winograd').  (A/B inside a bf16 x 6 model: same speed, five times
    closer to float64 -- profiles/r06_narrow_bf16x6.txt; the switch it was run with left with the measurement.)"""
    return (resolve_conv_form()[0] if form is None else form) == "bf16x6"


def make_amp_seg(x, u, k, center=None, direct=False):
    """One K segment of a narrow-stage group: conv weights `u` (pack_amp_weight; direct: pack_narrow_bf_weight, and the
    descriptor's ngrp field carries the tap count -- flowhigh_hip.h: fh_narrow_conv_bf16x6_f32) applied to x."""
    s = hip.AmpSeg()
    s.x, s.u, s.ngrp = _addr(x), _addr(u), k if direct else -(-k // 4)
    s.center = (k - 1) // 2 if center is None else center
    return s


def make_amp_group(segs, bias, res, out, length, scale=1.0, direct=False):
    g = hip.AmpGroup()
    # the kernel walks the segments in one pass per tap-group count, largest first
    segs = sorted(segs, key=lambda s: -s.ngrp)
    if any(s.ngrp > (11 if direct else 3) or s.center > 5 or (direct and s.center >= s.ngrp) for s in segs):
        raise NotImplementedError("narrow-stage kernel: kernels of at most 11 taps")
    for i, s in enumerate(segs):
        g.seg[i] = s
    g.nseg, g.nres = len(segs), len(res)
    g.bias = _addr(bias)
    for i, r in enumerate(res):
        g.res[i] = _addr(r)
    g.out, g.len, g.scale = _addr(out), length, scale
    return g


def amp_tile_list(lens, batch, dilation, interleave=True, direct=False):
    """The work list of a narrow-stage launch (fh_amp_tile: group, batch item, first output, len): int32 tensor [tiles, 4].
    The launch's persistent blocks take tiles b, b + grid, b + 2 grid, ... of this list.  Order: the groups' tiles dealt
    round-robin (group 0's first tile, group 1's first, ...), so that blocks with neighbouring ids work on tiles of DIFFERENT
    length at any moment: with all tiles of the heaviest group first, every block of the chip ran the same K loop and then
    stored its outputs in the same microseconds (no stores for 10 us, then 138 MB at once: the epilogue's HBM traffic cost a
    quarter of the launch, tools/exp/amp_ab.sh).  A block still gets the same share of every group.  interleave=False: group
    after group (heavy first).  (Both orders, and a layout that deals the tiles to the blocks by weight, longest first, measure
    the same to +-0.1 %: profiles/r05_amp_ablation.txt item 9; the environment switch for it left in round 6.)"""
    tb = hip.lib().fh_narrow_tile_len() if direct else amp_tile_len(dilation)
    per_group = []
    for gi, length in enumerate(lens):
        t0 = torch.arange(0, length, tb, dtype=torch.int32)
        e = torch.empty(batch, len(t0), 4, dtype=torch.int32)
        e[..., 0], e[..., 3] = gi, length
        e[..., 1] = torch.arange(batch, dtype=torch.int32)[:, None]
        e[..., 2] = t0[None, :]
        per_group.append(e.reshape(-1, 4))
    if not interleave or len(per_group) == 1:
        return torch.cat(per_group, dim=0).contiguous()
    n = max(len(e) for e in per_group)
    # position of tile i of group g in the dealt order: i * n_groups + g (groups that run out leave holes, dropped below)
    keys = torch.cat([torch.arange(len(e), dtype=torch.int64) * len(per_group) + gi for gi, e in enumerate(per_group)])
    rows = torch.cat(per_group, dim=0)
    return rows[torch.argsort(keys)].contiguous()


def amp_max_center(groups, direct=False):
    """max_center of a narrow-stage launch; raises if some segment's taps do not fit the slab it implies (direct form: every
    segment has its own slab, nothing to check)."""
    segs = [g.seg[i] for g in groups for i in range(g.nseg)]
    cmax = max(s.center for s in segs)
    if direct:
        return cmax
    if any(cmax - s.center + 4 * s.ngrp + 3 > 16 for s in segs):
        raise NotImplementedError("narrow-stage kernel: the kernel sizes of one launch are too far apart")
    return cmax


WINO_BF16X6 = 16          # FH_WINO_BF16X6 of flowhigh_hip.h: tile_cfg flag, three-piece bf16 weights
WINO_XCD_RANGES = 32      # FH_WINO_XCD_RANGES: tile_cfg flag, blocks -> XCDs by time range instead of by weight panel
WINO_NOVL = 64            # FH_WINO_NOVL: tile_cfg flag, some row of the launch is not 16-byte aligned although len % 4 == 0
WINO_MAX_K = 12           # the kernel instantiates 1..4 tap groups of 3


def use_bf16x6():
    """Does the environment (FH_CONV_FORM / FH_CONV_BF16X6 / FH_WINO, else the default form) ask for the bf16 x 6 form: the
    Winograd convs on the BF16 matrix cores with every fp32 operand split exactly into three bf16 pieces (6 bf16 MFMAs per
    16-channel k-block, fp32 accumulation), fp32-grade products at 0.375 of the matrix-pipe cycles.  (resolve_conv_form)"""
    return resolve_conv_form()[0] == "bf16x6"


def _addr(t):
    """Device address of a tensor, or an address computed by the caller (a channel slice of a batch item)."""
    return t if isinstance(t, int) else hip.ptr(t)


def wino_taps(cfg):
    """Taps per group of the kernel a plan tile id names: 4 (F(5,4)) or 3 (F(4,3))."""
    return 4 if cfg & WINO_F54 else 3


def wino_split_k(ks, c, wpad, length, dil, default_cfg, bf=False, enabled=None):
    """Number of input-channel slices (1, 2 or 3) of a residual-stack launch (one group per kernel size in ks)."""
    t = wino_taps(default_cfg)
    return wino_split_steps([c // 16 * -(-k // t) for k in ks], c, wpad, length, dil, default_cfg, bf, enabled)


def wino_split_steps(ksteps, cin, wpad, length, dil, default_cfg, bf=False, enabled=None):
    """Number of input-channel slices (1, 2 or 3) of a Winograd launch whose groups have `ksteps` K steps
    (cin / 16 x tap groups) each: more than one only where the batch-1 launch model says the blocks are too few and
    too long (clips under ~2 s); never a function of the batch size, so a clip gives the same bits alone and inside a
    batch.  FH_WINO_SPLITK=0 switches it off (enabled: a model's snapshot of that switch, Vocoder.sw; None: the environment)."""
    if not (os.environ.get("FH_WINO_SPLITK", "1") != "0" if enabled is None else enabled):
        return 1
    base = choose_wino_cfg(ksteps, 1, wpad, length, dil, default_cfg, bf)[1]
    best, n = base, 1
    for ns in (2, 3):
        if cin % (16 * ns):
            continue
        cost = choose_wino_cfg([k // ns for k in ksteps for _ in range(ns)], 1, wpad, length, dil,
                               default_cfg, bf)[1] + 7.0 * len(ksteps)      # + the adds of the partial outputs
        if cost < 0.95 * base and cost < best:     # (a slice must be estimated >= 5 % faster)
            best, n = cost, ns
    return n


def wino_block_mapping(groups, batch, wpad, length, dil, wcfg):
    """WINO_XCD_RANGES or 0: which block -> XCD mapping reads less from HBM (tools/traffic_per_launch.py: at batch 1 the
    launches of the 768 / 384 / 192-channel stages read 2.2-5.6 x their algorithmic bytes).  By weight panel (default):
    a panel's weights are fetched ~once, but the co tiles of a group sit on up to 8 XCDs and each reads the group's
    whole input: input x min(co tiles, 8), weights x min(runs per panel, 8).  By time range: the input is read once,
    every XCD fetches all weights once per rectangle of time tiles: weights x 8 x rectangles.  Same bits either way; transposed-conv phase groups (strided outputs) keep the default."""
    if wcfg & WINO_F54:
        return 0
    bm, bt = _WINO_TILES[wcfg]
    n_tiles = -(-(-(-length // dil)) // bt) * dil
    # (batch 1 only: with more batch items the by-panel blocks of different items already share a panel's weights through
    # the L2, and the time-range order measured 4-6 % SLOWER at B = 8 and 32; at B = 1 it is neutral in time)
    if batch > 1 or n_tiles < 16 or any(g.out_stride > 1 for g in groups):
        return 0
    weights = sum(g.seg[i].cin * g.seg[i].ngrp * 6 * wpad * 4.0 for g in groups for i in range(g.nseg))
    inputs = sum(g.seg[i].cin * length * 4.0 * batch for g in groups for i in range(g.nseg))
    co_tiles = wpad // bm
    run_len = -(-n_tiles // -(-n_tiles // _WINO_RUN))
    runs_per_panel = -(-n_tiles // run_len)             # (a panel's runs are dealt to different XCDs)
    by_panel = weights * min(runs_per_panel, 8) + inputs * min(co_tiles, 8)
    tpx = -(-n_tiles // 8)
    rect = max(1, 32 // co_tiles)                       # time tiles of a rectangle (conv_wino.hip, xcd_ranges branch)
    by_range = 8.0 * weights * -(-tpx // rect) + inputs
    return WINO_XCD_RANGES if by_range < 0.9 * by_panel else 0


def make_wino_seg(x, u, cin, k, center=None, xlen=0, taps=3):
    """taps: 3 for the F(4,3) kernel's weights (pack_wino_weight), 4 for F(5,4) (pack_wino54_weight)."""
    s = hip.WinoSeg()
    if taps == 4 and (-(-k // 4) > 3 or xlen):
        # (conv_wino54.hip walks segments of 1 .. 3 tap groups and takes no xlen: such a segment would be skipped silently)
        raise NotImplementedError("F(5,4) segments: at most 12 taps, no xlen")
    s.x, s.u, s.cin, s.ngrp = _addr(x), _addr(u), cin, -(-k // taps)
    s.center = (k - 1) // 2 if center is None else center
    s.xlen = xlen
    return s


def make_wino_group(segs, bias, res, out, cout, cpad, length, scale=1.0, stride=1, phase=0, out_len=0):
    g = hip.WinoGroup()
    # the kernel walks the segments in one pass per tap-group count, largest first
    for i, s in enumerate(sorted(segs, key=lambda s: -s.ngrp)):
        g.seg[i] = s
    g.nseg, g.nres = len(segs), len(res)
    g.bias = _addr(bias)
    for i, r in enumerate(res):
        g.res[i] = _addr(r)
    g.out = _addr(out)
    if max(cout * stride, max(s.cin for s in segs)) * length * 4 >= 2 ** 31:
        raise NotImplementedError("per-clip tensor exceeds the 2 GiB range of a buffer descriptor")
    g.cout, g.cout_pad, g.len, g.scale = cout, cpad, length, scale
    g.out_stride, g.out_phase, g.out_len = stride, phase, out_len
    return g


class _PlanBuilder:
    """Builds the launch plan of one [batch, num_mels, n_frames] shape (Vocoder.plan): workspace pool, descriptor
    arrays and the ordered list of launch steps.
    steps[i] = (kind, ...) is what Vocoder._launch runs; meta[i] = (position key, host descriptor structs): the position
    key names the step's place in the model -- (stage, sub-block, slot, index) -- so that the plans of different clips
    can be merged launch by launch (Vocoder.plan_ragged) although their optional steps differ."""

    def __init__(self, voc, batch, n_frames, ref_frames):
        self.v, self.B, self.N = voc, batch, n_frames
        self.f32 = dict(dtype=torch.float32, device=voc.device)
        self.steps, self.meta, self.keep = [], [], []         # keep: tensors the descriptors point into
        self.key = None                                       # position key of the steps being added
        self.executed = 0.0     # FLOPs issued to the matrix cores by all conv launches (Winograd: 1.5 G / k of the algorithmic)
        self.direct = 0.0       # ... of which by the direct-kernel launches
        self.conv_launches = []  # (kernel family, executed FLOPs, algorithmic FLOPs) of every conv launch, launch order (bench.py)
        self.L = n_frames                                     # current stage length
        self.Lref = ref_frames                                # ... of the whole clip (== L unless this plan is a chunk)
        self.parts = None                                     # split-K partial outputs, allocated on first use
        v = voc
        dils = sorted({d for dl in v.dil for d in dl})
        # (phase-major intermediates of the dilated convs are padded to d * phase_len >= L floats per row)
        lens = v.stage_lengths(n_frames)
        self.max_elems = max(st["c"] * max(d * phase_len(lens[i], d) for d in dils + [1]) for i, st in enumerate(v.stages))
        # (rows past the checkpoint's num_mels -- channel padding to a multiple of 8 -- stay zero)
        self.mel_in = torch.zeros(batch, v.num_mels, n_frames, **self.f32)
        self.pool = torch.empty(2 + 4 * v.nk, batch * self.max_elems, **self.f32)
        self.keep.append(self.pool)

    # ---- step bookkeeping -------------------------------------------------------------------------------------
    def at(self, *key):
        self.key = key

    def add(self, step, structs=None, key=None):
        self.meta.append((key if key is not None else self.key, structs))
        self.steps.append(step)

    def parts_buffer(self):
        if self.parts is None:
            self.parts = torch.empty(2 * self.v.nk, self.B * self.max_elems, **self.f32)
            self.keep.append(self.parts)
        return self.parts

    # ---- one launch each ----------------------------------------------------------------------------------------
    def conv(self, groups, cpad, n_len, tcfg, ck):
        """Direct-kernel launch.  Few-block launches (first-stage upsampler, fused stage-closing conv at short
        sequence lengths) switch from the 128 x 128 to the 128 x 64 tile to fill the 256 CUs."""
        if tcfg == 0 and len(groups) * self.B * (cpad // 128) * -(-n_len // 128) < 512:
            tcfg = 5
        d = hip.to_device_struct_array(groups, self.v.device)
        self.keep.append(d)
        flops = sum(2.0 * g.cout * g.seg[i].cin * g.seg[i].ntaps * n_len * self.B for g in groups for i in range(g.nseg))
        self.executed += flops
        self.direct += flops
        self.conv_launches.append(("direct", flops, flops))
        self.add(("conv", d, len(groups), cpad, n_len, tcfg, ck, flops), groups)

    def convt(self, groups, cpad, n_len, tcfg, phases):
        """Direct-kernel launch of phase-fused transposed-conv groups (fh_conv_transpose_fused_f32)."""
        d = hip.to_device_struct_array(groups, self.v.device)
        self.keep.append(d)
        flops = sum(2.0 * g.cout * g.seg[i].cin * g.seg[i].ntaps * n_len * self.B for g in groups for i in range(g.nseg))
        self.executed += flops
        self.direct += flops
        self.conv_launches.append(("direct", flops, flops))
        self.add(("convt", d, len(groups), cpad, n_len, tcfg, phases, flops), groups)

    def wino(self, groups, wpad, length, dil, wcfg, pm=False, flops=None, batch=None, novl=False):
        """Winograd launch; the tile shape is the launch model's (choose_wino_cfg).  batch: launches whose groups are
        per batch item (input-channel slices) pass 1.  novl: some row of the launch is not 16-byte aligned although
        `length` may be a multiple of 4 (segments with xlen)."""
        B = self.B if batch is None else batch
        wcfg, _ = choose_wino_cfg([sum(g.seg[i].cin // 16 * g.seg[i].ngrp for i in range(g.nseg)) for g in groups],
                                  B, wpad, length, dil, default=wcfg, bf=self.v.bf)
        wcfg |= wino_block_mapping(groups, B, wpad, length, dil, wcfg)
        if wcfg & WINO_F54 and any(g.out_stride > 1 or g.out_len or g.seg[i].ngrp > 3 or g.seg[i].xlen
                                   for g in groups for i in range(g.nseg)):
            raise NotImplementedError("the F(5,4) kernel takes plain convs of at most 12 taps (no strided outputs, xlen, out_len)")
        if novl:
            if wcfg & WINO_F54:
                raise NotImplementedError("the F(5,4) kernel takes no segments with xlen")
            wcfg |= WINO_NOVL
        d = hip.to_device_struct_array(groups, self.v.device)
        self.keep.append(d)
        if flops is None:
            flops = sum(2.0 * g.cout * g.seg[i].cin * (2 * g.seg[i].center + 1) * length * B
                        for g in groups for i in range(g.nseg))
        # multiply-adds the matrix cores actually execute: 6 per 4 outputs per group of 3 taps, or 8 per 5 per group of 4
        per_out = 1.6 if wcfg & WINO_F54 else 1.5
        ex = sum(2.0 * g.cout * g.seg[i].cin * per_out * g.seg[i].ngrp * length * B for g in groups for i in range(g.nseg))
        self.executed += ex
        self.conv_launches.append((("wino54" if wcfg & WINO_F54 else "wino43") + ("_bf16x6" if self.v.bf else ""), ex, flops))
        self.add(("wino", d, len(groups), wpad, length, dil, flops, wcfg, int(pm), B), groups)

    def amp(self, groups, c, length, dil):
        """Narrow-stage launch: the groups' convs (fh_amp_actconv_f32; the model's narrow stages in the direct bf16 x 6 form:
        fh_narrow_conv_bf16x6_f32, plan flag AMP_DIRECT)."""
        B = self.B
        direct = self.v.amp_direct
        tiles = amp_tile_list([g.len for g in groups], B, dil, direct=direct).to(self.v.device)
        d = hip.to_device_struct_array(groups, self.v.device)
        self.keep += [d, tiles]
        flops = sum(2.0 * c * c * (2 * g.seg[i].center + 1) * length * B for g in groups for i in range(g.nseg))
        # executed: Winograd F(5,4) 1.6 ceil(k / 4) multiply-adds per output; direct: the taps themselves
        ex = flops if direct else sum(2.0 * c * c * 1.6 * g.seg[i].ngrp * length * B for g in groups for i in range(g.nseg))
        self.executed += ex
        self.conv_launches.append(("narrow_bf16x6" if direct else "amp", ex, flops))
        flags = int(all(g.len % 4 == 0 for g in groups)) | (AMP_DIRECT if direct else 2)
        self.add(("amp", d, len(groups), tiles, tiles.shape[0], c, dil, amp_max_center(groups, direct), flags, flops), groups)

    def act(self, groups, c, length, din=1, dout=1):
        d = hip.to_device_struct_array(groups, self.v.device)
        self.keep.append(d)
        self.add(("act", d, len(groups), c, length, din, dout), groups)

    def res_conv(self, st, ents, xs_in, ks, dil, outs, res, pm=False, defer_sum=False):
        """One launch of the same conv position in the nk AMP blocks (one group per block) at the current stage
        length.  Returns, per block, the tensors whose sum is the conv's output (more than one: input-channel slices
        whose partial outputs the caller adds, defer_sum)."""
        c, cpad, wpad, L, B = st["c"], st["cpad"], st["wpad"], self.L, self.B
        biases = [e["b"] for e in ents]
        if all("ua" in e for e in ents):               # narrow stage: plain tensors whatever the dilation
            dr = self.v.amp_direct
            self.amp([make_amp_group([make_amp_seg(xs_in[i], ents[i]["ua"], ks[i], direct=dr)], biases[i], res[i],
                                     outs[i], L, direct=dr) for i in range(len(ents))], c, L, dil)
            return [[o] for o in outs]
        all_wino = all("u" in e for e in ents)
        nsplit = wino_split_k(ks, c, wpad, self.Lref, dil, st["wcfg"], self.v.bf, self.v.sw["splitk"]) if all_wino else 1
        if nsplit > 1:
            # Short clips: a launch of a few dozen blocks is bound by the K loop of ONE block.  The input channels
            # are cut into nsplit slices, one group each (first slice: bias and residual), and the partial
            # outputs are added in a fixed order.  Decided from the clip length alone, so a clip gives the same
            # bits alone and inside a batch; one group per batch item (a slice is not a whole [B, C, L] tensor).
            parts = self.parts_buffer()
            pitch = dil * phase_len(L, dil) if pm else L
            cs = c // nsplit
            addr = lambda t, b, ch: t.data_ptr() + 4 * (b * c + ch) * pitch
            groups = []
            for i, e in enumerate(ents):
                for sl in range(nsplit):
                    dst = outs[i] if sl == 0 else parts[2 * i + sl - 1]
                    for b in range(B):
                        seg = make_wino_seg(addr(xs_in[i], b, sl * cs), e["u"][sl * cs // 16:], cs, ks[i], taps=st["taps"])
                        groups.append(make_wino_group([seg], biases[i] if sl == 0 else None,
                                                      [addr(r, b, 0) for r in res[i]] if sl == 0 else [],
                                                      addr(dst, b, 0), c, wpad, L))
            self.wino(groups, wpad, L, dil, st["wcfg"], pm, batch=1, flops=sum(2.0 * c * c * k * L * B for k in ks))
            pieces = [[outs[i]] + [parts[2 * i + sl] for sl in range(nsplit - 1)] for i in range(len(ents))]
            if not defer_sum:
                for i in range(len(ents)):
                    self.add(("sum", pieces[i], outs[i], B * c * pitch, 1.0), key=self.key[:2] + (self.key[2] + 1, i))
            return pieces
        if all_wino:
            self.wino([make_wino_group([make_wino_seg(xs_in[i], ents[i]["u"], c, ks[i], taps=st["taps"])], biases[i], res[i],
                                       outs[i], c, wpad, L) for i in range(len(ents))], wpad, L, dil, st["wcfg"], pm)
        else:
            groups = []
            for i, e in enumerate(ents):
                offs = [(t - (ks[i] - 1) // 2) * dil for t in range(ks[i])]
                groups.append(make_conv_group([make_conv_seg(xs_in[i], e["w"], c, offs)], biases[i], res[i], outs[i],
                                              c, cpad, L, L, L))
            self.conv(groups, cpad, L, st["tile_cfg"], st["ck"])
        return [[o] for o in outs]

    def mixed_dilation_conv(self, st, order, ents, ks, ds, xs_in, outs, res):
        """The nk convs of one position with DIFFERENT dilations: one direct launch, per-group tap offsets."""
        c, cpad, L = st["c"], st["cpad"], self.L
        self.conv([make_conv_group([make_conv_seg(xs_in[n], e["w"], c, [(t - (k - 1) // 2) * d for t in range(k)])],
                                   e["b"], res[n], outs[n], c, cpad, L, L, L)
                   for n, (e, k, d) in enumerate(zip(ents, ks, ds))], cpad, L, st["tile_cfg"], st["ck"])

    # ---- model sections -------------------------------------------------------------------------------------------
    def conv_pre(self):
        v, N = self.v, self.N
        pre = torch.empty(self.B, v.c0, N, **self.f32)
        self.keep.append(pre)       # descriptors hold raw pointers: every buffer they name must outlive the plan
        self.at(-1, 0, 0, 0)
        if v.pre_u is not None:
            self.wino([make_wino_group([make_wino_seg(self.mel_in, v.pre_u, v.num_mels, 7)], v.pre_b, [], pre, v.c0,
                                       v.pre_wpad, N)], v.pre_wpad, N, 1, v.pre_wcfg)
        else:
            k7 = [j - 3 for j in range(7)]
            self.conv([make_conv_group([make_conv_seg(self.mel_in, v.pre_w, v.num_mels, k7)], v.pre_b, [], pre, v.c0,
                                       v.pre_cpad, N, N, N)], v.pre_cpad, N, v.pre_cfg, v.pre_ck)
        return pre

    def average(self, ys, out, n, scale, key):
        """out = scale * (((ys[0] + ys[1]) + ys[2]) + ...): the `xs += resblock(x)` / `xs / num_kernels` of
        models.py:181-187 in the reference's block order, for stages whose closing conv is not fused."""
        if len(ys) in (2, 3):
            self.add(("mean", ys[0], ys[1], ys[2] if len(ys) == 3 else None, out, n, scale), key=key)
        elif len(ys) <= 12:
            self.add(("sum", list(ys), out, n, scale), key=key)
        else:
            raise NotImplementedError("more than 12 resblock kernel sizes")

    def enter_stage(self, i):
        """Stage i: lengths and the views of the workspace pool (slot 0 = X: upsampled input, 1 = S: stage output,
        then per AMP block j: 2 + 4 j = T1, 3 + 4 j = T2, 4 / 5 + 4 j = Y ping-pong)."""
        v, st = self.v, self.v.stages[i]
        # (ConvTranspose1d with an odd k - u returns u * L + 1 samples: models.py:141-146)
        self.lin, self.L = self.L, self.L * st["u"] + st["extra"]
        self.lin_ref, self.Lref = self.Lref, self.Lref * st["u"] + st["extra"]
        c, B, L = st["c"], self.B, self.L
        view = lambda idx: self.pool[idx, :B * c * L].view(B, c, L)
        self.X, self.S = view(0), view(1)
        self.T1 = [view(2 + 4 * j) for j in range(v.nk)]
        self.T2 = [view(3 + 4 * j) for j in range(v.nk)]
        self.Y = [[view(4 + 4 * j), view(5 + 4 * j)] for j in range(v.nk)]
        # heavy kernel sizes first (dispatch order == launch order of the panels)
        self.order = sorted(range(v.nk), key=lambda j: -st["blocks"][j]["k"])

    def upsampler(self, i, cur):
        """ConvTranspose1d(cin -> c, stride u) as u output-phase groups: Winograd groups with strided stores for the
        wide stages, direct-kernel groups otherwise."""
        v, st, B = self.v, self.v.stages[i], self.B
        c, u, lin, L, X = st["c"], st["u"], self.lin, self.L, self.X
        # k - u odd: L = u * lin + 1; phase 0 has lin + 1 output positions, the other phases lin
        extra = st["extra"]
        npos = lin + extra
        self.at(i, -1, 0, 0)
        if st["up_wino"] is None and ups_fused_ok(st, v.sw["ups_fuse"]):
            # all u output phases of a (co, time) tile in ONE block: segment p = phase p, whole-line stores
            segs = [make_conv_seg(cur, ph["w"], st["cin"], ph["offs"]) for ph in st["up_phases"]]
            self.convt([make_conv_group(segs, st["up_b"], [], X, c, st["cpad"], lin, L, lin, stride=u, phase=0)],
                       st["cpad"], lin, st["tile_cfg"], u)
            return
        if st["up_wino"] is None:
            self.conv([make_conv_group([make_conv_seg(cur, ph["w"], st["cin"], ph["offs"])], st["up_b"], [], X, c,
                                       st["cpad"], lin, L, npos if r == 0 else lin, stride=u, phase=r)
                       for r, ph in enumerate(st["up_phases"])], st["cpad"], npos, st["tile_cfg"], st["up_ck"])
            return
        # (Winograd phase groups: all have `npos` positions, writes at u * n + r >= L are masked: fh_wino_group.out_len)
        xlen, olen = (lin, L) if extra else (0, 0)
        up_flops = sum(2.0 * c * st["cin"] * ph["k"] * lin * B for ph in st["up_wino"])
        nsplit = wino_split_steps([st["cin"] // 16 * -(-ph["k"] // 3) for ph in st["up_wino"]], st["cin"], st["up_wpad"],
                                  self.lin_ref + extra, 1, st["up_wcfg"], v.bf, v.sw["splitk"])
        if nsplit == 1:
            self.wino([make_wino_group([make_wino_seg(cur, ph["u"], st["cin"], ph["k"], ph["center"], xlen=xlen)], st["up_b"],
                                       [], X, c, st["up_wpad"], npos, stride=u, phase=r, out_len=olen)
                       for r, ph in enumerate(st["up_wino"])],
                      st["up_wpad"], npos, 1, st["up_wcfg"], flops=up_flops, novl=bool(extra))
            return
        parts = self.parts_buffer()             # short clips: input channels in slices, as in res_conv
        cs = st["cin"] // nsplit
        dsts = [X] + [parts[sl] for sl in range(nsplit - 1)]
        groups = [make_wino_group([make_wino_seg(cur.data_ptr() + 4 * (b * st["cin"] + sl * cs) * lin,
                                                 ph["u"][sl * cs // 16:], cs, ph["k"], ph["center"], xlen=xlen)],
                                  st["up_b"] if sl == 0 else None, [], dsts[sl].data_ptr() + 4 * b * c * L,
                                  c, st["up_wpad"], npos, stride=u, phase=r, out_len=olen)
                  for r, ph in enumerate(st["up_wino"]) for sl in range(nsplit) for b in range(B)]
        self.wino(groups, st["up_wpad"], npos, 1, st["up_wcfg"], flops=up_flops, batch=1, novl=bool(extra))
        self.add(("sum", dsts, X, B * c * L, 1.0), key=(i, -1, 1, 0))

    def amp1_stack(self, i):
        """The nk AMPBlock1 of stage i (models.py:21-78), position by position: act -> conv1 (dilated) -> act -> conv2
        (+ x) per dilation; the last conv2 closes the stage (closing_conv)."""
        v, st = self.v, self.v.stages[i]
        c, L, order = st["c"], self.L, self.order
        T1, T2, Y = self.T1, self.T2, self.Y
        xin = [self.X] * v.nk
        blks = [st["blocks"][j] for j in order]
        ks = [b_["k"] for b_ in blks]
        for m in range(v.nm):
            d1 = blks[0]["dil"][m]
            same_d = all(b_["dil"][m] == d1 for b_ in blks)
            # dilated Winograd conv: the activations on both sides write / read phase-major tensors
            pm = same_d and 1 < d1 <= 16 and all("u" in b_["c1"][m] for b_ in blks)
            dpm = d1 if pm else 1
            ents = [b_["c1"][m] for b_ in blks]
            self.at(i, m, 0, 0)
            self.act([make_act_group(xin[j], T1[j], st["blocks"][j]["acts"][2 * m]) for j in order], c, L, dout=dpm)
            self.at(i, m, 1, 0)
            if same_d:
                self.res_conv(st, ents, [T1[j] for j in order], ks, d1, [T2[j] for j in order], [[] for _ in blks], pm=pm)
            else:
                self.mixed_dilation_conv(st, order, ents, ks, [b_["dil"][m] for b_ in blks], [T1[j] for j in order],
                                         [T2[j] for j in order], [[] for _ in blks])
            self.at(i, m, 3, 0)
            self.act([make_act_group(T2[j], T1[j], st["blocks"][j]["acts"][2 * m + 1]) for j in order], c, L, din=dpm)
            self.at(i, m, 4, 0)
            ents = [b_["c2"][m] for b_ in blks]
            if m < v.nm - 1:
                self.res_conv(st, ents, [T1[j] for j in order], ks, 1, [Y[j][m % 2] for j in order],
                              [[xin[j]] for j in order])
                xin = [Y[j][m % 2] for j in range(v.nk)]
            else:
                self.closing_conv(i, m, ents, ks, [1] * v.nk, xin)

    def amp2_stack(self, i):
        """The nk AMPBlock2 of stage i (models.py:81-121): one activation + one conv (+ x) per dilation; the last one
        closes the stage."""
        v, st = self.v, self.v.stages[i]
        c, L, order = st["c"], self.L, self.order
        T1, Y = self.T1, self.Y
        xin = [self.X] * v.nk
        blks = [st["blocks"][j] for j in order]
        ks = [b_["k"] for b_ in blks]
        for m in range(v.nm):
            ents = [b_["c1"][m] for b_ in blks]
            ds = [b_["dil"][m] for b_ in blks]
            self.at(i, m, 0, 0)
            self.act([make_act_group(xin[j], T1[j], st["blocks"][j]["acts"][m]) for j in order], c, L)
            self.at(i, m, 1, 0)
            if m == v.nm - 1 and (all("w" in e for e in ents) or (all("ua" in e for e in ents) and all(d == ds[0] for d in ds))):
                self.closing_conv(i, m, ents, ks, ds, xin)          # direct / narrow-stage kernel: K segments of one group
                continue
            outs = [Y[j][m % 2] for j in order]
            if all(d == ds[0] for d in ds):
                self.res_conv(st, ents, [T1[j] for j in order], ks, ds[0], outs, [[xin[j]] for j in order])
            else:
                self.mixed_dilation_conv(st, order, ents, ks, ds, [T1[j] for j in order], outs, [[xin[j]] for j in order])
            xin = [Y[j][m % 2] for j in range(v.nk)]
            if m == v.nm - 1:            # xs / num_kernels, block order = the reference's xs += order
                self.average(xin, self.S, self.B * c * L, 1.0 / v.nk, key=(i, m, 6, 0))

    def closing_conv(self, i, m, ents, ks, ds, xin):
        """The stage-closing conv position: xs = sum over blocks of (conv(T1_j) + x_j); S = xs / num_kernels
        (models.py:181-187).  Fused form: ONE group with nk K segments, the blocks are summed in the accumulator and
        / nk is the epilogue scale.  Unfused form: nk groups (more blocks for the 256 CUs) + one averaging pass; whichever
        the launch model says is faster for ONE clip of the whole clip's length (the two forms round differently, and a
        clip must give the same bits alone, inside a batch and in chunks)."""
        v, st = self.v, self.v.stages[i]
        c, cpad, wpad, L, B, order = st["c"], st["cpad"], st["wpad"], self.L, self.B, self.order
        T1, Y, S = self.T1, self.Y, self.S
        scale = 1.0 / v.nk
        fusable = v.nk <= hip.CONV_MAX_SEG          # (K segments of one group; more blocks: one group each + one averaging pass)
        if all("ua" in e for e in ents) and fusable and all(d == ds[0] for d in ds):
            # narrow stage: always the fused form (its blocks are short whatever the length: nothing to decide per clip)
            dr = self.v.amp_direct
            segs = [make_amp_seg(T1[j], e["ua"], k, direct=dr) for j, e, k in zip(order, ents, ks)]
            self.amp([make_amp_group(segs, st["last_bias"], [xin[j] for j in order], S, L, scale=scale, direct=dr)], c, L, ds[0])
            return
        if all("u" in e for e in ents):
            ksteps = [c // 16 * -(-k // st["taps"]) for k in ks]
            Lr = self.Lref
            # (averaging pass: 16 bytes per element that the conv launch has just written -- measured 19 us for 184 MB, i.e. it
            # runs out of the last-level cache, ~8 bytes per ns)
            unfuse = not fusable or (v.nk in (2, 3) and (choose_wino_cfg(ksteps, 1, wpad, Lr, 1, st["wcfg"], v.bf)[1] + 4.0 + c * Lr * 16 / 8.0e6
                                                         < choose_wino_cfg([sum(ksteps)], 1, wpad, Lr, 1, st["wcfg"], v.bf)[1]))
            if not unfuse:
                segs = [make_wino_seg(T1[j], e["u"], c, k, taps=st["taps"]) for j, e, k in zip(order, ents, ks)]
                self.wino([make_wino_group(segs, st["last_bias"], [xin[j] for j in order], S, c, wpad, L, scale=scale)],
                          wpad, L, 1, st["wcfg"])
                return
            one_pass = 3 * v.nk <= 12                           # (a sum pass takes 12 sources: up to 3 slices per block)
            pieces = self.res_conv(st, ents, [T1[j] for j in order], ks, 1, [Y[j][m % 2] for j in order],
                                   [[xin[j]] for j in order], defer_sum=one_pass)
            if len(pieces[0]) > 1 and one_pass:                 # input-channel slices: all partial outputs in one pass
                by_block = {j: pieces[n_] for n_, j in enumerate(order)}
                self.add(("sum", [t for j in range(v.nk) for t in by_block[j]], S, B * c * L, scale), key=(i, m, 6, 0))
            else:
                self.average([Y[j][m % 2] for j in range(v.nk)], S, B * c * L, scale, key=(i, m, 6, 0))
            return
        if not fusable or all("ua" in e for e in ents):
            outs, res = [Y[j][m % 2] for j in order], [[xin[j]] for j in order]
            if all(d == ds[0] for d in ds):
                self.res_conv(st, ents, [T1[j] for j in order], ks, ds[0], outs, res)
            else:
                self.mixed_dilation_conv(st, order, ents, ks, ds, [T1[j] for j in order], outs, res)
            self.average([Y[j][m % 2] for j in range(v.nk)], S, B * c * L, scale, key=(i, m, 6, 0))
            return
        segs = [make_conv_seg(T1[j], e["w"], c, [(t - (k - 1) // 2) * d for t in range(k)])
                for j, e, k, d in zip(order, ents, ks, ds)]
        self.conv([make_conv_group(segs, st["last_bias"], [xin[j] for j in order], S, c, cpad, L, L, L, scale=scale)],
                  cpad, L, st["tile_cfg"], st["ck"])

    def finish(self, cur):
        """activation_post + conv_post + tanh, and the plan record."""
        v, B, L = self.v, self.B, self.L
        c_last = v.stages[-1]["c"]
        wav = torch.empty(B, L, **self.f32)
        # (activation_post -> conv_post -> tanh as ONE launch was built in round 5: 66 us against 20 + 13 us at batch 1, 477 blocks
        # that each walk 24 channels serially; it left the library in round 6)
        post_t = self.pool[2, :B * c_last * L].view(B, c_last, L)
        self.at(99, 0, 0, 0)
        self.act([make_act_group(cur, post_t, v.post_act)], c_last, L)
        self.add(("post", post_t, wav, c_last, L), key=(99, 0, 1, 0))
        # algorithmic HBM bytes of the Activation1d launches: every site reads and writes its [B, C, L] tensor once
        act_bytes = sum(8.0 * s_[2] * B * s_[3] * s_[4] for s_ in self.steps if s_[0] == "act")
        return dict(steps=self.steps, meta=self.meta, keep=self.keep, mel_in=self.mel_in, wav=wav, B=B, N=self.N, L=L,
                    conv_executed_flops=self.executed, conv_direct_flops=self.direct, conv_launches=self.conv_launches, act_bytes=act_bytes,
                    n_act=sum(s_[0] == "act" for s_ in self.steps))


def merge_ragged(voc, frames):
    """Merged launch plan for clips of frame counts `frames` (any mix of lengths, batch 1 each).
    Every clip keeps the plan it has alone (`plan(1, N)`: its own buffers, the same groups, K segments,
    input-channel slices and partial-sum steps -- so the same bits); steps that sit at the same position of the
    model are then launched together: one conv / activation launch carries the groups of all clips (the kernels
    take a length per group; the grid is sized for the longest and blocks past a group's end exit), the partial-sum
    / averaging passes become one multi-job launch.  A mix of 24 clips runs ~120 launches instead of ~2 900."""
    frames = tuple(int(n) for n in frames)
    key = ("ragged",) + frames
    if key in voc._ragged:
        return voc._ragged[key]
    seen, subs = {}, []
    for n in frames:
        k = seen.get(n, 0)
        seen[n] = k + 1
        subs.append(voc.plan(1, n, inst=k))
    by_key = {}
    for ci, sp in enumerate(subs):
        ks = [m[0] for m in sp["meta"]]
        if len(set(ks)) != len(ks):
            raise NotImplementedError("this vocoder configuration has launch positions that cannot be merged")
        for step, (k, structs) in zip(sp["steps"], sp["meta"]):
            by_key.setdefault(k, []).append((ci, step, structs))
    blobs, merged = [], []          # host bytes of every descriptor array (one upload), merged steps

    def blob(structs):
        arr = (type(structs[0]) * len(structs))(*structs)
        off = sum(len(b) for b in blobs)
        raw = bytes(arr)
        blobs.append(raw + bytes(-len(raw) % 16))
        return off

    tt = hip.lib().fh_act_tile_len()
    for k in sorted(by_key):
        items = by_key[k]
        # (a mean is a 2-3 term sum job; the fused tail launch and conv_post share a position: both run per clip)
        kinds = {{"mean": "sum"}.get(it[1][0], it[1][0]) for it in items}
        if len(kinds) != 1:
            raise NotImplementedError(f"launch position {k}: kinds {kinds} cannot be merged")
        kind = kinds.pop()
        if kind == "wino":
            classes = {}
            for ci, st_, groups in items:
                _, _d, ng, wpad, length, dil, _fl, wcfg, pm, bb = st_
                assert bb == 1 and ng == len(groups)
                classes.setdefault((wpad, dil, pm, wcfg & WINO_F54), []).append((length, wcfg, groups))
            for (wpad, dil, pm, fam), lst in classes.items():
                allg = [(sum(g.seg[i].cin // 16 * g.seg[i].ngrp for i in range(g.nseg)), length, g)
                        for length, _, groups in lst for g in groups]
                allg.sort(key=lambda t: (-t[0], -t[1]))                 # heavy groups first (dispatch order)
                maxlen = max(t[1] for t in allg)
                default = max(lst, key=lambda t: t[0])[1] & (15 | WINO_F54)          # the longest clip's tile shape
                wcfg = default
                if default in (0, 1, 4, 5) or fam:
                    # the launch model takes one length: the mean one keeps the block count honest
                    mean_len = max(1, sum(t[1] for t in allg) // len(allg))
                    wcfg, _ = choose_wino_cfg([t[0] for t in allg], 1, wpad, mean_len, dil, default=default, bf=voc.bf)
                novl = 0 if (pm or all(t[1] % 4 == 0 for t in allg)) else 2
                if any(w & WINO_NOVL for _, w, _ in lst):               # (segments with xlen: odd-(k - u) upsamplers)
                    novl = 2
                # runs (consecutive tiles of one (group, co tile) panel, dealt to one XCD) that hold real tiles
                bm, bt = _WINO_TILES[wcfg]
                cot = wpad // bm
                n_tiles = wino_n_tiles(wcfg, maxlen, dil, pm)
                run_len = (hip.lib().fh_wino54_run_len if fam else hip.lib().fh_wino_run_len)(n_tiles)
                rpp = -(-n_tiles // run_len)
                runs = []
                for gi, (_, length, _) in enumerate(allg):
                    # tile index = (block of bt outputs within the phase) * dil + phase: real while its first
                    # output (phase + dil * bt * block) lies inside the row
                    # (blocks per phase differ by at most one, so the real tiles are 0 .. t_last without holes)
                    if fam and pm:                  # (the F(5,4) kernel's concatenated tiling: a group's real tiles are the first ones)
                        t_last = wino_n_tiles(wcfg, length, dil, pm) - 1
                    else:
                        nb = [max(0, -(-(length - ph) // (dil * bt))) for ph in range(dil)]
                        t_last = dil * (nb[0] - 1) + sum(1 for v in nb if v == nb[0]) - 1
                    own = range(t_last // run_len + 1)
                    for ct in range(cot):
                        runs += [(gi * cot + ct) * rpp + r for r in own]
                rmap = (C.c_int32 * len(runs))(*runs)
                off_map = sum(len(b) for b in blobs)
                raw = bytes(rmap)
                blobs.append(raw + bytes(-len(raw) % 16))
                merged.append(("rwino", blob([t[2] for t in allg]), len(allg), wpad, maxlen, dil, wcfg,
                               int(pm) | novl, off_map, len(runs)))
        elif kind == "conv":
            classes = {}
            for ci, st_, groups in items:
                _, _d, ng, cpad, n_len, tcfg, ck, _fl = st_
                classes.setdefault((cpad, ck), []).append((n_len, tcfg, groups))
            for (cpad, ck), lst in classes.items():
                allg = [(sum(g.seg[i].cin * g.seg[i].ntaps for i in range(g.nseg)), n_len, g)
                        for n_len, _, groups in lst for g in groups]
                allg.sort(key=lambda t: (-t[0], -t[1]))
                tcfg = max(lst, key=lambda t: t[0])[1]
                merged.append(("rconv", blob([t[2] for t in allg]), len(allg), cpad, max(t[1] for t in allg), tcfg, ck))
        elif kind == "convt":
            classes = {}
            for ci, st_, groups in items:
                _, _d, ng, cpad, n_len, tcfg, phases, _fl = st_
                classes.setdefault((cpad, tcfg, phases), []).append((n_len, groups))
            for (cpad, tcfg, phases), lst in classes.items():
                allg = sorted(((n_len, g) for n_len, groups in lst for g in groups), key=lambda t: -t[0])
                merged.append(("rconvt", blob([t[1] for t in allg]), len(allg), cpad, allg[0][0], tcfg, phases))
        elif kind == "amp":
            classes = {}
            for ci, st_, groups in items:
                _, _d, ng, _t, _nt, c, dil, _cm, flags, _fl = st_
                classes.setdefault((c, dil, flags & (2 | AMP_DIRECT)), []).append(groups)
            for (c, dil, form_bits), lst in classes.items():
                allg = [g for groups in lst for g in groups]
                direct = bool(form_bits & AMP_DIRECT)
                # heavy groups first (the persistent blocks walk the tile list in order), then long ones
                allg.sort(key=lambda g: (-sum(g.seg[i].ngrp for i in range(g.nseg)), -g.len))
                tl = amp_tile_list([g.len for g in allg], 1, dil, direct=direct)
                off_t = sum(len(b) for b in blobs)
                raw = tl.numpy().tobytes()
                blobs.append(raw + bytes(-len(raw) % 16))
                merged.append(("ramp", blob(allg), len(allg), off_t, tl.shape[0], c, dil, amp_max_center(allg, direct),
                               int(all(g.len % 4 == 0 for g in allg)) | form_bits))
        elif kind == "act":
            classes = {}
            for ci, st_, groups in items:
                _, _d, ng, c, length, din, dout = st_
                classes.setdefault((c, din, dout), []).append((length, groups))
            for (c, din, dout), lst in classes.items():
                out, base = [], 0
                for length, groups in sorted(lst, key=lambda t: -t[0]):
                    for g in groups:
                        g2 = hip.ActGroup.from_buffer_copy(g)
                        g2.len, g2.tile_base = length, base
                        base += c * -(-length // tt)
                        out.append(g2)
                merged.append(("ract", blob(out), len(out), c, din, dout, base, int(all(g.len % 4 == 0 for g in out))))
        elif kind == "sum":
            jobs = []
            for ci, st_, _ in items:
                if st_[0] == "sum":
                    _, srcs, out, n, scale = st_
                else:
                    _, a, b_, c_, out, n, scale = st_
                    srcs = [a, b_] + ([c_] if c_ is not None else [])
                j = hip.SumJob()
                for i, t in enumerate(srcs):
                    j.src[i] = t.data_ptr()
                j.out, j.n, j.n_src, j.scale = out.data_ptr(), n, len(srcs), scale
                if n % 4 or len(srcs) > 12:
                    raise NotImplementedError("partial-sum job shape")
                jobs.append(j)
            merged.append(("rsum", blob(jobs), len(jobs), max(j.n for j in jobs)))
        elif kind == "post":
            for ci, st_, _ in items:
                merged.append(st_)
        else:
            raise NotImplementedError(kind)
    host = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8)
    desc = host.to(voc.device)
    rp = dict(subs=subs, steps=merged, desc=desc, frames=frames)
    # The merged descriptors hold raw pointers into the clips' plans, so the entry keeps them alive -- and is
    # accounted with everything it keeps alive (the sub-plans' workspaces, ~65 MB per second of audio), so that
    # FH_CACHE_GB bounds what the ragged cache can pin whatever voc._plans has evicted meanwhile.
    voc._ragged[key] = rp
    return rp
