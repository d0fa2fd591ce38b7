"""Enqueueing: one-off launchers of the conv / activation entry points (tests, tools), the per-device occupancy setting of
the activation launches, and the step / ragged-step dispatch a plan is run with.
(Split out of vocoder.py in round 5; `flowhigh_amd.vocoder` re-exports every name.)
"""
import ctypes as C
import os

import torch

from . import hip
from .packing import pack_wino54_weight_any, pack_wino_weight_any
from .planner import (AMP_DIRECT, WINO_F54, amp_max_center, amp_tile_list, make_act_group, make_wino_group, make_wino_seg, pick_wino54_tile,
                      pick_wino_tile, use_wino54)

def amp_actconv(groups, batch, channels, dilation, device, direct=False):
    """Upload descriptors and enqueue one narrow-stage launch (test / one-off use).  direct: the bf16 x 6 form (groups made with
    make_amp_seg / make_amp_group(..., direct=True), weights from pack_narrow_bf_weight)."""
    tiles = amp_tile_list([g.len for g in groups], batch, dilation, direct=direct).to(device)
    d = hip.to_device_struct_array(groups, device)
    vec = int(all(g.len % 4 == 0 for g in groups))
    if direct:
        hip.check(hip.lib().fh_narrow_conv_bf16x6_f32(d.data_ptr(), len(groups), tiles.data_ptr(), tiles.shape[0], channels, dilation,
                                                      vec, hip.stream()), "fh_narrow_conv_bf16x6_f32")
    else:
        hip.check(hip.lib().fh_amp_actconv_f32(d.data_ptr(), len(groups), tiles.data_ptr(), tiles.shape[0], channels, dilation,
                                               amp_max_center(groups), vec | 2, hip.stream()), "fh_amp_actconv_f32")
    return d, tiles


def conv_wino(groups, batch, cout_pad, length, dilation, device, tile_cfg=0, phase_major=False):
    """Upload descriptors and enqueue one Winograd conv launch (test / one-off use)."""
    d = hip.to_device_struct_array(groups, device)
    if tile_cfg & WINO_F54:            # (F(5,4) kernel: the groups' weights are pack_wino54_weight, ngrp = ceil(k / 4))
        if any(g.out_stride > 1 or g.out_len or g.seg[i].ngrp > 3 or g.seg[i].xlen for g in groups for i in range(g.nseg)):
            raise NotImplementedError("the F(5,4) kernel takes plain convs of at most 12 taps (no strided outputs, xlen, out_len)")
        hip.check(hip.lib().fh_conv_wino54_f32(d.data_ptr(), len(groups), batch, cout_pad, length, dilation,
                                               int(phase_major), tile_cfg & 31, hip.stream()), "fh_conv_wino54_f32")
        return d
    hip.check(hip.lib().fh_conv_wino_f32(d.data_ptr(), len(groups), batch, cout_pad, length, dilation,
                                         int(phase_major), tile_cfg, hip.stream()), "fh_conv_wino_f32")
    return d


def conv_grouped(groups, batch, cout_pad, n_len, tile_cfg, device, ck=8):
    """Upload descriptors and enqueue one grouped conv launch (test / one-off use)."""
    d = hip.to_device_struct_array(groups, device)
    hip.check(hip.lib().fh_conv_grouped_f32(d.data_ptr(), len(groups), batch, cout_pad, n_len, tile_cfg, ck,
                                            hip.stream()), "fh_conv_grouped_f32")
    return d


def act1d_grouped(groups, batch, channels, length, device, din=1, dout=1):
    d = hip.to_device_struct_array(groups, device)
    hip.check(hip.lib().fh_act1d_grouped_pm_f32(d.data_ptr(), len(groups), batch, channels, length, din, dout,
                                                hip.stream()), "fh_act1d_grouped_pm_f32")
    return d


# ---- occupancy of the activation launches (fh_act_set_blocks_per_cu) --------------------------------------------------
# On some MI355X boxes an uncapped activation launch (7 blocks = 28 waves per CU: vector ALUs, LDS and HBM busy at once)
# makes the power management drop the shader clock for the duration of the NEXT launch: a Winograd launch that follows
# one runs at 2.11 instead of 2.38 GHz (tools/clock_dip_probe.py, tools/box_probe.sh; a device copy of the same bytes does
# not do it).  With 3 blocks per CU the dip is mostly gone (on such a box the capped activation is 10-15 % slower, the convs
# 10 % faster); on boxes without the dip the cap would only cost the activation 25 %.  So the setting is measured once per device and process.
# (tools/exp/occ_ab.sh on both kinds of affected boxes: 4 blocks help on one kind only (bench 500 -> 527) and do nothing on the
# other (the clock is clamped as before); 3 blocks help on both (488 -> 522, 500 -> 524); 2 blocks: the activation itself
# is too slow then (512))
ACT_BLOCKS_CHOICES = (0, 3, 4)          # 0 = no cap; 4 is enough on one kind of affected box (and costs the activation less), 3 on both
_act_blocks = {}                        # device ordinal -> setting in force
_act_choice = {}                        # (device ordinal, bf16 x 6 convs?) -> setting chosen for models of that form


def pick_act_blocks(pair_us, slack=0.98):
    """The decision rule: no cap unless a capped setting makes the (activation + conv) pair at least 2 % faster -- in EVERY
    measurement pass (pair_us: one {blocks: us} dict, or a list of them, one per pass): a single noisy pass on a shared
    or busy GPU must not flip the setting."""
    passes = [pair_us] if isinstance(pair_us, dict) else list(pair_us)
    wins = [b for b in passes[0] if b != 0 and all(p[b] < slack * p[0] for p in passes)]
    if not wins:
        return 0
    return min(wins, key=lambda b: sorted(p[b] for p in passes)[len(passes) // 2])


def parse_act_blocks(value):
    """FH_ACT_BLOCKS / Vocoder(act_blocks=): None, '' or 'auto' -> None (measure); 0 or 2..5 -> that setting."""
    if value is None or str(value).strip().lower() in ("", "auto"):
        return None
    try:
        v = int(str(value).strip())
    except ValueError:
        v = -1
    if v != 0 and not 2 <= v <= 5:
        raise ValueError(f"FH_ACT_BLOCKS / act_blocks must be 'auto', 0 (no cap) or 2..5 blocks per CU, got {value!r}")
    return v


def decide_act_blocks(measure, group=None, collective=False):
    """(choice, passes) from `measure()` (-> list of per-pass {blocks: us}).
    collective=False (what a constructor does): measured by the calling process, no communication -- a constructor must never
    enter a collective: whether a rank gets here depends on per-process state (the per-device cache, FH_ACT_BLOCKS, a second
    model on one rank), and the ranks that wait would hang.
    collective=True (sync_act_blocks: an explicit call that EVERY rank of `group` makes at the same point of the program): rank 0
    of the group alone measures -- eight ranks timing launch pairs while their neighbours load models under one power budget
    would each measure something else -- and broadcasts its choice, or the error it hit (a failing rank 0 must not leave the
    others waiting: they raise too)."""
    if not collective:
        passes = measure()
        return pick_act_blocks(passes), passes
    import torch.distributed as dist
    box = [None]
    if dist.get_rank(group) == 0:
        try:
            passes = measure()
            box[0] = ("ok", pick_act_blocks(passes), passes)
        except Exception as e:                            # noqa: BLE001  (reported on every rank)
            box[0] = ("error", f"{type(e).__name__}: {e}", None)
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    if box[0][0] != "ok":
        raise RuntimeError(f"activation occupancy calibration failed on rank 0 of the group: {box[0][1]}")
    return box[0][1], box[0][2]


def sync_act_blocks(device, group=None, bf=None):
    """One setting for all ranks of `group` (default: the world; pass a per-node group when the nodes differ): call it on EVERY
    rank, before the models are built; it returns the setting and puts it in force (and in the per-device cache, so that the
    constructors measure nothing).  FH_ACT_BLOCKS still overrides.  Without an initialised process group: the local calibration."""
    import torch.distributed as dist
    dev = hip.norm_device(device)
    if bf is None:                                   # (the conv form the environment's models will have)
        from .planner import use_bf16x6
        bf = use_bf16x6()
    fixed = parse_act_blocks(os.environ.get("FH_ACT_BLOCKS"))
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return calibrate_act_occupancy(dev, bf=bf)

    # EVERY rank enters the collective, whatever its own FH_ACT_BLOCKS says (a variable exported on some ranks only must not
    # leave the others waiting in the broadcast: ADVICE r05): rank 0 of the group decides -- its FH_ACT_BLOCKS if it has one,
    # else its measurement -- and every rank takes that.
    def measure():
        if fixed is not None:
            return [{0: 1.0, fixed: 0.5}] * 2 if fixed != 0 else [{0: 1.0}] * 2        # (pick_act_blocks -> `fixed`)
        return [{b: measure_act_conv_pair(dev, b, bf=bf) for b in ACT_BLOCKS_CHOICES} for _ in range(2)]
    choice, passes = decide_act_blocks(measure, group=group, collective=True)
    if fixed is not None and fixed != choice:
        import logging
        logging.getLogger("flowhigh_amd").warning("FH_ACT_BLOCKS=%s on this rank, rank 0 of the group chose %s: taking rank 0's", fixed, choice)
    calibrate_act_occupancy.last_measurement = passes
    return calibrate_act_occupancy(dev, force=True, act_blocks=choice, bf=bf)


def measure_act_conv_pair(device, blocks, c=192, length=60000, warm=60, reps=100, bf=False):
    """Average us of one (activation launch, Winograd launch) pair of a mid-network stage's size with the activation capped
    at `blocks` per CU (synthetic tensors; the launches are the model's: 3 groups, k = 11 / 7 / 3)."""
    dev = hip.norm_device(device)
    with hip.device_guard(dev):
        g = torch.Generator().manual_seed(0)
        ks = (11, 7, 3)
        xs = [torch.randn(1, c, length, generator=g).to(dev) for _ in ks]
        ys = [torch.empty(1, c, length, device=dev) for _ in ks]
        outs = [torch.empty(1, c, length, device=dev) for _ in ks]
        bias = torch.zeros(c, device=dev)
        # (the conv launch of the model at this width: the F(5,4) kernel unless it is switched off)
        # (bf: the conv of the pair in the bf16 x 6 form -- the cap exists for what the activation launch does to the clock of the
        # conv launch behind it, and that differs between the fp32 and the bf16 matrix instructions: round 6 measured a cap that
        # pays in front of fp32-MFMA convs and costs in front of bf16 x 6 ones on the same box)
        f54 = use_wino54(c, "bf16x6" if bf else "winograd")
        wcfg, wpad = pick_wino54_tile(c, bf) if f54 else pick_wino_tile(c)
        pack = (lambda w_, p_: pack_wino54_weight_any(w_, p_, bf)) if f54 else (lambda w_, p_: pack_wino_weight_any(w_, p_, bf))
        flag = 16 if bf else 0                         # FH_WINO_BF16X6
        us = [pack(torch.randn(c, c, k, generator=g) * 0.02, wpad).to(dev) for k in ks]
        gw = hip.to_device_struct_array([make_wino_group([make_wino_seg(ys[i], us[i], c, k, taps=4 if f54 else 3)], bias, [],
                                                         outs[i], c, wpad, length) for i, k in enumerate(ks)], dev)
        filt = [0.0] * 5 + [0.5, 0.5] + [0.0] * 5
        p = dict(alpha=torch.ones(c, device=dev), inv_beta=torch.ones(c, device=dev), up=filt, down=filt)
        ga = hip.to_device_struct_array([make_act_group(xs[i], ys[i], p) for i in range(len(ks))], dev)
        lib, st = hip.lib(), hip.stream()
        before = lib.fh_act_get_blocks_per_cu()
        hip.check(lib.fh_act_set_blocks_per_cu(blocks), "fh_act_set_blocks_per_cu")
        try:
            def pair():
                hip.check(lib.fh_act1d_grouped_pm_f32(ga.data_ptr(), len(ks), 1, c, length, 1, 1, st), "fh_act1d_grouped_pm_f32")
                if f54:
                    hip.check(lib.fh_conv_wino54_f32(gw.data_ptr(), len(ks), 1, wpad, length, 1, 0, (wcfg & 15) | flag, st), "fh_conv_wino54_f32")
                else:
                    hip.check(lib.fh_conv_wino_f32(gw.data_ptr(), len(ks), 1, wpad, length, 1, 0, wcfg | flag, st), "fh_conv_wino_f32")
            for _ in range(warm):
                pair()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                pair()
            e1.record()
            torch.cuda.synchronize(dev)
        finally:
            hip.check(lib.fh_act_set_blocks_per_cu(before), "fh_act_set_blocks_per_cu")
        return e0.elapsed_time(e1) * 1e3 / reps


def calibrate_act_occupancy(device, force=False, act_blocks=None, bf=False):
    """Choose and set the activation launches' blocks per CU on `device` (once per device and process).
    act_blocks (Vocoder(act_blocks=)) or FH_ACT_BLOCKS = auto | 0 | 2..5 override the measurement -- a deployment that
    knows its boxes, or a launcher that wants every rank alike, passes the number.  No communication here: ranks that want one
    setting call sync_act_blocks() before they build their models (bench.py does).  The measured pair times are logged
    (logger 'flowhigh_amd').  Returns the setting.  Results do not depend on it, only launch times."""
    dev = hip.norm_device(device)
    if dev.type != "cuda" or not torch.cuda.is_available():
        return 0
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (idx, bool(bf))                            # one measured choice per device and conv form (bf: see measure_act_conv_pair)
    fixed = parse_act_blocks(act_blocks)
    if fixed is None:
        fixed = parse_act_blocks(os.environ.get("FH_ACT_BLOCKS"))
    if key in _act_choice and not force and (fixed is None or fixed == _act_choice[key]):
        choice = _act_choice[key]
    elif fixed is not None:
        choice = fixed
    else:
        def measure():                               # two alternating passes: the chip's state drifts over the first 100 ms
            return [{b: measure_act_conv_pair(dev, b, bf=bf) for b in ACT_BLOCKS_CHOICES} for _ in range(2)]
        choice, passes = decide_act_blocks(measure)
        calibrate_act_occupancy.last_measurement = passes
        import logging
        logging.getLogger("flowhigh_amd").info("activation occupancy on cuda:%d (%s convs): %s blocks per CU; (activation, conv) pair us per pass: %s",
                                               idx, "bf16 x 6" if bf else "fp32-MFMA", choice or "uncapped (7)", passes)
    _act_choice[key] = choice
    ensure_act_blocks(dev, choice)
    return choice


def ensure_act_blocks(device, choice):
    """Put `choice` in force on `device` if another one is (the library holds ONE setting per device; models of both conv forms
    in one process -- bench.py builds both -- each ask for theirs in front of their launches: a host-side integer, no GPU work)."""
    dev = hip.norm_device(device)
    if dev.type != "cuda":
        return
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if _act_blocks.get(idx) != choice:
        with hip.device_guard(dev):
            hip.check(hip.lib().fh_act_set_blocks_per_cu(choice), "fh_act_set_blocks_per_cu")
        _act_blocks[idx] = choice


calibrate_act_occupancy.last_measurement = None


def launch_step(voc, s, B, st):
    L = hip.lib()
    if s[0] == "conv":
        _, d, ng, cpad, n_len, tcfg, ck, _flops = s
        timing = voc.conv_timing          # optional list of (start, end) events around conv launches
        if timing is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        hip.check(L.fh_conv_grouped_f32(d.data_ptr(), ng, B, cpad, n_len, tcfg, ck, st), "fh_conv_grouped_f32")
        if timing is not None:
            e1.record()
            timing.append((e0, e1))
    elif s[0] == "wino":
        _, d, ng, wpad, length, dil, _flops, wcfg, pm, bb = s
        timing = voc.conv_timing
        if timing is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if wcfg & WINO_F54:
            hip.check(L.fh_conv_wino54_f32(d.data_ptr(), ng, bb, wpad, length, dil, pm, (wcfg & 15) | voc.wino_flag, st), "fh_conv_wino54_f32")
        else:
            hip.check(L.fh_conv_wino_f32(d.data_ptr(), ng, bb, wpad, length, dil, pm, wcfg | voc.wino_flag, st), "fh_conv_wino_f32")
        if timing is not None:
            e1.record()
            timing.append((e0, e1))
    elif s[0] == "convt":
        _, d, ng, cpad, n_len, tcfg, phases, _flops = s
        timing = voc.conv_timing
        if timing is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        hip.check(L.fh_conv_transpose_fused_f32(d.data_ptr(), ng, B, cpad, n_len, tcfg, phases, st), "fh_conv_transpose_fused_f32")
        if timing is not None:
            e1.record()
            timing.append((e0, e1))
    elif s[0] == "amp":
        _, d, ng, tiles, nt, c, dil, cmax, flags, _flops = s
        timing = voc.conv_timing
        if timing is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if flags & AMP_DIRECT:
            hip.check(L.fh_narrow_conv_bf16x6_f32(d.data_ptr(), ng, tiles.data_ptr(), nt, c, dil, flags & 1, st), "fh_narrow_conv_bf16x6_f32")
        else:
            hip.check(L.fh_amp_actconv_f32(d.data_ptr(), ng, tiles.data_ptr(), nt, c, dil, cmax, flags, st), "fh_amp_actconv_f32")
        if timing is not None:
            e1.record()
            timing.append((e0, e1))
    elif s[0] == "mean":
        _, a, b_, c_, out, n, scale = s
        hip.check(L.fh_mean_f32(a.data_ptr(), b_.data_ptr(), c_.data_ptr() if c_ is not None else None,
                                out.data_ptr(), n, scale, st), "fh_mean_f32")
    elif s[0] == "sum":
        _, srcs, out, n, scale = s
        arr = (C.c_void_p * len(srcs))(*[t.data_ptr() for t in srcs])
        hip.check(L.fh_sum_f32(arr, len(srcs), out.data_ptr(), n, scale, st), "fh_sum_f32")
    elif s[0] == "act":
        _, d, ng, c, length, din, dout = s
        timing = voc.act_timing           # optional list of (start, end) events around activation launches
        if timing is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        hip.check(L.fh_act1d_grouped_pm_f32(d.data_ptr(), ng, B, c, length, din, dout, st), "fh_act1d_grouped_pm_f32")
        if timing is not None:
            e1.record()
            timing.append((e0, e1))
    else:
        _, x, wav, c, length = s
        hip.check(L.fh_conv_post_tanh_f32(x.data_ptr(), voc.post_w.data_ptr(), voc.post_b.data_ptr(),
                                          wav.data_ptr(), B, c, length, voc.post_k, st), "fh_conv_post_tanh_f32")


def run_steps(voc, steps, B, st):
    for s in steps:
        launch_step(voc, s, B, st)


def run_ragged_steps(voc, rp):
    L, st, base = hip.lib(), hip.stream(), rp["desc"].data_ptr()
    for s in rp["steps"]:
        if s[0] == "rwino":
            _, off, ng, wpad, maxlen, dil, wcfg, pmflag, off_map, n_runs = s
            if wcfg & WINO_F54:
                hip.check(L.fh_conv_wino54_ragged_f32(base + off, ng, wpad, maxlen, dil, pmflag, (wcfg & 15) | voc.wino_flag, base + off_map,
                                                      n_runs, st), "fh_conv_wino54_ragged_f32")
            else:
                hip.check(L.fh_conv_wino_ragged_f32(base + off, ng, wpad, maxlen, dil, pmflag, wcfg | voc.wino_flag,
                                                    base + off_map, n_runs, st), "fh_conv_wino_ragged_f32")
        elif s[0] == "rconv":
            _, off, ng, cpad, maxlen, tcfg, ck = s
            hip.check(L.fh_conv_grouped_f32(base + off, ng, 1, cpad, maxlen, tcfg, ck, st), "fh_conv_grouped_f32")
        elif s[0] == "ract":
            _, off, ng, c, din, dout, tiles, mult4 = s
            hip.check(L.fh_act1d_ragged_f32(base + off, ng, c, din, dout, tiles, mult4, st), "fh_act1d_ragged_f32")
        elif s[0] == "rconvt":
            _, off, ng, cpad, maxlen, tcfg, phases = s
            hip.check(L.fh_conv_transpose_fused_f32(base + off, ng, 1, cpad, maxlen, tcfg, phases, st), "fh_conv_transpose_fused_f32")
        elif s[0] == "ramp":
            _, off, ng, off_t, nt, c, dil, cmax, flags = s
            if flags & AMP_DIRECT:
                hip.check(L.fh_narrow_conv_bf16x6_f32(base + off, ng, base + off_t, nt, c, dil, flags & 1, st), "fh_narrow_conv_bf16x6_f32")
            else:
                hip.check(L.fh_amp_actconv_f32(base + off, ng, base + off_t, nt, c, dil, cmax, flags, st), "fh_amp_actconv_f32")
        elif s[0] == "rsum":
            _, off, nj, max_n = s
            hip.check(L.fh_sum_multi_f32(base + off, nj, max_n, st), "fh_sum_multi_f32")
        else:
            launch_step(voc, s, 1, st)
