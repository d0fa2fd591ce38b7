"""BigVGAN generator on the HIP kernels: device-resident weights, per-shape launch plans, forward / chunked / ragged runs.

Mirrors /root/reference/src/flowhigh/models/bigvgan/models.py:124-194 (BigVGAN, AMPBlock1 :21-78)
and init_vocoder.py:8-23 (JSON config -> generator), parametric in the JSON keys
`upsample_rates, upsample_kernel_sizes, upsample_initial_channel, num_mels, resblock,
resblock_kernel_sizes, resblock_dilation_sizes, activation, snake_logscale`.

Launch structure per stage (13 launches instead of the reference's ~1700 aten calls):
  1 x grouped conv : ConvTranspose1d as `u` output-phase groups (each a 2-3 tap conv)
  for m in 0..2    : act (nk groups) -> conv1 dilated (nk groups) -> act (nk groups) ->
                     conv2 (+ residual; nk groups)   [models.py:63-72]
  the last conv2 of the stage is ONE group with nk K-segments: it sums the three AMP blocks in
  the accumulator and applies "/ num_kernels" (models.py:181-187) in its epilogue.

Round 5 split this file: weight layouts live in packing.py, kernel / tile selection, the launch-time model and the plan builder
in planner.py, launchers and the step dispatch in runtime.py.  Every name they define is re-exported here (tests and tools use
`vocoder.<name>`; the mutable module-level tables are the same objects).
"""
import json
import math
import os

import torch

from . import hip
from .packing import (  # noqa: F401
    _WINO54_G, _WINO_G, fold_weight_norm, from_phase_major, pack_amp_weight, pack_conv_weight, pack_narrow_bf_weight,
    pack_wino54_weight, pack_wino54_weight_any, pack_wino_weight, pack_wino_weight_any, phase_len, pick_ck, split_bf3, to_phase_major,
    transposed_conv_extra, transposed_conv_phases, wino_phase_weight)
from .planner import (  # noqa: F401
    AMP_DIRECT, AMP_MAX_D, WINO54_MIN_C, WINO_BF16X6, WINO_BM, WINO_F54, WINO_MAX_K, WINO_MIN_C, WINO_NARROW, WINO_NOVL,
    WINO_UPS_MIN_CIN, WINO_UPS_MIN_CIN_BF, WINO_XCD_RANGES, _PlanBuilder, _TILE_PREF, _WINO_BF_SPEED, _WINO_COST, _WINO_RUN,
    _WINO_TILES, _WINO_TILES_OFF, _WINO_WIDE_PANEL_MAX, _addr, _choose_wino_cfg, amp_max_center,
    amp_tile_len, amp_tile_list, choose_wino_cfg, make_act_group, make_amp_group, make_amp_seg, make_conv_group,
    make_conv_seg, make_wino_group, make_wino_seg, merge_ragged, pick_tile_cfg, pick_wino54_tile,
    pick_wino_tile, plan_switches, resolve_conv_form, ups_fused_ok, use_amp, use_amp_bf16x6, use_bf16x6, wino_ups_min_cin, use_wino, use_wino54, wino_block_mapping,
    wino_conv_ok, wino_launch_cost, wino_n_tiles, wino_split_k, wino_split_steps, wino_taps)
from .runtime import (  # noqa: F401
    ACT_BLOCKS_CHOICES, _act_blocks, _act_choice, act1d_grouped, amp_actconv, calibrate_act_occupancy, conv_grouped, ensure_act_blocks,
    conv_wino, decide_act_blocks, launch_step, measure_act_conv_pair, parse_act_blocks, pick_act_blocks,
    run_ragged_steps, run_steps, sync_act_blocks)

VOC = "flowhigh.audio_enc_dec.vocoder."


class Vocoder:
    """Device-resident BigVGAN weights + per-shape launch plans."""

    def __init__(self, cfg, sd, device, prefix=VOC, bf16x6=None, act_blocks=None, store=None, conv_form=None):
        if isinstance(cfg, (str, bytes)) or hasattr(cfg, "read_text"):
            cfg = json.loads(open(cfg).read())
        self.cfg = dict(cfg)
        self.resblock = str(cfg["resblock"])
        if self.resblock not in ("1", "2"):
            raise NotImplementedError(f"resblock {cfg['resblock']!r}")
        if cfg["activation"] not in ("snake", "snakebeta"):
            raise NotImplementedError(cfg["activation"])
        self.device = hip.norm_device(device)
        # conv_form: 'winograd' | 'bf16x6' | 'direct' | 'auto' (planner.resolve_conv_form: keyword > FH_CONV_FORM and the older
        # switches > 'auto' = the default form; bf16x6 = the boolean keyword of rounds 2-5).  'auto' is resolved HERE to the default
        # form; a caller that holds the checkpoint may then probe it against the direct form (FLowHigh, probe_conv_form).
        self.form, self.form_auto = resolve_conv_form(conv_form, bf16x6)
        # bf: the Winograd convs contract on the BF16 matrix cores, operands split into three bf16 pieces
        self.bf = self.form == "bf16x6"
        # ... and the narrow stages (<= 48 channels) run the direct bf16 x 6 kernel (narrow_bf.hip) instead of the fp32 Winograd one
        self.amp_direct = use_amp_bf16x6(self.form)
        # plan-shaping environment switches, read once: every plan of this model uses this snapshot (planner.plan_switches)
        self.sw = plan_switches()
        self.rates = list(cfg["upsample_rates"])
        self.up_k = list(cfg["upsample_kernel_sizes"])
        self.c0 = int(cfg["upsample_initial_channel"])
        self.ks = list(cfg["resblock_kernel_sizes"])
        self.dil = [list(d) for d in cfg["resblock_dilation_sizes"]]
        self.nk = len(self.ks)
        if not 1 <= self.nk <= 12:
            raise NotImplementedError("1 .. 12 resblock kernel sizes")
        self.nm = len(self.dil[0])
        if any(len(d) != self.nm for d in self.dil):
            # (the reference zips convs with dilations per block, models.py:44-58: blocks of different depth would need
            # per-block launch positions)
            raise NotImplementedError("ragged dilation lists")
        if any(k % 2 == 0 for k in self.ks):
            # (get_padding(k, d) = (k d - d) / 2 is "same" padding for odd k only; an even k changes the length inside
            # the residual `+ x` and the reference itself fails there)
            raise NotImplementedError("even resblock kernel sizes")
        self.hop = math.prod(self.rates)
        # Channel counts as the checkpoint has them (models.py:141-146,152: c0 // 2^(i+1)) and as the kernels run them:
        # rounded up to a multiple of 8 (the direct kernel's smallest channel chunk).  The extra channels have zero
        # weights, zero bias and zero conv_post taps, so they carry exact zeros through the network (snake(0) = 0).
        self.true_mels, self.true_c0 = int(cfg["num_mels"]), int(cfg["upsample_initial_channel"])
        self.true_chans = [self.true_c0 // (2 ** (i + 1)) for i in range(len(self.rates))]
        if min(self.true_chans) < 1:
            raise ValueError("upsample_initial_channel is too small for the number of upsampling stages")
        up8 = lambda c: -(-c // 8) * 8
        self.num_mels, self.c0 = up8(self.true_mels), up8(self.true_c0)
        self.chans = [up8(c) for c in self.true_chans]
        # ConvTranspose1d(k, u, padding (k - u) // 2) for any (u, k) (models.py:141-146): k - u odd adds one sample per stage
        self.extra = [transposed_conv_extra(k, u) for u, k in zip(self.rates, self.up_k)]
        dev = self.device
        # every device tensor comes through the weight store: packed here from the checkpoint `sd`, or -- `store` opened on a
        # weight blob (weights.py; `sd` may then be None) -- taken from the uploaded file without touching the checkpoint
        from .weights import WeightStore
        W = store if store is not None else WeightStore(dev)
        if getattr(W, "form", None) not in (None, self.form):
            # (a blob's tensors are laid out for ONE form: fp32 Winograd tensors read as three-piece bf16 would be garbage)
            raise ValueError(f"the weight blob was packed for conv_form={W.form!r}, this model is being built for {self.form!r}")

        def g(name):
            """Checkpoint tensor, channel dimensions zero-padded to the counts above."""
            t = sd[prefix + name].detach().float().cpu()
            parts = name.split(".")
            if parts[0] == "conv_pre" and parts[-1] == "weight":
                want = (self.c0, self.num_mels, t.shape[2])
            elif parts[0] == "conv_pre":
                want = (self.c0,)
            elif parts[0] == "ups" and parts[-1] == "weight":
                i = int(parts[1])
                want = (self.c0 if i == 0 else self.chans[i - 1], self.chans[i], t.shape[2])
            elif parts[0] == "ups":
                want = (self.chans[int(parts[1])],)
            elif parts[0] == "resblocks" and parts[-1] in ("weight", "bias", "alpha", "beta"):
                c = self.chans[int(parts[1]) // self.nk]
                want = (c, c, t.shape[2]) if parts[-1] == "weight" else (c,)
            elif parts[0] == "activation_post" and parts[-1] in ("alpha", "beta"):
                want = (self.chans[-1],)
            elif parts[0] == "conv_post" and parts[-1] == "weight":
                want = (t.shape[0], self.chans[-1], t.shape[2])
            else:
                return t
            if tuple(t.shape) == want:
                return t
            out = torch.zeros(want, dtype=torch.float32)
            if parts[-1] in ("alpha", "beta") and not bool(cfg.get("snake_logscale", False)):
                out.fill_(1.0)            # (linear-scale snake parameters: any finite non-zero value; the channel is 0)
            out[tuple(slice(0, n) for n in t.shape)] = t
            return out

        is_beta = cfg["activation"] == "snakebeta"
        logscale = bool(cfg.get("snake_logscale", False))

        def act_ab(name):
            a = g(name + "act.alpha")
            b = g(name + "act.beta") if is_beta else a
            if logscale:
                a, b = torch.exp(a), torch.exp(b)
            inv_b = 1.0 / (b + 1e-9)                 # activations.py:57,118
            return a.contiguous(), inv_b.contiguous()

        def act_params(name):
            """Activation1d parameters of one site: alpha / inv_beta on the device, the two 12-tap filters as host lists."""
            ab = None

            def both():
                nonlocal ab
                if ab is None:
                    ab = act_ab(name)
                return ab
            return dict(alpha=W.dev("v." + name + "alpha", lambda: both()[0]), inv_beta=W.dev("v." + name + "inv_beta", lambda: both()[1]),
                        up=W.host("v." + name + "up", lambda: g(name + "upsample.filter").flatten()).tolist(),
                        down=W.host("v." + name + "down", lambda: g(name + "downsample.lowpass.filter").flatten()).tolist())

        # conv_pre
        self.pre_cfg, _, self.pre_cpad = pick_tile_cfg(self.c0)
        self.pre_ck = pick_ck(self.num_mels)
        self.pre_w = W.dev("v.conv_pre.w", lambda: pack_conv_weight(g("conv_pre.weight"), self.pre_cpad, self.pre_ck))
        self.pre_b = W.dev("v.conv_pre.b", lambda: g("conv_pre.bias"))
        # conv_pre (num_mels -> c0, 7 taps) in Winograd form as well when the shapes fit
        if sd is not None and g("conv_pre.weight").shape[-1] != 7:
            raise NotImplementedError("conv_pre kernel size other than 7")            # (models.py:134 fixes 7)
        self.pre_u = None
        if use_wino(self.c0, 1, self.form) and self.num_mels % 16 == 0 and self.c0 % 64 == 0:
            self.pre_wcfg, self.pre_wpad = pick_wino_tile(self.c0)
            self.pre_u = W.dev("v.conv_pre.u", lambda: pack_wino_weight_any(g("conv_pre.weight"), self.pre_wpad, self.bf))
        self.stages = []
        for i, (u, k) in enumerate(zip(self.rates, self.up_k)):
            c = self.chans[i]
            cin = self.c0 if i == 0 else self.chans[i - 1]
            tcfg, bm, cpad = pick_tile_cfg(c)
            st = dict(c=c, cin=cin, u=u, k=k, extra=self.extra[i], tile_cfg=tcfg, cpad=cpad,
                      ck=pick_ck(c), up_ck=pick_ck(cin))
            # Winograd residual stack only if every block's kernel fits its 4 tap groups (one launch per position)
            wino_k = max(self.ks) <= WINO_MAX_K
            for kk, dl in zip(self.ks, self.dil):
                if not (wino_k and use_wino(c, 1, self.form)) and (kk > hip.CONV_MAX_TAPS or (kk - 1) * max(dl) > hip.CONV_MAX_HALO):
                    raise NotImplementedError(f"resblock kernel {kk} x dilation {max(dl)} exceeds the direct kernel's "
                                              f"{hip.CONV_MAX_TAPS} taps / {hip.CONV_MAX_HALO} samples of reach")
            # residual stack: F(5,4) kernel from WINO54_MIN_C channels on, else F(4,3); the
            # transposed conv's phase groups always run in the F(4,3) kernel (strided outputs)
            st["up_wcfg"], st["up_wpad"] = pick_wino_tile(c)
            st["w54"] = use_wino54(c, self.form) and max(self.ks) <= WINO_MAX_K
            st["wcfg"], st["wpad"] = pick_wino54_tile(c, self.bf) if st["w54"] else (st["up_wcfg"], st["up_wpad"])
            st["taps"] = 4 if st["w54"] else 3
            # narrow stages (<= 48 channels): the residual-stack convs run on the narrow-stage kernel (planner.use_amp)
            # (a bf16 x 6 model: in the direct bf16 x 6 form, narrow_bf.hip -- self.amp_direct)
            st["amp"] = use_amp(c, self.ks, self.dil, self.form)
            pack_narrow = pack_narrow_bf_weight if self.amp_direct else pack_amp_weight
            ua_key = "unb" if self.amp_direct else "ua"
            pack_res = (lambda w_: pack_wino54_weight_any(w_, st["wpad"], self.bf)) if st["w54"] else \
                (lambda w_: pack_wino_weight_any(w_, st["wpad"], self.bf))
            wt = lambda i=i: g(f"ups.{i}.0.weight")               # [cin, c, k]
            st["up_b"] = W.dev(f"v.ups.{i}.b", lambda: g(f"ups.{i}.0.bias"))
            st["up_phases"] = []
            for r_, taps in enumerate(transposed_conv_phases(k, u)):
                def phase_w(taps=taps):
                    wsel = torch.stack([wt()[:, :, j] for j, _ in taps], dim=-1)        # [cin, c, nt]
                    return pack_conv_weight(wsel.permute(1, 0, 2), cpad, st["up_ck"])
                st["up_phases"].append(dict(w=W.dev(f"v.ups.{i}.phase{r_}.w", phase_w), offs=[o for _, o in taps]))
            # the same transposed conv as Winograd phase groups (strided output) where the tile shapes fit
            st["up_wino"] = None
            if use_wino(max(c, 48), 1, self.form) and c % 16 == 0 and c >= 48 and st["cin"] % 16 == 0 and st["cin"] >= wino_ups_min_cin(self.form):
                st["up_wino"] = []
                for r_, taps in enumerate(transposed_conv_phases(k, u)):
                    # (taps ordered by input offset: a stride-1 correlation of len(taps) taps, center = - smallest offset)
                    st["up_wino"].append(dict(u=W.dev(f"v.ups.{i}.phase{r_}.u", lambda taps=taps: pack_wino_weight_any(
                        wino_phase_weight(wt(), taps)[0], st["up_wpad"], self.bf)), k=len(taps), center=-min(o for _, o in taps)))
            st["blocks"] = []
            for j in range(self.nk):
                r = i * self.nk + j
                blk = dict(k=self.ks[j], dil=self.dil[j], c1=[], c2=[], acts=[])
                if self.resblock == "2":        # AMPBlock2 (models.py:81-121): x = conv_l(act_l(x)) + x per dilation
                    for m in range(self.nm):
                        d = self.dil[j][m]
                        kn = f"resblocks.{r}.convs.{m}"
                        ent = dict(b=W.dev(f"v.{kn}.b", lambda kn=kn: g(kn + ".bias")))
                        w = lambda kn=kn: g(kn + ".weight")
                        if st["amp"] and all(self.dil[jj][m] == d for jj in range(self.nk)):
                            ent["ua"] = W.dev(f"v.{kn}.{ua_key}", lambda w=w: pack_narrow(w(), c))
                        elif wino_k and use_wino(c, d, self.form) and all(self.dil[jj][m] == d for jj in range(self.nk)):
                            ent["u"] = W.dev(f"v.{kn}.u", lambda w=w: pack_res(w()))
                        else:
                            ent["w"] = W.dev(f"v.{kn}.w", lambda w=w: pack_conv_weight(w(), cpad, st["ck"]))
                        blk["c1"].append(ent)
                        blk["acts"].append(act_params(f"resblocks.{r}.activations.{m}."))
                    st["blocks"].append(blk)
                    continue
                for m in range(self.nm):
                    for tag, lst, d in (("convs1", blk["c1"], self.dil[j][m]), ("convs2", blk["c2"], 1)):
                        kn = f"resblocks.{r}.{tag}.{m}"
                        w = lambda kn=kn: g(kn + ".weight")
                        ent = dict(b=W.dev(f"v.{kn}.b", lambda kn=kn: g(kn + ".bias")))
                        # convs1[m] of the nk blocks share one launch: Winograd only if they share the dilation
                        same_d = tag == "convs2" or all(self.dil[jj][m] == d for jj in range(self.nk))
                        if st["amp"] and same_d:
                            ent["ua"] = W.dev(f"v.{kn}.{ua_key}", lambda w=w: pack_narrow(w(), c))
                        elif wino_k and use_wino(c, d, self.form) and same_d:
                            ent["u"] = W.dev(f"v.{kn}.u", lambda w=w: pack_res(w()))
                        else:
                            ent["w"] = W.dev(f"v.{kn}.w", lambda w=w: pack_conv_weight(w(), cpad, st["ck"]))
                        lst.append(ent)
                for a in range(2 * self.nm):
                    blk["acts"].append(act_params(f"resblocks.{r}.activations.{a}."))
                st["blocks"].append(blk)
            # fused last conv2: pre-summed bias
            st["last_bias"] = sum(b["c2" if self.resblock == "1" else "c1"][self.nm - 1]["b"]
                                  for b in st["blocks"]).contiguous()
            self.stages.append(st)
        self.post_act = act_params("activation_post.")
        # (its 24 filter taps on the device as well: the fused tail launch reads them there)
        self.post_taps = W.dev("v.activation_post.taps", lambda: torch.tensor(self.post_act["up"] + self.post_act["down"], dtype=torch.float32))
        self.post_w = W.dev("v.conv_post.w", lambda: g("conv_post.weight")[0])      # [c_last, 7]
        self.post_b = W.dev("v.conv_post.b", lambda: g("conv_post.bias"))
        self.post_k = self.post_w.shape[-1]
        self.wino_flag = WINO_BF16X6 if self.bf else 0             # (the weights above are packed accordingly)
        self._plans = hip.ShapeCache()
        self._ragged = hip.ShapeCache()
        self.conv_timing = None
        self.act_timing = None
        # blocks per CU of the activation launches on this device (measured once per device and process: see above)
        # (act_blocks: 'auto' / None = measure unless FH_ACT_BLOCKS says otherwise; 0 or 2..5 = take that, no measurement)
        self.act_blocks = calibrate_act_occupancy(self.device, act_blocks=act_blocks, bf=self.bf)

    def stage_lengths(self, n_frames):
        """Samples per row after every upsampling stage: L_i = u_i L_(i-1) + (k_i - u_i) % 2 (models.py:141-146,179)."""
        out, n = [], int(n_frames)
        for st in self.stages:
            n = n * st["u"] + st["extra"]
            out.append(n)
        return out

    def out_len(self, n_frames):
        """Waveform samples for n_frames mel frames: hop * n_frames when every k - u is even, a few more otherwise
        (PostProcessing trims to the input's length: postprocessing.py:30-39)."""
        return self.stage_lengths(n_frames)[-1]

    def conv_flops_per_frame(self):
        """Algorithmic FLOPs of the MFMA conv launches per mel frame (SURVEY.md 8d formula:
        2 * Cout * Cin * k per output sample; conv_post and the activations are not included)."""
        f = 2.0 * self.true_c0 * self.true_mels * 7
        length, cin = 1, self.true_c0
        per_dil = 2 if self.resblock == "1" else 1                      # AMPBlock1: convs1 + convs2 per dilation
        for st, c in zip(self.stages, self.true_chans):
            f += 2.0 * cin * c * st["k"] * length                       # transposed conv: per INPUT sample
            length *= st["u"]
            f += length * c ** 2 * 2.0 * per_dil * self.nm * sum(self.ks)
            cin = c
        return f

    @hip.on_device
    def plan(self, batch, n_frames, ref_frames=None, inst=0):
        """Launch plan for [batch, num_mels, n_frames].  ref_frames: the frame count of the WHOLE clip when this
        plan runs a time chunk of it (forward_chunked): every choice that changes the order of additions (input-
        channel slices of short clips, fused / unfused stage-closing conv) is then taken as for the whole clip, so
        a chunk gives the bits of the unchunked run; tile shapes (no effect on the arithmetic) follow the chunk."""
        ref_frames = n_frames if ref_frames is None else ref_frames
        key = (batch, n_frames) if ref_frames == n_frames else (batch, n_frames, ref_frames)
        if inst:                    # (plan_ragged: clips of equal length in one ragged batch need buffers of their own)
            key = (batch, n_frames, "inst", inst, ref_frames)
        if key in self._plans:
            return self._plans[key]
        pb = _PlanBuilder(self, batch, n_frames, ref_frames)
        cur = pb.conv_pre()
        for i in range(len(self.stages)):
            pb.enter_stage(i)
            pb.upsampler(i, cur)
            if self.resblock == "1":
                pb.amp1_stack(i)
            else:
                pb.amp2_stack(i)
            cur = pb.S                  # slot roles repeat every stage: the next up-conv reads S (slot 1), writes X (slot 0)
        p = pb.finish(cur)
        self._plans[key] = p
        return p

    # ---- ragged batches (clips of different lengths in ONE launch sequence) ------------------------
    @hip.on_device
    def plan_ragged(self, frames):
        """Merged launch plan for clips of frame counts `frames` (any mix of lengths, batch 1 each): planner.merge_ragged."""
        return merge_ragged(self, frames)

    @hip.on_device
    def run_ragged(self, rp):
        ensure_act_blocks(self.device, self.act_blocks)
        run_ragged_steps(self, rp)

    @hip.on_device
    def forward_ragged(self, mels):
        """mels: list of [N_i, num_mels] (token-major rows of each clip) -> list of wav [1, hop * N_i] (plan-owned
        buffers), every one bit-identical to forward() on that clip alone."""
        rp = self.plan_ragged([m.shape[0] for m in mels])
        for sp, m in zip(rp["subs"], mels):
            sp["mel_in"][:, :self.true_mels].copy_(m.view(1, m.shape[0], -1).transpose(1, 2))
        self.run_ragged(rp)
        return [sp["wav"] for sp in rp["subs"]]

    # ---- time-chunked execution (streaming / chunked vocoder) ----------------------------------
    def chunk_geometry(self):
        """(halo, align) in mel frames for time-chunked execution.
        halo: how far (in frames, rounded up) a waveform sample looks into the mel: BigVGAN is purely local
        (bigvgan/models.py:172-194) -- conv_pre 3 taps a side, per stage the transposed conv and three AMP blocks of
        (anti-aliased activation 6 | conv (k-1)/2 d | activation 6 | conv (k-1)/2) per dilation, then activation +
        conv_post -- plus the numerical reach of the Winograd tiles (below).  A chunk run with `halo` extra frames on each inner side reproduces the whole-clip values in
        its middle: the zero / replicate padding of a chunk edge only reaches samples that are thrown away.
        align: chunk starts are multiples of it, so that every sample keeps its position inside its Winograd
        F(4,3) tile and its dilation phase at every stage -- the per-sample arithmetic is then the same instruction
        sequence as in the whole-clip run and the result is bit-identical, not just close."""
        kmax = max(self.ks)
        dmax = [max(dl[m] for dl in self.dil) for m in range(self.nm)]
        per_stage = sum(12 + (kmax - 1) // 2 * (d + 1) for d in dmax)
        if self.resblock == "2":
            per_stage = sum(6 + (kmax - 1) // 2 * d for d in dmax)
        # Numerical reach of a Winograd conv: an output's value depends on its k taps only, but its ROUNDING depends
        # on every input of its F(4,3) tile (the other inputs cancel in exact arithmetic, not in floating point): up to
        # 3 more (decimated) samples per side and conv.  Such rounding-level influence is almost always absorbed by the
        # next layer's own rounding -- chunks with the taps-only halo matched bit for bit in every fp32 test -- but
        # "almost" is not the contract (and the bf16 x 6 form, with more rounding steps, did show it): the halo covers it.
        # (F(4,3): 6 inputs per 4 outputs, 3 beyond the taps.  F(5,4): 8 per 5; its odd kernels are packed into ceil(k / 4) groups of
        # 4 taps with the zero tap at the END, so a tile reads inputs 5 n - center .. 5 n + 4 ngrp + 3 - center: output 5 n's rounding
        # depends on up to 5 samples past its last real tap on the right (4 on the left): 5 per conv, not 4 -- ADVICE r05)
        reach1 = sum((d + 1) if self.resblock == "1" else d for d in dmax)
        h = 3.0 + 6.0                                   # conv_post (7 taps) + activation_post, in output samples
        for i in reversed(range(len(self.rates))):
            h += per_stage + (5 if self.stages[i]["w54"] or self.stages[i]["amp"] else 3) * reach1      # residual stack at this stage's rate
            h = h / self.rates[i] + self.up_k[i] / self.rates[i] + 1.0 + 3.0     # ... seen from the transposed conv's input
        h += 3.0 + 3.0                                  # conv_pre
        align, rate = 4, 1
        lcm = lambda a, b: a * b // math.gcd(a, b)
        dl = 1
        for dils in self.dil:
            for d in dils:
                dl = lcm(dl, d)
        for i, u in enumerate(self.rates):
            align = lcm(align, 4 // math.gcd(4, rate))                      # Winograd transposed conv reads at `rate`
            rate *= u
            m = 5 if self.stages[i]["w54"] or self.stages[i]["amp"] else 4  # outputs per tile of the stage's Winograd kernel
            align = lcm(align, m * dl // math.gcd(m * dl, rate))            # residual stack at `rate` samples per frame
        halo = -(-int(math.ceil(h)) // align) * align
        return halo, align

    @hip.on_device
    def forward_chunks(self, mel_bnd, chunk_frames):
        """Generator over time chunks of forward(mel): yields (first_sample, wav_chunk [B, n_samples]) in order, each
        chunk bit-identical to the same samples of the whole-clip run; workspace is O(chunk_frames + 2 halo).
        The yielded tensor is the plan's buffer view: consume (copy) it before the next iteration."""
        B, N, D = mel_bnd.shape
        halo, align = self.chunk_geometry()
        step = max(align, chunk_frames // align * align)
        s = 0
        while s < N:
            e = min(N, s + step)
            a, b = max(0, s - halo), min(N, e + halo)
            p = self.plan(B, b - a, ref_frames=N)
            p["mel_in"][:, :self.true_mels].copy_(mel_bnd[:, a:b].transpose(1, 2))
            self.run(p)
            # (the last chunk also carries the samples past hop * N of a vocoder with odd k - u: out_len)
            yield s * self.hop, p["wav"][:, (s - a) * self.hop:((e - a) * self.hop if e < N else p["wav"].shape[1])]
            s = e

    @hip.on_device
    def forward_chunked(self, mel_bnd, chunk_frames, out=None):
        """forward() in time chunks of `chunk_frames` mel frames: same bits, bounded workspace."""
        B, N, D = mel_bnd.shape
        wav = out if out is not None else torch.empty(B, self.out_len(N), dtype=torch.float32, device=self.device)
        for first, w in self.forward_chunks(mel_bnd, chunk_frames):
            wav[:, first:first + w.shape[1]].copy_(w)
        return wav

    @hip.on_device
    def forward(self, mel_bnd):
        """mel [B, N, num_mels] (token-major, as the sampler produces it) -> wav [B, hop * N].  Clips longer than
        FH_VOCODER_CHUNK_FRAMES (default 6000 = 60 s) run in time chunks of that many frames: same result, the
        workspace (645 MB per 10 s of audio and clip) stops growing with the clip length."""
        B, N, D = mel_bnd.shape
        limit = int(os.environ.get("FH_VOCODER_CHUNK_FRAMES", "6000"))
        if limit > 0 and N > limit + 2 * self.chunk_geometry()[0]:
            return self.forward_chunked(mel_bnd, limit)
        p = self.plan(B, N)
        p["mel_in"][:, :self.true_mels].copy_(mel_bnd.transpose(1, 2))       # 'b n d -> b d n' (melvoco.py:115); layout only
        self.run(p)
        return p["wav"]

    @hip.on_device
    def run(self, p):
        ensure_act_blocks(self.device, self.act_blocks)       # (this model's activation occupancy cap: runtime.ensure_act_blocks)
        run_steps(self, p["steps"], p["B"], hip.stream())
