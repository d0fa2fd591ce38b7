"""`FlowHighSR` -- drop-in host class for the reference's public API on MI355X.

Mirrors /root/reference/src/flowhigh/flowhighsr.py:21-149 (`FlowHighSR`: ctor kwargs,
`generate`, `set_cfm_method`, `from_local`, `from_pretrained`) and the inference half of
/root/reference/src/flowhigh/cfm_superresolution.py:94-284 (`ConditionalFlowMatcherWrapper`:
`sample`, `load`, `device`, `odeint_kwargs`, `sigma`, `cfm_method`).  torchdiffeq's fixed-grid
euler / midpoint steppers (call site cfm:243) are restated in `_integrate`.

Everything numerical runs in the HIP kernels of libflowhigh_hip.so; this file only moves
tensors, picks shapes and sequences launches.  There is no CPU path: constructing the model on
a non-CUDA device, or without the built library, raises.

Extensions over the reference (all keyword-only, defaults keep reference behaviour):
  generate(..., noise=, generator=)   explicit prior draw / CPU generator (parity hook)
  sample(..., cond_scale=, mel_pp=)    as in the reference, evaluated on the device (no python bin loops)
  generate_batch(clips, sr, ...)      B equal-length clips, every per-clip normalisation kept per clip
  upsampling_method='hip'             resample_poly on the device instead of scipy on the host
"""
import json
from pathlib import Path

import os

import numpy as np
import torch

from . import hip
from .flow import FlowNet
from .frontend import LogMel, PostProcessor, Resampler
from .tables import HOP
from .vocoder import VOC, Vocoder, fold_weight_norm

REPO_ID = "ResembleAI/FlowHigh"
# the checkpoint files from_local reads (flowhighsr.py:110-137 of the reference); a weight blob records their digests
CKPT_FILES = ("bigvgan_48khz_256band.json", "bigvgan_48khz_256band.pt", "FLowHigh_basic_400k.pt")


def weights_conv_form():
    """The conv form the environment asks for ('auto' resolved to the default form): what convert.py packs."""
    from .planner import resolve_conv_form
    return resolve_conv_form()[0]


def read_checkpoints(ckpt_dir):
    """(state dict with the wrapper's keys, vocoder JSON) from the reference's three checkpoint files: weight norm folded
    (init_vocoder.py:13-17), key sets checked as load_state_dict(strict=True) would (flowhighsr.py:135)."""
    ckpt_dir = Path(ckpt_dir)
    cfg = json.loads((ckpt_dir / "bigvgan_48khz_256band.json").read_text())
    gen = _load_checkpoint(ckpt_dir / "bigvgan_48khz_256band.pt")['generator']
    sd = {VOC + k: v for k, v in fold_weight_norm(gen).items()}            # init_vocoder.py:13-17
    model = _load_checkpoint(ckpt_dir / "FLowHigh_basic_400k.pt")['model']
    check_state_dict_keys(sd, cfg, only_prefix=VOC)                        # vocoder.load_state_dict (init_vocoder.py:16)
    check_state_dict_keys(model, cfg)                                      # load_state_dict(strict=True), flowhighsr.py:135
    sd.update(model)                                                       # wrapper checkpoint wins
    return sd, cfg
_CFM_METHODS = ("basic_cfm", "independent_cfm_adaptive", "independent_cfm_constant", "independent_cfm_mix")


def reference_prior_draw(n_frames, n_mels=256, generator=None):
    """What `torch.randn_like(cond)` yields in the reference on CPU (cfm_superresolution.py:220):
    `cond` is the 'b d n -> b n d' *view* of the mel (melvoco.py:85); randn_like keeps its strides
    and torch's CPU normal_() takes the scalar path for non-contiguous outputs, so both the fill
    order and the values differ from a contiguous torch.randn(1, N, 256)."""
    t = torch.empty_strided((1, n_frames, n_mels), (n_frames * n_mels, 1, n_frames))
    return t.normal_(generator=generator)


def _load_checkpoint(path):
    """torch.load of a reference checkpoint file.  Only plain state-dict entries are read ('generator', 'model'), so
    the safe unpickler is enough (`weights_only=True`: a downloaded file cannot run code); a checkpoint that carries
    arbitrary pickled objects needs the explicit opt-in FH_UNSAFE_LOAD=1."""
    try:
        return torch.load(str(path), map_location='cpu', weights_only=True)
    except Exception as e:                         # noqa: BLE001  (torch raises pickle.UnpicklingError subclasses)
        if os.environ.get("FH_UNSAFE_LOAD", "0") != "1":
            raise RuntimeError(f"{path}: cannot be read with weights_only=True ({type(e).__name__}: {e}); "
                               "set FH_UNSAFE_LOAD=1 to unpickle it anyway (executes code from the file)") from e
        return torch.load(str(path), map_location='cpu', weights_only=False)


def expected_state_keys(vocoder_cfg, depth=2):
    """Key set of the reference module's state_dict (SURVEY.md 8a "State-dict contract"): what
    `load_state_dict(strict=True)` (flowhighsr.py:135, cfm_superresolution.py:125-131) accepts, no more, no less."""
    fh = "flowhigh."
    keys = [fh + k for k in ("null_cond", "sinu_pos_emb.0.weights", "sinu_pos_emb.1.weight", "sinu_pos_emb.1.bias",
                             "to_embed.weight", "to_embed.bias", "conv_embed.dw_conv1d.0.weight",
                             "conv_embed.dw_conv1d.0.bias", "transformer.rotary_emb.inv_freq",
                             "transformer.final_norm.gamma", "to_pred.weight")]
    for layer in range(depth):
        p = f"{fh}transformer.layers.{layer}."
        for nidx in ("2", "4"):
            keys += [p + f"{nidx}.{w}.{t}" for w in ("to_gamma", "to_beta") for t in ("weight", "bias")]
        keys += [p + "3.q_norm.gamma", p + "3.k_norm.gamma", p + "3.to_qkv.weight", p + "3.to_out.weight",
                 p + "5.0.weight", p + "5.0.bias", p + "5.3.weight", p + "5.3.bias"]
    cfg = vocoder_cfg
    beta = cfg["activation"] == "snakebeta"

    def act(name):
        return [name + "act.alpha"] + ([name + "act.beta"] if beta else []) + \
               [name + "upsample.filter", name + "downsample.lowpass.filter"]

    keys += [VOC + "conv_pre.weight", VOC + "conv_pre.bias", VOC + "conv_post.weight", VOC + "conv_post.bias"]
    keys += [VOC + k for k in act("activation_post.")]
    nk, nm = len(cfg["resblock_kernel_sizes"]), len(cfg["resblock_dilation_sizes"][0])
    for i in range(len(cfg["upsample_rates"])):
        keys += [VOC + f"ups.{i}.0.weight", VOC + f"ups.{i}.0.bias"]
        for j in range(nk):
            r = VOC + f"resblocks.{i * nk + j}."
            if str(cfg["resblock"]) == "1":
                keys += [r + f"{c}.{m}.{t}" for c in ("convs1", "convs2") for m in range(nm) for t in ("weight", "bias")]
                nact = 2 * nm
            else:
                keys += [r + f"convs.{m}.{t}" for m in range(nm) for t in ("weight", "bias")]
                nact = nm
            for a in range(nact):
                keys += [k for k in act(r + f"activations.{a}.")]
    return keys


def check_state_dict_keys(sd, vocoder_cfg, depth=2, only_prefix=None):
    """load_state_dict(strict=True) semantics on the key set: missing AND unexpected keys raise RuntimeError."""
    want = expected_state_keys(vocoder_cfg, depth)
    if only_prefix is not None:
        want = [k for k in want if k.startswith(only_prefix)]
    have = set(sd)
    missing = [k for k in want if k not in have]
    wset = set(want)
    unexpected = [k for k in sd if k not in wset]
    if missing or unexpected:
        def short(v):
            return f"{v[:8]}{' ...' if len(v) > 8 else ''}"
        msg = "Error(s) in loading state_dict:"
        if missing:
            msg += f" Missing key(s) in state_dict: {short(missing)}."
        if unexpected:
            msg += f" Unexpected key(s) in state_dict: {short(unexpected)}."
        raise RuntimeError(msg)


class GraphedGenerate:
    """One captured generate_from_device call (FlowHighSR.capture)."""

    def __init__(self, model, batch, n_in, sr, timestep):
        self.device = dev = model.device
        with hip.device_guard(dev):
            self._capture(model, batch, n_in, sr, timestep, dev)

    def _capture(self, model, batch, n_in, sr, timestep, dev):
        self.x = torch.zeros(batch, n_in, dtype=torch.float32, device=dev)
        t48 = -(-n_in * 48000 // sr)
        self.noise = torch.zeros(batch * (t48 // 480), model.flowhigh.n_mels, dtype=torch.float32, device=dev)
        self.x[:, 0] = 1.0                                  # any non-silent clip: the peak normalisation divides by max |x|
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                       # warm-up outside the capture: plans, workspaces, LDS opt-in
            for _ in range(2):
                model.generate_from_device(self.x, sr, timestep, noise=self.noise)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = model.generate_from_device(self.x, sr, timestep, noise=self.noise)
        # The graph holds raw pointers into the per-shape plans / workspaces of the model (vocoder pool and
        # descriptor arrays, transformer / front-end / post-processing buffers).  Those live in byte-bounded LRU
        # caches (hip.ShapeCache): keep a strong reference to every entry of this shape, so that an eviction only
        # drops the cache's reference and replay() never touches memory that went back to the allocator.
        fh = model.flowhigh
        n = t48 // 480
        # (PostProcessor keys its workspace on the vocoder's output length: hop * n, or a few samples more when some
        # upsampler has an odd k - u: Vocoder.out_len)
        self._keep = [c.get(k) for c, k in ((fh.net._ws, (batch, n)), (fh.logmel._ws, (batch, t48)),
                                            (model.postproc._ws, (batch, fh.vocoder.out_len(n), t48, t48)))]
        # vocoder: the plan of the clip, or the plans of its time chunks (key (batch, chunk frames, clip frames))
        voc_plans = [v for k, v in dict.items(fh.vocoder._plans) if k[0] == batch and k[-1] == n]
        self._keep += voc_plans if voc_plans else [None]
        if any(v is None for v in self._keep):
            raise hip.HipError("capture: a workspace of the captured shape is not in its cache (cache bound too small "
                               "for this shape: raise FH_CACHE_GB)")
        self._keep.append(model)

    @hip.on_device
    def replay(self):
        self.graph.replay()
        return self.out


class FLowHigh:
    """Device-resident weights of the vector-field net + its mel codec (the reference's
    `FLowHigh` with `audio_enc_dec = MelVoco`, models/flow.py:54-142, models/melvoco.py:16-46)."""

    def __init__(self, state_dict, vocoder_config, device="cuda", depth=2, conv_bf16x6=None, store=None, conv_form=None):
        device = torch.device(device)
        if device.type != "cuda":
            raise hip.HipError(f"flowhigh_amd runs on MI355X only (got device '{device}'); there is no CPU path")
        hip.lib()                                   # fail loudly if the extension is not built
        if not torch.cuda.is_available():
            raise hip.HipError("no HIP device visible")
        # 'cuda' = the device current NOW; the model then stays on that ordinal whatever the caller makes current
        # later: every public entry below runs under hip.on_device (the reference's from_local(ckpt_dir, device),
        # flowhighsr.py:110-137)
        self.device = device = hip.norm_device(device)
        self.vocoder_config = dict(vocoder_config)
        # store: a weights.WeightStore opened on a weight blob (state_dict may then be None), or a recording one (convert.py)
        if store is None or state_dict is not None:
            missing = [k for k in ("flowhigh.to_embed.weight", VOC + "conv_pre.weight") if k not in state_dict]
            if missing:
                raise RuntimeError(f"Missing key(s) in state_dict: {missing}")
        with hip.device_guard(device):
            # (the transformer's linears follow the REQUESTED form -- bf16 x 6 linears have no Winograd transform to be ill-conditioned,
            # so a probe that moves the vocoder to the direct form below leaves them as they are)
            from .planner import resolve_conv_form, use_gemm_bf16x6
            self.net = FlowNet(state_dict, device, depth=depth, store=store, bf=use_gemm_bf16x6(resolve_conv_form(conv_form, conv_bf16x6)[0]))
            # conv_form: the arithmetic form of the vocoder's convs, 'auto' | 'winograd' | 'bf16x6' | 'direct'
            # (planner.resolve_conv_form; None: FH_CONV_FORM / the older switches, else 'auto'.  conv_bf16x6: the boolean keyword
            # of rounds 2-5.)  'auto' = the default form, checked once against the direct form through THESE weights when the
            # checkpoint is at hand (probe_conv_form below): a model whose weights amplify the Winograd transforms' rounding is
            # rebuilt in the direct form.
            self.vocoder = Vocoder(self.vocoder_config, state_dict, device, bf16x6=conv_bf16x6, store=store, conv_form=conv_form)
            self.conv_form_probe = None
            if self.vocoder.form_auto and state_dict is not None and os.environ.get("FH_CONV_PROBE", "1") != "0":
                self.probe_conv_form(state_dict)
            self.logmel = LogMel(device)
        self.n_mels = self.net.dim_in

    @property
    def conv_form(self):
        return self.vocoder.form

    def probe_conv_form(self, state_dict, frames=20, limit=None):
        """The load-time estimate behind conv_form='auto': the vocoder in its default form against the direct form (no Winograd
        transform anywhere: its distance to a float64 run is the reference's own fp32 noise, profiles/r05_regime_sweep.txt), both
        through the loaded weights on a `frames`-frame random mel.  max |difference| is logged and kept (conv_form_probe); above
        `limit` (planner.PROBE_LIMIT = 3e-5: a third of the 1e-4 bar) the model keeps the direct-form vocoder instead.
        Oracle-free; costs one packing of the direct-form weights (a reshape) and two 0.2 s forwards."""
        import logging
        from .planner import PROBE_LIMIT
        limit = PROBE_LIMIT if limit is None else limit
        g = torch.Generator().manual_seed(175)
        mel = (torch.randn(1, frames, self.vocoder.true_mels, generator=g) * 2.0 - 3.0).to(self.device)
        direct = Vocoder(self.vocoder_config, state_dict, self.device, conv_form="direct", act_blocks=self.vocoder.act_blocks)
        a = self.vocoder.forward(mel).clone()
        b = direct.forward(mel)
        est = float((a - b).abs().max())
        peak = float(b.abs().max())
        self.conv_form_probe = dict(estimate=est, peak=peak, limit=limit, default=self.vocoder.form, frames=frames)
        log = logging.getLogger("flowhigh_amd")
        if est > limit:
            log.warning("conv_form='auto': the %s form is %.2e from the direct form on a %d-frame probe (|wav| <= %.2f; limit %.0e): "
                        "using the direct form", self.vocoder.form, est, frames, peak, limit)
            self.vocoder = direct
            self.conv_form_probe["chosen"] = "direct"
        else:
            log.info("conv_form='auto': the %s form is %.2e from the direct form on a %d-frame probe (|wav| <= %.2f; limit %.0e): kept",
                     self.vocoder.form, est, frames, peak, limit)
            self.conv_form_probe["chosen"] = self.vocoder.form
            del direct
        return self.conv_form_probe


class FlowHighSR:
    def __init__(
        self,
        flowhigh: FLowHigh,
        sigma=0.,
        ode_atol=1e-5,
        ode_rtol=1e-5,
        use_torchode=False,
        cfm_method='basic_cfm',
        torchdiffeq_ode_method='midpoint',   # [euler, midpoint]
        torchode_method_klass=None,
        cond_drop_prob=0.,
        #
        upsampling_method='scipy',
    ):
        if use_torchode:
            raise NotImplementedError("the torchode adaptive solver path is out of scope (SURVEY.md 8a row 2)")
        self.flowhigh = flowhigh
        self.sigma = sigma
        self.cond_drop_prob = cond_drop_prob
        self.use_torchode = use_torchode
        self.torchode_method_klass = torchode_method_klass
        self.cfm_method = cfm_method
        self.odeint_kwargs = dict(atol=ode_atol, rtol=ode_rtol, method=torchdiffeq_ode_method)
        self.upsampling_method = upsampling_method
        self.postproc = PostProcessor(flowhigh.device)
        self.resampler = Resampler(flowhigh.device)

    # ---- reference surface -----------------------------------------------------------------
    @property
    def device(self):
        return self.flowhigh.device

    def set_cfm_method(self, cfm_method):
        self.cfm_method = cfm_method

    def eval(self):
        return self

    @hip.on_device
    def load(self, path, strict=True):
        path = Path(path)
        assert path.exists()
        pkg = _load_checkpoint(path)
        if strict:
            check_state_dict_keys(pkg['model'], self.flowhigh.vocoder_config)
        self.flowhigh = FLowHigh(pkg['model'], self.flowhigh.vocoder_config, self.device)
        return pkg

    @classmethod
    def from_local(cls, ckpt_dir, device='cuda', conv_form=None, **kwargs) -> 'FlowHighSR':
        """from_local of the reference (flowhighsr.py:110-137) + conv_form = 'auto' (default) | 'winograd' | 'bf16x6' | 'direct': the
        arithmetic form of the vocoder's convs (planner.resolve_conv_form; the environment's
        FH_CONV_FORM overrides nothing a caller passes here)."""
        from .planner import resolve_conv_form
        form, form_auto = resolve_conv_form(conv_form)
        ckpt_dir = Path(ckpt_dir)
        dev = device if torch.device(device).type == 'cuda' else 'cuda'        # the reference always .cuda()s
        # A weight blob next to the checkpoints (python -m flowhigh_amd.convert <ckpt_dir>; FH_BLOB = another path, FH_BLOB=0 =
        # ignore): the packed device weights in one file, mapped and uploaded with one copy -- if it was made from THESE
        # checkpoint files (content digests) under the current layout switches.  Otherwise the checkpoints are read as always.
        from . import weights
        blob = os.environ.get("FH_BLOB", str(ckpt_dir / weights.BLOB_NAME))
        if blob != "0" and Path(blob).exists():
            srcs = {f: weights.file_digest(ckpt_dir / f) for f in CKPT_FILES} if os.environ.get("FH_BLOB_VERIFY", "1") != "0" else None
            # ('auto': a blob of the default form is taken as it is -- the probe needs the checkpoint; a blob written by
            # `python -m flowhigh_amd.convert --probe` on a GPU box already holds the form the probe chose)
            tags = [weights.format_tag(form)] + ([weights.format_tag("direct")] if form_auto else [])
            store, why = None, None
            for tag in tags:
                store = weights.WeightStore.open(blob, hip.norm_device(dev), expect_format=tag, sources=srcs)
                if store is not None:
                    form = json.loads(tag)["form"]
                    break
                why = why or weights.WeightStore.why          # (the reason the blob is not one of the FIRST form asked for)
            if store is None:
                weights.WeightStore.why = why
            if store is not None:
                try:
                    return cls(flowhigh=FLowHigh(None, store.cfg, dev, store=store, conv_form=form), **kwargs)
                except (RuntimeError, KeyError, ValueError) as e:          # a damaged or stale blob must not stop the load
                    weights.WeightStore.why = f"{type(e).__name__}: {e}"
            import logging
            logging.getLogger("flowhigh_amd").warning("weight blob %s not used (%s): reading the checkpoints", blob, weights.WeightStore.why)
        sd, cfg = read_checkpoints(ckpt_dir)
        return cls(flowhigh=FLowHigh(sd, cfg, dev, conv_form="auto" if form_auto else form), **kwargs)

    @classmethod
    def from_pretrained(cls, device='cuda', conv_form=None, **kwargs) -> 'FlowHighSR':
        from huggingface_hub import hf_hub_download
        for fpath in ["FLowHigh_basic_400k.json", "bigvgan_48khz_256band.json",
                      "FLowHigh_basic_400k.pt", "bigvgan_48khz_256band.pt"]:
            local_path = hf_hub_download(repo_id=REPO_ID, filename=fpath)
        return cls.from_local(Path(local_path).parent, device, conv_form=conv_form, **kwargs)

    # ---- host pre-step (flowhighsr.py:59-86) -----------------------------------------------------
    def _upload(self, t):
        """Host tensor -> device through pinned staging, asynchronously: a pageable .to(device) blocks the host
        until everything queued before it on the stream has run, which serialises the host work of the next
        request with the GPU work of the current one (generate_many / the batching server)."""
        if t.device.type != "cpu" or self.device.type != "cuda":
            return t.to(self.device)
        return t.pin_memory().to(self.device, non_blocking=True)

    def _prepare_cond(self, clips, sr, target_sampling_rate):
        """list of 1-D arrays (equal length) -> cond [B, T48] float32 on device, peak-normalised per clip."""
        if target_sampling_rate != 48000:
            raise NotImplementedError("the mel codec is fixed at 48 kHz")
        prepared = []
        for audio in clips:
            if isinstance(audio, torch.Tensor):
                audio = audio.detach().cpu().numpy()
            audio = np.asarray(audio)
            if len(audio.shape) == 2:
                audio = audio.squeeze(0)
            if audio.max() > 1:
                audio = audio / 32768.0
            prepared.append(audio)
        if self.upsampling_method == 'scipy':
            import scipy.signal
            conds = []
            for audio in prepared:
                cond = scipy.signal.resample_poly(audio, target_sampling_rate, sr)
                cond = cond / np.max(np.abs(cond))
                conds.append(torch.tensor(cond).float())
            return self._upload(torch.stack(conds))
        if self.upsampling_method == 'hip':
            x = self._upload(torch.from_numpy(np.stack([a.astype(np.float32) for a in prepared])))
            return self.resampler(x, sr, target_sampling_rate)
        raise UnboundLocalError(f"cond: unsupported upsampling_method '{self.upsampling_method}'")

    # ---- sampler (cfm_superresolution.py:162-284) ------------------------------------------------
    def _draw_noise(self, batch, n_frames, generator):
        n_mels = self.flowhigh.n_mels
        return torch.cat([reference_prior_draw(n_frames, n_mels, generator) for _ in range(batch)], 0)

    def _integrate(self, y0, cond_mel, batch, n, time_steps, cond_scale=1., ragged=None):
        """Fixed-grid euler / midpoint (torchdiffeq semantics); y0, cond_mel [B*n, n_mels] on device.
        Every update `out = base + h * v(x, t)` is the epilogue of the last GEMM of the vector field;
        with classifier-free guidance v = null + s (cond - null) it is two chained epilogues."""
        net = self.flowhigh.net
        method = self.odeint_kwargs['method']
        if method not in ('euler', 'midpoint'):
            raise NotImplementedError(f"ode method '{method}'")
        net.set_cond(cond_mel, batch, n, ragged=ragged)
        t = torch.linspace(0, 1, time_steps + 1)
        y = y0
        bufs = [torch.empty_like(y0) for _ in range(4)]

        def axpy_field(x, tt, out, h, base):           # out = base + h * v(x, tt)
            if cond_scale == 1.:
                net.forward(x, tt, out, batch, n, alpha=h, res=base, ragged=ragged)
            else:
                net.forward(x, tt, bufs[3], batch, n, alpha=h * (1. - cond_scale), res=base, null_cond=True, ragged=ragged)
                net.forward(x, tt, out, batch, n, alpha=h * cond_scale, res=bufs[3], ragged=ragged)

        for i in range(time_steps):
            t0, dt = t[i], t[i + 1] - t[i]
            out = bufs[i % 2]
            if method == 'euler':
                axpy_field(y, float(t0), out, float(dt), y)
            else:
                half = 0.5 * dt
                axpy_field(y, float(t0), bufs[2], float(half), y)
                axpy_field(bufs[2], float(t0 + half), out, float(dt), y)
            y = out
        return y

    @hip.on_device
    def mel_cutoff_bins(self, cond_mel, batch, n):
        """Device version of mel_cutoff_bins (cfm:134-159): int32 [B], no host loop, no sync."""
        L, st = hip.lib(), hip.stream()
        d = cond_mel.shape[-1]
        energy = torch.empty(batch, d, dtype=torch.float32, device=self.device)
        cut = torch.empty(batch, dtype=torch.int32, device=self.device)
        hip.check(L.fh_mel_energy_f32(cond_mel.data_ptr(), energy.data_ptr(), batch, n, d, st), "fh_mel_energy_f32")
        hip.check(L.fh_cutoff_index_f32(energy.data_ptr(), cut.data_ptr(), batch, d, 0.9995, st), "fh_cutoff_index_f32")
        return cut

    def _mel_replace(self, high, low, cut, batch, n):
        out = torch.empty_like(high)
        hip.check(hip.lib().fh_mel_splice_f32(low.data_ptr(), high.data_ptr(), cut.data_ptr(), out.data_ptr(), batch, n,
                                              high.shape[-1], hip.stream()), "fh_mel_splice_f32")
        return out

    @torch.no_grad()
    @hip.on_device
    def sample(self, *, cond=None, cond_mask=None, time_steps=4, cond_scale=1., decode_to_audio=True,
               std_1=None, std_2=None, mel_pp=False, cfm_method=None, noise=None, generator=None):
        """The reference's `sample` (cfm:162-284).  The returned tensor is the caller's own (the vocoder's output
        buffer belongs to a per-shape launch plan and is overwritten by the next call of the same shape, so the
        public entry hands out a copy; `generate*` read the plan's buffer in place)."""
        out = self._sample(cond=cond, cond_mask=cond_mask, time_steps=time_steps, cond_scale=cond_scale,
                           decode_to_audio=decode_to_audio, std_1=std_1, std_2=std_2, mel_pp=mel_pp,
                           cfm_method=cfm_method, noise=noise, generator=generator)
        return out.clone() if decode_to_audio else out

    def _sample(self, *, cond=None, cond_mask=None, time_steps=4, cond_scale=1., decode_to_audio=True,
                std_1=None, std_2=None, mel_pp=False, cfm_method=None, noise=None, generator=None):
        if cfm_method not in _CFM_METHODS:
            cfm_method = self.cfm_method
        if cfm_method in _CFM_METHODS[1:]:
            if std_1 is None or std_2 is None:          # cfm:180-183 (resets BOTH)
                std_1, std_2 = 1.0, self.sigma
        if cond_mask is not None:
            raise NotImplementedError("cond_mask is a training-time option (SURVEY.md 8a row 2)")
        fh = self.flowhigh
        cond = cond.to(self.device, torch.float32)
        if cond.ndim == 2 or (cond.ndim == 3 and cond.shape[1] == 1):      # raw audio (cfm:91-92,185)
            if cond.ndim == 3:
                cond = cond.squeeze(1)
            batch = cond.shape[0]
            cond_mel = fh.logmel(cond)
            n = cond_mel.shape[0] // batch
        else:
            batch, n, _ = cond.shape
            cond_mel = cond.reshape(batch * n, -1).contiguous()
        if noise is None:
            noise = self._draw_noise(batch, n, generator)
        noise = self._upload(noise.to(torch.float32)).reshape(batch * n, -1).contiguous()
        cut = None
        if cfm_method == 'basic_cfm':
            y0 = noise
        else:
            y0 = torch.empty_like(noise)                                    # cond * std_1 + eps * std_2
            hip.check(hip.lib().fh_axpby_f32(cond_mel.data_ptr(), float(std_1), noise.data_ptr(), float(std_2),
                                             y0.data_ptr(), y0.numel(), hip.stream()), "fh_axpby_f32")
            if cfm_method == 'independent_cfm_mix':                         # cfm:231-237
                cut = self.mel_cutoff_bins(cond_mel, batch, n)
                y0 = self._mel_replace(noise, y0, cut, batch, n)
        mel = self._integrate(y0, cond_mel, batch, n, time_steps, float(cond_scale))
        if mel_pp:                                                          # cfm:278-279
            cut = cut if cut is not None else self.mel_cutoff_bins(cond_mel, batch, n)
            mel = self._mel_replace(mel, cond_mel, cut, batch, n)
        mel = mel.view(batch, n, -1)
        if not decode_to_audio:
            return mel
        return fh.vocoder.forward(mel).unsqueeze(1)           # [B, 1, hop * n]

    def _cutoff_bins_seg(self, cond_mel, seg, n_seg):
        """mel_cutoff_bins per clip of a packed batch: int32 [n_seg], two launches for the whole list."""
        L, st = hip.lib(), hip.stream()
        d = cond_mel.shape[-1]
        energy = torch.empty(n_seg, d, dtype=torch.float32, device=self.device)
        cut = torch.empty(n_seg, dtype=torch.int32, device=self.device)
        hip.check(L.fh_mel_energy_seg_f32(cond_mel.data_ptr(), energy.data_ptr(), seg.data_ptr(), n_seg, d, st),
                  "fh_mel_energy_seg_f32")
        hip.check(L.fh_cutoff_index_f32(energy.data_ptr(), cut.data_ptr(), n_seg, d, 0.9995, st), "fh_cutoff_index_f32")
        return cut

    def _mel_replace_seg(self, high, low, cut, seg, n_seg, max_n):
        out = torch.empty_like(high)
        hip.check(hip.lib().fh_mel_splice_seg_f32(low.data_ptr(), high.data_ptr(), cut.data_ptr(), out.data_ptr(),
                                                  seg.data_ptr(), n_seg, max_n, high.shape[-1], hip.stream()),
                  "fh_mel_splice_seg_f32")
        return out

    def _sample_ragged(self, conds, noises, time_steps, cfm_method, std_1=None, std_2=None, mels=None, cond_scale=1.,
                       mel_pp=False, decode_to_audio=True):
        """`sample()` (cfm:162-284, incl. cond_scale != 1 and mel_pp, cfm:162-175,278-279) for clips of DIFFERENT
        lengths as one launch sequence.
        conds: list of [T48_i] device tensors (peak-normalised), noises: list of [1, N_i, n_mels] host tensors.
        Returns the vocoder's waveforms, a list of [1, 480 N_i] (plan-owned buffers; decode_to_audio=False: the mels,
        a list of [N_i, n_mels]), each what _sample gives for that clip alone: the log-mels are made per clip, every
        row-wise operator runs on the packed rows, the operators that look across rows take the clip boundaries (the
        mel cutoff bins are per clip: fh_mel_*_seg_f32), the vocoder runs its merged plan."""
        fh = self.flowhigh
        if cfm_method in _CFM_METHODS[1:]:
            if std_1 is None or std_2 is None:               # cfm:180-183 (resets BOTH; generate() never passes std_1)
                std_1, std_2 = 1.0, self.sigma
        if mels is None:
            mels = [fh.logmel(c[None]) for c in conds]       # [N_i, n_mels] each
        frames = [m.shape[0] for m in mels]
        cond_mel = torch.cat(mels, 0)
        noise = self._upload(torch.cat([z.reshape(-1, z.shape[-1]).to(torch.float32) for z in noises], 0)).contiguous()
        if noise.shape != cond_mel.shape:
            raise ValueError(f"noise rows {tuple(noise.shape)} do not match the clips' frames {tuple(cond_mel.shape)}")
        rws = fh.net.ragged_workspace(frames)
        seg, n_seg, max_n = rws["seg"], len(frames), max(frames)
        cut = None
        if cfm_method == 'basic_cfm':
            y0 = noise
        else:
            y0 = torch.empty_like(noise)
            hip.check(hip.lib().fh_axpby_f32(cond_mel.data_ptr(), float(std_1), noise.data_ptr(), float(std_2),
                                             y0.data_ptr(), y0.numel(), hip.stream()), "fh_axpby_f32")
            if cfm_method == 'independent_cfm_mix':          # per-clip cutoff bins (cfm:231-237)
                cut = self._cutoff_bins_seg(cond_mel, seg, n_seg)
                y0 = self._mel_replace_seg(noise, y0, cut, seg, n_seg, max_n)
        mel = self._integrate(y0, cond_mel, n_seg, max_n, time_steps, float(cond_scale), ragged=rws)
        if mel_pp:                                           # cfm:278-279, per clip
            cut = cut if cut is not None else self._cutoff_bins_seg(cond_mel, seg, n_seg)
            mel = self._mel_replace_seg(mel, cond_mel, cut, seg, n_seg, max_n)
        rows, out = 0, []
        for n in frames:
            out.append(mel[rows:rows + n])
            rows += n
        return fh.vocoder.forward_ragged(out) if decode_to_audio else out

    @torch.no_grad()
    @hip.on_device
    def sample_many(self, conds, *, time_steps=4, cond_scale=1., decode_to_audio=True, std_1=None, std_2=None, mel_pp=False,
                    cfm_method=None, noise=None, generator=None):
        """`sample()` for a LIST of conditioning clips of different lengths (each [T48_i], 48 kHz, peak-normalised) as
        one masked / ragged launch sequence, with the reference's sampler options (cond_scale: classifier-free
        guidance against null_cond, mel_pp: low-band replacement with per-clip cutoff bins, std_1 / std_2: prior scales of the
        independent_cfm_* paths, both reset unless both are given; cfm:162-183,278-279).
        Every result is bit-identical to `sample(cond=clip[None], ...)` on that clip alone.  Returns a list of
        [1, 1, 480 N_i] waveforms (or [1, N_i, n_mels] mels)."""
        if cfm_method not in _CFM_METHODS:
            cfm_method = self.cfm_method
        conds = [c.to(self.device, torch.float32).reshape(-1) for c in conds]
        frames = [c.shape[0] // 480 for c in conds]
        if noise is None:
            noise = [self._draw_noise(1, n, generator) for n in frames]
        outs = self._sample_ragged(conds, noise, time_steps, cfm_method, std_1=std_1, std_2=std_2, cond_scale=cond_scale,
                                   mel_pp=mel_pp, decode_to_audio=decode_to_audio)
        return [o.clone().unsqueeze(1) if decode_to_audio else o.clone()[None] for o in outs]

    @torch.no_grad()
    @hip.on_device
    def generate_batch(self, clips, sr, target_sampling_rate=48000, timestep=1, *, noise=None,
                       generator=None, return_stages=False):
        cond = self._prepare_cond(list(clips), sr, target_sampling_rate)
        kw = dict(std_2=1.) if self.cfm_method == 'independent_cfm_adaptive' else {}
        HR_audio = self._sample(cond=cond, time_steps=timestep, cfm_method=self.cfm_method, noise=noise,
                                generator=generator, **kw)
        HR_audio = HR_audio.squeeze(1)
        out = self.postproc(HR_audio, cond, cond.size(-1), return_cr=return_stages)
        if return_stages:
            return out[0], dict(cond=cond, wav=HR_audio.clone(), cr=out[1].clone())      # (plan-owned buffers)
        return out

    @torch.no_grad()
    @hip.on_device
    def generate_many(self, clips, sr, target_sampling_rate=48000, timestep=1, *, noise=None, generator=None,
                      max_batch=64, streams=None, ragged=None, max_frames=None):
        """Serving-side entry (the gradio caller of app.py:8-26, many requests at once): clips of ANY lengths,
        int16 or float.  Clips of equal length run as one batch (at most max_batch rows), so every result is
        what generate() returns for that clip alone; the prior noise is drawn in the order of `clips`, as a loop
        over generate() would.  noise: optional list of [1, N_i, n_mels] tensors.  Returns a list of [1, T48_i].
        ragged (default on, FH_RAGGED=0 switches it off): clips of different lengths run as ONE launch sequence
        (masked / ragged batch, the reference's mask paths transformer.py:35-44, attend.py:127-128): ~120 launches for
        the whole list instead of ~120 per distinct length; at most max_frames (FH_RAGGED_MAX_FRAMES, default 12 000 =
        120 s of audio) frames per sequence; clips too long for that (or for the unchunked vocoder) run alone.
        Results are bit-identical to generate() per clip either way.
        streams (ragged off): batches of different FRAME COUNTS can be enqueued round-robin on several HIP streams (FH_SERVE_STREAMS,
        default 1), so that the launches of a short clip - a few dozen blocks each, a fraction of the 256 CUs - overlap
        with those of the next one.  The per-shape workspaces are keyed by (batch, frames): two input lengths with the
        same frame count (6000 and 6001 samples at 12 kHz: 50 frames both) share them, so every bucket of one frame
        count runs on the same stream, in order; results do not depend on `streams`.  Measured on a
        mix of 0.5-4 s clips: between -15 % and +40 % of the single-stream time from run to run (the host enqueues
        ~120 launches per clip and is the bottleneck either way), hence off by default."""
        clips = list(clips)
        if noise is None:
            noise = []
            for a in clips:
                n_in = int(np.asarray(a.detach().cpu() if isinstance(a, torch.Tensor) else a).shape[-1])
                t48 = n_in * target_sampling_rate // sr if (n_in * target_sampling_rate) % sr == 0 \
                    else -(-n_in * target_sampling_rate // sr)
                noise.append(self._draw_noise(1, t48 // 480, generator))
        if len(noise) != len(clips):
            raise ValueError("one noise tensor per clip")
        if ragged is None:
            ragged = os.environ.get("FH_RAGGED", "1") != "0"
        lengths = [int(np.asarray(a.detach().cpu() if isinstance(a, torch.Tensor) else a).shape[-1]) for a in clips]
        if ragged and len(set(lengths)) > 1 and target_sampling_rate == 48000:
            try:
                return self._generate_many_ragged(clips, lengths, sr, timestep, noise, max_frames)
            except NotImplementedError as e:
                # a vocoder configuration whose launch positions cannot be merged: one batch per length (said once)
                if not getattr(self, "_ragged_fallback_logged", False):
                    self._ragged_fallback_logged = True
                    import logging
                    logging.getLogger("flowhigh_amd").warning("generate_many: ragged launch sequence not available (%s); "
                                                              "running one batch per clip length", e)
        buckets = {}
        for i, a in enumerate(clips):
            key = (int(np.asarray(a.detach().cpu() if isinstance(a, torch.Tensor) else a).shape[-1]), tuple(noise[i].shape))
            buckets.setdefault(key, []).append(i)
        out = [None] * len(clips)
        if streams is None:
            streams = int(os.environ.get("FH_SERVE_STREAMS", "1"))
        frame_counts = sorted({key[1][1] for key in buckets})
        n_streams = max(1, min(int(streams), len(frame_counts)))
        main = torch.cuda.current_stream(self.device)
        side = self._serve_streams(n_streams) if n_streams > 1 else [main]
        for s_ in side:
            if s_ is not main:
                s_.wait_stream(main)
        stream_of = {n: side[i % len(side)] for i, n in enumerate(frame_counts)}
        for key, idx in buckets.items():
            st = stream_of[key[1][1]]
            with torch.cuda.stream(st):
                for k in range(0, len(idx), max_batch):
                    part = idx[k:k + max_batch]
                    y = self.generate_batch([clips[i] for i in part], sr, target_sampling_rate, timestep,
                                            noise=noise[part[0]] if len(part) == 1 else
                                            torch.cat([noise[i] for i in part], 0))
                    for r, i in enumerate(part):
                        out[i] = y[r:r + 1].clone()
                        if st is not main:
                            out[i].record_stream(main)
        for s_ in side:
            if s_ is not main:
                main.wait_stream(s_)
        return out

    def _generate_many_ragged(self, clips, lengths, sr, timestep, noise, max_frames):
        if max_frames is None:
            max_frames = int(os.environ.get("FH_RAGGED_MAX_FRAMES", "12000"))
        chunk_limit = int(os.environ.get("FH_VOCODER_CHUNK_FRAMES", "6000"))
        frames = [n.shape[1] for n in noise]
        out = [None] * len(clips)
        kw = dict(std_2=1.) if self.cfm_method == 'independent_cfm_adaptive' else {}
        # greedy packing in list order; a clip that does not fit a sequence of its own runs through generate()
        groups, cur, tot = [], [], 0
        for i, n in enumerate(frames):
            if n > max_frames or (chunk_limit > 0 and n > chunk_limit):
                out[i] = self.generate_batch([clips[i]], sr, 48000, timestep, noise=noise[i]).clone()
                continue
            if cur and tot + n > max_frames:
                groups.append(cur)
                cur, tot = [], 0
            cur.append(i)
            tot += n
        if cur:
            groups.append(cur)
        for idx in groups:
            if len(idx) == 1:
                out[idx[0]] = self.generate_batch([clips[idx[0]]], sr, 48000, timestep, noise=noise[idx[0]]).clone()
                continue
            # (the per-clip front and back ends -- ~8 + ~12 small launches per clip -- on up to 4 side streams measured
            # 121.5 ms against 121.3 ms on one stream for the 24-clip mix: not worth the cross-stream bookkeeping)
            conds = [self._prepare_cond([clips[i]], sr, 48000)[0] for i in idx]
            wavs = self._sample_ragged(conds, [noise[i] for i in idx], timestep, self.cfm_method, **kw)
            for i, cond, wav in zip(idx, conds, wavs):
                out[i] = self.postproc(wav, cond[None], cond.shape[0]).clone()
        return out

    def _serve_streams(self, n):
        pool = getattr(self, "_side_streams", None)
        if pool is None or len(pool) < n:
            pool = self._side_streams = [torch.cuda.Stream(self.device) for _ in range(n)]
        return pool[:n]

    @torch.no_grad()
    @hip.on_device
    def generate_from_device(self, x, sr, timestep=1, *, noise):
        """Device-resident variant (no host work, no sync; graph-capturable): x [B, T_in] float32
        low-rate clips already in HBM (|x| <= 1), noise [B, N, n_mels] -> [B, T48].  Same
        arithmetic as generate_batch with upsampling_method='hip'."""
        cond = self.resampler(x, sr, 48000)
        kw = dict(std_2=1.) if self.cfm_method == 'independent_cfm_adaptive' else {}
        wav = self._sample(cond=cond, time_steps=timestep, cfm_method=self.cfm_method, noise=noise, **kw).squeeze(1)
        return self.postproc(wav, cond, cond.size(-1))

    @torch.no_grad()
    @hip.on_device
    def capture(self, batch, n_in, sr, timestep=1):
        """HIP-graph form of generate_from_device for one input shape: the ~150 launches of a call are recorded once
        and replayed with a single enqueue (short clips are launch-bound from Python).  Returns a `GraphedGenerate`
        with static buffers `.x` [batch, n_in] and `.noise` [batch * N, n_mels]; fill them and call `.replay()`."""
        return GraphedGenerate(self, batch, n_in, sr, timestep)

    @torch.no_grad()
    @hip.on_device
    def generate(self, audio, sr: int, target_sampling_rate=48000, timestep=1, *, noise=None, generator=None):
        """One clip, reference contract: returns float32 [1, T48] on the model device."""
        return self.generate_batch([audio], sr, target_sampling_rate, timestep, noise=noise, generator=generator)
