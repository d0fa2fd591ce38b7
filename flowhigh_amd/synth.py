"""Seeded synthetic checkpoints and inputs.

The reference's trained weights and its BigVGAN JSON are Hugging Face artefacts
(`ResembleAI/FlowHigh`, /root/reference/src/flowhigh/flowhighsr.py:18,141-147)
that are not reachable without a network, so benchmarks, tests and the smoke
run use random-init weights of the same architecture, emitted under the
reference's own state-dict keys (SURVEY.md section 8a "State-dict contract").

Every tensor is drawn from a Philox stream keyed by crc32(tensor name) ^ seed,
so the values do not depend on generation order and are bit-identical on any
box with the same numpy.
"""
import json
import math
import zlib
from pathlib import Path

import numpy as np
import torch

# Declared synthetic vocoder config (SURVEY.md section 8d "SYNTH-CFG"): the real
# bigvgan_48khz_256band.json is not in the reference tree; constraints from the
# reference are num_mels = 256 and prod(upsample_rates) = hop = 480.
SYNTH_CFG = {
    "resblock": "1",
    "upsample_rates": [5, 4, 3, 2, 2, 2],
    "upsample_kernel_sizes": [11, 8, 7, 4, 4, 4],
    "upsample_initial_channel": 1536,
    "resblock_kernel_sizes": [3, 7, 11],
    "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]],
    "activation": "snakebeta",
    "snake_logscale": True,
    "num_mels": 256,
    "sampling_rate": 48000,
}

# Reduced configs for parity tests (same code path, seconds on CPU).
TINY_CFG = dict(SYNTH_CFG, upsample_initial_channel=512)
ALT_CFG = dict(SYNTH_CFG, upsample_rates=[8, 6, 5, 2], upsample_kernel_sizes=[16, 12, 11, 4],
               upsample_initial_channel=128, resblock_kernel_sizes=[3, 5],
               resblock_dilation_sizes=[[1, 2, 4], [1, 3, 5]], activation="snake",
               snake_logscale=False)

# AMPBlock2 variant (resblock "2"): channel counts 192 / 96 / 48 / 24 cover both Winograd tiles and the direct
# kernel; the second dilation differs between the two blocks (mixed-dilation launch, unfused closing)
ALT2_CFG = dict(SYNTH_CFG, resblock="2", upsample_rates=[8, 6, 5, 2], upsample_kernel_sizes=[16, 12, 11, 4],
                upsample_initial_channel=384, resblock_kernel_sizes=[3, 7],
                resblock_dilation_sizes=[[1, 3], [1, 5]])
# ... and with 3 blocks sharing the dilations (fused closing conv with dilation 3)
ALT3_CFG = dict(ALT2_CFG, resblock_kernel_sizes=[3, 5, 7],
                resblock_dilation_sizes=[[1, 3], [1, 3], [1, 3]])

# The upstream BigVGAN convention k = 2 u on the ODD rates a hop of 480 = 2^5 * 3 * 5 forces: k - u is odd for the
# u = 5 and u = 3 stages, ConvTranspose1d(k, u, padding (k - u) // 2) then returns u L + 1 samples (models.py:141-146),
# the waveform has 480 N + 98 samples and PostProcessing trims it (postprocessing.py:30-39).  Channel counts
# 384 / 192 / 96 / 48 / 24 cover the Winograd upsampler, every Winograd tile and the direct kernel at those lengths.
ODD_CFG = dict(SYNTH_CFG, upsample_rates=[5, 4, 4, 3, 2], upsample_kernel_sizes=[10, 8, 8, 6, 4],
               upsample_initial_channel=768)
# four kernel sizes (models.py:130,182-187 take any count): the stage-closing conv cannot be one 3-segment group
NK4_CFG = dict(SYNTH_CFG, upsample_rates=[8, 6, 5, 2], upsample_kernel_sizes=[16, 12, 11, 4],
               upsample_initial_channel=384, resblock_kernel_sizes=[3, 5, 7, 11],
               resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5], [1, 3, 5]])
# ... and as AMPBlock2 with five
NK5_AMP2_CFG = dict(NK4_CFG, resblock="2", resblock_kernel_sizes=[3, 5, 7, 9, 11],
                    resblock_dilation_sizes=[[1, 3]] * 5)
# channel counts that are not multiples of 8 (200 -> 100 / 50 / 25 / 12), an odd k - u, one kernel size,
# a 13-tap kernel (longer than the Winograd form's 4 tap groups)
PAD_CFG = dict(SYNTH_CFG, upsample_rates=[8, 6, 5, 2], upsample_kernel_sizes=[16, 13, 11, 4],
               upsample_initial_channel=200, resblock_kernel_sizes=[13], resblock_dilation_sizes=[[1, 2, 5]])

VOC = "flowhigh.audio_enc_dec.vocoder."
FH = "flowhigh."


def _rng(name, seed):
    return np.random.Generator(np.random.Philox(key=(zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0xFFFFFFFFFFFFFFFF))


def _normal(name, shape, std, seed, mean=0.0):
    a = _rng(name, seed).standard_normal(size=shape, dtype=np.float32)
    return torch.from_numpy(a * np.float32(std) + np.float32(mean))


def _uniform(name, shape, bound, seed):
    a = _rng(name, seed).random(size=shape, dtype=np.float32)
    return torch.from_numpy((a * 2.0 - 1.0) * np.float32(bound))


def kaiser_sinc_filter(cutoff=0.25, half_width=0.3, kernel_size=12):
    """12-tap kaiser-windowed sinc of the anti-aliased activation
    (/root/reference/src/flowhigh/models/bigvgan/alias_free_torch/filter.py:28-57);
    stored as a buffer in real checkpoints, so only synthetic ones need this."""
    half = kernel_size // 2
    delta_f = 4 * half_width
    a = 2.285 * (half - 1) * math.pi * delta_f + 7.95
    if a > 50.0:
        beta = 0.1102 * (a - 8.7)
    elif a >= 21.0:
        beta = 0.5842 * (a - 21) ** 0.4 + 0.07886 * (a - 21.0)
    else:
        beta = 0.0
    window = torch.kaiser_window(kernel_size, beta=beta, periodic=False)
    time = torch.arange(-half, half) + 0.5 if kernel_size % 2 == 0 else torch.arange(kernel_size) - half
    filt = 2 * cutoff * window * torch.sinc(2 * cutoff * time)
    filt = filt / filt.sum()
    return filt.view(1, 1, kernel_size).float()


def vocoder_channels(cfg):
    c0 = cfg["upsample_initial_channel"]
    return [c0 // (2 ** (i + 1)) for i in range(len(cfg["upsample_rates"]))]


def make_vocoder_state_dict(cfg, seed=0, prefix=VOC, convs2_gain=0.2, snake_bound=0.5, post_gain=0.3):
    """BigVGAN generator tensors (weight-norm already folded, as in the wrapper checkpoint).

    convs2_gain / snake_bound / post_gain: the knobs of the regime sweep (tests/tools/regime_sweep.py: how much error
    headroom the Winograd forms keep as the residual stack's gain, the spread of the snake parameters and the output scale
    grow towards what trained weights may have); the defaults are the regime every other test and the benchmark use.

    Weight regime (SURVEY.md 8c regime iii, chosen so that end-to-end parity is *sensitive*:
    with the reference's N(0, 0.01) init the output barely depends on the mel): conv_pre,
    upsamplers and convs1 are variance preserving (std = g / sqrt(fan_in)), convs2 are damped
    so that the residual stack stays well conditioned in fp32, snake parameters are random
    per channel, conv_post keeps |pre-tanh| below ~0.5."""
    sd = {}
    c0 = cfg["upsample_initial_channel"]
    is_beta = cfg["activation"] == "snakebeta"
    logscale = bool(cfg.get("snake_logscale", False))
    filt = kaiser_sinc_filter()

    def conv(name, co, ci, k, gain):
        sd[prefix + name + ".weight"] = _normal(prefix + name + ".weight", (co, ci, k), gain / math.sqrt(ci * k), seed)
        sd[prefix + name + ".bias"] = _uniform(prefix + name + ".bias", (co,), 0.05, seed)

    def act(name, c):
        # log-scale params in +-0.5 (alpha, beta in [0.6, 1.65]); linear-scale in [0.6, 1.6]
        lo = _uniform(prefix + name + "act.alpha", (c,), snake_bound, seed)
        sd[prefix + name + "act.alpha"] = lo if logscale else lo + 1.1
        if is_beta:
            lb = _uniform(prefix + name + "act.beta", (c,), snake_bound, seed)
            sd[prefix + name + "act.beta"] = lb if logscale else lb + 1.1
        sd[prefix + name + "upsample.filter"] = filt.clone()
        sd[prefix + name + "downsample.lowpass.filter"] = filt.clone()

    conv("conv_pre", c0, cfg["num_mels"], 7, 0.4)       # log-mel spans ~[-11.5, 2]
    chans = vocoder_channels(cfg)
    nk = len(cfg["resblock_kernel_sizes"])
    for i, (u, k) in enumerate(zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"])):
        cin = c0 // (2 ** i)
        name = f"ups.{i}.0"
        sd[prefix + name + ".weight"] = _normal(prefix + name + ".weight", (cin, chans[i], k),
                                                1.0 / math.sqrt(cin * k / u), seed)
        sd[prefix + name + ".bias"] = _uniform(prefix + name + ".bias", (chans[i],), 0.05, seed)
        for j in range(nk):
            r = i * nk + j
            ks = cfg["resblock_kernel_sizes"][j]
            if str(cfg["resblock"]) == "2":       # AMPBlock2: one conv + one activation per dilation
                for m in range(len(cfg["resblock_dilation_sizes"][j])):
                    conv(f"resblocks.{r}.convs.{m}", chans[i], chans[i], ks, 0.4)
                    act(f"resblocks.{r}.activations.{m}.", chans[i])
                continue
            for m in range(len(cfg["resblock_dilation_sizes"][j])):
                conv(f"resblocks.{r}.convs1.{m}", chans[i], chans[i], ks, 1.0)
                conv(f"resblocks.{r}.convs2.{m}", chans[i], chans[i], ks, convs2_gain)
            for a in range(2 * len(cfg["resblock_dilation_sizes"][j])):
                act(f"resblocks.{r}.activations.{a}.", chans[i])
    act("activation_post.", chans[-1])
    conv("conv_post", 1, chans[-1], 7, post_gain)
    return sd


def make_flow_state_dict(seed=0, dim=1024, dim_in=256, depth=2, heads=16, dim_head=64,
                         ff_mult=4, conv_k=31):
    """FLowHigh (transformer) tensors with reference shapes (flow.py:92-142, transformer.py,
    attend.py:144-171).  Linear inits follow torch defaults (uniform +-1/sqrt(fan_in)); the
    adaptive-norm projections, which the reference initialises to identity, get small random
    weights here so that the time conditioning is exercised."""
    sd = {}

    def lin(name, out_f, in_f, bias=True, wscale=1.0, bias_mean=0.0):
        b = 1.0 / math.sqrt(in_f)
        sd[name + ".weight"] = _uniform(name + ".weight", (out_f, in_f), b * wscale, seed)
        if bias:
            sd[name + ".bias"] = _uniform(name + ".bias", (out_f,), b, seed) + bias_mean

    sd[FH + "null_cond"] = _normal(FH + "null_cond", (dim_in,), 0.5, seed)   # zeros in the reference init
    sd[FH + "sinu_pos_emb.0.weights"] = _normal(FH + "sinu_pos_emb.0.weights", (dim // 2,), 1.0, seed)
    lin(FH + "sinu_pos_emb.1", dim, dim)
    lin(FH + "to_embed", dim, dim_in * 2)
    sd[FH + "conv_embed.dw_conv1d.0.weight"] = _uniform(FH + "conv_embed.dw_conv1d.0.weight", (dim, 1, conv_k), 1.0 / math.sqrt(conv_k), seed)
    sd[FH + "conv_embed.dw_conv1d.0.bias"] = _uniform(FH + "conv_embed.dw_conv1d.0.bias", (dim,), 1.0 / math.sqrt(conv_k), seed)
    inner = int(dim * ff_mult * 2 / 3)
    for layer in range(depth):
        p = f"{FH}transformer.layers.{layer}."
        for nidx in ("2", "4"):
            lin(p + nidx + ".to_gamma", dim, dim, wscale=0.3, bias_mean=1.0)
            lin(p + nidx + ".to_beta", dim, dim, wscale=0.3)
        sd[p + "3.q_norm.gamma"] = 1.0 + _normal(p + "3.q_norm.gamma", (heads, 1, dim_head), 0.1, seed)
        sd[p + "3.k_norm.gamma"] = 1.0 + _normal(p + "3.k_norm.gamma", (heads, 1, dim_head), 0.1, seed)
        lin(p + "3.to_qkv", heads * dim_head * 3, dim, bias=False)
        lin(p + "3.to_out", dim, heads * dim_head, bias=False)
        lin(p + "5.0", inner * 2, dim)
        lin(p + "5.3", dim, inner)
    sd[FH + "transformer.rotary_emb.inv_freq"] = 1.0 / (50000 ** (torch.arange(0, dim_head, 2).float() / dim_head))
    sd[FH + "transformer.final_norm.gamma"] = 1.0 + _normal(FH + "transformer.final_norm.gamma", (dim,), 0.1, seed)
    lin(FH + "to_pred", dim_in, dim, bias=False)
    return sd


def make_state_dict(cfg=None, seed=0):
    """Full `FLowHigh_basic_400k.pt['model']`-shaped state dict."""
    cfg = cfg or SYNTH_CFG
    sd = make_flow_state_dict(seed)
    sd.update(make_vocoder_state_dict(cfg, seed))
    return sd


def write_checkpoint_dir(path, cfg=None, seed=0, weight_norm=True):
    """Write the three files `FlowHighSR.from_local` reads (flowhighsr.py:110-137).
    With weight_norm=True the vocoder file holds weight_g / weight_v pairs like a real
    BigVGAN generator checkpoint (init_vocoder.py:13-17)."""
    cfg = cfg or SYNTH_CFG
    path = Path(path)
    path.mkdir(parents=True, exist_ok=True)
    sd = make_state_dict(cfg, seed)
    (path / "bigvgan_48khz_256band.json").write_text(json.dumps(cfg))
    gen = {}
    for k, v in sd.items():
        if not k.startswith(VOC):
            continue
        name = k[len(VOC):]
        is_conv_w = name.endswith(".weight") and v.ndim == 3 and "filter" not in name
        if weight_norm and is_conv_w:
            norm = v.flatten(1).norm(dim=1).view(-1, 1, 1)      # norm over all dims but 0
            gen[name[:-len("weight")] + "weight_g"] = norm.clone()
            gen[name[:-len("weight")] + "weight_v"] = v * 1.7  # any positive rescale folds back
        else:
            gen[name] = v
    torch.save({"generator": gen}, path / "bigvgan_48khz_256band.pt")
    torch.save({"model": sd, "optim": {}, "scheduler": {}}, path / "FLowHigh_basic_400k.pt")
    return sd


def lowres_clip(i, seconds, sr_in):
    """Synthetic low-rate clip i: 0.1 * N(0,1), float32."""
    n = int(round(seconds * sr_in))
    return (0.1 * np.random.default_rng(1000 + i).standard_normal(n)).astype(np.float32)


def prior_noise(i, n_frames, n_mels=256):
    """Prior draw for clip i from the torch CPU generator (parity contract)."""
    from .flowhighsr import reference_prior_draw
    g = torch.Generator().manual_seed(2000 + i)
    return reference_prior_draw(n_frames, n_mels, g)
