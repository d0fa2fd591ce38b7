"""CPU restatement of FlowHighSR.generate() (oracle; TEST INFRASTRUCTURE ONLY).

Plain PyTorch-CPU fp32 functional ops in the same sequence as the reference,
driven by a state-dict that uses the reference's own key names
(`flowhigh.*`, `flowhigh.audio_enc_dec.vocoder.*`).  No
reference code is imported here; every function cites the reference lines it
follows (paths relative to /root/reference/src/flowhigh/).

Validated against the reference imported under shims by
`oracle/make_golden.py` (vectors in tests/golden/) -- see oracle/__init__.py.
"""
import math

import numpy as np
import scipy.signal
import torch
import torch.nn.functional as F

from .slaney import mel_filter_bank

VOC = "flowhigh.audio_enc_dec.vocoder."
FH = "flowhigh."

N_FFT = 2048
HOP = 480
N_MELS = 256
SR = 48000


# ----------------------------------------------------------------------------
# host pre-step                                            flowhighsr.py:59-72
# ----------------------------------------------------------------------------
def preprocess(audio, sr, target_sr=SR):
    """int16-scale detection, scipy polyphase resampling, peak normalise -> [1,T48] f32."""
    audio = np.asarray(audio)
    if audio.ndim == 2:
        audio = audio.squeeze(0)                      # :59-60
    if audio.max() > 1:                               # signed max, not abs (:62)
        audio = audio / 32768.0
    cond = scipy.signal.resample_poly(audio, target_sr, sr)   # :68
    cond = cond / np.max(np.abs(cond))                # :69
    return torch.tensor(cond).unsqueeze(0).float()    # :71, :86


# ----------------------------------------------------------------------------
# mel front end                                        models/melvoco.py:56-86
# ----------------------------------------------------------------------------
_MEL = None


def mel_basis():
    global _MEL
    if _MEL is None:
        _MEL = torch.from_numpy(mel_filter_bank(SR, N_FFT, N_MELS, 20.0, 24000.0))
    return _MEL


def logmel(audio):
    """audio [B,T] -> log-mel [B,N,256]; N = T // 480."""
    pad = (N_FFT - HOP) // 2                                          # melvoco.py:74
    x = F.pad(audio.unsqueeze(1), (pad, pad), mode="reflect").squeeze(1)
    spec = torch.stft(x, N_FFT, hop_length=HOP, win_length=N_FFT,
                      window=torch.hann_window(N_FFT, dtype=x.dtype), center=False,
                      normalized=False, onesided=True, return_complex=True)  # :78-79
    spec = torch.view_as_real(spec)
    spec = torch.sqrt(spec.pow(2).sum(-1) + 1e-9)                     # :81
    spec = torch.matmul(mel_basis().to(spec.dtype), spec)                            # :83
    spec = torch.log(torch.clamp(spec, min=1e-5))                     # modules.py:31-36
    return spec.transpose(1, 2).contiguous()                          # :85


# ----------------------------------------------------------------------------
# FLowHigh vector field                                   models/flow.py:185-261
# ----------------------------------------------------------------------------
def _rmsnorm_dir(x):
    return F.normalize(x, dim=-1)                     # x / max(||x||, 1e-12)


def ada_rmsnorm(sd, prefix, x, t_emb):
    """transformer.py:61-88"""
    dim = x.shape[-1]
    normed = _rmsnorm_dir(x) * (dim ** 0.5)
    gamma = F.linear(t_emb, sd[prefix + "to_gamma.weight"], sd[prefix + "to_gamma.bias"])
    beta = F.linear(t_emb, sd[prefix + "to_beta.weight"], sd[prefix + "to_beta.bias"])
    return normed * gamma[:, None, :] + beta[:, None, :]


def rotary_table(sd, n):
    """pos_emb.py:44-51 -- fp32 product of fp32 position and the state-dict inv_freq."""
    inv_freq = sd[FH + "transformer.rotary_emb.inv_freq"].float()
    t = torch.arange(n).type_as(inv_freq)
    freqs = torch.einsum("i,j->ij", t, inv_freq)
    return torch.cat((freqs, freqs), dim=-1)          # [n, 64]


def _rotate_half(x):
    x1, x2 = x.chunk(2, dim=-1)
    return torch.cat((-x2, x1), dim=-1)               # pos_emb.py:53-55


def attention(sd, prefix, x, rot, heads=16):
    """attend.py:173-189 (+ Attend.forward :102-139, scale = 10, no mask)."""
    b, n, _ = x.shape
    qkv = F.linear(x, sd[prefix + "to_qkv.weight"])
    q, k, v = qkv.chunk(3, dim=-1)
    q, k, v = (t.reshape(b, n, heads, -1).permute(0, 2, 1, 3) for t in (q, k, v))
    dh = q.shape[-1]
    q = _rmsnorm_dir(q) * sd[prefix + "q_norm.gamma"] * (dh ** 0.5)   # attend.py:144-151
    k = _rmsnorm_dir(k) * sd[prefix + "k_norm.gamma"] * (dh ** 0.5)
    q = q * rot.cos() + _rotate_half(q) * rot.sin()                   # pos_emb.py:57-59
    k = k * rot.cos() + _rotate_half(k) * rot.sin()
    sim = torch.einsum("bhid,bhjd->bhij", q, k) * 10.0                # attend.py:123
    attn = sim.softmax(dim=-1)
    out = torch.einsum("bhij,bhjd->bhid", attn, v)
    out = out.permute(0, 2, 1, 3).reshape(b, n, heads * dh)
    return F.linear(out, sd[prefix + "to_out.weight"])


def feedforward(sd, prefix, x):
    """transformer.py:92-104 -- GEGLU: first half value, second half gate, exact gelu."""
    h = F.linear(x, sd[prefix + "0.weight"], sd[prefix + "0.bias"])
    val, gate = h.chunk(2, dim=-1)
    h = F.gelu(gate) * val
    return F.linear(h, sd[prefix + "3.weight"], sd[prefix + "3.bias"])


def time_embedding(sd, times):
    """pos_emb.py:22-26 + flow.py:92-96: SiLU(Linear(cat(sin, cos)(t * w * 2pi)))."""
    w = sd[FH + "sinu_pos_emb.0.weights"]
    freqs = times[:, None] * w[None, :] * 2 * math.pi
    four = torch.cat((freqs.sin(), freqs.cos()), dim=-1)
    return F.silu(F.linear(four, sd[FH + "sinu_pos_emb.1.weight"], sd[FH + "sinu_pos_emb.1.bias"]))


def flow_forward(sd, x, cond, t, depth=2, return_stages=False):
    """flow.py:185-261 with cond_drop_prob = 0, masks None.  x, cond [B,N,256]; t scalar/[B]."""
    b = cond.shape[0]
    times = torch.as_tensor(t, dtype=x.dtype)
    if times.ndim == 0:
        times = times.repeat(b)                                       # flow.py:208-211
    stages = {}
    embed = torch.cat((x, cond), dim=-1)                              # :234-237
    h = F.linear(embed, sd[FH + "to_embed.weight"], sd[FH + "to_embed.bias"])   # :239
    stages["to_embed"] = h
    w = sd[FH + "conv_embed.dw_conv1d.0.weight"]
    pe = F.conv1d(h.transpose(1, 2), w, sd[FH + "conv_embed.dw_conv1d.0.bias"],
                  padding=w.shape[-1] // 2, groups=w.shape[0])
    h = F.gelu(pe).transpose(1, 2) + h                                # :240, transformer.py:16-46
    stages["conv_embed"] = h
    t_emb = time_embedding(sd, times)                                 # :242
    stages["time_emb"] = t_emb
    rot = rotary_table(sd, h.shape[1])
    for layer in range(depth):                                        # transformer.py:208-227
        p = f"{FH}transformer.layers.{layer}."
        a_in = ada_rmsnorm(sd, p + "2.", h, t_emb)
        h = attention(sd, p + "3.", a_in, rot) + h
        stages[f"attn{layer}"] = h
        f_in = ada_rmsnorm(sd, p + "4.", h, t_emb)
        h = feedforward(sd, p + "5.", f_in) + h
        stages[f"ff{layer}"] = h
    h = _rmsnorm_dir(h) * (h.shape[-1] ** 0.5) * sd[FH + "transformer.final_norm.gamma"]  # :234
    stages["final_norm"] = h
    out = F.linear(h, sd[FH + "to_pred.weight"])                      # flow.py:261
    return (out, stages) if return_stages else out


# ----------------------------------------------------------------------------
# fixed-grid ODE (torchdiffeq >= 0.2.3, pyproject.toml:14; call site cfm:243)
# ----------------------------------------------------------------------------
def odeint_fixed(fn, y0, t, method):
    y = y0
    for i in range(len(t) - 1):
        t0, t1 = t[i], t[i + 1]
        dt = t1 - t0
        if method == "euler":
            y = y + dt * fn(t0, y)
        elif method == "midpoint":
            half = 0.5 * dt
            y_mid = y + fn(t0, y) * half
            y = y + dt * fn(t0 + half, y_mid)
        else:
            raise ValueError(method)
    return y


def mel_cutoff_bins(cond_mel, percentile=0.9995):
    """cfm_superresolution.py:134-144,154-159: per clip, cumulative energy of exp(mel) over the mel
    bins, summed over time; the python scan is kept verbatim in meaning."""
    cuts = []
    for i in range(cond_mel.size(0)):
        energy = torch.cumsum(torch.sum(torch.abs(torch.exp(cond_mel[i])), dim=0), dim=0)
        thr = energy[-1] * percentile
        cut = 0
        for k in range(1, energy.shape[0]):
            if energy[-k] < thr:
                cut = energy.shape[0] - k
                break
        cuts.append(cut)
    return cuts


def mel_replace(high, low, cuts):
    """cfm_superresolution.py:146-152: bins >= cut from `high`, bins < cut from `low`."""
    out = torch.zeros_like(high)
    for i, c in enumerate(cuts):
        out[i][..., c:] = high[i][..., c:]
        out[i][..., :c] = low[i][..., :c]
    return out


def prior(cond_mel, noise, cfm_method="basic_cfm", std_1=1.0, std_2=0.0):
    """cfm_superresolution.py:219-237."""
    if cfm_method == "basic_cfm":
        return noise
    if cfm_method in ("independent_cfm_adaptive", "independent_cfm_constant"):
        return cond_mel * std_1 + noise * std_2
    if cfm_method == "independent_cfm_mix":
        return mel_replace(noise, cond_mel * std_1 + noise * std_2, mel_cutoff_bins(cond_mel))
    raise NotImplementedError(cfm_method)


# ----------------------------------------------------------------------------
# BigVGAN                                     models/bigvgan/models.py:172-194
# ----------------------------------------------------------------------------
def upsample2x(x, filt):
    """alias_free_torch/resample.py:25-33 (ratio 2, kernel 12)."""
    c = x.shape[1]
    k = filt.shape[-1]
    ratio = 2
    pad = k // ratio - 1
    pad_left = pad * ratio + (k - ratio) // 2
    pad_right = pad * ratio + (k - ratio + 1) // 2
    x = F.pad(x, (pad, pad), mode="replicate")
    x = ratio * F.conv_transpose1d(x, filt.expand(c, -1, -1), stride=ratio, groups=c)
    return x[..., pad_left:-pad_right]


def downsample2x(x, filt):
    """alias_free_torch/filter.py:86-95 (stride 2, kernel 12, replicate pad 5|6)."""
    c = x.shape[1]
    k = filt.shape[-1]
    x = F.pad(x, (k // 2 - 1, k // 2), mode="replicate")
    return F.conv1d(x, filt.expand(c, -1, -1), stride=2, groups=c)


def snake(x, alpha, beta, logscale, is_beta):
    """activations.py:48-59 (Snake) / :107-120 (SnakeBeta)."""
    a = alpha[None, :, None]
    bparam = beta[None, :, None] if is_beta else a
    if logscale:
        a = torch.exp(a)
        bparam = torch.exp(bparam) if is_beta else a
    return x + (1.0 / (bparam + 1e-9)) * torch.pow(torch.sin(x * a), 2)


def activation1d(sd, prefix, x, h):
    """alias_free_torch/act.py:23-28: up2x -> snake -> down2x."""
    is_beta = h["activation"] == "snakebeta"
    x = upsample2x(x, sd[prefix + "upsample.filter"])
    x = snake(x, sd[prefix + "act.alpha"], sd[prefix + "act.beta"] if is_beta else None,
              bool(h.get("snake_logscale", False)), is_beta)
    return downsample2x(x, sd[prefix + "downsample.lowpass.filter"])


def amp_block1(sd, prefix, x, h, ksize, dilations):
    """models.py:63-72"""
    for m, d in enumerate(dilations):
        xt = activation1d(sd, f"{prefix}activations.{2 * m}.", x, h)
        xt = F.conv1d(xt, sd[f"{prefix}convs1.{m}.weight"], sd[f"{prefix}convs1.{m}.bias"],
                      dilation=d, padding=(ksize * d - d) // 2)
        xt = activation1d(sd, f"{prefix}activations.{2 * m + 1}.", xt, h)
        xt = F.conv1d(xt, sd[f"{prefix}convs2.{m}.weight"], sd[f"{prefix}convs2.{m}.bias"],
                      padding=(ksize - 1) // 2)
        x = xt + x
    return x


def amp_block2(sd, prefix, x, h, ksize, dilations):
    """models.py:115-121"""
    for m, d in enumerate(dilations):
        xt = activation1d(sd, f"{prefix}activations.{m}.", x, h)
        xt = F.conv1d(xt, sd[f"{prefix}convs.{m}.weight"], sd[f"{prefix}convs.{m}.bias"],
                      dilation=d, padding=(ksize * d - d) // 2)
        x = xt + x
    return x


def bigvgan_forward(sd, h, mel, return_stages=False):
    """mel [B,256,N] -> [B,1,480N].  h = vocoder JSON dict."""
    amp_block = amp_block1 if str(h["resblock"]) == "1" else amp_block2
    stages = {}
    x = F.conv1d(mel, sd[VOC + "conv_pre.weight"], sd[VOC + "conv_pre.bias"], padding=3)
    stages["conv_pre"] = x
    nk = len(h["resblock_kernel_sizes"])
    for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
        x = F.conv_transpose1d(x, sd[f"{VOC}ups.{i}.0.weight"], sd[f"{VOC}ups.{i}.0.bias"],
                               stride=u, padding=(k - u) // 2)
        stages[f"up{i}"] = x
        xs = None
        for j in range(nk):
            y = amp_block(sd, f"{VOC}resblocks.{i * nk + j}.", x, h,
                           h["resblock_kernel_sizes"][j], h["resblock_dilation_sizes"][j])
            xs = y if xs is None else xs + y
        x = xs / nk
        stages[f"stage{i}"] = x
    x = activation1d(sd, VOC + "activation_post.", x, h)
    x = F.conv1d(x, sd[VOC + "conv_post.weight"], sd[VOC + "conv_post.bias"], padding=3)
    x = torch.tanh(x)
    return (x, stages) if return_stages else x


# ----------------------------------------------------------------------------
# post-processing                                       postprocessing.py:5-41
# ----------------------------------------------------------------------------
def stft_center(x):
    """torchaudio Spectrogram(2048, hop 480, power=None, pad_mode='constant')."""
    return torch.stft(x, N_FFT, hop_length=HOP, win_length=N_FFT,
                      window=torch.hann_window(N_FFT, dtype=x.dtype), center=True, pad_mode="constant", normalized=False, onesided=True,
                      return_complex=True)


def cutoff_index(spec, threshold=0.99):
    """postprocessing.py:10-16 (python loop kept verbatim in meaning)."""
    energy = torch.cumsum(torch.sum(spec.squeeze().abs(), dim=-1), dim=0)
    thr = energy[-1] * threshold
    for i in range(1, energy.size(0)):
        if energy[-i] < thr:
            return energy.size(0) - i
    return 0


def post_processing(pred, src, length, return_cr=False):
    """pred, src [1,T] -> [1,length]."""
    sp, ss = stft_center(pred), stft_center(src)
    cr = cutoff_index(ss)
    n = min(sp.size(-1), ss.size(-1))
    res = torch.empty_like(sp)[:, :, :n]
    res[:, cr:] = sp[:, cr:, :n]
    res[:, :cr] = ss[:, :cr, :n]
    audio = torch.istft(res, N_FFT, hop_length=HOP, win_length=N_FFT,
                        window=torch.hann_window(N_FFT, dtype=pred.dtype), center=True, length=length)
    audio = audio / torch.abs(audio).max() * 0.99
    return (audio, cr) if return_cr else audio


# ----------------------------------------------------------------------------
# whole path                          flowhighsr.py:51-102 + cfm:162-284
# ----------------------------------------------------------------------------
def vector_field(sd, y, cond_mel, t, depth=2, cond_scale=1.0):
    """flow.py:165-178 forward_with_cond_scale: classifier-free guidance against the null condition."""
    logits = flow_forward(sd, y, cond_mel, t, depth)
    if cond_scale == 1.0:
        return logits
    null = sd[FH + "null_cond"].to(cond_mel.dtype).expand_as(cond_mel)
    null_logits = flow_forward(sd, y, null, t, depth)
    return null_logits + (logits - null_logits) * cond_scale


@torch.no_grad()
def sample(sd, h, cond48, noise, time_steps=1, method="euler", cfm_method="basic_cfm",
           sigma=0.0, depth=2, return_stages=False, cond_scale=1.0, mel_pp=False, decode=True):
    """cond48 [B,T] (48 kHz, peak-normalised), noise [B,N,256] -> waveform [B,1,480N]."""
    cond_mel = logmel(cond48)
    y0 = prior(cond_mel, noise, cfm_method, 1.0, sigma)
    t = torch.linspace(0, 1, time_steps + 1, dtype=cond48.dtype)
    mel = odeint_fixed(lambda tt, y: vector_field(sd, y, cond_mel, tt, depth, cond_scale), y0, t, method)
    if mel_pp:                                                        # cfm:278-279
        mel = mel_replace(mel, cond_mel, mel_cutoff_bins(cond_mel))
    if not decode:
        return mel
    wav = bigvgan_forward(sd, h, mel.transpose(1, 2))
    if return_stages:
        return wav, {"cond_mel": cond_mel, "mel": mel}
    return wav


@torch.no_grad()
def generate(sd, h, audio, sr, noise, timestep=1, method="euler", cfm_method="basic_cfm",
             sigma=0.0, return_stages=False):
    """One clip, exactly the reference's generate() contract.  Returns [1,T48]."""
    cond = preprocess(audio, sr)
    wav, st = sample(sd, h, cond, noise, timestep, method, cfm_method, sigma, return_stages=True)
    wav = wav.squeeze(1)
    out, cr = post_processing(wav, cond, cond.size(-1), return_cr=True)
    if return_stages:
        st.update(cond=cond, wav=wav, cr=cr)
        return out, st
    return out
