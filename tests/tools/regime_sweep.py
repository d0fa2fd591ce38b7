"""Regime sweep: how much error headroom each conv form keeps as the synthetic weights move towards what trained weights may
look like (trained-weight parity cannot be pinned here: SURVEY.md 8c "Weights").

For every regime -- convs2 gain x snake log-parameter spread x conv_post scale (flowhigh_amd/synth.py:
make_vocoder_state_dict) -- the oracle's vocoder (oracle/ref_cpu.py) runs in float32 and in float64 on the CPU; their
difference is the REFERENCE'S OWN rounding noise in that regime.  The HIP vocoder runs in its four conv forms

    bf16x6    conv_form='bf16x6' (the default since round 6): the Winograd convs of the wide stages, conv_pre and the first two
              upsamplers on the BF16 matrix cores over exact three-piece splits; the narrow-stage kernel in fp32
    winograd  conv_form='winograd': F(5,4) Winograd (>= 96 channels) + the narrow-stage F(5,4) kernel (<= 48 channels), fp32 MFMA
    f43       ... with FH_WINO54=0 FH_AMP=0: F(4,3) Winograd everywhere it applies
    direct    conv_form='direct': the direct implicit-GEMM form everywhere

and every form's max-abs distance to the float64 oracle is printed next to that noise.

    python tests/tools/regime_sweep.py [TINY|SYNTH] [frames=60]         (GPU box; ~1 min at TINY, ~10 min at SYNTH)
"""
import itertools
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from flowhigh_amd import synth          # noqa: E402
from oracle import ref_cpu              # noqa: E402

GAINS, BOUNDS, POSTS = (0.2, 0.4, 0.6), (0.5, 1.5, 2.5), (0.3, 1.0)
FORMS = {"bf16x6": {"FH_CONV_FORM": "bf16x6"}, "winograd": {"FH_CONV_FORM": "winograd"},
         "f43": {"FH_CONV_FORM": "winograd", "FH_WINO54": "0", "FH_AMP": "0"}, "direct": {"FH_CONV_FORM": "direct"}}


def oracle_pair(cfg, sd, mel):
    """(float32 waveform, float64 waveform) of the oracle's vocoder for mel [B, N, 256]."""
    m = mel.transpose(1, 2).contiguous()
    o32 = ref_cpu.bigvgan_forward(sd, cfg, m).squeeze(1)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    o64 = ref_cpu.bigvgan_forward(sd64, cfg, m.double()).squeeze(1)
    return o32, o64


def hip_forms(cfg, sd, mel, device="cuda:0"):
    """{form: waveform} of the HIP vocoder; the switches are read when the model is built."""
    from flowhigh_amd import vocoder as V
    out = {}
    for name, env in FORMS.items():
        saved = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            voc = V.Vocoder(cfg, sd, device)
            out[name] = voc.forward(mel.to(device)).cpu()
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    return out


def sweep(cfg, frames, seed=1, regimes=None):
    g = torch.Generator().manual_seed(175)
    mel = torch.randn(1, frames, 256, generator=g) * 2.0 - 3.0
    rows = []
    for gain, bound, post in (regimes or itertools.product(GAINS, BOUNDS, POSTS)):
        sd = synth.make_vocoder_state_dict(cfg, seed=seed, convs2_gain=gain, snake_bound=bound, post_gain=post)
        o32, o64 = oracle_pair(cfg, sd, mel)
        noise = float((o32.double() - o64).abs().max())
        errs = {k: float((w.double() - o64).abs().max()) for k, w in hip_forms(cfg, sd, mel).items()}
        rows.append(dict(gain=gain, bound=bound, post=post, amp=float(o64.abs().max()), noise=noise, **errs))
    return rows


def main():
    width = sys.argv[1] if len(sys.argv) > 1 else "TINY"
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    cfg = synth.TINY_CFG if width == "TINY" else synth.SYNTH_CFG
    torch.set_num_threads(16)
    print(f"# regime sweep, {width}-CFG (C0 = {cfg['upsample_initial_channel']}), {frames} frames, vocoder output (pre post-processing)")
    print("# max |.| over the waveform: oracle fp32 vs fp64 (its own noise), then HIP form vs oracle fp64; bar = 1e-4")
    print(f"{'gain':>5} {'snake':>6} {'post':>5} {'|wav|':>7} {'noise':>9} {'bf16x6':>9} {'winograd':>9} {'f43':>9} {'direct':>9}  bf16x6/noise")
    for r in sweep(cfg, frames):
        ratio = r["bf16x6"] / max(r["noise"], 1e-12)
        print(f"{r['gain']:5.1f} {r['bound']:6.1f} {r['post']:5.1f} {r['amp']:7.3f} {r['noise']:9.2e} {r['bf16x6']:9.2e} {r['winograd']:9.2e} "
              f"{r['f43']:9.2e} {r['direct']:9.2e}  {ratio:6.1f}", flush=True)


if __name__ == "__main__":
    main()
