"""CPU: the oracle restatement against the reference-generated golden vectors."""
import numpy as np
import pytest
import torch

from conftest import E2E_CASES, load_golden
from flowhigh_amd import synth
from oracle import ref_cpu, slaney

_SD = {}


def state_dict(cfg, seed):
    key = (repr(sorted(cfg.items())), seed)
    if key not in _SD:
        _SD[key] = synth.make_state_dict(cfg, seed)
    return _SD[key]


@pytest.mark.parametrize("name", E2E_CASES)
def test_oracle_matches_reference_golden(name):
    g = load_golden(name)
    sd = state_dict(g["cfg"], g["seed"])
    chk = float(sum(float(v.double().sum()) for v in sd.values()))
    assert chk == pytest.approx(float(g["sd_checksum"]), rel=1e-12), "synthetic weight generator drifted"
    out, st = ref_cpu.generate(sd, g["cfg"], g["audio"], g["sr_in"], torch.from_numpy(g["noise"]),
                               g["steps"], g["method"], g["cfm_method"], g["sigma"], return_stages=True)
    assert st["cr"] == g["cr"]
    assert np.abs(st["cond"].numpy()[0] - g["cond48"]).max() <= 1e-6
    assert np.abs(st["cond_mel"].numpy() - g["cond_mel"]).max() <= 1e-5
    assert np.abs(st["mel"].numpy() - g["mel"]).max() <= 2e-5
    assert np.abs(st["wav"].numpy() - g["wav"]).max() <= 1e-6
    assert out.shape == g["out"].shape
    assert np.abs(out.numpy() - g["out"]).max() <= 2e-6      # tolerance: fp32 thread-order noise


@pytest.mark.parametrize("name", ["tiny_euler", "alt_midpoint", "tiny_mix"])
def test_oracle_sampler_options(name):
    """cond_scale != 1 (classifier-free guidance against null_cond) and mel_pp=True."""
    g = load_golden(name)
    sd = state_dict(g["cfg"], g["seed"])
    cond = torch.from_numpy(g["cond48"])[None]
    assert ref_cpu.mel_cutoff_bins(torch.from_numpy(g["cond_mel"])) == g["mel_cutoff_bins"].tolist()
    mel = ref_cpu.sample(sd, g["cfg"], cond, torch.from_numpy(g["noise"]), g["steps"], g["method"], g["cfm_method"],
                         g["sigma"], cond_scale=1.3, mel_pp=True, decode=False)
    assert np.abs(mel.numpy() - g["mel_cfg13_melpp"]).max() <= 3e-5


@pytest.mark.parametrize("name", ["tiny_euler", "alt_midpoint"])
def test_oracle_flow_forward(name):
    g = load_golden(name)
    sd = state_dict(g["cfg"], g["seed"])
    pred = ref_cpu.flow_forward(sd, torch.from_numpy(g["noise"]), torch.from_numpy(g["cond_mel"]), 0.3)
    assert np.abs(pred.numpy() - g["flow_pred_t03"]).max() <= 1e-5


def test_oracle_ops_golden():
    g = load_golden("ops")
    filt = torch.from_numpy(g["kaiser_taps"]).view(1, 1, 12)
    assert np.array_equal(synth.kaiser_sinc_filter().numpy().ravel(), g["kaiser_taps"])
    x = torch.from_numpy(g["act_x"])
    sd = {"a.upsample.filter": filt, "a.downsample.lowpass.filter": filt,
          "a.act.alpha": torch.from_numpy(g["act_alpha"]), "a.act.beta": torch.from_numpy(g["act_beta"])}
    y = ref_cpu.activation1d(sd, "a.", x, {"activation": "snakebeta", "snake_logscale": True})
    assert np.abs(y.numpy() - g["act_snakebeta_log"]).max() <= 1e-6
    sd["a.act.alpha"] = torch.from_numpy(np.abs(g["act_alpha"]) + 0.5)
    y2 = ref_cpu.activation1d(sd, "a.", x, {"activation": "snake", "snake_logscale": False})
    assert np.abs(y2.numpy() - g["act_snake_lin"]).max() <= 1e-6
    out, cr = ref_cpu.post_processing(torch.from_numpy(g["pp_pred"]), torch.from_numpy(g["pp_src"]), 4999, return_cr=True)
    assert cr == int(g["pp_cr"])
    assert np.abs(out.numpy() - g["pp_out"]).max() <= 1e-6


def test_slaney_mel_against_independent_implementation():
    """librosa is absent; transformers ships an independent Slaney filter bank."""
    from transformers.audio_utils import mel_filter_bank
    ours = slaney.mel_filter_bank(48000, 2048, 256, 20.0, 24000.0)
    theirs = mel_filter_bank(num_frequency_bins=1025, num_mel_filters=256, min_frequency=20.0,
                             max_frequency=24000.0, sampling_rate=48000, norm="slaney", mel_scale="slaney").T
    assert ours.shape == (256, 1025) and ours.dtype == np.float32
    assert np.abs(ours - theirs).max() < 1e-7
    assert (ours.sum(axis=1) > 0).all()
