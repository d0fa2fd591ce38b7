"""GPU: FlowHighSR.generate() through the C-ABI library against (a) the golden vectors produced by
the real reference, (b) the CPU oracle on larger seeded cases, (c) size-independent properties at
the BASELINE.json sizes.  The bar from BASELINE.json: <= 1e-4 max-abs on the 48 kHz waveform."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import E2E_CASES, load_golden          # noqa: E402
from flowhigh_amd import FLowHigh, FlowHighSR, synth  # noqa: E402
from oracle import ref_cpu                            # noqa: E402

TOL_WAVEFORM = 1e-4
_MODELS = {}


def model_for(cfg, seed, method="euler", cfm_method="basic_cfm", sigma=0.0, upsampling="scipy", form=None, fresh=False):
    """form: conv_form of the model (None = what a deployment gets: 'auto' -> the default form, probed against the direct form at
    load).  fresh: build a model now (under the environment as it is now) instead of taking the cached one."""
    key = (repr(sorted(cfg.items())), seed, form)
    if fresh:
        sd = synth.make_state_dict(cfg, seed)
        fh = FLowHigh(sd, cfg, "cuda", conv_form=form)
        return FlowHighSR(fh, sigma=sigma, cfm_method=cfm_method, torchdiffeq_ode_method=method, upsampling_method=upsampling), sd
    if key not in _MODELS:
        sd = synth.make_state_dict(cfg, seed)
        _MODELS[key] = (FLowHigh(sd, cfg, "cuda", conv_form=form), sd)
    fh, sd = _MODELS[key]
    m = FlowHighSR(fh, sigma=sigma, cfm_method=cfm_method, torchdiffeq_ode_method=method,
                   upsampling_method=upsampling)
    return m, sd


@pytest.mark.parametrize("name", E2E_CASES)
def test_generate_matches_reference_golden(name):
    g = load_golden(name)
    m, _ = model_for(g["cfg"], g["seed"], g["method"], g["cfm_method"], g["sigma"])
    out, st = m.generate_batch([g["audio"]], g["sr_in"], 48000, g["steps"], noise=torch.from_numpy(g["noise"]),
                               return_stages=True)
    assert out.dtype == torch.float32 and out.is_cuda and tuple(out.shape) == g["out"].shape
    assert int(st["cr"][0].item()) == g["cr"]                               # integer: exact
    assert np.abs(st["wav"].cpu().numpy() - g["wav"]).max() <= TOL_WAVEFORM
    assert np.abs(out.cpu().numpy() - g["out"]).max() <= TOL_WAVEFORM


def test_generate_draws_reference_prior_when_noise_is_none():
    g = load_golden("tiny_euler")
    m, _ = model_for(g["cfg"], g["seed"], g["method"])
    gen = torch.Generator().manual_seed(2000 + g["seed"])
    out = m.generate(g["audio"], g["sr_in"], 48000, g["steps"], generator=gen)
    assert np.abs(out.cpu().numpy() - g["out"]).max() <= TOL_WAVEFORM


@pytest.mark.parametrize("sr_in,method,steps,secs", [(12000, "euler", 1, 1.0), (16000, "midpoint", 1, 0.6),
                                                     (24000, "midpoint", 2, 0.5), (8000, "euler", 1, 0.7)])
def test_generate_synth_cfg_vs_oracle(sr_in, method, steps, secs):
    """Full-width vocoder (SYNTH-CFG, 1536 channels) on short clips the oracle finishes in seconds."""
    cfg = synth.SYNTH_CFG
    m, sd = model_for(cfg, 0, method)
    audio = synth.lowres_clip(7, secs, sr_in)
    n = int(round(secs * sr_in)) * (48000 // sr_in) // 480
    noise = synth.prior_noise(7, n)
    ref, st = ref_cpu.generate(sd, cfg, audio, sr_in, noise, steps, method, return_stages=True)
    out, got = m.generate_batch([audio], sr_in, 48000, steps, noise=noise, return_stages=True)
    assert int(got["cr"][0].item()) == st["cr"]
    assert (got["wav"].cpu() - st["wav"]).abs().max().item() <= TOL_WAVEFORM
    assert (out.cpu() - ref).abs().max().item() <= TOL_WAVEFORM


@pytest.mark.parametrize("sr_in,method,secs,B", [(12000, "euler", 1.0, 1), (16000, "midpoint", 0.45, 2)])
def test_generate_odd_upsamplers_full_width_vs_oracle(sr_in, method, secs, B):
    """Full-width vocoder (1536 channels) in the upstream BigVGAN convention k = 2 u on the odd rates of hop 480
    (rates [5, 4, 4, 3, 2], kernels [10, 8, 8, 6, 4]): every u = 5 / u = 3 stage returns u L + 1 samples
    (/root/reference/src/flowhigh/models/bigvgan/models.py:141-146), the vocoder 480 N + 98, and PostProcessing trims to
    the input's length (postprocessing.py:30-39).  Batch rows equal single-clip runs bit for bit."""
    cfg = dict(synth.ODD_CFG, upsample_initial_channel=1536)
    m, sd = model_for(cfg, 0, method)
    clips = [synth.lowres_clip(40 + b, secs, sr_in) for b in range(B)]
    n = int(round(secs * sr_in)) * (48000 // sr_in) // 480
    noise = torch.cat([synth.prior_noise(40 + b, n) for b in range(B)], 0)
    out, got = m.generate_batch(clips, sr_in, 48000, 1, noise=noise, return_stages=True)
    assert tuple(got["wav"].shape) == (B, 480 * n + 98) and tuple(out.shape) == (B, int(round(secs * sr_in)) * (48000 // sr_in))
    for b in range(B):
        ref, st = ref_cpu.generate(sd, cfg, clips[b], sr_in, noise[b:b + 1], 1, method, return_stages=True)
        assert int(got["cr"][b].item()) == st["cr"]
        assert (got["wav"][b:b + 1].cpu() - st["wav"]).abs().max().item() <= TOL_WAVEFORM
        assert (out[b:b + 1].cpu() - ref).abs().max().item() <= TOL_WAVEFORM
        if B > 1:
            assert torch.equal(m.generate(clips[b], sr_in, 48000, 1, noise=noise[b:b + 1]), out[b:b + 1])


def test_full_size_properties_odd_upsamplers():
    """BASELINE.json size (10 s clips, full-width vocoder) with the k = 2 u upsamplers on the odd rates (480 N + 98 vocoder
    samples, stage lengths 5 N + 1, 20 N + 4, 80 N + 16, 240 N + 49, 480 N + 98: rows that are not 16-byte aligned at three of
    the five stages): determinism, peak normalisation, batch rows = single-clip runs, and the time-chunked vocoder = the
    unchunked one, all bit for bit."""
    cfg = dict(synth.ODD_CFG, upsample_initial_channel=1536)
    m, _ = model_for(cfg, 0, "euler", upsampling="hip")
    clips = [synth.lowres_clip(70 + i, 10.0, 12000) for i in range(2)]
    noise = torch.cat([synth.prior_noise(70 + i, 1000) for i in range(2)], 0)
    out, st = m.generate_batch(clips, 12000, 48000, 1, noise=noise, return_stages=True)
    assert tuple(out.shape) == (2, 480000) and tuple(st["wav"].shape) == (2, 480098) and torch.isfinite(out).all()
    assert torch.allclose(out.abs().amax(dim=1).cpu(), torch.full((2,), 0.99), atol=1e-6)
    assert torch.equal(out, m.generate_batch(clips, 12000, 48000, 1, noise=noise))
    for b in range(2):
        assert torch.equal(m.generate(clips[b], 12000, 48000, 1, noise=noise[b:b + 1]), out[b:b + 1])
    voc = m.flowhigh.vocoder
    mel = torch.randn(1, 1000, 256, generator=torch.Generator().manual_seed(5)).cuda() * 2.0 - 3.0
    whole = voc.forward(mel).clone()
    halo, align = voc.chunk_geometry()
    assert torch.equal(voc.forward_chunked(mel, 5 * align), whole) and whole.shape[1] == 480098


def test_device_resampler_path_vs_oracle():
    cfg = synth.TINY_CFG
    m, sd = model_for(cfg, 0, "euler", upsampling="hip")
    audio = synth.lowres_clip(3, 0.5, 12000)
    noise = synth.prior_noise(3, 50)
    ref = ref_cpu.generate(sd, cfg, audio, 12000, noise, 1, "euler")
    out = m.generate(audio, 12000, 48000, 1, noise=noise)
    assert (out.cpu() - ref).abs().max().item() <= TOL_WAVEFORM


def test_batch_equals_single_clip_runs_bitwise():
    """Clips are independent: a batched run must reproduce per-clip runs exactly (this is also the
    multi-GPU sharding invariant)."""
    cfg = synth.TINY_CFG
    m, _ = model_for(cfg, 0, "midpoint")
    clips = [synth.lowres_clip(i, 0.5, 16000) for i in range(3)]
    noise = torch.cat([synth.prior_noise(i, 50) for i in range(3)], 0)
    both = m.generate_batch(clips, 16000, 48000, 1, noise=noise)
    for i in range(3):
        one = m.generate(clips[i], 16000, 48000, 1, noise=noise[i:i + 1])
        assert torch.equal(both[i:i + 1], one)


@pytest.mark.parametrize("secs,sr_in,method,steps,B", [(10.0, 12000, "euler", 1, 1), (10.0, 16000, "midpoint", 1, 2)])
def test_full_size_properties(secs, sr_in, method, steps, B):
    """BASELINE.json sizes (10 s clips, SYNTH-CFG): determinism, peak normalisation, finite output,
    and the low band of the output equals the low band of the input (the splice invariant)."""
    cfg = synth.SYNTH_CFG
    m, _ = model_for(cfg, 0, method, upsampling="hip")
    clips = [synth.lowres_clip(i, secs, sr_in) for i in range(B)]
    n = int(secs * 100)
    noise = torch.cat([synth.prior_noise(i, n) for i in range(B)], 0)
    out1, st = m.generate_batch(clips, sr_in, 48000, steps, noise=noise, return_stages=True)
    out2 = m.generate_batch(clips, sr_in, 48000, steps, noise=noise)
    assert tuple(out1.shape) == (B, int(secs * 48000))
    assert torch.isfinite(out1).all()
    assert torch.equal(out1, out2)
    assert torch.allclose(out1.abs().amax(dim=1).cpu(), torch.full((B,), 0.99), atol=1e-6)
    # splice invariant: below the cutoff bin the output spectrum is the source spectrum (up to the
    # common peak gain): compare band-limited energies through torch.stft on the host.
    cond = st["cond"].cpu()
    win = torch.hann_window(2048)
    for b in range(B):
        cr = int(st["cr"][b].item())
        assert 1 <= cr <= 1024
        so = torch.stft(out1[b].cpu(), 2048, 480, 2048, win, return_complex=True)
        sc = torch.stft(cond[b], 2048, 480, 2048, win, return_complex=True)
        lo = slice(2, max(3, cr - 4))
        gain = (so[lo].abs().sum() / sc[lo].abs().sum()).item()
        err = (so[lo] - gain * sc[lo]).abs().max().item() / sc[lo].abs().max().item()
        assert err <= 2e-2      # an iSTFT of a spliced STFT is only approximately consistent at the band edge


def test_from_local_reads_reference_checkpoint_files(tmp_path):
    """from_local: vocoder JSON + generator .pt with weight_g/weight_v + wrapper .pt['model']
    (flowhighsr.py:110-137, init_vocoder.py:8-23), defaults midpoint like the reference."""
    g = load_golden("tiny_ragged_16k")                       # a midpoint case
    synth.write_checkpoint_dir(tmp_path, g["cfg"], g["seed"], weight_norm=True)
    m = FlowHighSR.from_local(tmp_path, "cuda")
    assert m.odeint_kwargs["method"] == "midpoint" and m.cfm_method == "basic_cfm"
    out = m.generate(g["audio"], g["sr_in"], 48000, g["steps"], noise=torch.from_numpy(g["noise"]))
    assert np.abs(out.cpu().numpy() - g["out"]).max() <= TOL_WAVEFORM
    # missing key -> RuntimeError like load_state_dict(strict=True)
    import torch as _t
    ck = _t.load(tmp_path / "FLowHigh_basic_400k.pt", weights_only=False)
    ck["model"].pop("flowhigh.audio_enc_dec.vocoder.conv_post.bias")
    _t.save(ck, tmp_path / "FLowHigh_basic_400k.pt")
    with pytest.raises(RuntimeError):
        FlowHighSR.from_local(tmp_path, "cuda")


def test_conv_form_auto_probes_the_loaded_weights(caplog):
    """conv_form='auto' (the default of from_local / from_pretrained, reference flowhighsr.py:110-149): on a checkpoint whose
    convs2 gain is 0.6 and conv_post scale 1.0 -- the hardest regime of the sweep that still has a meaningful fp32 reference -- the
    load-time probe logs its estimate (default form against direct form on a 20-frame mel), keeps the default form or switches to
    the direct one by the 3e-5 rule, and the model it leaves stays under the 1e-4 bar against the FLOAT64 oracle."""
    import logging
    cfg = synth.SYNTH_CFG
    sd = synth.make_state_dict(cfg, 0)
    voc_sd = synth.make_vocoder_state_dict(cfg, seed=1, convs2_gain=0.6, snake_bound=0.5, post_gain=1.0)
    sd.update(voc_sd)
    with caplog.at_level(logging.INFO, logger="flowhigh_amd"):
        fh = FLowHigh(sd, cfg, "cuda", conv_form="auto")
    pr = fh.conv_form_probe
    assert pr is not None and 0.0 < pr["estimate"] < 1e-3 and pr["limit"] == 3e-5 and pr["frames"] == 20
    assert pr["chosen"] == ("direct" if pr["estimate"] > pr["limit"] else pr["default"]) == fh.conv_form
    assert any("conv_form='auto'" in r.getMessage() and "probe" in r.getMessage() for r in caplog.records)
    print(f"probe: {pr}")
    mel = torch.randn(1, 20, 256, generator=torch.Generator().manual_seed(175)) * 2.0 - 3.0
    wav = fh.vocoder.forward(mel.cuda()).cpu()
    torch.set_num_threads(min(16, max(1, torch.get_num_threads())))
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in voc_sd.items()}
    o64 = ref_cpu.bigvgan_forward(sd64, cfg, mel.transpose(1, 2).contiguous().double()).squeeze(1)
    err = float((wav.double() - o64).abs().max())
    print(f"'auto' -> {fh.conv_form}: {err:.2e} from the float64 oracle")
    assert err <= TOL_WAVEFORM
    # a limit of zero forces the switch: the model then runs the direct form (no Winograd launch in its plan)
    fh.vocoder = __import__("flowhigh_amd").vocoder.Vocoder(cfg, sd, "cuda", conv_form="bf16x6", act_blocks=fh.vocoder.act_blocks)
    assert fh.probe_conv_form(sd, limit=0.0)["chosen"] == "direct" == fh.conv_form
    assert {n for n, _, _ in fh.vocoder.plan(1, 20)["conv_launches"]} == {"direct"}
    # no probe for a named form
    assert FLowHigh(sd, cfg, "cuda", conv_form="winograd").conv_form_probe is None


def test_unsupported_options_raise():
    m, _ = model_for(synth.TINY_CFG, 0)
    audio = synth.lowres_clip(0, 0.3, 12000)
    with pytest.raises(NotImplementedError):
        m.sample(cond=torch.zeros(1, 4800), cond_mask=torch.ones(1, 10, dtype=torch.bool))
    m.upsampling_method = "soxr"
    with pytest.raises(UnboundLocalError):
        m.generate(audio, 12000)
    with pytest.raises(NotImplementedError):
        FlowHighSR(m.flowhigh, use_torchode=True)


@pytest.mark.parametrize("name", ["tiny_euler", "alt_midpoint", "tiny_mix"])
def test_sampler_options_match_reference_golden(name):
    """sample(cond_scale=1.3, mel_pp=True): classifier-free guidance against null_cond and the mel
    low-band replacement, both evaluated on the device (vectors produced by the reference)."""
    g = load_golden(name)
    m, _ = model_for(g["cfg"], g["seed"], g["method"], g["cfm_method"], g["sigma"])
    cond = torch.from_numpy(g["cond48"])[None]
    kw = dict(std_2=1.) if g["cfm_method"] == "independent_cfm_adaptive" else {}
    mel = m.sample(cond=cond, time_steps=g["steps"], cfm_method=g["cfm_method"], cond_scale=1.3, mel_pp=True,
                   decode_to_audio=False, noise=torch.from_numpy(g["noise"]), **kw)
    assert np.abs(mel.cpu().numpy() - g["mel_cfg13_melpp"]).max() <= 2e-4      # |mel| ~ 10, two fp32 transformer passes
    n = g["cond_mel"].shape[1]
    cut = m.mel_cutoff_bins(torch.from_numpy(g["cond_mel"]).cuda().reshape(n, -1).contiguous(), 1, n)
    assert cut.cpu().tolist() == g["mel_cutoff_bins"].tolist()


def test_generate_many_ragged_list_equals_single_calls():
    """Serving entry: clips of different lengths (and an int16 one) bucketed by length; every result is
    bit-identical to generate() on that clip alone with the same noise."""
    m, _ = model_for(synth.TINY_CFG, 0)
    secs = [0.2, 0.31, 0.2, 0.25, 0.31, 0.2]
    clips = [synth.lowres_clip(40 + i, s, 12000) for i, s in enumerate(secs)]
    clips[3] = (clips[3] * 20000).astype(np.int16)
    noise = [synth.prior_noise(40 + i, (len(c) * 4) // 480) for i, c in enumerate(clips)]
    many = m.generate_many(clips, 12000, 48000, 1, noise=noise, max_batch=2, ragged=False)      # bucketed by length
    assert len(many) == len(clips)
    for i, c in enumerate(clips):
        one = m.generate(c, 12000, 48000, 1, noise=noise[i])
        assert tuple(many[i].shape) == tuple(one.shape) == (1, len(c) * 4)
        assert torch.equal(many[i], one)
    # without explicit noise: drawn per clip in list order from the generator, like a loop over generate()
    g1, g2 = torch.Generator().manual_seed(7), torch.Generator().manual_seed(7)
    a = m.generate_many(clips[:3], 12000, generator=g1)
    b = [m.generate(c, 12000, generator=g2) for c in clips[:3]]
    assert all(torch.equal(x, y) for x, y in zip(a, b))


def test_batching_server_matches_single_generate():
    """flowhigh_amd.serve.BatchingServer: concurrent requests (mixed lengths, an int16 clip) come back equal to
    generate() on each clip alone with the same seeded prior."""
    import threading
    from flowhigh_amd.serve import BatchingServer
    m, _ = model_for(synth.TINY_CFG, 0)
    srv = BatchingServer(m, max_batch=4, max_wait_ms=50)
    clips = [synth.lowres_clip(60 + i, 0.2 + 0.05 * (i % 2), 12000) for i in range(5)]
    clips[2] = (clips[2] * 20000).astype(np.int16)
    futs = [None] * len(clips)

    def client(i):
        futs[i] = srv.submit(clips[i], 12000, 1, seed=100 + i)
    ts = [threading.Thread(target=client, args=(i,)) for i in range(len(clips))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    outs = [f.result(timeout=120) for f in futs]
    srv.close()
    for i, c in enumerate(clips):
        g = torch.Generator().manual_seed(100 + i)
        ref = m.generate(c, 12000, 48000, 1, generator=g)
        assert outs[i].shape == (len(c) * 4,)
        assert np.array_equal(outs[i], ref.cpu().squeeze(0).numpy())


@pytest.mark.parametrize("cfgname", ["TINY_CFG", "ODD_CFG"])
def test_graph_capture_replays_bit_identical(cfgname):
    """FlowHighSR.capture: the whole device path as one HIP graph; replays equal the eager call bit for bit.  (ODD_CFG: the
    vocoder returns 480 N + 98 samples, which keys the post-processing workspace the capture must hold on to.)"""
    m, _ = model_for(getattr(synth, cfgname), 0, upsampling="hip")
    n_in = 6000
    g = m.capture(2, n_in, 12000, 1)
    for seed in (50, 51):
        x = torch.from_numpy(np.stack([synth.lowres_clip(seed + i, n_in / 12000, 12000) for i in range(2)])).cuda()
        noise = torch.cat([synth.prior_noise(seed + i, 50) for i in range(2)], 0).cuda().reshape(100, -1).contiguous()
        g.x.copy_(x)
        g.noise.copy_(noise)
        got = g.replay().clone()
        ref = m.generate_from_device(x, 12000, 1, noise=noise)
        assert torch.equal(got, ref)


def test_baseline_config5_long_clip_multi_nfe():
    """BASELINE.json configs[4] shape: 30 s clips, 24 -> 48 kHz, time_step = 4 midpoint (8 NFE,
    N = 3000 frames, attention over 3000 keys), B = 2 here: finite, deterministic, peak-normalised."""
    cfg = synth.SYNTH_CFG
    m, _ = model_for(cfg, 0, "midpoint", upsampling="hip")
    clips = [synth.lowres_clip(20 + i, 30.0, 24000) for i in range(2)]
    noise = torch.cat([synth.prior_noise(20 + i, 3000) for i in range(2)], 0)
    out1 = m.generate_batch(clips, 24000, 48000, 4, noise=noise)
    out2 = m.generate_batch(clips, 24000, 48000, 4, noise=noise)
    assert tuple(out1.shape) == (2, 1440000) and torch.isfinite(out1).all()
    assert torch.equal(out1, out2)
    assert torch.allclose(out1.abs().amax(dim=1).cpu(), torch.full((2,), 0.99), atol=1e-6)


def test_generate_randomised_shapes_vs_oracle():
    """tests/tools/e2e_fuzz.py as a test: random clip lengths (0.01-3 s), input rates, batch sizes, solvers and
    resamplers through the full-width vocoder - every plan-time choice depends on the shape - vs the CPU oracle."""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    r = subprocess.run([sys.executable, str(root / "tests" / "tools" / "e2e_fuzz.py"), "12", "3"], cwd=root,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "FAIL" not in r.stdout, r.stdout[-2000:]
    assert "12 cases" in r.stdout


def test_short_clip_split_k_launches(monkeypatch):
    """Clips under ~2 s: the wide stages cut their input channels into slices (vocoder.wino_split_k) and add the
    partial outputs with fh_sum_f32.  The plan really does that for a 0.5 s clip, the result matches the oracle
    (also checked by test_generate_synth_cfg_vs_oracle), it is bit-identical alone and inside a batch, and it
    agrees with the unsplit launches to rounding."""
    cfg = synth.SYNTH_CFG
    m, sd = model_for(cfg, 0, "euler")
    voc = m.flowhigh.vocoder
    voc._plans.clear()
    kinds = [s_[0] for s_ in voc.plan(1, 50)["steps"]]
    assert kinds.count("sum") >= 5
    a, b = synth.lowres_clip(70, 0.5, 12000), synth.lowres_clip(71, 0.5, 12000)
    na, nb = synth.prior_noise(70, 50), synth.prior_noise(71, 50)
    one = m.generate_batch([a], 12000, 48000, 1, noise=na)
    two = m.generate_batch([b, a], 12000, 48000, 1, noise=torch.cat([nb, na], 0))
    assert torch.equal(one[0], two[1])
    ref = ref_cpu.generate(sd, cfg, a, 12000, na, 1, "euler")
    assert (one.cpu() - ref).abs().max().item() <= TOL_WAVEFORM
    # FH_WINO_SPLITK=0 is read when a model is BUILT (Vocoder.sw): this model keeps slicing whatever the environment says now ...
    monkeypatch.setenv("FH_WINO_SPLITK", "0")
    voc._plans.clear()
    assert [s_[0] for s_ in voc.plan(1, 50)["steps"]].count("sum") >= 5
    # ... and a model built under the switch never slices
    m2, _ = model_for(cfg, 0, "euler", fresh=True)
    assert m2.flowhigh.vocoder.sw["splitk"] is False
    assert "sum" not in [s_[0] for s_ in m2.flowhigh.vocoder.plan(1, 50)["steps"]]
    plain = m2.generate_batch([a], 12000, 48000, 1, noise=na)
    assert (plain - one).abs().max().item() <= 2e-5


@pytest.mark.parametrize("secs,sr_in,B", [(0.05, 12000, 1), (0.113, 16000, 3), (0.31, 8000, 2)])
def test_short_and_odd_length_clips_vs_oracle(secs, sr_in, B):
    """Few-frame clips, lengths that are not multiples of the hop, the vector width or the tile sizes."""
    cfg = synth.TINY_CFG
    m, sd = model_for(cfg, 0, "euler")
    clips = [synth.lowres_clip(40 + i, secs, sr_in) for i in range(B)]
    t48 = len(clips[0]) * (48000 // sr_in)
    n = t48 // 480
    noise = torch.cat([synth.prior_noise(40 + i, n) for i in range(B)], 0)
    out = m.generate_batch(clips, sr_in, 48000, 1, noise=noise)
    assert tuple(out.shape) == (B, t48)
    for i in range(B):
        ref = ref_cpu.generate(sd, cfg, clips[i], sr_in, noise[i:i + 1], 1, "euler")
        assert (out[i:i + 1].cpu() - ref).abs().max().item() <= TOL_WAVEFORM


def test_baseline_config_full_size_vs_oracle():
    """BASELINE.json configs[1] at its real size (one 10 s clip, 12 -> 48 kHz, euler x 1, full-width
    SYNTH-CFG vocoder) against the CPU oracle (~20 s of host time on 16 threads)."""
    torch.set_num_threads(min(16, torch.get_num_threads() if torch.get_num_threads() > 0 else 16))
    cfg = synth.SYNTH_CFG
    m, sd = model_for(cfg, 0, "euler")
    audio = synth.lowres_clip(0, 10.0, 12000)
    noise = synth.prior_noise(0, 1000)
    out, st = m.generate_batch([audio], 12000, 48000, 1, noise=noise, return_stages=True)
    ref, rs = ref_cpu.generate(sd, cfg, audio, 12000, noise, 1, "euler", return_stages=True)
    assert int(st["cr"][0].item()) == rs["cr"]
    assert (st["wav"].cpu() - rs["wav"]).abs().max().item() <= TOL_WAVEFORM
    assert (out.cpu() - ref).abs().max().item() <= TOL_WAVEFORM


# ------------------------------------------------------------------------------------------
# BASELINE.json configs[2..4] at their STATED batch sizes.  Every plan-time choice (Winograd tile shapes,
# fused / unfused stage-closing convs, the 128 x 64 direct tile, GEMM tile shapes, the attention kernel
# shape, the 20 GB workspace pool) depends on the batch, so B = 2 does not cover B = 32.
# ------------------------------------------------------------------------------------------
def _batch_case(secs, sr_in, method, steps, B, seed0):
    cfg = synth.SYNTH_CFG
    m, sd = model_for(cfg, 0, method, upsampling="hip")
    clips = [synth.lowres_clip(seed0 + i, secs, sr_in) for i in range(B)]
    n = int(secs * 100)
    noise = torch.cat([synth.prior_noise(seed0 + i, n) for i in range(B)], 0)
    out1 = m.generate_batch(clips, sr_in, 48000, steps, noise=noise).clone()
    out2 = m.generate_batch(clips, sr_in, 48000, steps, noise=noise)
    assert tuple(out1.shape) == (B, int(secs * 48000)) and torch.isfinite(out1).all()
    assert torch.equal(out1, out2)                                                   # deterministic
    assert torch.allclose(out1.abs().amax(dim=1).cpu(), torch.full((B,), 0.99), atol=1e-6)      # per-clip peak
    for i in (0, B - 1):                # the same clip alone (B = 1 plans): bit-identical
        one = m.generate_batch([clips[i]], sr_in, 48000, steps, noise=noise[i:i + 1])
        assert torch.equal(one[0], out1[i]), f"row {i} of the batch differs from the clip run alone"
    return m, sd, clips, noise, out1


def test_baseline_config3_batch32_16k_midpoint():
    """configs[2]: B = 32, 10 s clips, 16 -> 48 kHz, time_step = 1 midpoint.  Row 0 against the CPU oracle
    (/root/reference/src/flowhigh/cfm_superresolution.py:162-284 semantics) at the 1e-4 bar."""
    m, sd, clips, noise, out = _batch_case(10.0, 16000, "midpoint", 1, 32, 300)
    torch.set_num_threads(min(16, max(1, torch.get_num_threads())))
    ref = ref_cpu.generate(sd, synth.SYNTH_CFG, clips[0], 16000, noise[0:1], 1, "midpoint")
    assert (out[0:1].cpu() - ref).abs().max().item() <= TOL_WAVEFORM


def test_baseline_config4_share_batch32_8k_euler():
    """configs[3], one GPU's share: B = 32 of the 256 clips, 10 s, 8 -> 48 kHz, time_step = 1 euler.  Row 31
    against the CPU oracle."""
    m, sd, clips, noise, out = _batch_case(10.0, 8000, "euler", 1, 32, 400)
    torch.set_num_threads(min(16, max(1, torch.get_num_threads())))
    ref = ref_cpu.generate(sd, synth.SYNTH_CFG, clips[31], 8000, noise[31:32], 1, "euler")
    assert (out[31:32].cpu() - ref).abs().max().item() <= TOL_WAVEFORM


def test_baseline_config5_batch8_30s_midpoint4():
    """configs[4]: B = 8, 30 s clips, 24 -> 48 kHz, time_step = 4 midpoint (8 transformer evaluations at
    N = 3000).  Row 0 against the CPU oracle at the 1e-4 bar (~70 s of host time)."""
    m, sd, clips, noise, out = _batch_case(30.0, 24000, "midpoint", 4, 8, 500)
    torch.set_num_threads(min(16, max(1, torch.get_num_threads())))
    ref, rs = ref_cpu.generate(sd, synth.SYNTH_CFG, clips[0], 24000, noise[0:1], 4, "midpoint", return_stages=True)
    assert (out[0:1].cpu() - ref).abs().max().item() <= TOL_WAVEFORM


def test_baseline_config5_mel_level_vs_oracle():
    """configs[4] at the mel level: one 30 s 24 kHz clip, sample(time_steps=4, midpoint, decode_to_audio=False) =
    8 transformer evaluations at N = 3000 (cfm_superresolution.py:239-244, pos_emb.py:47-59, attend.py:102-139)
    against the oracle's sample(decode=False)."""
    cfg = synth.SYNTH_CFG
    m, sd = model_for(cfg, 0, "midpoint")
    torch.set_num_threads(min(16, max(1, torch.get_num_threads())))
    cond = ref_cpu.preprocess(synth.lowres_clip(510, 30.0, 24000), 24000)
    noise = synth.prior_noise(510, 3000)
    ref = ref_cpu.sample(sd, cfg, cond, noise, 4, "midpoint", decode=False)
    mel = m.sample(cond=cond, time_steps=4, decode_to_audio=False, noise=noise)
    assert tuple(mel.shape) == tuple(ref.shape) == (1, 3000, 256)
    assert (mel.cpu() - ref).abs().max().item() <= 2e-4       # |mel| ~ 10: 8 chained fp32 transformer passes


def test_baseline_config1_two_second_clip_vs_oracle():
    """configs[0] at its exact size: one 2 s clip, 12 -> 48 kHz, time_step = 1 euler, transformer 2 x 16 x 64,
    full-width SYNTH-CFG vocoder, against the CPU oracle."""
    cfg = synth.SYNTH_CFG
    m, sd = model_for(cfg, 0, "euler")
    audio = synth.lowres_clip(20, 2.0, 12000)
    noise = synth.prior_noise(20, 200)
    out, st = m.generate_batch([audio], 12000, 48000, 1, noise=noise, return_stages=True)
    ref, rs = ref_cpu.generate(sd, cfg, audio, 12000, noise, 1, "euler", return_stages=True)
    assert tuple(out.shape) == (1, 96000)
    assert int(st["cr"][0].item()) == rs["cr"]
    assert (st["wav"].cpu() - rs["wav"]).abs().max().item() <= TOL_WAVEFORM
    assert (out.cpu() - ref).abs().max().item() <= TOL_WAVEFORM


def test_two_minute_clip_runs_through_the_chunked_vocoder(monkeypatch):
    """A 120 s clip (N = 12 000 frames, app.py:8-26 takes arbitrary uploads): the vocoder runs in 60 s chunks with
    bounded workspace; forcing smaller chunks gives the same waveform bit for bit."""
    cfg = synth.SYNTH_CFG
    m, _ = model_for(cfg, 0, "euler", upsampling="hip")
    audio = synth.lowres_clip(90, 120.0, 12000)
    noise = synth.prior_noise(90, 12000)
    voc = m.flowhigh.vocoder
    out = m.generate(audio, 12000, 48000, 1, noise=noise).clone()
    assert tuple(out.shape) == (1, 5760000) and torch.isfinite(out).all()
    assert abs(out.abs().max().item() - 0.99) <= 1e-6
    assert any(len(k) == 3 and k[2] == 12000 for k in dict.keys(voc._plans))        # it did run in chunks
    monkeypatch.setenv("FH_VOCODER_CHUNK_FRAMES", "2400")
    out2 = m.generate(audio, 12000, 48000, 1, noise=noise)
    assert torch.equal(out, out2)


@pytest.mark.parametrize("cfgname,method,cfm", [("TINY_CFG", "euler", "basic_cfm"), ("SYNTH_CFG", "midpoint", "basic_cfm"),
                                                ("TINY_CFG", "midpoint", "independent_cfm_adaptive"),
                                                ("TINY_CFG", "euler", "independent_cfm_mix"),
                                                ("ODD_CFG", "euler", "basic_cfm")])         # vocoder returns 480 N + 98 samples
def test_generate_many_ragged_batch_equals_single_calls_bitwise(cfgname, method, cfm):
    """Masked / ragged batches (reference mask paths transformer.py:35-44, attend.py:127-128): clips of
    different lengths -- odd sample counts, an int16 clip, two of equal length, sub-2-s clips whose wide stages run as
    input-channel slices -- go through ONE launch sequence and come back bit-identical to generate() per clip."""
    cfg = getattr(synth, cfgname)
    m, _ = model_for(cfg, 0, method, cfm, sigma=1e-4 if cfm != "basic_cfm" else 0.0)
    secs = [0.5, 1.31, 0.2, 0.5, 2.2, 0.7713, 0.05]
    clips = [synth.lowres_clip(140 + i, s_, 12000) for i, s_ in enumerate(secs)]
    clips[2] = (clips[2] * 20000).astype(np.int16)
    noise = [synth.prior_noise(140 + i, (len(c) * 4) // 480) for i, c in enumerate(clips)]
    many = m.generate_many(clips, 12000, 48000, 2, noise=noise, ragged=True)
    assert len(many) == len(clips)
    for i, c in enumerate(clips):
        one = m.generate(c, 12000, 48000, 2, noise=noise[i])
        assert tuple(many[i].shape) == tuple(one.shape) == (1, len(c) * 4)
        assert torch.equal(many[i], one), f"clip {i} ({secs[i]} s) differs from generate() alone"
    # max_frames splits the list into several launch sequences; same results
    again = m.generate_many(clips, 12000, 48000, 2, noise=noise, ragged=True, max_frames=200)
    assert all(torch.equal(a, b) for a, b in zip(again, many))


# ------------------------------------------------------------------------------------------
# The conv forms a deployer can ask for by name (FlowHighSR.from_local(..., conv_form=) / FLowHigh(..., conv_form=)): 'bf16x6' (the
# default of round 6: every fp32 operand of the Winograd convs split exactly into three bf16 pieces, six bf16 MFMAs per 16-channel
# k-block, fp32 accumulation) and 'winograd' (the fp32-MFMA Winograd forms, the default of rounds 3-5).  Same tolerances, same
# bitwise invariants (batch / ragged / chunked against a clip alone) in both; every test above runs the DEFAULT form.
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("form", ["bf16x6", "winograd"])
@pytest.mark.parametrize("name", E2E_CASES)
def test_named_conv_form_generate_matches_reference_golden(name, form):
    g = load_golden(name)
    m, _ = model_for(g["cfg"], g["seed"], g["method"], g["cfm_method"], g["sigma"], form=form)
    voc = m.flowhigh.vocoder
    assert voc.form == form and voc.bf == (form == "bf16x6") and voc.wino_flag == (16 if voc.bf else 0)
    out, st = m.generate_batch([g["audio"]], g["sr_in"], 48000, g["steps"], noise=torch.from_numpy(g["noise"]),
                               return_stages=True)
    assert int(st["cr"][0].item()) == g["cr"]
    assert np.abs(st["wav"].cpu().numpy() - g["wav"]).max() <= TOL_WAVEFORM
    assert np.abs(out.cpu().numpy() - g["out"]).max() <= TOL_WAVEFORM


def test_bf16x6_full_size_vs_oracle_and_vs_fp32_form():
    """BASELINE configs[1] at full size in the bf16 x 6 form: against the CPU oracle at the 1e-4 bar, and its distance
    from the fp32-MFMA form next to both forms' distance from the oracle (the split is fp32-grade: the two errors are
    of the same size and the two forms differ by rounding only)."""
    torch.set_num_threads(min(16, max(1, torch.get_num_threads())))
    cfg = synth.SYNTH_CFG
    m16, sd = model_for(cfg, 0, "euler", form="bf16x6")
    m32, _ = model_for(cfg, 0, "euler", form="winograd")
    audio = synth.lowres_clip(0, 10.0, 12000)
    noise = synth.prior_noise(0, 1000)
    o16, s16 = m16.generate_batch([audio], 12000, 48000, 1, noise=noise, return_stages=True)
    o32, s32 = m32.generate_batch([audio], 12000, 48000, 1, noise=noise, return_stages=True)
    ref, rs = ref_cpu.generate(sd, cfg, audio, 12000, noise, 1, "euler", return_stages=True)
    assert int(s16["cr"][0].item()) == rs["cr"]
    e16 = (s16["wav"].cpu() - rs["wav"]).abs().max().item()
    e32 = (s32["wav"].cpu() - rs["wav"]).abs().max().item()
    print(f"vocoder output vs oracle: bf16 x 6 {e16:.2e}, fp32 MFMA {e32:.2e}; between the forms "
          f"{(s16['wav'] - s32['wav']).abs().max().item():.2e}")
    assert e16 <= TOL_WAVEFORM and (o16.cpu() - ref).abs().max().item() <= TOL_WAVEFORM
    assert e16 <= 3.0 * e32 + 2e-6                  # fp32-grade, not merely inside the bar


@pytest.mark.parametrize("form", ["bf16x6", "winograd"])
def test_named_conv_form_batch_ragged_and_chunked_invariants_bitwise(form):
    cfg = synth.SYNTH_CFG
    m, _ = model_for(cfg, 0, "euler", form=form)
    secs = [0.5, 1.31, 0.5, 2.2]
    clips = [synth.lowres_clip(240 + i, s_, 12000) for i, s_ in enumerate(secs)]
    noise = [synth.prior_noise(240 + i, (len(c) * 4) // 480) for i, c in enumerate(clips)]
    alone = [m.generate(c, 12000, 48000, 1, noise=z).clone() for c, z in zip(clips, noise)]
    many = m.generate_many(clips, 12000, 48000, 1, noise=noise, ragged=True)
    assert all(torch.equal(a, b) for a, b in zip(alone, many))
    both = m.generate_batch([clips[0], clips[2]], 12000, 48000, 1, noise=torch.cat([noise[0], noise[2]], 0))
    assert torch.equal(both[0:1], alone[0]) and torch.equal(both[1:2], alone[2])
    voc = m.flowhigh.vocoder
    mel = (torch.randn(1, 150, 256, generator=torch.Generator().manual_seed(5)) * 2.0 - 3.0).cuda()
    assert torch.equal(voc.forward_chunked(mel, 48), voc.forward(mel))


@pytest.mark.parametrize("args,workload", [(["--steps", "9", "--warmup", "1", "--no-cpu-baseline"], "configs[1]"),
                                           (["--config", "4", "--batch", "2", "--steps", "2", "--warmup", "1"], "configs[3]")])
def test_bench_prints_one_contract_line(args, workload):
    """bench.py on the GPU box: exactly one JSON line with the driver's keys, both roofline objects measured live
    (HIP events) and self-consistent, and -- for the default workload -- the opt-in bf16 x 6 side measurement with its
    distance from the fp32 form."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    r = subprocess.run([sys.executable, str(root / "bench.py")] + args, cwd=root, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "roofline_hbm", "cpu_baseline"):
        assert k in line, k
    form = line["config"]["conv_form"]
    assert workload in line["config"]["workload"] and line["n_gpus"] == 1
    assert line["dtype"] == "f32" if form != "bf16x6" else line["dtype"].startswith("f32 in / out / accumulate")
    clips = line["config"]["clips_per_gpu"]
    assert abs(line["value"] - clips * 10.0 / (line["ms_per_step"] / 1e3)) / line["value"] < 0.02
    rl, rh = line["roofline"], line["roofline_hbm"]
    assert rl["bound"] == "mfma" and 0.1 < rl["frac"] <= 1.0 and abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-3
    assert rl["peak"] in (157.3, 2500.0) and rl["conv_ms_per_step"] < line["ms_per_step"]
    fam = rl["by_family"]
    assert fam and next(iter(fam)) in ("wino54", "wino54_bf16x6") and all(0.0 < f["frac"] <= 1.0 for f in fam.values())
    assert abs(sum(f["ms_per_step"] for f in fam.values()) - rl["conv_ms_per_step"]) < 0.05 * rl["conv_ms_per_step"]
    assert sum(f["launches_per_step"] for f in fam.values()) == rl["all_conv"]["launches_per_step"]
    assert rl["all_conv"]["executed_fp32_equiv_tflops"] < rl["all_conv"]["algorithmic_equiv"]
    assert rh["bound"] == "hbm" and 0.1 < rh["frac"] <= 1.0 and rh["act_ms_per_step"] < line["ms_per_step"]
    if "--config" not in args:
        alt = line["alt_conv_form"]
        assert alt["conv_form"] != form and alt["max_abs_diff_vs_headline_waveform"] <= 5e-5 and alt["value"] > 0


# ------------------------------------------------------------------------------------------
# `device` semantics of from_local(ckpt_dir, device) (reference flowhighsr.py:110-137)
# ------------------------------------------------------------------------------------------
def test_model_runs_on_its_own_device_whatever_is_current(monkeypatch):
    """A model built on "cuda:0" launches on cuda:0's current stream whatever device / stream the CALLER has
    current.  With >= 2 GPUs the caller really sits on cuda:1; on a 1-GPU box every launch is spied on: the device
    current at launch time and the stream handed to the C ABI must be the model's."""
    from flowhigh_amd import hip
    g = load_golden("tiny_euler")
    sd = synth.make_state_dict(g["cfg"], g["seed"])
    fh = FLowHigh(sd, g["cfg"], "cuda:0")
    assert fh.device == torch.device("cuda", 0) == fh.vocoder.device == fh.net.device == fh.logmel.device
    m = FlowHighSR(fh, torchdiffeq_ode_method=g["method"])
    assert FLowHigh(sd, g["cfg"], "cuda").device.index == torch.cuda.current_device()      # 'cuda' pins the ordinal
    seen = []
    real_stream = hip.stream

    def spy(device=None):
        h = real_stream(device)
        seen.append((torch.cuda.current_device(), h))
        return h
    monkeypatch.setattr(hip, "stream", spy)
    side = torch.cuda.Stream(device="cuda:0")
    other = torch.device("cuda", 1) if torch.cuda.device_count() > 1 else None
    with torch.cuda.device(other if other is not None else 0):
        with torch.cuda.stream(side):           # current stream of cuda:0 := side (also while cuda:1 is current)
            out = m.generate(g["audio"], g["sr_in"], 48000, g["steps"], noise=torch.from_numpy(g["noise"]))
        assert torch.cuda.current_device() == (1 if other is not None else 0)               # the caller's, restored
    side.synchronize()
    assert out.device == torch.device("cuda", 0)
    assert np.abs(out.cpu().numpy() - g["out"]).max() <= TOL_WAVEFORM
    assert len(seen) >= 8 and all(d == 0 for d, _ in seen)       # (one hip.stream() per operator group)
    assert {h for _, h in seen} == {side.cuda_stream}


def test_lds_opt_in_is_per_device():
    """The > 64 KB LDS opt-in of the Winograd kernel is kept per device ordinal, not per process: the entry point
    works (again) after the current device was set explicitly, and on every visible device."""
    cfg = synth.TINY_CFG
    sd = synth.make_vocoder_state_dict(cfg, seed=1)
    mel = (torch.randn(1, 30, 256, generator=torch.Generator().manual_seed(5)) * 2.0 - 3.0)
    ref = None
    for d in range(torch.cuda.device_count()):
        from flowhigh_amd.vocoder import Vocoder
        voc = Vocoder(cfg, sd, f"cuda:{d}")
        wav = voc.forward(mel.to(f"cuda:{d}")).cpu()
        ref = wav if ref is None else ref
        assert torch.equal(wav, ref)


# ------------------------------------------------------------------------------------------
# sampler options on the ragged path (cfm_superresolution.py:162-175,278-279)
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfm_method", ["basic_cfm", "independent_cfm_mix"])
def test_sample_many_with_guidance_and_mel_pp_equals_single_calls_bitwise(cfm_method):
    """sample_many(cond_scale=1.3, mel_pp=True): clips of different lengths in one launch sequence, per-clip mel
    cutoff bins (fh_mel_energy_seg_f32 / fh_mel_splice_seg_f32): bit-identical to sample() per clip, mels and waveforms."""
    m, _ = model_for(synth.TINY_CFG, 0, "midpoint", cfm_method=cfm_method, sigma=0.3)
    secs = [0.5, 0.21, 1.0, 0.5]
    conds = [torch.from_numpy(ref_cpu.preprocess(synth.lowres_clip(60 + i, s, 12000), 12000).numpy()[0]) for i, s in enumerate(secs)]
    noise = [synth.prior_noise(60 + i, c.shape[0] // 480) for i, c in enumerate(conds)]
    for decode in (False, True):
        many = m.sample_many(conds, time_steps=2, cond_scale=1.3, mel_pp=True, cfm_method=cfm_method, noise=noise,
                             decode_to_audio=decode)
        for c, z, got in zip(conds, noise, many):
            one = m.sample(cond=c[None], time_steps=2, cond_scale=1.3, mel_pp=True, cfm_method=cfm_method, noise=z,
                           decode_to_audio=decode)
            assert got.shape == one.shape and torch.equal(got, one)
    if cfm_method != "basic_cfm":
        # caller-supplied prior scales (cfm:171-183: honoured only when BOTH are given) reach the ragged path as they reach sample()
        many = m.sample_many(conds, time_steps=1, cfm_method=cfm_method, noise=noise, decode_to_audio=False, std_1=0.9, std_2=0.2)
        half = m.sample_many(conds, time_steps=1, cfm_method=cfm_method, noise=noise, decode_to_audio=False, std_2=0.2)
        dflt = m.sample_many(conds, time_steps=1, cfm_method=cfm_method, noise=noise, decode_to_audio=False)
        for c, z, got, h, d in zip(conds, noise, many, half, dflt):
            one = m.sample(cond=c[None], time_steps=1, cfm_method=cfm_method, noise=z, decode_to_audio=False, std_1=0.9, std_2=0.2)
            assert torch.equal(got, one) and torch.equal(h, d) and not torch.equal(got, d)


# ------------------------------------------------------------------------------------------
# bench.py under a launcher with ONE rank: the RCCL init path and the scatter / gather check on a 1-GPU box
# ------------------------------------------------------------------------------------------
def test_bench_under_torchrun_one_rank_initialises_rccl_and_checks_the_sharded_path():
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", "29611", str(root / "bench.py"), "--gpus", "1", "--config", "4", "--batch", "2", "--steps", "2",
           "--warmup", "1", "--no-cpu-baseline", "--no-alt"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["config"]["rccl_world_size"] == 1 and line["n_gpus"] == 1
    assert "bit-identical" in line["config"]["sharded_check"]


def test_from_local_through_a_weight_blob_gives_the_same_bits(tmp_path, monkeypatch):
    """SURVEY.md 8f-3 (reference load path flowhighsr.py:110-149): `python -m flowhigh_amd.convert` once, then from_local maps the
    blob and uploads it with one copy -- no torch.load, no packing -- and generate() returns the bits of the model built from
    the checkpoint files; a blob that does not belong to the files is ignored."""
    from flowhigh_amd import convert, weights
    cfg = synth.TINY_CFG
    synth.write_checkpoint_dir(tmp_path, cfg, seed=5)
    monkeypatch.setenv("FH_BLOB", "0")
    ref_model = FlowHighSR.from_local(tmp_path, "cuda", torchdiffeq_ode_method="euler")
    clip, noise = synth.lowres_clip(3, 1.0, 12000), synth.prior_noise(3, 100)
    ref = ref_model.generate(clip, 12000, 48000, 1, noise=noise)
    convert.convert(tmp_path)
    monkeypatch.delenv("FH_BLOB")
    calls = []
    import flowhigh_amd.flowhighsr as M
    monkeypatch.setattr(M, "_load_checkpoint", lambda p: calls.append(p) or (_ for _ in ()).throw(AssertionError("checkpoint read")))
    m = FlowHighSR.from_local(tmp_path, "cuda", torchdiffeq_ode_method="euler")
    assert not calls
    got = m.generate(clip, 12000, 48000, 1, noise=noise)
    assert torch.equal(got, ref)
    # other checkpoint content behind the same blob: refused (and then the checkpoints ARE read)
    synth.write_checkpoint_dir(tmp_path, cfg, seed=6)
    with pytest.raises(AssertionError, match="checkpoint read"):
        FlowHighSR.from_local(tmp_path, "cuda")
    assert "other checkpoint files" in weights.WeightStore.why
