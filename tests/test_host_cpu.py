"""CPU: host-side logic of the drop-in class that needs no GPU (checkpoint key contract, safe loading)."""
import pytest
from pathlib import Path
import torch

from flowhigh_amd import synth
from flowhigh_amd.flowhighsr import _load_checkpoint, check_state_dict_keys, expected_state_keys


@pytest.mark.parametrize("cfgname", ["SYNTH_CFG", "TINY_CFG", "ALT_CFG", "ALT2_CFG", "ALT3_CFG", "ODD_CFG", "NK4_CFG",
                                     "NK5_AMP2_CFG", "PAD_CFG"])
def test_expected_keys_are_the_reference_state_dict_keys(cfgname):
    """synth.make_state_dict is loaded into the REAL reference modules with strict=True by oracle/make_golden.py
    and tests/test_reference_pin.py, so its key set is the reference's; expected_state_keys must reproduce it."""
    cfg = getattr(synth, cfgname)
    sd = synth.make_state_dict(cfg, 0)
    assert sorted(expected_state_keys(cfg)) == sorted(sd)
    check_state_dict_keys(sd, cfg)


def test_strict_key_check_reports_missing_and_unexpected():
    cfg = synth.TINY_CFG
    sd = synth.make_state_dict(cfg, 0)
    sd.pop("flowhigh.to_pred.weight")
    sd["flowhigh.some_new_buffer"] = torch.zeros(1)
    with pytest.raises(RuntimeError) as e:
        check_state_dict_keys(sd, cfg)
    assert "Missing key(s)" in str(e.value) and "flowhigh.to_pred.weight" in str(e.value)
    assert "Unexpected key(s)" in str(e.value) and "flowhigh.some_new_buffer" in str(e.value)


class _Evil:
    def __reduce__(self):
        return (print, ("pickle payload executed",))


def test_checkpoints_load_with_the_safe_unpickler(tmp_path, monkeypatch):
    synth.write_checkpoint_dir(tmp_path, synth.TINY_CFG, 0, weight_norm=True)
    pkg = _load_checkpoint(tmp_path / "FLowHigh_basic_400k.pt")
    assert "model" in pkg and "flowhigh.to_embed.weight" in pkg["model"]
    assert "generator" in _load_checkpoint(tmp_path / "bigvgan_48khz_256band.pt")
    torch.save({"model": {}, "extra": _Evil()}, tmp_path / "evil.pt")
    monkeypatch.delenv("FH_UNSAFE_LOAD", raising=False)
    with pytest.raises(RuntimeError, match="weights_only"):
        _load_checkpoint(tmp_path / "evil.pt")


# ------------------------------------------------------------------------------------------
# vocoder launch plans: host logic only (descriptors are built against CPU tensors, nothing is launched)
# ------------------------------------------------------------------------------------------
def _cpu_vocoder(cfgname):
    from flowhigh_amd.vocoder import Vocoder
    cfg = getattr(synth, cfgname)
    return Vocoder(cfg, synth.make_vocoder_state_dict(cfg, 1), "cpu")


@pytest.mark.parametrize("cfgname", ["SYNTH_CFG", "ALT_CFG", "ALT2_CFG", "ALT3_CFG", "ODD_CFG", "NK4_CFG", "NK5_AMP2_CFG",
                                     "PAD_CFG"])
def test_plan_steps_have_unique_sorted_position_keys(cfgname):
    """Every launch of a plan is tagged (stage, sub-block, slot, index); plan_ragged merges the plans of different
    clips by that key, so within one plan the keys must be unique and already in launch order."""
    voc = _cpu_vocoder(cfgname)
    for n in (7, 50, 333):
        p = voc.plan(1, n)
        keys = [m[0] for m in p["meta"]]
        assert len(keys) == len(p["steps"]) and len(set(keys)) == len(keys) and keys == sorted(keys)
        for step, (_, structs) in zip(p["steps"], p["meta"]):
            assert (structs is not None) == (step[0] in ("conv", "convt", "wino", "act", "amp"))
            if structs is not None:
                assert len(structs) == step[2]


def test_odd_upsamplers_carry_the_reference_lengths():
    """ConvTranspose1d(k, u, padding (k - u) // 2) with k - u odd returns u L + 1 samples
    (/root/reference/src/flowhigh/models/bigvgan/models.py:141-146): the plan's stage lengths, the phase groups of the
    upsampler launches (phase 0 has one position more; Winograd groups mask by out_len and read rows of xlen) and the
    waveform length follow torch's ConvTranspose1d."""
    from flowhigh_amd import vocoder as V
    voc = _cpu_vocoder("ODD_CFG")
    cfg = synth.ODD_CFG
    for n in (7, 20, 25):
        want, L = [], n
        for u, k in zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"]):
            L = torch.nn.ConvTranspose1d(1, 1, k, u, padding=(k - u) // 2)(torch.zeros(1, 1, L)).shape[-1]
            want.append(L)
        assert voc.stage_lengths(n) == want and voc.out_len(n) == 480 * n + 98
        p = voc.plan(2, n)
        assert tuple(p["wav"].shape) == (2, want[-1])
        ups = [(st, m[1]) for st, m in zip(p["steps"], p["meta"]) if m[0][1] == -1 and st[0] in ("conv", "convt", "wino")]
        assert len(ups) == 5
        lin = n
        for i, (st, groups) in enumerate(ups):
            u, extra = cfg["upsample_rates"][i], (cfg["upsample_kernel_sizes"][i] - cfg["upsample_rates"][i]) % 2
            if st[0] == "wino":
                assert st[4] == lin + extra and bool(st[7] & V.WINO_NOVL) == bool(extra)
                for g in groups:
                    assert g.len == lin + extra and g.out_stride == u
                    assert (g.out_len, g.seg[0].xlen) == ((want[i], lin) if extra else (0, 0))
            elif st[0] == "convt":          # all u phases in one block (even k - u, stride 2 or 3): one group, segment = phase
                assert extra == 0 and st[4] == lin and st[6] == u and len(groups) == 1
                g = groups[0]
                assert (g.lin, g.lout, g.n_len, g.out_stride, g.out_phase, g.nseg) == (lin, want[i], lin, u, 0, u)
            else:
                assert st[4] == lin + extra
                for r, g in enumerate(groups):
                    assert (g.lin, g.lout, g.out_stride, g.out_phase) == (lin, want[i], u, r)
                    assert g.n_len == lin + (extra if r == 0 else 0)
            lin = want[i]
    voc = _cpu_vocoder("SYNTH_CFG")
    assert voc.stage_lengths(10) == [50, 200, 600, 1200, 2400, 4800]


def test_channel_counts_are_padded_to_multiples_of_eight_with_zero_weights():
    """200 -> 100 / 50 / 25 / 12 channels run as 104 / 56 / 32 / 16 with zero rows / columns: exact zeros flow through."""
    voc = _cpu_vocoder("PAD_CFG")
    assert (voc.true_c0, voc.c0) == (200, 200) and voc.true_chans == [100, 50, 25, 12] and voc.chans == [104, 56, 32, 16]
    assert voc.post_w.shape == (16, 7) and float(voc.post_w[12:].abs().max()) == 0.0
    assert voc.stages[1]["extra"] == 1 and voc.stages[0]["extra"] == 0
    assert float(voc.stages[0]["up_b"][100:].abs().max()) == 0.0


def test_plan_ragged_merges_launch_by_launch():
    """One merged launch per position and kernel class: the group count of a merged launch is the sum over the
    clips, activation groups carry their own length and an exact tile prefix, Winograd run maps list only runs that
    hold real tiles, and two clips of equal length get plans (buffers) of their own."""
    import ctypes as C
    from flowhigh_amd import hip
    from flowhigh_amd import vocoder as V
    voc = _cpu_vocoder("SYNTH_CFG")
    frames = [50, 333, 50, 120]
    rp = voc.plan_ragged(frames)
    assert rp["subs"][0] is not rp["subs"][2] and rp["subs"][0]["wav"].data_ptr() != rp["subs"][2]["wav"].data_ptr()
    per_clip = [voc.plan(1, n, inst=i) for n, i in zip(frames, (0, 0, 1, 0))]
    assert all(a is b for a, b in zip(per_clip, rp["subs"]))
    raw = bytes(rp["desc"].numpy().tobytes())
    tt = hip.lib().fh_act_tile_len()
    n_act_groups = sum(s[2] for p in per_clip for s in p["steps"] if s[0] == "act")
    n_wino_groups = sum(s[2] for p in per_clip for s in p["steps"] if s[0] == "wino")
    got_act = got_wino = 0
    for s in rp["steps"]:
        if s[0] == "ract":
            _, off, ng, c, din, dout, tiles, mult4 = s
            gs = (hip.ActGroup * ng).from_buffer_copy(raw[off:off + ng * C.sizeof(hip.ActGroup)])
            base = 0
            for g in gs:
                assert g.tile_base == base and g.len > 0
                base += c * -(-g.len // tt)
            assert base == tiles and mult4 == int(all(g.len % 4 == 0 for g in gs))
            assert [g.len for g in gs] == sorted((g.len for g in gs), reverse=True)
            got_act += ng
        elif s[0] == "rwino":
            _, off, ng, wpad, maxlen, dil, wcfg, pmflag, off_map, n_runs = s
            gs = (hip.WinoGroup * ng).from_buffer_copy(raw[off:off + ng * C.sizeof(hip.WinoGroup)])
            assert max(g.len for g in gs) == maxlen
            runs = (C.c_int32 * n_runs).from_buffer_copy(raw[off_map:off_map + 4 * n_runs])
            f54 = wcfg & V.WINO_F54                             # (the F(5,4) kernel: 128 / 96 / 64 co x 320 outputs)
            bm, bt = {0: (64, 512), 1: (96, 256), 4: (64, 256), 5: (32, 256), 6: (128, 256),
                      V.WINO_F54: (128, 320), V.WINO_F54 | 1: (96, 320), V.WINO_F54 | 2: (64, 320), V.WINO_F54 | 3: (48, 320)}[wcfg]
            assert hip.lib().fh_wino54_tile_n() == 320 and hip.lib().fh_wino54_tile_m(wcfg & 15) == bm if f54 else True
            pm = pmflag & 1
            n_tiles = V.wino_n_tiles(wcfg, maxlen, dil, pm)
            if f54:
                assert n_tiles == hip.lib().fh_wino54_n_tiles(maxlen, dil, pm)
            run_len = (hip.lib().fh_wino54_run_len if f54 else hip.lib().fh_wino_run_len)(n_tiles)
            rpp = -(-n_tiles // run_len)
            assert len(set(runs)) == n_runs and all(0 <= r < ng * (wpad // bm) * rpp for r in runs)
            # real tiles of a group: per phase (tile = block-in-phase * dil + phase) in general; the F(5,4) kernel tiles
            # phase-major rows as one sequence, a group's real tiles are then the first fh_wino54_n_tiles(its len)
            real = (lambda t, g: t < V.wino_n_tiles(wcfg, g.len, dil, pm)) if (f54 and pm) else \
                (lambda t, g: (t % dil) + dil * bt * (t // dil) < g.len)
            for r in runs:                                   # the run's first tile is a real tile of its group
                g = gs[(r // rpp) // (wpad // bm)]
                assert real((r % rpp) * run_len, g)
            # ... and every real tile is covered
            want = sum((wpad // bm) * len({t // run_len for t in range(n_tiles) if real(t, g)}) for g in gs)
            assert want == n_runs
            got_wino += ng
    assert got_act == n_act_groups and got_wino == n_wino_groups
    assert voc.plan_ragged(frames) is rp                               # cached


def test_chunk_geometry_covers_the_receptive_field():
    """halo >= the receptive field of a waveform sample in mel frames (counted layer by layer here), and chunk starts
    keep every stage's Winograd tile position and dilation phase."""
    import math
    for cfgname in ("SYNTH_CFG", "ALT_CFG", "ALT2_CFG", "ALT3_CFG", "ODD_CFG", "NK4_CFG"):
        voc = _cpu_vocoder(cfgname)
        halo, align = voc.chunk_geometry()
        cfg = getattr(synth, cfgname)
        rates, ks, dils = cfg["upsample_rates"], cfg["resblock_kernel_sizes"], cfg["resblock_dilation_sizes"]
        h = 3 + 6                                                # conv_post, activation_post (samples at the output rate)
        for i in reversed(range(len(rates))):
            per_block = []
            for k, dl in zip(ks, dils):
                if str(cfg["resblock"]) == "1":
                    per_block.append(sum(6 + (k - 1) // 2 * d + 6 + (k - 1) // 2 for d in dl))
                else:
                    per_block.append(sum(6 + (k - 1) // 2 * d for d in dl))
            h += max(per_block)
            h = math.ceil((h + cfg["upsample_kernel_sizes"][i]) / rates[i])      # transposed conv, seen from its input
        h += 3                                                   # conv_pre
        assert halo >= h, (cfgname, halo, h)
        rate = 1
        for u in rates:
            assert (align * rate) % 4 == 0                       # input of the (Winograd) transposed conv
            rate *= u
            for dl in dils:
                for d in dl:
                    assert (align * rate) % (4 * d) == 0         # tile position and phase at this stage


def test_on_device_runs_methods_under_the_models_device(monkeypatch):
    """hip.on_device: every decorated entry (plain and generator methods) runs with the object's device current, whatever
    the caller has current (the reference's from_local(ckpt_dir, device), flowhighsr.py:110-137).  No GPU here: the
    guard is replaced by a recorder."""
    import contextlib
    from flowhigh_amd import hip
    log = []

    @contextlib.contextmanager
    def fake_guard(device):
        log.append(("enter", str(device)))
        yield
        log.append(("exit", str(device)))
    monkeypatch.setattr(hip, "device_guard", fake_guard)

    class Obj:
        device = torch.device("cuda", 3)

        @hip.on_device
        def f(self, a, b=2):
            log.append(("f", a, b))
            return a + b

        @hip.on_device
        def g(self, n):
            for i in range(n):
                log.append(("g", i))
                yield i

    o = Obj()
    assert o.f(1, b=5) == 6
    assert log == [("enter", "cuda:3"), ("f", 1, 5), ("exit", "cuda:3")]
    del log[:]
    it = o.g(2)
    assert log == []                                     # nothing runs before the first next()
    assert next(it) == 0
    assert log == [("enter", "cuda:3"), ("g", 0), ("exit", "cuda:3")]     # the consumer's code runs OUTSIDE the guard
    assert list(it) == [1]
    assert log.count(("enter", "cuda:3")) == log.count(("exit", "cuda:3")) == 3
    # every public entry of the product classes is guarded
    from flowhigh_amd import flow, flowhighsr, frontend, vocoder
    for cls, names in ((flowhighsr.FlowHighSR, ["generate", "generate_batch", "generate_many", "generate_from_device",
                                                "sample", "sample_many", "capture", "load", "mel_cutoff_bins"]),
                       (flowhighsr.GraphedGenerate, ["replay"]),
                       (vocoder.Vocoder, ["plan", "plan_ragged", "run", "run_ragged", "forward", "forward_ragged",
                                          "forward_chunks", "forward_chunked"]),
                       (flow.FlowNet, ["workspace", "ragged_workspace", "set_cond", "forward"]),
                       (frontend.LogMel, ["__call__"]), (frontend.PostProcessor, ["__call__"]),
                       (frontend.Resampler, ["__call__"])):
        for n in names:
            assert hasattr(getattr(cls, n), "__wrapped__"), f"{cls.__name__}.{n} is not under hip.on_device"


def test_act_occupancy_rule_keeps_the_default_unless_a_cap_pays():
    from flowhigh_amd import vocoder as V
    assert V.pick_act_blocks({0: 615.0, 4: 590.0}) == 4                  # a box with the clock dip
    assert V.pick_act_blocks({0: 615.0, 4: 590.0, 3: 562.0}) == 3
    assert V.pick_act_blocks({0: 540.0, 4: 548.0}) == 0                  # a box without it
    assert V.pick_act_blocks({0: 600.0, 4: 592.0}) == 0                  # within 2 %: keep the default
    # the cap must pay in EVERY pass; among those that do, the lowest median wins
    assert V.pick_act_blocks([{0: 615.0, 3: 560.0}, {0: 600.0, 3: 597.0}]) == 0
    assert V.pick_act_blocks([{0: 615.0, 3: 560.0, 4: 570.0}, {0: 610.0, 3: 575.0, 4: 560.0}]) in (3, 4)
    assert V.pick_act_blocks([{0: 615.0, 3: 560.0, 4: 600.0}, {0: 610.0, 3: 575.0, 4: 605.0}]) == 3
    assert V.calibrate_act_occupancy(torch.device("cpu")) == 0            # nothing to set without a GPU
    # FH_ACT_BLOCKS / act_blocks=: validated, empty = auto (an empty export must not kill every construction)
    assert V.parse_act_blocks(None) is None and V.parse_act_blocks("") is None and V.parse_act_blocks(" auto ") is None
    assert V.parse_act_blocks("0") == 0 and V.parse_act_blocks(3) == 3 and V.parse_act_blocks("5") == 5
    for bad in ("1", "6", "three", "-2"):
        with pytest.raises(ValueError, match="FH_ACT_BLOCKS"):
            V.parse_act_blocks(bad)


def test_norm_device_pins_the_ordinal():
    from flowhigh_amd import hip
    assert hip.norm_device("cuda:2") == torch.device("cuda", 2)
    assert hip.norm_device("cpu") == torch.device("cpu")
    if not torch.cuda.is_available():
        assert hip.norm_device("cuda") == torch.device("cuda")          # nothing to pin without a GPU


def test_f54_tile_family_is_fixed_by_the_channel_count(monkeypatch):
    """Which F(5,4) block shape a stage's weights are packed for depends on its channel count only; the 48-row block (three
    16-row MFMA tiles: another summation order than the 32-row-tile blocks) is never an alternative of the launch model where
    a 96-row block fits, so a conv's bits do not depend on batch or length (conv_wino54.hip, vocoder._choose_wino_cfg)."""
    from flowhigh_amd import vocoder as V
    for var in ("FH_WINO54", "FH_WINO54_H16", "FH_WINO", "FH_CONV_FORM"):
        monkeypatch.delenv(var, raising=False)
    assert V.pick_wino54_tile(768) == (V.WINO_F54 | 0, 768) and V.pick_wino54_tile(96) == (V.WINO_F54 | 1, 96)
    assert V.pick_wino54_tile(48) == (V.WINO_F54 | 3, 48) and V.pick_wino54_tile(144) == (V.WINO_F54 | 3, 144)
    assert V.pick_wino54_tile(64) == (V.WINO_F54 | 2, 64) and V.pick_wino54_tile(24) == (V.WINO_F54 | 2, 64)
    assert V.pick_wino54_tile(240) == (V.WINO_F54 | 3, 240)
    # (144 channels are not a Winograd stage at all: use_wino)
    assert V.use_wino54(96) and V.use_wino54(48) and V.use_wino54(240) and not V.use_wino54(64) and not V.use_wino54(24)
    monkeypatch.setenv("FH_WINO54_H16", "0")
    assert V.use_wino54(96) and not V.use_wino54(48)
    monkeypatch.delenv("FH_WINO54_H16")
    # the launch model's candidates: every shape of the family that divides cout_pad, the 48-row block only where it is the one
    for wpad, batch, length in ((96, 1, 120000), (96, 32, 120000), (192, 1, 60000), (384, 8, 2000), (288, 1, 5000)):
        cfg, _ = V.choose_wino_cfg([6, 12, 18], batch, wpad, length, 1, default=V.pick_wino54_tile(wpad)[0])
        assert cfg & V.WINO_F54 and cfg != V.WINO_F54 | 3, (wpad, batch, length, cfg)
    for wpad, batch, length in ((48, 1, 240000), (48, 32, 240000), (144, 2, 700)):
        cfg, _ = V.choose_wino_cfg([3, 6, 9], batch, wpad, length, 1, default=V.WINO_F54 | 3)
        assert cfg == V.WINO_F54 | 3


def _all_tensors(obj, out, prefix=""):
    """Every tensor reachable from a constructed model's weight attributes, by path."""
    if isinstance(obj, torch.Tensor):
        out[prefix] = obj
    elif isinstance(obj, dict):
        for k, v in obj.items():
            _all_tensors(v, out, f"{prefix}.{k}")
    elif isinstance(obj, (list, tuple)):
        for i, v in enumerate(obj):
            _all_tensors(v, out, f"{prefix}[{i}]")


@pytest.mark.parametrize("cfgname", ["TINY_CFG", "NK5_AMP2_CFG"])
def test_weight_blob_gives_the_packers_tensors_bit_for_bit(tmp_path, cfgname, monkeypatch):
    """flowhigh_amd.convert (reference load path flowhighsr.py:110-149, init_vocoder.py:8-23): the blob written
    from the reference's three checkpoint files gives back, without reading them, every tensor the in-memory packers make --
    same values, shapes and host-side lists -- and is refused when the checkpoints or the layout switches changed."""
    from flowhigh_amd import convert, weights
    from flowhigh_amd.flow import FlowNet
    from flowhigh_amd.flowhighsr import CKPT_FILES, read_checkpoints
    from flowhigh_amd.vocoder import Vocoder
    cfg = getattr(synth, cfgname)
    synth.write_checkpoint_dir(tmp_path, cfg, seed=3)
    info = convert.convert(tmp_path)
    assert Path(info["blob"]).name == weights.BLOB_NAME and info["tensors"] > 100
    sd, cfg_read = read_checkpoints(tmp_path)
    from flowhigh_amd.planner import use_gemm_bf16x6
    gbf = use_gemm_bf16x6(info["form"])
    ref_net, ref_voc = FlowNet(sd, "cpu", bf=gbf), Vocoder(cfg_read, sd, "cpu")
    srcs = {f: weights.file_digest(tmp_path / f) for f in CKPT_FILES}
    store = weights.WeightStore.open(info["blob"], "cpu", expect_format=weights.format_tag(info["form"]), sources=srcs)
    assert store is not None and store.cfg == cfg_read
    net, voc = FlowNet(None, "cpu", store=store, bf=gbf), Vocoder(store.cfg, None, "cpu", store=store)
    for a, b in ((ref_net, net), (ref_voc, voc)):
        ta, tb = {}, {}
        _all_tensors({k: v for k, v in vars(a).items() if not k.startswith("_")}, ta)
        _all_tensors({k: v for k, v in vars(b).items() if not k.startswith("_")}, tb)
        assert ta.keys() == tb.keys() and len(ta) > 20
        for k in ta:
            assert ta[k].dtype == tb[k].dtype and ta[k].shape == tb[k].shape and torch.equal(ta[k], tb[k]), k
    assert ref_voc.post_act["up"] == voc.post_act["up"] and ref_voc.stages[0]["blocks"][0]["acts"][0]["down"] == voc.stages[0]["blocks"][0]["acts"][0]["down"]
    assert (net.dim, net.dim_in, net.dw_k, net.layers[0]["inner_pad"]) == (ref_net.dim, ref_net.dim_in, ref_net.dw_k, ref_net.layers[0]["inner_pad"])
    # other checkpoint content, other layout switches: not used
    assert weights.WeightStore.open(info["blob"], "cpu", sources=dict(srcs, **{CKPT_FILES[1]: "0" * 32})) is None
    monkeypatch.setenv("FH_WINO54", "0")
    assert weights.WeightStore.open(info["blob"], "cpu", expect_format=weights.format_tag(info["form"])) is None
    assert "layout switches" in weights.WeightStore.why
    monkeypatch.delenv("FH_WINO54")
    # ... another conv form: not used either; a truncated copy does not open at all; --verify finds nothing wrong with the original
    other = "winograd" if info["form"] != "winograd" else "direct"
    assert weights.WeightStore.open(info["blob"], "cpu", expect_format=weights.format_tag(other)) is None
    cut = tmp_path / "cut.blob"
    cut.write_bytes(Path(info["blob"]).read_bytes()[:-4096])
    assert weights.WeightStore.open(cut, "cpu") is None and "truncated" in weights.WeightStore.why
    (tmp_path / "short.blob").write_bytes(weights.MAGIC + b"\x10\x00")
    assert weights.WeightStore.open(tmp_path / "short.blob", "cpu") is None
    assert convert.verify(tmp_path) == []
    raw = bytearray(Path(info["blob"]).read_bytes())
    raw[-5000] ^= 1                                   # one flipped bit in the tensor bytes
    (tmp_path / "rot.blob").write_bytes(bytes(raw))
    assert any("digest" in p for p in convert.verify(tmp_path, tmp_path / "rot.blob"))


def test_conv_form_keyword_and_environment(monkeypatch):
    """conv_form = 'auto' | 'winograd' | 'bf16x6' | 'direct' (FlowHighSR.from_local / FLowHigh / Vocoder keyword, reference
    constructor flowhighsr.py:110-137): keyword > FH_CONV_FORM > the older switches > 'auto' = the default form; the form
    decides which kernels a model's weights are packed for, and a blob of another form is refused."""
    from flowhigh_amd import planner as P, weights
    from flowhigh_amd.vocoder import Vocoder
    for var in ("FH_CONV_FORM", "FH_WINO", "FH_CONV_BF16X6"):
        monkeypatch.delenv(var, raising=False)
    assert P.resolve_conv_form() == (P.DEFAULT_CONV_FORM, True) == P.resolve_conv_form("auto")
    assert P.resolve_conv_form("direct") == ("direct", False) and P.resolve_conv_form(None, bf16x6=False) == ("winograd", False)
    monkeypatch.setenv("FH_CONV_FORM", "winograd")
    assert P.resolve_conv_form() == ("winograd", False) and P.resolve_conv_form("bf16x6") == ("bf16x6", False)     # keyword wins
    monkeypatch.delenv("FH_CONV_FORM")
    monkeypatch.setenv("FH_WINO", "0")
    assert P.resolve_conv_form() == ("direct", False) and not P.use_wino(768, 1) and P.use_wino(768, 1, "winograd")
    monkeypatch.delenv("FH_WINO")
    monkeypatch.setenv("FH_CONV_BF16X6", "0")
    assert P.resolve_conv_form() == ("winograd", False)
    monkeypatch.delenv("FH_CONV_BF16X6")
    with pytest.raises(ValueError):
        P.resolve_conv_form("fp8")
    cfg, sd = synth.SYNTH_CFG, synth.make_vocoder_state_dict(synth.SYNTH_CFG, 1)
    forms = {f: Vocoder(cfg, sd, "cpu", conv_form=f) for f in ("winograd", "bf16x6", "direct")}
    for f, voc in forms.items():
        assert voc.form == f and not voc.form_auto and voc.bf == (f == "bf16x6")
        wide, narrow = voc.stages[0], voc.stages[-1]
        e = wide["blocks"][0]["c1"][0]
        assert ("u" in e) == (f != "direct") and ("w" in e) == (f == "direct")
        assert ("ua" in narrow["blocks"][0]["c1"][0]) == (f != "direct")            # a narrow-stage kernel in both other forms:
        assert voc.amp_direct == (f == "bf16x6")                                    # fp32 Winograd / direct bf16 x 6
        if f != "direct":
            c, k = narrow["c"], voc.ks[0]
            from flowhigh_amd import packing
            want = packing.pack_narrow_bf_weight if f == "bf16x6" else packing.pack_amp_weight
            assert narrow["blocks"][0]["c1"][0]["ua"].numel() == want(torch.zeros(c, c, k), c).numel()
        if f != "direct":
            assert e["u"].dtype == (torch.int16 if f == "bf16x6" else torch.float32)
            assert wide["w54"] and wide["wcfg"] == (P.WINO_F54 | 1 if f == "bf16x6" else P.WINO_F54 | 0)
    assert Vocoder(cfg, sd, "cpu").form == P.DEFAULT_CONV_FORM and Vocoder(cfg, sd, "cpu").form_auto
    # the bf16 x 6 launch plan: wide stages on the F(5,4) bf16 x 6 kernel (96- / 64-row blocks only), narrow ones on the direct
    # bf16 x 6 kernel
    fams = {n for n, _, _ in forms["bf16x6"].plan(1, 100)["conv_launches"]}
    assert fams == {"wino54_bf16x6", "wino43_bf16x6", "narrow_bf16x6", "direct"}
    assert {n for n, _, _ in forms["winograd"].plan(1, 100)["conv_launches"]} == {"wino54", "wino43", "amp", "direct"}
    # transposed convs as Winograd phase groups: from 768 input channels on in the fp32 form, from 192 in the bf16 x 6 form
    # (SYNTH-CFG: 1536 -> 768 -> 384 -> 192 -> 96 -> 48 -> 24)
    count = lambda f, fam: sum(n == fam for n, _, _ in forms[f].plan(1, 100)["conv_launches"])
    assert (P.wino_ups_min_cin("winograd"), P.wino_ups_min_cin("bf16x6")) == (768, 192)
    assert (count("winograd", "wino43"), count("winograd", "direct")) == (3, 4)          # conv_pre + 2 upsamplers | 4 upsamplers
    assert (count("bf16x6", "wino43_bf16x6"), count("bf16x6", "direct")) == (5, 2)
    assert {n for n, _, _ in forms["direct"].plan(1, 100)["conv_launches"]} == {"direct"}
    # a store that holds another form's tensors is refused at construction
    st = weights.WeightStore("cpu")
    st.form = "winograd"
    with pytest.raises(ValueError, match="packed for"):
        Vocoder(cfg, sd, "cpu", conv_form="bf16x6", store=st)
    from flowhigh_amd.flow import FlowNet
    with pytest.raises(ValueError, match="packed for"):                  # ... and by the transformer, whose linears have two forms too
        FlowNet(synth.make_flow_state_dict(seed=0), "cpu", store=st, bf=True)
    assert weights.format_tag("bf16x6") != weights.format_tag("winograd") != weights.format_tag("direct")
    monkeypatch.setenv("FH_AMP", "1")
    tag = weights.format_tag("bf16x6")
    monkeypatch.delenv("FH_AMP")
    assert tag == weights.format_tag("bf16x6")                # unset and "1" are the same setting


def test_plan_switches_are_read_when_the_model_is_built(monkeypatch):
    """FH_WINO_SPLITK / FH_UPS_FUSE shape launch plans (the first one also the order of additions of short
    clips): a model takes them as they are when it is BUILT (Vocoder.sw) and plans with that snapshot for its whole life,
    whatever the environment says later.  (FH_FUSE_TAIL, FH_AMP_FUSE_ACT and FH_AMP_INTERLEAVE left with the forms they switched: round 6.)"""
    for var in ("FH_WINO_SPLITK", "FH_UPS_FUSE"):
        monkeypatch.delenv(var, raising=False)
    voc = _cpu_vocoder("SYNTH_CFG")
    assert voc.sw == dict(splitk=True, ups_fuse=True)
    kinds = [s_[0] for s_ in voc.plan(1, 50)["steps"]]
    assert "sum" in kinds and "convt" in kinds and kinds.count("act") == 37
    monkeypatch.setenv("FH_WINO_SPLITK", "0")
    monkeypatch.setenv("FH_UPS_FUSE", "0")
    voc._plans.clear()
    assert [s_[0] for s_ in voc.plan(1, 50)["steps"]] == kinds                 # the living model does not follow the environment
    voc2 = _cpu_vocoder("SYNTH_CFG")
    kinds2 = [s_[0] for s_ in voc2.plan(1, 50)["steps"]]
    assert "sum" not in kinds2 and "convt" not in kinds2 and kinds2.count("act") == 37
