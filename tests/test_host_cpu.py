"""CPU: host-side logic of the drop-in class that needs no GPU (checkpoint key contract, safe loading)."""
import pytest
import torch

from flowhigh_amd import synth
from flowhigh_amd.flowhighsr import _load_checkpoint, check_state_dict_keys, expected_state_keys


@pytest.mark.parametrize("cfgname", ["SYNTH_CFG", "TINY_CFG", "ALT_CFG", "ALT2_CFG", "ALT3_CFG"])
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
