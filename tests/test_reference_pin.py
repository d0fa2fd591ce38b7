"""CPU, build container only: the oracle against the LIVE reference (imported under the shims of
oracle/ref_shim.py) on a case that is not among the stored golden vectors.  Skipped where
/root/reference does not exist (the GPU box)."""
import tempfile

import numpy as np
import pytest
import torch

from flowhigh_amd import synth
from oracle import ref_cpu, ref_shim

pytestmark = pytest.mark.skipif(not ref_shim.available(), reason="reference tree not present")


@pytest.mark.parametrize("cfgname,sr_in,method,steps", [("ALT_CFG", 12000, "euler", 2), ("TINY_CFG", 24000, "midpoint", 1),
                                                        # upsamplers with k - u odd (the vocoder returns 480 N + 98 samples), five kernel sizes
                                                        ("ODD_CFG", 16000, "midpoint", 1), ("NK5_AMP2_CFG", 8000, "euler", 1)])
def test_oracle_equals_live_reference(cfgname, sr_in, method, steps):
    cfg = getattr(synth, cfgname)
    d = tempfile.mkdtemp(prefix="fh_pin_")
    sd = synth.write_checkpoint_dir(d, cfg, seed=11)
    model = ref_shim.build_reference_model(d, method)
    audio = synth.lowres_clip(11, 0.23, sr_in)
    torch.manual_seed(4242)
    with torch.no_grad():
        ref = model.generate(audio, sr_in, 48000, steps)
    n = ref.shape[-1] // 480
    g = torch.Generator().manual_seed(4242)
    from flowhigh_amd.flowhighsr import reference_prior_draw
    noise = reference_prior_draw(n, 256, g)
    out = ref_cpu.generate(sd, cfg, audio, sr_in, noise, steps, method)
    assert out.shape == ref.shape
    assert np.abs(out.numpy() - ref.numpy()).max() <= 2e-6


def test_reference_state_dict_contract():
    """The synthetic checkpoint loads into the reference with strict=True: key names and shapes of
    flowhigh_amd.synth are exactly the reference's (flowhighsr.py:131-135)."""
    d = tempfile.mkdtemp(prefix="fh_pin_")
    sd = synth.write_checkpoint_dir(d, synth.TINY_CFG, seed=0)
    model = ref_shim.build_reference_model(d, "euler")
    ref_sd = model.state_dict()
    assert set(ref_sd) == set(sd)
    for k, v in ref_sd.items():
        assert tuple(v.shape) == tuple(sd[k].shape), k
