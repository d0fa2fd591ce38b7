import json
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    z = np.load(GOLDEN / f"{name}.npz", allow_pickle=False)
    d = {k: z[k] for k in z.files}
    if "cfg" in d:
        d["cfg"] = json.loads(str(d["cfg"]))
        for k in ("seed", "sr_in", "steps", "cr"):
            d[k] = int(d[k])
        for k in ("method", "cfm_method"):
            d[k] = str(d[k])
        d["sigma"] = float(d["sigma"])
    return d


E2E_CASES = ["tiny_euler", "alt_midpoint", "tiny_adaptive_i16", "tiny_ragged_16k", "tiny_mix",
             "amp2_euler", "amp2_three_blocks",         # AMPBlock2 vocoders (resblock "2")
             "odd_euler",                               # upsamplers with k - u odd (u L + 1 samples per stage)
             "nk4_midpoint"]                            # four resblock kernel sizes
