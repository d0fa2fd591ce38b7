import os, sys, subprocess, json
import numpy as np, torch
sys.path.insert(0, '.')
mode = sys.argv[1]
os.environ["FH_FFT"] = mode
from flowhigh_amd import frontend
from oracle import ref_cpu
g = np.load("tests/golden/tiny_euler.npz")
cond = torch.from_numpy(g["cond48"])[None]
lm = frontend.LogMel("cuda:0")
mel = lm(cond.cuda()).cpu()
ref32 = ref_cpu.logmel(cond)[0] if hasattr(ref_cpu, "logmel") else None
ref64 = ref_cpu.logmel(cond.double())[0].float()
gold = torch.from_numpy(g["cond_mel"])[0]
d_gold = (mel - gold).abs(); d64 = (mel - ref64).abs(); g64 = (gold - ref64).abs()
hi = ref64 > -8
print("FH_FFT", mode, "vs golden max", d_gold.max().item(), "(mel>-8:", d_gold[hi].max().item(), ") vs f64 max", d64.max().item(), "(mel>-8:", d64[hi].max().item(), ") golden vs f64 max", g64.max().item(), "(mel>-8:", g64[hi].max().item(), ")")
