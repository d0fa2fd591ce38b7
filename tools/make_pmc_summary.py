"""gpurun_out/pmc_<tag>.txt (tools/pmc_round.sh) -> profiles/<tag>_pmc_kernels.txt with the matrix-pipe busy fraction
and the instruction mix per MFMA of every MFMA kernel in the header.  python tools/make_pmc_summary.py r01"""
import re
import sys
from pathlib import Path

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out_name = sys.argv[2] if len(sys.argv) > 2 else f"{tag}_pmc_kernels.txt"        # e.g. r04_pmc_kernels_B32.txt for tag r04b32
root = Path(__file__).resolve().parents[1]
txt = (root / "gpurun_out" / f"pmc_{tag}.txt").read_text()
agg = {}
for b in re.split(r'\n(?=\S)', txt):
    lines = b.strip().split('\n')
    if not lines or 'dispatches' not in lines[0]:
        continue
    d = agg.setdefault(lines[0].split(' dispatches')[0], {})
    d['dispatches'] = int(lines[0].split()[-1])
    for line in lines[1:]:
        k, v = line.split()
        d[k] = float(v)
hdr = ["# rocprofv3 --pmc, per kernel, averages per dispatch over `python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline`",
       "# (B = 1, 10 s clip; reproduce: tools/pmc_round.sh on the GPU box, then this script).  Three passes (6 counters each):",
       "#   SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS",
       "#   SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES",
       "#   SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_LEVEL_VMEM",
       "# SQ_BUSY_CYCLES sums 32 shader engines, SQ_VALU_MFMA_BUSY_CYCLES 1024 SIMDs:",
       "#   matrix-pipe busy fraction = (MFMA_BUSY / 1024) / (BUSY_CYCLES / 32), whole launches (prologue, epilogue, tails included):"]
for n, d in sorted(agg.items()):
    if d.get('SQ_INSTS_MFMA') and d.get('SQ_BUSY_CYCLES'):
        m = d['SQ_INSTS_MFMA']
        f = (d['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024) / (d['SQ_BUSY_CYCLES'] / 32)
        hdr.append(f"#   {n} ({d['dispatches'] // 2} launches/step): {f:.2f}; per MFMA: {(d['SQ_INSTS_VALU'] - m) / m:.1f} other VALU, "
                   f"{d['SQ_INSTS_SALU'] / m:.1f} SALU, {d['SQ_INSTS_LDS'] / m:.2f} LDS, "
                   f"{(d['SQ_INSTS_VMEM_RD'] + d['SQ_INSTS_VMEM_WR']) / m:.2f} VMEM; LDS bank conflict cycles "
                   f"{d.get('SQ_LDS_BANK_CONFLICT', 0):.3g}")
(root / "profiles" / out_name).write_text("\n".join(hdr) + "\n" + txt)
print("\n".join(hdr[7:]))
