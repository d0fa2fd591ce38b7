"""profiles/<tag>_* from gpurun_out/prof_<tag>/ (tools/profile_round.sh): kernel stats CSV, bench line, summary."""
import csv, json, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
src = f'gpurun_out/prof_{tag}'
shutil.copy(f'{src}/kernel_stats.csv', f'profiles/{tag}_kernel_stats_bench_B1.csv')
shutil.copy(f'{src}/bench_line.json', f'profiles/{tag}_bench_line_B1.json')
shutil.copy(f'{src}/conv_hbm_bytes_per_launch.json', 'profiles/conv_hbm_bytes_per_launch.json')
line = json.loads(open(f'{src}/bench_line.json').read())
prof = json.loads([l for l in open(f'{src}/stats.log').read().splitlines() if l.startswith('{')][-1])
traffic = json.loads(open(f'{src}/conv_hbm_bytes_per_launch.json').read())
rows = list(csv.DictReader(open(f'{src}/kernel_stats.csv')))
steps = 10.0    # 8 timed + 2 warm-up steps in the profiled command
short = lambda n: n.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
conv = [r for r in rows if 'conv_mfma_kernel' in r['Name'] or 'conv_wino_kernel' in r['Name']]
calls = sum(int(r['Calls']) for r in conv); tot = sum(float(r['TotalDurationNs']) for r in conv)
rl = line['roofline']
o = [f"# Round 1 profile summary (1 x MI355X, B = 1, 10 s clip, 12 -> 48 kHz, euler x 1, SYNTH-CFG)", "",
     "Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline`",
     f"(raw: `{tag}_kernel_stats_bench_B1.csv`; HBM traffic PMC passes `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE`: `conv_hbm_bytes_per_launch.json`;",
     "reproduce with `tools/profile_round.sh` on the GPU box).", "",
     f"* all conv launches (conv_wino_kernel + conv_mfma_kernel): {calls} launches, average duration **{tot / calls / 1e3:.1f} us** under the "
     f"profiler; bench.py HIP events in the same run: {prof['roofline']['avg_launch_us']} us; un-profiled bench run: {rl['avg_launch_us']} us "
     f"-> {rl['achieved']} TFLOP/s algorithmic = {rl['frac']} of the 157.3 TFLOP/s fp32 MFMA peak "
     f"({rl['mfma_executed']} TFLOP/s actually executed on the matrix cores = {rl['mfma_executed_frac']}: the Winograd launches do 1.5 ceil(k/3) "
     f"instead of k multiply-adds per output).",
     f"* HBM traffic per conv launch (PMC, corrected as the guide prescribes): {traffic['bytes_per_launch'] / 1e6:.1f} MB.",
     f"* bench line: value {line['value']} audio-s/s, {line['ms_per_step']} ms per step"
     + (f", cpu_baseline {line['cpu_baseline']['value']} audio-s/s on {line['cpu_baseline']['cores']} threads." if line.get('cpu_baseline') else "."),
     "", "| kernel | launches/step | ms/step | avg us |", "|---|---|---|---|"]
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:18]:
    o.append(f"| `{short(r['Name'])}` | {int(r['Calls']) / steps:.1f} | {float(r['TotalDurationNs']) / steps / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.1f} |")
open(f'profiles/{tag}_summary.md', 'w').write("\n".join(o) + "\n")
print("\n".join(o))
