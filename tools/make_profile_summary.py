"""profiles/<tag>_* from gpurun_out/prof_<tag>/ (tools/profile_round.sh): kernel stats CSV, bench line, summary."""
import csv, json, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else 'r02'
bsuf = sys.argv[2] if len(sys.argv) > 2 else 'B1'        # file-name suffix: B1 (default workload) or e.g. B32 (PROF_ARGS='--batch 32')
src = f'gpurun_out/prof_{sys.argv[3] if len(sys.argv) > 3 else tag}'         # (third argument: the profile_round.sh tag when it differs, e.g. r04b32)
shutil.copy(f'{src}/kernel_stats.csv', f'profiles/{tag}_kernel_stats_bench_{bsuf}.csv')
shutil.copy(f'{src}/bench_line.json', f'profiles/{tag}_bench_line_{bsuf}.json')
for k in ('conv', 'act'):
    shutil.copy(f'{src}/{k}_hbm_bytes_per_launch.json', f'profiles/{k}_hbm_bytes_per_launch.json' if bsuf == 'B1' else f'profiles/{k}_hbm_bytes_per_launch_{bsuf}.json')
line = json.loads(open(f'{src}/bench_line.json').read())
prof = json.loads([l for l in open(f'{src}/stats.log').read().splitlines() if l.startswith('{')][-1])
tconv = json.loads(open(f'{src}/conv_hbm_bytes_per_launch.json').read())
tact = json.loads(open(f'{src}/act_hbm_bytes_per_launch.json').read())
rows = list(csv.DictReader(open(f'{src}/kernel_stats.csv')))
steps = float(prof['steps'] + prof['warmup'])       # timed + warm-up steps of the profiled command
short = lambda n: n.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
def agg(keys):
    sel = [r for r in rows if any(k in r['Name'] for k in keys)]
    return sum(int(r['Calls']) for r in sel), sum(float(r['TotalDurationNs']) for r in sel)
ccalls, ctot = agg(('conv_mfma_kernel', 'conv_wino_kernel', 'conv_wino54_kernel', 'amp_actconv_kernel', 'narrow_bf_kernel'))
acalls, atot = agg(('act1d_strip_kernel',))
rl, rh = line['roofline'], line['roofline_hbm']
rnd = int(''.join(ch for ch in tag[1:3] if ch.isdigit()))
prl, prh = prof['roofline'], prof['roofline_hbm']
o = [f"# Round {rnd} profile summary (1 x MI355X, B = {bsuf[1:]}, 10 s clips, 12 -> 48 kHz, euler x 1, SYNTH-CFG)", "",
     f"Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps {prof['steps']} --warmup {prof['warmup']} --no-cpu-baseline --no-alt`",
     f"(raw: `{tag}_kernel_stats_bench_{bsuf}.csv`; HBM traffic PMC passes `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` of `bench.py --steps 1 --warmup 1`:",
     "`conv_hbm_bytes_per_launch.json`, `act_hbm_bytes_per_launch.json`; reproduce with `tools/profile_round.sh` on the GPU box).", "",
     f"* conv form `{line['config'].get('conv_form')}` (dtype: {line['dtype']}).  Dominant kernel family `{next(iter(rl['by_family']))}`: {rl['kernel']}: "
     f"{rl['launches_per_step']} launches per step, {rl['avg_launch_us']} us each (HIP events, un-profiled run; profiled run: {prl['avg_launch_us']} us) "
     f"-> **{rl['achieved']} TFLOP/s on its matrix instructions = {rl['frac']} of the {rl['peak']} TFLOP/s peak**.",
     f"* all conv launches (conv_wino54_kernel + narrow_bf_kernel / amp_actconv_kernel + conv_wino_kernel + conv_mfma_kernel): {ccalls} launches, average duration **{ctot / ccalls / 1e3:.1f} us** under the "
     f"profiler; bench.py HIP events in the same run: {prl['all_conv']['avg_launch_us']} us; un-profiled bench run: {rl['all_conv']['avg_launch_us']} us "
     f"-> {rl['all_conv']['executed_fp32_equiv_tflops']} TFLOP/s executed in fp32-equivalent FLOPs = {rl['all_conv']['frac_of_fp32_mfma_peak']} of the 157.3 TFLOP/s fp32 MFMA peak "
     f"({rl['all_conv']['executed_gflop_per_launch']} GFLOP per launch; direct-form equivalent {rl['all_conv']['algorithmic_equiv']} TFLOP/s = {rl['all_conv']['algorithmic_equiv_frac']}: "
     f"the Winograd launches do 1.6 ceil(k/4) (F(5,4)) or 1.5 ceil(k/3) (F(4,3)) instead of k multiply-adds per output).",
     "* by kernel family (un-profiled bench run, HIP events): " + "; ".join(
         f"`{k}` {v['launches_per_step']} launches {v['ms_per_step']} ms {v['matrix_tflops']} TFLOP/s ({v['matrix_instructions']}) = {v['frac']} of {v['peak']}" for k, v in rl['by_family'].items()) + ".",
     f"* HBM traffic per conv launch (PMC, corrected as the guide prescribes): {tconv['bytes_per_launch'] / 1e6:.1f} MB.",
     f"* all Activation1d launches (act1d_strip_kernel): {acalls} launches, average duration **{atot / acalls / 1e3:.1f} us** under the profiler; "
     f"bench.py HIP events in the same run: {prh['avg_launch_us']} us; un-profiled: {rh['avg_launch_us']} us -> "
     f"**{rh['achieved']} GB/s algorithmic = {rh['frac']} of 8 TB/s** ({rh['algorithmic_mb_per_launch']} MB per launch: every sample read and written once).",
     f"* HBM traffic per activation launch (PMC): {tact['bytes_per_launch'] / 1e6:.1f} MB "
     f"(read {tact['read_bytes_per_launch'] / 1e6:.1f} + written {tact['write_bytes_per_launch'] / 1e6:.1f}) against {rh['algorithmic_mb_per_launch']} MB algorithmic.",
     f"* bench line: value {line['value']} audio-s/s, {line['ms_per_step']} ms per step"
     + (f", cpu_baseline {line['cpu_baseline']['value']} audio-s/s on {line['cpu_baseline']['cores']} threads ({line['cpu_baseline']['sample']})." if line.get('cpu_baseline') else "."),
     "", "| kernel | launches/step | ms/step | avg us |", "|---|---|---|---|"]
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:20]:
    o.append(f"| `{short(r['Name'])}` | {int(r['Calls']) / steps:.1f} | {float(r['TotalDurationNs']) / steps / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.1f} |")
open(f'profiles/{tag}_summary.md' if bsuf == 'B1' else f'profiles/{tag}_summary_{bsuf}.md', 'w').write("\n".join(o) + "\n")
print("\n".join(o))
