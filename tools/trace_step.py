"""Print the per-launch timeline of the last generate() step in a rocprofv3 kernel-trace CSV."""
import csv, glob, sys
f = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob('gpurun_out/prof*/**/*_kernel_trace.csv', recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'resample_poly' in r['Kernel_Name']]
step = rows[idx[-1]:]
t0 = int(step[0]['Start_Timestamp'])
prev_end, tot_gap = t0, 0
for r in step:
    name = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('(anonymous namespace)::', '')[:34]
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = s - prev_end
    tot_gap += max(gap, 0)
    print(f"{(s-t0)/1e3:9.1f}us dur {(e-s)/1e3:8.1f}us gap {gap/1e3:6.1f} grid {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):>7} "
          f"vgpr {r['VGPR_Count']}+{r['Accum_VGPR_Count']} lds {r['LDS_Block_Size']} {name}")
    prev_end = e
print('total', (prev_end - t0) / 1e6, 'ms; gaps', tot_gap / 1e6, 'ms')
