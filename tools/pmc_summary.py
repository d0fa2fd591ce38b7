"""Summarise rocprofv3 --pmc counter_collection.csv per kernel (sum over dispatches of the last step)."""
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + '/**/*_counter_collection.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
# keys: Dispatch_Id, Kernel_Name, Counter_Name, Counter_Value, ...
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for r in rows:
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    if 'conv_mfma' in name or 'act1d' in name or 'gemm_kernel' in name:
        agg[name][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[name].add(r['Dispatch_Id'])
for name, c in sorted(agg.items()):
    print(name, 'dispatches', len(cnt[name]))
    for k, v in sorted(c.items()):
        print(f'    {k:28s} {v:.4g}')
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and 'SQ_BUSY_CYCLES' in c:
        print('    mfma_busy/busy_cycles', c['SQ_VALU_MFMA_BUSY_CYCLES'] / c['SQ_BUSY_CYCLES'])
    if 'SQ_WAVE_CYCLES' in c:
        wc = c['SQ_WAVE_CYCLES']
        for k in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY'):
            if k in c: print(f'    {k}/WAVE_CYCLES {c[k]/wc:.3f}')
