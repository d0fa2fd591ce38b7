"""Per-kernel sums of every counter in a rocprofv3 --pmc output directory."""
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + '/**/*_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        if len(sys.argv) > 2 and sys.argv[2] not in name: continue
        agg[name][r['Counter_Name']] += float(r['Counter_Value']); cnt[name].add(r['Dispatch_Id'])
for name, c in sorted(agg.items()):
    print(name, 'dispatches', len(cnt[name]))
    for k, v in sorted(c.items()):
        print(f'    {k:30s} {v / len(cnt[name]):.4g}')
