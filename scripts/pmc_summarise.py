"""Per-kernel sums of the rocprofv3 --pmc passes written by scripts/pmc_step.sh -> markdown + csv."""
import csv, glob, os, sys, collections
root = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
dur = collections.defaultdict(float)
for f in sorted(glob.glob(os.path.join(root, 'set*', '*', '*counter_collection.csv'))):
    first = f.split('/set')[1].startswith('1/')
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')[:48]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if first and r['Counter_Name'] == 'SQ_WAVE_CYCLES':
            calls[k] += 1
            if 'Start_Timestamp' in r and r.get('End_Timestamp'):
                dur[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
names = sorted({c for v in agg.values() for c in v})
keys = sorted(agg, key=lambda k: -agg[k].get('SQ_BUSY_CU_CYCLES', 0))
with open(os.path.join(root, 'summary.csv'), 'w') as o:
    w = csv.writer(o); w.writerow(['kernel', 'launches_per_step'] + names)
    for k in keys:
        w.writerow([k, calls[k] / steps] + ['%.6g' % (agg[k].get(c, 0.0) / steps) for c in names])
print('| kernel | launches/step | ' + ' | '.join(names) + ' |')
print('|' + '---|' * (len(names) + 2))
for k in keys[:16]:
    print('| %s | %.0f | ' % (k, calls[k] / steps) + ' | '.join('%.4g' % (agg[k].get(c, 0.0) / steps) for c in names) + ' |')
