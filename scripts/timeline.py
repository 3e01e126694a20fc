"""Timeline of one steady-state training step from a rocprofv3 --kernel-trace CSV: every kernel longer than --min us (and all
persistent-cluster launches) with its start offset, duration and stream, in start order; step boundaries = the hipMemset/zero of
the gradient buffer is not a kernel, so steps are cut at embed_gather_k.   python scripts/timeline.py <kernel_trace.csv> [--step K]"""
import argparse, csv, re, sys
ap = argparse.ArgumentParser()
ap.add_argument('csv'); ap.add_argument('--step', type=int, default=8); ap.add_argument('--min', type=float, default=25.0)
a = ap.parse_args()
rows = list(csv.DictReader(open(a.csv)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('embed_gather')]
i0, i1 = starts[a.step], starts[a.step + 1]
t0 = int(rows[i0]['Start_Timestamp'])
def short(n):
    n = re.sub(r'^void ', '', n); n = re.sub(r'\(.*', '', n)
    return n[:44]
print('step %d: %.3f ms between step starts, %d kernels' % (a.step, (int(rows[i1]['Start_Timestamp']) - t0) / 1e6, i1 - i0))
busy = 0
for r in rows[i0:i1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    d = (e - s) / 1e3
    if d >= a.min or 'cluster' in r['Kernel_Name'] or 'gru128' in r['Kernel_Name']:
        print('%8.1f us  +%7.1f  s%-2s q%-2s %-44s grid %s' % ((s - t0) / 1e3, d, r['Stream_Id'], r['Queue_Id'], short(r['Kernel_Name']), r['Grid_Size_X']))
