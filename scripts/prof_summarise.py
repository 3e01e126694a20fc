"""Per-step kernel table from a rocprofv3 --kernel-trace CSV of scripts/dev_prof.py: the first training step (which also
allocates and zero-fills every buffer once) is left out; a step ends with its adam_k launch.
usage: python scripts/prof_summarise.py <dir with *_kernel_trace.csv> [out.md]"""
import collections, csv, glob, sys

f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[0]
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))))
ends = [i for i, k in enumerate(rows) if k[2].startswith('adam_k')]
body = rows[ends[0] + 1:ends[-1] + 1]
nsteps = len(ends) - 1
acc = collections.OrderedDict()
for s, e, n in body:
    a = acc.setdefault(n, [0, 0])
    a[0] += 1; a[1] += e - s
tot = sum(v[1] for v in acc.values())
span = (body[-1][1] - body[0][0]) / nsteps / 1e6
busy, cur = 0, body[0][0]
for s, e, n in body:
    if e > cur:
        busy += e - max(s, cur); cur = e
lines = ['| kernel | calls/step | ms/step | avg us | % |', '|---|---|---|---|---|']
for n, (c, d) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    lines.append('| %s | %.1f | %.3f | %.2f | %.2f |' % (n[:110], c / nsteps, d / nsteps / 1e6, d / c / 1e3, 100.0 * d / tot))
head = ('%d steps (first step excluded); sum of kernel durations %.2f ms/step (kernels overlap), wall %.2f ms/step under the profiler, '
        'some kernel running %.1f %% of the wall time' % (nsteps, tot / nsteps / 1e6, span, 100.0 * busy / (span * nsteps * 1e6)))
out = head + '\n\n' + '\n'.join(lines) + '\n'
if len(sys.argv) > 2:
    open(sys.argv[2], 'w').write(out)
print(out)
