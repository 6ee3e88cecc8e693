"""Summarise the rocprofv3 kernel trace of scripts/plugin_iter.py: kernels per BO iteration of the plug-in loop.  The
first k_sweep_rankq launch marks the end of the cold iteration; every later one starts a warm iteration's grid stage.
Prints, per phase, the kernels launched and their summed durations -- the witness that a warm iteration launches no
k_sweep_trmm / k_cross_gram."""
import collections
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
short = lambda n: n.split('(')[0].replace('void ', '').replace('gpx::', '')     # noqa: E731
marks = [i for i, r in enumerate(rows) if 'k_sweep_rankq' in r['Kernel_Name']]
if not marks:
    print('no k_sweep_rankq launch found: the warm path did not run')
    sys.exit(1)
phases = [('cold iteration (fit + full sweep + refinement + append + recommender)', rows[:marks[0]])]
for a, b in zip(marks, marks[1:] + [len(rows)]):
    phases.append(('warm iteration', rows[a:b]))
for title, rs in phases:
    agg = collections.OrderedDict()
    for r in rs:
        k = short(r['Kernel_Name'])
        n, t = agg.get(k, (0, 0.0))
        agg[k] = (n + 1, t + (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6)
    total = sum(t for _, t in agg.values())
    print('%s: %d launches, %.2f ms of kernel time' % (title, len(rs), total))
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
        print('    %-34s x%-4d %9.3f ms' % (k[:34], n, t))
    heavy = [k for k in agg if 'k_sweep_trmm' in k or 'k_cross_gram' in k]
    print('    -> k_sweep_trmm / k_cross_gram launches: %s' % (', '.join('%s x%d' % (k, agg[k][0]) for k in heavy) or 'NONE'))
