#!/usr/bin/env python
"""Steady-state kernel statistics of a training bench from a `rocprofv3 --kernel-trace` csv: rocprofv3's own --stats
summary covers the WHOLE process (plan building, first-touch packs, warm-up), which is not what a step costs.  This
keeps only the launches of the LAST `nsteps` optimisation steps — a step ends with its second `adam_kernel` launch
(G's and D's optimizers; one per step for the generator-only bench: --adams 1) — and writes the same columns as
rocprofv3's kernel_stats.csv plus launches and summed kernel time PER STEP.
Usage: python tools/steady_stats.py <kernel_trace.csv> <out.csv> [nsteps=10] [--adams 2]"""
import collections
import csv
import sys

args = [a for a in sys.argv[1:] if not a.startswith('--')]
adams = int(sys.argv[sys.argv.index('--adams') + 1]) if '--adams' in sys.argv else 2
src, dst = args[0], args[1]
nsteps = int(args[2]) if len(args) > 2 else 10
rows = list(csv.DictReader(open(src)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ends = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
ends = ends[adams - 1::adams]                      # index of the launch that closes each step
assert len(ends) > nsteps + 1, 'trace holds %d steps, asked for the last %d' % (len(ends), nsteps)
# steps are pipelined (the next step's first launches overlap the previous step's tail on the other stream): cut at the
# closing Adam launches, which gives every launch to exactly one step
lo, hi = ends[-nsteps - 1] + 1, ends[-1] + 1
sel = rows[lo:hi]
agg = collections.OrderedDict()
for r in sel:
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    agg.setdefault(r['Kernel_Name'], []).append(d)
tot = sum(sum(v) for v in agg.values())
wall = int(sel[-1]['End_Timestamp']) - int(sel[0]['Start_Timestamp'])
with open(dst, 'w', newline='') as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs', 'CallsPerStep', 'NsPerStep'])
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([k, len(v), sum(v), round(sum(v) / len(v), 1), round(100.0 * sum(v) / tot, 2), min(v), max(v),
                    round(len(v) / nsteps, 2), round(sum(v) / nsteps, 1)])
    w.writerow(['_STEADY_STATE_TOTAL (last %d steps of the trace; wall %.3f ms per step)' % (nsteps, wall / nsteps / 1e6),
                len(sel), tot, round(tot / len(sel), 1), 100.0, 0, 0, round(len(sel) / nsteps, 2), round(tot / nsteps, 1)])
lib = sum(len(v) for k, v in agg.items() if 'at::native' not in k and '__amd_rocclr' not in k and 'Cijk' not in k)
print('%d steps: %.1f launches per step (%.1f of them the library\'s), %.3f ms of kernel time per step, %.3f ms wall per step'
      % (nsteps, len(sel) / nsteps, lib / nsteps, tot / nsteps / 1e6, wall / nsteps / 1e6))
