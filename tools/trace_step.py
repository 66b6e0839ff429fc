"""Summarise one train step of a rocprofv3 --kernel-trace csv (steps are delimited by the Adam launches):
busy time, idle gaps, per-kernel totals, and optionally the timeline of a range of launches.
Usage: python tools/trace_step.py <kernel_trace.csv> [lo hi]"""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
idx = [i for i, n in enumerate(names) if 'adam' in n]
a, b = idx[-5] + 1, idx[-3] + 1
st = rows[a:b]
t0 = int(st[0]['Start_Timestamp'])
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'_ZN12_GLOBAL__N_1\d+', '', n)
    n = re.sub(r'void at::native::', 'at::', n)
    return n[:64]
ce = int(st[0]['End_Timestamp']); idle = 0; gaps = []
for i, r in enumerate(st[1:], 1):
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if s > ce:
        idle += s - ce
        if s - ce > 30000: gaps.append((i, (s - ce) / 1e3, (s - t0) / 1e6))
    ce = max(ce, e)
wall = (ce - t0) / 1e6
print('step: %d launches, wall %.2f ms, busy %.2f ms, idle %.2f ms' % (len(st), wall, wall - idle / 1e6, idle / 1e6))
for i, g, t in gaps: print('  gap %.0f us before #%d at %.2f ms (%s)' % (g, i, t, short(st[i]['Kernel_Name'])[:40]))
if len(sys.argv) > 3:
    pe = None
    for i in range(int(sys.argv[2]), min(int(sys.argv[3]), len(st))):
        r = st[i]; s = int(r['Start_Timestamp']) - t0; e = int(r['End_Timestamp']) - t0
        g = '%sx%sx%s' % (int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), r['Grid_Size_Y'], r['Grid_Size_Z'])
        print('%4d %8.3f dur %6.1f gap %6.1f s%s %-10s %s' % (i, s / 1e6, (e - s) / 1e3, (s - pe) / 1e3 if pe is not None else 0, r['Stream_Id'], g, short(r['Kernel_Name'])))
        pe = e
else:
    agg = collections.defaultdict(lambda: [0, 0])
    for r in st:
        k = short(r['Kernel_Name']); agg[k][0] += 1; agg[k][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
        print('%5d %9.1f us %7.1f avg  %s' % (v[0], v[1] / 1e3, v[1] / 1e3 / v[0], k))
