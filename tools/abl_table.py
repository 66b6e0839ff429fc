"""tabulate the output of tools/abl_run.sh"""
import re, sys
txt = open(sys.argv[1]).read()
tabs = {}
for s in txt.split('=== ')[1:]:
    name = s.split('\n')[0].strip()
    rows = {}
    for l in s.split('\n'):
        m = re.match(r'\s+(\S+)\s+([\d.]+)\s+\[', l)
        if m: rows[m.group(1)] = float(m.group(2))
        m = re.match(r'\s+total\s+([\d.]+)', l)
        if m: rows['total'] = float(m.group(1))
        m = re.search(r'net forward.* ([\d.]+) ms', l)
        if m: rows['net_ms'] = float(m.group(1))
    if 'total' in rows: tabs[name] = rows
    else: print(name, 'FAILED:', s[-300:])
names = list(next(iter(tabs.values())).keys())
print('%-10s' % '', ' '.join('%7s' % k for k in tabs))
for n in names:
    print('%-10s' % n, ' '.join('%7.2f' % tabs[k].get(n, 0) for k in tabs))
print('%-10s' % 'mfma segs', ' '.join('%7.2f' % sum(v for n, v in tabs[k].items() if n.startswith(('crit', 'bulk', '1x1'))) for k in tabs))
