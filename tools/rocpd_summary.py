#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel stats table —
the same content as `rocprofv3 --stats` CSV (name, calls, total/avg/min/max ns, %)."""
import re
import sqlite3
import sys


def short(name):
    m = re.search(r'conv_kernelI(DF16_|f)Li(\d)ELi(\d)ELb(\d)ELi(\d)ELi(\d)ELi(\d)ELb(\d)', name)
    if m:
        t, ks, s, ups, wr, wc, ncg, h1 = m.groups()
        return 'conv_kernel<%s,ks%s,s%s,ups%s,%sx%sx%s,1x1=%s>' % ('f16' if t != 'f' else 'f32', ks, s, ups, wr, wc, ncg, h1)
    return name[:90]


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute('select name, start, end from kernels').fetchall()
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(short(name), [])
        a.append(e - s)
    tot = sum(sum(v) for v in agg.values())
    print('%-64s %8s %14s %12s %10s %10s %6s' % ('kernel', 'calls', 'total_ns', 'avg_ns', 'min_ns', 'max_ns', '%'))
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print('%-64s %8d %14d %12.0f %10d %10d %6.2f' % (k, len(v), sum(v), sum(v) / len(v), min(v), max(v), 100.0 * sum(v) / tot))


if __name__ == '__main__':
    main(sys.argv[1])
