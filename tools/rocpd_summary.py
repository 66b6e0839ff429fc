#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel stats table —
the same content as `rocprofv3 --stats` CSV (name, calls, total/avg/min/max ns, %)."""
import re
import sqlite3
import sys


def short(name):
    """conv_kernel<T, KS, S, UPS, WR, WC, NCG, NCW, WLDS, HAS1X1, BWD> from its mangled name."""
    m = re.search(r'conv_kernelI(DF16_|f)((?:L[ib]\d+E)+)', name)
    if m:
        a = re.findall(r'L[ib](\d+)E', m.group(2))
        keys = ['k', 's', 'ups', 'wr', 'wc', 'ncg', 'ncw', 'wlds', '1x1', 'bwd']
        return 'conv<%s,%s>' % ('f16' if m.group(1) != 'f' else 'f32', ','.join('%s%s' % kv for kv in zip(keys, a)))
    m = re.search(r'rdb_chain_kernelI(DF16_|f)E', name)
    if m:
        return 'rdb_chain<%s>' % ('f16' if m.group(1) != 'f' else 'f32')
    m = re.search(r'(wgrad16_kernel<[^>]*>|wgrad_kernel\w*|pack_batch_kernel|unpermute_kernel|bn_\w+|pool_kernel\w*|linear_\w+)', name)
    return m.group(1)[:70] if m else name[:70]


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
