#!/usr/bin/env python
"""Per-kernel summary of a rocprofv3 --kernel-trace CSV: calls, avg / total time, stream.
Usage: python tools/trace_summary.py <trace_kernel_trace.csv> [steps]"""
import collections
import csv
import re
import sys


def short(n):
    m = re.search(r'conv_kernelI(DF16_|f)((?:L[ib]\d+E)+)', n)
    if m:
        a = re.findall(r'L[ib](\d+)E', m.group(2))
        keys = ['k', 's', 'ups', 'wr', 'wc', 'ncg', 'ncw', 'wlds', '1x1', 'bwd']
        return 'conv<' + ','.join('%s%s' % kv for kv in zip(keys, a)) + '>'
    m = re.search(r'(rdb_wgrad\w*kernel|rdb_chain_kernel\w*|wgrad\w+|unpermute_kernel|pack_batch_kernel|to_g32_kernel|from_g32\w*|frag_gather\w*)', n)
    if m:
        return m.group(1)
    return n[:50]


def main(path, steps=1):
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(list)
    for r in rows:
        agg[(short(r['Kernel_Name']), r['Stream_Id'])].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    tot = sum(sum(v) for v in agg.values())
    print('%-64s %3s %8s %10s %10s %6s' % ('kernel', 'st', 'n/step', 'avg_us', 'ms/step', '%'))
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print('%-64s %3s %8.1f %10.1f %10.3f %6.2f' % (k[0], k[1], len(v) / steps, sum(v) / len(v) / 1e3, sum(v) / steps / 1e6, 100.0 * sum(v) / tot))
    print('total kernel ms/step %.3f' % (tot / steps / 1e6))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1)
