#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel (sum / per-dispatch mean)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    """conv_kernel<T, KS, S, UPS, WR, WC, NCG, NCW, WLDS, HAS1X1, BWD> from its mangled name."""
    m = re.search(r'conv_kernelI(DF16_|f)((?:L[ib]\d+E)+)', name)
    if m:
        a = re.findall(r'L[ib](\d+)E', m.group(2))
        keys = ['k', 's', 'ups', 'wr', 'wc', 'ncg', 'ncw', 'wlds', '1x1', 'bwd']
        return 'conv<%s,%s>' % ('f16' if m.group(1) != 'f' else 'f32', ','.join('%s%s' % kv for kv in zip(keys, a)))
    m = re.search(r'rdb_chain_kernelI(DF16_|f)(?:Li(\d)E)?', name)
    if m:
        return 'rdb_chain<%s,%s>' % ('f16' if m.group(1) != 'f' else 'f32', {'0': 'forward', '1': 'train-forward', '2': 'backward'}.get(m.group(2) or '0'))
    m = re.search(r'(rdb_wgrad_reduce_kernel|rdb_wgrad_follow_kernel|rdb_wgrad_kernel|wgrad_reduce_kernel)', name)
    if m:
        return m.group(1)
    m = re.search(r'(wgrad16_kernel<[^>]*>|wgrad_kernel\w*|pack_batch_kernel|unpermute_kernel|bn_\w+|pool_kernel\w*|linear_\w+)', name)
    return m.group(1)[:70] if m else name[:70]


def main(root):
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in sorted(glob.glob(os.path.join(root, 'pmc_*', '*counter_collection.csv'))):
        for r in csv.DictReader(open(f)):
            a = agg[short(r['Kernel_Name'])][r['Counter_Name']]
            a[0] += float(r['Counter_Value'])
            a[1] += 1
    for k in sorted(agg, key=lambda k: -agg[k].get('SQ_BUSY_CYCLES', [0, 0])[0]):
        if not (k.startswith('conv<') or k.startswith('rdb_chain<') or k.startswith('rdb_wgrad') or k.startswith('wgrad')):
            continue
        c = agg[k]
        n = max(v[1] for v in c.values())
        print(k, 'dispatches', n)
        for name in sorted(c):
            print('    %-28s mean/dispatch %16.1f' % (name, c[name][0] / c[name][1]))
        if 'FETCH_SIZE' in c:
            # gfx950: FETCH_SIZE (KB) under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md §HBM)
            fk = c['FETCH_SIZE'][0] / c['FETCH_SIZE'][1]
            wk = c['WRITE_SIZE'][0] / c['WRITE_SIZE'][1] if 'WRITE_SIZE' in c else float('nan')
            print('    => HBM-side read ~ %.1f MB (2x-corrected FETCH_SIZE), write ~ %.1f MB per dispatch' % (2 * fk / 1024, wk / 1024))
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and 'GRBM_GUI_ACTIVE' in c:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA busy cycles over the 1024 SIMDs
            print('    => MfmaUtil = MFMA busy / (GUI_ACTIVE/8 * 1024 SIMDs) = %.3f' % (c['SQ_VALU_MFMA_BUSY_CYCLES'][0] / (c['GRBM_GUI_ACTIVE'][0] / 8 * 1024)))
        if 'TCC_HIT_sum' in c:
            h, m = c['TCC_HIT_sum'][0], c['TCC_MISS_sum'][0]
            print('    => L2 hit rate %.3f' % (h / max(h + m, 1)))
        if 'SQ_LDS_BANK_CONFLICT' in c:
            print('    => LDS bank-conflict cycles / LDS active cycles = %.3f' % (c['SQ_LDS_BANK_CONFLICT'][0] / max(c['SQ_LDS_IDX_ACTIVE'][0], 1)))


if __name__ == '__main__':
    main(sys.argv[1])
