#!/usr/bin/env python
"""profiles/roofline_traffic.json holds PMC-measured HBM bytes per launch that bench.py quotes as `roofline.traffic`; a
number measured on another version of a kernel is stale.  Every row therefore carries `source_sha`: the sha256 (first
16 hex digits) over the source files that define that kernel.  `python tools/traffic_hashes.py` prints the current
hashes and which rows match; `--update` stamps the rows (run it right after refreshing the numbers from a
`rocprofv3 --pmc` pass over the CURRENT build: tools/pmc_summary.py)."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'esrganplus_amd', 'csrc')
CHAIN = ('rdb_chain_kernel.h', 'mfma_tile.h', 'common.h')
CONV = ('conv_mfma.hip', 'mfma_tile.h', 'common.h')
SOURCES = {'rdb_chain': CHAIN, 'rdb_chain_train': CHAIN, 'rdb_chain_bwd': CHAIN, 'rdb_wgrad': ('rdb_wgrad.hip', 'common.h'),
           # the same kernels at the train step's shape (16 x 32^2 LR crops, 4-row tiles; tools/pmc_train.sh)
           'rdb_chain_train@train': CHAIN, 'rdb_chain_bwd@train': CHAIN, 'rdb_wgrad@train': ('rdb_wgrad.hip', 'common.h'),
           'conv3x3_c32': CONV, 'conv3x3_c64': CONV, 'upconv_subpix_c64': CONV}


def source_sha(kernel):
    h = hashlib.sha256()
    for f in SOURCES[kernel]:
        h.update(open(os.path.join(CSRC, f), 'rb').read())
    return h.hexdigest()[:16]


def fresh(table, kernel):
    row = table.get(kernel)
    return bool(row) and kernel in SOURCES and row.get('source_sha') == source_sha(kernel)


if __name__ == '__main__':
    path = os.path.join(ROOT, 'profiles', 'roofline_traffic.json')
    tj = json.load(open(path))
    for k in SOURCES:
        print('%-20s %s  %s' % (k, source_sha(k), 'fresh' if fresh(tj, k) else 'STALE / unstamped'))
        if '--update' in sys.argv and k in tj:
            tj[k]['source_sha'] = source_sha(k)
    if '--update' in sys.argv:
        json.dump(tj, open(path, 'w'), indent=2)
        print('stamped', path)
