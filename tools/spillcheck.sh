#!/bin/bash
# usage: spillcheck.sh <tu> [flags]: device-only asm; count scratch ops inside the MFMA region of the fp16 kernel
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --offload-device-only "$@" -S /root/repo/esrganplus_amd/csrc/$f.hip -o /tmp/$f.s 2>/dev/null
awk '/^_ZN12_GLOBAL__N_116rdb_chain_kernelIDF16_/{p=1} /\.end_amdhsa_kernel|^\.Lfunc_end/{if(p)p=0} p' /tmp/$f.s > /tmp/$f.k.s
python3 - <<PY
L = open('/tmp/$f.k.s').read().split('\n')
mf = [i for i, l in enumerate(L) if 'v_mfma' in l]
sc = [(i, L[i]) for i, l in enumerate(L) if 'scratch_' in l and mf[0] < i < mf[-1]]
print('$f', 'scratch inside: stores', sum('store' in t for _, t in sc), 'loads', sum('load' in t for _, t in sc))
import re
m = [l for l in open('/tmp/$f.s') if 'vgpr_count' in l or 'scratch_en' in l or '.private_segment_fixed_size' in l or 'vgpr_spill' in l]
print(''.join(m[:8]))
PY
