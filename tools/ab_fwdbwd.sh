#!/bin/bash
# A/B two builds of libesrgan_hip.so on the fwd+bwd probe inside ONE gpurun call (box-to-box variance is +-3 %):
#   tools/ab_fwdbwd.sh esrganplus_amd/lib_base.so esrganplus_amd/libesrgan_hip.so
for rep in 1 2; do
  for lib in "$@"; do
    ESR_LIB_PATH=$PWD/$lib python tools/fwd_bwd_probe.py 2>/dev/null | python -c "
import sys, ast
d = ast.literal_eval(sys.stdin.read())
k = d['kernels']
print('$rep $lib total %.2f' % d['ms_per_step'], ' '.join('%s %.3f' % (n, k[n]['ms_per_step']) for n in ('rdb_chain_train', 'rdb_chain_bwd', 'rdb_wgrad') if n in k))"
  done
done
