#!/bin/bash
# A/B two builds of libesrgan_hip.so inside ONE gpurun call (box-to-box variance is +-3 %):
#   tools/ab_bench.sh esrganplus_amd/libesrgan_base.so esrganplus_amd/libesrgan_hip.so
# NB: a build whose C structs differ from the current _lib.py mirrors cannot be loaded this way.
for rep in 1 2 3; do
  for lib in "$@"; do
    ms=$(ESR_LIB_PATH=$PWD/$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fwd-bwd 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$rep $lib $ms"
  done
done
