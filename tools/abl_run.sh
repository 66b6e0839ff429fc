#!/bin/bash
# tools/abl_run.sh <out> <chain_trace args...> -- runs tools/chain_trace.py with the base library and every esrganplus_amd/lib_abl*.so
OUT=$1; shift
{ echo "=== base"; timeout 120 python tools/chain_trace.py "$@"
  for f in esrganplus_amd/lib_abl*.so; do v=$(basename $f .so); v=${v#lib_}; echo "=== $v"; ESR_LIB_PATH=$f timeout 120 python tools/chain_trace.py "$@"; done; } > $OUT 2>&1
