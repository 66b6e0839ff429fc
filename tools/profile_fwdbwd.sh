#!/bin/bash
# Run on the GPU box: rocprofv3 kernel-trace stats (+ optional PMC passes) of the fwd+bwd probe
# (bench.py's `fwd_bwd` object: batch 16 of 128x128 LR, fp16 train-mode forward + backward).
# Usage: tools/profile_fwdbwd.sh <tag> [pmc]     (writes gpurun_out/prof_<tag>/...)
set -u
TAG=${1:-r03_fwdbwd}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/tools/fwd_bwd_probe.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
tail -3 $OUT/trace.log
if [ "${2:-}" = "pmc" ]; then
  for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum"; do
    N=$(echo $C | tr ' ' '_' | cut -c1-40)
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$N -o pmc -- $CMD > $OUT/pmc_$N.log 2>&1
  done
fi
find $OUT -name "*stats*.csv" | head
