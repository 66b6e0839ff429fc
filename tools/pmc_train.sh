#!/bin/bash
# Run on the GPU box: HBM-side traffic (FETCH_SIZE / WRITE_SIZE, one counter per pass) of the train step's kernels —
# bench.py --mode train, BASELINE configs[2] — summarised per kernel by tools/pmc_summary.py.  Usage: tools/pmc_train.sh <tag>
set -u
TAG=${1:-r05}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_trainpmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --mode train --steps 10 --warmup 3"
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$N -o pmc -- $CMD > $OUT/pmc_$N.log 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $OUT > gpurun_out/${TAG}_train_pmc_summary.txt 2>&1
grep -A8 "^rdb_chain\|^rdb_wgrad_kernel" gpurun_out/${TAG}_train_pmc_summary.txt | head -60
find $OUT -name "*.csv" -size +4M -delete
