#!/bin/bash
# Run on the GPU box: FETCH_SIZE / WRITE_SIZE / L2 hit-miss of rdb_wgrad_kernel on the probe (tools/rdb_wgrad_probe.py,
# 69 blocks at 16 x 128^2) for the environment given on the command line.  Usage: [ENV=..] tools/pmc_wgrad.sh <tag>
set -u
TAG=${1:-wg}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/tools/rdb_wgrad_probe.py"
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$N -o pmc -- $CMD > $OUT/pmc_$N.log 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $OUT | grep -A8 "^rdb_wgrad_kernel"
grep "rdb_wgrad:" $OUT/pmc_FETCH_SIZE.log
find $OUT -name "*.csv" -size +2M -delete
