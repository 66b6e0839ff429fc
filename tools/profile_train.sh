#!/bin/bash
# Run on the GPU box: rocprofv3 kernel-trace stats of the train / gtrain benches.  Usage: tools/profile_train.sh <tag>
set -u
TAG=${1:-r03}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_train
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train -o train -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 20 --warmup 3 > $OUT/train.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/gtrain -o gtrain -- python $GRAFT_REPO_ROOT/bench.py --mode gtrain --steps 4 --warmup 1 > $OUT/gtrain.log 2>&1
grep '"metric"' $OUT/train.log $OUT/gtrain.log | cut -c1-260
