#!/bin/bash
# Run on the GPU box: kernel trace of the train bench, its steady-state statistics (tools/steady_stats.py) and the
# timeline of two steps (tools/trace_step.py).  Usage: tools/trace_train.sh <tag> [mode=train] [steps=20] [adams=2]
set -u
TAG=${1:-r05}; MODE=${2:-train}; STEPS=${3:-20}; ADAMS=${4:-2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_${TAG}_${MODE}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o tr -- python $GRAFT_REPO_ROOT/bench.py --mode $MODE --steps $STEPS --warmup 5 > $OUT/bench.log 2>&1
cd $GRAFT_REPO_ROOT
F=$(find $OUT -name '*kernel_trace.csv' | head -1)
echo "trace: $F"
python tools/steady_stats.py $F $OUT/steady_kernel_stats.csv 10 --adams $ADAMS
python tools/trace_step.py $F 0 500 > $OUT/timeline.txt
python tools/trace_step.py $F | head -45
grep '"metric"' $OUT/bench.log | cut -c1-300
rm -f $F   # (tens of MB; the summaries stay)
