#!/bin/bash
# Run on the GPU box: everything profiles/<tag>_* is made from.  Usage: tools/profile_all.sh r03
TAG=${1:-r05}
O=$GRAFT_REPO_ROOT/gpurun_out
python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
python bench.py --mode train --steps 30 --warmup 5 > $O/${TAG}_bench_train.json 2>> $O/${TAG}_bench.err
python bench.py --mode gtrain --steps 5 --warmup 2 > $O/${TAG}_bench_gtrain.json 2>> $O/${TAG}_bench.err
bash tools/profile_bench.sh ${TAG} > $O/${TAG}_profile_bench.log 2>&1
bash tools/profile_fwdbwd.sh ${TAG}_fwdbwd pmc > $O/${TAG}_profile_fwdbwd.log 2>&1
bash tools/profile_train.sh ${TAG} > $O/${TAG}_profile_train.log 2>&1
python tools/pmc_summary.py $O/prof_${TAG} > $O/${TAG}_fwd_pmc_summary.txt 2>&1
python tools/pmc_summary.py $O/prof_${TAG}_fwdbwd > $O/${TAG}_fwdbwd_pmc_summary.txt 2>&1
ls $O | head -40
