#!/bin/bash
# After `gpurun -- bash tools/profile_all.sh <tag>`: copy what is judged from gpurun_out/ into profiles/ and re-stamp the
# traffic rows.  Usage: tools/collect_profiles.sh r04   (then run bench.py once more on the GPU for the committed line:
# the bench quotes a traffic row only while its hash matches the tree)
set -e
TAG=${1:-r04}
G=gpurun_out; P=profiles
cp $G/${TAG}_bench.json $G/${TAG}_bench_train.json $G/${TAG}_bench_gtrain.json $G/${TAG}_fwd_pmc_summary.txt $G/${TAG}_fwdbwd_pmc_summary.txt $P/
cp $G/prof_${TAG}/trace/trace_kernel_stats.csv $P/${TAG}_fwd_kernel_stats.csv
cp $G/prof_${TAG}_fwdbwd/trace/trace_kernel_stats.csv $P/${TAG}_fwdbwd_kernel_stats.csv
cp $G/prof_${TAG}_train/train/train_kernel_stats.csv $P/${TAG}_train_kernel_stats.csv
cp $G/prof_${TAG}_train/gtrain/gtrain_kernel_stats.csv $P/${TAG}_gtrain_kernel_stats.csv
python tools/update_traffic.py $TAG
