#!/bin/bash
# Instruction-fetch counters of the forward kernels (run on the GPU box): tools/pmc_icache.sh
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_icache
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_IFETCH[A-Z_]*\|SQ_INST_CACHE[A-Z_]*\|SQ_INSTS_[A-Z_]*" | sort -u | tr '\n' ' ' > $OUT/avail.txt
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-fwd-bwd"
for C in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$N -o pmc -- $CMD > $OUT/pmc_$N.log 2>&1
  f=$(find $OUT/pmc_$N -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'][:40]
    acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
for k in acc:
    if 'rdb_chain' in k or 'conv' in k:
        print(k, {c: round(v / max(1, n[(k, c)])) for c, v in acc[k].items()})
PY
done
cat $OUT/avail.txt
