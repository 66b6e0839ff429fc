for rep in 1 2; do
for cfg in "last 80" "mid 80" "first 80" "mid 112" "first 112" "last 112"; do
  set -- $cfg
  ESR_TRAIN_DSTEP=$1 ESR_BWD_FOLLOW_WGS=$2 python bench.py --mode train --steps 40 --warmup 6 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dstep=$1 wgs=$2', d['ms_per_step'])"
done; done
