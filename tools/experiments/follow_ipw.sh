for rep in 1 2; do
for cfg in "4 80" "2 80" "2 96" "1 96" "8 80"; do
  set -- $cfg
  ESR_RDB_WGRAD_IPW=$1 ESR_BWD_FOLLOW=1 ESR_BWD_FOLLOW_WGS=$2 python bench.py --mode train --steps 40 --warmup 6 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ipw=$1 wgs=$2', d['ms_per_step'])"
done; done
