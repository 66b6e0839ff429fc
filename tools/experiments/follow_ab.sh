# same-box A/B of the follower form of the backward chain's weight gradients (bench.py --mode train, pipelined ms per step)
for rep in 1 2; do
for cfg in $CFGS; do
  if [ $cfg = off ]; then export ESR_BWD_FOLLOW=0; unset ESR_BWD_FOLLOW_WGS; else export ESR_BWD_FOLLOW=1 ESR_BWD_FOLLOW_WGS=$cfg; fi
  python bench.py --mode train --steps 40 --warmup 6 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('follow=$cfg', d['ms_per_step'], d.get('ms_per_step_sync_log'))"
done; done
