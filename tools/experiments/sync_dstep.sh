# logging-form train step (tools/train_marks.py sync): D-step placement x follower size, same box
for rep in 1 2; do
for cfg in "last 80" "first 112" "first 128" "mid 128" "first 96" "mid 112"; do
  set -- $cfg
  ESR_TRAIN_DSTEP_SYNC=$1 ESR_TRAIN_SYNC_FOLLOW_WGS=$2 python tools/train_marks.py 40 sync 2>/dev/null | grep -E "logging-form" | cut -c1-32 | tr '\n' ' '; echo " dstep=$1 wgs=$2"
done; done
