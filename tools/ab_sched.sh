# in-run A/B of library variants: bash tools/ab_sched.sh [--train] hip h1 ...
MODE=forward; if [ "$1" = "--train" ]; then MODE=train; shift; fi
for rep in 1 2; do
for v in "$@"; do
  echo "== $v"; ESR_LIB_PATH=/root/repo/esrganplus_amd/libesrgan_$v.so python bench.py --mode $MODE --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:round(v['ms_per_step'],3) for k,v in d.get('kernels',{}).items()})"
done; done
