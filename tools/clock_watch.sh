#!/bin/bash
# Samples sclk / socket power with rocm-smi while the forward bench, then the train-mode fwd+bwd probe run:
# evidence for the sustained clock under the MFMA load.  Usage on the GPU box: bash tools/clock_watch.sh
watch_run() {
  "$@" > /tmp/cw_out.txt 2>/dev/null &
  BP=$!
  sleep 8
  while kill -0 $BP 2>/dev/null; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Package Power" | sed -e 's/.*(\([0-9]*Mhz\)).*/sclk \1/' -e 's/.*(W): /W /' | tr '\n' ' '; echo
    sleep 0.5
  done
}
echo "== forward: python bench.py --steps 2500 --no-fwd-bwd --no-cpu-baseline --no-train"
watch_run python bench.py --steps 2500 --warmup 10 --no-fwd-bwd --no-cpu-baseline --no-train
tail -1 /tmp/cw_out.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('forward ms/step', d['ms_per_step'], 'whole-net frac', d['whole_net_frac_of_f16_mfma_peak'])"
echo "== fwd+bwd: python tools/fwd_bwd_probe.py 600"
watch_run python tools/fwd_bwd_probe.py 600
tail -2 /tmp/cw_out.txt
