#!/bin/bash
# Same-box A/B of environment switches: bash tools/ab_env_loop.sh ROUNDS "VAR=a" "VAR=b VAR2=c" ...  -> ms/step of bench.py per setting, interleaved
ROUNDS=$1; shift
for r in $(seq 1 $ROUNDS); do
  for setting in "$@"; do
    ms=$(env $setting python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-side-values --no-roofline 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.readlines()[-1])['ms_per_step'])")
    echo "round $r  $setting  $ms ms/step"
  done
done
