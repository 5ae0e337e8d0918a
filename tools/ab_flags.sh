#!/bin/bash
# Same-box A/B of bench.py flag sets: bash tools/ab_flags.sh ROUNDS "flags a" "flags b" ...
ROUNDS=$1; shift
for r in $(seq 1 $ROUNDS); do
  for f in "$@"; do
    ms=$(python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-side-values --no-roofline $f 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.readlines()[-1])['ms_per_step'])")
    echo "round $r  [$f]  $ms ms/step"
  done
done
