#!/bin/bash
# Same-box A/B of an environment switch of the engine: alternating bench runs, ms per step of each setting.
# usage (on the GPU box): bash tools/ab_env.sh VAR valueA valueB [rounds] [extra bench.py flags]
VAR=$1; A=$2; B=$3; shift 3; N=${1:-3}; [ $# -gt 0 ] && shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for i in $(seq 1 "$N"); do
  for V in "$A" "$B"; do
    ms=$(env "$VAR=$V" python3 "$ROOT/bench.py" --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-parity --no-side-values "$@" | python3 -c 'import json,sys; print(json.loads(sys.stdin.readline())["ms_per_step"])')
    echo "round $i  $VAR=$V  $ms ms/step"
  done
done
