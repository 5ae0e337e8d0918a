#!/bin/bash
# Same-box A/B of two builds of the HIP library: alternating bench runs, ms per step of each.
# usage (on the GPU box): bash tools/ab_run.sh <libA.so> <libB.so> [rounds] [extra bench.py flags]
A=$(realpath "$1"); B=$(realpath "$2"); shift 2; N=${1:-3}; [ $# -gt 0 ] && shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for i in $(seq 1 "$N"); do
  for L in "$A" "$B"; do
    ms=$(VLT5_LIB=$L python3 "$ROOT/bench.py" --steps 30 --warmup 5 --no-cpu-baseline --no-roofline "$@" | python3 -c 'import json,sys; print(json.loads(sys.stdin.readline())["ms_per_step"])')
    echo "round $i  $(basename "$L")  $ms ms/step"
  done
done
