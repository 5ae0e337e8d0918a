#!/bin/bash
# Same-box A/B of several builds of the HIP library: alternating bench runs, ms per step of each.  usage: bash tools/ab3.sh rounds lib1.so lib2.so ...
N=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for i in $(seq 1 "$N"); do
  for L in "$@"; do
    ms=$(VLT5_LIB=$(realpath $L) python3 "$ROOT/bench.py" --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-parity --no-side-values | python3 -c 'import json,sys; print(json.loads(sys.stdin.readline())["ms_per_step"])')
    echo "round $i  $(basename "$L")  $ms ms/step"
  done
done
