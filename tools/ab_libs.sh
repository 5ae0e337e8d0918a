#!/bin/bash
# Same-box A/B of several builds of the HIP library (tools/build_variant.sh): bash tools/ab_libs.sh ROUNDS lib1.so lib2.so ...
ROUNDS=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for r in $(seq 1 $ROUNDS); do
  for L in "$@"; do
    ms=$(VLT5_LIB=$(realpath $L) python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-side-values --no-roofline 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.readlines()[-1])['ms_per_step'])")
    echo "round $r  $(basename $L)  $ms ms/step"
  done
done
