#!/bin/bash
# Same-box A/B of builds of the HIP library on the greedy-decoding loop: bash tools/ab_decode.sh ROUNDS lib1.so lib2.so ... -> ms per token-step
ROUNDS=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for r in $(seq 1 $ROUNDS); do
  for L in "$@"; do
    out=$(VLT5_LIB=$(realpath $L) python3 $ROOT/tools/decode_bench.py --fast-only 2>&1 | grep "per token-step" | head -1)
    echo "round $r  $(basename $L)  $out"
  done
done
