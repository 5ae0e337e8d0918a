#!/bin/bash
# Same-box A/B of two source trees (each with its own built library): alternating bench runs.  usage: bash tools/ab_trees.sh rounds treeA treeB [bench flags]
N=$1; A=$2; B=$3; shift 3
for i in $(seq 1 "$N"); do
  for T in "$A" "$B"; do
    ms=$(cd "$T" && python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-parity --no-side-values "$@" 2>/dev/null | python3 -c 'import json,sys; print(json.loads(sys.stdin.readline())["ms_per_step"])')
    echo "round $i  $T  $ms ms/step"
  done
done
