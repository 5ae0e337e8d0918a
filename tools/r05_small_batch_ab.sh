#!/bin/bash
# small batches are bound by the length of the dependent launch chain (tools/step_graph_probe.py): do the existing launch-saving switches of the
# decoder (fused attention sublayers, folded norms) pay there?  same-box A/B per batch size
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
for B in 4 16 32; do
  for r in 1 2; do
    for setting in "VLT5_NOP=0" "VLT5_DEC_FUSED=1" "VLT5_FOLD_NORM_DEC=1"; do
      ms=$(env $setting python3 bench.py --batch $B --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-side-values --no-roofline 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.readlines()[-1])['ms_per_step'])")
      echo "B=$B round $r  $setting  $ms ms/step"
    done
  done
done
