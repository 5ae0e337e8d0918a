#!/bin/bash
# full GPU suite with the parity log + pin table recorded, then plain vs --force-dist bench lines (same box)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out; mkdir -p $OUT
TAG=${1:-r04_j}
cd $ROOT
rm -f $OUT/${TAG}_parity.txt $OUT/${TAG}_parity_pins.json
VQACL_PARITY_LOG=$OUT/${TAG}_parity.txt VQACL_PARITY_PINS_OUT=$OUT/${TAG}_parity_pins.json timeout 1500 python3 -m pytest tests -m gpu -q 2>&1 | tail -25
B="--steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-side-values --no-roofline"
for i in 1 2; do
  python3 bench.py $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('plain', d['ms_per_step'])"
  python3 bench.py --gpus 1 --force-dist $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('force-dist', d['ms_per_step'], d['grad_exchange']['algo'])"
done
