#!/bin/bash
# round 5, first GPU call: full GPU suite (parity log + pin table), the trajectory proxy, plain vs --force-dist bench lines (same box)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out; mkdir -p $OUT
TAG=${1:-r05_a}
cd $ROOT
rm -f $OUT/${TAG}_parity.txt $OUT/${TAG}_parity_pins.json
VQACL_PARITY_LOG=$OUT/${TAG}_parity.txt VQACL_PARITY_PINS_OUT=$OUT/${TAG}_parity_pins.json timeout 2400 python3 -m pytest tests -m gpu -q -x --durations=15 2>&1 | tail -45 | tee $OUT/${TAG}_pytest_tail.txt
timeout 1500 python3 tools/trajectory.py --out gpurun_out/${TAG}_trajectory.txt 2>&1 | tail -80
B="--steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-side-values --no-roofline"
for i in 1 2; do
  python3 bench.py $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('plain', d['ms_per_step'])"
  python3 bench.py --gpus 1 --force-dist $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('force-dist', d['ms_per_step'], d['grad_exchange']['algo'])"
done
