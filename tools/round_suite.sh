#!/bin/bash
# full GPU suite with the parity log + pin table recorded, smoke, then the evidence pass of the round
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out; mkdir -p $OUT
TAG=${1:-r06_z}
cd $ROOT
rm -f $OUT/${TAG}_parity.txt $OUT/${TAG}_parity_pins.json
( time VQACL_PARITY_LOG=$OUT/${TAG}_parity.txt VQACL_PARITY_PINS_OUT=$OUT/${TAG}_parity_pins.json timeout 2400 python3 -m pytest tests -m gpu -q --durations=10 ) 2>&1 | tail -40 | tee $OUT/${TAG}_pytest_tail.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/profile_round.sh $TAG 2>&1 | tail -40
