#!/bin/bash
# The 8-rank rehearsal launch N times in a row with wall time and exit code per launch (looking for the stall of DESIGN 9 "open"):
#   bash tools/rehearsal_loop.sh [N=4]  -> gpurun_out/rehearsal_loop.txt (+ the stderr of a launch that failed or took > 300 s)
N=${1:-4}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd $ROOT
: > $OUT/rehearsal_loop.txt
for i in $(seq 1 $N); do
  t0=$(date +%s.%N)
  python3 bench.py --gpus 8 --rehearsal --steps 1 --warmup 1 --batch 16 --store-images 256 --no-cpu-baseline --no-parity --no-roofline --launch-timeout 400 > $OUT/rehearsal_loop_$i.json 2> $OUT/rehearsal_loop_$i.err
  rc=$?
  t1=$(date +%s.%N)
  dt=$(python3 -c "print(round($t1-$t0,1))")
  ok=$(python3 -c "import json,sys; d=json.load(open('$OUT/rehearsal_loop_$i.json')); print(d.get('weights_in_sync'), d.get('ranks_seen'))" 2>/dev/null)
  echo "launch $i: rc $rc  ${dt} s  weights_in_sync/ranks $ok" | tee -a $OUT/rehearsal_loop.txt
  if [ "$rc" = "0" ] && python3 -c "import sys; sys.exit(0 if $dt < 300 else 1)"; then rm -f $OUT/rehearsal_loop_$i.err; fi
done
