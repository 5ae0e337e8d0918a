#!/bin/bash
# The multi-process GPU tests N times in a row (looking for the one stalled suite run of round 5): a launch that runs into its
# bound now leaves the Python stacks of every rank under gpurun_out/stall_dumps/ and FAILS (tests/test_gpu_bench_line.py::run_bounded).
#   bash tools/stall_hunt.sh [N=4]  -> gpurun_out/stall_hunt.txt
N=${1:-4}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd $ROOT
: > $OUT/stall_hunt.txt
for i in $(seq 1 $N); do
  t0=$(date +%s)
  timeout 1200 python3 -m pytest tests/test_gpu_dp2.py tests/test_gpu_bench_line.py -m gpu -q -x --durations=3 > $OUT/stall_hunt_$i.log 2>&1
  rc=$?
  t1=$(date +%s)
  echo "round $i: rc $rc  $((t1-t0)) s  $(tail -1 $OUT/stall_hunt_$i.log)" | tee -a $OUT/stall_hunt.txt
  if [ "$rc" = "0" ]; then rm -f $OUT/stall_hunt_$i.log; fi
done
ls $OUT/stall_dumps 2>/dev/null | tee -a $OUT/stall_hunt.txt
