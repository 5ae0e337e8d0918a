#!/bin/bash
# the data-parallel files of the evidence pass again (after the communication stream went to normal priority): force-dist line, its kernel trace as a rank, the overhead table
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out; TAG=${1:-r05_z}
cd $ROOT
for i in 1 2 3; do
  p=$(python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-side-values --no-roofline 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.readlines()[-1])['ms_per_step'])")
  python3 bench.py --gpus 1 --force-dist --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-side-values > $OUT/${TAG}_bench_line_force_dist.json 2> $OUT/${TAG}_force_dist.err
  f=$(python3 -c "import json; print(json.loads(open('$OUT/${TAG}_bench_line_force_dist.json').read().strip().splitlines()[-1])['ms_per_step'])")
  echo "round $i plain $p force-dist $f"
done | tee $OUT/${TAG}_plain_vs_force_dist.txt
cd /tmp && export TMPDIR=/tmp
F="--steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-parity --no-side-values"
rocprofv3 --kernel-trace -d $OUT/${TAG}_kp -o r -- python3 $ROOT/bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-roofline --no-parity --no-side-values > $OUT/${TAG}_kp.log 2>&1
python3 $ROOT/tools/rocpd_stats.py $(find $OUT/${TAG}_kp -name "*.db" | head -1) > $OUT/${TAG}_kernel_stats_plain_same_box.txt
WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 rocprofv3 --kernel-trace -d $OUT/${TAG}_dd -o r -- python3 $ROOT/bench.py --gpus 1 --force-dist $F > $OUT/${TAG}_dd.log 2>&1
python3 $ROOT/tools/rocpd_stats.py $(find $OUT/${TAG}_dd -name "*.db" | head -1) > $OUT/${TAG}_kernel_stats_force_dist.txt
rm -rf $OUT/${TAG}_kp $OUT/${TAG}_dd
cd $ROOT && python3 tools/dp_overhead_table.py $OUT/${TAG}_kernel_stats_plain_same_box.txt $OUT/${TAG}_kernel_stats_force_dist.txt > $OUT/${TAG}_dp_overhead.txt
grep "^# kernel time" $OUT/${TAG}_dp_overhead.txt
