#!/bin/bash
# kernel trace of a short bench run -> per-kernel stats + one-step timeline under gpurun_out/<tag>_*   usage: bash tools/kt.sh <tag> [bench flags]
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out; mkdir -p $OUT
rocprofv3 --kernel-trace -d $OUT/${TAG}_kt -o r -- python3 $ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-parity --no-side-values "$@" > $OUT/${TAG}_kt.log 2>&1
DB=$(find $OUT/${TAG}_kt -name "*.db" | head -1)
python3 $ROOT/tools/rocpd_stats.py $DB > $OUT/${TAG}_kernel_stats.txt
python3 $ROOT/tools/rocpd_timeline.py $DB > $OUT/${TAG}_step_timeline.txt
rm -rf $OUT/${TAG}_kt
head -40 $OUT/${TAG}_kernel_stats.txt
