#!/bin/bash
# per-kernel stats of the greedy-decoding loop (decode kernels only): bash tools/r04_decode_stats.sh TAG -> gpurun_out/TAG_decode_kernel_stats.txt
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out; mkdir -p $OUT
TAG=${1:-r04_q}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/${TAG}_dkt -o r -- python3 $ROOT/tools/decode_bench.py --fast-only > $OUT/${TAG}_dkt.log 2>&1
DB=$(find $OUT/${TAG}_dkt -name "*.db" | head -1)
python3 $ROOT/tools/rocpd_stats.py $DB --steps 1 > $OUT/${TAG}_decode_kernel_stats.txt
rm -rf $OUT/${TAG}_dkt
head -24 $OUT/${TAG}_decode_kernel_stats.txt | cut -c1-170
