#!/bin/bash
# decode kernels: parity tests, throughput, kernel trace of the decode loop
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out; mkdir -p $OUT
TAG=${1:-r04_b}
cd $ROOT
timeout 1200 python3 -m pytest tests/test_gpu_decode.py tests/test_gpu_model.py -m gpu -x -q > $OUT/${TAG}_decode_tests.log 2>&1; echo "tests rc=$?"
tail -15 $OUT/${TAG}_decode_tests.log
python3 tools/decode_bench.py > $OUT/${TAG}_decode_bench.txt 2>&1; cat $OUT/${TAG}_decode_bench.txt | grep -v amdgpu.ids
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/${TAG}_dkt -o r -- python3 $ROOT/tools/decode_bench.py > $OUT/${TAG}_dkt.log 2>&1
DB=$(find $OUT/${TAG}_dkt -name "*.db" | head -1)
python3 $ROOT/tools/rocpd_stats.py $DB --steps 1 > $OUT/${TAG}_decode_kernel_stats.txt
rm -rf $OUT/${TAG}_dkt
head -30 $OUT/${TAG}_decode_kernel_stats.txt
