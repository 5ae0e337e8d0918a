#!/bin/bash
# same-box A/B of decode-kernel build variants + per-kernel stats and a one-step launch timeline of two of them
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out; mkdir -p $OUT
TAG=${1:-r04_d}; shift
cd $ROOT
LIBS="$@"
bash tools/ab_decode.sh 2 $LIBS 2>&1 | grep -v amdgpu.ids | tee $OUT/${TAG}_ab_decode.txt
cd /tmp && export TMPDIR=/tmp
for L in $LIBS; do
  v=$(basename $L .so)
  VLT5_LIB=$ROOT/$L rocprofv3 --kernel-trace -d $OUT/${TAG}_dkt_$v -o r -- python3 $ROOT/tools/decode_bench.py --fast-only > $OUT/${TAG}_dkt_$v.log 2>&1
  DB=$(find $OUT/${TAG}_dkt_$v -name "*.db" | head -1)
  python3 $ROOT/tools/rocpd_stats.py $DB --steps 1 | grep -v "at::native" | head -24 > $OUT/${TAG}_decode_kernel_stats_$v.txt
  python3 $ROOT/tools/rocpd_window.py $DB dec_io_kernel -4 > $OUT/${TAG}_decode_step_timeline_$v.txt
  rm -rf $OUT/${TAG}_dkt_$v
  echo "== $v"; cat $OUT/${TAG}_decode_kernel_stats_$v.txt | head -16
done
head -30 $OUT/${TAG}_decode_step_timeline_libvlt5_hip.txt
