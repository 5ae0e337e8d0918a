#!/bin/bash
cd "$(dirname "$0")/.."
export VLT5_LIB=$PWD/vqacl_amd/libvlt5_tl.so
O=gpurun_out/timeline.txt; : > $O
for args in "4480 3072 768 0 0 128 128" "4480 3072 768 0 0 256 256" "4480 768 768 0 0 64 128 f32" "4480 768 3072 0 1 64 128" "400 768 768 0 0 64 64 f32" "4640 18432 768 0 0 256 256"; do
  python tools/gemm_timeline.py $args >> $O 2>&1
done
cat $O
