#!/bin/bash
# per-workgroup timelines of chosen GEMM launches with the -DGEMM_TIMELINE library (tools/build_variant.sh tl -DGEMM_TIMELINE)
cd "$(dirname "$0")/.."
export VLT5_LIB=$PWD/vqacl_amd/libvlt5_tl.so
O=gpurun_out/timeline.txt; : > $O
while read -r args; do
  [ -z "$args" ] && continue
  python tools/gemm_timeline.py $args 2>&1 | grep -v amdgpu.ids >> $O
done <<< "${1:-4096 4096 4480 0 0 256 256 f32
4096 4096 4480 0 1 256 256 f32
4096 4096 4480 1 1 256 256 f32
4096 4096 4480 1 1 128 128 f32
4096 4096 4480 0 0 128 128 f32}"
cat $O
