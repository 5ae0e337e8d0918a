#!/bin/bash
# per-workgroup phase timelines of the step's main GEMM shapes on the current build (needs: bash tools/build_variant.sh tl -DGEMM_TIMELINE)
export VLT5_LIB=$PWD/vqacl_amd/libvlt5_tl.so
O=gpurun_out/r04_gemm_timeline.txt
: > $O
for a in "4480 3072 768 0 0 224 256" "4480 2304 768 0 0 224 256" "3072 768 4480 1 1 256 256" "768 768 4480 1 1 256 256" "4480 768 768 0 0 64 128" "4480 768 3072 0 0 64 128" "4480 768 768 0 1 128 64" "4480 768 3072 0 1 128 64" "4480 768 2304 0 1 128 64" "4480 3072 768 0 1 224 256"; do
  python tools/gemm_timeline.py $a 2>&1 | grep -v amdgpu.ids >> $O
done
cat $O
