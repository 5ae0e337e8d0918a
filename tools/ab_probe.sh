#!/bin/bash
# graph-replayed timings of a few GEMM launches under several builds of the library: bash tools/ab_probe.sh lib1.so lib2.so ...
SHAPES=("4096 4096 4480 0 0 256 256" "4096 4096 4480 0 1 256 256" "4096 4096 4480 1 1 256 256" "4480 3072 768 0 0 224 256" "4480 3072 768 0 1 224 256" "768 3072 4480 1 1 256 256 6" "3072 768 4480 1 1 256 256 6" "18432 768 4640 1 1 256 256" "4640 18432 768 0 0 224 256")
for L in "$@"; do
  echo "== $L"
  VLT5_LIB=$PWD/$L python tools/gemm_probe2.py "${SHAPES[@]}" 2>&1 | grep -v amdgpu
done
