#!/bin/bash
# the two GEMM candidates of the round-3 review, measured before any rewrite: (a) MFMA shape in the 8-wave k-step (tools/mfma_shape_probe.hip),
# (b) the ceiling of a stream-K / persistent tile walk for the 420-tile N = 768 launches: the same launches with 10 instead of 12 k-steps per
# workgroup (= the k-steps a perfectly balanced walk over 512 slots would leave each workgroup, with no fix-up cost at all), and with 512 tiles
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out; mkdir -p $OUT
TAG=${1:-r04_m}
cd $ROOT
{
echo "# (a) MFMA shape, 8-wave 256x256 k-step (no global traffic)"
build/mfma_shape_probe
echo
echo "# (b) stream-K ceiling: N = 768 launches of the train step, graph-replayed (tools/gemm_probe2.py: M N K a_kmajor b_kmajor tile_m tile_n)"
echo "#     420 tiles x 12 k-steps (as launched) | 420 tiles x 10 k-steps (ideal balanced walk: 420 x 12 / 512 = 9.84) | 512 tiles x 12 k-steps (all slots busy)"
python3 tools/gemm_probe2.py "4480 768 768 0 1 128 64" "4480 768 640 0 1 128 64" "5461 768 768 0 1 128 64" \
                             "4480 768 768 0 0 64 128" "4480 768 640 0 0 64 128" "5461 768 768 0 0 64 128" \
                             "4480 768 3072 0 1 128 64" "4480 768 2560 0 1 128 64" "4480 768 3072 0 0 64 128" "4480 768 2560 0 0 64 128" \
                             "4480 768 2304 0 1 128 64" "4480 768 1920 0 1 128 64" 2>&1 | grep -v amdgpu
} | tee $OUT/${TAG}_gemm_candidates.txt
