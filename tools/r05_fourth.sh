#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out; mkdir -p $OUT
TAG=${1:-r05_d}
cd $ROOT
VQACL_PARITY_LOG=$OUT/${TAG}_parity.txt timeout 1500 python3 -m pytest -m gpu -q --tb=short \
   "tests/test_gpu_model.py::test_data_parallel_overlap_path_on_one_gpu_and_rank_equivalence" tests/test_gpu_trajectory.py \
   "tests/test_gpu_model.py::test_encoder_kernel_choice_switches_with_the_batch_size_and_both_sides_match_the_oracle" \
   "tests/test_gpu_model.py::test_alternative_engine_paths_still_match_the_oracle" "tests/test_gpu_model.py::test_base_model_forward_backward_vs_oracle" \
   tests/test_gpu_decode.py tests/test_gpu_kernels.py 2>&1 | tail -60 | tee $OUT/${TAG}_pytest_tail.txt
bash tools/ab_env_loop.sh 3 "VLT5_GEMM_SPLIT_CAP=8" "VLT5_GEMM_SPLIT_CAP=6" "VLT5_GEMM_SPLIT_CAP=4" "VLT5_GEMM_SPLIT_CAP=3" "VLT5_GEMM_SPLIT_CAP=2" "VLT5_GEMM_SPLIT_CAP=4 VLT5_LIB=$ROOT/vqacl_amd/libvlt5_lnb640.so" 2>&1 | tee $OUT/${TAG}_ab_split_cap.txt
python3 bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench.err; tail -c 6000 $OUT/${TAG}_bench_line.json
