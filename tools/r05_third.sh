#!/bin/bash
# round 5, third GPU call: failing tests with full traces, in-step A/B of the tile-policy switches the step-faithful sweep suggests, norm-backward block count
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out; mkdir -p $OUT
TAG=${1:-r05_c}
cd $ROOT
VQACL_PARITY_LOG=$OUT/${TAG}_parity.txt timeout 1500 python3 -m pytest -m gpu -q -x --tb=long \
   "tests/test_gpu_model.py::test_data_parallel_overlap_path_on_one_gpu_and_rank_equivalence" tests/test_gpu_trajectory.py \
   "tests/test_gpu_model.py::test_encoder_kernel_choice_switches_with_the_batch_size_and_both_sides_match_the_oracle" \
   "tests/test_gpu_model.py::test_alternative_engine_paths_still_match_the_oracle" "tests/test_gpu_model.py::test_base_model_forward_backward_vs_oracle" \
   tests/test_gpu_decode.py 2>&1 | tail -80 | tee $OUT/${TAG}_pytest_tail.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
bash tools/ab_env_loop.sh 3 "VLT5_NOP=0" "VLT5_GEMM_RMKM_TILE=2" "VLT5_GEMM_RMRM_F32_TILE=2" "VLT5_GEMM_SPLIT_CAP=4" "VLT5_GEMM_RMKM_TILE=2 VLT5_GEMM_RMRM_F32_TILE=2 VLT5_GEMM_SPLIT_CAP=4" 2>&1 | tee $OUT/${TAG}_ab_tile_policy.txt
