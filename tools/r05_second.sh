#!/bin/bash
# round 5, second GPU call: the tests the first call did not reach, DP overhead table, B = 4 host/graph probes, norm-backward A/B, step-faithful GEMM sweep
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out; mkdir -p $OUT
TAG=${1:-r05_b}
cd $ROOT
VQACL_PARITY_LOG=$OUT/${TAG}_parity.txt VQACL_PARITY_PINS_OUT=$OUT/${TAG}_parity_pins.json timeout 2400 python3 -m pytest -m gpu -q --durations=8 \
   tests/test_gpu_model.py tests/test_gpu_trajectory.py tests/test_gpu_kernels.py tests/test_gpu_next_rows.py tests/test_gpu_sublayers.py tests/test_gpu_probe.py \
   "tests/test_gpu_bench_line.py::test_bench_line_contract_short_run" "tests/test_gpu_bench_line.py::test_bench_rehearsal_n_ranks_on_one_gpu[8-auto]" 2>&1 | tail -40 | tee $OUT/${TAG}_pytest_tail.txt
# DP overhead at world size 1: kernel stats of the plain step and of the step through the wrapper (as a rank: the profiled process IS the rank)
cd /tmp && export TMPDIR=/tmp
F="--steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-parity --no-side-values"
rocprofv3 --kernel-trace -d $OUT/${TAG}_kp -o r -- python3 $ROOT/bench.py $F > $OUT/${TAG}_kp.log 2>&1
python3 $ROOT/tools/rocpd_stats.py $(find $OUT/${TAG}_kp -name "*.db" | head -1) > $OUT/${TAG}_kernel_stats_plain.txt
WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 rocprofv3 --kernel-trace -d $OUT/${TAG}_kd -o r -- python3 $ROOT/bench.py --gpus 1 --force-dist $F > $OUT/${TAG}_kd.log 2>&1
python3 $ROOT/tools/rocpd_stats.py $(find $OUT/${TAG}_kd -name "*.db" | head -1) > $OUT/${TAG}_kernel_stats_force_dist.txt
rocprofv3 --kernel-trace -d $OUT/${TAG}_k4 -o r -- python3 $ROOT/bench.py --batch 4 $F > $OUT/${TAG}_k4.log 2>&1
python3 $ROOT/tools/rocpd_stats.py $(find $OUT/${TAG}_k4 -name "*.db" | head -1) > $OUT/${TAG}_kernel_stats_b4.txt
rm -rf $OUT/${TAG}_kp $OUT/${TAG}_kd $OUT/${TAG}_k4
cd $ROOT
python3 tools/dp_overhead_table.py $OUT/${TAG}_kernel_stats_plain.txt $OUT/${TAG}_kernel_stats_force_dist.txt > $OUT/${TAG}_dp_overhead.txt 2>&1
tail -4 $OUT/${TAG}_dp_overhead.txt
B="--steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-side-values --no-roofline"
for i in 1 2; do
  python3 bench.py $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('plain', d['ms_per_step'])"
  python3 bench.py --gpus 1 --force-dist $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('force-dist', d['ms_per_step'], d['grad_exchange']['algo'])"
  VLT5_WGRAD_SHADOW=3 python3 bench.py --gpus 1 --force-dist $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('force-dist, decoder wgrads as own launches (r04)', d['ms_per_step'])"
done
python3 tools/step_graph_probe.py 4 8 16 32 80 2>&1 | grep -v amdgpu.ids | tee $OUT/${TAG}_step_graph_probe.txt
python3 tools/ln_bench.py 2>&1 | grep -v amdgpu.ids | tee $OUT/${TAG}_ln_bench.txt
bash tools/ab_run.sh vqacl_amd/libvlt5_hip.so vqacl_amd/libvlt5_lnb640.so 2 --no-side-values --no-parity 2>/dev/null | tee $OUT/${TAG}_ab_lnb640.txt
bash tools/ab_run.sh vqacl_amd/libvlt5_hip.so vqacl_amd/libvlt5_lnb1120.so 2 --no-side-values --no-parity 2>/dev/null | tee $OUT/${TAG}_ab_lnb1120.txt
timeout 1200 python3 tools/gemm_step_sweep.py 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_gemm_step_sweep.txt; tail -25 $OUT/${TAG}_gemm_step_sweep.txt
