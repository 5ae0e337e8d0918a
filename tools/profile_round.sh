#!/bin/bash
# Round-end evidence run on the GPU box: bench line, rocprofv3 kernel stats of the same command, PMC passes (HBM traffic,
# MFMA utilisation), GEMM tile sweep.  usage: bash tools/profile_round.sh r01_e   -> gpurun_out/<tag>_*
set -u
TAG=${1:-rXX}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out
mkdir -p $OUT
python3 $ROOT/bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench.err
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_kt -o r -- python3 $ROOT/bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-roofline --no-parity --no-side-values > $OUT/${TAG}_kt.log 2>&1
python3 $ROOT/tools/rocpd_stats.py $(find $OUT/${TAG}_kt -name "*.db" | head -1) > $OUT/${TAG}_kernel_stats_bench_b80.txt
python3 $ROOT/tools/rocpd_timeline.py $(find $OUT/${TAG}_kt -name "*.db" | head -1) > $OUT/${TAG}_step_timeline.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/${TAG}_pf -o r -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-parity --no-side-values > $OUT/${TAG}_pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/${TAG}_pw -o r -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-parity --no-side-values > $OUT/${TAG}_pw.log 2>&1
python3 $ROOT/tools/rocpd_pmc.py $(find $OUT/${TAG}_pf -name "*.db" | head -1) $(find $OUT/${TAG}_pw -name "*.db" | head -1) $OUT/${TAG}_pmc_hbm_traffic.json > $OUT/${TAG}_pmc_hbm_traffic.txt
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --kernel-trace -d $OUT/${TAG}_pm -o r -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-parity --no-side-values > $OUT/${TAG}_pm.log 2>&1
python3 $ROOT/tools/rocpd_counters.py $(find $OUT/${TAG}_pm -name "*.db" | head -1) --match _kernel > $OUT/${TAG}_pmc_mfma_util.txt
cd $ROOT && python3 tools/gemm_sweep.py --graph --torch-ref > $OUT/${TAG}_gemm_tile_sweep.txt 2>&1
# the greedy-decoding loop: throughput of the three paths, kernel stats and one token-step launch by launch
python3 $ROOT/tools/decode_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_decode_bench.txt
cd /tmp
rocprofv3 --kernel-trace -d $OUT/${TAG}_dk -o r -- python3 $ROOT/tools/decode_bench.py --fast-only > $OUT/${TAG}_dk.log 2>&1
DDB=$(find $OUT/${TAG}_dk -name "*.db" | head -1)
python3 $ROOT/tools/rocpd_stats.py $DDB --steps 1 | grep -v "at::native" | head -24 > $OUT/${TAG}_decode_kernel_stats.txt
python3 $ROOT/tools/rocpd_window.py $DDB dec_io_kernel -4 > $OUT/${TAG}_decode_step_timeline.txt
# the data-parallel machinery on one GPU: the same bench through the rank launcher (RCCL, world size 1) and its kernel trace as a rank
cd $ROOT
python3 bench.py --gpus 1 --force-dist --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-side-values > $OUT/${TAG}_bench_line_force_dist.json 2> $OUT/${TAG}_force_dist.err
cd /tmp
WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 rocprofv3 --kernel-trace -d $OUT/${TAG}_dd -o r -- python3 $ROOT/bench.py --gpus 1 --force-dist --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-parity --no-side-values > $OUT/${TAG}_dd.log 2>&1
python3 $ROOT/tools/rocpd_stats.py $(find $OUT/${TAG}_dd -name "*.db" | head -1) > $OUT/${TAG}_kernel_stats_force_dist.txt
cd $ROOT && python3 tools/dp_overhead_table.py $OUT/${TAG}_kernel_stats_bench_b80.txt $OUT/${TAG}_kernel_stats_force_dist.txt > $OUT/${TAG}_dp_overhead.txt 2>&1
# round 5: the step-faithful GEMM sweep, the dress rehearsal of the N-rank bench path on this one GPU (flagged lines, no value), the
# accuracy proxy against the oracle on the same GPU, the step-graph probe (small batches: host-bound or not)
python3 tools/gemm_step_sweep.py 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_gemm_step_sweep.txt
for N in 2 4 8; do
  python3 bench.py --gpus $N --rehearsal --steps 2 --warmup 1 --no-cpu-baseline --no-parity > $OUT/${TAG}_rehearsal_n$N.json 2> $OUT/${TAG}_rehearsal_n$N.err
done
python3 tools/trajectory.py --steps-per-stage 100 --out gpurun_out/${TAG}_trajectory.txt > /dev/null 2>&1
# dropout on against the oracle in distribution: moments over dropout seeds from the same weights on the same batch
python3 tools/dropout_moments.py --seeds 48 --warm 0 20 --out gpurun_out/${TAG}_dropout_moments.txt > /dev/null 2>&1
python3 tools/dropout_moments.py --seeds 192 --warm 60 --out gpurun_out/${TAG}_dropout_moments_late.txt > /dev/null 2>&1
python3 tools/step_graph_probe.py 4 8 16 32 80 2>&1 | grep -v "amdgpu.ids\|UserWarning\|detach()\|print(f" > $OUT/${TAG}_step_graph_probe.txt
cd /tmp
rm -rf $OUT/${TAG}_dk $OUT/${TAG}_dd
# the raw rocpd databases are tens of MB each and gpurun only copies 64 MiB back: keep the summaries, drop the databases
rm -rf $OUT/${TAG}_kt $OUT/${TAG}_pf $OUT/${TAG}_pw $OUT/${TAG}_pm
ls -la $OUT | grep ${TAG}_ | head -30
