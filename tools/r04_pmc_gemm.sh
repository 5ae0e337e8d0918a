#!/bin/bash
# PMC passes (tools/pmc_gemm.sh) over the three GEMM kernel classes of the step, current build -> gpurun_out/r04_pmc_gemm_*.txt
cd "$(dirname "$0")/.."
TILES=224x256 bash tools/pmc_gemm.sh 4480 3072 768 0 0 > /dev/null 2>&1; cp gpurun_out/pmc_gemm.txt gpurun_out/r04_pmc_gemm_4480x3072x768_224x256.txt
TILES=256x256 bash tools/pmc_gemm.sh 3072 3072 4480 1 1 > /dev/null 2>&1; cp gpurun_out/pmc_gemm.txt gpurun_out/r04_pmc_gemm_wgrad_3072x3072x4480_256x256.txt
TILES=128x64 bash tools/pmc_gemm.sh 4480 768 768 0 1 > /dev/null 2>&1; cp gpurun_out/pmc_gemm.txt gpurun_out/r04_pmc_gemm_4480x768x768_128x64.txt
TILES=64x128 bash tools/pmc_gemm.sh 4480 768 3072 0 0 > /dev/null 2>&1; cp gpurun_out/pmc_gemm.txt gpurun_out/r04_pmc_gemm_4480x768x3072_64x128.txt
tail -n +1 gpurun_out/r04_pmc_gemm_*.txt
