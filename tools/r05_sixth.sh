#!/bin/bash
# default bench line of the final tree (quotes the committed r05_zzz trace) + dropout moments
mkdir -p gpurun_out
cd /root/repo
( time python bench.py --steps 20 --warmup 5 > gpurun_out/r05_zzz_bench_line_with_trace.json 2> gpurun_out/r05_six_bench.err ) 2> gpurun_out/r05_six_bench_time.txt
python tools/dropout_moments.py --seeds 48 --warm 0 60 --out gpurun_out/r05_zzz_dropout_moments.txt > gpurun_out/r05_six_dm.log 2>&1
python tools/dropout_moments.py --batch 32 --seeds 16 --warm 30 --out gpurun_out/r05_six_dropout_moments_b32_k16.txt >> gpurun_out/r05_six_dm.log 2>&1
python tools/dropout_moments.py --batch 32 --seeds 16 --warm 30 --out gpurun_out/r05_six_dropout_moments_b32_k16_again.txt >> gpurun_out/r05_six_dm.log 2>&1
tail -5 gpurun_out/r05_six_dm.log
cat gpurun_out/r05_six_bench_time.txt
