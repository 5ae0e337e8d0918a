#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
python -m pytest tests/test_gpu_trajectory.py -q -m gpu -k dropout_on 2>&1 | tail -5 > gpurun_out/r05_seven_pytest.txt
python tools/dropout_moments.py --seeds 48 --warm 0 20 --out gpurun_out/r05_zzz_dropout_moments.txt > gpurun_out/r05_seven_dm.log 2>&1
python tools/dropout_moments.py --seeds 192 --warm 60 --out gpurun_out/r05_zzz_dropout_moments_late.txt >> gpurun_out/r05_seven_dm.log 2>&1
cat gpurun_out/r05_seven_pytest.txt; tail -3 gpurun_out/r05_seven_dm.log
