#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out; mkdir -p $OUT
TAG=${1:-r05_t}
cd $ROOT
timeout 900 python3 -m pytest -m gpu -q --tb=short "tests/test_gpu_model.py::test_data_parallel_overlap_path_on_one_gpu_and_rank_equivalence" 2>&1 | tail -15
python3 bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench.err; python3 -c "
import json; d=json.loads(open('$OUT/${TAG}_bench_line.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r.get('frac_trace'), r.get('traffic'), r.get('traffic_build_matches'), d['feed']['frac'], d['feed'].get('frac_in_step_trace'))"
bash tools/r05_small_batch_ab.sh 2>&1 | tee $OUT/${TAG}_small_batch_ab.txt
