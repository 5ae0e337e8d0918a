#!/bin/bash
# round-4 first GPU pass: regression tests, the launcher on the GPU box (--force-dist), decode baseline, kernel traces plain vs force-dist
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd $ROOT
timeout 900 python3 -m pytest tests -m gpu -x -q > $OUT/r04_a_gputests.log 2>&1; echo "gpu tests rc=$?" | tee -a $OUT/r04_a_gputests.log
tail -3 $OUT/r04_a_gputests.log
B="--steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-side-values"
for i in 1 2; do
  python3 bench.py $B > $OUT/r04_a_plain_$i.json 2> $OUT/r04_a_plain_$i.err
  python3 bench.py --gpus 1 --force-dist $B > $OUT/r04_a_forcedist_$i.json 2> $OUT/r04_a_forcedist_$i.err; echo "launcher rc=$?"
done
python3 -c "
import json,glob
for f in sorted(glob.glob('$OUT/r04_a_*_?.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['ms_per_step'], d.get('rccl_ranks_seen'), d.get('grad_exchange'), d.get('launcher'))
    except Exception as e: print(f, 'ERR', e)
"
python3 tools/decode_bench.py > $OUT/r04_a_decode_before.txt 2>&1; cat $OUT/r04_a_decode_before.txt | tail -3
bash tools/kt.sh r04_a_plain > /dev/null 2>&1
WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 bash tools/kt.sh r04_a_dist --gpus 1 --force-dist > /dev/null 2>&1
head -5 $OUT/r04_a_plain_kernel_stats.txt $OUT/r04_a_dist_kernel_stats.txt
