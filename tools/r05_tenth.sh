#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
python tools/soak.py 1000 > gpurun_out/r05_zzz_soak.txt 2> gpurun_out/r05_ten_soak.err
for algo in allreduce rs_ag; do
  python bench.py --gpus 8 --rehearsal --dp-algo $algo --steps 2 --warmup 1 > gpurun_out/r05_zzz_rehearsal_n8_${algo}.json 2> gpurun_out/r05_ten_reh_${algo}.err
done
python bench.py --gpus 1 --force-dist --dp-algo zero1 --steps 20 --warmup 5 > gpurun_out/r05_zzz_bench_line_force_dist_zero1.json 2> gpurun_out/r05_ten_fd.err
tail -4 gpurun_out/r05_zzz_soak.txt
for f in gpurun_out/r05_zzz_rehearsal_n8_allreduce.json gpurun_out/r05_zzz_rehearsal_n8_rs_ag.json gpurun_out/r05_zzz_bench_line_force_dist_zero1.json; do python - "$f" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1], d.get("value"), d.get("ms_per_step"), d.get("rehearsal"), d.get("weights_in_sync"), d.get("ranks_seen", d.get("rccl_ranks_seen")), str(d.get("grad_exchange"))[:160])
PY
done
