#!/bin/bash
# PMC passes over the decode loop (one pass per counter set): where a declin launch spends its cycles in the memory path
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out; mkdir -p $OUT
TAG=${1:-r04_f}
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "TCP_[A-Z0-9_]*\|TCC_[A-Z0-9_]*\|TA_[A-Z0-9_]*\|TD_[A-Z0-9_]*" | sort -u > $OUT/${TAG}_counter_names.txt
i=0
for SET in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" "TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-trace -d $OUT/${TAG}_p$i -o r -- python3 $ROOT/tools/decode_bench.py --fast-only > $OUT/${TAG}_p$i.log 2>&1
  echo "pass $i ($SET): rc=$? $(find $OUT/${TAG}_p$i -name '*.db' | wc -l) db"
  tail -2 $OUT/${TAG}_p$i.log | cut -c1-200
done
python3 $ROOT/tools/rocpd_counters.py $(find $OUT/${TAG}_p* -name "*.db") --match dec > $OUT/${TAG}_decode_pmc.txt
rm -rf $OUT/${TAG}_p[0-9]
cat $OUT/${TAG}_decode_pmc.txt
