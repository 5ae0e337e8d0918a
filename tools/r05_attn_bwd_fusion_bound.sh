#!/bin/bash
# Upper bound of what fusing the encoder attention backward with the q|k|v input-gradient GEMM could save, measured
# in the step with two experiment builds whose RESULTS ARE WRONG but whose work is the fused kernel's best case:
#   nostore: attn_bwd_kernel computes everything and stores no dq / dk / dv at the encoder shape (producer side: 20.6 MB per layer stay on chip)
#   hota:    the input-gradient GEMM reads one cache-resident row for every row of dq|dk|dv (consumer side: the operand never comes from memory)
#   both:    the two together = a fusion with zero cost of its own
# build: bash tools/build_variant.sh nostore -DATTN_BWD_NO_STORE; ... hota -DENC_DGRAD_HOT_A; ... both -DATTN_BWD_NO_STORE -DENC_DGRAD_HOT_A
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
for r in 1 2 3; do
  for L in hip nostore hota both; do
    ms=$(VLT5_LIB=$ROOT/vqacl_amd/libvlt5_$L.so python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-parity --no-side-values 2>/dev/null | python3 -c 'import json,sys; print(json.loads(sys.stdin.readlines()[-1])["ms_per_step"])')
    echo "round $r  libvlt5_$L.so  $ms ms/step"
  done
done
