#!/bin/bash
# same-box A/B of one environment variable: bash tools/ab_env.sh VAR A B [reps]   (value "unset" leaves VAR unset)
cd "$(dirname "$0")/.."
V=$1; A=$2; B=$3; N=${4:-3}
for i in $(seq $N); do
 for x in $A $B; do
  if [ "$x" = unset ]; then E="env -u $V"; else E="env $V=$x"; fi
  $E python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-parity --no-side-values 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V=$x', d['ms_per_step'])"
 done
done
