#!/bin/bash
# same-box A/B of one environment variable: bash tools/ab_env.sh VAR A B [reps] [extra bench.py flags]   (value "unset" leaves VAR unset)
cd "$(dirname "$0")/.." || exit 1
V="$1"; A="$2"; B="$3"; N="${4:-3}"
shift 4 2>/dev/null || shift $#
ERR=$(mktemp)
for i in $(seq "$N"); do
  for x in "$A" "$B"; do
    if [ "$x" = unset ]; then
      OUT=$(env -u "$V" python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-parity --no-side-values "$@" 2>"$ERR")
    else
      OUT=$(env "$V=$x" python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-parity --no-side-values "$@" 2>"$ERR")
    fi
    rc=$?
    if [ $rc -ne 0 ]; then            # a failing bench run shows its own error, not a JSON traceback of the parser below
      echo "$V=$x: bench.py exited with $rc" >&2
      tail -20 "$ERR" >&2
      continue
    fi
    printf '%s\n' "$OUT" | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V=$x', d['ms_per_step'])"
  done
done
rm -f "$ERR"
