#!/bin/bash
# same-box A/B of two bench.py argument sets (alternating runs): tools/ab_args.sh "--persist 0" "--persist 1" [rounds]
A="$1"; B="$2"; R=${3:-3}
for i in $(seq $R); do
  a=$(python bench.py --no-cpu-baseline $A 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms']['median'])")
  b=$(python bench.py --no-cpu-baseline $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms']['median'])")
  echo "  [$A] $a   [$B] $b"
done
