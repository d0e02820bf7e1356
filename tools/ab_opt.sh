#!/bin/bash
# usage (GPU box): [BENCH_ARGS="--model mcil"] [REPS=3] tools/ab_opt.sh "name=0" "name=1" ... ; alternates short bench runs with each hulc_set_option setting
# (bench.py --opt) on the same box in the PRODUCTION build; prints ms/step (wall mean) and the median step per arm
cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py --no-cpu-baseline --steps 80 $BENCH_ARGS --opt $1 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.3f/%.3f" % (d["ms_per_step"], d["step_ms"]["median"]), end="")'; }
for rep in $(seq 1 ${REPS:-3}); do
  line=""
  for s in "$@"; do line="$line  $s $(run $s)"; done
  echo "$line"
done
