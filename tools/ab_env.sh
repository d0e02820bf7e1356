#!/bin/bash
# usage: [BENCH_ARGS="--model mcil"] [REPS=2] tools/ab_env.sh "VAR=a" "VAR=b" ... ; alternates short bench runs with each env setting on the same box
cd $GRAFT_REPO_ROOT
run() { env $1 timeout 300 python bench.py --no-cpu-baseline --steps 60 $BENCH_ARGS 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])'; }
for rep in $(seq 1 ${REPS:-2}); do
  line=""
  for s in "$@"; do line="$line  $s $(run $s)"; done
  echo "$line"
done
