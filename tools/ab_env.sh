#!/bin/bash
# usage: tools/ab_env.sh "VAR=a" "VAR=b" ... ; alternates short bench runs with each env setting on the same box (2 repetitions)
cd $GRAFT_REPO_ROOT
run() { env $1 timeout 200 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])'; }
for rep in 1 2; do
  line=""
  for s in "$@"; do line="$line  $s $(run $s)"; done
  echo "$line"
done
