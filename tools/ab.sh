#!/bin/bash
# usage: tools/ab.sh VAR val1 val2 [reps] ; alternates bench runs with VAR=val1 / VAR=val2 on the same box
cd $GRAFT_REPO_ROOT
for i in $(seq 1 ${4:-2}); do
  for v in $2 $3; do
    echo -n "$1=$v  "
    env $1=$v timeout 200 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])'
  done
done
