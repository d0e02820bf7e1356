#!/bin/bash
# usage (GPU box): [BENCH_ARGS="--ingest u8"] [REPS=3] tools/ab_lib.sh A.so B.so ... ; alternates short bench runs with each BUILD of libhulc_hip.so (HULC_LIB_PATH,
# hulc_amd/lib.py) on the same box; "-" = the in-tree build.  Build the other side with e.g.
#   git stash / git checkout <commit> -- hulc_amd/csrc && python __graft_entry__.py && cp hulc_amd/csrc/libhulc_hip.so tools/bin/libhulc_old.so   (tools/bin travels with gpurun, not with git)
cd $GRAFT_REPO_ROOT
run() { if [ "$1" = "-" ]; then unset HULC_LIB_PATH; else export HULC_LIB_PATH=$GRAFT_REPO_ROOT/$1; fi
        timeout 300 python bench.py --no-cpu-baseline --steps 80 $BENCH_ARGS 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.3f/%.3f" % (d["ms_per_step"], d["step_ms"]["median"]), end="")'; }
for rep in $(seq 1 ${REPS:-3}); do
  line=""
  for s in "$@"; do line="$line  $s $(run $s)"; done
  echo "$line"
done
