#!/bin/bash
# usage: tools/prof_step.sh <outdir-name> [topN] ; rocprofv3 kernel trace of bench.py, single-step breakdown of the last step
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --preroll 0 --no-cpu-baseline > $out.log 2>&1 </dev/null
cd $GRAFT_REPO_ROOT
python tools/step_breakdown.py $out ${2:-30}
