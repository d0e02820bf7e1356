# on the GPU box: per-kernel durations of one bench configuration.  usage: bash tools/prof_model.sh OUTTAG <bench.py args...>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=$1; shift
O=$R/gpurun_out/$T; mkdir -p $O; rm -rf $O/stats
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 5 --warmup 2 --preroll 0 --no-cpu-baseline "$@" > $O/stats.log 2>&1 </dev/null
f=$(find $O/stats -name "*kernel_stats.csv" | head -1)
(cd $R; python tools/prof_summary.py $f 9 30 > $O/summary.txt; python tools/step_seq.py $O/stats > $O/seq.txt)
cut -c1-170 $O/summary.txt; head -1 $O/seq.txt
