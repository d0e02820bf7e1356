# on the GPU box (experiment build): per-kernel durations of the bench step under environment settings.  usage: bash tools/prof_ab.sh OUTTAG "VAR=a" "VAR=b" ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=$1; shift
O=$R/gpurun_out/$T; mkdir -p $O
for s in "$@"; do
  tag=$(echo "$s" | tr ' =' '__')
  rm -rf $O/stats_$tag
  env $s timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$tag -- python $R/bench.py --steps 5 --warmup 2 --preroll 0 --no-cpu-baseline > $O/stats_$tag.log 2>&1 </dev/null
  f=$(find $O/stats_$tag -name "*kernel_stats.csv" | head -1)
  (cd $R; python tools/prof_summary.py $f 9 40 > $O/summary_$tag.txt)
  echo "== $s"; grep -h "total kernel\|conv_wgrad\|conv_reg" $O/summary_$tag.txt | cut -c1-120
done
