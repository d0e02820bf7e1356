cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04e
mkdir -p $O
for v in 0 15; do
  HULC_CONV_REG_W4=$v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_w$v -- python $R/bench.py --steps 5 --warmup 2 --preroll 0 --no-cpu-baseline > $O/stats_w$v.log 2>&1 </dev/null
  f=$(find $O/stats_w$v -name "*kernel_stats.csv" | head -1)
  cd $R; python tools/prof_summary.py $f 9 30 > $O/summary_w$v.txt; cd /tmp
done
cd $R
grep -h "conv_reg\|total kernel" $O/summary_w0.txt $O/summary_w15.txt
