# on the GPU box (experiment build): rocprof kernel durations of the weight-gradient kernels under environment settings
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04g; mkdir -p $O
for s in "$@"; do
  tag=$(echo "$s" | tr ' =' '__')
  env $s timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/wg_$tag -- python $R/tools/time_wgrad.py > $O/wg_$tag.log 2>&1 </dev/null
  f=$(find $O/wg_$tag -name "*kernel_stats.csv" | head -1)
  echo "== $s"; grep "conv_wgrad" $f | awk -F'","' '{gsub(/"/,"",$1); printf "  %-70s calls %s avg_ns %s\n", substr($1,1,70), $2, $4}'
done
