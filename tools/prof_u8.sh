# on the GPU box (experiment build): per-kernel durations of the uint8-ingest bench step.  usage: bash tools/prof_u8.sh OUTTAG "VAR=a" "VAR=b" ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=$1; shift
O=$R/gpurun_out/$T; mkdir -p $O
for s in "$@"; do
  tag=$(echo "$s" | tr ' =' '__')
  rm -rf $O/stats_$tag
  env $s timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$tag -- python $R/bench.py --steps 5 --warmup 2 --preroll 0 --no-cpu-baseline --ingest u8 > $O/stats_$tag.log 2>&1 </dev/null
  echo "== $s"; (cd $R; python tools/step_seq.py $O/stats_$tag conv1)
done
