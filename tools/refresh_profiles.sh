#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): regenerates every measurement that profiles/ holds for this round into gpurun_out/r01/.
#   1. official bench line (N=1, defaults, with cpu_baseline)
#   2. rocprofv3 --kernel-trace --stats of the same command (9 steps: 2 warm-up + 2 survey + 5 timed)
#   3. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) -> per-kernel-class HBM bytes per launch
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r01
mkdir -p $O
timeout 600 python $R/bench.py > $O/bench_n1.json 2> $O/bench_n1.err </dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/stats.log 2>&1 </dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_$c.log 2>&1 </dev/null
done
cd $R
f=$(find $O/stats -name "*kernel_stats.csv" | head -1)
test -n "$f" && cp $f $O/kernel_stats.csv && python tools/prof_summary.py $O/kernel_stats.csv 9 45 > $O/kernel_stats_summary.txt
python tools/pmc_traffic.py $O > $O/pmc_traffic.log
tail -3 $O/bench_n1.json | cut -c1-600
head -12 $O/kernel_stats_summary.txt
cat $O/pmc_traffic.log | tail -12
