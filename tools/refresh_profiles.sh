#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): regenerates every measurement that profiles/ holds for this round into gpurun_out/r01/.
#   1. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) -> per-kernel-class HBM bytes per launch (profiles/r01_pmc_traffic.json)
#   2. rocprofv3 --kernel-trace --stats of the bench command (9 steps: 2 warm-up + 2 survey + 5 timed)
#   3. the official bench line (N=1, defaults, with cpu_baseline), which picks roofline.traffic from step 1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r01
mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python $R/bench.py --steps 2 --warmup 1 --preroll 0 --no-cpu-baseline > $O/pmc_$c.log 2>&1 </dev/null
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 5 --warmup 2 --preroll 0 --no-cpu-baseline > $O/stats.log 2>&1 </dev/null
cd $R
python tools/pmc_traffic.py $O > $O/pmc_traffic.log
cp $O/pmc_traffic.json $R/profiles/r01_pmc_traffic.json
f=$(find $O/stats -name "*kernel_stats.csv" | head -1)
test -n "$f" && cp $f $O/kernel_stats.csv && python tools/prof_summary.py $O/kernel_stats.csv 9 45 > $O/kernel_stats_summary.txt
timeout 600 python $R/bench.py > $O/bench_n1.json 2> $O/bench_n1.err </dev/null
timeout 300 python $R/bench.py --ingest u8 --no-cpu-baseline > $O/bench_n1_u8.json 2>/dev/null </dev/null
timeout 300 python $R/bench.py --lang 1 --no-cpu-baseline > $O/bench_n1_vislang.json 2>/dev/null </dev/null
timeout 300 python $R/bench.py --lang 1 --pair 0 --no-cpu-baseline > $O/bench_n1_vislang_seq.json 2>/dev/null </dev/null
timeout 300 python $R/bench.py --model mcil --no-cpu-baseline > $O/bench_n1_mcil.json 2>/dev/null </dev/null
timeout 300 python $R/bench.py --model mcil_gru --no-cpu-baseline > $O/bench_n1_mcil_gru.json 2>/dev/null </dev/null
timeout 300 python $R/bench.py --seq 64 --batch 32 --no-cpu-baseline > $O/bench_n1_s64.json 2>/dev/null </dev/null
tail -1 $O/bench_n1.json | cut -c1-400
head -8 $O/kernel_stats_summary.txt
cat $O/pmc_traffic.log | tail -12
