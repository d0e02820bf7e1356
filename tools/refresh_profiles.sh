#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): regenerates every measurement profiles/ holds for a round into gpurun_out/<tag>/ (default tag r03);
# `tools/collect_profiles.py <tag>` then copies the summaries into profiles/ and rewrites profiles/README.md.
#   1. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) -> per-kernel-class HBM bytes per launch (pmc_traffic.json)
#   2. rocprofv3 --kernel-trace --stats of the bench command (9 steps: 2 warm-up + 2 survey + 5 timed)
#   3. two SQ counter passes (tools/pmc_sq.sh): MFMA-pipe utilisation, LDS bank conflicts, wait breakdown per kernel (mfma_util.csv)
#   4. the bench lines: N=1 default (with cpu_baseline), fp16, config-5 shapes, uint8 ingest, vis+lang, mcil variants
#   5. the probes behind DESIGN.md's numbers: tools/bin/gridbar2 (XCD barrier + sc1 publish),
#      tools/bin/rnn_persist_bench_st (the persistent recurrence alone: check against a CPU recurrence, us per step, phase stamps)
T=${1:-r06}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$T
mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python $R/bench.py --steps 2 --warmup 1 --preroll 0 --no-cpu-baseline > $O/pmc_$c.log 2>&1 </dev/null
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 5 --warmup 2 --preroll 0 --no-cpu-baseline > $O/stats.log 2>&1 </dev/null
cd $R
python tools/pmc_traffic.py $O > $O/pmc_traffic.log
mkdir -p $R/profiles && cp $O/pmc_traffic.json $R/profiles/${T}_pmc_traffic.json      # bench.py reads the newest profiles/r*_pmc_traffic.json
f=$(find $O/stats -name "*kernel_stats.csv" | head -1)
test -n "$f" && cp $f $O/kernel_stats.csv && python tools/prof_summary.py $O/kernel_stats.csv 9 45 > $O/kernel_stats_summary.txt
python tools/step_seq.py $O/stats > $O/step_sequence.txt 2>&1
( HULC_CT_FPB=1 python tools/time_conv_tile_gripper.py 2>/dev/null | tail -1; python tools/time_conv_tile_gripper.py 2>/dev/null | tail -1 ) > $O/conv_tile_gripper.txt
bash tools/pmc_sq.sh $T/sq > $O/sq.log 2>&1
cd $R
timeout 900 python $R/bench.py > $O/bench_n1.json 2> $O/bench_n1.err </dev/null
b() { name=$1; shift; timeout 400 python $R/bench.py --no-cpu-baseline "$@" > $O/bench_n1_$name.json 2>/dev/null </dev/null; }
b fp16 --dtype fp16
b s64 --seq 64 --batch 32
b s64_fp16 --seq 64 --batch 32 --dtype fp16
b s64_fp16_vislang --seq 64 --batch 32 --dtype fp16 --lang 1
b u8 --ingest u8
b vislang --lang 1
b vislang_seq --lang 1 --pair 0
b mcil --model mcil
b mcil_gru --model mcil_gru
b fp32 --dtype fp32 --steps 20                # the parity engine's throughput (v_mfma_f32_16x16x4_f32: exact fp32, 1/16 of the bf16 rate)
MASTER_PORT=29877 b rehearsal --force-comm 1          # 1-GPU rehearsal of the N > 1 code path: allreduce.selfcheck + allreduce.timeline (issue times of the five buckets)
b u8_h2d --ingest u8 --h2d 1                  # every step's uint8 frames copied from PINNED HOST memory (SURVEY 8(d)'s PCIe-inclusive row)
b u8_store --ingest u8 --store 16384          # windows gathered by index from a device-resident frame store, new random windows every step (hulc_batch::window_start)
MASTER_PORT=29878 b rehearsal_mcil --force-comm 1 --model mcil --lang 1                               # config 4 on the N > 1 code path: buckets held behind the persistent BiRNN backward
MASTER_PORT=29879 b rehearsal_mcil_early --force-comm 1 --model mcil --lang 1 --opt dp_hold_buckets=0    # ... round 5's schedule: early buckets, BiRNN backward one launch per step
python tools/time_conv_reg.py > $O/conv_reg_vs_tile.txt 2>/dev/null
ABLATE=1 python tools/time_conv_reg.py 2>/dev/null | tail -12 > $O/conv_reg_ablation.txt
test -x tools/bin/cr_bench && timeout 400 tools/bin/cr_bench ablate > $O/cr_bench.txt 2>&1          # conv_reg.h forms: slot decode in registers, pipelined epilogue, loader waves (round 5)
test -x tools/bin/cr_stamps && timeout 200 tools/bin/cr_stamps > $O/cr_stamps.txt 2>&1              # ... phase stamps of a band
test -x tools/bin/vmcnt_probe && timeout 200 tools/bin/vmcnt_probe > $O/vmcnt_probe.txt 2>&1        # LDS-DMA then store, counted vmcnt: in-order retirement probe
test -x tools/bin/storebench && timeout 120 tools/bin/storebench > $O/storebench.txt 2>&1
test -x tools/bin/mixbench && timeout 120 tools/bin/mixbench > $O/mixbench.txt 2>&1
python tools/step_timeline.py $O/stats "" 400 > $O/step_timeline.txt 2>&1
test -x tools/bin/rnn_persist_bench_st || { mkdir -p tools/bin; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -DRP_STAMPS tools/rnn_persist_bench.hip -o tools/bin/rnn_persist_bench_st; }
( timeout 120 tools/bin/rnn_persist_bench_st 64 32; timeout 120 tools/bin/rnn_persist_bench_st 128 32 | tail -5; timeout 120 tools/bin/rnn_persist_bench_st 32 64 | tail -5 ) > $O/rnn_persist_stamps.txt 2>&1
test -x tools/bin/gridbar2 && timeout 120 tools/bin/gridbar2 > $O/gridbar2.txt 2>&1
test -x tools/bin/gridbar && timeout 120 tools/bin/gridbar > $O/gridbar.txt 2>&1
tail -1 $O/bench_n1.json | cut -c1-400
head -8 $O/kernel_stats_summary.txt
tail -12 $O/pmc_traffic.log
head -14 $O/sq/summary.txt
