#!/bin/bash
# usage (on the GPU box, via gpurun): tools/pmc_sq.sh <outdir-name> [bench args...]
# Two SQ counter passes (8 SQ slots each, MI355X_MICROARCH.md §rocprofv3 PMC slots; --kernel-trace only, no other trace domain) over a short
# bench run, summarised per kernel by tools/pmc_sq_summary.py: MFMA-pipe utilisation, LDS bank-conflict rate, wait breakdown.
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $out
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --output-format csv -d $out/pass1 -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --preroll 0 --no-cpu-baseline "$@" > $out/pass1.log 2>&1 </dev/null
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM \
  --output-format csv -d $out/pass2 -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --preroll 0 --no-cpu-baseline "$@" > $out/pass2.log 2>&1 </dev/null
cd $GRAFT_REPO_ROOT
python tools/pmc_sq_summary.py $out > $out/summary.txt 2>&1
head -40 $out/summary.txt
