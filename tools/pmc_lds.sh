#!/bin/bash
# usage: tools/pmc_lds.sh <outdir-name> ; collects LDS conflict counters for one bench step and prints the conv / gemm kernels
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 </dev/null
cd $GRAFT_REPO_ROOT
python - <<PY
import csv,glob,collections
fs=glob.glob("$out/**/*counter_collection.csv",recursive=True)
agg=collections.defaultdict(lambda:[0,0])
for r in csv.DictReader(open(fs[0])):
    n=r["Kernel_Name"]
    if any(k in n for k in ("conv_wgrad","conv_tile_kernel","conv1_","lin_bwd")):
        k=(n[:64],r["Counter_Name"]); agg[k][0]+=float(r["Counter_Value"]); agg[k][1]+=1
names=sorted(set(k[0] for k in agg))
for n in names:
    c=agg[(n,"SQ_LDS_BANK_CONFLICT")][0]; a=agg[(n,"SQ_LDS_IDX_ACTIVE")][0]
    print(f"{n:66s} conflict {c:.3g} active {a:.3g}  ratio {c/max(a,1):.2f}")
PY
