"""Per-kernel summary of the two SQ counter passes of tools/pmc_sq.sh -> <dir>/mfma_util.csv + a text table.

Units (MI355X_MICROARCH.md §Per-instruction cycle constants): SQ_VALU_MFMA_BUSY_CYCLES counts cycles of the matrix pipes (summed over SIMDs),
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves, SQ_BUSY_CYCLES is per SQ (one per XCD-SE slice).
  mfma_util  = SQ_VALU_MFMA_BUSY_CYCLES / (kernel time x 2.4 GHz x 1024 SIMDs)        fraction of the chip's matrix-pipe cycles in use
  mfma/busy  = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES                               the ratio VERDICT r1 asks for (raw counters)
  lds_confl  = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE                                extra LDS cycles per LDS-array cycle
  wait_any / wait_inst / wait_lds = share of SQ_WAVE_CYCLES a wave is parked (s_waitcnt / barrier), issue-stalled, LDS-issue-stalled
  MOPS: SQ_INSTS_VALU_MFMA_MOPS_BF16 / _F16 count 512-FLOP units ("MOPS") -> algorithmic MFMA TFLOP/s = MOPS x 512 / time
"""
import collections, csv, glob, sys

d = sys.argv[1]
CLOCK_HZ, NSIMD = 2.4e9, 1024


def load(passdir):
    fs = glob.glob(f"{passdir}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].replace("hulc_bf16::", "").replace("hulc_f16::", "").replace("unsigned short", "h16")
        if "skinny_lds_kernel" in k or "skinny_gemm_kernel" in k:
            k = k.split("(")[0] + f" [grid={r['Grid_Size']}]"
        else:
            k = k.split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    tr = glob.glob(f"{passdir}/**/*kernel_trace.csv", recursive=True)
    dur = collections.defaultdict(float)
    for r in csv.DictReader(open(tr[0])):
        k = r["Kernel_Name"].replace("hulc_bf16::", "").replace("hulc_f16::", "").replace("unsigned short", "h16")
        if "skinny_lds_kernel" in k or "skinny_gemm_kernel" in k:
            grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            k = k.split("(")[0] + f" [grid={grid}]"
        else:
            k = k.split("(")[0]
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    return agg, {k: len(v) for k, v in disp.items()}, dur


a1, n1, t1 = load(f"{d}/pass1")
a2, n2, t2 = load(f"{d}/pass2")
rows = []
for k in a1:
    c, c2 = a1[k], a2.get(k, {})
    t = t1[k]
    if t <= 0:
        continue
    wc = max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0)
    mops = c2.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0) + c2.get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0.0)
    rows.append(dict(kernel=k, launches=n1[k], us_per_launch=t / n1[k] * 1e6, total_us=t * 1e6,
                     mfma_util=c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (t * CLOCK_HZ * NSIMD),
                     mfma_over_busy=c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(c.get("SQ_BUSY_CYCLES", 0.0), 1.0),
                     lds_conflict=c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(c.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0),
                     wait_any=c.get("SQ_WAIT_ANY", 0.0) / wc, wait_inst=c.get("SQ_WAIT_INST_ANY", 0.0) / wc, wait_lds=c.get("SQ_WAIT_INST_LDS", 0.0) / wc,
                     mfma_tflops=mops * 512 / max(t2.get(k, t), 1e-12) / 1e12 if mops else 0.0,
                     lds_insts=c2.get("SQ_INSTS_LDS", 0.0) / max(n2.get(k, 1), 1), valu_insts=c2.get("SQ_INSTS_VALU", 0.0) / max(n2.get(k, 1), 1)))
rows.sort(key=lambda r: -r["total_us"])
with open(f"{d}/mfma_util.csv", "w") as fo:
    w = csv.DictWriter(fo, fieldnames=list(rows[0].keys()))
    w.writeheader()
    for r in rows:
        w.writerow({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()})
print(f"{'kernel':72s} {'n':>4s} {'us/launch':>9s} {'mfma_util':>9s} {'mfma/busy':>9s} {'MFMA TF/s':>9s} {'lds_confl':>9s} {'wait_any':>8s} {'wait_inst':>9s} {'wait_lds':>8s}")
for r in rows[:45]:
    print(f"{r['kernel'][:72]:72s} {r['launches']:4d} {r['us_per_launch']:9.1f} {r['mfma_util']:9.3f} {r['mfma_over_busy']:9.3f} {r['mfma_tflops']:9.1f} {r['lds_conflict']:9.3f} "
          f"{r['wait_any']:8.3f} {r['wait_inst']:9.3f} {r['wait_lds']:8.3f}")
