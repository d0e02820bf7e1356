"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), as MI355X_MICROARCH.md §HBM prescribes:
bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per dispatch (gfx950 FETCH_SIZE reports half of a wide coalesced read; both
counters are in KiB).  Writes <dir>/pmc_hbm_per_kernel.csv and <dir>/pmc_traffic.json (class -> bytes per launch)."""
import collections, csv, glob, json, sys

d = sys.argv[1]
# the recurrent step is the M=64, N=K=2048 launch of the skinny kernel: grid (128, 2) x 1024 threads (the same kernel also serves
# many-row GEMMs with other grids; dispatches are keyed by kernel name + grid size so those stay out of the class)
# patterns are substrings that survive re-templating (round 2 lost the conv classes when the tile kernels gained a wave-count template
# parameter: "false>(" no longer matched "false, 8>("): the REV flag is matched as ", false," / ", true," anywhere in the argument list
CLASSES = [("rnn_persist", ("rnn_persist_kernel",)), ("rnn_step_gemm", (("skinny_lds_kernel<2, 4, 16, false>", "[grid=262144]"), ("skinny_lds_kernel<2, 8>", "[grid=131072]"))), ("skinny_gemm", ("skinny_lds_kernel", "skinny_gemm_kernel")), ("gemm_128x128", ("gemm_glds_kernel", "gemm_kernel<unsigned short, 128, 128", "gemm_kernel<h16, 128, 128")),
           ("conv1_fwd", ("conv1_fwd_kernel",)), ("conv1_wgrad", ("conv1_wgrad_tr",)), ("conv_wgrad_tr", ("conv_wgrad_tr8_kernel", "conv_wgrad_tr_kernel<", "conv_wgrad_dma_kernel")),
           ("conv_tile_fwd", (("conv_reg_kernel<", ", false,"), ("conv_tile_kernel<", ", false,"))), ("conv_tile_dgrad", (("conv_reg_kernel<", ", true,"), ("conv_tile_kernel<", ", true,"))), ("adam", ("adam_kernel", "adam_tiled_kernel"))]
per = {}
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob(f"{d}/pmc_{name}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: [0.0, set()])
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] != name:
            continue
        k = r["Kernel_Name"]
        if "skinny_lds_kernel" in k:
            k += f" [grid={r['Grid_Size']}]"
        agg[k][0] += float(r["Counter_Value"]); agg[k][1].add(r["Dispatch_Id"])
    per[name] = {k: (v[0], len(v[1])) for k, v in agg.items()}
rows = []
for k in sorted(set(per["FETCH_SIZE"]) | set(per["WRITE_SIZE"])):
    f, nf = per["FETCH_SIZE"].get(k, (0.0, 1)); w, nw = per["WRITE_SIZE"].get(k, (0.0, 1))
    rows.append((k, nf, f / max(nf, 1), w / max(nw, 1), (2 * f / max(nf, 1) + w / max(nw, 1)) * 1024))
with open(f"{d}/pmc_hbm_per_kernel.csv", "w") as fo:
    wr = csv.writer(fo); wr.writerow(["kernel", "dispatches", "FETCH_SIZE_KiB_per_dispatch", "WRITE_SIZE_KiB_per_dispatch", "hbm_bytes_per_dispatch_corrected"])
    for r in sorted(rows, key=lambda r: -r[4] * r[1]):
        wr.writerow([r[0] if len(r[0]) <= 160 else r[0][:160] + (r[0][r[0].rfind(' [grid='):] if ' [grid=' in r[0] else ''), r[1], round(r[2], 1), round(r[3], 1), int(r[4])])
traffic = {}
acc = {cls: [0.0, 0] for cls, _ in CLASSES}
for r in rows:          # every dispatch row belongs to the FIRST class it matches: the recurrent-step dispatches (listed first, keyed by grid size)
    for cls, pats in CLASSES:      # therefore do not leak into `skinny_gemm`, whose algorithmic bytes they would not be comparable with
        if any((all(q in r[0] for q in p) if isinstance(p, tuple) else p in r[0]) for p in pats):
            acc[cls][0] += r[4] * r[1]; acc[cls][1] += r[1]
            break
for cls, (tot, n) in acc.items():
    if n:
        traffic[cls] = int(tot / n)
missing = [cls for cls, (tot, n) in acc.items() if not n]
if missing:
    print("WARNING: no dispatch matched class(es) " + ", ".join(missing) + " — update CLASSES in tools/pmc_traffic.py", file=sys.stderr)
json.dump(traffic, open(f"{d}/pmc_traffic.json", "w"), indent=1)
for k, v in traffic.items():
    print(f"{k:18s} {v / 1e6:10.2f} MB per launch")
