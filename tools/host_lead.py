import csv, sys, glob, collections
d = sys.argv[1]
kt = list(csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])))
ha = list(csv.DictReader(open(glob.glob(d + "/**/*hip_api_trace.csv", recursive=True)[0])))
print("hip api columns", list(ha[0].keys()))
api = {r["Correlation_Id"]: r for r in ha}
kt.sort(key=lambda r: int(r["Start_Timestamp"]))
# last full step: find the last adam_kernel, walk back to the previous one
idx = [i for i, r in enumerate(kt) if ("adam_kernel" in r["Kernel_Name"] or "adam_tiled_kernel" in r["Kernel_Name"])]
lo, hi = idx[-2] + 1, idx[-1] + 1
prev_end = int(kt[lo - 1]["End_Timestamp"])
t0 = int(kt[lo]["Start_Timestamp"])
for r in kt[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    a = api.get(r["Correlation_Id"])
    hs = (int(a["Start_Timestamp"]) - t0) / 1000 if a else float("nan")
    he = (int(a["End_Timestamp"]) - t0) / 1000 if a else float("nan")
    print("gpu %8.1f +%7.1f gap %5.1f | host call %9.1f .. %9.1f  lead %8.1f us  %s  %s" % ((s - t0) / 1000, (e - s) / 1000, (s - prev_end) / 1000, hs, he, (s - t0) / 1000 - he, a["Function"] if a else "?", r["Kernel_Name"][:60]))
    prev_end = e
