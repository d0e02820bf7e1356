import csv, glob, os, sys
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r04g"
for d in sorted(glob.glob(os.path.join(root, "wg_*"))):
    if not os.path.isdir(d):
        continue
    fs = sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    if not fs:
        continue
    print("==", os.path.basename(d))
    for r in csv.DictReader(open(fs[-1])):
        if "conv_wgrad" in r["Name"] and "unpack" not in r["Name"]:
            print("   ", r["Name"][17:62], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "min", round(float(r["MinNs"]) / 1e3, 1))
