"""Summarise a rocprofv3 --kernel-trace --stats CSV (kernel_stats.csv): top kernels, per-step totals."""
import csv
import sys

f = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.3f} ms over {steps:g} steps -> {tot / 1e6 / steps:.3f} ms/step")
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    n = r["Name"].replace("gemm_kernel", "G").replace("Loader", "L").replace("unsigned short", "bf16")[:120]
    print(f"{float(r['TotalDurationNs']) / 1e6 / steps:8.3f} ms/step {float(r['Percentage']):6.2f}% calls/step {float(r['Calls']) / steps:7.1f} avg {float(r['AverageNs']) / 1e3:9.1f} us  {n}")
