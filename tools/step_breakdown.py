"""Single-step kernel breakdown from a rocprofv3 --kernel-trace CSV (the step before the last adam_kernel)."""
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if ("adam_kernel" in r["Kernel_Name"] or "adam_tiled_kernel" in r["Kernel_Name"])]
step = rows[idx[-2] + 1: idx[-1] + 1]
d = collections.defaultdict(lambda: [0.0, 0])
for r in step:
    n = r["Kernel_Name"].replace("unsigned short", "h16").replace("hulc_bf16::", "").replace("hulc_f16::", "")[:110]
    d[n][0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; d[n][1] += 1
tot = sum(v[0] for v in d.values())
print(f"{len(step)} kernels, busy {tot:.1f} us")
acc = 0
for n, v in sorted(d.items(), key=lambda kv: -kv[1][0])[: int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    acc += v[0]
    print(f"{v[0]:8.1f} us {v[1]:4d}x  cum {acc / tot * 100:5.1f}%  {n}")
