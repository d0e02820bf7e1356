"""Timeline of the last full step in a rocprofv3 --kernel-trace directory: every launch with its start offset, duration, gap to the previous
launch on the same queue and the queue it ran on — what shows whether the side stream's weight-gradient launches really overlap the main chain.
python tools/step_timeline.py <dir> [from-kernel-substring] [max-lines]"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
start_at = sys.argv[2] if len(sys.argv) > 2 else ""
maxn = int(sys.argv[3]) if len(sys.argv) > 3 else 400
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if ("adam_kernel" in r["Kernel_Name"] or "adam_tiled_kernel" in r["Kernel_Name"])]
step = rows[idx[-2] + 1: idx[-1] + 1]
short = lambda n: n.replace("hulc_bf16::", "").replace("hulc_f16::", "").replace("void ", "").split("(")[0][:60]
t0 = int(step[0]["Start_Timestamp"])
qs = {}
last_end = {}
on = start_at == ""
n = 0
busy = {}
for r in step:
    q = r.get("Queue_Id", "?")
    qs.setdefault(q, len(qs))
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    busy[q] = busy.get(q, 0) + (e - s)
    name = short(r["Kernel_Name"])
    if not on and start_at in name: on = True
    if on and n < maxn:
        print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  gap {gap:6.1f}  q{qs[q]}  {name}  grid={r['Grid_Size_X']}x{r['Grid_Size_Y']}/{r['Workgroup_Size_X']}")
        n += 1
print("span %.1f us; busy per queue: %s" % ((int(step[-1]["End_Timestamp"]) - t0) / 1e3, {f"q{qs[q]}": round(v / 1e3, 1) for q, v in busy.items()}))
