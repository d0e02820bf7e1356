"""Sequential kernel list of the last full step in a rocprofv3 --kernel-trace directory (consecutive identical launches collapsed):
python tools/step_seq.py <dir> [name-filter]"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if ("adam_kernel" in r["Kernel_Name"] or "adam_tiled_kernel" in r["Kernel_Name"])]
step = rows[idx[-2] + 1: idx[-1] + 1]
short = lambda n: n.replace("hulc_bf16::", "").replace("hulc_f16::", "").replace("void ", "").split("(")[0][:80]
out = []
for r in step:
    n = short(r["Kernel_Name"]); d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    g = (r["Grid_Size_X"], r["Grid_Size_Y"], r["Workgroup_Size_X"])
    if out and out[-1][0] == n and out[-1][2] == g: out[-1][1].append(d)
    else: out.append([n, [d], g])
print(f"{len(step)} launches, busy {sum(sum(o[1]) for o in out):.0f} us, span {(int(step[-1]['End_Timestamp']) - int(step[0]['Start_Timestamp'])) / 1e3:.0f} us")
for n, ds, g in out:
    if flt in n: print(f"{len(ds):3d} x {sum(ds) / len(ds):7.1f} us  {n}  grid={g}")
