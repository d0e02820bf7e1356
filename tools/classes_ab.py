"""Same-box comparison of bench.py's per-class HIP-event survey for two argument sets: python tools/classes_ab.py "--persist 0" "--persist 1"."""
import json, subprocess, sys
def run(args):
    out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline"] + args.split(), capture_output=True, text=True).stdout.strip().splitlines()[-1]
    return json.loads(out)
a, b = run(sys.argv[1]), run(sys.argv[2])
print("ms/step", a["ms_per_step"], b["ms_per_step"], "median", a["step_ms"]["median"], b["step_ms"]["median"])
keys = sorted(set(a["kernel_classes"]) | set(b["kernel_classes"]))
ta = tb = 0
for k in keys:
    x = a["kernel_classes"].get(k, {"ms_per_step": 0, "launches_per_step": 0}); y = b["kernel_classes"].get(k, {"ms_per_step": 0, "launches_per_step": 0})
    ta += x["ms_per_step"]; tb += y["ms_per_step"]
    print("%-22s %7.3f (%5.1f)   %7.3f (%5.1f)   %+.3f" % (k, x["ms_per_step"], x["launches_per_step"], y["ms_per_step"], y["launches_per_step"], y["ms_per_step"] - x["ms_per_step"]))
print("sum of classes %.3f %.3f" % (ta, tb))
