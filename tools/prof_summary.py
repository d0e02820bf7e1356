"""Summarise a rocprofv3 --kernel-trace --stats CSV (kernel_stats.csv): top kernels, per-step totals.

The profiled command is the whole of `python bench.py` (context creation, input synthesis, warm-up, survey and timed steps), so the CSV also holds
launches that are NOT part of a training step; rows are tagged:
  [outside]  torch's own kernels (`at::native::*`: bench.py synthesises the batch with torch.rand / randint / fills, once per run) and the runtime's
             blit kernels (`__amd_rocclr_fillBufferAligned` / `copyBuffer`: the engine's one-time workspace zero-fills at hulc_ctx_create — one per
             allocation — and hipMemcpy of descriptors / weights at bind time).  A steady-state step launches none of them since round 5
             (`zero_grads` is a `multi_zero_kernel` launch; rounds <= 4: one 188 MB fillBufferAligned per step).
  [init+step] the step's own repack kernels that ALSO run once at bind time (`cast_kernel`, `weight_pack_kernel`, `batched_transpose64_kernel`,
             `frag_pack_kernel`): calls/step is slightly above the per-step count.
The per-launch sequence of ONE step is `<tag>_step_sequence.txt` (tools/step_seq.py)."""
import csv
import sys

f = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(f)))


def tag(name):
    if "at::native" in name or "__amd_rocclr" in name:
        return "[outside]  "
    if any(k in name for k in ("cast_kernel<float", "weight_pack_kernel", "batched_transpose64_kernel", "frag_pack_kernel")):
        return "[init+step]"
    return "           "


tot = sum(float(r["TotalDurationNs"]) for r in rows)
outside = sum(float(r["TotalDurationNs"]) for r in rows if tag(r["Name"]).startswith("[outside]"))
print(f"total kernel time {tot / 1e6:.3f} ms over {steps:g} steps -> {tot / 1e6 / steps:.3f} ms/step;  of it [outside] the step (bench.py input synthesis, one-time "
      f"workspace fills): {outside / 1e6:.3f} ms -> the library's kernels {(tot - outside) / 1e6 / steps:.3f} ms/step")
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    n = r["Name"].replace("gemm_kernel", "G").replace("Loader", "L").replace("unsigned short", "bf16")[:120]
    print(f"{float(r['TotalDurationNs']) / 1e6 / steps:8.3f} ms/step {float(r['Percentage']):6.2f}% calls/step {float(r['Calls']) / steps:7.1f} avg {float(r['AverageNs']) / 1e3:9.1f} us  {tag(r['Name'])} {n}")
