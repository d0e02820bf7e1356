"""Copy the summaries of gpurun_out/<tag>/ (written by tools/refresh_profiles.sh on the GPU box) into profiles/<tag>_* and regenerate the
number-bearing tables of profiles/README.md from them.   usage: python tools/collect_profiles.py r03"""
import json, os, shutil, sys, csv
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T = sys.argv[1] if len(sys.argv) > 1 else "r03"
RN = int(T[1:])
O = os.path.join(ROOT, "gpurun_out", T)
P = lambda f: os.path.join(ROOT, "profiles", f)
copies = {"bench_n1.json": "bench_n1.json", "kernel_stats.csv": "kernel_stats.csv", "kernel_stats_summary.txt": "kernel_stats_summary.txt",
          "pmc_hbm_per_kernel.csv": "pmc_hbm_per_kernel.csv", "pmc_traffic.json": "pmc_traffic.json", "sq/mfma_util.csv": "mfma_util.csv",
          "sq/summary.txt": "mfma_util_summary.txt", "gridbar2.txt": "gridbar_xcd_barrier.txt", "gridbar.txt": "gridbar_naive_barrier.txt", "step_sequence.txt": "step_sequence.txt",
          "conv_tile_gripper.txt": "conv_tile_gripper_fpb.txt", "conv_reg_vs_tile.txt": "conv_reg_vs_tile.txt", "conv_reg_ablation.txt": "conv_reg_ablation.txt",
          "step_timeline.txt": "step_timeline.txt", "rnn_persist_stamps.txt": "rnn_persist_stamps.txt",
          "storebench.txt": "storebench_write_patterns.txt", "mixbench.txt": "mixbench_conv1_traffic_shape.txt",
          "cr_bench.txt": "conv_reg_forms.txt", "cr_stamps.txt": "conv_reg_phase_stamps.txt", "vmcnt_probe.txt": "vmcnt_order_probe.txt"}
for k in ("fp16", "s64", "s64_fp16", "s64_fp16_vislang", "u8", "vislang", "vislang_seq", "mcil", "mcil_gru", "fp32", "u8_h2d", "rehearsal", "u8_store", "rehearsal_mcil", "rehearsal_mcil_early"):
    copies[f"bench_n1_{k}.json"] = f"bench_n1_{k}.json"
for src, dst in copies.items():
    if os.path.exists(os.path.join(O, src)):
        if src.startswith("bench_n1") and src.endswith(".json"):      # keep the JSON line only (a run that brings RCCL up prints its version banner first)
            lines = [l for l in open(os.path.join(O, src)).read().splitlines() if l.startswith('{"metric"')]
            if lines:
                open(P(f"{T}_{dst}"), "w").write(lines[-1] + "\n")
            continue
        shutil.copy(os.path.join(O, src), P(f"{T}_{dst}"))
J = lambda k: json.load(open(P(f"{T}_bench_n1{k}.json")))
d = J("")
t = json.load(open(P(f"{T}_pmc_traffic.json")))
rl, cb = d["roofline"], d["cpu_baseline"]
rows = "| class | ms/step | launches/step | algorithmic TFLOP/s | algorithmic GB/s | PMC HBM MB/launch |\n|---|---|---|---|---|---|\n"
for k, c in d["kernel_classes"].items():
    rows += f"| {k} | {c['ms_per_step']} | {c['launches_per_step']:.0f} | {c['tflops']} | {c['gbs']} | {t.get(k, 0) / 1e6:.1f} |\n"
mu = list(csv.DictReader(open(P(f"{T}_mfma_util.csv"))))
def grp(pred):
    sel = [r for r in mu if pred(r["kernel"])]
    tus = sum(float(r["total_us"]) for r in sel)
    if tus <= 0:
        return 0.0, 0.0, 0.0
    return (sum(float(r["mfma_util"]) * float(r["total_us"]) for r in sel) / tus, sum(float(r["mfma_tflops"]) * float(r["total_us"]) for r in sel) / tus, tus)
groups = [("conv (conv1 fwd/wgrad, conv2/3 fwd, dgrad, wgrad)", lambda k: "conv" in k and "unpack" not in k),
          ("RNN decoder, recurrences (rnn_persist: one launch per layer and direction; round 2: skinny_lds, M=64, one launch per step)", lambda k: "rnn_persist_kernel" in k or "skinny_lds_kernel<2, 4, 16, false> [grid=262144]" in k),
          ("RNN decoder, batched GEMMs (gemm_glds 128x128)", lambda k: "gemm_glds" in k),
          ("transformer + MLP small GEMMs (gemm_kernel 32/64 tiles, other skinny)", lambda k: ("gemm_kernel<" in k) or ("skinny" in k and "[grid=262144]" not in k) or "lin_bwd" in k)]
grows = "| GEMM group | MFMA-pipe utilisation (SQ_VALU_MFMA_BUSY_CYCLES / (time x 2.4 GHz x 1024 SIMDs)) | MFMA TFLOP/s (SQ_INSTS_VALU_MFMA_MOPS x 512 / time) | % of 2.5 PF | profiled us in the 5 dispatched steps |\n|---|---|---|---|---|\n"
for name, pred in groups:
    u, tf, tus = grp(pred)
    grows += f"| {name} | {u:.3f} | {tf:.0f} | {tf / 25.0:.1f} % | {tus:.0f} |\n"
var = {k: J("_" + k) for k in ("fp16", "s64", "s64_fp16", "s64_fp16_vislang", "u8", "vislang", "vislang_seq", "mcil", "mcil_gru", "fp32", "u8_h2d")}
sm, rs = d.get("step_ms") or {}, d.get("roofline_step") or {}
sc = var["s64_fp16"].get("loss_scaler") or {}
u8_store_row = ""
if os.path.exists(P(f"{T}_bench_n1_u8_store.json")):
    us = J("_u8_store")
    u8_store_row = (f"| `{T}_bench_n1_u8_store.json` | uint8 frames from an HBM-resident frame store (`hulc_batch::window_start`: B new random windows per step gathered by index, nothing crosses PCIe, "
                    f"no (B,S,H,W,C) tensor is materialised): {us['value']:.0f} windows/s, {us['ms_per_step']} ms/step | `python bench.py --ingest u8 --store 16384 --no-cpu-baseline` |\n")
files = f"""| file | what | command |
|---|---|---|
| `{T}_bench_n1.json` | the bench line (N=1): **{d['value']:.0f} windows/s, {d['ms_per_step']} ms/step**, `roofline` + `cpu_baseline` objects | `python bench.py` |
| `{T}_bench_n1_fp16.json` | the same step on the fp16 engine + on-device GradScaler (the reference's `precision: 16`): {var['fp16']['value']:.0f} windows/s, {var['fp16']['ms_per_step']} ms/step | `python bench.py --dtype fp16 --no-cpu-baseline` |
| `{T}_bench_n1_s64_fp16.json` | BASELINE config 5's shape AND precision: seq_len 64, 32 windows/GPU, fp16 + loss scaling: **{var['s64_fp16']['value']:.0f} windows/s, {var['s64_fp16']['ms_per_step']} ms/step** (scale {sc.get('scale')}, {sc.get('skipped_in_timed_region')} step(s) skipped in the timed region) | `python bench.py --seq 64 --batch 32 --dtype fp16 --no-cpu-baseline` |
| `{T}_bench_n1_s64_fp16_vislang.json` | config 5 with 16 vis + 16 lang windows + CLIP loss (paired pass): {var['s64_fp16_vislang']['value']:.0f} windows/s, {var['s64_fp16_vislang']['ms_per_step']} ms/step | `… --seq 64 --batch 32 --dtype fp16 --lang 1` |
| `{T}_bench_n1_s64.json` | seq_len 64, 32 windows/GPU in bf16: {var['s64']['value']:.0f} windows/s, {var['s64']['ms_per_step']} ms/step | `python bench.py --seq 64 --batch 32 --no-cpu-baseline` |
| `{T}_bench_n1_u8.json` | uint8 (B,S,H,W,C) ingest, transforms fused into conv1 (SURVEY §8(f) row 1): {var['u8']['value']:.0f} windows/s, {var['u8']['ms_per_step']} ms/step (boxes differ by ±2 %: the same-box ratio to the fp32 boundary is in `r06_ab_fp32_vs_u8_final.txt`) | `python bench.py --ingest u8 --no-cpu-baseline` |
{u8_store_row}| `{T}_bench_n1_vislang.json` | 32 vis + 32 lang + CLIP (BASELINE config 3 per GPU), one paired pass: {var['vislang']['value']:.0f} windows/s, {var['vislang']['ms_per_step']} ms/step | `python bench.py --lang 1 --no-cpu-baseline` |
| `{T}_bench_n1_vislang_seq.json` | the same, one pass per modality (the reference's order): {var['vislang_seq']['value']:.0f} windows/s, {var['vislang_seq']['ms_per_step']} ms/step | `python bench.py --lang 1 --pair 0 --no-cpu-baseline` |
| `{T}_bench_n1_mcil.json` | `model=mcil` (BiRNN plan recognition): {var['mcil']['value']:.0f} windows/s, {var['mcil']['ms_per_step']} ms/step | `python bench.py --model mcil --no-cpu-baseline` |
| `{T}_bench_n1_mcil_gru.json` | `rnn_type=nn.GRU` (BASELINE config 4's GRU plan encoder): {var['mcil_gru']['value']:.0f} windows/s, {var['mcil_gru']['ms_per_step']} ms/step (round 1: 12.44 ms) | `python bench.py --model mcil_gru --no-cpu-baseline` |
| `{T}_kernel_stats.csv`, `{T}_kernel_stats_summary.txt` | rocprofv3 per-kernel stats of the bench command (9 steps: 2 warm-up + 2 survey + 5 timed), top 45 per step | `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 5 --warmup 2 --preroll 0 --no-cpu-baseline`, `tools/prof_summary.py` |
| `{T}_pmc_hbm_per_kernel.csv`, `{T}_pmc_traffic.json` | FETCH_SIZE / WRITE_SIZE per dispatch (two separate `--pmc` passes) and the per-launch HBM bytes per kernel class, `(2 x FETCH_SIZE + WRITE_SIZE) x 1024` (MI355X_MICROARCH.md §HBM: gfx950 FETCH_SIZE reports half of a wide coalesced read; calibration: `adam` reads 5 and writes 3.5 arrays of 47.05 M fp32 = 1.41 GB algorithmic (+ 0.07 GB of transposed 16-bit copies since round 5) against {t.get('adam', 0) / 1e9:.2f} GB measured); every dispatch is counted in the FIRST class it matches, so the recurrent-step dispatches (keyed by their grid) are not in `skinny_gemm`; `bench.py` reports the dominant class's value as `roofline.traffic` | `tools/pmc_traffic.py` |
| `{T}_mfma_util.csv`, `{T}_mfma_util_summary.txt` | per kernel: MFMA-pipe utilisation, `SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES`, MFMA TFLOP/s from `SQ_INSTS_VALU_MFMA_MOPS_BF16`, LDS bank-conflict rate (`SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE`), wait breakdown — two SQ passes of 8 counters | `tools/pmc_sq.sh`, `tools/pmc_sq_summary.py` |
| `{T}_step_sequence.txt` | every launch of ONE step in order (consecutive identical launches collapsed) from the same kernel trace: launches per step, which memsets / transposes / small kernels remain and how long each takes.  The per-step call counts of `{T}_kernel_stats_summary.txt` divide a 9-step profile that also holds the engine's one-time workspace zero-fills (≈160 `fillBufferAligned`) and bench.py's input generation; a step itself issues 3 memsets (gradient buffer, loss slots, the backward's zero arena) | `tools/step_seq.py` |
| `{T}_bench_n1_fp32.json` | the fp32 PARITY engine (exact-fp32 MFMA, `v_mfma_f32_16x16x4_f32`: 1/16 of the bf16 matrix rate): {var['fp32']['value']:.0f} windows/s, {var['fp32']['ms_per_step']} ms/step | `python bench.py --dtype fp32 --steps 20 --no-cpu-baseline` |
| `{T}_bench_n1_u8_h2d.json` | uint8 ingest with every step's frames copied from PINNED HOST memory (async H2D on a copy stream, double-buffered; 289 MB per step): {var['u8_h2d']['value']:.0f} windows/s, median step {(var['u8_h2d'].get('step_ms') or {{}}).get('median')} ms — PCIe-bound, SURVEY §8(d)'s H2D-inclusive row (never the headline) | `python bench.py --ingest u8 --h2d 1 --no-cpu-baseline` |
| `{T}_conv_reg_vs_tile.txt`, `{T}_conv_reg_ablation.txt` | conv2 / conv3 forward and data gradient on 2048 frames: LDS-resident-weights kernels (conv_tile.h) against the weights-in-registers kernels (conv_reg.h); and conv_reg with phases switched off (no DMA / no multiply loop / no epilogue) | `tools/time_conv_reg.py`, `ABLATE=1 tools/time_conv_reg.py` |
| `{T}_rnn_persist_stamps.txt` | the persistent recurrence alone (`csrc/rnn_persist.h`): every step checked against a CPU recurrence, us per step at B = 64 / 128 (S = 32) and B = 32 (S = 64), shader-clock stamps of the phases of a step (poll, payload, MFMA + LDS, barrier, epilogue, drain) | `tools/bin/rnn_persist_bench_st` (tools/rnn_persist_bench.hip, -DRP_STAMPS) |
| `{T}_step_timeline.txt` | every launch of one step with start offset, duration, gap and queue | `tools/step_timeline.py` |
| `{T}_conv_tile_gripper_fpb.txt` | the four conv tile kernels on the gripper camera's shapes with 1 frame per band and with the stacked bands the launch picks | `tools/time_conv_tile_gripper.py` |
| `{T}_gridbar_xcd_barrier.txt`, `{T}_gridbar_naive_barrier.txt` | grid barrier + cross-XCD exchange cost with the fast primitives (XCD-hierarchical barrier, relaxed polls, `sc1` write-through publish) and with round 1's acquire-polled single counter | `tools/bin/gridbar2`, `tools/bin/gridbar` |
"""
s = open(P("README.md")).read()
marker = f"<!-- BEGIN generated {T} (tools/collect_profiles.py) -->"
end = f"<!-- END generated {T} -->"
gen = f"""{marker}
## Round {RN} files (`{T}_*`)

{files}
### Kernel classes, HIP-event timed inside bench.py (survey pass; includes event overhead)

{rows}
Dominant class `{rl['kernel']}` ({rl['bound']}-bound ruler): {rl['launches_per_step']:.0f} launches/step, {rl['avg_launch_us']} us per launch by HIP events on the engine's stream, algorithmic
{rl['per_launch']['algorithmic_bytes'] / 1e6:.2f} MB / {rl['per_launch']['algorithmic_flops'] / 1e9:.2f} GFLOP per launch -> {rl['achieved']:.0f} {rl['unit']} = **{rl['frac'] * 100:.1f} % of the {'8 TB/s HBM' if rl['bound'] == 'hbm' else '2.5 PFLOP/s bf16 MFMA'} roofline**; PMC traffic {(rl['traffic'] or 0) / 1e6:.1f} MB/launch.

### MFMA utilisation per GEMM group (rocprofv3 SQ counters, `{T}_mfma_util.csv`)

{grows}
Per-step device times (HIP events at the step boundaries, {d['steps']} timed steps): median {sm.get('median')} ms, p10 {sm.get('p10')}, p90 {sm.get('p90')} (wall-clock mean {d['ms_per_step']}).
Whole step against both ceilings with SURVEY §8(d)'s algorithmic work (`roofline_step`): {rs.get('algorithmic_bytes_per_step', 0) / 1e9:.2f} GB -> **{(rs.get('hbm_frac') or 0) * 100:.1f} % of the 8 TB/s HBM
roofline**, {(rs.get('mfma_frac') or 0) * 100:.1f} % of the bf16 MFMA peak; binding ceiling {rs.get('binding')} = {rs.get('ceiling_windows_per_s')} windows/s.
Whole step: {d['step_tflops']} TFLOP/s algorithmic (13.02 GFLOP/window x {d['value']:.0f} windows/s) = {d['step_tflops'] / 2500 * 100:.1f} % of the 2.5 PFLOP/s bf16 MFMA peak.
cpu_baseline (`kind: port`): {cb['value']} windows/s with {cb['cores']} BLAS threads — {cb['sample']}; sweep {[(r['threads'], r['windows_per_s']) for r in cb['sweep']]};
reference_anchor (the unmodified reference in the survey container, BASELINE.md §2): {cb['reference_anchor']['value']} windows/s on 8 vCPU.
{end}"""
if marker in s:
    s = s[:s.index(marker)] + gen + s[s.index(end) + len(end):]
else:                       # a new round: its block goes where the README marks it (or at the end)
    slot = f"<!-- {T} generated block goes here -->"
    s = s.replace(slot, gen) if slot in s else s.rstrip() + "\n\n" + gen + "\n"
open(P("README.md"), "w").write(s)
r = open(os.path.join(ROOT, "README.md")).read()
m0, m1 = "<!-- BEGIN numbers (tools/collect_profiles.py) -->", "<!-- END numbers -->"
txt = f"""{m0}
Round {RN} (1 x MI355X, B=64 windows, seq_len 32, bf16): **{d['value'] / 1000:.1f} k trajectory-windows/s, {d['ms_per_step']} ms/step** (median step {sm.get('median')} ms) at the
reference's fp32 boundary (round 1: 13.7 k / 4.686 ms; round 2: 14.9 k / 4.304; round 3: 18.4 k / 3.474; round 4: 19.5 k / 3.277; the pool's boxes differ by ±2 %, see profiles/README.md); {var['fp32']['value'] / 1000:.2f} k on the fp32 parity engine; {var['fp16']['value'] / 1000:.1f} k in fp16 with the on-device GradScaler (the reference's `precision: 16`);
{var['u8']['value'] / 1000:.1f} k with uint8 ingest ({var['u8_h2d']['value'] / 1000:.1f} k when every step's frames also cross PCIe from pinned host memory); {var['vislang']['value'] / 1000:.1f} k for 32 vis + 32 lang + CLIP as one paired pass ({var['vislang_seq']['value'] / 1000:.1f} k with the reference's one pass per
modality); BASELINE config 5 (seq_len 64 x 32 windows, fp16): {var['s64_fp16']['value'] / 1000:.2f} k.  CPU baseline (the step on torch's CPU library kernels, host cores of
the GPU box): {cb['value']} windows/s with {cb['cores']} threads; the reference itself did 8.0-12.2 windows/s on 8 vCPU (BASELINE.md).
Validation forward and stateful rollout (`validation_step`, `reset`/`step`, also for GCBC) run on the same kernels; the reference's `model=mcil`
configuration (BiRNN plan recognition, continuous latent plan) trains, validates and rolls out on the same engine: {var['mcil']['value'] / 1000:.1f} k windows/s
(`bench.py --model mcil`), {var['mcil_gru']['value'] / 1000:.1f} k with the `rnn_type=nn.GRU` plan encoder of BASELINE config 4 (`--model mcil_gru`).  Multi-GPU: one process per
GPU, the gradient all-reduce is the library's own (RCCL over xGMI, bucketed in reverse-forward order under the backward: `hulc_backward_allreduce`).
{m1}"""
r = r[:r.index(m0)] + txt + r[r.index(m1) + len(m1):]
open(os.path.join(ROOT, "README.md"), "w").write(r)
print("ok", d["value"], d["ms_per_step"])
