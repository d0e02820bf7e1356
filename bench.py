#!/usr/bin/env python
"""bench.py — trajectory-windows/s of the HULC training step on N MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = forward + loss + backward + gradient all-reduce (N>1, RCCL via torch.distributed "nccl") + Adam, on one
batch of synthetic CALVIN-shaped windows already resident in HBM.  Workload (BASELINE.json configs[1]):
HULC, vision goal only (use_clip_auxiliary_loss=false), B=64 windows/GPU, seq_len=32, 200x200 static + 84x84
gripper frames, bf16 MFMA operands with fp32 accumulation / fp32 master weights, transformer dropout 0.1 (train mode).
Prints ONE JSON line on rank 0 (contract in the task statement) incl. `roofline` and `cpu_baseline`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from hulc_amd import parallel, spec  # noqa: E402
from hulc_amd.engine import StepEngine  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_BF16_PEAK_TFLOPS = 2500.0
FLOP_PER_WINDOW_S32 = 13.02e9   # fwd+bwd algorithmic FLOPs per window at S=32

# GEMM groups of the step (SURVEY §8(d) "per-group MFMA utilisation", north_star "MFMA utilisation against gfx950 peak"): the engine's HIP-event
# kernel classes (csrc/engine.h TimerScope names) that make up each group.  `roofline` is taken on the heaviest GROUP (a group's classes are
# timed together), not on whichever single class happens to be a few microseconds ahead in the survey pass.
MFMA_GROUPS = {
    "conv": ["conv_tile_fwd", "conv_tile_dgrad", "conv_wgrad_tr"],        # conv2 / conv3 forward, data and weight gradients (MFMA-bound by design)
    "conv1": ["conv1_fwd", "conv1_wgrad"],                                # first-layer convolutions on the fp32 / uint8 boundary frames (HBM-bound)
    "rnn_recurrent": ["rnn_persist", "rnn_step_gemm", "gru_step"],        # the recurrences' dependent M = B GEMMs
    "rnn_batched": ["gemm_128x128"],                                      # the hoisted input projections / 2048^2 weight gradients (128 x 128 tiles)
    "transformer_mlp": ["transformer_fused", "skinny_gemm_m64", "skinny_gemm_rows", "gemm_64x64_splitk"],
}


def synth_batch(B, S, dev, seed, lang=False, ingest="fp32", store_frames=0):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    def u8img(h):
        if ingest == "u8" and store_frames:      # HBM-resident frame store (F,H,W,C): the step's windows are gathered by index (hulc_batch::window_start)
            return torch.cat([torch.randint(0, 256, (min(1024, store_frames - f0), h, h, 3), device=dev, generator=g, dtype=torch.int32).to(torch.uint8) for f0 in range(0, store_frames, 1024)]).contiguous()
        if ingest == "u8":      # dataset layout: uint8 (B,S,H,W,C); scale / normalise / RandomShiftsAug run inside conv1's load path
            return torch.randint(0, 256, (B, S, h, h, 3), device=dev, generator=g, dtype=torch.int32).to(torch.uint8).contiguous()
        u = torch.randint(0, 256, (B, S, 3, h, h), device=dev, generator=g, dtype=torch.int32).float()
        return ((u / 255.0 - 0.5) / 0.5).contiguous()
    act = torch.rand(B, S, 7, device=dev, generator=g) * 2 - 1
    e = torch.rand(B, S, 7, device=dev, generator=g)
    act = torch.where(e < 0.025, -torch.ones_like(act), torch.where(e > 0.975, torch.ones_like(act), act))
    act[..., 6] = torch.where(torch.rand(B, S, device=dev, generator=g) < 0.5, -1.0, 1.0)
    ro = torch.randn(B, S, 15, device=dev, generator=g) * 0.3
    ro[..., 3:6] = torch.rand(B, S, 3, device=dev, generator=g) * 2 - 1
    mb = dict(rgb_static=u8img(200), rgb_gripper=u8img(84), actions=act.contiguous(), robot_obs=ro.contiguous())
    if ingest == "u8":
        mb.update(shift_static=torch.randint(0, 21, (B * S, 2), device=dev, generator=g, dtype=torch.int32), pad_static=10,
                  shift_gripper=torch.randint(0, 9, (B * S, 2), device=dev, generator=g, dtype=torch.int32), pad_gripper=4)
    if ingest == "u8" and store_frames:
        # 64 pre-drawn vectors of B random window starts (episode boundaries are the datamodule's business: any start in [0, F - S] is a valid read)
        mb["window_starts"] = torch.randint(0, store_frames - S + 1, (64, B), device=dev, generator=g, dtype=torch.int64)
        mb["window_start"] = mb["window_starts"][0].contiguous()
    if lang:
        l = torch.randn(B, 384, device=dev, generator=g)
        mb["lang"] = (l / l.norm(dim=-1, keepdim=True)).contiguous()
        mb["aux_rows"] = np.arange(B, dtype=np.int32)
    return mb


def cpu_baseline(S, budget_s=6.0, kind="hulc", rnn_type="rnn"):
    """The reference's CPU path, restated, timed on this host's cores on a bounded sample of the same workload (kind "port": the reference
    itself never travels to the GPU box).  For the HULC / GCBC kinds the port is oracle/hulc_torch_port.py — the step on torch's CPU library
    kernels with autograd + torch.optim.Adam, i.e. the reference's arithmetic on the very ATen / mkldnn kernels its own CPU path runs,
    pinned against the reference fixtures (tests/test_oracle_golden.py::test_torch_port_matches_reference); the numpy oracle (the parity
    checker) is timed next to it for one setting.  B = 8 windows per step, one WARM-UP step discarded per setting, thread count swept
    (1 / 8 / 32 / min(all cores, 64)) and the best setting reported.  `reference_anchor` carries the unmodified reference's own figure from the
    survey container."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import hulc_oracle as O
    from hulc_amd.utils import synthetic
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count() or 1
    dims = spec.ModelDims(kind=kind, max_window=max(32, S), use_clip=False, rnn_type=rnn_type)
    anchor = dict(value=[8.0, 12.2], unit="windows/s", cores=8, kind="reference",
                  sample="unmodified reference (torch 2.10 CPU, fp32), HULC vision-only S=32, B=8 / B=16, 8 vCPU Xeon 2.1 GHz in the survey "
                         "container (BASELINE.md §2); not re-measurable on the GPU box")

    def timed(one, Bc):
        one(0)                                        # warm-up: page-in, thread pool start, first touch of every buffer
        t0 = time.time()
        n = 0
        while True:
            one(n + 1)
            n += 1
            if time.time() - t0 > budget_s or n >= 3:
                break
        return Bc * n / (time.time() - t0), n

    def numpy_point(nt, Bc):
        from threadpoolctl import threadpool_limits
        P = spec.init_all(dims, seed=0)
        batch = synthetic.make_batch(Bc, 0, S, seed=0)
        if kind == "mcil":
            for mb in batch.values():
                mb["plan_eps"] = np.random.default_rng(0).standard_normal((mb["actions"].shape[0], 256)).astype(np.float32)
        st = {}

        def one(i):
            _, G = O.training_step(P, dims, batch)
            O.adam_step(P, G, st, i + 1)
        with threadpool_limits(limits=nt):
            return timed(one, Bc)

    sweep = []
    # thread sweep capped at 64: on the 256-core GPU host the all-cores point of the torch port ran 0.065 windows/s (123 s for ONE step of
    # 8 windows: intra-op oversubscription on a step this small) and would alone eat the bench's time budget; 8-32 threads is where both ports peak
    settings = sorted({t for t in (1, 8, 32, min(cores, 64)) if t <= cores})
    if kind in ("hulc", "gcbc"):
        import hulc_torch_port as TP
        nt0 = torch.get_num_threads()
        for nt in settings:
            Bc = 2 if nt == 1 else 8                  # the 1-thread point on a quarter of the windows (it would eat the whole budget otherwise)
            torch.set_num_threads(nt)
            stp = TP.Stepper(spec.init_all(dims, seed=0), kind=kind, use_clip=False)
            batch = synthetic.make_batch(Bc, 0, S, seed=0)
            wps, n = timed(lambda i: stp.step(batch), Bc)
            sweep.append(dict(threads=nt, windows_per_s=round(wps, 3), steps=n, batch=Bc))
        torch.set_num_threads(nt0)
        wps_np, n_np = numpy_point(min(8, cores), 8)
        best = max(sweep, key=lambda r: r["windows_per_s"])
        return dict(value=best["windows_per_s"], unit="windows/s", cores=best["threads"], kind="port",
                    sample=f"{best['steps']} warm step(s) of B={best['batch']} S={S} vis windows after one discarded warm-up step; oracle/hulc_torch_port.py = the step on "
                           f"torch's CPU library kernels (ATen/mkldnn conv, addmm, RNN) with autograd + torch.optim.Adam, fp32; best of the thread sweep "
                           f"{[r['threads'] for r in sweep]} on a {cores}-core host",
                    sweep=sweep, host_cores=cores, numpy_oracle=dict(value=round(wps_np, 3), threads=min(8, cores), steps=n_np, batch=8), reference_anchor=anchor)
    for nt in settings:                               # mcil kinds: the numpy oracle is the only port
        Bc = 2 if nt == 1 else 8
        wps, n = numpy_point(nt, Bc)
        sweep.append(dict(threads=nt, windows_per_s=round(wps, 3), steps=n, batch=Bc))
    best = max(sweep, key=lambda r: r["windows_per_s"])
    return dict(value=best["windows_per_s"], unit="windows/s", cores=best["threads"], kind="port",
                sample=f"{best['steps']} warm step(s) of B={best['batch']} S={S} vis windows after one discarded warm-up step, numpy oracle fp32 (fwd+bwd+Adam), "
                       f"best of the BLAS thread sweep {[r['threads'] for r in sweep]} on a {cores}-core host",
                sweep=sweep, host_cores=cores, reference_anchor=anchor)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="GPUs of this node = ranks of the job (default: WORLD_SIZE, else 1).  Given without a launcher "
                                                       "(no WORLD_SIZE in the environment) and > 1, bench.py re-launches itself under torch.distributed.run with that many ranks")
    ap.add_argument("--steps", type=int, default=100)    # ~0.5 s of timed steps: one host-side stall of a few ms no longer moves the mean
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64, help="windows per GPU")
    ap.add_argument("--seq", type=int, default=32)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"],
                    help="bf16: the headline (BASELINE configs[1]); fp16: the reference's `precision: 16` with the on-device dynamic loss scaler (config 5)")
    ap.add_argument("--lang", type=int, default=0, help="1: 32 vis + 32 lang per GPU with CLIP aux loss (config 3)")
    ap.add_argument("--ingest", default="fp32", choices=["fp32", "u8"],
                    help="fp32: the reference's boundary (transformed fp32 NCHW frames, the headline); u8: uint8 HWC dataset frames, "
                         "scale/normalise/RandomShiftsAug fused into conv1 (SURVEY §8(f) row 1)")
    ap.add_argument("--model", default="hulc", choices=["hulc", "mcil", "mcil_gru"],
                    help="hulc: the headline configuration; mcil: conf/model/mcil.yaml (BiRNN plan recognition, continuous plan, no CLIP loss); "
                         "mcil_gru: the same with plan_recognition.rnn_type=nn.GRU (BASELINE config 4)")
    ap.add_argument("--preroll", type=int, default=300, help="untimed steps before the warm-up (≈1.5 s: clock ramp of an idle GPU)")
    ap.add_argument("--persist", type=int, default=1, help="hulc_set_option persistent_rnn: 1 = each 2048-wide recurrence as one persistent launch (default), 0 = one launch per time step")
    ap.add_argument("--pair", type=int, default=1, help="with --lang 1: both modalities as ONE 2B-window pass (hulc_forward_loss_pair); 0 = one pass per modality like the reference")
    ap.add_argument("--bucket", default="fp32", choices=["fp32", "bf16", "fp16"],
                    help="N > 1: wire format of the gradient all-reduce buckets (fp32 = the reference's; the engine's 16-bit type halves the bytes per link)")
    ap.add_argument("--h2d", type=int, default=0, help="with --ingest u8: 1 = every step's uint8 frames come from PINNED HOST memory (async H2D on a copy stream into a "
                                                      "double buffer, overlapped with the previous step) — the PCIe-inclusive row SURVEY §8(d) asks for; never the headline `value`")
    ap.add_argument("--store", type=int, default=0, metavar="F", help="with --ingest u8: F > 0 = the frames live in a device-resident uint8 STORE of F frames per camera "
                                                                     "and every step draws B random windows from it by index (hulc_batch::window_start): no materialised batch, no per-step H2D")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="hulc_set_option(NAME, VALUE) before the run: same-box A/B of a library switch, e.g. adam_fused_transposes=0")
    ap.add_argument("--timer-stride", type=int, default=4, help="the live class timers record their HIP events on every N-th step of the timed region (1 = every step)")
    ap.add_argument("--live-timers", default="on", choices=["on", "fenced", "off"],
                    help="A/B of the measurement itself (never a reported configuration): 'on' (default) = the dominant group's kernel classes are timed by HIP events "
                         "inside the timed region (timing-only events, hipEventDisableSystemFence); 'fenced' = the same with default events (rounds <= 4); 'off' = no "
                         "events in the timed region, `roofline` comes from the survey pass")
    ap.add_argument("--force-comm", type=int, default=0, help="1-GPU REHEARSAL of the N > 1 code path (never a reported number): a 1-rank torch.distributed group and a "
                                                              "1-rank library RCCL communicator, so that the self-check, hulc_backward_allreduce in the timed loop and the bucket "
                                                              "timeline run before the first multi-GPU box meets them (tests/test_gpu_dp.py)")
    args = ap.parse_args()

    # --gpus N MEANS N: under a launcher it must equal WORLD_SIZE; without one, N > 1 re-launches this script with one rank per GPU.
    # A `python bench.py --gpus 8` that quietly ran one rank and printed n_gpus: 1 is the failure this guards against.
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus is None:
        args.gpus = int(env_world) if env_world else 1
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if env_world is not None and int(env_world) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={env_world}: refusing to report a line whose n_gpus is not what was asked for")
    if env_world is None and args.gpus > 1:
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible on this node")
        import socket
        import subprocess
        with socket.socket() as sk:                       # a free rendezvous port on the loopback interface
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or args.force_comm:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local}"))
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)

    B, S = args.batch, args.seq
    mcil = args.model in ("mcil", "mcil_gru")
    use_clip = bool(args.lang) and not mcil
    dims = spec.ModelDims(kind="mcil" if mcil else args.model, max_window=max(32, S), use_clip=use_clip, rnn_type="gru" if args.model == "mcil_gru" else "rnn")
    Bmod = B // 2 if args.lang else B
    paired = bool(args.lang) and bool(args.pair)
    eng = StepEngine(dims, B if paired else Bmod, S, dtype=args.dtype, device=str(dev), dropout_p=0.0 if mcil else 0.1, seed=42, num_classes=dims.mix_classes)
    eng.set_option("persistent_rnn", args.persist)
    for kv in args.opt:                                  # same-box A/B of a library switch (hulc_set_option); never a reported configuration
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    eng.load_numpy(spec.init_all(dims, seed=0))      # identical weights on every rank (seeded init = the DDP broadcast)
    # N > 1: the library's own RCCL communicator (hulc_backward_allreduce: reverse-forward buckets overlapped with the backward); the
    # torch.distributed group above only carries the ncclUniqueId, the barriers and the timing reduction.  HULC_DP_COMM=capi (the default)
    # makes a communicator that cannot be brought up an ERROR on every rank — a run that quietly measured torch.distributed all-reduces
    # would be Lightning-DDP-shaped, not the product; HULC_DP_COMM=torch / auto select that path explicitly.
    if world > 1:
        parallel.configure_shared_gpu(eng)              # ranks sharing a device (never the driver's layout: one rank per GPU) switch the persistent recurrences off
    lib_comm = parallel.setup_comm(eng, args.bucket) if world > 1 else False
    if args.force_comm and world == 1:
        eng.comm_init(eng.comm_unique_id(), 0, 1)
        eng.comm_bucket_dtype = args.bucket
        parallel.check_bucket_plan(eng.comm_buckets(), eng.numel)
        lib_comm = True
    if world > 1 and not lib_comm and parallel.comm_mode() == "capi":
        raise SystemExit("bench.py: the library RCCL communicator is not up and HULC_DP_COMM=capi — refusing to time the torch.distributed fallback")
    # the gate of the N > 1 line (VERDICT r5 #7 iii): the LIVE communicator's own size and rank (ncclCommCount / ncclCommUserRank through hulc_comm_size),
    # not an echo of what this script passed in — a line with n_gpus: N is printed only if RCCL itself says N ranks carry the gradients
    rccl_rank, rccl_ranks = (eng.comm_size() if lib_comm else (None, None))
    if lib_comm and (rccl_ranks != world or rccl_rank != rank):
        raise SystemExit(f"bench.py rank {rank}: the library communicator reports rank {rccl_rank} of {rccl_ranks}, the job is rank {rank} of {world} — refusing to print a line")
    if args.store and (args.ingest != "u8" or args.h2d or args.store < S):
        raise SystemExit("--store F needs --ingest u8, no --h2d, and F >= seq")
    mods = [("vis", synth_batch(Bmod, S, dev, 1000 * rank + 1, False, args.ingest, args.store))]
    if args.lang:
        mods.append(("lang", synth_batch(Bmod, S, dev, 1000 * rank + 2, True, args.ingest, args.store)))
    nmod = len(mods)

    # --h2d 1: the frames of every step cross PCIe.  Pinned host copies of the uint8 (B,S,H,W,C) frames; two device buffers per camera; the copy
    # of step i+1's frames is issued on a copy stream when step i starts (it may not start earlier: the buffer it overwrites is read by step i-1's
    # conv1 weight gradient, the last kernels of that step) and step i+1 waits for its event — the transfer hides under step i's compute as far as
    # PCIe allows (289 MB per step at B = 64: ~5 ms at 55-60 GB/s, longer than the step itself).
    h2d = None
    if args.h2d:
        if args.ingest != "u8" or args.lang:
            raise SystemExit("--h2d 1 needs --ingest u8 (vision-only)")
        mb0 = mods[0][1]
        h2d = dict(host={k: mb0[k].cpu().pin_memory() for k in ("rgb_static", "rgb_gripper")},
                   dev=[{k: torch.empty_like(mb0[k]) for k in ("rgb_static", "rgb_gripper")} for _ in range(2)],
                   stream=torch.cuda.Stream(device=dev), ready=[torch.cuda.Event(), torch.cuda.Event()], free=[torch.cuda.Event(), torch.cuda.Event()], n=0)

        def h2d_issue(slot):
            with torch.cuda.stream(h2d["stream"]):
                h2d["stream"].wait_event(h2d["free"][slot])          # the step that last read this buffer has finished
                for k in ("rgb_static", "rgb_gripper"):
                    h2d["dev"][slot][k].copy_(h2d["host"][k], non_blocking=True)
                h2d["ready"][slot].record(h2d["stream"])
        for sl in range(2):
            h2d["free"][sl].record(torch.cuda.current_stream(dev))
        h2d_issue(0)

    def step(i):
        if h2d is not None:
            slot = h2d["n"] % 2
            h2d["n"] += 1
            h2d_issue(1 - slot)                                       # next step's frames: in flight while this step computes
            torch.cuda.current_stream(dev).wait_event(h2d["ready"][slot])
            mods[0] = (mods[0][0], dict(mods[0][1], **h2d["dev"][slot]))
            _step(i)
            h2d["free"][slot].record(torch.cuda.current_stream(dev))
            return
        _step(i)

    def last_backward():
        if lib_comm:
            eng.backward_allreduce(args.bucket)      # the library's bucketed RCCL all-reduce, overlapped with the backward
        elif world > 1:
            parallel.backward_overlapped(eng)        # torch.distributed fallback (HULC_DP_COMM=auto / torch)
        else:
            eng.backward()

    def _step(i):
        if args.store:                                   # this step's windows: another pre-drawn vector of starts (a device pointer swap, no copy)
            for _, mb in mods:
                mb["window_start"] = mb["window_starts"][i % 64]
        eng.zero_grads()
        if paired:
            eng.forward_loss_pair(mods[0][1], mods[1][1], 0.5, 3.0, step=i, sync_losses=False)
            last_backward()
            eng.adam_step(lr=2e-4, grad_scale=1.0 / world)
            return
        for k, (name, mb) in enumerate(mods):
            eng.forward_loss(mb, name == "lang", 1.0 / nmod, 3.0, step=i, sync_losses=False)
            if k == nmod - 1:
                last_backward()
            else:
                eng.backward()
        eng.adam_step(lr=2e-4, grad_scale=1.0 / world)   # DP mean folded into the Adam kernel

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    # N > 1, before anything is timed: the collective checks ITSELF on this job's real gradients — one step's library-RCCL bucketed SUM
    # (hulc_backward_allreduce) against the same local gradients through ONE flat torch.distributed all-reduce.  The first multi-GPU box
    # this code meets is the driver's; the numbers it reports carry their own evidence (JSON: allreduce.selfcheck).
    selfcheck = None
    if lib_comm:
        import torch.distributed as dist
        eng.zero_grads()
        if paired:
            eng.forward_loss_pair(mods[0][1], mods[1][1], 0.5, 3.0, step=0, sync_losses=False)
        else:
            for k, (name, mb) in enumerate(mods):
                eng.forward_loss(mb, name == "lang", 1.0 / nmod, 3.0, step=0, sync_losses=False)
                if k < nmod - 1:
                    eng.backward()
        eng.backward()
        torch.cuda.synchronize()
        g_loc = eng.flat_grads.clone()
        g_ref = g_loc.clone()
        dist.all_reduce(g_ref, op=dist.ReduceOp.SUM)
        eng.allreduce_grads(args.bucket)                  # the library communicator on the same local gradients (whole-buffer form)
        torch.cuda.synchronize()
        nref = g_ref.double().norm().item()
        rel_flat = (eng.flat_grads.double() - g_ref.double()).norm().item() / max(nref, 1e-30)
        # and the bucketed, overlapped form on a fresh backward of the same step.  The categorical plan sample of the first pass is injected
        # (16-bit engines reorder atomic sums run to run: a logit moving by an ulp could flip a draw and change the gradient by percents);
        # dropout masks and the mcil eps are counter-based on (seed, step) and repeat by themselves
        mods2 = mods
        if not mcil and (paired or nmod == 1):
            pidx = torch.from_numpy(eng.plan_idx(B if paired else Bmod)).to(dev)
            mods2 = [(name, dict(mb, plan_idx=pidx[k * Bmod:(k + 1) * Bmod].contiguous())) for k, (name, mb) in enumerate(mods)]
        eng.zero_grads()
        if paired:
            eng.forward_loss_pair(mods2[0][1], mods2[1][1], 0.5, 3.0, step=0, sync_losses=False)
        else:
            for k, (name, mb) in enumerate(mods2):
                eng.forward_loss(mb, name == "lang", 1.0 / nmod, 3.0, step=0, sync_losses=False)
                if k < nmod - 1:
                    eng.backward()
        eng.backward_allreduce(args.bucket)
        torch.cuda.synchronize()
        rel_bkt = (eng.flat_grads.double() - g_ref.double()).norm().item() / max(nref, 1e-30)
        differs = (g_loc.double() - g_ref.double()).norm().item() / max(nref, 1e-30)
        # 16-bit engines: two evaluations of the same backward differ by their atomics' summation order, amplified through ReLU / rounding
        # boundaries (measured 5.5e-3 at B = 16 vis + lang in the 1-GPU rehearsal); a missing, doubled or early bucket is an O(0.3) error
        tol = 1e-6 if (args.dtype == "fp32" and args.bucket == "fp32") else 2e-2
        if mods2 is mods and not mcil:
            tol = 0.2                                     # one pass per modality: the first modality's draw is not injected
        selfcheck = dict(rel_l2_whole_buffer=rel_flat, rel_l2_bucketed_overlapped=rel_bkt, local_vs_sum=differs, tolerance=tol,
                         reference="one flat torch.distributed SUM all-reduce of the same local gradients")
        passed = rel_flat <= (2e-2 if args.bucket != "fp32" else 1e-6) and rel_bkt <= tol and (world == 1 or differs > 1e-4)
        # A collective that is WRONG (a missing, doubled or early bucket; a rank left out) is an O(0.3) error or leaves the buffer at its local value:
        # that aborts the run on every rank.  A miss of the noise tolerance alone (the 16-bit engines' run-to-run atomics order) is REPORTED — the line is
        # printed with selfcheck.passed = false, so an 8-GPU run is not lost to a tolerance that no multi-GPU box ever calibrated.
        wrong = rel_flat > 0.1 or rel_bkt > (0.1 if tol < 0.1 else 0.5) or (world > 1 and differs <= 1e-4) or not np.isfinite(rel_flat + rel_bkt)
        selfcheck["passed"] = bool(passed)
        ok = torch.tensor([0 if wrong else 1, 1 if passed else 0], device=dev, dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        selfcheck["passed_on_every_rank"] = bool(int(ok[1].item()) == 1)
        if int(ok[0].item()) != 1:
            raise SystemExit(f"bench.py rank {rank}: library RCCL all-reduce self-check FAILED: {selfcheck}")

    # pre-roll (untimed, before the W warm-up steps): a GPU that has just been idle needs a few hundred ms of load before its clocks
    # settle — the first bench process on a fresh box measured 4.89 ms/step against 4.61 for every later one with warm-up alone
    for i in range(args.preroll):           # a fixed count, identical on every rank (the steps contain collectives)
        step(i)
    torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)
    # survey pass (untimed): HIP events around every kernel class -> which class dominates the step
    eng.timers_enable(True)
    eng.timers_read(reset=True)
    for i in range(2):
        step(args.warmup + i)
    survey = eng.timers_read(reset=True)

    def group_sum(tm, classes):
        rows = [tm[c] for c in classes if c in tm]
        return dict(ms=sum(r["ms"] for r in rows), flops=sum(r["flops"] for r in rows), bytes=sum(r["bytes"] for r in rows),
                    launches=sum(r["launches"] for r in rows), classes=[c for c in classes if c in tm], bounds=sorted({r["bound"] for r in rows})) if rows else None
    groups = {g: group_sum(survey, cl) for g, cl in MFMA_GROUPS.items()}
    groups = {g: v for g, v in groups.items() if v}
    dom_group = max(groups.items(), key=lambda kv: kv[1]["ms"])[0] if groups else ""
    # timed region: events only around the classes of the dominant GROUP (on the engine's stream), so the timers do not perturb the rest of the step
    if args.live_timers == "fenced":
        eng.set_option("timer_event_fence", 1)
    # The live class timers are themselves a load on the stream: one timing-only event record costs ~1.8 us of stream time (a default, system-fenced event
    # 3 us; profiles/r05_live_timer_cost.txt), 20 records per step around the dominant group's 10 launches.  They stay inside the timed region, on every
    # `--timer-stride`-th timed step (default 4: 25 timed samples of each launch per 100 steps), so that the measurement moves the measured step by ~0.3 %.
    tfilter = ",".join(groups[dom_group]["classes"]) if dom_group else ""
    stride = max(1, args.timer_stride)
    sampled = [i for i in range(args.steps) if i % stride == 0] if args.live_timers != "off" else []
    eng.timers_enable(False)
    sc0 = eng.scaler_state() if args.dtype == "fp16" else None
    # one event per step boundary on the engine's stream (= torch's current stream): per-step device times for median / p10 / p90 next to the
    # wall-clock mean the contract's `ms_per_step` is (an event record costs no synchronisation and ~1 us of stream time)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    barrier()
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(args.steps):
        if sampled and stride > 1 or i == 0:
            eng.timers_enable(bool(sampled) and i % stride == 0, tfilter)      # a host-side flag + filter string: no stream work
        step(args.warmup + i)
        evs[i + 1].record()
    barrier()
    dt = time.perf_counter() - t0
    per_step = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps)])
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms = dt / args.steps * 1e3
    wps = B * world / (dt / args.steps)
    loss = eng._loss_dev.cpu().numpy().tolist()

    timers = eng.timers_read(reset=True) if args.live_timers != "off" else survey
    eng.timers_enable(False)
    # N > 1: one more (untimed) step with events around every bucket's collective: when each bucket was issued / finished relative to the END of
    # the backward on the engine stream (negative = hidden under the backward), per rank 0
    comm_tl = None
    if lib_comm:
        eng.set_option("comm_timing", 1)
        step(args.warmup + args.steps)
        comm_tl = eng.comm_timeline()
        eng.set_option("comm_timing", 0)

    tsteps = len(sampled) if sampled else 2      # the timed steps that carried events; 'off': the survey pass's two steps

    def roofline(tm):
        """The dominant GEMM group (all of its kernel classes, timed live by HIP events inside the timed region); achieved = the group's
        algorithmic FLOPs (or bytes) / its measured device time.  `classes` breaks it down per kernel class."""
        if not tm or not dom_group:
            return None
        t = group_sum(tm, MFMA_GROUPS[dom_group])
        # Which roof binds is decided by the group's ARITHMETIC INTENSITY (algorithmic FLOPs / algorithmic bytes) against the machine's ridge point
        # (dense MFMA peak / HBM peak = 312 FLOP/B in bf16): below the ridge the time floor bytes / 8 TB/s is the larger of the two floors.  conv2 / conv3 with
        # 32 - 64 channels of 16-bit activations sit at 157 FLOP/B (conv2: 71 GFLOP over 453 MB per pass, conv3: 66.6 over 253 MB): the HBM roof binds, although
        # every launch is a GEMM on the matrix cores; `other_roof` carries the fraction of the roof that does not bind.
        name = dom_group
        sec = t["ms"] * 1e-3
        peak_tf = MFMA_BF16_PEAK_TFLOPS if args.dtype in ("bf16", "fp16") else 157.3
        ai, ridge = t["flops"] / max(t["bytes"], 1.0), peak_tf * 1e12 / (HBM_PEAK_GBS * 1e9)
        t["bound"] = "mfma" if ai >= ridge else "hbm"
        roofs = {"mfma": (t["flops"] / sec / 1e12, peak_tf, "TFLOP/s"), "hbm": (t["bytes"] / sec / 1e9, HBM_PEAK_GBS, "GB/s")}
        ach, peak, unit = roofs[t["bound"]]
        ob = "hbm" if t["bound"] == "mfma" else "mfma"
        other = {"bound": ob, "achieved": round(roofs[ob][0], 2), "peak": roofs[ob][1], "unit": roofs[ob][2], "frac": round(roofs[ob][0] / roofs[ob][1], 4)}
        # HBM bytes per launch of this class from the PMC counters (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc passes, corrected as
        # MI355X_MICROARCH.md §HBM prescribes).  Counters cannot be collected from inside this process: the value is the one
        # tools/refresh_profiles.sh measured for the SAME build and command and committed as profiles/rNN_pmc_traffic.json (newest round);
        # `traffic_source` names the file.  null when no such file exists.
        traffic, traffic_source = None, None
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
        if cands and args.dtype == "bf16" and not args.lang and args.model == "hulc" and S == 32:
            try:
                tj = json.load(open(cands[-1]))
                # bytes per launch, averaged over the group's launches (PMC bytes per launch of each class x its launches per step)
                tb = [(tj.get(c), tm[c]["launches"]) for c in t["classes"]]
                traffic = None if any(b is None for b, _ in tb) else sum(b * n for b, n in tb) / max(1, sum(n for _, n in tb))
                traffic_source = os.path.relpath(cands[-1], ROOT)
            except Exception:
                traffic = None
        per_class = {c: {"ms_per_step": round(tm[c]["ms"] / tsteps, 4), "launches_per_step": tm[c]["launches"] / tsteps,
                         "achieved": round((tm[c]["flops"] / 1e12 if t["bound"] == "mfma" else tm[c]["bytes"] / 1e9) / max(tm[c]["ms"] * 1e-3, 1e-12), 2)} for c in t["classes"]}
        # the two fractions north_star names next to the dominant kernel's: the decoder's GEMM groups against the dense MFMA peak
        # (survey pass: HIP events around every class) and the whole step against the HBM peak with SURVEY 8(d)'s algorithmic bytes
        def mf(g):
            v = groups.get(g)
            return None if not v else round(v["flops"] / max(v["ms"], 1e-9) / 1e9 / peak_tf, 4)
        sr = None if mcil else step_roofline()
        # the same group priced with the PMC traffic instead of the algorithmic bytes: HBM bytes actually moved per launch x launches / time / peak
        frac_by_traffic = None if traffic is None else round(traffic * t["launches"] / sec / 1e9 / HBM_PEAK_GBS, 4)
        return {"kernel": name + " = " + " + ".join(t["classes"]), "classes": per_class, "frac_by_traffic": frac_by_traffic,
                "decoder_gemm_mfma_frac": {"rnn_batched": mf("rnn_batched"), "rnn_recurrent": mf("rnn_recurrent"), "transformer_mlp": mf("transformer_mlp"), "peak_tflops": peak_tf},
                "step_hbm_frac": None if sr is None else sr["hbm_frac"], "step_mfma_frac": None if sr is None else sr["mfma_frac"], "bound": t["bound"], "achieved": round(ach, 2), "peak": peak, "unit": unit, "frac": round(ach / peak, 4),
                "arithmetic_intensity_flop_per_byte": round(ai, 1), "ridge_flop_per_byte": round(ridge, 1), "other_roof": other,
                "traffic": traffic, "traffic_source": traffic_source, "launches_per_step": t["launches"] / tsteps, "avg_launch_us": round(t["ms"] * 1e3 / max(1, t["launches"]), 2),
                "ms_per_step": round(t["ms"] / tsteps, 4), "event_timed_steps": tsteps, "of_timed_steps": args.steps,
                "per_launch": {"algorithmic_flops": t["flops"] / max(1, t["launches"]), "algorithmic_bytes": t["bytes"] / max(1, t["launches"])}}

    def step_roofline():
        """Whole step against both ceilings with SURVEY §8(d)'s ALGORITHMIC work (every tensor moved the minimum number of times without
        inter-layer fusion): per window (S = 32) input frames 4 517 376 elements read twice (conv1 forward + weight gradient), conv activations
        5 121 024 elements x 5 touches, post-encoder activations ~0.3 M x 5, in the engine's element size (frames: 4 B at the fp32 boundary,
        1 B with uint8 ingest); per step 470 M fp32 elements of weight / gradient / optimizer traffic (independent of B)."""
        es = 4 if args.dtype == "fp32" else 2
        fe = 1 if args.ingest == "u8" else 4
        r = S / 32.0
        bytes_window = 4517376 * r * fe * 2 + (5121024 + 300000) * r * es * 5
        bytes_step = bytes_window * B + 470e6 * 4
        flops_step = FLOP_PER_WINDOW_S32 * r * B
        t = float(np.median(per_step)) * 1e-3
        peak_f = (MFMA_BF16_PEAK_TFLOPS if args.dtype != "fp32" else 157.3) * 1e12
        t_hbm, t_mfma = bytes_step / (HBM_PEAK_GBS * 1e9), flops_step / peak_f
        return {"algorithmic_bytes_per_step": bytes_step, "algorithmic_flops_per_step": flops_step, "hbm_frac": round(t_hbm / t, 4), "mfma_frac": round(t_mfma / t, 4),
                "binding": "hbm" if t_hbm > t_mfma else "mfma", "ceiling_windows_per_s": round(B / max(t_hbm, t_mfma), 1),
                "achieved_over_min_ceiling": round(max(t_hbm, t_mfma) / t, 4), "at": "median step time, per GPU"}

    # RCCL prints its version banner through C stdio, which would otherwise be flushed at process exit — AFTER the JSON line; the contract is that
    # the JSON line is the last thing rank 0 prints
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if rank == 0:
        rl = roofline(timers)
        kernel_classes = {k: {"ms_per_step": round(v["ms"] / args.steps, 4), "launches_per_step": v["launches"] / args.steps,
                              "tflops": round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 1), "gbs": round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1)}
                          for k, v in sorted(survey.items(), key=lambda kv: -kv[1]["ms"])}
        for v in kernel_classes.values():
            v["ms_per_step"] = round(v["ms_per_step"] * args.steps / 2, 4); v["launches_per_step"] = v["launches_per_step"] * args.steps / 2
        out = {
            "metric": "trajectory-windows/sec (seq_len=%d, bs=%d/GPU)" % (S, B), "value": round(wps, 2), "unit": "windows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "%s training step, %s, B=%d windows/GPU, seq_len=%d, 200x200 static + 84x84 gripper %s frames, "
                                   "fwd+loss+bwd+%sAdam, dropout %s" % (("HULC (model=mcil: Bi%s plan recognition, continuous plan)" % ("GRU" if args.model == "mcil_gru" else "RNN")) if mcil else "HULC",
                                                                        ("32 vis + 32 lang" + ("" if mcil else " + CLIP aux") + (" as one paired pass" if paired else ", one pass per modality")) if args.lang else "vision goal only (use_clip_auxiliary_loss=false)",
                                                                        B, S, ("uint8 HWC (scale+normalise+RandomShiftsAug fused into conv1)" + (", frames copied from PINNED HOST memory every step (async H2D, double-buffered)" if args.h2d else "") + ((", windows gathered by index from a device-resident store of %d frames per camera (new random windows every step)" % args.store) if args.store else "")) if args.ingest == "u8" else "fp32 NCHW",
                                                                        "RCCL all-reduce+" if world > 1 else "", "0.0" if mcil else "0.1"),
                       "global_batch": B * world, "seq_len": S, "parallelism": "dp%d" % world},
            "last_losses": {"total_mod": loss[0], "kl": loss[1], "action": loss[2], "clip": loss[3]},
            "model_flops_per_window": None if mcil else FLOP_PER_WINDOW_S32 * S / 32.0,      # SURVEY §8(d) counts the headline model only
            "step_tflops": None if mcil else round(wps / world * FLOP_PER_WINDOW_S32 * S / 32.0 / 1e12, 2),
            # per-step device times (HIP events at the step boundaries): the mean above is wall clock incl. host stalls; SURVEY §8(d) asks for the median
            "step_ms": {"median": round(float(np.median(per_step)), 4), "p10": round(float(np.percentile(per_step, 10)), 4), "p90": round(float(np.percentile(per_step, 90)), 4),
                        "min": round(float(per_step.min()), 4), "max": round(float(per_step.max()), 4), "windows_per_s_at_median": round(B * world / (float(np.median(per_step)) * 1e-3), 1)},
            "roofline_step": None if mcil else step_roofline(),
            "roofline": rl,
            # per GEMM group (survey pass: HIP events around every class, 2 steps): algorithmic TFLOP/s and the fraction of the dense MFMA peak
            "mfma_groups": {g: {"ms_per_step": round(v["ms"] / 2, 4), "launches_per_step": v["launches"] / 2, "tflops": round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 1),
                                "mfma_frac": round(v["flops"] / max(v["ms"], 1e-9) / 1e9 / (MFMA_BF16_PEAK_TFLOPS if args.dtype != "fp32" else 157.3), 4),
                                "gbs": round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1), "classes": v["classes"]} for g, v in sorted(groups.items(), key=lambda kv: -kv[1]["ms"])},
            "allreduce": None if (world == 1 and not lib_comm) else ({"path": "libhulc_hip RCCL (hulc_backward_allreduce)" + (" — 1-rank REHEARSAL (--force-comm)" if world == 1 else ""), "rccl_ranks": rccl_ranks, "rccl_ranks_source": "ncclCommCount of the live communicator (hulc_comm_size)", "bucket_dtype": args.bucket, "buckets": eng.comm_buckets(),
                                                  "bucket_bytes": [(hi - lo) * (4 if args.bucket == "fp32" else 2) for lo, hi in eng.comm_buckets()],
                                                  "selfcheck": selfcheck, "timeline": comm_tl, **eng.comm_stats()} if lib_comm else
                                                 {"path": "torch.distributed nccl (HULC_DP_COMM=%s)" % parallel.comm_mode(), "bucket_dtype": "fp32"}),
            "kernel_classes": kernel_classes,
            # fp16: GradScaler state after the timed steps; skipped_in_timed_region counts optimizer steps the scaler skipped (inf/nan) inside it
            "loss_scaler": None if sc0 is None else dict(eng.scaler_state(), skipped_in_timed_region=eng.scaler_state()["skipped_steps"] - sc0["skipped_steps"]),
            "cpu_baseline": None if (args.no_cpu_baseline or world > 1) else cpu_baseline(S, kind=dims.kind, rnn_type=dims.rnn_type),
        }
        print(json.dumps(out), flush=True)
    if world > 1 or args.force_comm:
        import torch.distributed as dist
        dist.destroy_process_group()
    sys.stdout.flush()
    if world > 1 or args.force_comm:
        os._exit(0)          # nothing after the JSON line: RCCL / torch teardown may print on exit


if __name__ == "__main__":
    main()
