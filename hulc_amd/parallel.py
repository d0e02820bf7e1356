"""Data parallelism for the HULC step: one process per GPU, gradients summed by ONE flat all-reduce over RCCL/xGMI.

Replaces Lightning's DDPStrategy (reference hulc/training.py:64-69: DDP(find_unused_parameters=False), mean of
per-rank gradients).  Because parameters and gradients live in one contiguous fp32 buffer (hulc_amd.spec.layout),
the whole 188 MB gradient is a single collective (or a few large buckets) — sized for xGMI's per-link bandwidth rather
than NCCL's 25 MB default buckets; the 1/world factor is folded into the Adam kernel (grad_scale), so no extra pass.
Unused parameters (GCBC's plan_proposal / fc_state) simply contribute zeros on every rank.

`backend="nccl"` is RCCL on ROCm; the same code runs on CPU tensors with `gloo` (tests/test_ddp_gloo.py).
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> tuple:
    """Returns (rank, world, local_rank); initialises torch.distributed when WORLD_SIZE > 1."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device(f"cuda:{local}")
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def allreduce_sum_(flat: torch.Tensor, bucket_elems: int = 0) -> torch.Tensor:
    """In-place SUM all-reduce of the flat gradient buffer.  bucket_elems=0 -> one collective; otherwise contiguous
    buckets of that many elements issued back to back (async), which lets RCCL pipeline reduce-scatter/all-gather phases."""
    if world_size() == 1:
        return flat
    if bucket_elems <= 0 or bucket_elems >= flat.numel():
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        return flat
    works = []
    for off in range(0, flat.numel(), bucket_elems):
        works.append(dist.all_reduce(flat[off:off + bucket_elems], op=dist.ReduceOp.SUM, async_op=True))
    for w in works:
        w.wait()
    return flat


def broadcast_(flat: torch.Tensor, src: int = 0) -> torch.Tensor:
    """DDP's initial parameter broadcast from rank 0."""
    if world_size() > 1:
        dist.broadcast(flat, src=src)
    return flat


def mean_scalar(x: float, device=None) -> float:
    """`self.log(..., sync_dist=True)` semantics for a scalar metric (hulc.py:512-532)."""
    if world_size() == 1:
        return float(x)
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item()) / world_size()


def mean_metrics(sums: Dict[str, float], counts: Dict[str, int], device=None) -> Dict[str, float]:
    """Epoch means of per-batch metrics over ALL ranks' batches.  The key set may differ between ranks (`lang_gt/*` is only logged by batches
    with masked language rows, hulc.py:988-989): the ranks first agree on the union of their keys (one all_gather_object), then ONE
    all-reduce carries a (sum, count) pair per key in sorted order — a rank that never logged a key contributes 0 / 0 instead of issuing
    fewer collectives (a hang) or pairing its k-th metric with another rank's different k-th metric."""
    if world_size() == 1:
        return {k: float(v) / max(int(counts.get(k, 0)), 1) for k, v in sums.items()}
    gathered = [None] * world_size()
    dist.all_gather_object(gathered, sorted(sums))
    keys = sorted(set().union(*[set(g) for g in gathered]))
    if not keys:
        return {}
    t = torch.tensor([[float(sums.get(k, 0.0)), float(counts.get(k, 0))] for k in keys], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    t = t.cpu()
    return {k: float(t[i, 0]) / float(t[i, 1]) for i, k in enumerate(keys) if float(t[i, 1]) > 0}


BUCKET_GROUPS = (("action_decoder.", None), ("plan_proposal.",), ("plan_recognition.",), ("visual_goal.", "language_goal."), ("perceptual_encoder.",))


def bucket_schedule(layout, numel: int):
    """The bucket ranges of the flat gradient buffer in the order the backward finalises them (reverse-forward: decoder first, the
    perceptual encoders last) — the host-side mirror of Engine::bucket_plan (hulc_amd/csrc/engine.h; `hulc_comm_buckets` returns the
    library's own list, tests compare the two).  layout: name -> (offset, shape) from hulc_amd.spec.layout."""
    import numpy as np

    order = sorted(layout.items(), key=lambda kv: kv[1][0])

    def rng(prefixes):       # [first element, start of the tensor behind the group's last one): inter-tensor padding rides with the group
        lo, hi = numel, 0
        for i, (n, (off, shape)) in enumerate(order):
            if any(n.startswith(p) for p in prefixes if p):
                lo, hi = min(lo, off), max(hi, order[i + 1][1][0] if i + 1 < len(order) else numel)
        return (lo, min(hi, numel)) if hi > lo else (0, 0)
    out = []
    for g in BUCKET_GROUPS:
        lo, hi = rng(g)
        if g[-1] is None:                  # the first bucket runs to the end of the buffer (CLIP head, logit_scale)
            hi = numel
        out.append((lo, hi))
    return out


def comm_mode() -> str:
    """HULC_DP_COMM: `capi` (default) = the library's own RCCL communicator, and failing to bring it up is an ERROR on every rank;
    `auto` = try it, fall back (loudly) to torch.distributed all-reduces; `torch` = torch.distributed only."""
    m = os.environ.get("HULC_DP_COMM", "capi")
    if m not in ("capi", "auto", "torch"):
        raise ValueError(f"HULC_DP_COMM={m!r}: expected capi | auto | torch")
    return m


def _vote(ok: bool, device) -> bool:
    """True iff EVERY rank reports ok (MIN all-reduce): all ranks leave with the same answer."""
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item()) == 1


def setup_comm(engine, bucket_dtype: str = "fp32") -> bool:
    """Create the library's own RCCL communicator for `engine` (world > 1, GPU): rank 0's ncclUniqueId travels over the already
    initialised torch.distributed group, everything after that is RCCL inside libhulc_hip (hulc_backward_allreduce).

    Three phases, each closed by a vote of all ranks, so that a failure on one rank can neither split the job between two collective
    paths nor leave the healthy ranks blocked in ncclCommInitRank: (1) hulc_comm_prepare — RCCL resolved, private stream
    created: the rank-local failure modes; (2) rank 0 draws the unique id and broadcasts it; (3) hulc_comm_init.  Returns True when the
    library path is up.  Otherwise: HULC_DP_COMM=capi (default) raises on every rank; =auto prints the reason once and returns False
    (gradients then go through torch.distributed); =torch never tries."""
    mode = comm_mode()
    # (`comm_rehearsal`: an engine stand-in of the CPU tests drives the three votes over gloo — tests/test_config_and_ddp.py, 8 ranks)
    if world_size() == 1 or not (torch.cuda.is_available() or getattr(engine, "comm_rehearsal", False)) or mode == "torch":
        return False
    dev = engine.device
    err = None

    def fail(phase):
        if getattr(engine, "has_comm", False):
            engine.comm_destroy()
        msg = f"library RCCL communicator unavailable ({phase}: {err if err is not None else 'failed on another rank'})"
        if mode == "capi":
            raise RuntimeError(f"[hulc_amd] {msg}; set HULC_DP_COMM=auto (fall back to torch.distributed) or HULC_DP_COMM=torch to run without it")
        if dist.get_rank() == 0:
            print(f"[hulc_amd] {msg}; gradients go through torch.distributed", flush=True)
        return False

    try:
        engine.comm_prepare()
    except Exception as e:                 # pragma: no cover  (needs a box without RCCL)
        err = e
    if not _vote(err is None, dev):
        return fail("hulc_comm_prepare")
    box = [None]
    if dist.get_rank() == 0:
        try:
            box = [engine.comm_unique_id()]
        except Exception as e:             # pragma: no cover
            err = e
    dist.broadcast_object_list(box, src=0)
    if box[0] is None:
        return fail("ncclGetUniqueId on rank 0")
    try:
        engine.comm_init(box[0], dist.get_rank(), dist.get_world_size())
    except Exception as e:
        err = e
    if not _vote(err is None, dev):
        return fail("hulc_comm_init")
    engine.comm_bucket_dtype = bucket_dtype
    check_bucket_plan(engine.comm_buckets(), engine.numel)
    return True


def shared_device_ranks(device) -> int:
    """How many ranks of this job run on the SAME physical GPU as this one (1 = exclusive).  Ranks are matched by host name and the device's
    PCI bus id / UUID, not by LOCAL_RANK (two ranks may both have been given cuda:0)."""
    if world_size() == 1 or not torch.cuda.is_available():
        return 1
    import socket
    ident = device_identity(device)
    if ident is None:
        # unknown identity: do NOT conclude that the ranks share a device — with per-rank HIP_VISIBLE_DEVICES isolation every rank sees "cuda:0",
        # and keying on the index would switch the persistent recurrences off for the whole job (a silent slowdown).  A rank that
        # really shares its GPU still falls back by itself when a persistent launch times out (csrc/engine.h persist_check).
        ident = ("unknown", rank())
    key = (socket.gethostname(), ident)
    keys = [None] * world_size()
    dist.all_gather_object(keys, key)
    return sum(1 for k in keys if k == key)


def device_identity(device):
    """A job-wide identity of the physical GPU behind `device`: its UUID or PCI address from the device properties; else the entry of
    HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES that the index maps to (the launcher's per-rank isolation); None if
    nothing identifies it."""
    props = torch.cuda.get_device_properties(device)
    uuid = getattr(props, "uuid", None)
    if uuid is not None and str(uuid).strip("0-") != "":
        return ("uuid", str(uuid))
    bus = getattr(props, "pci_bus_id", None)
    if bus is not None and getattr(props, "pci_device_id", None) is not None:
        return ("pci", int(getattr(props, "pci_domain_id", 0) or 0), int(bus), int(props.pci_device_id))
    idx = torch.device(device).index
    idx = torch.cuda.current_device() if idx is None else idx
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        vis = os.environ.get(var)
        if vis:
            ids = [v.strip() for v in vis.split(",") if v.strip()]
            if idx < len(ids):
                return (var, ids[idx])
    return None


def configure_shared_gpu(engine) -> int:
    """the persistent recurrences (csrc/rnn_persist.h) need all 256 CUs of the GPU resident at once.  When several ranks share one
    device that cannot hold, so `persistent_rnn` is switched off up front (one launch per time step) instead of waiting for a bounded poll to
    time out and the library to fall back by itself.  Returns the number of ranks on this rank's GPU."""
    n = shared_device_ranks(engine.device)
    if n > 1:
        engine.set_option("persistent_rnn", 0)
        if rank() == 0:
            print(f"[hulc_amd] {n} ranks share one GPU: persistent_rnn off (one launch per time step)", flush=True)
    return n


def check_bucket_plan(buckets, numel: int) -> None:
    """The buckets of hulc_backward_allreduce must partition [0, numel): disjoint, gap-free, every element reduced exactly once."""
    t = sorted(buckets)
    if not t or t[0][0] != 0 or t[-1][1] != numel or any(t[i][1] != t[i + 1][0] for i in range(len(t) - 1)) or any(hi < lo for lo, hi in t):
        raise RuntimeError(f"gradient buckets {buckets} do not partition [0, {numel})")


def backward_overlapped(engine) -> None:
    """Backward of the LAST modality pass of a step with the gradient all-reduce overlapped (world > 1).

    98 % of the gradient bytes (decoder, plan networks, goal encoders: everything but the 0.75 M encoder parameters) are final
    after `backward(part=0)`; their all-reduce is issued asynchronously (RCCL runs it on its own stream, after the kernels
    already enqueued) and proceeds over xGMI while the encoder backward — the longest part of the step — still computes.
    The small encoder slice is reduced afterwards.  After this call `engine.flat_grads` holds the SUM over ranks.
    """
    if world_size() == 1:
        engine.backward()
        return
    if getattr(engine, "has_comm", False):                    # the library's own RCCL communicator: bucketed, reverse-forward order, event-ordered
        engine.backward_allreduce(getattr(engine, "comm_bucket_dtype", "fp32"))
        return
    # torch.distributed fallback.  The job-wide "a persistent recurrence of this step timed out on some rank" vote rides in a padding element of
    # the encoder slice (hulc_set_option dp_skip_vote: 1 = write this rank's vote before the collective, 2 = read the SUM after it) so that every
    # rank drops the optimizer step together — the failing rank's garbage is in everybody's sum
    vote = getattr(engine, "set_option", None)
    if os.environ.get("HULC_DP_OVERLAP", "1") == "0":        # experiment knob: plain backward, then one all-reduce
        engine.backward()
        if vote: vote("dp_skip_vote", 1)
        dist.all_reduce(engine.flat_grads, op=dist.ReduceOp.SUM)
        if vote: vote("dp_skip_vote", 2)
        return
    n_enc = engine.encoder_numel
    engine.backward(0)
    work = dist.all_reduce(engine.flat_grads[n_enc:], op=dist.ReduceOp.SUM, async_op=True)
    engine.backward(1)
    if vote: vote("dp_skip_vote", 1)
    dist.all_reduce(engine.flat_grads[:n_enc], op=dist.ReduceOp.SUM)
    if vote: vote("dp_skip_vote", 2)
    work.wait()
