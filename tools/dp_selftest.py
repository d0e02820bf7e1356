#!/usr/bin/env python
"""Multi-GPU self-test of the library's data-parallel collective (reference hulc/training.py:64-69 = DDP mean of per-rank gradients).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/dp_selftest.py      (N >= 2 GPUs)

One process per GPU.  For every (model kind, engine precision, bucket wire format) each rank runs the SAME step twice on its OWN,
rank-different batch:
  A. hulc_backward                         -> local gradients -> ONE flat torch.distributed (RCCL via torch) SUM all-reduce   = the reference sum
  B. hulc_backward_allreduce(bucket dtype) -> the library's own communicator: five reverse-forward buckets on the private stream, gated by
     events from the engine stream, overlapped with the backward
and compares B with A on every rank:
  * fp32 engine (deterministic kernels) + fp32 buckets: BIT-FOR-BIT at world 2 (a two-operand fp32 sum has one rounding whatever the ring
    order), <= 1e-6 relative at world > 2 (ring chunking differs between a 188 MB and a 60 MB collective);
  * 16-bit engines: their atomics reorder local sums run to run -> 2e-3 relative L2; 16-bit buckets: + the wire rounding -> 2e-2.
A bucket issued before its last gradient write, a missing event edge or overlapping / missing bucket ranges all show up as a mismatch,
because at world > 1 the SUM is not the identity (the 1-rank test in tests/test_gpu_fp16.py cannot see them).  Then hulc_adam_step with
grad_scale = 1 / world must leave IDENTICAL parameters on every rank, and the whole-buffer form hulc_allreduce_grads must agree too.
Prints DP_SELFTEST_OK on rank 0; any failure exits non-zero on every rank.  tests/test_gpu_dp.py runs it when >= 2 GPUs are visible."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def to_dev(mb, dev):
    out = {}
    for k, v in mb.items():
        if k == "use_for_aux":
            out["aux_rows"] = np.nonzero(v)[0].astype(np.int32)
        elif k == "plan_idx":
            out[k] = torch.from_numpy(v.astype(np.int32)).to(dev)
        else:
            out[k] = torch.from_numpy(v).to(dev)
    return out


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    if world < 2:
        print("dp_selftest needs >= 2 ranks (launch with torch.distributed.run --nproc-per-node N)")
        sys.exit(2)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    os.environ["HULC_DP_COMM"] = "capi"            # strict: the library path or an error
    from hulc_amd import parallel, spec
    from hulc_amd.engine import StepEngine
    from hulc_amd.utils import synthetic

    cases = [("hulc", "fp32", "fp32", 2, 2, 4), ("hulc", "bf16", "fp32", 2, 2, 4), ("hulc", "bf16", "bf16", 2, 2, 4), ("hulc", "fp16", "fp16", 2, 2, 4),
             ("gcbc", "fp32", "fp32", 2, 2, 4), ("mcil", "fp32", "fp32", 2, 0, 4), ("mcil", "bf16", "bf16", 2, 0, 4),
             ("hulc", "fp32", "fp32", 8, 0, 16), ("hulc", "bf16", "fp32", 16, 16, 32),     # these two: a backward long enough for the buckets to overlap it
             # mcil at a recurrence length the persistent launches take (S >= 3, 16-bit engine): the plan encoder's BiRNN backward runs AFTER the decoder
             # bucket has been issued.  Default routing = one launch per step under a collective in flight; persist_under_comm = 1 keeps the persistent
             # launches next to RCCL's kernels — if they ever lose their CUs the bounded polls time out, the optimizer step skips itself and the context
             # falls back (persistent_rnn_fallbacks in the report); either way the reduced gradients must match (VERDICT r3 #4 iii)
             ("mcil", "bf16", "fp32", 16, 0, 32, dict(persist_under_comm=0)), ("mcil", "bf16", "fp32", 16, 0, 32, dict(persist_under_comm=1)),
             ("mcil", "bf16", "bf16", 32, 0, 32, dict(persist_under_comm=1))]
    report = []
    for case in cases:
        kind, dtype, bucket, Bv, Bl, S = case[:6]
        opts = case[6] if len(case) > 6 else {}
        dims = spec.ModelDims(kind=kind, max_window=32, use_clip=(kind == "hulc" and Bl > 0))
        eng = StepEngine(dims, max(Bv, Bl), S, dtype=dtype, device=str(dev), dropout_p=0.0, seed=5, num_classes=dims.mix_classes)
        eng.load_numpy(spec.init_all(dims, seed=0, ln_jitter=True))
        if dtype == "fp16":
            eng.scaler_enable(init_scale=256.0)
        for k, v in opts.items():
            eng.set_option(k, v)
        assert parallel.setup_comm(eng, bucket) is True and eng.has_comm
        numel = eng.numel
        parallel.check_bucket_plan(eng.comm_buckets(), numel)
        batch = synthetic.make_batch(Bv, Bl, S, seed=100 + rank)            # rank-different windows
        if kind == "mcil":
            for mb in batch.values():
                mb["plan_eps"] = np.random.default_rng(7 + rank).standard_normal((mb["actions"].shape[0], 256)).astype(np.float32)
        mods = [(sc, to_dev(mb, dev)) for sc, mb in batch.items()]

        def run(last_backward):
            eng.zero_grads()
            for i, (sc, mb) in enumerate(mods):
                eng.forward_loss(mb, "lang" in sc, 1.0 / len(mods), 3.0, step=0)
                if i == len(mods) - 1:
                    last_backward()
                else:
                    eng.backward()
            torch.cuda.synchronize()

        run(eng.backward)
        g_local = eng.flat_grads.clone()
        g_ref = g_local.clone()
        dist.all_reduce(g_ref, op=dist.ReduceOp.SUM)
        n0 = eng.comm_stats()["collectives"]
        run(lambda: eng.backward_allreduce(bucket))
        g_lib = eng.flat_grads.clone()
        st = eng.comm_stats()
        assert st["collectives"] - n0 == 5, st
        assert torch.isfinite(g_lib).all()
        nref = g_ref.double().norm().item()
        rel = (g_lib.double() - g_ref.double()).norm().item() / max(nref, 1e-30)
        differs = (g_local.double() - g_ref.double()).norm().item() / max(nref, 1e-30)
        assert differs > 1e-3, f"{kind}: the ranks' gradients do not differ ({differs}) — the test would be vacuous"
        exact = bool(torch.equal(g_lib, g_ref))
        if dtype == "fp32" and bucket == "fp32":
            if world == 2:
                assert exact, f"{kind}/{dtype}/{bucket} B={Bv}+{Bl} S={S}: library RCCL SUM differs from the flat all-reduce (rel {rel:.3e})"
            else:
                assert rel < 1e-6, (kind, rel)
        elif bucket == "fp32":
            assert rel < 2e-3, (kind, dtype, bucket, rel)
        else:
            assert 0 < rel < 2e-2, (kind, dtype, bucket, rel)
        # whole-buffer form on the local gradients
        eng.flat_grads.copy_(g_local)
        eng.allreduce_grads(bucket)
        torch.cuda.synchronize()
        rel_flat = (eng.flat_grads.double() - g_ref.double()).norm().item() / max(nref, 1e-30)
        assert rel_flat < (1e-6 if bucket == "fp32" else 2e-2), (kind, dtype, bucket, rel_flat)
        # Adam on the reduced buffer (ordered after the collectives by events only) -> identical parameters everywhere
        eng.flat_grads.copy_(g_lib)
        eng.adam_step(lr=2e-4, grad_scale=1.0 / world)
        torch.cuda.synchronize()
        p = eng.flat_params
        stat = torch.stack([p.double().sum(), p.double().abs().sum(), p.double().square().sum()])
        allstat = [torch.zeros_like(stat) for _ in range(world)]
        dist.all_gather(allstat, stat)
        assert all(torch.equal(allstat[0], s) for s in allstat), f"{kind}: parameters differ between ranks after Adam"
        report.append(dict(kind=kind, engine=dtype, bucket=bucket, B=[Bv, Bl], S=S, rel_vs_flat_allreduce=rel, bit_exact=exact, rel_whole_buffer=rel_flat,
                           buckets=eng.comm_buckets(), options=opts, persistent_rnn=eng.get_option("persistent_rnn"),
                           persistent_rnn_fallbacks=eng.get_option("persistent_rnn_fallbacks")))
        eng.close()
        dist.barrier()
    if rank == 0:
        import json
        print(json.dumps(dict(world=world, cases=report)))
        print("DP_SELFTEST_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
