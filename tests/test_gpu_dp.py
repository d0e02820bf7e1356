"""Two data-parallel ranks on ONE GPU (gloo between the processes; RCCL refuses two ranks on one device): the real engine's split backward
(`backward(0)` -> async all-reduce of the non-encoder slice -> `backward(1)` -> encoder slice, hulc_amd.parallel.backward_overlapped) must leave
on every rank the SUM of the ranks' gradients, and SUM / world must equal the single-process gradient of the concatenated batch (every loss term
is a mean over windows — what data parallelism relies on).  fp32 (parity) engine, dropout off, injected plan sample."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

B, S = 8, 8


def _make(dev):
    from bench import synth_batch
    mb = synth_batch(B, S, dev, seed=5)
    g = torch.Generator(device=dev); g.manual_seed(17)
    mb["plan_idx"] = torch.randint(0, 32, (B, 32), device=dev, generator=g, dtype=torch.int32)
    return mb


def _slice(mb, lo, hi):
    return {k: (v[lo:hi].contiguous() if torch.is_tensor(v) else v) for k, v in mb.items()}


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", HULC_DP_COMM="torch")
    import torch.distributed as dist
    from hulc_amd import parallel, spec
    from hulc_amd.engine import StepEngine
    parallel.init_from_env("gloo")
    dev = torch.device("cuda:0")
    dims = spec.ModelDims(kind="hulc", max_window=S, use_clip=False)
    eng = StepEngine(dims, B, S, dtype="fp32", device="cuda:0", dropout_p=0.0, seed=3)
    eng.load_numpy(spec.init_all(dims, seed=0, ln_jitter=True))
    mb = _slice(_make(dev), rank * B // world, (rank + 1) * B // world)
    eng.zero_grads()
    eng.forward_loss(mb, False, 1.0, 3.0, step=0)
    parallel.backward_overlapped(eng)
    torch.cuda.synchronize()
    out[rank] = eng.flat_grads.cpu().numpy()
    eng.close()
    dist.destroy_process_group()


def test_two_ranks_one_gpu_sum_equals_full_batch_gradient():
    import torch.multiprocessing as mp
    from hulc_amd import spec
    from hulc_amd.engine import StepEngine
    out = mp.Manager().dict()
    try:
        mp.spawn(_worker, args=(2, 29800 + os.getpid() % 150, out), nprocs=2, join=True)
    except Exception as e:                      # a torch build whose gloo cannot reduce device tensors
        if "gloo" in str(e).lower() or "not supported" in str(e).lower():
            pytest.skip(f"gloo cannot all-reduce device tensors here: {e}")
        raise
    g0, g1 = out[0], out[1]
    assert np.array_equal(g0, g1)               # both ranks hold the same SUM
    dev = torch.device("cuda:0")
    dims = spec.ModelDims(kind="hulc", max_window=S, use_clip=False)
    eng = StepEngine(dims, B, S, dtype="fp32", device="cuda:0", dropout_p=0.0, seed=3)
    eng.load_numpy(spec.init_all(dims, seed=0, ln_jitter=True))
    eng.zero_grads()
    eng.forward_loss(_make(dev), False, 1.0, 3.0, step=0)
    eng.backward()
    torch.cuda.synchronize()
    full = eng.flat_grads.cpu().numpy().astype(np.float64)
    eng.close()
    rel = np.linalg.norm(g0.astype(np.float64) / 2 - full) / np.linalg.norm(full)
    assert rel < 1e-4, rel


def _module_worker(rank, world, port, out):
    """Hulc.training_step (vis + lang + CLIP: the logged-scalar all-reduce of hulc.py:512-532) + FusedAdam.step on two ranks of one GPU.
    HULC_DP_COMM=auto: the library RCCL communicator is attempted (hulc_comm_prepare succeeds on both ranks), RCCL refuses two ranks on one
    device inside hulc_comm_init, every rank must agree on the torch.distributed fallback (parallel.setup_comm's votes) and the step must
    complete with identical parameters on both ranks.  (The default, HULC_DP_COMM=capi, turns the same failure into an error on every rank.)"""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    os.environ["HULC_DP_COMM"] = "auto"
    import torch.distributed as dist
    from golden_util import load_case
    from hulc_amd import config, parallel
    from test_gpu_module import ref_style_batch
    parallel.init_from_env("gloo")
    dims, P, batch, fx = load_case("hulc_tiny")
    cfg = config.compose(os.path.join(ROOT, "conf"), "config", ["model=hulc", "trainer.precision=fp32", "datamodule.batch_size=4"])
    model = config.instantiate(cfg.model, device="cuda:0", max_seq_len=32)
    model.load_state_dict({n: torch.from_numpy(P[n]) for n in P}, strict=False)
    model.eval()
    loss = float(model.training_step(ref_style_batch(batch), 0))
    opt = model.configure_optimizers()["optimizer"]
    opt.step()
    torch.cuda.synchronize()
    w = dict(model.named_parameters())["action_decoder.mean_fc.weight"].detach().cpu().numpy().copy()
    # two ranks on one GPU: the module switched the persistent recurrences off up front (parallel.configure_shared_gpu)
    assert parallel.shared_device_ranks(model.engine.device) == 2 and model.engine.get_option("persistent_rnn") == 0
    out[rank] = (loss, model.logged["train/lang_clip_loss"], bool(getattr(model.engine, "has_comm", False)), w)
    dist.destroy_process_group()


def test_module_training_step_and_adam_two_ranks_one_gpu():
    import torch.multiprocessing as mp
    from golden_util import load_case
    out = mp.Manager().dict()
    mp.spawn(_module_worker, args=(2, 29500 + os.getpid() % 150, out), nprocs=2, join=True)
    (l0, c0, comm0, w0), (l1, c1, comm1, w1) = out[0], out[1]
    _, _, _, fx = load_case("hulc_tiny")
    ref = float(fx["loss_total"])
    assert abs(l0 - ref) <= 1e-3 * abs(ref) and abs(l1 - ref) <= 1e-3 * abs(ref)      # both ranks ran the same batch
    assert abs(c0 - c1) < 1e-6                                                         # the logged scalar is the mean over ranks on both
    assert comm0 == comm1                                                              # one collective path for the whole job
    assert np.array_equal(w0, w1) and np.isfinite(w0).all()                            # identical parameters after the step (mean gradient of identical ranks)


def _strict_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", HULC_DP_COMM="capi")
    import torch.distributed as dist
    from hulc_amd import parallel, spec
    from hulc_amd.engine import StepEngine
    parallel.init_from_env("gloo")
    dims = spec.ModelDims(kind="hulc", max_window=32, use_clip=False)
    eng = StepEngine(dims, 2, 4, dtype="bf16", device="cuda:0", dropout_p=0.0)
    eng.load_numpy(spec.init_all(dims, seed=0))
    try:
        up = parallel.setup_comm(eng)
        out[rank] = ("up", up)
    except RuntimeError as e:
        out[rank] = ("raised", str(e))
    eng.close()
    dist.destroy_process_group()


def test_strict_mode_raises_on_every_rank_when_rccl_cannot_come_up():
    """HULC_DP_COMM=capi (the default): two ranks on one device -> ncclCommInitRank fails -> BOTH ranks raise (no silent torch.distributed
    path), with the way out named in the message.  On a box where RCCL does accept the two ranks the communicator is up."""
    import torch.multiprocessing as mp
    out = mp.Manager().dict()
    mp.spawn(_strict_worker, args=(2, 29350 + os.getpid() % 100, out), nprocs=2, join=True)
    assert out[0][0] == out[1][0]
    if out[0][0] == "raised":
        assert "HULC_DP_COMM=auto" in out[0][1] and "HULC_DP_COMM=auto" in out[1][1]
    else:
        assert out[0][1] is True and out[1][1] is True


def test_library_rccl_selftest_on_two_gpus():
    """a17: `hulc_backward_allreduce` between REAL ranks — tools/dp_selftest.py under torchrun on 2 GPUs: rank-different gradients, the
    library's bucketed RCCL SUM against plain backward + one flat torch.distributed all-reduce, hulc / gcbc / mcil, fp32 and 16-bit buckets,
    then Adam -> identical parameters on every rank.  Skipped (not absent) on a 1-GPU box; the driver's multi-GPU node runs it."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL refuses two ranks on one device)")
    import subprocess
    n = min(torch.cuda.device_count(), 8)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port",
           str(29650 + os.getpid() % 100), os.path.join(ROOT, "tools", "dp_selftest.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-4000:]
    assert "DP_SELFTEST_OK" in r.stdout, r.stdout[-4000:]


@pytest.mark.parametrize("extra", [[], ["--lang", "1"], ["--model", "mcil", "--bucket", "bf16"]])
def test_bench_multi_gpu_code_path_rehearsal_on_one_gpu(extra):
    """The N > 1 branch of bench.py — library communicator, the self-check of the bucketed SUM against a flat torch.distributed all-reduce,
    hulc_backward_allreduce inside the timed loop, the bucket timeline in the JSON — has no multi-GPU box to run on before the driver's.
    `--force-comm 1` runs exactly that code with a 1-rank process group and a 1-rank RCCL communicator; the line must parse, carry the
    `allreduce` object with five timed buckets that partition the buffer, and keep roofline / mfma_groups."""
    import json
    import subprocess
    env = dict(os.environ, MASTER_PORT=str(29700 + os.getpid() % 200))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--force-comm", "1", "--steps", "4", "--warmup", "1", "--preroll", "2", "--no-cpu-baseline", "--batch", "16"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = r.stdout.strip().splitlines()
    assert lines[-1].startswith('{"metric"'), lines[-3:]          # the JSON line is the LAST line of stdout (the driver's contract)
    d = json.loads(lines[-1])
    ar = d["allreduce"]
    assert ar and "REHEARSAL" in ar["path"] and ar["collectives"] >= 5 * 4
    sc = ar["selfcheck"]
    assert sc["rel_l2_whole_buffer"] <= 2e-2 and sc["rel_l2_bucketed_overlapped"] <= sc["tolerance"]
    tl = ar["timeline"]
    assert tl["backward_us"] > 0 and len(tl["buckets"]) >= 4
    assert sum(b["bytes"] for b in tl["buckets"]) == sum(ar["bucket_bytes"])
    assert tl["buckets"][0]["issued_at_us"] < 0 <= tl["exposed_after_backward_us"]
    assert d["n_gpus"] == 1 and d["roofline"] and d["mfma_groups"] and d["value"] > 0


def test_bench_gpus_flag_means_what_it_says():
    """`--gpus N` is the number of ranks: without a launcher `--gpus 2` re-launches bench.py under torch.distributed.run with two
    ranks — on a box with fewer GPUs it must FAIL, never print an `n_gpus: 1` line; under a launcher a WORLD_SIZE that disagrees is an error."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--preroll", "1", "--no-cpu-baseline", "--batch", "8"]
    n = torch.cuda.device_count()
    r = subprocess.run(base + ["--gpus", str(n + 1)], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode != 0 and '"metric"' not in r.stdout, r.stdout[-2000:]
    assert "GPU(s) are visible" in r.stderr
    r = subprocess.run(base + ["--gpus", "2"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and '"metric"' not in r.stdout and "WORLD_SIZE=1" in r.stderr
    if n >= 2:              # the self-launch itself, wherever two GPUs exist
        import json
        r = subprocess.run(base + ["--gpus", "2"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        d = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith('{"metric"')][-1])
        assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2"
