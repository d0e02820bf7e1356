"""CPU: the mini-Hydra loader on the repo's conf/ tree; data-parallel semantics over gloo (world_size 2)."""
import os
import sys

import numpy as np
import pytest
import torch

from hulc_amd import config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONF = os.path.join(ROOT, "conf")


def test_compose_defaults_and_interpolation():
    c = config.compose(CONF, "config")
    m = c.model
    assert m._target_ == "hulc.models.hulc.Hulc" and m._recursive_ is False
    assert m.kl_beta == 0.01 and m.kl_balancing_mix == 0.8 and m.clip_auxiliary_loss_beta == 3.0        # ${loss.*}
    assert m.plan_recognition.max_position_embeddings == 32                                            # ${datamodule.max_window_size}
    assert m.action_decoder.out_features == 7 and m.action_decoder.act_max_bound == [1.0] * 7
    assert m.proj_vis_lang.im_dim == 4096 and m.proj_vis_lang.lang_dim == 32                          # nested ${model.*}
    assert m.optimizer.lr == 2e-4
    assert m.perceptual_encoder.rgb_static.visual_features == 64 and m.perceptual_encoder.proprio == {}
    assert sorted(config.missing_keys(c)) == sorted([
        "model.plan_proposal.perceptual_features", "model.plan_proposal.plan_features", "model.plan_recognition.in_features",
        "model.plan_recognition.plan_features", "model.visual_goal.in_features", "model.action_decoder.plan_features",
        "model.action_decoder.perceptual_features"])   # filled by setup_input_sizes in the reference (hulc.py:155-187)


def test_overrides_group_value_delete():
    c = config.compose(CONF, "config", ["model=gcbc", "trainer.precision=fp32", "datamodule.batch_size=8", "loss.kl_beta=0.5",
                                        "callbacks/kl_schedule=sigmoid", "~callbacks/checkpoint", "+extra.flag=true"])
    assert c.model._target_ == "hulc.models.gcbc.GCBC"
    assert c.model.precision == "fp32" and c.model.max_batch_size == 8 and c.model.kl_beta == 0.5
    assert c.callbacks.kl_schedule._target_.endswith("KLSigmoidSchedule") and c.callbacks.kl_schedule.max_kl_beta == 0.5
    assert "checkpoint" not in c.callbacks and c.extra.flag is True


def test_instantiate_target_map():
    assert config.TARGET_MAP["hulc.models.hulc.Hulc"] == "hulc_amd.hulc.Hulc"
    obj = config.instantiate({"_target_": "collections.OrderedDict", "a": 1})
    assert obj["a"] == 1 and config.instantiate({}) is None and config.instantiate(None) is None


def test_kl_schedules():
    from hulc_amd.trainer import KLLinearSchedule, KLSigmoidSchedule
    lin = KLLinearSchedule(10, 50, 0.01)
    assert lin._beta(0) == 0 and abs(lin._beta(30) - 0.005) < 1e-9 and lin._beta(60) == 0.01
    sig = KLSigmoidSchedule(10, 50, 0.01)
    assert sig._beta(10) < 1e-4 and abs(sig._beta(30) - 0.005) < 1e-9 and sig._beta(50) > 0.0099
    # outside [start, end] the reference clamps exactly (kl_callbacks.py:41-44): 0 before, max after — no extrapolated sigmoid tail
    assert sig._beta(0) == 0.0 and sig._beta(9) == 0.0 and sig._beta(51) == 0.01 and sig._beta(1000) == 0.01
    # values of the reference formula sigmoid((x - shift) / (scale / 12)), shift = (end + start) / 2, scale = end - start
    import math
    for ep in (10, 17, 30, 42, 50):
        want = 0.01 / (1.0 + math.exp(-((ep - 30.0) / (40.0 / 12.0))))
        assert abs(sig._beta(ep) - want) < 1e-12


def test_state_manifest_covers_spec_layout():
    """tests/golden/state_manifest.json (tools/gen_golden_ckpt.py = the reference's state_dict()): every PARAMETER key and shape of every
    model kind equals hulc_amd.spec.layout — the flat-buffer layout the engine binds by name."""
    import json
    man = json.load(open(os.path.join(ROOT, "tests", "golden", "state_manifest.json")))
    from hulc_amd import spec
    for name, kind, kw in (("hulc_w32", "hulc", dict(max_window=32)), ("hulc_w64", "hulc", dict(max_window=64)), ("gcbc_w32", "gcbc", dict(max_window=32)),
                           ("mcil_w32", "mcil", dict(max_window=32, use_clip=False)), ("mcil_gru_w32", "mcil", dict(max_window=32, use_clip=False, rnn_type="gru"))):
        lay, total = spec.layout(spec.ModelDims(kind=kind, **kw))
        ref = {k: tuple(e["shape"]) for k, e in man[name]["state_dict"].items() if e["param"]}
        assert set(ref) == set(lay), (name, set(ref) ^ set(lay))
        for k, shp in ref.items():
            assert tuple(lay[k][1]) == shp, (name, k)
        assert sum(int(np.prod(s)) if len(s) else 1 for s in ref.values()) == man[name]["n_params"]


def _ddp_worker(rank, world, port, out):
    import torch.distributed as dist
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from hulc_amd import parallel, spec
    parallel.init_from_env("gloo")
    d = spec.ModelDims()
    lay, total = spec.layout(d)
    # each rank: a different deterministic "gradient"; unused tensors (GCBC-style) stay zero on both
    g = torch.zeros(total)
    torch.manual_seed(100 + rank)
    g[: total // 2] = torch.randn(total // 2)
    flat = g.clone()
    parallel.allreduce_sum_(flat, bucket_elems=0 if rank == 0 else 0)
    flat_b = g.clone()
    parallel.allreduce_sum_(flat_b, bucket_elems=5_000_000)          # bucketed path must agree with the single collective
    p = torch.full((8,), float(rank))
    parallel.broadcast_(p, 0)
    m = parallel.mean_scalar(float(rank + 1))
    out[rank] = (flat[:1000].clone(), bool(torch.equal(flat, flat_b)), g[:1000].clone(), p, m, float(flat[total // 2:].abs().max()))
    dist.destroy_process_group()


def test_data_parallel_mean_gradient_gloo():
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    out = mgr.dict()
    port = 29600 + os.getpid() % 300
    mp.spawn(_ddp_worker, args=(2, port, out), nprocs=2, join=True)
    (s0, eq0, g0, p0, m0, z0), (s1, eq1, g1, p1, m1, z1) = out[0], out[1]
    assert eq0 and eq1
    assert torch.allclose(s0, g0 + g1) and torch.equal(s0, s1)            # both ranks hold the SUM; Adam folds 1/world
    assert torch.equal(p1, torch.zeros(8)) and m0 == m1 == 1.5
    assert z0 == 0.0 and z1 == 0.0                                        # tensors without a gradient stay exactly zero
    # mean-of-rank-gradients + Adam(grad_scale=1/world) == Adam on the averaged gradient (oracle Adam as the checker)
    import hulc_oracle as O
    P = {"w": np.ones(1000, np.float32)}
    Pa = {"w": np.ones(1000, np.float32)}
    O.adam_step(P, {"w": ((g0 + g1) / 2).numpy()}, {}, 1)
    O.adam_step(Pa, {"w": (s0 * 0.5).numpy()}, {}, 1)
    assert np.array_equal(P["w"], Pa["w"])


class _FakeEngine:
    """CPU stand-in with the StepEngine surface backward_overlapped touches (flat_grads, encoder_numel, backward(part))."""

    def __init__(self, rank):
        self.flat_grads = torch.zeros(1000)
        self.encoder_numel = 128
        self.rank = rank
        self.calls = []

    def backward(self, part=-1):
        self.calls.append(part)
        if part in (-1, 0):
            self.flat_grads[self.encoder_numel:] += float(self.rank + 1)
        if part in (-1, 1):
            self.flat_grads[: self.encoder_numel] += 10.0 * (self.rank + 1)


def _overlap_worker(rank, world, port, out):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from hulc_amd import parallel
    parallel.init_from_env("gloo")
    e = _FakeEngine(rank)
    parallel.backward_overlapped(e)
    out[rank] = (e.calls, float(e.flat_grads[0]), float(e.flat_grads[-1]))
    dist.destroy_process_group()


def test_backward_overlapped_allreduce_gloo():
    import torch.multiprocessing as mp
    out = mp.Manager().dict()
    mp.spawn(_overlap_worker, args=(2, 29700 + os.getpid() % 200, out), nprocs=2, join=True)
    for r in (0, 1):
        calls, enc, rest = out[r]
        assert calls == [0, 1] and enc == 30.0 and rest == 3.0      # both slices hold the SUM over ranks
    from hulc_amd import parallel
    e = _FakeEngine(0)
    parallel.backward_overlapped(e)                                   # world 1: plain whole backward
    assert e.calls == [-1]


def _bucket_worker(rank, world, port, out):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from hulc_amd import parallel, spec
    parallel.init_from_env("gloo")
    d = spec.ModelDims()
    lay, total = spec.layout(d)
    sched = parallel.bucket_schedule(lay, total)
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(total, generator=g)
    mine = flat.clone()
    # the REAL bucket schedule of the library (reverse-forward module groups), one async collective per bucket in issue order
    works = [dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True) for lo, hi in sched]
    for w in works:
        w.wait()
    other = torch.randn(total, generator=torch.Generator().manual_seed(100 + (1 - rank)))
    out.put((rank, bool(torch.allclose(flat, mine + other, rtol=0, atol=1e-6)), sched))
    dist.destroy_process_group()


def test_bucket_schedule_partitions_buffer_and_reduces_world2():
    """The library's bucket plan (hulc_amd.parallel.bucket_schedule mirrors Engine::bucket_plan; the GPU test compares the two) driven with
    real tensors over 2 gloo ranks: reverse-forward order, a partition of the flat buffer, every element reduced exactly once."""
    import torch.multiprocessing as mp
    from hulc_amd import parallel, spec
    for kind, kw in (("hulc", {}), ("gcbc", {}), ("mcil", dict(use_clip=False)), ("mcil", dict(use_clip=False, rnn_type="gru"))):
        lay, total = spec.layout(spec.ModelDims(kind=kind, **kw))
        s = parallel.bucket_schedule(lay, total)
        assert len(s) == 5 and s[0][1] == total and s[-1][0] == 0
        t = sorted(s)
        assert t[0][0] == 0 and t[-1][1] == total and all(t[i][1] == t[i + 1][0] for i in range(4)), (kind, s)   # a partition of the buffer
        assert [n for n in ("action_decoder.", "plan_proposal.", "plan_recognition.", "visual_goal.", "perceptual_encoder.")
                if not (s[["action_decoder.", "plan_proposal.", "plan_recognition.", "visual_goal.", "perceptual_encoder."].index(n)][0]
                        <= min(off for k, (off, _) in lay.items() if k.startswith(n)) < s[["action_decoder.", "plan_proposal.", "plan_recognition.", "visual_goal.", "perceptual_encoder."].index(n)][1])] == []   # issue order = the order the backward finalises the groups
        first = min(off for n, (off, _) in lay.items() if n.startswith("action_decoder."))
        assert s[0][0] == first and s[-1][1] == min(off for n, (off, _) in lay.items() if not n.startswith("perceptual_encoder."))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 300
    ps = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=180) for _ in ps]
    for p in ps:
        p.join(60)
    assert all(ok for _, ok, _ in res) and res[0][2] == res[1][2]


def test_resume_true_reenters_newest_run_with_checkpoint(tmp_path):
    """`resume=true` picks the newest run directory under the log_dir pattern that holds a checkpoint; without one (or without the
    flag) `{now}` expands to a fresh directory; an explicit log_dir is used as is."""
    import time
    from hulc_amd import training
    pat = str(tmp_path / "runs" / "{now}")
    fresh = training.resolve_log_dir(pat, resume=True)
    assert fresh.startswith(str(tmp_path / "runs")) and "{now}" not in fresh          # nothing to resume: a fresh run
    for i, name in enumerate(("2026-01-01/10-00-00", "2026-01-02/09-30-00", "2026-01-03/08-00-00")):
        d = tmp_path / "runs" / name / "saved_models"
        d.mkdir(parents=True)
        if i < 2:                                                                      # the newest directory holds NO checkpoint
            (d / f"epoch={i}.ckpt").write_bytes(b"x")
            t = time.time() - 100 + i
            os.utime(d / f"epoch={i}.ckpt", (t, t))
    assert training.resolve_log_dir(pat, resume=True) == str(tmp_path / "runs" / "2026-01-02/09-30-00")
    assert training.resolve_log_dir(pat, resume=False) not in [str(tmp_path / "runs" / n) for n in ("2026-01-01/10-00-00", "2026-01-02/09-30-00")]
    explicit = str(tmp_path / "runs" / "2026-01-01/10-00-00")
    assert training.resolve_log_dir(explicit, resume=False) == explicit


def test_resume_skips_runs_of_another_configuration(tmp_path, capsys):
    """`resume=true` used to re-enter the newest run directory that held ANY checkpoint.  A run directory now records the
    configuration fingerprint it was started with (model / loss / training / datamodule groups, seed, precision); resume only re-enters a run
    whose fingerprint matches, reports the ones it skips, and an explicit log_dir of another configuration is refused by `train`."""
    import json
    import os
    import time
    from hulc_amd import config, training
    conf = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "conf")
    cfg_hulc = config.compose(conf, "config", [])
    cfg_gcbc = config.compose(conf, "config", ["model=gcbc"])
    fp_h, fp_g = training.config_fingerprint(cfg_hulc), training.config_fingerprint(cfg_gcbc)
    assert fp_h != fp_g and fp_h == training.config_fingerprint(config.compose(conf, "config", ["trainer.max_epochs=7", "resume=true"]))   # bookkeeping keys do not count
    pat = str(tmp_path / "runs" / "{now}")
    for name, fp in (("2026-01-01/10-00-00", fp_h), ("2026-01-02/09-30-00", fp_g)):
        d = tmp_path / "runs" / name / "saved_models"
        d.mkdir(parents=True)
        (d / "epoch=0.ckpt").write_bytes(b"x")
        json.dump(fp, open(tmp_path / "runs" / name / training.RUN_CONFIG, "w"))
        time.sleep(0.02)
    # the newest run is a GCBC run: a HULC job must skip it and continue the older HULC run
    assert training.resolve_log_dir(pat, resume=True, fp=fp_h) == str(tmp_path / "runs" / "2026-01-01/10-00-00")
    assert "different configuration" in capsys.readouterr().out
    assert training.resolve_log_dir(pat, resume=True, fp=fp_g) == str(tmp_path / "runs" / "2026-01-02/09-30-00")
    ok, why = training.run_config_matches(str(tmp_path / "runs" / "2026-01-02/09-30-00"), fp_h)
    assert not ok and "model" in why
    # a run directory without a recorded fingerprint (older runs) is still resumable
    d = tmp_path / "runs" / "2026-01-03/08-00-00" / "saved_models"
    d.mkdir(parents=True)
    (d / "epoch=0.ckpt").write_bytes(b"x")
    assert training.resolve_log_dir(pat, resume=True, fp=fp_h) == str(tmp_path / "runs" / "2026-01-03/08-00-00")


def _metrics_worker(rank, world, port, out):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hulc_amd import parallel
    # rank 0 saw masked language rows in 2 of its 3 validation batches, rank 1 in none; the ranks' dicts also differ in insertion order
    if rank == 0:
        sums, counts = {"val_act/a": 3.0, "lang_gt/train_sr": 1.0, "val_kl/k": 6.0}, {"val_act/a": 3, "lang_gt/train_sr": 2, "val_kl/k": 3}
    else:
        sums, counts = {"val_kl/k": 12.0, "val_act/a": 9.0}, {"val_kl/k": 3, "val_act/a": 3}
    out[rank] = parallel.mean_metrics(sums, counts)
    dist.destroy_process_group()


def test_validation_metric_reduction_with_rank_dependent_key_sets():
    """ADVICE r4 (trainer.validate): `lang_gt/*` is logged only by batches with masked language rows, so the ranks' key sets and orders differ;
    the reduction must neither hang nor average different metrics together."""
    import torch.multiprocessing as mp
    out = mp.Manager().dict()
    mp.spawn(_metrics_worker, args=(2, 29900 + os.getpid() % 90, out), nprocs=2, join=True)
    assert out[0] == out[1] == {"lang_gt/train_sr": 0.5, "val_act/a": 2.0, "val_kl/k": 3.0}
    from hulc_amd import parallel
    assert parallel.mean_metrics({"a": 3.0}, {"a": 2}) == {"a": 1.5}          # world 1: no collective


# ---- 8 ranks on gloo (VERDICT r5 #7 i): everything of the N > 1 bring-up that is host logic, at the world size the driver's SCALE run uses ----------
class _CommEngine:
    """CPU stand-in with the StepEngine surface parallel.setup_comm / backward_overlapped touch; `fail` = the phase this rank fails in."""
    comm_rehearsal = True

    def __init__(self, rank, fail=None):
        self.rank, self.fail, self.device, self.has_comm = rank, fail, "cpu", False
        self.calls, self.numel, self.encoder_numel = [], 4096, 512
        self.flat_grads = torch.zeros(self.numel)
        self.skip_pad, self.my_vote, self.skipped = 100, 0.0, None      # one padding element of the encoder slice carries the skip vote

    def comm_prepare(self):
        self.calls.append("prepare")
        if self.fail == "prepare":
            raise RuntimeError("RCCL not found (rehearsal)")

    def comm_unique_id(self):
        self.calls.append("id")
        if self.fail == "id":
            raise RuntimeError("ncclGetUniqueId failed (rehearsal)")
        return bytes(range(128))

    def comm_init(self, uid, rank, world):
        self.calls.append(("init", bytes(uid), rank, world))
        if self.fail == "init":
            raise RuntimeError("ncclCommInitRank failed (rehearsal)")
        self.has_comm = True

    def comm_destroy(self):
        self.calls.append("destroy")
        self.has_comm = False

    def comm_buckets(self):
        return [(2048, 4096), (1536, 2048), (1024, 1536), (512, 1024), (0, 512)]

    def backward(self, part=-1):
        if part in (-1, 0):
            self.flat_grads[self.encoder_numel:] += float(self.rank + 1)
        if part in (-1, 1):
            self.flat_grads[: self.encoder_numel] += 10.0 * (self.rank + 1)
            self.flat_grads[self.skip_pad] = 0.0

    def set_option(self, name, value):                   # the library's dp_skip_vote protocol (csrc/engine.h skip_vote_put / skip_vote_get)
        assert name == "dp_skip_vote"
        if value == 1:
            self.flat_grads[self.skip_pad] = self.my_vote
        else:
            self.skipped = bool(self.flat_grads[self.skip_pad] != 0)
            self.flat_grads[self.skip_pad] = 0.0


def _world8_worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from hulc_amd import parallel, spec
    parallel.init_from_env("gloo")
    res = {}
    # (1) setup_comm: every rank leaves every scenario with the SAME answer, nobody is left inside comm_init alone
    for name, failing, phase in (("ok", None, None), ("prepare", 5, "prepare"), ("id", 0, "id"), ("init", 3, "init")):
        for mode in ("auto", "capi"):
            os.environ["HULC_DP_COMM"] = mode
            e = _CommEngine(rank, phase if rank == failing else None)
            try:
                up = parallel.setup_comm(e, "fp32")
            except RuntimeError as ex:
                up = "raised:" + str(ex)[:40]
            res[(name, mode)] = (up, [c if isinstance(c, str) else c[0] for c in e.calls], [c for c in e.calls if not isinstance(c, str)], e.has_comm)
    os.environ["HULC_DP_COMM"] = "capi"
    # (2) the library's real bucket schedule with real tensors: a partition, every element reduced once, at world 8
    lay, total = spec.layout(spec.ModelDims())
    sched = parallel.bucket_schedule(lay, total)
    parallel.check_bucket_plan(sched, total)
    n = 1 << 20                                           # the first 1 M elements of every bucket (8 ranks x 188 MB would not fit the CI box)
    flat = torch.full((total,), float(rank + 1)) if total <= n else None
    parts = [torch.full((min(hi - lo, n),), float(rank + 1)) for lo, hi in sched]
    works = [dist.all_reduce(p, op=dist.ReduceOp.SUM, async_op=True) for p in parts]
    for w in works:
        w.wait()
    res["buckets"] = (sched, all(bool((p == 36.0).all()) for p in parts))
    # (3) the torch.distributed fallback of a step + the job-wide skip vote: rank 6's recurrence "timed out" -> EVERY rank skips; nobody votes -> nobody skips
    for tag, voter in (("vote", 6), ("novote", None)):
        e = _CommEngine(rank)
        e.my_vote = 1.0 if rank == voter else 0.0
        parallel.backward_overlapped(e)
        res[tag] = (e.skipped, float(e.flat_grads[0]), float(e.flat_grads[-1]), float(e.flat_grads[e.skip_pad]))
    # (4) epoch metrics with rank-dependent key sets (lang_gt/* only on ranks whose batches had masked rows)
    sums, counts = {"val/a": float(rank)}, {"val/a": 1}
    if rank % 3 == 0:
        sums["lang_gt/x"], counts["lang_gt/x"] = 2.0 * rank, 2
    res["metrics"] = parallel.mean_metrics(sums, counts)
    res["mean_scalar"] = parallel.mean_scalar(float(rank))
    q.put((rank, res))
    dist.destroy_process_group()


def test_world8_bringup_votes_buckets_skip_vote_and_metrics_on_gloo():
    """The host side of the first 8-GPU run, rehearsed at world 8 on CPU (VERDICT r5 #7 i): parallel.setup_comm's three votes with a failure
    injected in each phase on a different rank (HULC_DP_COMM=auto -> every rank falls back together; =capi -> every rank raises), the library's
    bucket schedule as 5 async collectives, the skip vote riding the gradients' own SUM through the torch.distributed fallback, and the
    rank-dependent metric keys."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + os.getpid() % 90
    W = 8
    ps = [ctx.Process(target=_world8_worker, args=(r, W, port, q)) for r in range(W)]
    for p in ps:
        p.start()
    got = dict(q.get(timeout=600) for _ in ps)
    for p in ps:
        p.join(60)
    assert sorted(got) == list(range(W))
    for r in range(W):
        res = got[r]
        up, names, inits, has = res[("ok", "auto")]
        assert up is True and has and names == (["prepare", "id", "init"] if r == 0 else ["prepare", "init"]) and inits == [("init", bytes(range(128)), r, W)]
        assert res[("ok", "capi")][0] is True
        # a rank-local failure in ANY phase: nobody reports the library path up, nobody keeps a communicator, nobody entered a later phase alone
        for name in ("prepare", "id", "init"):
            up, names, inits, has = res[(name, "auto")]
            assert up is False and not has, (r, name, up)
            assert str(res[(name, "capi")][0]).startswith("raised:"), (r, name)
        assert "init" not in res[("prepare", "auto")][1] and "init" not in res[("id", "auto")][1]
        assert ("destroy" in res[("init", "auto")][1]) == (r != 3)               # the ranks whose init succeeded tear their communicator down again
        sched, okb = res["buckets"]
        assert okb and sched == got[0]["buckets"][0] and len(sched) == 5
        skipped, enc, rest, pad = res["vote"]
        assert skipped is True and enc == 10.0 * 36 and rest == 36.0 and pad == 0.0
        assert res["novote"][0] is False and res["novote"][1:] == (360.0, 36.0, 0.0)
        assert res["metrics"] == {"val/a": 3.5, "lang_gt/x": (0 + 6 + 12) / 6.0} and res["mean_scalar"] == 3.5
