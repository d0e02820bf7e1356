"""Chunked evaluation of the numpy oracle in worker processes (spawned: the parent holds a HIP context).  Test infrastructure only.

Every loss of the step except the CLIP term is a mean over windows, so losses and gradients of a batch are the means of its window chunks; a
worker evaluates a contiguous run of chunks with `oracle/hulc_oracle.py` (optionally in its rounding-aware mode) and returns the SUMS.  The
CLIP term couples all language rows of a batch (an n x n contrastive loss): a modality that carries it is evaluated by ONE worker on all its
windows (`full` jobs)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def case_dims(case):
    from hulc_amd import spec
    return spec.ModelDims(kind=case["kind"], max_window=case["max_window"], use_clip=bool(case.get("use_clip", False)), rnn_type=case.get("rnn_type", "rnn"))


def case_batch(case):
    """The numpy batch of a case — the SAME call in the test process (which feeds the engine) and in every worker.  mcil: the injected N(0,1)
    reparametrisation draw (`plan_eps`) from a seeded numpy generator."""
    import numpy as np
    from hulc_amd.utils import synthetic
    Bl = case.get("B_lang", 0)
    batch = synthetic.make_batch(case["B"], Bl, case["S"], seed=case["seed"], edge_frac=0.05, aux_mask="all")
    if case["kind"] == "mcil":
        for i, (scope, mb) in enumerate(batch.items()):
            mb["plan_eps"] = np.random.default_rng(case["seed"] * 31 + i).standard_normal((mb["actions"].shape[0], 256)).astype(np.float32)
    return batch


def _share(case, P, batch):
    """Parameters and batch written ONCE by the parent (flat .npy files in a temp directory, memory-mapped by the workers): every worker regenerating
    74 - 143 M parameters and a 1 GB batch with the portable counter RNG was most of the full-size tests' run time."""
    import tempfile
    import numpy as np
    from hulc_amd import spec
    d = tempfile.mkdtemp(prefix="hulc_oracle_pool_")
    lay, numel = spec.layout(case_dims(case))
    flat = np.zeros(numel, np.float32)
    for n, (off, shape) in lay.items():
        flat[off:off + P[n].size] = np.asarray(P[n], np.float32).reshape(-1)
    np.save(os.path.join(d, "params.npy"), flat)
    keys = {}
    for scope, mb in batch.items():
        keys[scope] = sorted(mb)
        for k, v in mb.items():
            np.save(os.path.join(d, f"b_{scope}_{k}.npy"), np.ascontiguousarray(v))
    return d, keys


def _load_shared(case, d, keys, scope):
    import numpy as np
    from hulc_amd import spec
    lay, _ = spec.layout(case_dims(case))
    flat = np.load(os.path.join(d, "params.npy"), mmap_mode="r")
    P = {n: np.asarray(flat[off:off + int(np.prod(shape)) if len(shape) else off + 1]).reshape(shape) for n, (off, shape) in lay.items()}
    mb = {k: np.load(os.path.join(d, f"b_{scope}_{k}.npy"), mmap_mode="r") for k in keys[scope]}
    return P, mb


def _work(args):
    case, scope, chunks, CH, w_mod, w_clip, threads = args[:7]
    shared = args[7] if len(args) > 7 else None
    os.environ["OMP_NUM_THREADS"] = os.environ["OPENBLAS_NUM_THREADS"] = os.environ["MKL_NUM_THREADS"] = str(threads)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import dataclasses
    import numpy as np
    import hulc_oracle as O
    from hulc_amd import spec
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=threads)
    except Exception:
        pass
    dims = case_dims(case)
    if shared:
        P, mb = _load_shared(case, shared[0], shared[1], scope)
    else:
        P = spec.init_all(dims, seed=case["seed"], ln_jitter=True)
        mb = case_batch(case)[scope]
    is_lang = "lang" in scope
    dims_run = dims if w_clip else dataclasses.replace(dims, use_clip=False)      # chunked jobs never carry the CLIP term
    O.set_operand_rounding(case.get("mode"), case.get("gscale", 1.0))
    O.CONDITION_SUMS = bool(case.get("cond", False))        # + G["abs/..."]: the conv gradients' condition sums (hulc_oracle._cond)
    G, sums, embs = {}, dict(kl=0.0, action=0.0, total=0.0, clip=0.0), {}
    for c in chunks:
        if case.get("dropout"):                                  # (p, seed, step): TRAIN mode with the engine's masks of windows [c CH, (c + 1) CH)
            O.TRAIN_DROPOUT = tuple(case["dropout"]) + (c * CH,)
        chunk = {k: np.ascontiguousarray(v[c * CH:(c + 1) * CH]) for k, v in mb.items()}
        o, cache = O.modality_fwd(P, dims_run, chunk, is_lang)
        for k in sums:
            sums[k] += float(o[k])
        embs[c] = np.asarray(cache["emb"], np.float32)
        O.modality_bwd(P, G, dims_run, cache, is_lang, w_mod, w_clip)
        del cache
    return sums, {n: np.asarray(v, np.float32) for n, v in G.items()}, embs


def oracle_case(case, CH=4, workers=None, P=None, batch=None):
    """case: dict(seed, kind, rnn_type, max_window, B, S, [B_lang], [use_clip], mode, gscale).  Evaluates the step the engine runs —
    one modality ("vis", weight 1) or the pair vis + lang (weights 1/2 each, CLIP x 3 on the lang rows) — and returns
    (gradients {name: array}, {scope: dict(kl, action, total, clip)}, {scope: emb (B,S,128)}).  P / batch: the caller's own copies of
    `spec.init_all(case_dims(case), seed=case["seed"], ln_jitter=True)` / `case_batch(case)` — handed to the workers through memory-mapped files."""
    import multiprocessing as mp
    import numpy as np
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count() or 1
    Bl = case.get("B_lang", 0)
    nmod = 2 if Bl else 1
    jobs, meta = [], []
    nch = case["B"] // CH
    assert nch * CH == case["B"]
    clip = bool(case.get("use_clip", False)) and Bl > 0
    workers = workers or max(1, min(16, nch, cores // 8))
    threads = max(1, min(16, cores // (workers + (4 if clip else 0))))
    for w in range(workers):
        part = list(range(nch))[w::workers]
        if part:
            jobs.append((case, "vis", part, CH, 1.0 / nmod, 0.0, threads)); meta.append(("vis", nch))
    if Bl:
        if clip:       # all language rows in one evaluation (the contrastive loss is not a mean over windows); wide BLAS
            jobs.insert(0, (case, "lang", [0], Bl, 1.0 / nmod, 3.0, max(threads, min(64, cores // 2)))); meta.insert(0, ("lang", 1))
        else:
            nl = Bl // CH
            for w in range(workers):
                part = list(range(nl))[w::workers]
                if part:
                    jobs.append((case, "lang", part, CH, 1.0 / nmod, 0.0, threads)); meta.append(("lang", nl))
    shared = None
    if P is not None and batch is not None:
        shared = _share(case, P, batch)
        jobs = [j + (shared,) for j in jobs]
    try:
        if len(jobs) == 1:
            res = [_work(jobs[0])]
        else:
            with mp.get_context("spawn").Pool(len(jobs)) as pool:
                res = pool.map(_work, jobs, chunksize=1)
    finally:
        if shared:
            import shutil
            shutil.rmtree(shared[0], ignore_errors=True)
    G, losses, embs = {}, {}, {}
    for (scope, n), (sums, g, e) in zip(meta, res):
        for k, v in g.items():            # modality_bwd applied the modality weight; a chunk's gradient is of its own window mean -> / number of chunks
            G[k] = G.get(k, 0.0) + v.astype(np.float64) / n
        d = losses.setdefault(scope, dict(kl=0.0, action=0.0, total=0.0, clip=0.0))
        for k, v in sums.items():
            d[k] += v / n
        embs.setdefault(scope, {}).update(e)
    emb_out = {}
    for scope, e in embs.items():
        emb_out[scope] = np.concatenate([e[c] for c in sorted(e)], 0)
    return {n: np.asarray(v, np.float32) for n, v in G.items()}, losses, emb_out


def oracle_batch(seed, kind, max_window, Bt, St, CH=4, mode=None, gscale=1.0, workers=None, P=None, mb=None, cond=False, dropout=None):
    """(gradients, total loss, emb) of the vision-only batch `synthetic.make_batch(Bt, 0, St, seed=seed, edge_frac=0.05, aux_mask='all')` under
    `spec.init_all(dims, seed=seed, ln_jitter=True)`, evaluated by the numpy oracle in chunks of CH windows."""
    G, losses, embs = oracle_case(dict(seed=seed, kind=kind, max_window=max_window, B=Bt, S=St, mode=mode, gscale=gscale, cond=cond, dropout=dropout), CH, workers,
                                  P=P, batch=None if mb is None else {"vis": mb})
    return G, losses["vis"]["total"], embs["vis"]
