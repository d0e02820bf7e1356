"""Chunked evaluation of the numpy oracle in worker processes (spawned: the parent holds a HIP context).  Test infrastructure only.

Every loss of the step is a mean over windows, so losses and gradients of a batch are the means of its window chunks; a worker evaluates a
contiguous run of chunks with `oracle/hulc_oracle.py` (optionally in its rounding-aware mode) and returns the SUMS."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _work(args):
    seed, kind, max_window, Bt, St, CH, chunks, mode, gscale, threads = args
    os.environ["OMP_NUM_THREADS"] = os.environ["OPENBLAS_NUM_THREADS"] = os.environ["MKL_NUM_THREADS"] = str(threads)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import hulc_oracle as O
    from hulc_amd import spec
    from hulc_amd.utils import synthetic
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=threads)
    except Exception:
        pass
    dims = spec.ModelDims(kind=kind, max_window=max_window, use_clip=False)
    P = spec.init_all(dims, seed=seed, ln_jitter=True)
    mb = synthetic.make_batch(Bt, 0, St, seed=seed, edge_frac=0.05, aux_mask="all")["vis"]
    O.set_operand_rounding(mode, gscale)
    G, loss, embs = None, 0.0, {}
    for c in chunks:
        chunk = {"vis": {k: v[c * CH:(c + 1) * CH] for k, v in mb.items()}}
        l, g, caches = O.training_step(P, dims, chunk, keep_cache=True)
        loss += float(l["total"])
        embs[c] = np.asarray(caches["vis"]["emb"], np.float32)
        G = g if G is None else {n: G[n] + g[n] for n in g}
        del caches
    return loss, {n: np.asarray(v, np.float32) for n, v in G.items()}, embs


def oracle_batch(seed, kind, max_window, Bt, St, CH=4, mode=None, gscale=1.0, workers=None):
    """(gradients, total loss, emb) of the batch `synthetic.make_batch(Bt, 0, St, seed=seed, edge_frac=0.05, aux_mask='all')` under
    `spec.init_all(dims, seed=seed, ln_jitter=True)`, evaluated by the numpy oracle in chunks of CH windows."""
    import multiprocessing as mp
    import numpy as np
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count() or 1
    nch = Bt // CH
    workers = workers or min(8, nch, max(1, cores // 4))
    threads = max(1, min(16, cores // workers))
    parts = [list(range(nch))[w::workers] for w in range(workers)]
    jobs = [(seed, kind, max_window, Bt, St, CH, p, mode, gscale, threads) for p in parts if p]
    if len(jobs) == 1:
        res = [_work(jobs[0])]
    else:
        with mp.get_context("spawn").Pool(len(jobs)) as pool:
            res = pool.map(_work, jobs)
    loss = sum(r[0] for r in res) / nch
    G = {n: sum(r[1][n].astype(np.float64) for r in res) / nch for n in res[0][1]}
    embs = {}
    for r in res:
        embs.update(r[2])
    return {n: v.astype(np.float32) for n, v in G.items()}, loss, np.concatenate([embs[c] for c in range(nch)], 0)
