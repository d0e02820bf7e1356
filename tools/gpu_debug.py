"""GPU bring-up report: runs one golden case through libhulc_hip (fp32 or bf16) and prints per-tensor errors
vs the numpy oracle.  Not a test (tests/test_gpu_parity.py asserts); used while debugging on the GPU box."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import hulc_oracle as O  # noqa: E402
from golden_util import load_case, rel_l2  # noqa: E402
from hulc_amd.engine import StepEngine  # noqa: E402


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
    case = sys.argv[1] if len(sys.argv) > 1 else "hulc_tiny"
    dtype = sys.argv[2] if len(sys.argv) > 2 else "fp32"
    dims, P, batch, fx = load_case(case)
    Bmax = max(mb["actions"].shape[0] for mb in batch.values())
    S = next(iter(batch.values()))["actions"].shape[1]
    eng = StepEngine(dims, Bmax, S, dtype=dtype, dropout_p=0.0)
    eng.load_numpy(P)
    print("workspace MB", eng.workspace_bytes() / 1e6)
    losses_o, G, caches = O.training_step(P, dims, batch, keep_cache=True)
    eng.zero_grads()
    nmod = len(batch)
    tot = 0.0
    for sc, mb in batch.items():
        is_lang = "lang" in sc
        t0 = time.time()
        l = eng.forward_loss(to_dev(mb, eng.device), is_lang, 1.0 / nmod, 3.0)
        B, S = mb["actions"].shape[:2]
        c = caches[sc]
        print(f"--- {sc}: losses gpu {l}  oracle action {losses_o['action_' + sc]:.6f} kl {losses_o['kl_' + sc]:.6f} clip {losses_o['clip_' + sc]:.6f}")

        def rep(name, ref, n=None, tf=None):
            got = eng.get_tensor(name, ref.size if n is None else n)
            if tf is not None:
                got = tf(got)
            got = got.reshape(ref.shape)
            print(f"   {name:14s} rel_l2 {rel_l2(got, ref):.3e}  maxabs {np.abs(got - ref).max():.3e}  (ref max {np.abs(ref).max():.3e})")

        N = B * S
        rep("s_a1", c["enc_s"]["a1"], tf=lambda g: g.reshape(N, 49, 49, 32).transpose(0, 3, 1, 2))
        rep("s_a2", c["enc_s"]["a2"], tf=lambda g: g.reshape(N, 23, 23, 64).transpose(0, 3, 1, 2))
        rep("s_a3", c["enc_s"]["a3"], tf=lambda g: g.reshape(N, 21, 21, 64).transpose(0, 3, 1, 2))
        rep("s_ss", c["enc_s"]["ss"])
        rep("s_f1", c["enc_s"]["f1"])
        rep("s_f2", c["enc_s"]["f2"])
        rep("g_a3", c["enc_g"]["a3"], tf=lambda g: g.reshape(N, 7, 7, 64).transpose(0, 3, 1, 2))
        rep("g_g0", c["enc_g"]["g0"])
        rep("emb", c["emb"])
        rep("goal", c["goal"])
        rep("pr_x0", c["pr"]["layers"][0]["x_in"])
        rep("pr_x1", c["pr"]["layers"][1]["x_in"])
        rep("pr_x_final", c["pr"]["x_final"])
        rep("seq_feat", c["seq_feat"])
        rep("pr_logits", c["pr_logits"])
        if dims.kind == "hulc":
            rep("pp_logits", c["pp_logits"])
        H1 = c["dec"]["H1"]  # (B,S,H) -> time-major
        rep("dec_h0", c["dec"]["rc"]["H0"].transpose(1, 0, 2))
        rep("dec_h1", H1.transpose(1, 0, 2))
        heads_ref = np.concatenate([c["dec"]["probs"].reshape(B, S, 60), c["dec"]["means"].reshape(B, S, 60),
                                    c["dec"]["log_scales"].reshape(B, S, 60), c["dec"]["gripper"]], -1).transpose(1, 0, 2)
        rep("heads", heads_ref, n=S * B * 192, tf=lambda g: g.reshape(S * B, 192)[:, :182])
        rep("a_tcp", c["dec"]["a_tcp"])
        eng.backward()
        torch.cuda.synchronize()
        print(f"   fwd+bwd wall {time.time() - t0:.3f}s")
        tot += l["total_mod"] / nmod + 3.0 * l["clip"]
    print(f"TOTAL gpu {tot:.6f} oracle {losses_o['total']:.6f} ref {float(fx['loss_total']):.6f}")
    gv = eng.views(eng.flat_grads)
    worst = []
    for n, g in G.items():
        got = gv[n].detach().cpu().numpy()
        e = rel_l2(got, g)
        worst.append((e, n, float(np.linalg.norm(g)), float(np.abs(got - g.reshape(got.shape)).max())))
    worst.sort(reverse=True)
    print("worst gradient tensors (rel_l2, name, |ref|, maxabs):")
    for w in worst[:25]:
        print("   %.3e  %-70s %.3e %.3e" % w)
    print("median rel_l2 %.3e" % np.median([w[0] for w in worst]))
    # Adam
    st = {}
    O.adam_step(P, G, st, 1)
    eng.adam_step()
    pv = eng.views(eng.flat_params)
    errs = []
    for n in P:
        got = pv[n].detach().cpu().numpy()
        errs.append((float(np.abs(got - P[n].reshape(got.shape)).max()), n))
    errs.sort(reverse=True)
    print("adam worst:", errs[:5])


if __name__ == "__main__":
    main()
