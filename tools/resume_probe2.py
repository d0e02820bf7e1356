import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
from hulc_amd.hulc import Hulc
from hulc_amd.utils import synthetic
from test_gpu_parity import to_dev
def run(persist):
    def make():
        m = Hulc(precision=16, max_batch_size=2, max_seq_len=4, use_clip_auxiliary_loss=False, seed=11)
        m.engine.set_option("persistent_rnn", persist)
        m.engine.set_dropout(0.0)
        m.engine.scaler_enable(init_scale=1024.0)
        return m, m.configure_optimizers()["optimizer"]
    rec = {}
    def steps(m, opt, lo, hi, tag, overflow_at=-1):
        eng = m.engine
        for i in range(lo, hi):
            opt.zero_grad()
            for sc, mb in synthetic.make_batch(2, 0, 4, seed=100 + i).items():
                l = eng.forward_loss(to_dev(mb), "lang" in sc, 1.0, 3.0, step=i)
                eng.backward()
            torch.cuda.synchronize()
            G = {n: t.detach().clone() for n, t in eng.views(eng.flat_grads).items()}
            rec[(tag, i)] = (l, G, eng.flat_params.clone())
            if i == overflow_at:
                eng.flat_grads[5] = float("inf")
            m._grads_reduced = True
            opt.step()
    m, opt = make(); steps(m, opt, 0, 6, "ref", overflow_at=1); m.engine.close()
    m1, opt1 = make(); steps(m1, opt1, 0, 3, "a", overflow_at=1); sd = opt1.state_dict(); params = m1.engine.flat_params.clone(); m1.engine.close()
    m2, opt2 = make(); m2.engine.flat_params.copy_(params); m2.engine.prepare_weights(); opt2.load_state_dict(sd); steps(m2, opt2, 3, 6, "b")
    for i in range(6):
        tag = "a" if i < 3 else "b"
        lr, Gr, Pr = rec[("ref", i)]; l2, G2, P2 = rec[(tag, i)]
        worst = sorted(((float((G2[n] - Gr[n]).double().norm() / (Gr[n].double().norm() + 1e-30)), n) for n in Gr if float(Gr[n].double().norm()) > 0), reverse=True)[:3]
        print("persist", persist, "step", i, "loss", lr["action"], l2["action"], "params rel", float((P2 - Pr).double().norm() / Pr.double().norm()), "worst grads", [(round(a, 5), b[-40:]) for a, b in worst])
for p in (1, 0):
    run(p)
