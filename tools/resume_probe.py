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
    def steps(m, opt, lo, hi, overflow_at=-1):
        eng = m.engine
        for i in range(lo, hi):
            opt.zero_grad()
            for sc, mb in synthetic.make_batch(2, 0, 4, seed=100 + i).items():
                eng.forward_loss(to_dev(mb), "lang" in sc, 1.0, 3.0, step=i)
                eng.backward()
            if i == overflow_at:
                eng.flat_grads[5] = float("inf")
            m._grads_reduced = True
            opt.step()
    m, opt = make(); p0 = m.engine.flat_params.clone(); steps(m, opt, 0, 6, overflow_at=1); ref = m.engine.flat_params.clone(); m.engine.close()
    ma, opta = make(); steps(ma, opta, 0, 6, overflow_at=1); ref2 = ma.engine.flat_params.clone(); ma.engine.close()
    m1, opt1 = make(); steps(m1, opt1, 0, 3, overflow_at=1); sd = opt1.state_dict(); params = m1.engine.flat_params.clone(); m1.engine.close()
    m2, opt2 = make(); m2.engine.flat_params.copy_(params); m2.engine.prepare_weights(); opt2.load_state_dict(sd); steps(m2, opt2, 3, 6)
    got = m2.engine.flat_params
    d = (ref - p0).double().norm()
    print("persist", persist, "repeat-run rel", ((ref2 - ref).double().norm() / d).item(), "resume rel", ((got - ref).double().norm() / d).item())
for p in (0, 1, 0, 1):
    run(p)
