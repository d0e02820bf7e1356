"""Where does a 16-bit engine's gradient leave the rounding-aware oracle?  (GPU box; diagnostic behind tests/test_gpu_fullsize.py's gates)
B = 4 windows x S = 32: per-tensor rel-L2 of every gradient tensor + the named intermediate gradients (demb, dgoal, dseq_feat, dplan) for
the bf16 and fp16 engines against oracle/hulc_oracle.py in the matching operand-rounding mode.   python tools/parity_diag.py [B] [S]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import hulc_oracle as O
from hulc_amd import spec
from hulc_amd.engine import StepEngine
from hulc_amd.utils import synthetic

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
S = int(sys.argv[2]) if len(sys.argv) > 2 else 32
rel = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64)) / max(np.linalg.norm(np.asarray(b, np.float64)), 1e-30))
dims = spec.ModelDims(kind="hulc", max_window=32, use_clip=False)
P = spec.init_all(dims, seed=21, ln_jitter=True)
mb = synthetic.make_batch(B, 0, S, seed=21)["vis"]
dev_mb = {k: torch.from_numpy(v.astype(np.int32) if k == "plan_idx" else v).cuda() for k, v in mb.items()}
cap = {}
_sb, _gb, _prb = O.static_encoder_bwd, O.gripper_encoder_bwd, O.plan_recognition_bwd
def sb(P_, G, pre, c, dout): cap["demb_s"] = dout.copy(); return _sb(P_, G, pre, c, dout)
def gb(P_, G, pre, c, dout): cap["demb_g"] = dout.copy(); return _gb(P_, G, pre, c, dout)
def prb(P_, G, c, dlogits, dsf, heads=8, fc_state_used=True):
    cap["dpr_logits"] = None if dlogits is None else dlogits.copy()
    r = _prb(P_, G, c, dlogits, dsf, heads, fc_state_used); cap["demb_from_pr"] = r.copy(); return r
O.static_encoder_bwd, O.gripper_encoder_bwd, O.plan_recognition_bwd = sb, gb, prb
for dtype in ("bf16", "fp16"):
    gs = 8192.0 if dtype == "fp16" else 1.0
    O.set_operand_rounding(dtype, gs)
    lo, G, caches = O.training_step(P, dims, {"vis": mb}, keep_cache=True)
    O.set_operand_rounding(None)
    eng = StepEngine(dims, B, S, dtype=dtype, device="cuda:0", dropout_p=0.0, seed=3)
    if dtype == "fp16": eng.scaler_enable(init_scale=gs)
    eng.load_numpy(P)
    eng.zero_grads()
    l = eng.forward_loss(dev_mb, False, 1.0, 3.0, step=0)
    eng.backward()
    torch.cuda.synchronize()
    Gg = {n: t.detach().cpu().numpy() / gs for n, t in eng.views(eng.flat_grads).items()}
    demb = eng.get_tensor("demb", B * S * 128).reshape(B * S, 128) / gs
    demb_o = np.concatenate([cap["demb_s"], cap["demb_g"]], -1)
    print(f"== {dtype}: loss {l['total_mod']:.6f} vs {float(lo['total']):.6f}; demb static {rel(demb[:, :64], demb_o[:, :64]):.4f} gripper {rel(demb[:, 64:], demb_o[:, 64:]):.4f}")
    d3 = demb.reshape(B, S, 128); o3 = demb_o.reshape(B, S, 128)
    print("   demb by token group: t=0 %.4f, t=1..S-2 %.4f, t=S-1 %.4f" % (rel(d3[:, 0], o3[:, 0]), rel(d3[:, 1:-1], o3[:, 1:-1]), rel(d3[:, -1], o3[:, -1])))
    for nm, n in (("dgoal", B * 32), ("dseq_feat", B * 4096), ("dplan", B * 1024), ("dpr_logits", B * 1024)):
        try:
            t = eng.get_tensor(nm, n) / gs
            print(f"   {nm}: norm {np.linalg.norm(t):.4e}", ("rel vs oracle %.4f" % rel(t, cap["dpr_logits"].reshape(-1))) if nm == "dpr_logits" and cap.get("dpr_logits") is not None else "")
        except Exception as e:
            print("  ", nm, "unavailable:", e)
    errs = sorted(((rel(Gg[n], G[n]), n) for n in G if np.linalg.norm(G[n]) > 1e-9), reverse=True)
    for e, n in errs:
        print(f"   {e:.4f}  {n}")
    eng.close()
