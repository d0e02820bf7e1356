"""Throughput of the MiniLM sentence encoder on the HIP path, with transformers.BertModel on the host cores beside it
(same seeded synthetic weights of the real all-MiniLM-L6-v2 sizes; the published checkpoint is not in this image).  CALVIN's task annotations are
short (<= 16 word pieces); the reference encodes them once per dataset (language_network.py:8-17), so this is not on the training step's path.
    python tools/time_sbert.py  ->  one JSON line"""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hulc_amd import sbert as S

cfg = S.SBertConfig()
W = S.init_params(cfg, seed=7)
rows = []
for B, Ln in ((64, 16), (64, 32), (64, 128)):
    rng = np.random.default_rng(B + Ln)
    ids = rng.integers(1000, cfg.vocab, size=(B, Ln)).astype(np.int32)
    lens = rng.integers(max(4, Ln // 2), Ln + 1, size=B)
    mask = (np.arange(Ln)[None, :] < lens[:, None]).astype(np.int32)
    ids = ids * mask
    enc = S.SentenceEncoder(cfg, max_sentences=B)
    enc.load_state_dict(W)
    e = enc.encode_ids(ids, mask)
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        e = enc.encode_ids(ids, mask)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    row = {"sentences": B, "tokens": Ln, "hip_ms": round(dt * 1e3, 3), "hip_sentences_per_s": round(B / dt, 1)}
    enc.close()
    try:
        import transformers
        torch.set_num_threads(8)
        hc = transformers.BertConfig(vocab_size=cfg.vocab, hidden_size=cfg.hidden, num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads,
                                     intermediate_size=cfg.intermediate, max_position_embeddings=cfg.max_position, layer_norm_eps=cfg.ln_eps)
        m = transformers.BertModel(hc, add_pooling_layer=False).eval()
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in W.items()}, strict=False)
        ti, tm = torch.from_numpy(ids.astype(np.int64)), torch.from_numpy(mask.astype(np.int64))
        with torch.no_grad():
            def f():
                h = m(input_ids=ti, attention_mask=tm).last_hidden_state
                mk = tm[:, :, None].float()
                v = (h * mk).sum(1) / mk.sum(1).clamp(min=1e-9)
                return torch.nn.functional.normalize(v, dim=1)
            ref = f()
            t0 = time.perf_counter()
            for _ in range(3):
                ref = f()
            dc = (time.perf_counter() - t0) / 3
        row.update(cpu_ms=round(dc * 1e3, 1), cpu_sentences_per_s=round(B / dc, 1), cpu_threads=8,
                   max_abs_diff=float((e.cpu().double() - ref.double()).abs().max()), emb_abs_mean=float(ref.abs().mean()))
    except Exception as ex:      # transformers absent / API drift: the HIP numbers stand on their own
        row["cpu_error"] = str(ex)[:120]
    rows.append(row)
print(json.dumps({"metric": "MiniLM-L6 sentence embeddings/s (fp32, seeded synthetic weights)", "rows": rows}))
