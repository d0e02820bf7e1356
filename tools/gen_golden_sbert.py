"""tests/golden/sbert_minilm.npz: transformers.BertModel (the library sentence_transformers' all-MiniLM-L6-v2 wraps) with the seeded
synthetic weights of hulc_amd.sbert.init_params, + mean pooling + L2 normalisation as sentence_transformers' Pooling / Normalize
modules do them.  Inputs (token ids, attention mask) and outputs only; the weights are regenerated from the seed.
Run in the build container only (needs `transformers`)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transformers import BertConfig, BertModel  # noqa: E402

import importlib.util  # noqa: E402
spec_ = importlib.util.spec_from_file_location("sbert_host", os.path.join(ROOT, "hulc_amd", "sbert.py"))

# hulc_amd.sbert imports the HIP library lazily only inside SentenceEncoder; param_table / init_params are pure numpy
sys.modules.setdefault("hulc_amd", __import__("hulc_amd"))
from hulc_amd import sbert as S  # noqa: E402

cfg = S.SBertConfig()
W = S.init_params(cfg, seed=3)
hf = BertConfig(vocab_size=cfg.vocab, hidden_size=cfg.hidden, num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads, intermediate_size=cfg.intermediate,
                max_position_embeddings=cfg.max_position, type_vocab_size=cfg.type_vocab, layer_norm_eps=cfg.ln_eps, hidden_act="gelu",
                hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
model = BertModel(hf, add_pooling_layer=False).eval()
sd = model.state_dict()
missing = [k for k in sd if k not in W and "position_ids" not in k]
assert not missing, missing
model.load_state_dict({k: torch.from_numpy(W[k]) for k in W}, strict=False)
rng = np.random.default_rng(5)
B, Ln = 5, 19
ids = rng.integers(1000, cfg.vocab, (B, Ln)).astype(np.int64)
lens = np.array([19, 7, 12, 3, 19])
mask = (np.arange(Ln)[None] < lens[:, None]).astype(np.int64)
ids = np.where(mask != 0, ids, 0)
with torch.no_grad():
    h = model(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask)).last_hidden_state
    m = torch.from_numpy(mask).unsqueeze(-1).float()
    emb = (h * m).sum(1) / torch.clamp(m.sum(1), min=1e-9)                      # sentence_transformers Pooling (mean)
    emb = torch.nn.functional.normalize(emb, p=2, dim=1)                        # sentence_transformers Normalize
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "sbert_minilm.npz"), ids=ids.astype(np.int32), mask=mask.astype(np.int32), emb=emb.numpy(),
                    hidden_sample=h.numpy()[:, :, :8], seed=np.int32(3))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import hulc_oracle as O  # noqa: E402
e, x = O.sbert_forward(W, ids, mask, heads=cfg.heads)
print("oracle vs transformers: emb", np.abs(e - emb.numpy()).max(), "hidden", np.abs(x[:, :, :8] - h.numpy()[:, :, :8])[mask != 0].max())
