"""tests/golden/state_manifest.json: the FULL state_dict() contract of the unmodified reference models — every key (parameters AND
buffers) with shape and dtype, the values of the small buffers, and which keys are parameters — for hulc (position table 32 and 64 rows),
gcbc and the mcil configuration (RNN / GRU).  It is what a Lightning checkpoint's "state_dict" of the reference holds
(LightningModule.state_dict = nn.Module.state_dict), so tests can build a reference-layout checkpoint dict and load it
(hulc/training.py:38-46, hulc/utils/utils.py:7-16).  Run in the build container only:  python tools/gen_golden_ckpt.py"""
from __future__ import annotations

import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_harness  # noqa: E402

out = {}
for name, kind, kw in (("hulc_w32", "hulc", dict(max_window=32, use_clip=True)), ("hulc_w64", "hulc", dict(max_window=64, use_clip=True)),
                       ("gcbc_w32", "gcbc", dict(max_window=32, use_clip=True)), ("mcil_w32", "mcil", dict(max_window=32)),
                       ("mcil_gru_w32", "mcil_gru", dict(max_window=32))):
    model = ref_harness.build_reference(kind, **kw)
    params = {n for n, _ in model.named_parameters()}
    entries = {}
    for k, v in model.state_dict().items():
        e = dict(shape=list(v.shape), dtype=str(v.dtype).replace("torch.", ""), param=k in params)
        if k not in params and v.numel() <= 512:
            e["value"] = v.detach().reshape(-1).tolist()
        entries[k] = e
    out[name] = dict(kind=kind, state_dict=entries, n_params=sum(int(p.numel()) for p in model.parameters()))
    print(name, len(entries), "keys,", out[name]["n_params"], "parameters,", sum(1 for e in entries.values() if not e["param"]), "buffers")
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "state_manifest.json"), "w"), indent=0, sort_keys=True)
