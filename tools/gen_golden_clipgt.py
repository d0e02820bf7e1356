"""Generate tests/golden/clipgt_hulc.npz by running the UNMODIFIED reference (/root/reference) on CPU: `Hulc.on_fit_start` (hulc.py:697-737) on a
small synthetic language-annotation dataset written to a temporary directory in the CALVIN layout (auto_lang_ann.npy / embeddings.npy), then
`on_validation_epoch_start` (hulc.py:967-974) and the lang branch of `validation_step` up to `clip_groundtruth` (hulc.py:804-808, 980-1043).  The
fixture holds the annotation data (strings + embeddings), the batch seed, and the four `lang_gt/*` metrics the reference logged.

Run in the build container only:  python tools/gen_golden_clipgt.py
"""
from __future__ import annotations

import os
import pathlib
import sys
import tempfile
import types
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
warnings.filterwarnings("ignore")

from hulc_amd import spec  # noqa: E402
from hulc_amd.utils import portable_rng as prng  # noqa: E402
from hulc_amd.utils import synthetic  # noqa: E402
import ref_harness  # noqa: E402
from gen_golden import to_ref_batch  # noqa: E402
from gen_golden_val import RandRecorder, load_params  # noqa: E402

SEED, BL, S = 16, 8, 4
IDX = [3, 0, 7, 5, 1, 6, 2, 4]                       # episode indices of the validation batch (dataset order is not batch order)
MASK = [True, True, False, True, True, True, False, True]
TASKS = ["open_drawer", "close_drawer", "lift_red_block", "push_block_left", "turn_on_led"]


def annotation_data():
    """A miniature of the CALVIN language annotations: 11 training annotations over 5 tasks (two instructions appear twice), a validation
    annotation file with one task per annotated episode, and the validation instruction embeddings keyed by task (one task is unknown to training)."""
    tr_ann = ["pull the drawer open", "open the drawer", "shut the drawer", "close the drawer", "lift the red block", "pick up the red block",
              "push the block to the left", "slide the block left", "switch on the led", "open the drawer", "lift the red block"]
    tr_task = [TASKS[0], TASKS[0], TASKS[1], TASKS[1], TASKS[2], TASKS[2], TASKS[3], TASKS[3], TASKS[4], TASKS[0], TASKS[2]]
    tr_emb = prng.normal("clipgt.train_emb", (len(tr_ann), 1, 384), 1.0, SEED).astype(np.float32)
    tr_emb /= np.linalg.norm(tr_emb, axis=-1, keepdims=True)
    va_task = [TASKS[1], TASKS[3], TASKS[0], TASKS[4], TASKS[2], TASKS[3], TASKS[0], TASKS[2], TASKS[4]]   # per annotated validation episode
    lookup = [2, 0, 5, 3, 8, 6, 4, 1]                                                        # dataset index -> row of the validation annotation file
    val_instr = {TASKS[0]: ["open the drawer"], TASKS[1]: ["close the drawer"], TASKS[2]: ["lift the red block"], TASKS[3]: ["push the block left"],
                 TASKS[4]: ["turn on the led"], "rotate_pink_block": ["rotate the pink block"]}
    va_emb = {}
    for i, (t, ins) in enumerate(val_instr.items()):
        e = prng.normal("clipgt.val_emb." + t, (1, 1, 384), 1.0, SEED).astype(np.float32)
        va_emb[t] = {"emb": e / np.linalg.norm(e, axis=-1, keepdims=True), "ann": ins}
    return tr_ann, tr_task, tr_emb, va_task, lookup, val_instr, va_emb


def main():
    out = os.path.join(ROOT, "tests", "golden", "clipgt_hulc.npz")
    tr_ann, tr_task, tr_emb, va_task, lookup, val_instr, va_emb = annotation_data()
    dims = spec.ModelDims(kind="hulc", max_window=32, use_clip=True)
    P = spec.init_all(dims, seed=SEED, ln_jitter=True)
    batch = synthetic.make_batch(0, BL, S, seed=SEED, edge_frac=0.05, aux_mask="some")
    ref_harness.install_stubs()
    cfg = ref_harness.model_cfg("hulc", max_window=32, use_clip=True)
    cfg["val_instructions"] = ref_harness.to_cfg(val_instr)
    from hulc.models.hulc import Hulc
    model = Hulc(**cfg)
    model.eval()
    load_params(model, P)
    with tempfile.TemporaryDirectory() as tmp:
        root = pathlib.Path(tmp)
        for split in ("training", "validation"):
            (root / split / "lang_annotations").mkdir(parents=True)
        np.save(root / "training" / "lang_annotations" / "auto_lang_ann.npy", {"language": {"ann": tr_ann, "task": tr_task, "emb": tr_emb}}, allow_pickle=True)
        np.save(root / "validation" / "lang_annotations" / "auto_lang_ann.npy", {"language": {"ann": ["-"] * len(va_task), "task": va_task, "emb": np.zeros((len(va_task), 1, 384), np.float32)}},
                allow_pickle=True)
        np.save(root / "validation" / "lang_annotations" / "embeddings.npy", va_emb, allow_pickle=True)
        ds_tr = types.SimpleNamespace(abs_datasets_dir=root / "training", lang_folder="lang_annotations")
        ds_va = types.SimpleNamespace(abs_datasets_dir=root / "validation", lang_folder="lang_annotations", lang_lookup=lookup)
        model.trainer = types.SimpleNamespace(datamodule=types.SimpleNamespace(train_datasets={"lang": ds_tr}, val_datasets={"lang": ds_va}, modalities=["lang"]))
        model.on_fit_start()
    model.current_epoch = 0
    rb = to_ref_batch(batch)["lang"]
    rb["idx"] = torch.tensor(IDX)
    rb["use_for_aux_lang_loss"] = torch.tensor(MASK)
    torch.manual_seed(99)
    with torch.no_grad():
        model.on_validation_epoch_start()
        emb = model.perceptual_encoder(rb["rgb_obs"], rb["depth_obs"], rb["robot_obs"])
        goal = model.language_goal(rb["lang"])
        with RandRecorder():
            res = model.lmp_val(emb, goal, rb["actions"], rb["state_info"]["robot_obs"])
        seq_feat = res[-1]
        model.clip_groundtruth(seq_feat, rb["idx"], rb["use_for_aux_lang_loss"])
    fx = dict(meta=np.array([BL, S, SEED], np.int64), train_ann=np.array(tr_ann), train_task=np.array(tr_task), train_emb=tr_emb, val_task=np.array(va_task),
              lang_lookup=np.array(lookup, np.int64), val_instr_task=np.array(list(val_instr)), val_instr_text=np.array([v[0] for v in val_instr.values()]),
              val_emb=np.stack([va_emb[t]["emb"] for t in val_instr]), seq_feat=seq_feat.numpy(), use_for_aux=rb["use_for_aux_lang_loss"].numpy(), idx=np.array(IDX, np.int64),
              n_train_unique=np.int64(model.train_lang_emb.shape[0]), n_val=np.int64(model.val_lang_emb.shape[0]))
    for k in ("lang_gt/train_gt", "lang_gt/val_gt", "lang_gt/train_sr", "lang_gt/val_sr"):
        fx[k.replace("/", "__")] = np.float64(model.logged[k])
    np.savez_compressed(out, **fx)
    print("wrote", out, {k: float(v) for k, v in fx.items() if k.startswith("lang_gt")}, "unique train instructions", int(fx["n_train_unique"]), "val", int(fx["n_val"]),
          "aux mask", fx["use_for_aux"])


if __name__ == "__main__":
    main()
