"""MiniLM sentence encoder.  CPU: the oracle against transformers.BertModel fixtures, the WordPiece tokenizer
against transformers.BertTokenizer.  GPU (-m gpu): hulc_sbert_encode through the C-ABI against the same fixture, and the drop-in
SBert class end to end from a local Hugging Face-layout directory."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import hulc_oracle as O  # noqa: E402
from hulc_amd import sbert as S  # noqa: E402

VOCAB = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "open", "the", "drawer", "draw", "##er", "push", "##ing", "block", "red", "blue", ",", ".", "!", "'", "s",
         "slide", "##r", "left", "right", "turn", "on", "off", "light", "##bulb", "un", "##aff", "##able", "cafe", "robot", "gripper", "a", "##b", "##c", "1", "2", "##3"]


def _fixture():
    fx = np.load(os.path.join(ROOT, "tests", "golden", "sbert_minilm.npz"))
    cfg = S.SBertConfig()
    return cfg, S.init_params(cfg, seed=int(fx["seed"])), fx


def test_oracle_matches_transformers_fixture():
    cfg, W, fx = _fixture()
    e, x = O.sbert_forward(W, fx["ids"], fx["mask"], heads=cfg.heads)
    assert np.abs(e - fx["emb"]).max() <= 2e-5
    m = fx["mask"] != 0
    assert np.abs(x[:, :, :8] - fx["hidden_sample"])[m].max() <= 2e-4
    assert np.allclose(np.linalg.norm(e, axis=-1), 1.0, atol=1e-5)


def test_wordpiece_tokenizer_matches_transformers(tmp_path):
    transformers = pytest.importorskip("transformers")
    vf = tmp_path / "vocab.txt"
    vf.write_text("\n".join(VOCAB) + "\n", encoding="utf-8")
    ref = transformers.BertTokenizer(str(vf), do_lower_case=True)
    tok = S.WordPieceTokenizer(str(vf))
    sents = ["Open the drawer.", "push the RED block, left!", "turn on the lightbulb", "unaffable café robot's gripper", "abc 123 xyz", "  pushing   slider\tright ", ""]
    for s_ in sents:
        assert tok.encode(s_, 128) == ref.encode(s_, add_special_tokens=True), s_
    ids, mask = tok.batch(sents, 8)
    assert ids.shape == mask.shape and ids.shape[1] <= 8 and (ids[mask == 0] == tok.pad).all()
    assert tok.encode("push " * 50, 16) == ref.encode("push " * 50, add_special_tokens=True, truncation=True, max_length=16)


@pytest.mark.gpu
def test_hip_encoder_matches_transformers_fixture():
    cfg, W, fx = _fixture()
    enc = S.SentenceEncoder(cfg, max_sentences=8)
    enc.load_state_dict(W)
    e = enc.encode_ids(fx["ids"], fx["mask"]).cpu().numpy()
    assert np.abs(e - fx["emb"]).max() <= 1e-4, np.abs(e - fx["emb"]).max()
    # padding must not matter: the same sentences padded to a longer length give the same embeddings
    pad = 9
    ids2 = np.concatenate([fx["ids"], np.zeros((fx["ids"].shape[0], pad), np.int32)], 1)
    mask2 = np.concatenate([fx["mask"], np.zeros((fx["mask"].shape[0], pad), np.int32)], 1)
    e2 = enc.encode_ids(ids2, mask2).cpu().numpy()
    assert np.abs(e2 - e).max() <= 2e-6
    with pytest.raises(RuntimeError):
        enc.encode_ids(np.zeros((9, 4), np.int32), np.ones((9, 4), np.int32))       # more sentences than the context was created for
    enc.close()


@pytest.mark.gpu
def test_sbert_class_end_to_end(tmp_path):
    """The drop-in SBert: local directory with vocab.txt + model.safetensors -> (B, 1, 384), equal to the oracle on the same tokens."""
    import torch
    from safetensors.numpy import save_file
    cfg = S.SBertConfig(layers=2, vocab=len(VOCAB))
    W = S.init_params(cfg, seed=9)
    d = tmp_path / "all-MiniLM-L6-v2"
    d.mkdir()
    (d / "vocab.txt").write_text("\n".join(VOCAB) + "\n", encoding="utf-8")
    save_file({("bert." + k): v for k, v in W.items()}, str(d / "model.safetensors"))
    with pytest.raises(FileNotFoundError):
        S.SBert("all-MiniLM-L6-v2", model_dir=str(tmp_path / "nowhere"))
    m = S.SBert("all-MiniLM-L6-v2", model_dir=str(d), max_sentences=4)
    assert m.cfg.layers == 2 and m.cfg.vocab == len(VOCAB) and m.cfg.heads == 12          # sizes are read from the checkpoint
    sents = ["open the drawer", "push the red block left!", "turn off the lightbulb."]
    out = m(sents)
    assert tuple(out.shape) == (3, 1, 384)
    ids, mask = m.tokenizer.batch(sents, 128)
    e, _ = O.sbert_forward(W, ids, mask, heads=cfg.heads)
    assert np.abs(out[:, 0].cpu().numpy() - e).max() <= 1e-4
    m.encoder.close()
