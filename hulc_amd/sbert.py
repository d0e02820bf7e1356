"""MiniLM sentence encoder on the HIP path — the reference's `SBert` (hulc/models/encoders/language_network.py:8-17,
conf/model/sbert.yaml: sentence_transformers "all-MiniLM-L6-v2") without sentence_transformers / transformers at run time:

    SBert(nlp_model)(["open the drawer", ...]) -> (B, 1, 384) tensor            # same call surface as the reference

= WordPiece tokenizer (host, pure Python: BasicTokenizer + greedy longest-match, the BERT uncased recipe) -> `hulc_sbert_encode`
(BERT encoder, masked mean pooling, L2 normalisation: hulc_amd/csrc/sbert.h).  Weights are NOT in this repository (no network in the
build container): `SBert` loads a local Hugging Face layout (vocab.txt + model.safetensors / pytorch_model.bin) from
`$HULC_SBERT_DIR/<nlp_model>` or the path given, and raises FileNotFoundError otherwise.  Parity of the encoder is pinned against
`transformers.BertModel` with seeded random weights (tests/golden/sbert_minilm.npz), the tokenizer against `transformers.BertTokenizer`.
"""
from __future__ import annotations

import ctypes as C
import os
import unicodedata
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import lib as L


@dataclass(frozen=True)
class SBertConfig:            # all-MiniLM-L6-v2
    layers: int = 6
    hidden: int = 384
    heads: int = 12
    intermediate: int = 1536
    vocab: int = 30522
    max_position: int = 512
    type_vocab: int = 2
    ln_eps: float = 1e-12
    normalize: bool = True
    max_seq_length: int = 128     # sentence_transformers truncates at 256 for this model; the kernel handles <= 128 tokens


def param_table(cfg: SBertConfig) -> List[Tuple[str, Tuple[int, ...]]]:
    """transformers.BertModel(add_pooling_layer=False).state_dict() names and shapes."""
    H, I = cfg.hidden, cfg.intermediate
    t = [("embeddings.word_embeddings.weight", (cfg.vocab, H)), ("embeddings.position_embeddings.weight", (cfg.max_position, H)),
         ("embeddings.token_type_embeddings.weight", (cfg.type_vocab, H)), ("embeddings.LayerNorm.weight", (H,)), ("embeddings.LayerNorm.bias", (H,))]
    for l in range(cfg.layers):
        p = f"encoder.layer.{l}."
        for n in ("query", "key", "value"):
            t += [(p + f"attention.self.{n}.weight", (H, H)), (p + f"attention.self.{n}.bias", (H,))]
        t += [(p + "attention.output.dense.weight", (H, H)), (p + "attention.output.dense.bias", (H,)),
              (p + "attention.output.LayerNorm.weight", (H,)), (p + "attention.output.LayerNorm.bias", (H,)),
              (p + "intermediate.dense.weight", (I, H)), (p + "intermediate.dense.bias", (I,)),
              (p + "output.dense.weight", (H, I)), (p + "output.dense.bias", (H,)), (p + "output.LayerNorm.weight", (H,)), (p + "output.LayerNorm.bias", (H,))]
    return t


def init_params(cfg: SBertConfig, seed: int = 0) -> Dict[str, np.ndarray]:
    """Seeded synthetic weights (portable RNG, BERT's N(0, 0.02) init scaled up so attention is not uniform; LayerNorm scales jittered)."""
    from .utils import portable_rng as prng
    out = {}
    for n, shape in param_table(cfg):
        if "LayerNorm.weight" in n:
            out[n] = (1.0 + 0.1 * prng.normal("sbert." + n, shape, seed=seed)).astype(np.float32)
        elif n.endswith(".bias"):
            out[n] = (0.02 * prng.normal("sbert." + n, shape, seed=seed)).astype(np.float32)
        else:
            out[n] = (0.06 * prng.normal("sbert." + n, shape, seed=seed)).astype(np.float32)
    return out


class SentenceEncoder:
    """Owner of one hulc_sbert context + the flat fp32 weight buffer."""

    def __init__(self, cfg: SBertConfig = SBertConfig(), max_sentences: int = 64, device: str = "cuda:0"):
        self.lib = L.load()
        if not torch.cuda.is_available():
            raise RuntimeError("hulc_amd.sbert needs a HIP device; no CPU fallback")
        self.cfg, self.device = cfg, torch.device(device)
        torch.cuda.set_device(self.device)
        self.max_sentences = max_sentences
        c = L.HulcSbertConfig(layers=cfg.layers, hidden=cfg.hidden, heads=cfg.heads, intermediate=cfg.intermediate, vocab=cfg.vocab,
                              max_position=cfg.max_position, max_sentences=max_sentences, max_tokens=min(128, cfg.max_seq_length),
                              normalize=int(cfg.normalize), ln_eps=cfg.ln_eps)
        self.ctx = C.c_void_p()
        L.check(self.lib.hulc_sbert_create(C.byref(c), C.byref(self.ctx)))
        L.check(self.lib.hulc_sbert_set_stream(self.ctx, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        self.table = param_table(cfg)
        self.offsets, off = {}, 0
        for n, shape in self.table:
            self.offsets[n] = off
            off += (int(np.prod(shape)) + 63) // 64 * 64
        self.flat = torch.zeros(off, dtype=torch.float32, device=self.device)

    def load_state_dict(self, sd: Dict) -> None:
        """sd: BertModel state_dict (numpy arrays or tensors); a leading "bert." / "0.auto_model." prefix is stripped."""
        clean = {}
        for k, v in sd.items():
            for pre in ("bert.", "0.auto_model.", "auto_model."):
                if k.startswith(pre):
                    k = k[len(pre):]
            clean[k] = v
        for n, shape in self.table:
            if n not in clean:
                raise KeyError(f"sentence encoder weight {n} missing")
            v = clean[n]
            t = (v if torch.is_tensor(v) else torch.from_numpy(np.ascontiguousarray(v))).to(torch.float32).reshape(-1)
            if t.numel() != int(np.prod(shape)):
                raise ValueError(f"{n}: expected shape {shape}")
            self.flat[self.offsets[n]: self.offsets[n] + t.numel()].copy_(t)
        names = (C.c_char_p * len(self.table))(*[n.encode() for n, _ in self.table])
        offs = (C.c_int64 * len(self.table))(*[self.offsets[n] for n, _ in self.table])
        nums = (C.c_int64 * len(self.table))(*[int(np.prod(s)) for _, s in self.table])
        L.check(self.lib.hulc_sbert_bind(self.ctx, self.flat.data_ptr(), self.flat.numel(), len(self.table), names, offs, nums))

    def encode_ids(self, ids: np.ndarray, mask: np.ndarray) -> torch.Tensor:
        ids = np.ascontiguousarray(ids, np.int32)
        mask = np.ascontiguousarray(mask, np.int32)
        B, Ln = ids.shape
        out = torch.zeros(B, self.cfg.hidden, dtype=torch.float32, device=self.device)
        L.check(self.lib.hulc_sbert_encode(self.ctx, ids.ctypes.data, mask.ctypes.data, B, Ln, out.data_ptr()))
        return out

    def close(self):
        if self.ctx:
            self.lib.hulc_sbert_destroy(self.ctx)
            self.ctx = None


# ---------------------------------------------------------------------------------------------------------------------
# BERT uncased tokenizer (the algorithm of google-research/bert tokenization.py; checked against transformers.BertTokenizer)
# ---------------------------------------------------------------------------------------------------------------------
def _is_punct(ch: str) -> bool:
    cp = ord(ch)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:
        return True
    return unicodedata.category(ch).startswith("P")


def _is_cjk(cp: int) -> bool:
    return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF or 0x2A700 <= cp <= 0x2B73F or 0x2B740 <= cp <= 0x2B81F
            or 0x2B820 <= cp <= 0x2CEAF or 0xF900 <= cp <= 0xFAFF or 0x2F800 <= cp <= 0x2FA1F)


class WordPieceTokenizer:
    def __init__(self, vocab_file: str, do_lower_case: bool = True, max_input_chars_per_word: int = 100):
        with open(vocab_file, encoding="utf-8") as f:
            self.vocab = {tok.rstrip("\n"): i for i, tok in enumerate(f)}
        self.lower = do_lower_case
        self.max_chars = max_input_chars_per_word
        self.unk, self.cls, self.sep, self.pad = (self.vocab[t] for t in ("[UNK]", "[CLS]", "[SEP]", "[PAD]"))

    def _basic(self, text: str) -> List[str]:
        out = []
        for ch in text:                                  # clean + pad CJK
            cp = ord(ch)
            if cp == 0 or cp == 0xFFFD or (unicodedata.category(ch) in ("Cc", "Cf") and ch not in "\t\n\r"):
                continue
            if _is_cjk(cp):
                out.append(f" {ch} ")
            elif ch in " \t\n\r" or unicodedata.category(ch) == "Zs":
                out.append(" ")
            else:
                out.append(ch)
        toks = []
        for tok in "".join(out).split():
            if self.lower:
                tok = "".join(c for c in unicodedata.normalize("NFD", tok.lower()) if unicodedata.category(c) != "Mn")
            cur = ""
            for ch in tok:                               # split on punctuation
                if _is_punct(ch):
                    if cur:
                        toks.append(cur)
                        cur = ""
                    toks.append(ch)
                else:
                    cur += ch
            if cur:
                toks.append(cur)
        return toks

    def _wordpiece(self, word: str) -> List[int]:
        if len(word) > self.max_chars:
            return [self.unk]
        ids, start = [], 0
        while start < len(word):
            end, cur = len(word), None
            while start < end:
                sub = ("##" if start > 0 else "") + word[start:end]
                if sub in self.vocab:
                    cur = self.vocab[sub]
                    break
                end -= 1
            if cur is None:
                return [self.unk]
            ids.append(cur)
            start = end
        return ids

    def encode(self, text: str, max_len: int) -> List[int]:
        ids = [i for w in self._basic(text) for i in self._wordpiece(w)]
        return [self.cls] + ids[: max_len - 2] + [self.sep]

    def batch(self, texts: Sequence[str], max_len: int) -> Tuple[np.ndarray, np.ndarray]:
        enc = [self.encode(t, max_len) for t in texts]
        Ln = max(len(e) for e in enc)
        ids = np.full((len(enc), Ln), self.pad, np.int32)
        mask = np.zeros((len(enc), Ln), np.int32)
        for i, e in enumerate(enc):
            ids[i, : len(e)] = e
            mask[i, : len(e)] = 1
        return ids, mask


class SBert(torch.nn.Module):
    """Drop-in for hulc.models.encoders.language_network.SBert: forward(list of sentences) -> (B, 1, 384)."""

    def __init__(self, nlp_model: str, model_dir: Optional[str] = None, device: str = "cuda:0", max_sentences: int = 64):
        super().__init__()
        assert isinstance(nlp_model, str)
        root = model_dir or os.path.join(os.environ.get("HULC_SBERT_DIR", os.path.expanduser("~/.cache/torch/sentence_transformers")), nlp_model.replace("/", "_"))
        vocab = os.path.join(root, "vocab.txt")
        if not os.path.exists(vocab):
            raise FileNotFoundError(f"sentence encoder files for {nlp_model!r} not found under {root} (vocab.txt + model.safetensors / pytorch_model.bin); "
                                    "there is no network access and no weights ship with this repository")
        self.tokenizer = WordPieceTokenizer(vocab)
        sd = None
        for fn in ("model.safetensors", "0_Transformer/model.safetensors"):
            if sd is None and os.path.exists(os.path.join(root, fn)):
                from safetensors.numpy import load_file
                sd = load_file(os.path.join(root, fn))
        for fn in ("pytorch_model.bin", "0_Transformer/pytorch_model.bin"):
            if sd is None and os.path.exists(os.path.join(root, fn)):
                sd = torch.load(os.path.join(root, fn), map_location="cpu", weights_only=True)
        if sd is None:
            raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin under {root}")
        layers = 1 + max(int(k.split("encoder.layer.")[1].split(".")[0]) for k in sd if "encoder.layer." in k)
        shp = {k.split("bert.")[-1].split("auto_model.")[-1]: tuple(v.shape) for k, v in sd.items()}     # sizes come from the checkpoint
        vocab, hidden = shp["embeddings.word_embeddings.weight"]
        self.cfg = SBertConfig(layers=layers, hidden=hidden, vocab=vocab, max_position=shp["embeddings.position_embeddings.weight"][0],
                               type_vocab=shp["embeddings.token_type_embeddings.weight"][0], intermediate=shp["encoder.layer.0.intermediate.dense.weight"][0],
                               heads=hidden // 32)
        self.encoder = SentenceEncoder(self.cfg, max_sentences=max_sentences, device=device)
        self.encoder.load_state_dict(sd)

    def forward(self, x: List[str]) -> torch.Tensor:
        embs = []
        for i in range(0, len(x), self.encoder.max_sentences):
            ids, mask = self.tokenizer.batch(x[i: i + self.encoder.max_sentences], min(128, self.cfg.max_seq_length))
            embs.append(self.encoder.encode_ids(ids, mask))
        return torch.unsqueeze(torch.cat(embs, 0), 1)
