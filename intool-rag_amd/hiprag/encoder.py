"""
HipEncoder -- Python handle on libhiprag's XLM-RoBERTa-shaped batch encoder (csrc/encoder.hip).

PyTorch-ROCm only HOLDS the weights (bf16 matrices, fp32 biases / LayerNorm parameters, on the GPU) and hands their
data_ptr()s to the library; every FLOP runs in hand-written HIP kernels.  Stands where the reference keeps
LangChain's HuggingFaceEmbeddings / sentence-transformers (rag/providers/hf/embeddings.py:32-35).
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _native as nat


@dataclass
class EncoderConfig:
    """Defaults = XLM-RoBERTa-large, the architecture of BAAI/bge-m3 and BAAI/bge-reranker-v2-m3
    (rag/config.py:9 EMBEDDING_MODEL, :25 RERANKER_MODEL)."""
    vocab: int = 250002
    hidden: int = 1024
    layers: int = 24
    heads: int = 16
    ffn: int = 4096
    max_pos: int = 8194
    pad_id: int = 1
    ln_eps: float = 1e-5
    bos_id: int = 0
    eos_id: int = 2
    max_seq_len: int = 512


def random_state(cfg: EncoderConfig, seed: int = 0, with_head: bool = False, std: float = 0.02) -> Dict[str, "object"]:
    """Seeded random weights under transformers' XLM-R parameter names (no checkpoint exists offline)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    H, F = cfg.hidden, cfg.ffn

    def w(*shape):
        return torch.randn(*shape, generator=g) * std

    sd = {
        "embeddings.word_embeddings.weight": w(cfg.vocab, H),
        "embeddings.position_embeddings.weight": w(cfg.max_pos, H),
        "embeddings.token_type_embeddings.weight": w(1, H),
        "embeddings.LayerNorm.weight": 1.0 + w(H),
        "embeddings.LayerNorm.bias": w(H),
    }
    for i in range(cfg.layers):
        p = f"encoder.layer.{i}."
        for name, shape in (("attention.self.query", (H, H)), ("attention.self.key", (H, H)),
                            ("attention.self.value", (H, H)), ("attention.output.dense", (H, H)),
                            ("intermediate.dense", (F, H)), ("output.dense", (H, F))):
            sd[p + name + ".weight"] = w(*shape)
            sd[p + name + ".bias"] = w(shape[0])
        for name in ("attention.output.LayerNorm", "output.LayerNorm"):
            sd[p + name + ".weight"] = 1.0 + w(H)
            sd[p + name + ".bias"] = w(H)
    if with_head:
        sd["classifier.dense.weight"] = w(H, H)
        sd["classifier.dense.bias"] = w(H)
        sd["classifier.out_proj.weight"] = w(1, H)
        sd["classifier.out_proj.bias"] = w(1)
    return sd


class _Cfg(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("vocab", "hidden", "layers", "heads", "ffn", "max_pos", "pad_id")] + \
               [("ln_eps", ctypes.c_float)]


class _Layer(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("wqkv", "bqkv", "wo", "bo", "ln1_g", "ln1_b", "w1", "b1", "w2", "b2",
                                               "ln2_g", "ln2_b")]


class _Weights(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("word_emb", "pos_emb", "type_emb", "emb_ln_g", "emb_ln_b")] + \
               [("layers", ctypes.POINTER(_Layer))] + \
               [(n, ctypes.c_void_p) for n in ("cls_dense_w", "cls_dense_b", "cls_out_w", "cls_out_b")]


class HipEncoder:
    def __init__(self, cfg: EncoderConfig, state: Optional[Dict[str, "object"]] = None, device: int = 0, seed: int = 0,
                 with_head: bool = False):
        import torch
        self.cfg = cfg
        self.device = int(device)
        dev = torch.device("cuda", self.device)
        if state is None:
            state = random_state(cfg, seed, with_head)
        state = {k.replace("roberta.", "", 1) if k.startswith("roberta.") else k: v for k, v in state.items()}
        self._t: List["torch.Tensor"] = []     # keeps every device tensor alive for the lifetime of the handle

        def mat(t):
            t = t.detach().to(device=dev, dtype=torch.bfloat16).contiguous()
            self._t.append(t)
            return t.data_ptr()

        def vec(t):
            t = t.detach().to(device=dev, dtype=torch.float32).contiguous()
            self._t.append(t)
            return t.data_ptr()

        layers = (_Layer * cfg.layers)()
        for i in range(cfg.layers):
            p = f"encoder.layer.{i}."
            wq, wk, wv = (state[p + f"attention.self.{n}.weight"] for n in ("query", "key", "value"))
            bq, bk, bv = (state[p + f"attention.self.{n}.bias"] for n in ("query", "key", "value"))
            L = layers[i]
            L.wqkv, L.bqkv = mat(torch.cat([wq, wk, wv], 0)), vec(torch.cat([bq, bk, bv], 0))
            L.wo, L.bo = mat(state[p + "attention.output.dense.weight"]), vec(state[p + "attention.output.dense.bias"])
            L.ln1_g, L.ln1_b = vec(state[p + "attention.output.LayerNorm.weight"]), vec(state[p + "attention.output.LayerNorm.bias"])
            L.w1, L.b1 = mat(state[p + "intermediate.dense.weight"]), vec(state[p + "intermediate.dense.bias"])
            L.w2, L.b2 = mat(state[p + "output.dense.weight"]), vec(state[p + "output.dense.bias"])
            L.ln2_g, L.ln2_b = vec(state[p + "output.LayerNorm.weight"]), vec(state[p + "output.LayerNorm.bias"])
        w = _Weights()
        w.word_emb = mat(state["embeddings.word_embeddings.weight"])
        w.pos_emb = mat(state["embeddings.position_embeddings.weight"])
        w.type_emb = mat(state["embeddings.token_type_embeddings.weight"][0])
        w.emb_ln_g, w.emb_ln_b = vec(state["embeddings.LayerNorm.weight"]), vec(state["embeddings.LayerNorm.bias"])
        w.layers = layers
        self.has_head = "classifier.dense.weight" in state
        if self.has_head:
            w.cls_dense_w, w.cls_dense_b = mat(state["classifier.dense.weight"]), vec(state["classifier.dense.bias"])
            w.cls_out_w = mat(state["classifier.out_proj.weight"].reshape(-1))
            w.cls_out_b = vec(state["classifier.out_proj.bias"].reshape(-1))
        c = _Cfg(cfg.vocab, cfg.hidden, cfg.layers, cfg.heads, cfg.ffn, cfg.max_pos, cfg.pad_id, cfg.ln_eps)
        h = ctypes.c_uint64()
        nat.call("hipenc_create", ctypes.byref(c), ctypes.byref(w), self.device, ctypes.byref(h))
        self._h = h.value

    def close(self) -> None:
        if getattr(self, "_h", None):
            try:
                nat.call("hipenc_destroy", self._h)
            finally:
                self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def dimension(self) -> int:
        return self.cfg.hidden

    @staticmethod
    def _pad(token_lists: Sequence[Sequence[int]]):
        lens = np.asarray([len(t) for t in token_lists], dtype=np.int32)
        max_len = max(1, int(lens.max()) if len(lens) else 1)
        ids = np.zeros((len(token_lists), max_len), dtype=np.int32)
        for i, t in enumerate(token_lists):
            ids[i, :len(t)] = np.asarray(t, dtype=np.int32)
        return ids, lens, max_len

    def _run(self, fn: str, token_lists: Sequence[Sequence[int]], out_cols: int, batch_size: int,
             max_tokens: Optional[int] = None):
        """Length-sorted batches (what sentence-transformers' encode does with batch_size=32); results are restored
        to input order and do not depend on the batching: padding never reaches a real token.
        `max_tokens`: form batches by PADDED TOKEN COUNT instead of sequence count -- as many of the next-longest
        sequences as fit nseq * S <= max_tokens (S = the batch's longest, rounded up to 64) and nseq <= batch_size; short
        texts then fill the GPU as well as long ones do (32 x 128 tokens is a twentieth of what one forward can take)."""
        import torch
        from .index import _stream_ptr
        n = len(token_lists)
        shape = (n, out_cols) if out_cols > 1 else (n,)
        out = torch.zeros(shape, dtype=torch.float32, device=torch.device("cuda", self.device))
        order = sorted(range(n), key=lambda i: -len(token_lists[i]))
        o = 0
        while o < n:
            take = batch_size
            if max_tokens is not None:
                s_pad = max(64, -(-len(token_lists[order[o]]) // 64) * 64)
                take = max(1, min(batch_size, max_tokens // s_pad))
            idx = order[o:o + take]
            o += take
            ids, lens, max_len = self._pad([token_lists[i] for i in idx])
            part = torch.empty((len(idx), out_cols) if out_cols > 1 else (len(idx),), dtype=torch.float32, device=out.device)
            nat.call(fn, self._h, ids.ctypes.data, lens.ctypes.data, len(idx), max_len, part.data_ptr(), _stream_ptr())
            out[torch.as_tensor(idx, device=out.device)] = part
        return out

    def encode_tokens(self, token_lists: Sequence[Sequence[int]], batch_size: int = 256, max_tokens: Optional[int] = None):
        """-> float32 CUDA tensor [n, hidden]: L2-normalised CLS embeddings (zero rows for empty token lists)."""
        return self._run("hipenc_forward", token_lists, self.cfg.hidden, batch_size, max_tokens)

    def score_tokens(self, token_lists: Sequence[Sequence[int]], batch_size: int = 64, max_tokens: Optional[int] = None):
        """-> float32 CUDA tensor [n]: classification-head logits of `<s> query </s></s> passage </s>` sequences."""
        if not self.has_head:
            raise ValueError("this encoder was created without a classification head")
        return self._run("hipenc_score_pairs", token_lists, 1, batch_size, max_tokens)

    def last_flops(self) -> float:
        v = ctypes.c_double()
        nat.call("hipenc_last_flops", self._h, ctypes.byref(v))
        return v.value
