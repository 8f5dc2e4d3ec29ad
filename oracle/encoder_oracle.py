"""
oracle/encoder_oracle.py -- fp32 CPU restatement of the XLM-RoBERTa forward the reference runs through
sentence-transformers (rag/providers/hf/embeddings.py:32-35,54,77): CLS pooling + L2 normalisation, and of
XLMRobertaForSequenceClassification's head for the reranker the reference only configures (rag/config.py:25-27).

TEST INFRASTRUCTURE ONLY (see oracle/hybrid_oracle.py header).  The third-party implementation it restates is
`transformers` (requirements.txt:29 pins >=4.37; 5.15 is installed here): tests/test_encoder_oracle.py pins this file
against transformers' own XLMRobertaModel / XLMRobertaForSequenceClassification built from a local config with the same
seeded weights.  No BGE-M3 checkpoint or tokenizer exists offline, so parity with the released model's numbers is
UNPINNED; what is pinned is the architecture's arithmetic.
"""
from __future__ import annotations

import math
from typing import Dict, Sequence

import numpy as np
import torch


def _ln(x, g, b, eps):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * g + b


def xlmr_hidden_fp32(sd: Dict[str, torch.Tensor], token_lists: Sequence[Sequence[int]], layers: int, heads: int,
                     pad_id: int = 1, eps: float = 1e-5) -> torch.Tensor:
    """Last hidden state [n, S, H] (fp32, CPU) for right-padded sequences; padded keys are masked."""
    sd = {k: v.float() for k, v in sd.items()}
    n = len(token_lists)
    S = max(1, max(len(t) for t in token_lists))
    ids = torch.full((n, S), pad_id, dtype=torch.long)
    mask = torch.zeros((n, S), dtype=torch.bool)
    for i, t in enumerate(token_lists):
        ids[i, :len(t)] = torch.as_tensor(list(t), dtype=torch.long)
        mask[i, :len(t)] = True
    pos = torch.cumsum(mask.long(), 1) * mask.long() + pad_id          # create_position_ids_from_input_ids
    x = sd["embeddings.word_embeddings.weight"][ids] + sd["embeddings.position_embeddings.weight"][pos] + \
        sd["embeddings.token_type_embeddings.weight"][0]
    x = _ln(x, sd["embeddings.LayerNorm.weight"], sd["embeddings.LayerNorm.bias"], eps)
    H = x.shape[-1]
    dh = H // heads
    neg = torch.zeros((n, 1, 1, S))
    neg.masked_fill_(~mask[:, None, None, :], float("-inf"))
    for i in range(layers):
        p = f"encoder.layer.{i}."

        def lin(name, t):
            return t @ sd[p + name + ".weight"].T + sd[p + name + ".bias"]

        q = lin("attention.self.query", x).view(n, S, heads, dh).transpose(1, 2)
        k = lin("attention.self.key", x).view(n, S, heads, dh).transpose(1, 2)
        v = lin("attention.self.value", x).view(n, S, heads, dh).transpose(1, 2)
        att = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(dh) + neg, dim=-1)
        ctx = (att @ v).transpose(1, 2).reshape(n, S, H)
        x = _ln(lin("attention.output.dense", ctx) + x, sd[p + "attention.output.LayerNorm.weight"],
                sd[p + "attention.output.LayerNorm.bias"], eps)
        h = lin("intermediate.dense", x)
        h = 0.5 * h * (1.0 + torch.erf(h / math.sqrt(2.0)))
        x = _ln(lin("output.dense", h) + x, sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"], eps)
    return x


def embed_fp32(sd, token_lists, layers, heads, pad_id=1, eps=1e-5) -> np.ndarray:
    """CLS pooling + L2 normalise (sentence-transformers BGE recipe); zero rows for empty inputs."""
    nonempty = [t if len(t) else [pad_id] for t in token_lists]
    cls = xlmr_hidden_fp32(sd, nonempty, layers, heads, pad_id, eps)[:, 0, :]
    out = cls / cls.norm(dim=1, keepdim=True).clamp_min(1e-12)
    for i, t in enumerate(token_lists):
        if len(t) == 0:
            out[i] = 0
    return out.numpy()


def rerank_logits_fp32(sd, token_lists, layers, heads, pad_id=1, eps=1e-5) -> np.ndarray:
    """XLMRobertaClassificationHead: out_proj(tanh(dense(cls)))."""
    cls = xlmr_hidden_fp32(sd, token_lists, layers, heads, pad_id, eps)[:, 0, :]
    sd = {k: v.float() for k, v in sd.items()}
    h = torch.tanh(cls @ sd["classifier.dense.weight"].T + sd["classifier.dense.bias"])
    return (h @ sd["classifier.out_proj.weight"].T + sd["classifier.out_proj.bias"]).reshape(-1).numpy()


def bf16_round_state(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Matrices / embedding tables as the GPU holds them (bf16), biases and LayerNorm parameters fp32."""
    out = {}
    for k, v in sd.items():
        is_mat = v.dim() == 2
        out[k] = v.to(torch.bfloat16).float() if is_mat else v.float()
    return out
