"""CPU: pins oracle/encoder_oracle.py against transformers' own XLM-RoBERTa implementation (same seeded weights)."""
import numpy as np
import pytest
import torch

from oracle import encoder_oracle as eo


def _cfg():
    from hiprag import EncoderConfig
    return EncoderConfig(vocab=600, hidden=128, layers=2, heads=2, ffn=256, max_pos=140, max_seq_len=128)


def _tokens(rng, n, lo, hi, vocab):
    return [[0] + rng.integers(3, vocab, size=int(rng.integers(lo, hi))).tolist() + [2] for _ in range(n)]


def test_oracle_matches_transformers_xlmr():
    transformers = pytest.importorskip("transformers")
    from hiprag import random_state
    cfg = _cfg()
    sd = random_state(cfg, seed=3, with_head=True)
    hf_cfg = transformers.XLMRobertaConfig(vocab_size=cfg.vocab, hidden_size=cfg.hidden, num_hidden_layers=cfg.layers,
                                           num_attention_heads=cfg.heads, intermediate_size=cfg.ffn,
                                           max_position_embeddings=cfg.max_pos, type_vocab_size=1, layer_norm_eps=cfg.ln_eps,
                                           pad_token_id=cfg.pad_id, hidden_dropout_prob=0.0,
                                           attention_probs_dropout_prob=0.0, num_labels=1)
    model = transformers.XLMRobertaForSequenceClassification(hf_cfg).eval()
    missing, unexpected = model.load_state_dict({("roberta." + k if not k.startswith("classifier") else k): v
                                                 for k, v in sd.items()}, strict=False)
    assert not unexpected and all("position_ids" in m or "pooler" in m for m in missing), (missing, unexpected)
    rng = np.random.default_rng(0)
    toks = _tokens(rng, 5, 3, 40, cfg.vocab)
    S = max(len(t) for t in toks)
    ids = torch.full((len(toks), S), cfg.pad_id, dtype=torch.long)
    att = torch.zeros((len(toks), S), dtype=torch.long)
    for i, t in enumerate(toks):
        ids[i, :len(t)] = torch.tensor(t)
        att[i, :len(t)] = 1
    with torch.no_grad():
        hid = model.roberta(input_ids=ids, attention_mask=att).last_hidden_state
        logits = model(input_ids=ids, attention_mask=att).logits.reshape(-1)
    mine = eo.xlmr_hidden_fp32(sd, toks, cfg.layers, cfg.heads, cfg.pad_id, cfg.ln_eps)
    for i, t in enumerate(toks):                     # compare real tokens only (padding rows are unspecified)
        assert torch.allclose(mine[i, :len(t)], hid[i, :len(t)], atol=2e-5, rtol=1e-4)
    assert np.allclose(eo.rerank_logits_fp32(sd, toks, cfg.layers, cfg.heads), logits.numpy(), atol=2e-5)
    emb = eo.embed_fp32(sd, toks, cfg.layers, cfg.heads)
    ref = torch.nn.functional.normalize(hid[:, 0], dim=1).numpy()
    assert np.allclose(emb, ref, atol=1e-5)


def test_oracle_batch_invariance_and_empty_rows():
    from hiprag import random_state
    cfg = _cfg()
    sd = random_state(cfg, seed=4)
    rng = np.random.default_rng(1)
    toks = _tokens(rng, 4, 2, 30, cfg.vocab) + [[]]
    all_at_once = eo.embed_fp32(sd, toks, cfg.layers, cfg.heads)
    one_by_one = np.concatenate([eo.embed_fp32(sd, [t], cfg.layers, cfg.heads) for t in toks])
    assert np.allclose(all_at_once, one_by_one, atol=1e-5)
    assert np.all(all_at_once[-1] == 0) and np.allclose(np.linalg.norm(all_at_once[:-1], axis=1), 1.0, atol=1e-5)
