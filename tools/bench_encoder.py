#!/usr/bin/env python3
"""Encoder throughput on the BASELINE config 5 shape: embed_batch(bs=256, 512-token sequences), XLM-R large, random
weights + synthetic token ids.  Prints tokens/s, algorithmic TFLOP/s and the fraction of the bf16 dense
MFMA peak (2.5 PFLOP/s, MI355X_MICROARCH.md)."""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "intool-rag_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bs", type=int, default=256)
    ap.add_argument("--seq", type=int, default=512)
    ap.add_argument("--layers", type=int, default=24)
    ap.add_argument("--iters", type=int, default=3)
    args = ap.parse_args()
    import torch
    from hiprag import EncoderConfig, HipEncoder
    cfg = EncoderConfig(layers=args.layers)
    enc = HipEncoder(cfg, seed=0)
    rng = np.random.default_rng(0)
    toks = [[0] + rng.integers(3, cfg.vocab, size=args.seq - 2).tolist() + [2] for _ in range(args.bs)]
    enc.encode_tokens(toks, batch_size=args.bs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        enc.encode_tokens(toks, batch_size=args.bs)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.iters
    flops = enc.last_flops()
    print(json.dumps({"bs": args.bs, "seq": args.seq, "layers": args.layers, "ms_per_batch": round(dt * 1e3, 2),
                      "tokens_per_s": round(args.bs * args.seq / dt, 1), "tflops": round(flops / dt / 1e12, 1),
                      "frac_of_2.5PF": round(flops / dt / 2.5e15, 4), "flops_per_batch": flops}))


if __name__ == "__main__":
    main()
