#!/usr/bin/env python3
"""
The encoder's four linear layers in isolation (hipenc_linear): correctness of both tiled kernels against an fp32 torch
reference on the same bf16 operands, their agreement with each other, and TFLOP/s per shape at the BASELINE config-5 batch
(256 x 512 tokens = 131072 rows).  Prints one JSON line per shape; `--yardstick` also times torch.matmul (the library GEMM
without any epilogue) on the same shapes.
"""
import argparse
import ctypes
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "intool-rag_amd"))


def linear(nat, sp, a, w, bias, epi, impl, resid=None, S=512, heads=16):
    import torch
    M, K = a.shape
    N = w.shape[0]
    dev = a.device
    if epi == 0:
        H = N // 3
        q = torch.empty((M // S, heads, S, 64), dtype=torch.bfloat16, device=dev)
        k = torch.empty_like(q)
        vt = torch.empty((M // S, heads, 64, S), dtype=torch.bfloat16, device=dev)
        nat.call("hipenc_linear", a.data_ptr(), w.data_ptr(), bias.data_ptr(), M, N, K, 0, None, q.data_ptr(), k.data_ptr(),
                 vt.data_ptr(), S, heads, impl, sp())
        return q, k, vt
    if epi == 1:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        nat.call("hipenc_linear", a.data_ptr(), w.data_ptr(), bias.data_ptr(), M, N, K, 1, None, out.data_ptr(), None, None, 0, 0,
                 impl, sp())
        return (out,)
    out = torch.empty((M, N), dtype=torch.float32 if epi == 2 else torch.bfloat16, device=dev)
    nat.call("hipenc_linear", a.data_ptr(), w.data_ptr(), bias.data_ptr(), M, N, K, epi, resid.data_ptr(), out.data_ptr(), None, None,
             0, 0, impl, sp())
    return (out,)


def reference(a, w, bias, epi, resid=None, S=512, heads=16, rows=2048):
    """fp32 torch reference on the first `rows` rows (exact-erf GELU, the encoder's layouts)."""
    import torch
    c = a[:rows].float() @ w.float().T + bias
    if epi == 1:
        return (torch.nn.functional.gelu(c),)
    if epi >= 2:
        return (c + resid[:rows].float(),)
    H = w.shape[0] // 3
    nseq = rows // S
    q = (c[:, :H] * 0.125).view(nseq, S, heads, 64).permute(0, 2, 1, 3)
    k = c[:, H:2 * H].view(nseq, S, heads, 64).permute(0, 2, 1, 3)
    vt = c[:, 2 * H:].view(nseq, S, heads, 64).permute(0, 2, 3, 1)
    return q, k, vt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=256 * 512)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--yardstick", action="store_true")
    ap.add_argument("--impls", default="1,2", help="1 = 128 x 128 tiles, 2 = persistent 256 x 256 tiles")
    args = ap.parse_args()
    import torch
    from hiprag import _native as nat
    from hiprag.index import _stream_ptr as sp
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    M, H, F, S, heads = args.rows, 1024, 4096, 512, 16
    shapes = [("qkv", 0, 3 * H, H), ("out_proj", 2, H, H), ("out_proj_bf16", 3, H, H), ("ffn_up", 1, F, H), ("ffn_down", 2, H, F),
              ("ffn_down_bf16", 3, H, F)]
    impls = [int(v) for v in args.impls.split(",")]
    for name, epi, N, K in shapes:
        a = (torch.randn((M, K), generator=g, device=dev) * 0.5).to(torch.bfloat16)
        w = (torch.randn((N, K), generator=g, device=dev) * 0.03).to(torch.bfloat16)
        bias = torch.randn((N,), generator=g, device=dev) * 0.1
        resid = (torch.randn((M, N), generator=g, device=dev)).to(torch.bfloat16) if epi >= 2 else None
        rows = min(M, 2048)
        ref = reference(a, w, bias, epi, resid, S, heads, rows)
        rec = {"shape": name, "M": M, "N": N, "K": K, "tflop": 2.0 * M * N * K / 1e12}
        outs = {}
        for impl in impls:
            if epi == 3 and impl == 1:
                continue
            out = linear(nat, sp, a, w, bias, epi, impl, resid, S, heads)
            torch.cuda.synchronize()
            outs[impl] = out
            err = 0.0
            for o, r in zip(out, ref):
                o = o.float()
                if epi == 0:
                    o = o[:rows // S]
                else:
                    o = o[:rows]
                err = max(err, float((o - r).abs().max() / r.abs().max()))
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(3):
                linear(nat, sp, a, w, bias, epi, impl, resid, S, heads)
            ev0.record()
            for _ in range(args.reps):
                linear(nat, sp, a, w, bias, epi, impl, resid, S, heads)
            ev1.record()
            torch.cuda.synchronize()
            ms = ev0.elapsed_time(ev1) / args.reps
            rec[f"impl{impl}_ms"] = round(ms, 4)
            rec[f"impl{impl}_pflops"] = round(rec["tflop"] / ms, 4)
            rec[f"impl{impl}_max_rel_err"] = float(f"{err:.3e}")
        if len(outs) == 2:
            rec["impls_bit_equal"] = all(torch.equal(x, y) for x, y in zip(outs[impls[0]], outs[impls[1]]))
        if args.yardstick:
            wt = w.T.contiguous()
            for _ in range(3):
                torch.matmul(a, wt)
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for _ in range(args.reps):
                torch.matmul(a, wt)
            ev1.record()
            torch.cuda.synchronize()
            ms = ev0.elapsed_time(ev1) / args.reps
            rec["torch_matmul_ms"] = round(ms, 4)
            rec["torch_matmul_pflops"] = round(rec["tflop"] / ms, 4)
        print(json.dumps(rec), flush=True)
        del a, w, bias, resid, outs


if __name__ == "__main__":
    main()
