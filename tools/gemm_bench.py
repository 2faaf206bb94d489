"""Time the dense products of one RMAT-24 training step (fp32) one by one.

    python tools/gemm_bench.py [--rows 16777216] [--iters 5]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1 << 24)
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--classes", type=int, default=16)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--only", default="", help="substring of the case name to run")
    ap.add_argument("--arith", type=int, default=0, help="0: split bf16 operands on the matrix pipe; 1: fp32 MFMA chains")
    args = ap.parse_args()
    from gcn_amd import ops
    ops.dense_arith(args.arith)
    dev = torch.device("cuda:0")
    n, h, c = args.rows, args.hidden, args.classes
    x = torch.rand((n, h), device=dev) - 0.5
    y = torch.empty((n, h), device=dev)
    w1 = torch.rand((h, h), device=dev) - 0.5
    w2 = torch.rand((h, c), device=dev) - 0.5
    p2 = torch.rand((n, c), device=dev) - 0.5
    g1 = torch.empty((h, h), device=dev)
    g2 = torch.empty((h, c), device=dev)
    drop = (0.5, 7, 0)
    bits = ops.pack_sign_bits(x)
    cases = [
        ("X.W1 [N,h]x[h,h]", 2.0 * n * h * h, lambda: ops.gemm(x, w1, out=y)),
        ("drop(X).W1", 2.0 * n * h * h, lambda: ops.gemm(x, w1, out=y, a_dropout=drop)),
        ("X^T.dP [h,N]x[N,h]", 2.0 * n * h * h, lambda: ops.gemm(x, y, trans_a=True, out=g1)),
        ("drop(X)^T.dP", 2.0 * n * h * h, lambda: ops.gemm(x, y, trans_a=True, out=g1, a_dropout=drop)),
        ("H.W2 [N,h]x[h,c]", 2.0 * n * h * c, lambda: ops.gemm(x, w2, out=p2)),
        ("H^T.dP2 [h,N]x[N,c]", 2.0 * n * h * c, lambda: ops.gemm(x, p2, trans_a=True, out=g2)),
        ("dP2.W2^T gated [N,c]x[c,h]", 2.0 * n * h * c,
         lambda: ops.gemm(p2, w2, trans_b=True, out=y, gate=x, gate_scale=2.0)),
        ("dP2.W2^T bit-gated", 2.0 * n * h * c,
         lambda: ops.gemm(p2, w2, trans_b=True, out=y, gate_bits=bits, gate_scale=2.0)),
    ]
    for name, flops, fn in cases:
        if args.only and args.only not in name:
            continue
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(args.iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        ms = ts[len(ts) // 2]
        print("%-30s %8.2f ms  %7.1f TFLOP/s" % (name, ms, flops / ms / 1e9), flush=True)


if __name__ == "__main__":
    main()
