"""Do a memory-bound SpMM and an MFMA-bound dense product overlap when they are
issued on two streams?  Times them back to back on one stream and concurrently.

    python tools/overlap_probe.py [--scale 24]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=24)
    args = ap.parse_args()
    from gcn_amd import ops, rmat
    dev = torch.device("cuda:0")
    a = rmat.rmat_ahat(args.scale, 8, 123, dev)
    n, h = a.n_rows, 256
    x = torch.rand((n, h), device=dev) - 0.5
    p = torch.rand((n, h), device=dev) - 0.5
    out = torch.empty((n, h), device=dev)
    g = torch.empty((h, h), device=dev)
    y = torch.empty((n, h), device=dev)
    w = torch.rand((h, h), device=dev) - 0.5
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return sorted(ts)[1]

    def seq(dense):
        ops.spmm(a, p, out=out)
        dense()

    def par(dense, dense_first):
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur)
        s2.wait_stream(cur)
        if dense_first:
            with torch.cuda.stream(s2):
                dense()
            with torch.cuda.stream(s1):
                ops.spmm(a, p, out=out)
        else:
            with torch.cuda.stream(s1):
                ops.spmm(a, p, out=out)
            with torch.cuda.stream(s2):
                dense()
        cur.wait_stream(s1)
        cur.wait_stream(s2)

    tn = lambda: ops.gemm(x, y, trans_a=True, out=g, a_dropout=(0.5, 7, 0))
    nn = lambda: ops.gemm(x, w, out=y, a_dropout=(0.5, 7, 0))
    print("spmm alone        %.2f ms" % timed(lambda: ops.spmm(a, p, out=out)))
    for name, d in (("tn gemm", tn), ("nn gemm", nn)):
        print("%s alone     %.2f ms" % (name, timed(d)))
        print("spmm ; %s    %.2f ms" % (name, timed(lambda: seq(d))))
        print("spmm || %s (spmm issued first)  %.2f ms" % (name, timed(lambda: par(d, False))))
        print("spmm || %s (gemm issued first)  %.2f ms" % (name, timed(lambda: par(d, True))))


if __name__ == "__main__":
    main()
