"""Does a Philox mask kernel (vector-ALU work, few bytes) hide under the
memory-bound forward SpMM when both are issued on two streams?

    python tools/philox_overlap_probe.py [--scale 24]
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
    p = torch.rand((n, h), device=dev) - 0.5
    out = torch.empty((n, h), device=dev)
    mask = torch.empty(n * h, dtype=torch.uint8, device=dev)
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

    spmm = lambda: ops.spmm(a, p, out=out, relu=True, dropout=(0.5, 7, 0, h, 0))
    draw = lambda: ops.bernoulli_mask(n * h, 0.5, 9, 1 << 44, dev, out=mask)

    def both(first):
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur)
        s2.wait_stream(cur)
        for which in first:
            with torch.cuda.stream(s1 if which == "spmm" else s2):
                (spmm if which == "spmm" else draw)()
        cur.wait_stream(s1)
        cur.wait_stream(s2)

    print("spmm alone %.2f ms, mask draw alone %.2f ms" % (timed(spmm), timed(draw)))
    print("two streams, spmm issued first %.2f ms" % timed(lambda: both(("spmm", "draw"))))
    print("two streams, draw issued first %.2f ms" % timed(lambda: both(("draw", "spmm"))))


if __name__ == "__main__":
    main()
