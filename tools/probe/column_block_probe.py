"""Development probe: the class-width product Z = A_hat P2 (16 columns) as a sum over
column blocks of A_hat, each block's slice of P2 small enough for the Infinity
Cache: Z = sum_b A_hat[:, block b] P2, sub-matrices by Csr.filter_columns."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gcn_amd import ops, rmat   # noqa: E402


def timed(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(iters))
    return ms[len(ms) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=24)
    ap.add_argument("--width", type=int, default=16)
    ap.add_argument("--blocks", type=int, nargs="+", default=[2, 4, 8, 16])
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    a = rmat.rmat_ahat(args.scale, 8, 123, dev)
    n = a.n_cols
    b = torch.rand((n, args.width), device=dev) - 0.5
    z = torch.empty((a.n_rows, args.width), device=dev)
    base = timed(lambda: ops.spmm(a, b, out=z))
    ref = z.clone()
    print("whole matrix: %.3f ms" % base, flush=True)
    col = torch.arange(n, device=dev)
    for nb in args.blocks:
        step = (n + nb - 1) // nb
        subs = [a.filter_columns((col >= i * step) & (col < (i + 1) * step)) for i in range(nb)]

        def run():
            for i, s in enumerate(subs):
                ops.spmm(s, b, out=z, beta=0.0 if i == 0 else 1.0)
        ms = timed(run)
        err = float((z - ref).abs().max()) / float(ref.abs().max())
        print("%2d column blocks of %d rows (%.0f MiB of operand each): %.3f ms, apart %.2g"
              % (nb, step, step * args.width * 4 / 2**20, ms, err), flush=True)
        del subs


if __name__ == "__main__":
    main()
