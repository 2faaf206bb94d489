"""Largest Laplacian eigenvalue of an RMAT graph by the device Lanczos iteration
(gcn_amd.spectral): iterations, residual and wall time.

usage: python tools/eigval_bench.py [--scale 24]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    import torch
    from gcn_amd import rmat, spectral
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=24)
    ap.add_argument("--max-iters", type=int, default=64)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    a = rmat.rmat_ahat(args.scale, 8, 123, dev)          # D^-1/2 (A + I) D^-1/2: same code path, other matrix
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lmax, iters, resid = spectral.largest_laplacian_eigval(a, max_iters=args.max_iters)
    torch.cuda.synchronize()
    print("scale %d nnz %d: lmax(I - A_hat) = %.9f after %d iterations, residual %.2e, %.2f s" %
          (args.scale, a.nnz, lmax, iters, resid, time.perf_counter() - t0))


if __name__ == "__main__":
    main()
