"""Development probe: how much of A_hat the backward pass of a 2-layer GCN touches
when only a fraction of the nodes is labelled (RMAT graph of bench.py).  dZ is
nonzero on the labelled rows, dP2 = A^T dZ on their neighbours, and the
hidden-width backward product gathers dP2 rows: the nonzeros whose column has a
zero dP2 row contribute nothing."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gcn_amd import rmat   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=24)
    ap.add_argument("--every", type=int, nargs="+", default=[100, 20, 10])
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    a = rmat.rmat_ahat(args.scale, 8, 123, dev)
    n = a.n_rows
    for every in args.every:
        lab = (torch.arange(n, device=dev) % every == 0)
        sub = a.filter_columns(lab)
        hit = (sub.row_ptr[1:] - sub.row_ptr[:-1]) > 0          # rows of dP2 that can be nonzero
        sub2 = a.filter_columns(hit)
        hit2 = (sub2.row_ptr[1:] - sub2.row_ptr[:-1]) > 0       # rows of dP1 that can be nonzero
        print("labelled 1/%d: A^T dZ multiplies %.2f %% of the nonzeros; dP2 rows nonzero %.1f %%; "
              "A^T d(A P1) multiplies %.1f %% of the nonzeros; dP1 rows nonzero %.1f %%"
              % (every, 100.0 * sub.nnz / a.nnz, 100.0 * float(hit.float().mean()), 100.0 * sub2.nnz / a.nnz,
                 100.0 * float(hit2.float().mean())), flush=True)


if __name__ == "__main__":
    main()
