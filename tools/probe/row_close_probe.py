"""Development probe: how much of the class-width SpMM (16 columns) is the closing of
rows rather than the gathers.  The same nonzeros (same gathers) with every 2 / 4 /
16 adjacent rows merged into one: fewer row ends, same memory traffic for the
operand."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gcn_amd import ops, rmat   # noqa: E402
from gcn_amd.sparse import Csr   # noqa: E402


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
    dev = torch.device("cuda:0")
    a = rmat.rmat_ahat(24, 8, 123, dev)
    for width in (16, 256):
        b = torch.rand((a.n_cols, width), device=dev) - 0.5
        for merge in (1, 2, 4, 16, 64):
            rp = a.row_ptr[::merge].contiguous()
            if int(rp[-1]) != a.nnz:
                rp = torch.cat([rp, a.row_ptr[-1:]])
            m = Csr(rp.numel() - 1, a.n_cols, rp, a.col_idx, a.val, a.input_index)
            out = torch.empty((m.n_rows, width), device=dev)
            ms = timed(lambda: ops.spmm(m, b, out=out))
            print("width %3d, %2d rows merged into one (%9d rows): %.3f ms" % (width, merge, m.n_rows, ms), flush=True)
            del m, out


if __name__ == "__main__":
    main()
