"""Probe helper: one kind of W = 256 product per process, for rocprofv3 --pmc passes
(L2 hit rate of whole rows against column slabs pinned to XCDs).  Needs the library
built from tools/probe/xcd_slab_spmm.patch (GCNX_LIB)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from gcn_amd import ops, rmat

mode = sys.argv[1]          # "rows" | "slabs" | "slabs_major"
os.environ["GCNX_XCD_SLABS"] = {"rows": "0", "slabs": "1", "slabs_major": "2"}[mode]
dev = torch.device("cuda:0")
a = rmat.rmat_ahat(24, 8, 123, dev)
b = torch.rand((a.n_cols, 256), device=dev) * 2 - 1
out = torch.empty((a.n_rows, 256), device=dev)
for _ in range(2):
    ops.spmm(a, b, out=out)
torch.cuda.synchronize()
print("done", mode)
