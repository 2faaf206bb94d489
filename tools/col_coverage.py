import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gcn_amd import rmat
a = rmat.rmat_ahat(24, 8, 123, torch.device("cuda:0"))
deg = torch.bincount(a.col_idx.long(), minlength=a.n_rows)
s, _ = torch.sort(deg, descending=True)
c = torch.cumsum(s, 0).double() / a.nnz
for k in (1024, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 1 << 20, 1 << 22):
    print(k, "%.3f" % float(c[k - 1]))
