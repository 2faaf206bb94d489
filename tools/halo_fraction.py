"""How many remote rows does one row shard of the RMAT graph reference?
(sizing of the halo exchange; development tool)"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gcn_amd import rmat

dev = torch.device("cuda:0")
EF = 8          # undirected edges per node; "--edge-factor 16" = the average degree 32 of the several-GPU default
if "--edge-factor" in sys.argv:
    i = sys.argv.index("--edge-factor")
    EF = int(sys.argv[i + 1])
    del sys.argv[i:i + 2]
if len(sys.argv) > 1 and sys.argv[1] == "--balance":
    # nonzeros of every equal-row shard (the scrambled vertex ids balance them)
    for scale, parts in ((24, 2), (24, 4), (24, 8), (25, 2), (26, 4), (27, 8)):
        n = 1 << scale
        n_loc = n // parts
        counts = []
        for p in range(parts):
            keys, nk = rmat.rmat_keys(scale, EF, 123, dev, p * n_loc, (p + 1) * n_loc)
            row_ptr, col_idx, val, nnz = rmat.csr_from_keys(keys, nk, p * n_loc, (p + 1) * n_loc, n)
            counts.append(int(nnz))
            del keys, row_ptr, col_idx, val
            torch.cuda.empty_cache()
        mean = sum(counts) / float(parts)
        print("scale %d, %d equal-row shards: nnz min %d max %d mean %.0f -> max/mean %.4f, min/mean %.4f"
              % (scale, parts, min(counts), max(counts), mean, max(counts) / mean, min(counts) / mean), flush=True)
    sys.exit(0)
for scale, parts in ((24, 8), (25, 2), (26, 4), (27, 8)):
    n = 1 << scale
    n_loc = n // parts
    t0 = time.time()
    keys, nk = rmat.rmat_keys(scale, EF, 123, dev, 0, n_loc)
    row_ptr, col_idx, val, nnz = rmat.csr_from_keys(keys, nk, 0, n_loc, n)
    del keys, val
    remote = col_idx[col_idx >= n_loc]
    uniq = torch.unique(remote)
    deg = row_ptr[1:] - row_ptr[:-1]
    print("edge factor %d, scale %d, %d shards: shard nnz %d, remote refs %d, unique remote rows %d = %.1f%% of the %d remote rows; "
          "isolated local rows %.1f%%; %.1fs" % (EF, scale, parts, nnz, remote.numel(), uniq.numel(),
          100.0 * uniq.numel() / (n - n_loc), n - n_loc, 100.0 * float((deg == 1).float().mean()), time.time() - t0), flush=True)
    del row_ptr, col_idx, remote, uniq
    torch.cuda.empty_cache()
