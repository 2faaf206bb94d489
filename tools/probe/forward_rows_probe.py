"""How much of the forward pass a training step needs when few nodes are labelled
(development probe).  The masked loss reads Z at the labelled rows only, those read
H' at the nodes with a labelled neighbour (N1), those read P1 everywhere.  Times the
hidden-width forward SpMM over all rows and over the rows of N1, and the class-width
product over all rows and over the labelled rows, on an RMAT graph."""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from gcn_amd import ops, rmat


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
    ap.add_argument("--edge-factor", type=int, default=8)
    ap.add_argument("--label-rate", type=float, default=0.01)
    ap.add_argument("--width", type=int, default=256)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    a = rmat.rmat_ahat(args.scale, args.edge_factor, 123, dev)
    n = a.n_rows
    g = torch.Generator(device="cpu")
    g.manual_seed(7)
    lab = (torch.rand(n, generator=g) < args.label_rate).to(dev)
    at_dz = a.filter_columns(lab)                     # A^T = A: nonzeros of the labelled columns
    hit = (at_dz.row_ptr[1:] - at_dz.row_ptr[:-1]) > 0
    at_wide = a.filter_columns(hit)
    fwd_rows, rows = at_wide.transpose().compact_rows()
    print("n=%d nnz=%d labelled=%d N1=%d (%.1f%% of nodes) nnz of the rows of N1=%d (%.1f%%)" % (
        n, a.nnz, int(lab.sum()), fwd_rows.n_rows, 100.0 * fwd_rows.n_rows / n, fwd_rows.nnz,
        100.0 * fwd_rows.nnz / a.nnz), flush=True)
    w = args.width
    a.set_gather_hint(w * 4)
    fwd_rows.set_gather_hint(w * 4)
    p1 = torch.rand((n, w), device=dev) - 0.5
    out = torch.empty((n, w), device=dev)
    out_c = torch.empty((fwd_rows.n_rows, w), device=dev)
    bits = torch.empty((n, w // 32), dtype=torch.int32, device=dev)
    bits_c = torch.empty((fwd_rows.n_rows, w // 32), dtype=torch.int32, device=dev)
    drop = (0.5, 7, 0, w, 0)
    t_full = timed(lambda: ops.spmm(a, p1, out=out, relu=True, dropout=drop, sign_bits=bits))
    t_rows = timed(lambda: ops.spmm(fwd_rows, p1, out=out_c, relu=True, dropout=drop, sign_bits=bits_c))
    print("W=%d forward SpMM: all rows %.2f ms, rows of N1 %.2f ms" % (w, t_full, t_rows), flush=True)
    # the result rows agree with the full product up to the order of partial sums
    ref = out[rows.long()]
    print("max |difference| against the same rows of the full product: %.3e (max |value| %.3e)" % (
        float((out_c - ref).abs().max()), float(ref.abs().max())), flush=True)
    # class width: Z at the labelled rows from P2 at the rows of N1 (columns renumbered)
    lab_rows_m, lab_rows = a.filter_columns(hit).transpose().filter_columns(lab).transpose().compact_rows()
    print("labelled rows %d, their nonzeros %d" % (lab_rows_m.n_rows, lab_rows_m.nnz), flush=True)
    p2 = torch.rand((n, 16), device=dev) - 0.5
    z = torch.empty((n, 16), device=dev)
    z_c = torch.empty((lab_rows_m.n_rows, 16), device=dev)
    t_zfull = timed(lambda: ops.spmm(a, p2, out=z))
    t_zrows = timed(lambda: ops.spmm(lab_rows_m, p2, out=z_c))
    print("W=16 forward SpMM: all rows %.3f ms, labelled rows %.3f ms" % (t_zfull, t_zrows), flush=True)
    h_c = torch.rand((fwd_rows.n_rows, w), device=dev)
    h = torch.rand((n, w), device=dev)
    w2 = torch.rand((w, 16), device=dev) - 0.5
    p2c = torch.empty((fwd_rows.n_rows, 16), device=dev)
    t_gfull = timed(lambda: ops.gemm(h, w2, out=p2))
    t_grows = timed(lambda: ops.gemm(h_c, w2, out=p2c))
    print("H' W2: all rows %.3f ms, rows of N1 %.3f ms" % (t_gfull, t_grows), flush=True)


if __name__ == "__main__":
    main()
