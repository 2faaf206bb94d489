"""Micro-benchmark of the SpMM kernel on an RMAT graph (development tool)."""
import argparse
import sys
import os
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gcn_amd import ops, rmat


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=20)
    ap.add_argument("--edge-factor", type=int, default=8)
    ap.add_argument("--width", type=int, nargs="+", default=[256])
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--signs", action="store_true", help="also time the relu + dropout epilogue without and "
                    "with the sign-bit output")
    ap.add_argument("--factored", action="store_true", help="time A (gate .* (D W^T)) as bit-gated product + SpMM "
                    "and as the one-kernel factored SpMM (inner 16)")
    ap.add_argument("--hint", action="store_true", help="attach the cache hint of the wide kernel "
                    "(Csr.set_gather_hint at the first width's row size)")
    ap.add_argument("--hint-budget-mib", type=float, default=176.0, help="bytes of operand rows the hint keeps cacheable")
    ap.add_argument("--field-of-labels", type=int, default=0, metavar="K",
                    help="multiply A_hat[N1, N2] instead of A_hat: every K-th node labelled, N1 = the nodes a labelled "
                         "row reads, N2 = the nodes a row of N1 reads, both stored next to each other (the matrix of "
                         "the hidden-width forward product of the engine's step over the labelled rows' field)")
    ap.add_argument("--relabel", choices=["none", "degree"], default="none",
                    help="degree: renumber the vertices by descending degree first (hub rows of the operand "
                    "become neighbours in memory)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    t0 = time.time()
    a = rmat.rmat_ahat(args.scale, args.edge_factor, 123, dev)
    if args.relabel == "degree":
        from gcn_amd.sparse import Csr
        deg = a.row_ptr[1:] - a.row_ptr[:-1]
        order = torch.argsort(deg, descending=True, stable=True)
        new_id = torch.empty_like(order)
        new_id[order] = torch.arange(a.n_rows, device=dev)
        rows = torch.repeat_interleave(torch.arange(a.n_rows, device=dev), deg)
        a = Csr.from_coo(new_id[rows].to(torch.int32), new_id[a.col_idx.long()].to(torch.int32), a.val,
                         (a.n_rows, a.n_cols), device=dev)
        del deg, order, new_id, rows
    if args.field_of_labels:
        lab = (torch.arange(a.n_rows, device=dev) % args.field_of_labels == 0)
        at = a                                        # the normalised adjacency is bitwise symmetric
        at_dz = ops.csr_filter_columns(at, lab)
        hit = (at_dz.row_ptr[1:] - at_dz.row_ptr[:-1]) > 0
        rows1 = torch.nonzero(hit).view(-1).to(torch.int32)
        sub, rows2 = ops.csr_compact_rows(ops.csr_filter_columns(at, hit))
        a = ops.csr_keep_rows(ops.csr_transpose(sub), rows1)
        print("field: %d labelled, N1 %d rows, N2 %d rows" % (int(lab.sum()), rows1.numel(), rows2.numel()), flush=True)
        del at, at_dz, hit, sub
    torch.cuda.synchronize()
    print("graph: n=%d nnz=%d max_deg=%d build %.2fs" % (
        a.n_rows, a.nnz, int((a.row_ptr[1:] - a.row_ptr[:-1]).max()), time.time() - t0), flush=True)
    if args.hint:
        print("gather hint: rows of columns with degree >= %d stay cacheable" % a.set_gather_hint(args.width[0] * 4, budget_bytes=int(args.hint_budget_mib * 2 ** 20)),
              flush=True)
    for w in args.width:
        b = torch.rand((a.n_cols, w), device=dev) * 2 - 1
        out = torch.empty((a.n_rows, w), device=dev)
        ops.spmm(a, b, out=out)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.iters + 1)]
        ev[0].record()
        for i in range(args.iters):
            ops.spmm(a, b, out=out)
            ev[i + 1].record()
        torch.cuda.synchronize()
        ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(args.iters))
        med = ms[len(ms) // 2]
        algo = a.nnz * (8 + 4 * w) + a.n_rows * 4 * w + 8 * (a.n_rows + 1)
        print("W=%d: %.3f ms (min %.3f)  nnz*F/s=%.3e  algorithmic %.1f GB/s (%.1f%% of 8 TB/s)" % (
            w, med, ms[0], a.nnz * w / (med * 1e-3), algo / (med * 1e-3) / 1e9,
            algo / (med * 1e-3) / 8e12 * 100), flush=True)
        if args.signs and ops.spmm_supports_sign_bits(w):
            bits = torch.empty((a.n_rows, w // 32), dtype=torch.int32, device=dev)
            drop = (0.5, 7, 0, w, 0)
            for name, kw in (("relu+dropout", {}), ("relu+dropout+signs", {"sign_bits": bits})):
                ops.spmm(a, b, out=out, relu=True, dropout=drop, **kw)
                torch.cuda.synchronize()
                ev[0].record()
                for i in range(args.iters):
                    ops.spmm(a, b, out=out, relu=True, dropout=drop, **kw)
                    ev[i + 1].record()
                torch.cuda.synchronize()
                ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(args.iters))
                print("W=%d %s: %.3f ms (min %.3f)" % (w, name, ms[len(ms) // 2], ms[0]), flush=True)
        if args.factored and ops.spmm_supports_factored(w, 16):
            d = torch.rand((a.n_cols, 16), device=dev) - 0.5
            w2 = torch.rand((w, 16), device=dev) - 0.5
            bits = ops.pack_sign_bits(torch.rand((a.n_cols, w), device=dev) - 0.75)    # a quarter of the bits set
            tmp = torch.empty((a.n_cols, w), device=dev)

            def two():
                ops.gemm(d, w2, trans_b=True, out=tmp, gate_bits=bits, gate_scale=2.0)
                ops.spmm(a, tmp, out=out)

            def one():
                ops.spmm_factored(a, d, w2, bits, gate_scale=2.0, trans_w=True, out=out)
            for name, fn in (("bit-gated product + SpMM", two), ("factored SpMM", one)):
                fn()
                torch.cuda.synchronize()
                ev[0].record()
                for i in range(args.iters):
                    fn()
                    ev[i + 1].record()
                torch.cuda.synchronize()
                ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(args.iters))
                print("W=%d %s: %.3f ms (min %.3f)" % (w, name, ms[len(ms) // 2], ms[0]), flush=True)
            del d, w2, bits, tmp
        del b, out


if __name__ == "__main__":
    main()
