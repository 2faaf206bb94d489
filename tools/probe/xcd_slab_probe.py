"""Probe: the hidden-width forward SpMM with the operand cut into 8 column slabs,
one per XCD (workgroup b works on slab b % 8), against the whole-row kernel; with
no cache hint and with the rows of the most referenced columns kept (plain loads)
and every other row streamed, for several sizes of the kept set."""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from gcn_amd import ops, rmat


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(iters))
    return ms[len(ms) // 2], ms[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=24)
    ap.add_argument("--edge-factor", type=int, default=8)
    ap.add_argument("--iters", type=int, default=4)
    ap.add_argument("--plain", action="store_true", help="the bare product (no relu / dropout / sign words)")
    ap.add_argument("--rows", type=int, nargs="*", default=[8192, 16384, 24576, 32768, 65536, 180224])
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    a = rmat.rmat_ahat(args.scale, args.edge_factor, 123, dev)
    w = 256
    b = torch.rand((a.n_cols, w), device=dev) * 2 - 1
    out = torch.empty((a.n_rows, w), device=dev)
    bits = torch.empty((a.n_rows, w // 32), dtype=torch.int32, device=dev)
    drop = (0.5, 7, 0, w, 0)
    run = lambda: ops.spmm(a, b, out=out, relu=True, dropout=drop, sign_bits=bits)
    if args.plain:
        run = lambda: ops.spmm(a, b, out=out)
    os.environ["GCNX_XCD_SLABS"] = "0"
    print("whole rows, no hint: %.3f ms (min %.3f)" % timed(run, args.iters), flush=True)
    ref, ref_bits = out.clone(), bits.clone()
    a.set_gather_hint(w * 4)
    print("whole rows, hint (176 MiB of rows kept): %.3f ms (min %.3f)" % timed(run, args.iters), flush=True)
    assert torch.equal(out, ref) and torch.equal(bits, ref_bits)
    a.clear_gather_hint()
    os.environ["GCNX_XCD_SLABS"] = "1"
    print("8 slabs, no hint: %.3f ms (min %.3f)" % timed(run, args.iters), flush=True)
    err = (out - ref).abs().max().item()
    print("   max |difference| to the whole-row result %.3g (other order of a row's sum), sign words equal: %s"
          % (err, bool(torch.equal(bits, ref_bits))), flush=True)
    a.clear_gather_hint()
    # slab-major operand and output: slab s of B is the [rows, 32] matrix of its columns
    os.environ["GCNX_XCD_SLABS"] = "2"
    b.copy_(b.view(a.n_cols, 8, 32).permute(1, 0, 2).contiguous().view(a.n_cols, w))
    print("8 slabs, slab-major B and C, no hint: %.3f ms (min %.3f)" % timed(run, args.iters), flush=True)
    got = out.view(8, a.n_rows, 32).permute(1, 0, 2).reshape(a.n_rows, w)
    # (rows cut by a chunk boundary are finished by the fix-up pass, which this probe leaves row-major)
    close = ((got - ref).abs().max(dim=1).values <= 1e-5)
    print("   rows equal to the whole-row result within 1e-5: %.4f %%" % (100.0 * close.float().mean().item()), flush=True)
    for rows in args.rows:
        thr = a.set_gather_hint(128, budget_bytes=rows * 128)
        print("8 slabs, slab-major, %6d rows kept per XCD (degree >= %d): %.3f ms (min %.3f)"
              % ((rows, thr) + timed(run, args.iters)), flush=True)


if __name__ == "__main__":
    main()
