"""How do the memory-bound SpMM and the MFMA-bound products behave on a share of
the compute units?  Streams are created with a CU mask
(hipExtStreamCreateWithCUMask); bit i of the mask is taken to select CU i/8 of
XCD i%8, so `c` CUs per XCD = bits with i/8 < c.

    python tools/cu_mask_probe.py [--scale 24]
"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def masked_stream(hip, bits):
    words = (ctypes.c_uint32 * 8)()
    for i in bits:
        words[i // 32] |= 1 << (i % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(8), words)
    if rc != 0:
        raise RuntimeError("hipExtStreamCreateWithCUMask -> %d" % rc)
    return torch.cuda.ExternalStream(s.value)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=24)
    ap.add_argument("--pipeline", action="store_true",
                    help="row blocks: SpMM of block b, then the weight-gradient product over the same rows")
    args = ap.parse_args()
    from gcn_amd import ops, rmat
    from gcn_amd.sparse import Csr
    dev = torch.device("cuda:0")
    torch.cuda.init()
    hip = ctypes.CDLL("libamdhip64.so")
    a = rmat.rmat_ahat(args.scale, 8, 123, dev)
    n, h = a.n_rows, 256
    x = torch.rand((n, h), device=dev) - 0.5
    p = torch.rand((n, h), device=dev) - 0.5
    out = torch.empty((n, h), device=dev)
    g = torch.empty((h, h), device=dev)
    y = torch.rand((n, h), device=dev) - 0.5

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return sorted(ts)[1]

    def on(stream, fn):
        cur = torch.cuda.current_stream()
        stream.wait_stream(cur)
        with torch.cuda.stream(stream):
            fn()
        cur.wait_stream(stream)

    spmm = lambda: ops.spmm(a, p, out=out)
    tn = lambda: ops.gemm(x, y, trans_a=True, out=g, a_dropout=(0.5, 7, 0))
    print("all CUs: spmm %.2f ms, tn gemm %.2f ms" % (timed(spmm), timed(tn)))
    if args.pipeline:
        for nb in (4, 8, 16):
            rb = n // nb
            blocks = []
            for b in range(nb):
                r0, r1 = b * rb, (b + 1) * rb
                rp = a.row_ptr[r0:r1 + 1]
                k0, k1 = int(rp[0]), int(rp[-1])
                blocks.append((r0, r1, Csr(r1 - r0, a.n_cols, (rp - k0).contiguous(), a.col_idx[k0:k1], a.val[k0:k1])))
            for c in (32, 22, 20, 18):
                if c == 32:
                    s_sp = s_mm = torch.cuda.current_stream()
                else:
                    s_sp = masked_stream(hip, [i for i in range(256) if i // 8 < c])
                    s_mm = masked_stream(hip, [i for i in range(256) if i // 8 >= c])

                def piped():
                    cur = torch.cuda.current_stream()
                    if c != 32:
                        s_sp.wait_stream(cur)
                        s_mm.wait_stream(cur)
                    for b, (r0, r1, blk) in enumerate(blocks):
                        with torch.cuda.stream(s_sp):
                            ops.spmm(blk, p, out=out[r0:r1])
                        if c != 32:
                            s_mm.wait_stream(s_sp)
                        with torch.cuda.stream(s_mm):
                            ops.gemm(x[r0:r1], out[r0:r1], trans_a=True, out=g, beta=0.0 if b == 0 else 1.0,
                                     a_dropout=(0.5, 7, r0 * h))
                    if c != 32:
                        cur.wait_stream(s_sp)
                        cur.wait_stream(s_mm)
                print("%2d row blocks, %2d/32 CUs per XCD for the spmm: %.2f ms" % (nb, c, timed(piped)))
        return
    for c in (28, 24, 20, 16, 12):
        s_sp = masked_stream(hip, [i for i in range(256) if i // 8 < c])
        s_mm = masked_stream(hip, [i for i in range(256) if i // 8 >= c])
        t_sp = timed(lambda: on(s_sp, spmm))
        t_mm = timed(lambda: on(s_mm, tn))

        def both():
            cur = torch.cuda.current_stream()
            s_sp.wait_stream(cur)
            s_mm.wait_stream(cur)
            with torch.cuda.stream(s_sp):
                spmm()
            with torch.cuda.stream(s_mm):
                tn()
            cur.wait_stream(s_sp)
            cur.wait_stream(s_mm)
        print("%2d/32 CUs per XCD for the spmm: spmm alone %.2f ms; tn gemm alone on the other %2d: %.2f ms; together %.2f ms" %
              (c, t_sp, 32 - c, t_mm, timed(both)))


if __name__ == "__main__":
    main()
