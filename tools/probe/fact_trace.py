"""Development probe: phase timing of spmm_factored_mfma_kernel.  Needs a
library built from gcnx_spmm.hip with tools/probe/factored_trace.patch applied
and -DFACT_TRACE (load it with GCNX_LIB=<that library>); the shipped library
carries no trace code.  Prints the mean cycle counts per round of 4096
wavefronts in the middle of the grid (RMAT-24, width 256)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gcn_amd import _lib, ops, rmat   # noqa: E402


def main():
    dev = torch.device("cuda:0")
    a = rmat.rmat_ahat(24, 8, 123, dev)
    w = 256
    d = torch.rand((a.n_cols, 16), device=dev) - 0.5
    w2 = torch.rand((w, 16), device=dev) - 0.5
    bits = ops.pack_sign_bits(torch.rand((a.n_cols, w), device=dev) - 0.75)
    out = torch.empty((a.n_rows, w), device=dev)
    for _ in range(2):
        ops.spmm_factored(a, d, w2, bits, gate_scale=2.0, trans_w=True, out=out)
    torch.cuda.synchronize()
    lib = _lib.load()
    host = np.zeros((4096, 8), dtype=np.uint64)
    fn = lib.gcnx_debug_fact_trace
    fn.argtypes = [C.c_void_p]
    rc = fn(host.ctypes.data)
    assert rc == 0, rc
    t = host.astype(np.float64)
    t = t[t[:, 7] > 0]
    r = t[:, 7]
    names = ["issue+wait", "lds read+split", "mfma issue", "mfma drain", "walk"]
    print("wavefronts %d, rounds per wavefront %.1f" % (len(t), r.mean()))
    tot = 0.0
    for i, nme in enumerate(names):
        v = (t[:, i] / r).mean()
        tot += v
        print("%-16s %8.0f ticks per round" % (nme, v))
    print("%-16s %8.0f ticks per round" % ("sum", tot))
    print("prologue %.0f ticks, wavefront life %.0f ticks, loop share %.3f" % (
        t[:, 5].mean(), t[:, 6].mean(), (t[:, :5].sum(1) / t[:, 6]).mean()))


if __name__ == "__main__":
    main()
