"""Probe: the tall split-operand product X' W1 with resident workgroups (one per
compute unit, epilogue of a tile under the next tile's MFMAs) against the
one-workgroup-per-tile form: same bits, time per launch."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from gcn_amd import ops


def main():
    dev = torch.device("cuda:0")
    for n in (1 << 17, 1 << 24):
        x = torch.rand((n, 256), device=dev) - 0.5
        w1 = torch.rand((256, 256), device=dev) - 0.5
        drop = (0.5, 7, 4096)
        for name, kw in (("plain", {}), ("dropout", {"a_dropout": drop}), ("relu", {"relu": True})):
            outs = {}
            for mode in ("0", "1"):
                os.environ["GCNX_NN_PERSIST"] = mode
                y = torch.full((n, 256), float("nan"), device=dev)
                ops.gemm(x, w1, out=y, **kw)
                torch.cuda.synchronize()
                ts = []
                for _ in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    ops.gemm(x, w1, out=y, **kw)
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1))
                ts.sort()
                outs[mode] = (y, ts[2])
            same = bool(torch.equal(outs["0"][0], outs["1"][0]))
            print("n=%d %-8s per-tile %.3f ms  resident %.3f ms  equal bits: %s" % (n, name, outs["0"][1], outs["1"][1], same),
                  flush=True)
            del outs, y


if __name__ == "__main__":
    main()
