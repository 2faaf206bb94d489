"""Time the masked softmax cross-entropy kernel on [N, classes] logits."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1 << 24)
    ap.add_argument("--classes", type=int, default=16)
    args = ap.parse_args()
    from gcn_amd import ops
    dev = torch.device("cuda:0")
    n, c = args.rows, args.classes
    z = torch.randn((n, c), device=dev)
    y = torch.zeros((n, c), device=dev)
    y[torch.arange(n, device=dev), torch.randint(0, c, (n,), device=dev)] = 1.0
    m = (torch.arange(n, device=dev) % 100 == 0).to(torch.uint8)
    g = torch.empty_like(z)
    fn = lambda: ops.masked_softmax_xent(z, y, m, want_grad=True, mask_total=float(n // 100), grad_out=g)
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = sorted(ts)[2]
    print("xent %d x %d: %.3f ms, %.0f GB/s" % (n, c, ms, (3 * n * c * 4 + n) / ms / 1e6))


if __name__ == "__main__":
    main()
