"""Sample rocm-smi's clocks while a kernel loop runs: is a kernel running at a reduced (power-capped) clock?
(The power reading itself is not available to an unprivileged user on this pool.)
usage: python tools/probe/power_probe.py [nn|spmm|factored]"""
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gcn_amd import ops, rmat  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "nn"
dev = torch.device("cuda:0")
n, h = 1 << 24, 256
if what == "nn":
    x = torch.randn(n, h, device=dev)
    w = torch.randn(h, h, device=dev) * 0.1
    out = torch.empty(n, h, device=dev)
    fn = lambda: ops.gemm(x, w, out=out, a_dropout=(0.5, 3, 0))
else:
    a = rmat.rmat_ahat(24, 8, 123, dev)
    out = torch.empty(n, h, device=dev)
    if what == "spmm":
        b = torch.randn(n, h, device=dev)
        fn = lambda: ops.spmm(a, b, out=out)
    else:
        d = torch.rand((n, 16), device=dev) - 0.5
        w2 = torch.rand((h, 16), device=dev) - 0.5
        bits = ops.pack_sign_bits(torch.rand((n, h), device=dev) - 0.75)
        fn = lambda: ops.spmm_factored(a, d, w2, bits, gate_scale=2.0, trans_w=True, out=out)
fn()
torch.cuda.synchronize()
stop = False
samples = []


def sampler():
    while not stop:
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "-d", "0"], capture_output=True, text=True, timeout=10)
            keep = [l.strip() for l in r.stdout.splitlines() if "sclk" in l or "mclk" in l]
            samples.append((time.time(), keep))
        except Exception as exc:  # noqa: BLE001
            samples.append((time.time(), [repr(exc)]))
            return
        time.sleep(0.2)


t = threading.Thread(target=sampler)
t.start()
t0 = time.time()
it = 0
while time.time() - t0 < 6.0:
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    it += 20
el = time.time() - t0
stop = True
t.join()
print("%s: %d launches, %.3f ms each" % (what, it, el / it * 1e3))
for ts, keep in samples[:: max(1, len(samples) // 8)]:
    print("%.1fs " % (ts - t0) + " | ".join(keep))
