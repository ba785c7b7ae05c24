#!/usr/bin/env python
"""What does the vendor library reach on the C5 product shape (fp32 in, fp32 MFMA, no TF32)?
(16384 x 512) @ (512 x 512), plus two larger shapes for the asymptote.  A yardstick for csrc/bjx_dense.hip."""
import json

import torch

torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda:0")
out = {}
for (m, k, n) in [(16384, 512, 512), (32768, 512, 512), (65536, 512, 512), (131072, 512, 512), (8192, 4096, 4096)]:
    a = torch.randn(m, k, device=dev)
    b = torch.randn(k, n, device=dev)
    c = torch.empty(m, n, device=dev)
    for _ in range(5):
        torch.mm(a, b, out=c)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50
    e0.record()
    for _ in range(reps):
        torch.mm(a, b, out=c)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    tf = 2.0 * m * k * n / us / 1e6
    out[f"{m}x{k}x{n}"] = {"us": round(us, 2), "TFLOPs": round(tf, 1), "frac_of_157.3": round(tf / 157.3, 3)}
print(json.dumps(out))
