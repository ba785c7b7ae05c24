#!/usr/bin/env python
"""Per-transition wall time in a fresh process (how long does the box take to reach steady state?)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blackjax_amd as bjx
dev = torch.device("cuda:0")
N, D, L = 65536, 1024, 50
sig = torch.as_tensor((10.0 ** (-1.0 + 2.0 * np.arange(D) / (D - 1))).astype(np.float32), device=dev)
imm = (sig * sig).contiguous()
alg = bjx.hmc(bjx.targets.DiagGaussian((1.0 / imm).contiguous()), 0.25, imm, L)
state = alg.init(sig * torch.randn(N, D, device=dev))
keys = bjx.random.split(bjx.random.key(0), 40)
ts = []
for t in range(40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    state, info = alg.step(keys[t], state)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(" ".join(f"{x:.1f}" for x in ts))
