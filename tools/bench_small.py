#!/usr/bin/env python
"""BASELINE.json configs[0] shape (128 chains x 1 024 dims, L = 10) and a few other small batches:
transitions per second with the default driver (use_graph="auto": HIP graph of the inner loop for
small blocks) against plain launches -- the regime where a launch is a few microseconds of GPU work
and the host's launch rate is the limit."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blackjax_amd as bjx  # noqa: E402

dev = torch.device("cuda:0")
out = {}
for N, D, L in ((128, 1024, 10), (1024, 256, 20), (4096, 100, 50)):
    iv = torch.ones(D, device=dev)
    fn = bjx.targets.DiagGaussian(iv)
    q0 = torch.randn(N, D, device=dev)
    keys = bjx.random.split(bjx.random.key(0), 220)
    row = {}
    for name, ug in (("auto", "auto"), ("plain", False)):
        alg = bjx.hmc(fn, 0.1, torch.ones(D, device=dev), L, use_graph=ug)
        st = alg.init(q0)
        for k in keys[:20]:
            st, _ = alg.step(k, st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in keys[20:]:
            st, _ = alg.step(k, st)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        row[name] = {"ms_per_transition": dt / 200 * 1e3, "chain_leapfrogs_per_s": N * L * 200 / dt}
    row["speedup"] = row["plain"]["ms_per_transition"] / row["auto"]["ms_per_transition"]
    out[f"{N}x{D},L={L}"] = row
print(json.dumps(out))
