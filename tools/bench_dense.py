#!/usr/bin/env python
"""Secondary benchmark (BASELINE.json configs[4], SURVEY.md 8d "C5"): dense mass-matrix HMC on a
512-dim AR(1) correlated Gaussian (Sigma_ij = 0.9^|i-j|), 16 384 chains, L = 20, eps = 0.5.
Bound: fp32 MFMA (2*D^2 flop per chain-leapfrog); peak 157.3 TFLOP/s (MI355X_MICROARCH.md)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blackjax_amd as bjx  # noqa: E402
from blackjax_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--chains", type=int, default=16384)
ap.add_argument("--dim", type=int, default=512)
ap.add_argument("--leapfrogs", type=int, default=20)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--streams", default="auto", help="chain blocks in flight (auto = 2 for a shared dense metric)")
args = ap.parse_args()
dev = torch.device("cuda:0")
N, D, L = args.chains, args.dim, args.leapfrogs
tgt = bjx.targets.AR1Gaussian(0.9, D)
cov = tgt.covariance(dev)
alg = bjx.hmc(tgt, 0.5, cov, L, streams=args.streams if args.streams == "auto" else int(args.streams))
g = torch.Generator(device=dev)
g.manual_seed(0)
state = alg.init(torch.randn(N, D, device=dev, generator=g))
keys = bjx.random.split(bjx.random.key(0), args.warmup + args.steps)
for t in range(args.warmup):
    state, info = alg.step(keys[t], state)
torch.cuda.synchronize()
timer = _lib.LaunchTimer(["bjx_leapfrog_dense"], every=4, capacity=L * args.steps)
_lib.set_timer(timer)
acc = 0.0
t0 = time.perf_counter()
for t in range(args.warmup, args.warmup + args.steps):
    state, info = alg.step(keys[t], state)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
_lib.set_timer(None)
d = timer.durations_ms("bjx_leapfrog_dense")
avg = float(np.mean(d)) * 1e-3
flops = 2.0 * N * D * D
print(json.dumps({
    "metric": "dense-mass HMC chain-leapfrog-steps/s", "value": N * L * args.steps / dt,
    "unit": "chain-leapfrog-steps/s",
    "config": {"workload": f"dense HMC, AR(1) rho=0.9 D={D}, {N} chains, L={L}, eps=0.5",
               "streams": args.streams},
    "ms_per_transition": dt / args.steps * 1e3, "mean_acceptance": float(info.acceptance_rate.mean()),
    "roofline": {"bound": "mfma", "kernel": "k_dense_gemm<EPI_DRIFT>", "achieved": flops / avg / 1e12,
                 "peak": 157.3, "unit": "TFLOP/s", "frac": flops / avg / 1e12 / 157.3,
                 "avg_launch_us": avg * 1e6, "flops_per_launch": flops},
}))
