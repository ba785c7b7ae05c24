#!/usr/bin/env python
"""NUTS with a SHARED dense inverse mass matrix (VERDICT r2 "next" #8): 512-dim AR(1) Gaussian,
16 384 chains, lockstep `step`, every product v = M^{-1} p of a leaf as one fp32 MFMA GEMM over the
live rows (`dense_gemm=True`) against one fp64 mat-vec per chain (`dense_gemm=False`, D^2 words per chain
and product).  Reports chain-leapfrog/s and, for the GEMM mode, the time and TFLOP/s of the product."""
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
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--warmup", type=int, default=1)
ap.add_argument("--eps", type=float, default=0.35)
ap.add_argument("--max-depth", type=int, default=6)
ap.add_argument("--mode", choices=["gemm", "matvec", "both"], default="both")
ap.add_argument("--run", type=int, default=0,
                help="also time alg.run(T = this many transitions) on the GEMM path: lockstep steps vs free-running "
                     "ticks (BJX_NUTS_FREE_GEMM=1)")
ap.add_argument("--target", choices=["ar1", "funnel"], default="ar1")
args = ap.parse_args()
dev = torch.device("cuda:0")
N, D = args.chains, args.dim
ar1 = bjx.targets.AR1Gaussian(0.9, D)
cov = ar1.covariance(dev)
tgt = ar1 if args.target == "ar1" else bjx.targets.NealFunnel()  # the funnel: trees of very different depths
g = torch.Generator(device=dev)
g.manual_seed(0)
q0 = torch.randn(N, D, device=dev, generator=g)
out = {"config": {"workload": f"NUTS shared dense metric (AR(1) rho=0.9 covariance), target {args.target}, D={D}, {N} chains, eps={args.eps}, "
                              f"max_depth={args.max_depth}, lockstep step (HIP-graph driver)"}}
for mode in (["gemm", "matvec"] if args.mode == "both" else [args.mode]):
    alg = bjx.nuts(tgt, args.eps, cov, max_num_doublings=args.max_depth, use_graph=True,
                   dense_gemm=(mode == "gemm"))
    state = alg.init(q0)
    keys = bjx.random.split(bjx.random.key(0), args.warmup + args.steps)
    for t in range(args.warmup):
        state, info = alg.step(keys[t], state)
    torch.cuda.synchronize()
    tot = 0
    t0 = time.perf_counter()
    for t in range(args.warmup, args.warmup + args.steps):
        state, info = alg.step(keys[t], state)
        tot += int(info.num_integration_steps.sum())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out[mode] = {"value": tot / dt, "unit": "useful chain-leapfrog-steps/s", "ms_per_transition": dt / args.steps * 1e3,
                 "mean_leapfrogs": tot / (N * args.steps), "mean_acceptance": float(info.acceptance_rate.mean()),
                 "mean_depth": float(info.num_trajectory_expansions.float().mean())}
# the product itself: one GEMM over all rows
p = torch.randn(N, D, device=dev, generator=g)
v = torch.empty_like(p)
s = _lib.current_stream()
for _ in range(3):
    _lib.call("bjx_dense_apply_imm", s, N, D, p.data_ptr(), cov.data_ptr(), v.data_ptr())
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    _lib.call("bjx_dense_apply_imm", s, N, D, p.data_ptr(), cov.data_ptr(), v.data_ptr())
b.record()
torch.cuda.synchronize()
us = a.elapsed_time(b) / 20 * 1e3
out["product_gemm"] = {"us": us, "TFLOPs": 2.0 * N * D * D / (us * 1e-6) / 1e12, "frac_of_157.3": 2.0 * N * D * D / (us * 1e-6) / 157.3e12}
if args.run:
    alg = bjx.nuts(tgt, args.eps, cov, max_num_doublings=args.max_depth, dense_gemm=True)
    st0 = alg.init(q0 if args.target == "ar1" else 0.1 * q0)
    res = {}
    for name, flag in (("lockstep_steps", "0"), ("free_running_ticks", "1")):
        os.environ["BJX_NUTS_FREE_GEMM"] = flag
        alg.run(bjx.random.key(4), st0, 2, store_positions=False)  # warm-up (kernels, graphs, allocator)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        final, _, rinfo = alg.run(bjx.random.key(5), st0, args.run, store_positions=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        tot = int(rinfo.num_integration_steps.sum())
        res[name] = {"value": tot / dt, "unit": "useful chain-leapfrog-steps/s", "ms_per_transition": dt / args.run * 1e3,
                     "mean_leapfrogs": tot / (N * args.run), "checksum": float(final.position.double().sum()),
                     "depth_histogram": torch.bincount(rinfo.num_trajectory_expansions.flatten().long()).tolist()}
    res["identical_final_positions"] = res["lockstep_steps"]["checksum"] == res["free_running_ticks"]["checksum"]
    res["speedup_free_running"] = res["free_running_ticks"]["value"] / res["lockstep_steps"]["value"]
    out["run_T%d" % args.run] = res
if "gemm" in out and "matvec" in out:
    out["speedup_gemm_vs_matvec"] = out["gemm"]["value"] / out["matvec"]["value"]
print(json.dumps(out))
