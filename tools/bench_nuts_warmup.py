#!/usr/bin/env python
"""NUTS warm-up (window_adaptation: per-chain dual averaging + Welford) at the C3 shape -- Neal's
funnel D = 256, 32 768 chains, max_depth = 10 -- lockstep (`run`) against free-running chains
(`run(..., free_running=True)`): same results bit for bit, different schedule."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blackjax_amd as bjx  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--chains", type=int, default=32768)
ap.add_argument("--dim", type=int, default=256)
ap.add_argument("--num-steps", type=int, default=60)
ap.add_argument("--max-depth", type=int, default=10)
ap.add_argument("--initial-step-size", type=float, default=0.1)
ap.add_argument("--skip-lockstep", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda:0")
N, D, T = args.chains, args.dim, args.num_steps
fn = bjx.targets.NealFunnel()
g = torch.Generator(device=dev)
g.manual_seed(0)
q0 = 0.1 * torch.randn(N, D, device=dev, generator=g)
warm = bjx.window_adaptation(bjx.nuts, fn, adaptation_info_fn=None, initial_step_size=args.initial_step_size,
                             max_num_doublings=args.max_depth)
key = bjx.random.key(0)
warm.run(bjx.random.key(9), q0[:1024].contiguous(), 20, free_running=True)  # first use of the kernels
warm.run(bjx.random.key(9), q0[:1024].contiguous(), 20)
torch.cuda.synchronize()
out = {"metric": "NUTS warm-up useful chain-leapfrog-steps/s", "unit": "chain-leapfrog-steps/s",
       "config": {"workload": f"window_adaptation(nuts) Neal funnel D={D}, {N} chains, {T} steps, "
                              f"max_depth={args.max_depth}, initial step size {args.initial_step_size}"}}
t0 = time.perf_counter()
(st_f, par_f), info = warm.run(key, q0, T, free_running=True)
torch.cuda.synchronize()
dt_f = time.perf_counter() - t0
tot = int(info.num_integration_steps.sum())
out["free_running"] = {"seconds": dt_f, "value": tot / dt_f, "total_leapfrogs": tot,
                       "mean_leapfrogs_per_chain_transition": tot / (N * T),
                       "max_chain_total_leapfrogs": int(info.num_integration_steps.sum(0).max()),
                       "final_step_size_median": float(par_f["step_size"].median())}
t0 = time.perf_counter()
(st_e, par_e), info_e = warm.run(key, q0, T, free_running=True, fuse_target=True)
torch.cuda.synchronize()
dt_e = time.perf_counter() - t0
out["free_running_engine_resident_target"] = {
    "seconds": dt_e, "value": tot / dt_e,
    "note": "fuse_target=True: the funnel evaluated inside the tick kernels (outside the external-callable contract)",
    "identical_to_free_running": bool(torch.equal(st_e.position, st_f.position)
                                      and torch.equal(par_e["step_size"], par_f["step_size"])
                                      and torch.equal(par_e["inverse_mass_matrix"], par_f["inverse_mass_matrix"]))}
if not args.skip_lockstep:
    t0 = time.perf_counter()
    (st_l, par_l), _ = warm.run(key, q0, T)
    torch.cuda.synchronize()
    dt_l = time.perf_counter() - t0
    out["lockstep"] = {"seconds": dt_l, "value": tot / dt_l}
    out["identical_results"] = bool(torch.equal(st_l.position, st_f.position)
                                    and torch.equal(par_l["step_size"], par_f["step_size"])
                                    and torch.equal(par_l["inverse_mass_matrix"], par_f["inverse_mass_matrix"]))
    out["speedup"] = dt_l / dt_f
print(json.dumps(out))
