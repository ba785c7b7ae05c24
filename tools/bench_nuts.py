#!/usr/bin/env python
"""Secondary benchmark (BASELINE.json configs[2], SURVEY.md 8d "C3"): NUTS, iterative tree doubling,
max_depth = 10, Neal's funnel D = 256, 32 768 chains, eps = 0.1, imm = ones, one MI355X.
Reports useful chain-leapfrog-steps/s and lockstep utilisation = sum(chain steps) / (N * launches)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blackjax_amd as bjx  # noqa: E402
from blackjax_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--chains", type=int, default=32768)
ap.add_argument("--dim", type=int, default=256)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--eps", type=float, default=0.1)
ap.add_argument("--max-depth", type=int, default=10)
ap.add_argument("--recompact", type=int, default=16)
ap.add_argument("--use-graph", action="store_true")
ap.add_argument("--free-running", action="store_true",
                help="drive the timed transitions with alg.run (asynchronous chains) instead of step")
ap.add_argument("--no-tick-timing", action="store_true",
                help="free-running: do not bracket ticks with HIP events (the brackets drain the queue)")
ap.add_argument("--fuse-target", action="store_true",
                help="free-running: the tick kernels evaluate the built-in funnel themselves (one launch per tick; "
                     "OUTSIDE the external-callable contract -- a second, separately labelled figure)")
ap.add_argument("--run-graph", default="auto", choices=["auto", "off", "on"],
                help="free-running: HIP-graph replay of tick chunks (auto = in the tail of the run)")
args = ap.parse_args()
dev = torch.device("cuda:0")
N, D = args.chains, args.dim
alg = bjx.nuts(bjx.targets.NealFunnel(), args.eps, torch.ones(D, device=dev),
               max_num_doublings=args.max_depth, recompact_every=args.recompact,
               use_graph=args.use_graph,
               run_use_graph={"auto": "auto", "off": False, "on": True}[args.run_graph],
               fuse_target=args.fuse_target)
g = torch.Generator(device=dev)
g.manual_seed(0)
state = alg.init(0.1 * torch.randn(N, D, device=dev, generator=g))
keys = bjx.random.split(bjx.random.key(0), args.warmup + args.steps)
for t in range(args.warmup):
    state, info = alg.step(keys[t], state)
torch.cuda.synchronize()
if args.free_running:
    alg.run(bjx.random.key(5), state, 2, store_positions=False, fuse_target=args.fuse_target)  # first use of the tick kernel
    tick_timer = _lib.LaunchTimer(["bjx_nuts_async_tick"], every=8, capacity=4096)
    # events cannot be recorded while a graph is being captured: ticks are only timed without graphs
    if args.run_graph == "off" and not args.use_graph and not args.no_tick_timing:
        _lib.set_timer(tick_timer)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    state, positions, rinfo = alg.run(bjx.random.key(1), state, args.steps, fuse_target=args.fuse_target)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    _lib.set_timer(None)
    tot = int(rinfo.num_integration_steps.sum())
    per_chain = rinfo.num_integration_steps.sum(0).float()
    ticks = tick_timer.seen["bjx_nuts_async_tick"] or int(per_chain.max())
    # tick launches are only bracketed with events when no HIP graph is involved (--run-graph off);
    # otherwise the per-launch fields are null and the wall-clock period per tick is what there is
    tick_ms = tick_timer.durations_ms("bjx_nuts_async_tick")
    avg_tick_us = sum(tick_ms) / len(tick_ms) * 1e3 if tick_ms else None
    mean_us = lambda xs: (sum(xs) / len(xs) * 1e3) if xs else None  # noqa: E731
    # per window of 100 transitions: what a schedule that synchronised the chains every 100 transitions
    # could use at best = leapfrogs of the window / (N x the busiest chain's leapfrogs in it)
    util_windows = []
    for w0 in range(0, args.steps, 100):
        win = rinfo.num_integration_steps[w0:w0 + 100].sum(0).float()
        util_windows.append({"transitions": [w0, min(w0 + 100, args.steps)],
                             "utilisation": float(win.sum() / (N * win.max())),
                             "busiest_chain_leapfrogs": int(win.max()), "mean_chain_leapfrogs": float(win.mean())})
    print(json.dumps({
        "metric": "NUTS useful chain-leapfrog-steps/s", "value": tot / dt, "unit": "chain-leapfrog-steps/s",
        # an engine-resident target does not move 52 B per element through HBM and is never
        # quoted as a fraction of the HBM roofline
        "frac_of_52B_roofline": None if args.fuse_target else tot / dt / (8e12 / (52.0 * D)),
        "utilisation_per_100_transitions": util_windows,
        "config": {"workload": f"NUTS max_depth={args.max_depth}, Neal funnel D={D}, {N} chains, eps={args.eps}",
                   "driver": "free-running chains (alg.run)" + (
                       ", log-density evaluated INSIDE the tick kernels (fuse_target=True: engine-resident target, "
                       "not the external-callable contract)" if args.fuse_target else ""),
                   "hip_graph": "on" if args.use_graph else args.run_graph},
        "steps": args.steps, "ms_per_transition": dt / args.steps * 1e3,
        "mean_leapfrogs_per_chain_transition": tot / (N * args.steps),
        "ticks": ticks, "max_chain_total_leapfrogs": int(per_chain.max()),
        "utilisation": tot / (N * ticks),
        "tick_period_avg_us": dt / max(ticks, 1) * 1e6,
        "tick_kernel_timed": bool(tick_ms),
        "tick_kernel_avg_us": avg_tick_us,
        "tick_kernel_us_first_20_samples": mean_us(tick_ms[:20]),
        "tick_kernel_us_last_20_samples": mean_us(tick_ms[-20:]),
        "tick_kernel_GBps_at_52B_per_element": (52.0 * N * D / (avg_tick_us * 1e-6) / 1e9)
        if avg_tick_us and not args.fuse_target else None,
        "mean_depth": float(rinfo.num_trajectory_expansions.float().mean()),
        "frac_divergent": float(rinfo.is_divergent.float().mean()),
    }))
    sys.exit(0)
timer = _lib.LaunchTimer(["bjx_nuts_pre", "bjx_nuts_post"])
if not args.use_graph:
    _lib.set_timer(timer)
tot_steps = 0
launches = 0
t0 = time.perf_counter()
for t in range(args.warmup, args.warmup + args.steps):
    state, info = alg.step(keys[t], state)
    tot_steps += int(info.num_integration_steps.sum())
    launches = len(timer.events["bjx_nuts_pre"])
torch.cuda.synchronize()
dt = time.perf_counter() - t0
_lib.set_timer(None)
pre = timer.durations_ms("bjx_nuts_pre")
post = timer.durations_ms("bjx_nuts_post")
print(json.dumps({
    "metric": "NUTS useful chain-leapfrog-steps/s", "value": tot_steps / dt, "unit": "chain-leapfrog-steps/s",
    "config": {"workload": f"NUTS max_depth={args.max_depth}, Neal funnel D={D}, {N} chains, eps={args.eps}"
               + (", step() = one free-running transition per chain with the log-density evaluated inside "
                  "the tick kernels (fuse_target=True: engine-resident target, not the external-callable "
                  "contract)" if args.fuse_target else ""),
               "recompact_every": args.recompact},
    "steps": args.steps, "ms_per_transition": dt / args.steps * 1e3,
    "frac_of_52B_roofline": None if args.fuse_target else tot_steps / dt / (8e12 / (52.0 * D)),
    "mean_leapfrogs_per_chain_transition": tot_steps / (N * args.steps),
    "leapfrog_launches_per_transition": launches / args.steps if launches else None,
    "lockstep_utilisation_vs_uncompacted": tot_steps / (N * launches) if launches else None,
    "hip_graph": bool(args.use_graph),
    "pre_kernel_total_ms": sum(pre), "post_kernel_total_ms": sum(post),
    "mean_depth": float(info.num_trajectory_expansions.float().mean()),
    "frac_divergent": float(info.is_divergent.float().mean()),
}))
