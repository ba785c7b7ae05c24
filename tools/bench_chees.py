#!/usr/bin/env python
"""Secondary benchmark: pooled ChEES-HMC warm-up on the C2 target
(65 536 chains x 1 024 dims, diagonal Gaussian sigma_i = 10^(-1+2i/(D-1))), one GPU.

Reports whole-run chain-leapfrog/s and, for the pooled-statistics kernels of include/bjx_pool.h, the
average launch time against their algorithmic bytes (HBM roofline):
  bjx_chees_weights_colstats  (weights fused in) reads q_prop, q_init   8 B / element
  bjx_chees_weights    reads q_prop                      4 B / element  (D > 1024 only)
  bjx_chees_colstats   reads q_prop, q_init              8 B / element
  bjx_chees_criterion  reads q_prop, p_prop, q_init     12 B / element
  bjx_pool_colsum      reads x                           4 B / element (x2 per step when estimating the metric)
"""
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
ap.add_argument("--chains", type=int, default=65536)
ap.add_argument("--dim", type=int, default=1024)
ap.add_argument("--num-steps", type=int, default=60)
ap.add_argument("--step-size", type=float, default=0.05)
ap.add_argument("--mass-matrix", action="store_true", help='mass_matrix_estimation="diagonal" (no length floor)')
ap.add_argument("--length-floor", action="store_true", help="also the slow-direction length floor (D x D block)")
ap.add_argument("--fuse-target", action="store_true",
                help="every warm-up transition as ONE launch with the built-in Gaussian evaluated in registers "
                     "(engine-resident target: OUTSIDE the external-callable contract, a separately labelled figure)")
args = ap.parse_args()
dev = torch.device("cuda:0")
N, D = args.chains, args.dim
sig = torch.as_tensor((10.0 ** (-1.0 + 2.0 * np.arange(D) / (D - 1))).astype(np.float32), device=dev)
fn = bjx.targets.DiagGaussian((1.0 / (sig * sig)).contiguous())
kw = {}
if args.mass_matrix or args.length_floor:
    kw = {"mass_matrix_estimation": "diagonal", "_length_floor": bool(args.length_floor),
          "mass_matrix_window_fraction": 0.25}
warm = bjx.chees_adaptation(fn, N, adaptation_info_fn=bjx.adaptation.get_filter_adapt_info_fn(
    set(), {"num_integration_steps"}, {"step_size", "trajectory_length"}), fuse_target=args.fuse_target, **kw)
g = torch.Generator(device=dev)
g.manual_seed(0)
q0 = sig * torch.randn(N, D, device=dev, generator=g)
opt = bjx.optim.adam(0.5, b1=0, b2=0.95)
# priming run (first use of every kernel / torch op)
warm.run(bjx.random.key(1), q0, args.step_size, opt, 3)
names = ["bjx_chees_weights_colstats", "bjx_chees_weights", "bjx_chees_colstats", "bjx_chees_criterion", "bjx_pool_colsum", "bjx_leapfrog_diag"]
# short leapfrog launches are sampled sparsely and all events come from a pre-recorded pool: creating
# events inside the timed region slowed bench.py's region by 13 % (NOTEBOOK.md section 5)
timer = _lib.LaunchTimer(names, every={"bjx_leapfrog_diag": 16}, capacity=4096)
_lib.set_timer(timer)
torch.cuda.synchronize()
t0 = time.perf_counter()
(state, params), info = warm.run(bjx.random.key(0), q0, args.step_size, opt, args.num_steps)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
_lib.set_timer(None)
L_total = int(info.info.num_integration_steps.sum())
bytes_per_elem = {"bjx_chees_weights_colstats": 8, "bjx_chees_weights": 4, "bjx_chees_colstats": 8, "bjx_chees_criterion": 12,
                  "bjx_pool_colsum": 4, "bjx_leapfrog_diag": 20}
from blackjax_amd.hmc import auto_chain_block  # noqa: E402

lf_chains = min(auto_chain_block(N, D, 3), N)
kernels = {}
pool_ms = 0.0
for n in names:
    d = timer.durations_ms(n)
    if not d:
        continue
    avg = float(np.mean(d))
    rows = lf_chains if n == "bjx_leapfrog_diag" else N  # the leapfrog runs chain block by chain block
    kernels[n] = {"launches": len(d), "avg_us": avg * 1e3, "chains_per_launch": rows,
                  "GBps": bytes_per_elem[n] * rows * D / (avg * 1e-3) / 1e9,
                  "frac_of_8TBps": bytes_per_elem[n] * rows * D / (avg * 1e-3) / 8e12}
    if n != "bjx_leapfrog_diag":
        pool_ms += float(np.sum(d))
ratio = params["inverse_mass_matrix"] / (sig * sig)
print(json.dumps({
    "metric": "ChEES warm-up chain-leapfrog-steps/s (pooled step size + trajectory length)",
    "value": N * L_total / dt, "unit": "chain-leapfrog-steps/s",
    "config": {"workload": f"chees_adaptation {N} chains x {D} dims, {args.num_steps} steps, {kw or 'identity metric'}"},
    "seconds": dt, "total_leapfrogs_per_chain": L_total,
    "pooled_statistics_ms_per_step": pool_ms / args.num_steps,
    "pooled_statistics_share_of_wall": pool_ms * 1e-3 / dt,
    "kernels": kernels,
    "final_step_size": params["step_size"], "final_num_leapfrog": params["integration_steps_params"][0],
    "imm_over_sigma2_median": float(ratio.median()),
    "peak_mem_GiB": torch.cuda.max_memory_allocated() / 2**30,
}))
