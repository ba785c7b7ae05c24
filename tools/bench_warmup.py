#!/usr/bin/env python
"""Secondary benchmark (BASELINE.json configs[3], SURVEY.md 8d "C4", one GPU's shard): HMC with
window_adaptation (per-chain dual averaging + Welford diagonal mass) on a 4 096-dim ill-conditioned
diagonal Gaussian (sigma_i = 10^(-1.5+3i/(D-1))), 32 768 chains per GPU, L = 50.
The full config is 262 144 chains = 8 such shards with no collective (chain_offset = rank * 32768)."""
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
ap.add_argument("--chains", type=int, default=32768)
ap.add_argument("--dim", type=int, default=4096)
ap.add_argument("--leapfrogs", type=int, default=50)
ap.add_argument("--num-steps", type=int, default=100)
ap.add_argument("--rank", type=int, default=0)
args = ap.parse_args()
dev = torch.device("cuda:0")
N, D, L = args.chains, args.dim, args.leapfrogs
sig = torch.as_tensor((10.0 ** (-1.5 + 3.0 * np.arange(D) / (D - 1))).astype(np.float32), device=dev)
fn = bjx.targets.DiagGaussian((1.0 / (sig * sig)).contiguous())
warm = bjx.window_adaptation(bjx.hmc, fn, adaptation_info_fn=None, num_integration_steps=L,
                             initial_step_size=0.01)
g = torch.Generator(device=dev)
g.manual_seed(args.rank)
q0 = sig * torch.randn(N, D, device=dev, generator=g)
# leapfrog launches are sampled sparsely, events come from a pre-recorded pool (NOTEBOOK.md section 5:
# creating events inside a timed region of ~50 us launches costs ~13 %)
timer = _lib.LaunchTimer(["bjx_welford_update_diag", "bjx_da_update", "bjx_leapfrog_diag"],
                         every={"bjx_leapfrog_diag": 16}, capacity=4096)
from blackjax_amd.hmc import auto_chain_block  # noqa: E402

lf_chains = min(auto_chain_block(N, D, 4), N)  # q, p, g + per-chain inverse mass matrix
_lib.set_timer(timer)
torch.cuda.synchronize()
t0 = time.perf_counter()
(state, params), _ = warm.run(bjx.random.key(0), q0, args.num_steps, chain_offset=args.rank * N)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
_lib.set_timer(None)
wel = timer.durations_ms("bjx_welford_update_diag")
lf = timer.durations_ms("bjx_leapfrog_diag")
ratio = (params["inverse_mass_matrix"] / (sig * sig))
print(json.dumps({
    "metric": "warmup chain-leapfrog-steps/s (window_adaptation, per-chain DA + Welford)",
    "value": N * L * args.num_steps / dt, "unit": "chain-leapfrog-steps/s",
    "config": {"workload": f"window_adaptation(hmc) {N} chains x {D} dims, L={L}, num_steps={args.num_steps}"},
    "seconds": dt, "welford_update_avg_us": float(np.mean(wel)) * 1e3 if wel else None,
    "welford_GBps": (20.0 * N * D / (np.mean(wel) * 1e-3) / 1e9) if wel else None,
    "leapfrog_avg_us": float(np.mean(lf)) * 1e3,
    "leapfrog_chains_per_launch": lf_chains,
    "leapfrog_GBps_24B": 24.0 * lf_chains * D / (np.mean(lf) * 1e-3) / 1e9,
    "final_step_size_mean": float(params["step_size"].mean()),
    "imm_over_sigma2_median": float(ratio.median()),
    "peak_mem_GiB": torch.cuda.max_memory_allocated() / 2**30,
}))
