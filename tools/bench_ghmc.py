#!/usr/bin/env python
"""Secondary benchmark: Generalized HMC transitions (one leapfrog each) at the C2 shape, and a short MEADS
warm-up at 4 096 chains x 1 024 dims (fold statistics = two D x D fp64 Gram matrices per fold and step).

Algorithmic bytes of a GHMC transition per (chain, dim) element (shared scale; +4 per launch for a
per-chain one):
  refresh + kick + drift  r p_prev, g0, q0     w p, p_half, q1          24 B  (one normal draw per element: VALU-bound)
  callable                r q1                 w g1                      8 B
  finish                  r p_half, g1, q1     w q, g, p, p_end         28 B  (accepted chain, D <= 1 024)
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

ap = argparse.ArgumentParser()
ap.add_argument("--chains", type=int, default=65536)
ap.add_argument("--dim", type=int, default=1024)
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--warmup", type=int, default=30, help="untimed transitions: the caching allocator needs a\n"
                "few dozen steps of 256 MiB temporaries before it stops calling hipMalloc")
ap.add_argument("--meads-chains", type=int, default=4096)
ap.add_argument("--meads-steps", type=int, default=40)
args = ap.parse_args()
dev = torch.device("cuda:0")
N, D = args.chains, args.dim
sig = torch.as_tensor((10.0 ** (-1.0 + 2.0 * np.arange(D) / (D - 1))).astype(np.float32), device=dev)
fn = bjx.targets.DiagGaussian((1.0 / (sig * sig)).contiguous())
g = torch.Generator(device=dev)
g.manual_seed(0)
q0 = sig * torch.randn(N, D, device=dev, generator=g)
alg = bjx.ghmc(fn, 0.3, sig, 0.3, 0.15)
state = alg.init(q0, bjx.random.key(0))
keys = bjx.random.split(bjx.random.key(1), args.steps + args.warmup)
acc = torch.zeros((), device=dev)
for k in keys[:args.warmup]:
    state, info = alg.step(k, state)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in keys[args.warmup:]:
    state, info = alg.step(k, state)
    acc += info.acceptance_rate.mean()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
bytes_per_elem = 24 + 8 + 28
out = {
    "metric": "GHMC chain-leapfrog-steps/s (one leapfrog per transition)",
    "value": N * args.steps / dt, "unit": "chain-leapfrog-steps/s",
    "config": {"workload": f"blackjax_amd.ghmc {N} chains x {D} dims, shared (D,) scale, {args.steps} transitions"},
    "ms_per_transition": dt / args.steps * 1e3,
    "mean_acceptance": float(acc) / args.steps,
    "algorithmic_bytes_per_element": bytes_per_elem,
    "achieved_GBps": bytes_per_elem * N * D * args.steps / dt / 1e9,
    "frac_of_8TBps": bytes_per_elem * N * D * args.steps / dt / 8e12,
}
Nm = args.meads_chains
warm = bjx.meads_adaptation(fn, Nm, num_folds=4, adaptation_info_fn=None)
qm = (sig * torch.randn(Nm, D, device=dev, generator=g)).contiguous()
warm.run(bjx.random.key(2), qm, 4)
torch.cuda.synchronize()
t0 = time.perf_counter()
(st, params), _ = warm.run(bjx.random.key(3), qm, args.meads_steps)
torch.cuda.synchronize()
dtm = time.perf_counter() - t0
out["meads"] = {"chains": Nm, "dim": D, "steps": args.meads_steps, "ms_per_step": dtm / args.meads_steps * 1e3,
                "chain_leapfrogs_per_s": Nm * args.meads_steps / dtm,
                "final_step_size": float(params["step_size"]), "final_alpha": float(params["alpha"])}
print(json.dumps(out))
