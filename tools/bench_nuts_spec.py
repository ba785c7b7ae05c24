#!/usr/bin/env python
"""A/B of the two-stream speculative tail (run_free(spec_rows=...)) at the C3 shape: NUTS, Neal's funnel D = 256,
32 768 chains, eps = 0.1, max_depth = 10, external-callable contract.  Same call, alternating runs; prints one JSON
line with useful chain-leapfrog-steps/s per setting, the tail counters and whether the records are identical."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blackjax_amd as bjx  # noqa: E402
from blackjax_amd import _nuts as bnuts  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--chains", type=int, default=32768)
ap.add_argument("--dim", type=int, default=256)
ap.add_argument("--T", type=int, nargs="+", default=[20, 100])
ap.add_argument("--spec", type=int, nargs="+", default=[0, 128])
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--max-depth", type=int, default=10)
args = ap.parse_args()
dev = torch.device("cuda:0")
N, D = args.chains, args.dim
fn = bjx.targets.NealFunnel()
imm = torch.ones(D, device=dev)
g = torch.Generator(device=dev)
g.manual_seed(0)
alg = bjx.nuts(fn, 0.1, imm, max_num_doublings=args.max_depth)
state = alg.init(0.1 * torch.randn(N, D, device=dev, generator=g))
for k in bjx.random.split(bjx.random.key(0), 4):
    state, _ = alg.step(k, state)
out = {"chains": N, "dim": D, "runs": []}
for T in args.T:
    ref = None
    for rep in range(args.reps):
        for sr in args.spec:
            if rep == 0:  # warm
                bnuts.run_free(bjx.random.key(5), state, fn, 0.1, imm, 2, args.max_depth, store_positions=False, spec_rows=sr)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fin, _, ri = bnuts.run_free(bjx.random.key(1), state, fn, 0.1, imm, T, args.max_depth,
                                        store_positions=False, spec_rows=sr)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            tot = int(ri.num_integration_steps.sum())
            busiest = int(ri.num_integration_steps.sum(0).max())
            same = None
            if ref is None:
                ref = (fin.position.clone(), ri.num_integration_steps.clone(), ri.energy.clone())
            else:
                same = bool(torch.equal(ref[0], fin.position) and torch.equal(ref[1], ri.num_integration_steps)
                            and torch.equal(torch.nan_to_num(ref[2]), torch.nan_to_num(ri.energy)))
            out["runs"].append({"T": T, "spec_rows": sr, "rep": rep, "M_per_s": tot / dt / 1e6, "seconds": dt,
                                "busiest_chain_leapfrogs": busiest, "period_us_lower_bound": dt / busiest * 1e6,
                                "identical_to_first": same, "spec": dict(bnuts._SPEC_STATS)})
print(json.dumps(out))
