#!/usr/bin/env python
"""Can the RNG-bound refresh kernel of one half of a GHMC ensemble overlap the memory-bound finish kernel of the
other half?  The C2-shaped ensemble as 1 / 2 / 4 independent blocks, each on its own stream (probe only: the
blocks are separate sampler objects here)."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import blackjax_amd as bjx  # noqa: E402

dev = torch.device("cuda:0")
N, D, STEPS, WARM = 65536, 1024, 200, 30
sig = torch.as_tensor((10.0 ** (-1.0 + 2.0 * np.arange(D) / (D - 1))).astype(np.float32), device=dev)
fn = bjx.targets.DiagGaussian((1.0 / (sig * sig)).contiguous())
g = torch.Generator(device=dev)
g.manual_seed(0)
q0 = sig * torch.randn(N, D, device=dev, generator=g)
keys = bjx.random.split(bjx.random.key(1), STEPS + WARM)
out = {}
for nb in (1, 2, 4, 1):
    n = N // nb
    algs = [bjx.ghmc(fn, 0.3, sig, 0.3, 0.15, chain_offset=b * n) for b in range(nb)]
    states = [algs[b].init(q0[b * n:(b + 1) * n].contiguous(), bjx.random.key(0)) for b in range(nb)]
    streams = [torch.cuda.Stream() for _ in range(nb)]
    torch.cuda.synchronize()

    def run(ks):
        for k in ks:
            for b in range(nb):
                with torch.cuda.stream(streams[b]):
                    states[b], _ = algs[b].step(k, states[b])

    run(keys[:WARM])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(keys[WARM:])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out[f"{nb}_blocks_rep{len(out)}"] = {"ms_per_transition": dt / STEPS * 1e3, "frac_of_8TBps_at_60B": 60.0 * N * D * STEPS / dt / 8e12}
print(json.dumps(out))
