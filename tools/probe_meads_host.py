#!/usr/bin/env python
"""MEADS step at 4 096 x 1 024: wall time per step against the GPU's own time (run under rocprofv3 --stats for
the latter) -- is the step host-bound?"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import blackjax_amd as bjx  # noqa: E402

dev = torch.device("cuda:0")
N, D, STEPS = 4096, 1024, 200
sig = torch.as_tensor((10.0 ** (-1.0 + 2.0 * np.arange(D) / (D - 1))).astype(np.float32), device=dev)
fn = bjx.targets.DiagGaussian((1.0 / (sig * sig)).contiguous())
g = torch.Generator(device=dev)
g.manual_seed(0)
qm = (sig * torch.randn(N, D, device=dev, generator=g)).contiguous()
warm = bjx.meads_adaptation(fn, N, num_folds=4, adaptation_info_fn=None)
warm.run(bjx.random.key(2), qm, 10)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
e0.record()
warm.run(bjx.random.key(3), qm, STEPS)
t_issue = time.perf_counter() - t0
e1.record()
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(json.dumps({"steps": STEPS, "wall_ms_per_step": wall / STEPS * 1e3, "host_issue_ms_per_step": t_issue / STEPS * 1e3,
                  "event_ms_per_step": e0.elapsed_time(e1) / STEPS}))
