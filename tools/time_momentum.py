#!/usr/bin/env python
"""Times the momentum draw at C2 (65 536 x 1 024): bjx_hmc_momentum_diag (threefry + ErfInv32 per element, 4 B per
element written) and bjx_hmc_momentum_kick_diag (the same + the trajectory's first kick and drift: 20 B per element, the
form the C2 transition uses), HIP-event averages over 20 launches, plus a checksum of the draws (variants must agree)."""
import hashlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blackjax_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
N, D = 65536, 1024
imm = torch.rand(D, device=dev, generator=torch.Generator(device=dev).manual_seed(0)) + 0.5
p = torch.empty(N, D, device=dev)
ke = torch.empty(N, device=dev)
q0 = torch.randn(N, D, device=dev)
g0 = torch.randn(N, D, device=dev)
q1 = torch.empty_like(q0)
ph = torch.empty_like(q0)
s = _lib.current_stream()


def draw(rep):
    _lib.call("bjx_hmc_momentum_diag", s, 1, 2 + rep, 0, -1, N, D, imm.data_ptr(), 0, p.data_ptr(), ke.data_ptr())


def draw_kick(rep):
    _lib.call("bjx_hmc_momentum_kick_diag", s, 1, 2 + rep, 0, -1, N, D, imm.data_ptr(), 0, 0.25, 0, q0.data_ptr(),
              g0.data_ptr(), p.data_ptr(), ke.data_ptr(), q1.data_ptr(), ph.data_ptr())


out = {}
for name, fn in (("momentum_diag_us", draw), ("momentum_kick_diag_us", draw_kick)):
    for rep in range(3):
        fn(rep)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for rep in range(20):
        fn(rep)
    b.record()
    torch.cuda.synchronize()
    out[name] = a.elapsed_time(b) / 20 * 1e3
draw(0)
torch.cuda.synchronize()
out["sha256_of_draw_key_1_2"] = hashlib.sha256(p.cpu().numpy().tobytes()).hexdigest()[:16]
out["mean"], out["std"] = float(p.mean()), float(p.std())
out["hbm_bound_us_at_6.4TBps"] = {"momentum_diag": 4 * N * D / 6.4e6, "momentum_kick_diag": 20 * N * D / 6.4e6}
print(json.dumps(out))
