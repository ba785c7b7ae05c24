#!/usr/bin/env python
"""Where the host time of one C2 transition goes (cProfile over 3 transitions, chain blocks of 16 384)."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blackjax_amd as bjx  # noqa: E402

dev = torch.device("cuda:0")
N, D, L = 65536, 1024, 50
imm = torch.ones(D, device=dev)
fn = bjx.targets.DiagGaussian(imm.clone())
alg = bjx.hmc(fn, 0.25, imm, L, chain_block=16384, use_graph=False)
st = alg.init(torch.randn(N, D, device=dev))
keys = bjx.random.split(bjx.random.key(0), 6)
for k in keys[:2]:
    st, _ = alg.step(k, st)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for k in keys[2:5]:
    st, _ = alg.step(k, st)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
