#!/usr/bin/env python
"""Times bjx_hmc_momentum_diag (threefry + ErfInv32 per element, once per transition) at C2."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blackjax_amd import _lib
dev = torch.device("cuda:0")
N, D = 65536, 1024
imm = torch.rand(D, device=dev) + 0.5
p = torch.empty(N, D, device=dev); ke = torch.empty(N, device=dev)
s = _lib.current_stream()
for rep in range(3):
    _lib.call("bjx_hmc_momentum_diag", s, 1, 2, 0, -1, N, D, imm.data_ptr(), 0, p.data_ptr(), ke.data_ptr())
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for rep in range(20):
    _lib.call("bjx_hmc_momentum_diag", s, 1, 2 + rep, 0, -1, N, D, imm.data_ptr(), 0, p.data_ptr(), ke.data_ptr())
b.record(); torch.cuda.synchronize()
print("momentum_diag us:", a.elapsed_time(b) / 20 * 1e3, "mean", float(p.mean()), "std", float(p.std()))
