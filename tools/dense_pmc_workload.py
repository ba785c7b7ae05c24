#!/usr/bin/env python
"""Workload for tools/pmc_dense5.sh (round 5): the three C5 launches side by side --
the fused dense leapfrog (k_dense_gemm_tn8<EPI_DRIFT, 2>: kick prologue, v = M^-1 p, drift epilogue), the plain
store-only GEMM on the same core loop (k_dense_gemm_tn8<EPI_STORE, 0>, bjx_dense_apply_imm) and the vendor
library's plain fp32 GEMM (torch.mm, TF32 off) -- REPS launches each at 16 384 x 512."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blackjax_amd as bjx  # noqa: E402
from blackjax_amd import _lib  # noqa: E402

REPS = int(os.environ.get("BJX_PMC_REPS", "12"))
N, D = int(os.environ.get("BJX_PMC_ROWS", "16384")), 512
dev = torch.device("cuda:0")
torch.backends.cuda.matmul.allow_tf32 = False
tgt = bjx.targets.AR1Gaussian(0.9, D)
cov = tgt.covariance(dev).contiguous()
g = torch.Generator(device=dev)
g.manual_seed(0)
q = torch.randn(N, D, device=dev, generator=g)
p = torch.randn(N, D, device=dev, generator=g)
gr = torch.randn(N, D, device=dev, generator=g)
q2, p2, v = torch.empty_like(q), torch.empty_like(p), torch.empty_like(p)
s = torch.cuda.current_stream().cuda_stream
for _ in range(REPS):
    _lib.call("bjx_leapfrog_dense", s, N, D, 2, 0.01, None, cov.data_ptr(), q.data_ptr(), p.data_ptr(), gr.data_ptr(),
              q2.data_ptr(), p2.data_ptr())
torch.cuda.synchronize()
for _ in range(REPS):
    _lib.call("bjx_dense_apply_imm", s, N, D, p.data_ptr(), cov.data_ptr(), v.data_ptr())
torch.cuda.synchronize()
for _ in range(REPS):
    torch.mm(p, cov, out=v)
torch.cuda.synchronize()
print("ok")
