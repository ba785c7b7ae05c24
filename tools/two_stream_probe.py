#!/usr/bin/env python
"""Probe: the inner loop of one 16 384-chain block (L x [leapfrog, DiagGaussian callable]) on one
stream against two 8 192-chain halves on two streams (same 192 MiB working set)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blackjax_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
N, D, L = 16384, 1024, 50
q = torch.randn(N, D, device=dev)
p = torch.randn(N, D, device=dev)
g = torch.randn(N, D, device=dev)
logp = torch.empty(N, device=dev)
imm = torch.ones(D, device=dev)
iv = torch.ones(D, device=dev)


def loop(stream, lo, hi):
    n = hi - lo
    qs, ps, gs, ls = q[lo:hi], p[lo:hi], g[lo:hi], logp[lo:hi]
    for _ in range(L):
        _lib.call("bjx_leapfrog_diag", stream, n, D, 2, 0.01, None, imm.data_ptr(), 0, qs.data_ptr(),
                  ps.data_ptr(), gs.data_ptr(), qs.data_ptr(), ps.data_ptr())
        _lib.call("bjx_target_diag_gaussian", stream, n, D, iv.data_ptr(), qs.data_ptr(), ls.data_ptr(),
                  gs.data_ptr())


s0 = torch.cuda.current_stream().cuda_stream
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for name in ("one stream, 16384", "two streams, 2 x 8192", "one stream, 16384", "two streams, 2 x 8192"):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        if name.startswith("one"):
            loop(s0, 0, N)
        else:
            # interleave the two halves' launches so neither stream runs ahead
            for _ in range(1):
                n = N // 2
                for _ in range(L):
                    for st, lo in ((sa.cuda_stream, 0), (sb.cuda_stream, n)):
                        _lib.call("bjx_leapfrog_diag", st, n, D, 2, 0.01, None, imm.data_ptr(), 0,
                                  q[lo:lo + n].data_ptr(), p[lo:lo + n].data_ptr(), g[lo:lo + n].data_ptr(),
                                  q[lo:lo + n].data_ptr(), p[lo:lo + n].data_ptr())
                        _lib.call("bjx_target_diag_gaussian", st, n, D, iv.data_ptr(), q[lo:lo + n].data_ptr(),
                                  logp[lo:lo + n].data_ptr(), g[lo:lo + n].data_ptr())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name}: {dt / (4 * L) * 1e6:.1f} us per step of 16384 chains")
