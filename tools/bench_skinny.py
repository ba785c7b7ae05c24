#!/usr/bin/env python
"""Duration of V = P imm^T (bjx_dense_apply_imm_t) against the row count: run once with BJX_DENSE_SKINNY_MAX=0 (the MFMA
kernels only) and once with a huge value (the latency-oriented kernel everywhere) to place the threshold."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blackjax_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
out = {"BJX_DENSE_SKINNY_MAX": os.environ.get("BJX_DENSE_SKINNY_MAX"), "us": {}}
s = _lib.current_stream()
for D in (128, 256, 512, 1024):
    imm = torch.randn(D, D, device=dev)
    imm_t = imm.t().contiguous()
    for M in (32, 128, 512, 1024, 2048, 4096, 8192, 16384):
        p = torch.randn(M, D, device=dev)
        v = torch.empty_like(p)
        args = ("bjx_dense_apply_imm_t", s, M, D, p.data_ptr(), imm.data_ptr(), imm_t.data_ptr(), v.data_ptr())
        for _ in range(3):
            _lib.call(*args)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50):
            _lib.call(*args)
        b.record()
        torch.cuda.synchronize()
        out["us"][f"D{D}_M{M}"] = round(a.elapsed_time(b) / 50 * 1e3, 2)
print(json.dumps(out))
