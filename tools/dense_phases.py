#!/usr/bin/env python
"""Round 6: per-phase timeline of the C5 dense leapfrog's K-tile loop from a PROBE2 build (`-DBJX_DENSE_PROBE2`: shader
clock stamps of waves 0 and 4 -- one SIMD pair -- at four points of every K-tile).  Ping-pong loop (BJX_DENSE_PP=1):
stamps = phase A start, phase A work issued, phase B start, phase B work issued; lockstep loop (BJX_DENSE_PP=0): tile
start, first 8 MFMAs issued, staging issued, last 8 MFMAs issued.
usage: BJX_DENSE_PP=0|1 python tools/dense_phases.py <libbjxhip_probe2.so>   JSON -> stdout"""
import ctypes
import json
import os
import sys

import numpy as np
import torch

lib = ctypes.CDLL(sys.argv[1])
N, D = 16384, 512
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(0)
idx = torch.arange(D, device=dev)
cov = (0.9 ** (idx[:, None] - idx[None, :]).abs().float()).contiguous()
q, p, gr = (torch.randn(N, D, device=dev, generator=g) for _ in range(3))
q2, p2 = torch.empty_like(q), torch.empty_like(p)
n_wg = (N // 128) * (D // 128)
stamps = torch.zeros(n_wg * 256, dtype=torch.int64, device=dev)
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
lib.bjx_dense_probe_set.argtypes = [ctypes.c_void_p]


def fused():
    rc = lib.bjx_leapfrog_dense(s, ctypes.c_int64(N), ctypes.c_int64(D), ctypes.c_int(2), ctypes.c_float(0.01), None,
                                P(cov), P(q), P(p), P(gr), P(q2), P(p2))
    assert rc == 0


lib.bjx_dense_probe_set(None)
for _ in range(3):
    fused()
torch.cuda.synchronize()
lib.bjx_dense_probe_set(P(stamps))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
fused()
e1.record()
torch.cuda.synchronize()
lib.bjx_dense_probe_set(None)
st = stamps.cpu().numpy().reshape(n_wg, 2, 32, 4).astype(np.float64)  # [wg, role (wave 0 / wave 4), tile, stamp]


def q_(x):
    x = np.asarray(x).ravel()
    return {"p10": float(np.percentile(x, 10)), "median": float(np.median(x)), "p90": float(np.percentile(x, 90))}


out = {"pp": os.environ.get("BJX_DENSE_PP", "1"), "event_us": e0.elapsed_time(e1) * 1e3, "unit": "shader cycles (s_memtime)",
       "note": "role 0 = wave 0 (rows 0-63), role 1 = wave 4 (rows 64-127, same SIMD); tiles 2..29 of every workgroup"}
tl = slice(2, 30)
for r in (0, 1):
    d = {}
    d["seg0 (stamp0->1)"] = q_(st[:, r, tl, 1] - st[:, r, tl, 0])
    d["seg1 (stamp1->2)"] = q_(st[:, r, tl, 2] - st[:, r, tl, 1])
    d["seg2 (stamp2->3)"] = q_(st[:, r, tl, 3] - st[:, r, tl, 2])
    d["seg3 (stamp3->next tile's stamp0)"] = q_(st[:, r, 3:31, 0] - st[:, r, tl, 3])
    d["tile period"] = q_(st[:, r, 3:31, 0] - st[:, r, tl, 0])
    out[f"role{r}"] = d
main = st[:, 0, 31, 3] - st[:, 0, 0, 0]
out["main_loop_cycles"] = q_(main)
# the workgroup that ends its loop first on ... (no placement info here): split at the median of the loop length
fast = main <= np.median(main)
for name, sel in (("faster_half_of_workgroups", fast), ("slower_half_of_workgroups", ~fast)):
    out[name] = {"tile period role0": q_(st[sel][:, 0, 3:31, 0] - st[sel][:, 0, tl, 0]),
                 "seg0 role0": q_(st[sel][:, 0, tl, 1] - st[sel][:, 0, tl, 0]),
                 "seg2 role1": q_(st[sel][:, 1, tl, 3] - st[sel][:, 1, tl, 2])}
print(json.dumps(out, indent=1))
