#!/usr/bin/env python
"""Round 5: per-workgroup timeline of the C5 dense leapfrog launch (k_dense_gemm_tn8<EPI_DRIFT, 2> and the plain
<EPI_STORE, 0>) from a PROBE build of the library (`-DBJX_DENSE_PROBE`: four wall-clock stamps per workgroup at
100 MHz -- entry, first K-tile staged, end of the main loop, end of the epilogue incl. its stores' acknowledgement).
usage: python tools/dense_timeline.py <libbjxhip_probe.so>      JSON -> stdout (profiles/r05/dense_c5_timeline.json)
Build the variant with tools/build_variant.sh probe -DBJX_DENSE_PROBE (cross-compiles here, travels with gpurun)."""
import ctypes
import json
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
q2, p2, v = torch.empty_like(q), torch.empty_like(p), torch.empty_like(p)
n_wg = (N // 128) * (D // 128)
stamps = torch.zeros(n_wg * 8, dtype=torch.int64, device=dev)
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
lib.bjx_dense_probe_set.argtypes = [ctypes.c_void_p]


def fused():
    rc = lib.bjx_leapfrog_dense(s, ctypes.c_int64(N), ctypes.c_int64(D), ctypes.c_int(2), ctypes.c_float(0.01), None,
                                P(cov), P(q), P(p), P(gr), P(q2), P(p2))
    assert rc == 0


def plain():
    rc = lib.bjx_dense_apply_imm(s, ctypes.c_int64(N), ctypes.c_int64(D), P(p), P(cov), P(v))
    assert rc == 0


def stats(x):
    x = np.asarray(x, dtype=np.float64) / 100.0  # 100 MHz ticks -> microseconds
    return {"min": float(x.min()), "p10": float(np.percentile(x, 10)), "median": float(np.median(x)),
            "p90": float(np.percentile(x, 90)), "max": float(x.max())}


import os
out = {"kernel": "k_dense_gemm_tn8 (128 x 128, eight waves, two workgroups per CU)", "workgroups": n_wg, "clock": "wall_clock64 (100 MHz); all figures in microseconds; 512 workgroups, two per CU, one round",
       "stamps": ["entry", "first K-tile staged (after the first barrier)", "end of main loop", "end (stores acknowledged)"]}
for name, fn in (("fused<EPI_DRIFT,2>", fused), ("plain<EPI_STORE,0>", plain)):
    runs = []
    for rep in range(6):
        lib.bjx_dense_probe_set(None)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        lib.bjx_dense_probe_set(P(stamps))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        raw = stamps.cpu().numpy().reshape(n_wg, 8)
        t = raw[:, :4]
        second = (raw[:, 4] & 0xFFF) != 0
        t0 = t[:, 0].min()
        runs.append({"event_us": e0.elapsed_time(e1) * 1e3,
                     "span_first_entry_to_last_end": float(t[:, 3].max() - t0) / 100.0,
                     "entry_skew": stats(t[:, 0] - t0), "prologue": stats(t[:, 1] - t[:, 0]),
                     "main_loop": stats(t[:, 2] - t[:, 1]), "epilogue": stats(t[:, 3] - t[:, 2]),
                     "main_loop_end_skew": stats(t[:, 2] - t[:, 2].min()),
                     "end_skew": stats(t[:, 3].max() - t[:, 3]),
                     "placed_second_workgroups": int(second.sum()),
                     "lds_alloc_values": sorted(set(int(x) for x in raw[:, 4]))[:8],
                     "main_loop_placed_first": stats((t[:, 2] - t[:, 1])[~second]) if (~second).any() else None,
                     "main_loop_placed_second": stats((t[:, 2] - t[:, 1])[second]) if second.any() else None})
    lib.bjx_dense_probe_set(None)
    runs.sort(key=lambda r: r["span_first_entry_to_last_end"])
    out[name] = {"median_run": runs[len(runs) // 2], "spans_us": [r["span_first_entry_to_last_end"] for r in runs]}
print(json.dumps(out, indent=1))
