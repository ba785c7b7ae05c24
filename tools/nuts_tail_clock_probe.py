#!/usr/bin/env python
"""Is the few-live-chains tail of a free-running NUTS run slowed by the GPU's clock management?
Runs the fused-target C3 workload (T transitions) (a) alone, sampling `rocm-smi --showclocks` in the
background, and (b) with a side stream that keeps the other CUs busy with fp32 GEMMs.
usage: python tools/nuts_tail_clock_probe.py [T]"""
import json
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import blackjax_amd as bjx  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dev = torch.device("cuda:0")
N, D = 32768, 256
alg = bjx.nuts(bjx.targets.NealFunnel(), 0.1, torch.ones(D, device=dev), max_num_doublings=10)
g = torch.Generator(device=dev)
g.manual_seed(0)
state0 = alg.init(0.1 * torch.randn(N, D, device=dev, generator=g))
alg.run(bjx.random.key(5), state0, 2, store_positions=False, fuse_target=True)
torch.cuda.synchronize()


def timed():
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _, _, info = alg.run(bjx.random.key(1), state0, T, store_positions=False, fuse_target=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dt, int(info.num_integration_steps.sum())


out = {}
clocks = []
stop = threading.Event()


def poll():
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--json"], capture_output=True, text=True, timeout=5)
            j = json.loads(r.stdout)
            card = next(iter(j.values()))
            clocks.append((time.perf_counter(), {k: v for k, v in card.items() if "sclk" in k.lower() or "mclk" in k.lower()}))
        except Exception as e:  # noqa: BLE001
            clocks.append((time.perf_counter(), {"error": str(e)[:80]}))
        time.sleep(0.05)


th = threading.Thread(target=poll)
th.start()
t_start = time.perf_counter()
dt, tot = timed()
stop.set()
th.join()
out["alone"] = {"s": dt, "M_per_s": tot / dt / 1e6,
                "clock_samples": [(round(t - t_start, 2), c) for t, c in clocks][:40]}

side = torch.cuda.Stream()
a = torch.randn(4096, 4096, device=dev)
b = torch.randn(4096, 4096, device=dev)
busy = threading.Event()


def heater():
    with torch.cuda.stream(side):
        while not busy.is_set():
            for _ in range(20):
                torch.mm(a, b)
            side.synchronize()


th = threading.Thread(target=heater)
th.start()
time.sleep(0.2)
dt2, tot2 = timed()
busy.set()
th.join()
out["with_gemm_load_on_a_side_stream"] = {"s": dt2, "M_per_s": tot2 / dt2 / 1e6}
print(json.dumps(out))
