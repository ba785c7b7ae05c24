#!/usr/bin/env python
"""Stage-by-stage timing of the multi-tick NUTS leaf (build-time instrumentation).
Needs a library built with `make -C blackjax_amd/csrc CXXFLAGS="... -DBJX_TICK_PROBE"` (a tools/gpu_call.sh command does
that into a scratch copy of the tree); prints the 100-MHz-tick sums of compact row 0's wave per stage."""
import ctypes
import json
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import blackjax_amd as bjx  # noqa: E402
from blackjax_amd import _lib  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 100
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda:0")
D = 256
alg = bjx.nuts(bjx.targets.NealFunnel(), 0.1, torch.ones(D, device=dev), max_num_doublings=10)
g = torch.Generator(device=dev)
g.manual_seed(0)
state0 = alg.init(0.1 * torch.randn(N, D, device=dev, generator=g))
alg.run(bjx.random.key(5), state0, 2, store_positions=False, fuse_target=True)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * 16)()
lib.bjx_debug_tick_probe.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
lib.bjx_debug_tick_probe.restype = ctypes.c_int
assert lib.bjx_debug_tick_probe(buf, 1) == 0
t0 = time.perf_counter()
_, _, info = alg.run(bjx.random.key(1), state0, T, store_positions=False, fuse_target=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
assert lib.bjx_debug_tick_probe(buf, 0) == 0
v = list(buf)
names = ["0 uniform draw", "1 pass 1 (kick, energy sum)", "2 scalars3 (progressive sampling)",
         "3 pass 2 (sum, checkpoint, stores)", "4 pass 3 (U-turn levels)", "5 continue: update + stores",
         "6 loop top (cold: fence + loads)", "7 merge + transition end", "8 merge + next doubling",
         "9 record + target stores / cold target", "10 ticks"]
n = max(v[10], 1)
tot = sum(v[:10])
print(json.dumps({"wall_s": dt, "row0_leapfrogs": int(info.num_integration_steps[:, 0].sum()),
                  "row0_ticks": v[10], "row0_us_total": tot / 100.0, "us_per_tick": tot / 100.0 / n,
                  "stages_us_per_tick": {names[k]: v[k] / 100.0 / n for k in range(10)}}, indent=1))
