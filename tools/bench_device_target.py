#!/usr/bin/env python
"""A USER-WRITTEN device log-density (blackjax_amd.targets.DeviceTarget: HIP source compiled by hiprtc at run
time) under the external-callable contract (it is then an ordinary recordable callable between two leapfrogs)
and compiled INTO the engine's kernels (fuse_target=True).  Quartic target of tests/test_device_target.py.
HMC at the C2 shape (65 536 x 1 024, L = 50) and free-running NUTS at the C3 shape (32 768 x 256, T = 20)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import blackjax_amd as bjx  # noqa: E402
from test_device_target import QUARTIC  # noqa: E402

dev = torch.device("cuda:0")
out = {"config": {"workload": "user-written device target (quartic: logp = -sum a q^2/2 + c q^4/4, c = 0.3), "
                              "blackjax_amd.targets.DeviceTarget"}}


def target(D, c=0.3):
    g = torch.Generator(device=dev)
    g.manual_seed(D)
    a = (0.5 + torch.rand(D, device=dev, generator=g)).float()
    t0 = time.perf_counter()
    tgt = bjx.targets.DeviceTarget(QUARTIC, torch.cat([a, torch.tensor([c], device=dev)]).contiguous())
    tgt.module()
    return tgt, a, time.perf_counter() - t0


# ---- HMC, C2 shape
N, D, L = 65536, 1024, 50
tgt, a, t_compile = target(D)
out["hiprtc_compile_s"] = {"hmc_and_callable_module": t_compile}
q0 = torch.randn(N, D, device=dev) / a.sqrt()
keys = bjx.random.split(bjx.random.key(0), 16)
res = {}
for name, kw in (("external_callable", {}), ("compiled_into_the_trajectory_kernel", {"fuse_target": "lean"})):
    alg = bjx.hmc(tgt, 0.2, (1.0 / a).contiguous(), L, **kw)
    st = alg.init(q0)
    for t in range(2):
        st, info = alg.step(keys[t], st)
    torch.cuda.synchronize()
    K = 6 if not kw else 12
    t0 = time.perf_counter()
    for t in range(K):
        st, info = alg.step(keys[2 + t], st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    res[name] = {"ms_per_transition": dt * 1e3, "chain_leapfrogs_per_s": N * L / dt,
                 "mean_acceptance": float(info.acceptance_rate.mean())}
res["speedup"] = res["compiled_into_the_trajectory_kernel"]["chain_leapfrogs_per_s"] / res["external_callable"]["chain_leapfrogs_per_s"]
out["hmc_65536x1024_L50"] = res

# ---- NUTS, C3 shape
N, D, T = 32768, 256, 20
tgt, a, _ = target(D, c=0.6)
t0 = time.perf_counter()
tgt.nuts_module()
out["hiprtc_compile_s"]["nuts_module"] = time.perf_counter() - t0
alg = bjx.nuts(tgt, 0.35, (1.0 / a).contiguous(), max_num_doublings=10)
st0 = alg.init(torch.randn(N, D, device=dev))
res = {}
for name, kw in (("external_callable", {}), ("compiled_into_the_tick_kernel", {"fuse_target": True})):
    alg.run(bjx.random.key(5), st0, 2, store_positions=False, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st, _, info = alg.run(bjx.random.key(1), st0, T, store_positions=False, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tot = int(info.num_integration_steps.sum())
    res[name] = {"seconds": dt, "useful_chain_leapfrogs_per_s": tot / dt,
                 "mean_leapfrogs_per_chain_transition": tot / (N * T)}
res["speedup"] = res["compiled_into_the_tick_kernel"]["useful_chain_leapfrogs_per_s"] / res["external_callable"]["useful_chain_leapfrogs_per_s"]
out["nuts_32768x256_T20"] = res
print(json.dumps(out))
