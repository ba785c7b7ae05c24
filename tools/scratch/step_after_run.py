import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import blackjax_amd as bjx
from blackjax_amd import _util, nuts as bn, hmc as bh
count = {"n": 0}
orig = _util.record_graph
def counting(graph, **kw):
    count["n"] += 1
    return orig(graph, **kw)
bn.record_graph = counting
bh.record_graph = counting
dev = torch.device("cuda:0")
N, D = 32768, 256
alg = bjx.nuts(bjx.targets.NealFunnel(), 0.1, torch.ones(D, device=dev), max_num_doublings=10)
st = alg.init(0.1 * torch.randn(N, D, device=dev, generator=torch.Generator(device=dev).manual_seed(0)))
keys = bjx.random.split(bjx.random.key(0), 40)
k = 0
def steps(n, tag):
    global st, k
    for _ in range(n):
        c0 = count["n"]; torch.cuda.synchronize(); t0 = time.perf_counter()
        st, info = alg.step(keys[k], st); k += 1
        torch.cuda.synchronize()
        print(tag, "step ms", round((time.perf_counter() - t0) * 1e3, 2), "recordings", count["n"] - c0, "max leaves", int(info.num_integration_steps.max()))
steps(5, "cold")
c0 = count["n"]; alg.run(bjx.random.key(5), st, 2, store_positions=False); torch.cuda.synchronize(); print("run(T=2) recordings", count["n"] - c0)
steps(3, "after run(2)")
c0 = count["n"]; alg.run(bjx.random.key(6), st, 400, store_positions=False); torch.cuda.synchronize(); print("run(T=400) recordings", count["n"] - c0)
steps(3, "after run(400)")
c0 = count["n"]; alg.run(bjx.random.key(7), st, 100, store_positions=False); torch.cuda.synchronize(); print("run(T=100) again recordings", count["n"] - c0)
steps(3, "after run(100) #2")
