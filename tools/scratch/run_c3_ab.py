import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import blackjax_amd as bjx
dev = torch.device("cuda:0")
N, D = 32768, 256
alg = bjx.nuts(bjx.targets.NealFunnel(), 0.1, torch.ones(D, device=dev), max_num_doublings=10, use_graph="auto")
g = torch.Generator(device=dev); g.manual_seed(0)
state = alg.init(0.1 * torch.randn(N, D, device=dev, generator=g))
keys = bjx.random.split(bjx.random.key(0), 16)
for t in range(4):
    state, info = alg.step(keys[t], state)
alg.run(bjx.random.key(5), state, 2, store_positions=False)
torch.cuda.synchronize()
for T in (100, 100, 400, 400, 100):
    t0 = time.perf_counter()
    _, _, ri = alg.run(bjx.random.key(1), state, T, store_positions=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"T={T} {dt*1e3:.1f} ms  {float(ri.num_integration_steps.sum())/dt/1e6:.1f} M/s", flush=True)
st = state
for t in range(2):
    st, _ = alg.step(keys[t], st)
torch.cuda.synchronize()
t0 = time.perf_counter(); tot = 0
for t in range(8):
    st, inf = alg.step(keys[4 + t], st)
    tot += int(inf.num_integration_steps.sum())
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"step: {dt/8*1e3:.2f} ms per transition, {tot/dt/1e6:.1f} M/s")
