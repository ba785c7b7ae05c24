"""Per-chain leapfrog counts of the C3 run (T = 400): the input of the lane-scheduling model (NOTEBOOK section 19)."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
import blackjax_amd as bjx
dev = torch.device("cuda:0")
N, D = 32768, 256
alg = bjx.nuts(bjx.targets.NealFunnel(), 0.1, torch.ones(D, device=dev), max_num_doublings=10, use_graph="auto")
g = torch.Generator(device=dev); g.manual_seed(0)
state = alg.init(0.1 * torch.randn(N, D, device=dev, generator=g))
keys = bjx.random.split(bjx.random.key(0), 4)
for t in range(4):
    state, info = alg.step(keys[t], state)
alg.run(bjx.random.key(5), state, 2, store_positions=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
_, _, ri = alg.run(bjx.random.key(1), state, 400, store_positions=False)
torch.cuda.synchronize()
print("T=400 seconds", time.perf_counter() - t0)
m = ri.num_integration_steps.to(torch.int16).cpu().numpy()
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/nuts_lengths_T400.npz", steps=m)
tot = m.astype(np.int64).sum(0)
print("max", tot.max(), "mean", tot.mean(), "top", np.sort(tot)[::-1][[0, 1, 7, 31, 127, 511, 2047, 8191]])
