import sys, time, json, torch
sys.path.insert(0, '/root/repo')
import blackjax_amd as bjx
dev = torch.device("cuda:0")
N, D = 32768, 256
alg = bjx.nuts(bjx.targets.NealFunnel(), 0.1, torch.ones(D, device=dev), max_num_doublings=10, use_graph=True)
g = torch.Generator(device=dev); g.manual_seed(0)
state = alg.init(0.1 * torch.randn(N, D, device=dev, generator=g))
keys = bjx.random.split(bjx.random.key(0), 40)
for t in range(4):
    state, info = alg.step(keys[t], state)
from blackjax_amd.nuts import run_free
def one_run(k, st):
    st2, _, ri = run_free(k, st, bjx.targets.NealFunnel(), 0.1, torch.ones(D, device=dev), 1, 10, key_layout="step", store_positions=False)
    return st2, ri
st = state
for t in range(4, 8):
    st, ri = one_run(keys[t], st)
torch.cuda.synchronize()
res = {}
for name in ("step", "run1", "step", "run1"):
    st = state; tot = 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(8, 16):
        if name == "step":
            st, info = alg.step(keys[t], st); tot += int(info.num_integration_steps.sum())
        else:
            st, ri = one_run(keys[t], st); tot += int(ri.num_integration_steps.sum())
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    res.setdefault(name, []).append({"ms_per_transition": dt / 8 * 1e3, "M_per_s": tot / dt / 1e6})
print(json.dumps(res))
