import sys, os, torch
sys.path.insert(0, os.getcwd())
import blackjax_amd as bjx
dev = torch.device("cuda:0")
N, D = 32768, 256
alg = bjx.nuts(bjx.targets.NealFunnel(), 0.1, torch.ones(D, device=dev), max_num_doublings=10, use_graph=True)
g = torch.Generator(device=dev); g.manual_seed(0)
state = alg.init(0.1 * torch.randn(N, D, device=dev, generator=g))
depths = []
for k in bjx.random.split(bjx.random.key(0), 14):
    state, info = alg.step(k, state)
    depths.append(info.num_trajectory_expansions.clone())
d = torch.stack(depths).cpu()
for t in range(14):
    print(t, "max", int(d[t].max()), "n>=9", int((d[t] >= 9).sum()), "n>=10", int((d[t] >= 10).sum()), "mean leaves", float(((2.0 ** d[t]) - 1).mean()))
deep = d >= 9
p = (deep[1:] & deep[:-1]).sum().item() / max(deep[:-1].sum().item(), 1)
print("P(deep at t+1 | deep at t) =", p, " base rate", deep.float().mean().item())
