import sys, torch, os
sys.path.insert(0, '/root/repo')
import blackjax_amd as bjx
from blackjax_amd import _nuts as bnuts
dev = torch.device("cuda:0")
N, D, max_depth, eps = 300, 64, 8, 0.1
g = torch.Generator(device=dev); g.manual_seed(N + D)
q0 = 0.2 * torch.randn(N, D, device=dev, generator=g)
fn = bjx.targets.NealFunnel(); imm = torch.ones(D, device=dev)
lock = bjx.nuts(fn, eps, imm, max_num_doublings=max_depth, step_driver="lockstep")
keys = bjx.random.split(bjx.random.key(3), 6)
states = [lock.init(q0)]
for k in keys:
    states.append(lock.step(k, states[-1])[0])
tot = 0
for rep in range(3):
    h = {}
    for i, k in enumerate(keys):
        if i == 0:
            st, _, ri = bnuts.run_free(k, states[i], fn, eps, imm, 1, max_depth, key_layout="step", store_positions=False, keep_ends=True, spec_rows=0, _handle=h)
        else:
            st, _, ri = h["rerun"](k, states[i], eps, imm)
        torch.cuda.synchronize()
        bad = int((states[i + 1].position != st.position).any(1).sum())
        tot += bad
        print("DBG", os.environ.get("BJX_DBG_PERSIST"), "rep", rep, "call", i, "bad", bad, flush=True)
print("DBG", os.environ.get("BJX_DBG_PERSIST"), "TOTAL", tot)
