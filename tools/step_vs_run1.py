import sys, time, json, torch
sys.path.insert(0, '/root/repo')
import blackjax_amd as bjx
from blackjax_amd import _nuts as bnuts
dev = torch.device("cuda:0")
N, D = 32768, 256
fn = bjx.targets.NealFunnel()
mk = lambda **kw: bjx.nuts(fn, 0.1, torch.ones(D, device=dev), max_num_doublings=10, **kw)
algs = {"lockstep": mk(step_driver="lockstep"), "free128": mk(step_driver="free", step_spec_rows=128),
        "free512": mk(step_driver="free", step_spec_rows=512), "free2048": mk(step_driver="free", step_spec_rows=2048)}
g = torch.Generator(device=dev); g.manual_seed(0)
state = algs["lockstep"].init(0.1 * torch.randn(N, D, device=dev, generator=g))
keys = bjx.random.split(bjx.random.key(0), 40)
for t in range(4):
    state, info = algs["lockstep"].step(keys[t], state)
for name, alg in algs.items():
    st = state
    for t in range(4, 7):
        st, _ = alg.step(keys[t], st)
torch.cuda.synchronize()
res = {}
ref = None
for rep in range(2):
    for name, alg in algs.items():
        st = state; tot = 0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for t in range(8, 16):
            st, info = alg.step(keys[t], st); tot += int(info.num_integration_steps.sum())
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if ref is None:
            ref = st.position.clone()
        res.setdefault(name, []).append({"ms_per_transition": round(dt / 8 * 1e3, 3), "M_per_s": round(tot / dt / 1e6, 1),
                                         "same_final": bool(torch.equal(ref, st.position)),
                                         "spec": {k: v for k, v in bnuts._SPEC_STATS.items() if k in ("rows", "sequences", "seconds", "stalls")}})
print(json.dumps(res))
