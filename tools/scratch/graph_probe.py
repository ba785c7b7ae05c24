import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import blackjax_amd as bjx
dev = torch.device("cuda:0")
N, D = 512, 16
g = torch.Generator(device=dev); g.manual_seed(0)
q0 = torch.randn(N, D, device=dev, generator=g)
iv = torch.linspace(0.5, 2.0, D, device=dev)
auto_fn = lambda q: -0.5 * (q * q * iv).sum(-1)          # autograd callable
def bad_fn(q):                                            # synchronises: cannot be captured
    lp = -0.5 * (q * q * iv).sum(-1)
    _ = float(lp.sum().item())
    return lp
for name, fn in (("autograd", auto_fn), ("syncing", bad_fn)):
    a_e = bjx.nuts(fn, 0.3, torch.ones(D, device=dev), max_num_doublings=5)
    a_g = bjx.nuts(fn, 0.3, torch.ones(D, device=dev), max_num_doublings=5, use_graph=True)
    st = a_e.init(q0)
    key = bjx.random.key(1)
    s_e, i_e = a_e.step(key, st)
    try:
        s_g, i_g = a_g.step(key, st)
        print(name, "graph ok, equal:", torch.equal(s_e.position, s_g.position), torch.equal(i_e.num_integration_steps, i_g.num_integration_steps))
    except Exception as e:
        print(name, "graph failed:", type(e).__name__, str(e)[:200].replace("\n", " "))
        try:
            torch.cuda.synchronize()
            s_e2, _ = a_e.step(key, st)
            print("   device usable afterwards, eager equal:", torch.equal(s_e.position, s_e2.position))
        except Exception as e2:
            print("   device NOT usable:", type(e2).__name__, str(e2)[:200].replace("\n", " "))
