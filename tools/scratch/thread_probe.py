"""Which combination aborts?  Scenarios run in subprocesses: argv[1] = scenario name."""
import os, sys, threading, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
SCEN = ["one_thread_capture", "two_threads_nograph", "two_threads_capture", "two_threads_capture_locked", "two_threads_hmc_only", "two_threads_nuts_only"]
if len(sys.argv) == 1:
    for s in SCEN:
        r = subprocess.run([sys.executable, "-X", "faulthandler", __file__, s], capture_output=True, text=True, timeout=600)
        print(s, "rc", r.returncode, (r.stdout.strip().splitlines() or [""])[-1], "|", (r.stderr.strip().splitlines() or [""])[-1][:200])
    sys.exit(0)
import torch
import blackjax_amd as bjx
scen = sys.argv[1]
dev = torch.device("cuda:0")
D = 32
imm = torch.ones(D, device=dev)
funnel = bjx.targets.NealFunnel()
gauss = bjx.targets.DiagGaussian(torch.linspace(0.5, 2.0, D, device=dev))
q_b = 0.3 * torch.randn(160, D, device=dev)
q_c = torch.randn(200, D, device=dev)
keys = list(bjx.random.split(bjx.random.key(3), 5))
ug = False if "nograph" in scen else "auto"
lock = threading.Lock()
def run(kind, stream):
    alg = bjx.nuts(funnel, 0.15, imm, max_num_doublings=5, use_graph=ug) if kind == "nuts" else bjx.hmc(gauss, 0.2, imm, 5, use_graph=ug)
    q = q_b if kind == "nuts" else q_c
    with torch.cuda.stream(stream):
        st = alg.init(q)
        for k in keys:
            if "locked" in scen:
                with lock:
                    st, _ = alg.step(k, st)
            else:
                st, _ = alg.step(k, st)
    stream.synchronize()
s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
if scen == "one_thread_capture":
    th = [threading.Thread(target=run, args=("hmc", s1))]
elif scen == "two_threads_hmc_only":
    th = [threading.Thread(target=run, args=("hmc", s1)), threading.Thread(target=run, args=("hmc", s2))]
elif scen == "two_threads_nuts_only":
    th = [threading.Thread(target=run, args=("nuts", s1)), threading.Thread(target=run, args=("nuts", s2))]
else:
    th = [threading.Thread(target=run, args=("hmc", s1)), threading.Thread(target=run, args=("nuts", s2))]
for t in th: t.start()
for t in th: t.join(300)
torch.cuda.synchronize()
print("ok", scen)
