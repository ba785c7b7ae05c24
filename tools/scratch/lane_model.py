"""Makespan model of free-running NUTS lanes on the measured per-chain leapfrog counts (tools/scratch/nuts_lengths.py).
Single lane (as built) vs bulk lane + one latency lane refilled in epochs with the chains that lag most."""
import sys, numpy as np
m = np.load("gpurun_out/nuts_lengths_T400.npz")["steps"].astype(np.int64)

def p_bulk(n, a):            # us per tick of a lane of n live rows on the graph / plain tick path
    return max(a, 2.35e-3 * n)
P_SPEC = 4.93

def single(T, a, spec_rows=128):
    L = np.sort((m[:T] + 1).sum(0))[::-1]          # ticks each chain needs
    # n(k) = chains with L >= k; integrate p(n(k)) dk
    ks = np.concatenate([[0], L[::-1]])             # ascending lengths
    t = 0.0
    N = len(L)
    asc = L[::-1]
    prev = 0
    for j, l in enumerate(asc):                     # between prev and l ticks, N - j chains live
        n = N - j
        per = P_SPEC if n <= spec_rows else p_bulk(n, a)
        t += (l - prev) * per
        prev = l
    return t * 1e-6

def lanes(T, a, cap=128, k0=64, slow=1.0, policy="lag", verbose=False):
    """event simulation: bulk lane B, latency lane F (cap rows, P_SPEC*slow us per tick while B is busy)."""
    steps = m[:T] + 1                                # ticks per (transition, chain)
    cum = np.cumsum(steps, 0)                        # ticks needed to have finished transition t
    total = cum[-1].copy()
    N = steps.shape[1]
    done_ticks = np.zeros(N, dtype=np.int64)         # ticks each chain has received
    in_f = np.zeros(N, bool)
    tB = tF = 0.0                                    # lane clocks
    now = 0.0
    live = np.ones(N, bool)
    f_active = False
    epochs = 0
    b_ticks = 0
    while live.any():
        nB = int((live & ~in_f).sum()); nF = int((live & in_f).sum())
        if nF == 0:
            f_active = False
            in_f[:] = False
        # (re)fill the latency lane at a bulk sync point
        if not f_active and nB > cap and b_ticks >= k0:
            cand = np.where(live)[0]
            tdone = (cum[:, cand] <= done_ticks[cand]).sum(0)     # transitions finished
            if policy == "lag":
                est = (T - tdone) * (done_ticks[cand] + 1.0) / (tdone + 0.5)
            else:
                est = (total - done_ticks)[cand]                  # oracle
            pick = cand[np.argsort(-est)[:cap]]
            in_f[pick] = True
            f_active = True
            epochs += 1
            nB -= len(pick); nF = len(pick)
        if nB == 0 and not f_active:
            break
        # advance both lanes by one "chunk": bulk 16 ticks (or until a chain event), F in proportion
        if nB > 0:
            perB = P_SPEC if (nB <= cap and not f_active) else p_bulk(nB, a)
            dt = 16 * perB
            selB = live & ~in_f
            done_ticks[selB] += 16
            b_ticks += 16
        else:
            dt = 64 * P_SPEC
        if f_active:
            perF = P_SPEC * (slow if nB > 2048 else 1.0)
            selF = live & in_f
            done_ticks[selF] += int(dt / perF)
        now += dt
        live &= done_ticks < total
    return now * 1e-6, epochs

for T, meas in ((400, 2.176), (100, None)):
    tot = m[:T].sum()
    for a in (7.0, 8.0, 9.0, 10.0):
        s = single(T, a)
        print(f"T={T} a={a}: single {s:.3f}s ({tot/s/1e6:.0f} M/s)", end="")
        for cap in (128,):
            for k0 in (64, 256):
                for slow in (1.0, 1.5):
                    for pol in ("lag", "oracle"):
                        t, e = lanes(T, a, cap, k0, slow, pol)
                        print(f" | cap{cap} k0={k0} slow{slow} {pol}: {t:.3f}s {tot/t/1e6:.0f}M/s e{e}", end="")
        print()
