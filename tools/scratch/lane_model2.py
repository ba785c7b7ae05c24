"""Lane model 2: latency lane kept full by ADDING the bulk chains with the largest estimated remaining work at every
refill point (never removed: a wrongly picked chain finishes fast there and frees its slot)."""
import sys, numpy as np
m = np.load("gpurun_out/nuts_lengths_T400.npz")["steps"].astype(np.int64)
P_SPEC = 4.93
def p_bulk(n, a): return max(a, 2.35e-3 * n)

def lanes(T, a=8.5, cap=128, refill_us=2000.0, slow=float(sys.argv[1]) if len(sys.argv) > 1 else 1.2, policy="recent", min_free=16, k0=32, p_f=None):
    steps = m[:T] + 1
    cum = np.cumsum(steps, 0)
    total = cum[-1].copy()
    N = steps.shape[1]
    done = np.zeros(N, dtype=np.int64)
    in_f = np.zeros(N, bool)
    live = np.ones(N, bool)
    now = 0.0; last_refill = -1e9; b_ticks = 0; adds = 0; epochs = 0
    while live.any():
        selB = live & ~in_f; selF = live & in_f
        nB = int(selB.sum()); nF = int(selF.sum())
        if nB > 0 and b_ticks >= k0 and cap - nF >= min(min_free, nB) and now - last_refill >= refill_us and (nB + nF > cap):
            cand = np.where(selB)[0]
            td = (cum[:, cand] <= done[cand]).sum(0)
            if policy == "oracle":
                est = (total - done)[cand].astype(float)
            elif policy == "lag":
                est = (T - td) * (done[cand] + 1.0) / (td + 0.5)
            else:  # recent: ticks spent in the current + previous transition
                tdc = np.minimum(td, T - 1)
                start_prev = np.where(td >= 2, cum[np.maximum(td - 2, 0), cand], 0)
                ntr = np.where(td >= 2, 2, td) + 0.5
                rate = (done[cand] - start_prev + 1.0) / ntr
                est = (T - td) * rate
            k = min(cap - nF, len(cand))
            pick = cand[np.argsort(-est)[:k]]
            in_f[pick] = True
            adds += k; epochs += 1; last_refill = now
            selB = live & ~in_f; selF = live & in_f
            nB = int(selB.sum()); nF = int(selF.sum())
            now += 60.0  # pause of the lane for the re-entry
        if nB > 0:
            perB = P_SPEC if (nB <= cap and nF == 0) else p_bulk(nB, a)
            dt = 16 * perB
            done[selB] += 16; b_ticks += 16
        else:
            dt = 64 * P_SPEC
        if nF > 0:
            perF = (p_f or P_SPEC) * (slow if nB > 2048 else 1.0)
            done[selF] += max(1, int(dt / perF))
        now += dt
        live &= done < total
    return now * 1e-6, epochs, adds

for T in (400, 100):
    tot = m[:T].sum()
    for pol in ("lag",):
        for cap in (128,):
            for refill in (2000.0, 10000.0):
                pf = P_SPEC if cap == 128 else 6.0
                t, e, a = lanes(T, cap=cap, refill_us=refill, policy=pol, p_f=pf)
                print(f"T={T} {pol:7s} cap={cap} refill={refill:7.0f}us: {t:.3f}s {tot/t/1e6:.0f} M/s  epochs {e} adds {a}")
