#!/usr/bin/env python
"""Headline benchmark: chain-leapfrog-steps/sec of batched diagonal-mass HMC on MI355X.

Workload (BASELINE.json configs[1], SURVEY.md section 8d "C2"): 65 536 chains x 1 024-dim
diagonal Gaussian (sigma_i = 10^(-1+2i/(D-1))), inverse mass = sigma^2, eps = 0.25,
L = 50 leapfrog steps per transition, fp32.  A "step" is one HMC transition of every
chain (momentum draw, L leapfrogs each followed by the log-density callable, Metropolis
accept).  Chains shard over GPUs with no data-path collective (weak scaling: 65 536
chains PER GPU); per-chain keys come from the global chain index.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- dominant kernel (fused kick+drift leapfrog): ALGORITHMIC bytes per launch
                  (20 B x D x N: read p,g,q; write p,q) / mean launch duration measured with
                  HIP events on the launch stream inside the timed region; peak 8000 GB/s.
  cpu_baseline -- the oracle's C/OpenMP port of the same transition timed on the host cores on a
                  bounded sample (rank 0, N=1 only).  A reported baseline, not the target.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def sigma_ladder(D):
    return (10.0 ** (-1.0 + 2.0 * np.arange(D) / (D - 1))).astype(np.float32)


def cpu_baseline(D, L, eps, target_seconds=15.0):
    """Time the oracle's C port (all host cores) on a bounded sample of the same workload."""
    from oracle import cport, prng

    sig = sigma_ladder(D)
    imm = (sig * sig).astype(np.float32)
    inv_var = (np.float32(1.0) / imm).astype(np.float32)
    threads = cport.num_threads()
    n = 8 * threads
    rng = np.random.default_rng(0)

    def make(n):
        q = (sig * rng.standard_normal((n, D))).astype(np.float32)
        g = -(q * inv_var)
        logp = (0.5 * np.sum(q.astype(np.float64) * g, axis=-1)).astype(np.float32)
        return q, logp, g.astype(np.float32)

    n_big = 2048 * threads
    q, logp, g = make(n_big)
    cport.hmc_diag_gaussian_step(prng.key(0), q[:n], logp[:n], g[:n], eps, imm, inv_var, L)  # warm
    keys = prng.split(prng.key(1), 1000)
    done = 0
    t0 = time.perf_counter()
    while done < 2 or (time.perf_counter() - t0 < target_seconds and done < len(keys)):
        cport.hmc_diag_gaussian_step(keys[done], q, logp, g, eps, imm, inv_var, L)
        done += 1
    dt = time.perf_counter() - t0
    return {
        "value": n_big * L * done / dt,
        "unit": "chain-leapfrog-steps/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{n_big} chains x {D} dims, L={L}, {done} transitions "
                  f"({dt:.1f} s) -- C/OpenMP port of the oracle (CPU restatement of BlackJAX "
                  "arithmetic, NOT JAX: no jax wheel on this box)",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--chains", type=int, default=65536, help="chains PER GPU")
    ap.add_argument("--dim", type=int, default=1024)
    ap.add_argument("--leapfrogs", type=int, default=50)
    ap.add_argument("--eps", type=float, default=0.25)
    ap.add_argument("--chain-block", type=int, default=0,
                    help="run each transition block-by-block over this many chains (0 = all at once)")
    ap.add_argument("--use-graph", action="store_true",
                    help="capture each block's inner leapfrog/callable loop in a HIP graph")
    ap.add_argument("--no-ic-mode", action="store_true",
                    help="skip the extra Infinity-Cache-mode measurement (chain blocks + HIP graph)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-launch-timing", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert torch.cuda.is_available(), "bench.py needs a GPU (blackjax_amd has no CPU fallback)"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    import blackjax_amd as bjx
    from blackjax_amd import _lib

    N, D, L = args.chains, args.dim, args.leapfrogs
    sig = torch.as_tensor(sigma_ladder(D), device=dev)
    imm = (sig * sig).contiguous()
    target = bjx.targets.DiagGaussian((1.0 / imm).contiguous())
    alg = bjx.hmc(target, args.eps, imm, L, chain_offset=rank * N,
                  chain_block=args.chain_block or None, use_graph=args.use_graph)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    state = alg.init(sig * torch.randn(N, D, device=dev, generator=gen))
    keys = bjx.random.split(bjx.random.key(0), args.warmup + args.steps)

    def barrier():
        if world > 1:
            dist.barrier()

    # Warm-up runs EXACTLY the timed loop's body (including the bookkeeping torch ops and the
    # launch-timer events): on a fresh box the first use of any kernel pages its code object in from
    # disk (tens of ms per torch op), which must not land in the timed region.
    n_sub = min(1024, N)
    warm_timer = None if args.no_launch_timing else _lib.LaunchTimer(["bjx_leapfrog_diag"])
    _lib.set_timer(warm_timer)
    warm_acc = torch.zeros((), device=dev)
    prime_key = bjx.random.key(12345)
    for t in range(-1, args.warmup):  # one extra priming pass (t = -1) in addition to the W warm-ups
        state, info = alg.step(prime_key if t < 0 else keys[t], state)
        warm_acc += info.acceptance_rate.mean()
        _ = state.position[:n_sub].clone()
    _lib.set_timer(None)
    if warm_timer is not None:
        warm_timer.durations_ms("bjx_leapfrog_diag")  # first elapsed_time() call warms that path too
    torch.cuda.synchronize()

    timer = None
    if not args.no_launch_timing and rank == 0:
        timer = _lib.LaunchTimer(["bjx_leapfrog_diag"])
        _lib.set_timer(timer)
    acc_sum = torch.zeros((), device=dev)
    draws = []  # retained draws of a fixed chain subset for ESS/sec (4 MiB per step at C2)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(args.warmup, args.warmup + args.steps):
        state, info = alg.step(keys[t], state)
        acc_sum += info.acceptance_rate.mean()
        draws.append(state.position[:n_sub].clone())
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    _lib.set_timer(None)

    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        # final draws / statistics are the only thing that crosses xGMI (RCCL all-gather)
        sub = state.position[:256].contiguous()
        gathered = [torch.empty_like(sub) for _ in range(world)]
        dist.all_gather(gathered, sub)
        final_draws = torch.cat(gathered, 0)
    else:
        final_draws = state.position[:256]

    # Extra, separately reported region: the same workload run block-by-block (16 384 chains at a
    # time, inner loop captured in a HIP graph) so a block's q/p/g stay in the 256 MiB Infinity
    # Cache across the L steps.  Not the headline `value` (its kernels are not HBM streams).
    ic_mode = None
    if not args.no_ic_mode and not args.chain_block and not args.use_graph and N > 16384:
        alg_ic = bjx.hmc(target, args.eps, imm, L, chain_offset=rank * N, chain_block=16384,
                         use_graph=True)
        st_ic = state
        st_ic, _ = alg_ic.step(keys[0], st_ic)  # captures the graph
        barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for t in range(args.warmup, args.warmup + args.steps):
            st_ic, _ = alg_ic.step(keys[t], st_ic)
        torch.cuda.synchronize()
        barrier()
        dt_ic = time.perf_counter() - t1
        if world > 1:
            tt = torch.tensor([dt_ic], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt_ic = float(tt.item())
        ic_mode = {"value": world * N * L * args.steps / dt_ic, "unit": "chain-leapfrog-steps/s",
                   "chain_block": 16384, "hip_graph": True, "ms_per_step": dt_ic / args.steps * 1e3}

    # ESS/sec (second half of BASELINE.json's metric): min over dimensions of
    # effective_sample_size (blackjax/diagnostics.py:157-304) on the retained subset / wall time
    ess_min = None
    if args.steps >= 4:
        ess = bjx.diagnostics.effective_sample_size(torch.stack(draws, dim=1))  # (n_sub, T, D)
        ess_min = float(ess.min().item())

    if rank == 0:
        total_chain_leapfrogs = world * N * L * args.steps
        value = total_chain_leapfrogs / dt
        roofline = None
        if timer is not None:
            d_ms = timer.durations_ms("bjx_leapfrog_diag")
            avg_s = float(np.mean(d_ms)) * 1e-3
            alg_bytes = 20.0 * D * N  # read p,g,q ; write p,q (imm (D,) is shared and cached)
            achieved = alg_bytes / avg_s / 1e9
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
            if os.path.exists(tpath):
                try:
                    tj = json.load(open(tpath))
                    if tj.get("chains") == N and tj.get("dim") == D:
                        traffic = tj.get("hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            roofline = {
                "bound": "hbm", "kernel": "k_leapfrog_diag<4,2>", "achieved": achieved,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "algorithmic_bytes_per_launch": alg_bytes,
                "avg_launch_us": avg_s * 1e6, "launches_timed": len(d_ms),
            }
        out = {
            "metric": "chain-leapfrog-steps/sec (whole node), 65 536 chains x 1 024-dim diag-mass HMC",
            "value": value,
            "unit": "chain-leapfrog-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"HMC diag mass, {N} chains/GPU x {D}-dim Gaussian (sigma ladder 0.1..10), "
                            f"L={L}, eps={args.eps}, user log-density = HIP DiagGaussian callable",
                "chains_per_gpu": N, "dim": D, "leapfrogs": L, "global_chains": world * N,
                "chain_block": args.chain_block or N, "hip_graph": bool(args.use_graph),
                "parallelism": f"chains sharded x{world}, no data-path collective",
            },
            "mean_acceptance": float(acc_sum.item()) / args.steps,
            "ess": None if ess_min is None else {
                "min_ess_subset": ess_min, "subset_chains": n_sub, "draws_per_chain": args.steps,
                "min_ess_per_sec_subset": ess_min / dt,
                "min_ess_per_sec_all_chains": ess_min / dt * (world * N / n_sub),
                "note": "rank-0 subset of chains, min over the D dimensions"},
            "end_to_end_frac_of_28B_roofline": value / world / (HBM_PEAK_GBS * 1e9 / (28.0 * D)),
            "final_draws_gathered": list(final_draws.shape),
            "infinity_cache_mode": ic_mode,
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(D, L, args.eps)
            except Exception as e:  # the baseline is a reported extra; never fail the GPU number
                out["cpu_baseline"] = {"value": None, "error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
