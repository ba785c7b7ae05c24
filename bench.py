#!/usr/bin/env python
"""Headline benchmark: chain-leapfrog-steps/sec of batched diagonal-mass HMC on MI355X.

Workload (BASELINE.json configs[1], SURVEY.md section 8d "C2"): 65 536 chains x 1 024-dim
diagonal Gaussian (sigma_i = 10^(-1+2i/(D-1))), inverse mass = sigma^2, eps = 0.25,
L = 50 leapfrog steps per transition, fp32.  A "step" is one HMC transition of every
chain (momentum draw, L leapfrogs each followed by the log-density callable, Metropolis
accept).  Chains shard over GPUs with no data-path collective (weak scaling: 65 536
chains PER GPU); per-chain keys come from the global chain index.

Scheduling of a transition (``--chain-block``): chains are independent, so the engine may run a
transition block by block over chains -- same kernels, same results bit for bit.  A block whose
q, p, g fit the 256 MiB Infinity Cache (16 384 chains at D = 1 024) re-reads its state from the cache
across the L steps; whether that beats one launch for all chains (pure HBM streaming) depends on the
box (+13 % ... -3 % measured across the pool).  The default (-1) therefore AUTOTUNES during warm-up:
two untimed transitions in each mode, the faster one is benchmarked and named in
``config.chain_block``; the other mode is timed afterwards and reported as ``alternate_mode``.
``--chain-block 0`` / ``n`` force one launch for all chains / blocks of n chains.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- dominant kernel (fused kick+drift leapfrog): ALGORITHMIC bytes per launch
                  (20 B x D x chains per launch: read p,g,q; write p,q) / mean launch duration
                  measured with HIP events on the launch stream inside the timed region (every 16th
                  launch is bracketed when a transition has >= 100 launches, events from a pool
                  recorded before the region; the cost of an empty bracket is reported beside it);
                  peak 8000 GB/s.
Also reported: gpu_ms_of_each_step (HIP events around every timed transition) and
host_enqueue_ms_per_step -- a run in which the host fell behind shows as long steps at unchanged
kernel durations.
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
    # 40 timed transitions = 0.6 s: a single host/driver hiccup (one 28 ms step among 14.5 ms ones was
    # seen in 1 of 8 back-to-back runs) moves a 10-step region by 8 %, a 40-step region by 2 %
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--chains", type=int, default=65536, help="chains PER GPU")
    ap.add_argument("--dim", type=int, default=1024)
    ap.add_argument("--leapfrogs", type=int, default=50)
    ap.add_argument("--eps", type=float, default=0.25)
    ap.add_argument("--chain-block", type=int, default=-1,
                    help="chains per launch: -1 = auto (block sized for the Infinity Cache), "
                         "0 = all chains at once, n = n chains")
    ap.add_argument("--use-graph", action="store_true",
                    help="capture each block's inner leapfrog/callable loop in a HIP graph")
    ap.add_argument("--no-plain-mode", action="store_true",
                    help="skip the extra all-chains-at-once (HBM streaming) measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-launch-timing", action="store_true")
    ap.add_argument("--time-every", type=int, default=0,
                    help="bracket every k-th leapfrog launch with HIP events (0 = 16 for short launches, else 1)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # One rank per GPU over RCCL (backend "nccl").  BJX_BENCH_BACKEND=gloo exists only so the
        # multi-rank control flow can be exercised on a single-GPU box (ranks then share cuda:0).
        backend = os.environ.get("BJX_BENCH_BACKEND", "nccl")
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    assert torch.cuda.is_available(), "bench.py needs a GPU (blackjax_amd has no CPU fallback)"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    import blackjax_amd as bjx
    from blackjax_amd import _lib
    from blackjax_amd.hmc import auto_chain_block

    N, D, L = args.chains, args.dim, args.leapfrogs
    blk = auto_chain_block(N, D) if args.chain_block < 0 else (args.chain_block or N)
    blk = min(blk, N)
    n_blocks = (N + blk - 1) // blk
    sig = torch.as_tensor(sigma_ladder(D), device=dev)
    imm = (sig * sig).contiguous()
    target = bjx.targets.DiagGaussian((1.0 / imm).contiguous())
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    q_init = sig * torch.randn(N, D, device=dev, generator=gen)
    keys = bjx.random.split(bjx.random.key(0), args.warmup + args.steps)
    n_sub = min(1024, N)
    timed_kernel = "bjx_leapfrog_diag"

    def barrier():
        if world > 1:
            dist.barrier()

    host_enqueue_ms = []  # per measure() call: host time to queue one transition
    per_step_ms = []      # per measure() call: GPU time of every timed transition (HIP events)

    def measure(chain_block, use_graph, collect_draws):
        """W warm-up + K timed transitions in one scheduling mode.  The warm-up runs EXACTLY the
        timed loop's body (bookkeeping torch ops and launch-timer events included) plus one priming
        pass: on a fresh box the first use of any kernel pages its code object in from disk, which
        must not land in the timed region."""
        alg = bjx.hmc(target, args.eps, imm, L, chain_offset=rank * N, chain_block=chain_block,
                      use_graph=use_graph)
        state = alg.init(q_init)
        launches = L * ((N + chain_block - 1) // chain_block)
        # Sampling rate of the HIP-event brackets.  A bracket costs host time and drains the
        # queue around the launch; at ~50 us launches, bracketing every 4th one slowed the whole
        # timed region by 13 % (16.4 vs 14.5 ms per transition), so short launches are sampled sparsely.
        every = args.time_every or (16 if launches >= 100 else 1)
        timing = not args.no_launch_timing and not use_graph  # events cannot be recorded inside a graph
        cap = (launches // every + 1) * (max(args.steps, args.warmup + 1))
        warm_timer = _lib.LaunchTimer([timed_kernel], every, cap) if timing else None
        _lib.set_timer(warm_timer)
        warm_acc = torch.zeros((), device=dev)
        prime_key = bjx.random.key(12345)
        for t in range(-1, args.warmup):
            state, info = alg.step(prime_key if t < 0 else keys[t], state)
            warm_acc += info.acceptance_rate.mean()
            _ = state.position[:n_sub].clone()
        _lib.set_timer(None)
        if warm_timer is not None:
            warm_timer.durations_ms(timed_kernel)  # first elapsed_time() call warms that path too
        torch.cuda.synchronize()

        timer = None
        if timing and rank == 0:
            timer = _lib.LaunchTimer([timed_kernel], every, cap)
            torch.cuda.synchronize()
            _lib.set_timer(timer)
        acc_sum = torch.zeros((), device=dev)
        draws = []  # retained draws of a fixed chain subset for ESS/sec (4 MiB per step at C2)
        step_marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        for ev in step_marks:
            ev.record()  # HIP events are created at the first record(): do that outside the region
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step_marks[0].record()
        for t in range(args.warmup, args.warmup + args.steps):
            state, info = alg.step(keys[t], state)
            acc_sum += info.acceptance_rate.mean()
            if collect_draws:
                draws.append(state.position[:n_sub].clone())
            step_marks[t - args.warmup + 1].record()
        t_enq = time.perf_counter() - t0  # the host has queued everything; the GPU may still be working
        torch.cuda.synchronize()
        barrier()
        dt = time.perf_counter() - t0
        host_enqueue_ms.append(t_enq / max(args.steps, 1) * 1e3)
        per_step_ms.append([round(a.elapsed_time(b), 3) for a, b in zip(step_marks, step_marks[1:])])
        _lib.set_timer(None)
        if world > 1:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        roof = None
        if timer is not None:
            d_ms = timer.durations_ms(timed_kernel)
            avg_s = float(np.mean(d_ms)) * 1e-3
            alg_bytes = 20.0 * D * min(chain_block, N)  # read p,g,q ; write p,q (imm (D,) is shared and cached)
            achieved = alg_bytes / avg_s / 1e9
            # bjx_leapfrog_diag dispatches rows of a multiple of 1 024 floats to the flat kernel
            flat = D % 1024 == 0 and os.environ.get("BJX_LF_FLAT", "1") != "0"
            roof = {"bound": "hbm", "kernel": "k_leapfrog_diag_flat<2>" if flat else "k_leapfrog_diag<4,2>",
                    "achieved": achieved,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "traffic": None, "algorithmic_bytes_per_launch": alg_bytes,
                    "chains_per_launch": min(chain_block, N), "avg_launch_us": avg_s * 1e6,
                    "launches_timed": len(d_ms), "timed_every": every}
            # What the bracket itself costs: a pair of event records around NOTHING, behind a kernel
            # of the same kind so the queue is in the same state.  Reported beside the raw figure
            # (`achieved` / `frac` stay on the raw one); rocprofv3's kernel-trace average for this
            # kernel (profiles/) should sit near avg_launch_us - event_bracket_overhead_us.
            try:
                empties = []
                probe = torch.zeros(min(chain_block, N), D, device=dev)
                for _ in range(32):
                    probe.add_(1.0)
                    s_ev, e_ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s_ev.record()
                    e_ev.record()
                    empties.append((s_ev, e_ev))
                torch.cuda.synchronize()
                over_us = float(np.median([a.elapsed_time(b) for a, b in empties[8:]])) * 1e3
                roof["event_bracket_overhead_us"] = over_us
                roof["avg_launch_us_net_of_bracket"] = avg_s * 1e6 - over_us
            except Exception:
                pass
        return state, dt, roof, float(acc_sum.item()) / max(args.steps, 1), draws

    # Scheduling autotune (untimed, part of the warm-up): whether Infinity-Cache blocking beats one
    # launch for all chains depends on the box (+13 % ... -3 % across the pool), so with the default
    # --chain-block -1 both are tried for two transitions each and the faster one is benchmarked.
    tuning = None
    if args.chain_block < 0 and n_blocks > 1:
        tuning = {}
        for cb in (blk, N):
            alg_t = bjx.hmc(target, args.eps, imm, L, chain_offset=rank * N, chain_block=cb)
            st_t = alg_t.init(q_init)
            st_t, _ = alg_t.step(bjx.random.key(777), st_t)
            torch.cuda.synchronize()
            t_t = time.perf_counter()
            for kk in bjx.random.split(bjx.random.key(778), 2):
                st_t, _ = alg_t.step(kk, st_t)
            torch.cuda.synchronize()
            tuning[cb] = (time.perf_counter() - t_t) / 2 * 1e3
        del alg_t, st_t
        if tuning[N] < tuning[blk]:
            blk, n_blocks = N, 1
    other_blk = None
    if args.chain_block < 0 and tuning is not None:
        other_blk = N if blk != N else min(auto_chain_block(N, D), N)

    state, dt, roofline, mean_acc, draws = measure(blk, args.use_graph, True)
    if roofline is not None:
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if (tj.get("chains") == N and tj.get("dim") == D
                        and tj.get("chains_per_launch", N) == roofline["chains_per_launch"]):
                    roofline["traffic"] = tj.get("hbm_bytes_per_launch")
            except Exception:
                pass
        if n_blocks > 1:
            roofline["note"] = ("chain-block scheduling: a block's q/p/g stay resident in the 256 MiB Infinity "
                                "Cache across the L steps, so part of this kernel's traffic never reaches "
                                "HBM; alternate_mode.roofline is the same kernel streaming from HBM")

    if world > 1:
        # final draws / statistics are the only thing that crosses xGMI (RCCL all-gather)
        sub = state.position[:256].contiguous()
        gathered = [torch.empty_like(sub) for _ in range(world)]
        dist.all_gather(gathered, sub)
        final_draws = torch.cat(gathered, 0)
    else:
        final_draws = state.position[:256]

    # Extra, separately reported region: the same workload with all chains in one launch (every
    # leapfrog launch streams its 1.3 GB from HBM).
    plain_mode = None
    if other_blk is None and n_blocks > 1:
        other_blk = N  # an explicit --chain-block: still show the all-at-once mode beside it
    if not args.no_plain_mode and other_blk is not None and other_blk != blk:
        _, dt_p, roof_p, _, _ = measure(other_blk, False, False)
        plain_mode = {"chain_block": other_blk, "value": world * N * L * args.steps / dt_p,
                      "unit": "chain-leapfrog-steps/s", "ms_per_step": dt_p / args.steps * 1e3,
                      "roofline": roof_p}

    # ESS/sec (second half of BASELINE.json's metric): min over dimensions of
    # effective_sample_size (blackjax/diagnostics.py:157-304) on the retained subset / wall time
    ess_min = None
    if args.steps >= 4:
        ess = bjx.diagnostics.effective_sample_size(torch.stack(draws, dim=1))  # (n_sub, T, D)
        ess_min = float(ess.min().item())

    if rank == 0:
        value = world * N * L * args.steps / dt
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
                "chain_block": blk, "hip_graph": bool(args.use_graph),
                "parallelism": f"chains sharded x{world}, no data-path collective",
            },
            "gpu_ms_of_each_step": per_step_ms[0],
            "host_enqueue_ms_per_step": host_enqueue_ms[0],  # close to ms_per_step = the host's launch rate is the limit
            "mean_acceptance": mean_acc,
            "ess": None if ess_min is None else {
                "min_ess_subset": ess_min, "subset_chains": n_sub, "draws_per_chain": args.steps,
                "min_ess_per_sec_subset": ess_min / dt,
                "min_ess_per_sec_all_chains": ess_min / dt * (world * N / n_sub),
                "note": "rank-0 subset of chains, min over the D dimensions"},
            "end_to_end_frac_of_28B_roofline": value / world / (HBM_PEAK_GBS * 1e9 / (28.0 * D)),
            "final_draws_gathered": list(final_draws.shape),
            "scheduling_autotune_ms_per_step": (None if tuning is None else
                                                {str(k): v for k, v in tuning.items()}),
            "alternate_mode": plain_mode,
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
