// NUTS two-stream speculative tail (gfx950; include/bjx_nuts.h "Speculative tail"): stream A's light integrator,
// stream B's bookkeeper replaying the UNCHANGED tick arithmetic (bjx_nuts_tick_dev.h: async_leaf3_row /
// async_end2_chain) over a ring, the stream-concurrency probe, and their C-ABI entry points.
#include "bjx_nuts_tick_dev.h"
#include "bjx_nuts_host.h"

namespace {

// ------------------------------------------------------------------------------------ free-running chains: speculative tail
// (round 5; include/bjx_nuts.h "Speculative tail")  Stream A: k_nuts_spec_integrate, the serial part of a leaf
// (closing kick, next position along the key's direction schedule) + a ring push of the callable's outputs.
// Stream B: k_nuts_spec_book, the UNCHANGED tick arithmetic (async_leaf3_row / async_end2_chain) replayed over the
// ring on the bookkeeper's own replica of the pending position -- every decision and record is the one-stream tick's.
enum { SW_EP = 0, SW_DEPTH, SW_S, SW_DIRS, SW_EPS, SW_CNT, SW_STATE };  // iw: stream A's words of a row
enum { BWD_EP = 0, BWD_CNT, BWD_IK0, BWD_IK1, BWD_EPS };                // bw: stream B -> A
enum { TG_EP = 0, TG_DEPTH, TG_S, TG_LP, TG_X0 };                       // ring_tag: identity of a pushed leaf
enum { SPD_MISMATCH = 0, SPD_STALL, SPD_RESTART, SPD_STALE, SPD_TIMEOUT, SPD_ORDER };
enum { SPS_RUN = 0, SPS_WAIT = 1, SPS_DONE = 2 };  // integrator state: integrating / tree exhausted or end pending / chain finished
static_assert(SW_STATE < BJX_NUTS_SPEC_IW && BWD_EPS < BJX_NUTS_SPEC_IW && TG_X0 + 4 <= BJX_NUTS_SPEC_TAG, "spec layout");

__device__ __forceinline__ int ld_agent(const int* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(int* p, int v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ long long ld64_agent(const long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st64_agent(long long* p, long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int64_t spec_n_rows(const bjx_nuts_spec_t& sp) {
  if (!sp.n_rows_dev) return sp.n_rows;
  const int64_t n = (int64_t)__builtin_amdgcn_readfirstlane(*sp.n_rows_dev);
  return n < sp.n_rows ? n : sp.n_rows;
}
// bit d = 1: doubling d goes forward (begin_doubling_rec's draw for every depth at once, one lane per depth)
__device__ __forceinline__ int spec_dirs(Key ik, int max_depth) {
  const int lane = threadIdx.x & 63;
  const bool in = lane < max_depth;
  const Key kd = key_child(key_child(ik, (uint64_t)(in ? lane : 0)), 0);
  const bool fwd = key_uniform(kd) < 0.5f;
  return (int)(uint32_t)__ballot(in && fwd);
}

template <int NI>
__device__ __forceinline__ void spec_integrate_row(const bjx_nuts_t& nt, const bjx_nuts_async_t& ax,
                                                   const bjx_nuts_spec_t& sp, const float* __restrict__ logp_f,
                                                   const float* __restrict__ gf, int64_t b) {
  constexpr int VEC = 4;
  const int g = threadIdx.x;
  const int64_t c = (int64_t)__builtin_amdgcn_readfirstlane(sp.rows[b]);
  int* iwp = sp.iw + b * BJX_NUTS_SPEC_IW;
  int* bwp = sp.bw + b * BJX_NUTS_SPEC_IW;
  // one round trip: the row's words, the bookkeeper's epoch / consumed count, the callable's outputs, the rows
  const int4 i0 = *reinterpret_cast<const int4*>(iwp);
  const int4 i1 = *reinterpret_cast<const int4*>(iwp + 4);
  // (agent-scope loads; plain loads -- a launch starts with an acquire, so they would be at most one launch stale --
  // measured 2.5 % slower over the tail of the C3 T = 400 run)
  int ep_b = ld_agent(bwp + BWD_EP);
  int cnt_b = ld_agent(bwp + BWD_CNT);
  float lp = logp_f[b];
  const int64_t base = c * nt.D, rbase = b * nt.D;
  const float* im = nt.imm + c * nt.imm_stride;
  uint32_t j0[NI];
  bool ok[NI];
  Row<VEC> G[NI], X[NI], P[NI], M[NI];
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    j0[k] = ((uint32_t)g + 64u * k) * VEC;
    ok[k] = j0[k] < (uint32_t)nt.D;
    if (ok[k]) {
      G[k] = ldr<VEC>(gf + rbase + j0[k]);
      X[k] = ldr<VEC>(sp.qf + rbase + j0[k]);
      P[k] = ldr<VEC>(sp.fp + rbase + j0[k]);
      M[k] = ldr<VEC>(im + j0[k]);
    }
  }
  const int ep = __builtin_amdgcn_readfirstlane(i0.x);
  int depth = __builtin_amdgcn_readfirstlane(i0.y);
  int s = __builtin_amdgcn_readfirstlane(i0.z);
  const int dirs = __builtin_amdgcn_readfirstlane(i0.w);
  const float eps = __int_as_float(__builtin_amdgcn_readfirstlane(i1.x));
  int cnt = __builtin_amdgcn_readfirstlane(i1.y);
  int state = __builtin_amdgcn_readfirstlane(i1.z);
  ep_b = __builtin_amdgcn_readfirstlane(ep_b);
  cnt_b = __builtin_amdgcn_readfirstlane(cnt_b);
  lp = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(lp)));
  if (state == SPS_DONE) return;
  // every record pushed by EARLIER launches is complete in memory (those launches have ended): announce them
  if (g == 0) st_agent(sp.avail + b, cnt);
  if (ep_b == -1) {  // the chain has completed its last transition
    if (g == 0) iwp[SW_STATE] = SPS_DONE;
    return;
  }
  if (ep_b != ep) {
    // ---- the bookkeeper has started a new transition: restart from its first pending position
    __threadfence();  // acquire: what stream B wrote before it published the epoch
    const Key ik{(uint32_t)ld_agent(bwp + BWD_IK0), (uint32_t)ld_agent(bwp + BWD_IK1)};
    const int eps_bits = ld_agent(bwp + BWD_EPS);
    const int dirs2 = spec_dirs(ik, nt.max_depth);
#pragma unroll
    for (int k = 0; k < NI; ++k)
      if (ok[k]) {
        const Row<VEC> q0 = ldr<VEC>(ax.q + base + j0[k]);
        const Row<VEC> p0 = ldr<VEC>(ax.p + base + j0[k]);
        const Row<VEC> g0 = ldr<VEC>(ax.g + base + j0[k]);
        const Row<VEC> x1 = ldr<VEC>(sp.qf_book + rbase + j0[k]);
        const Row<VEC> p1 = ldr<VEC>(ax.front_p + base + j0[k]);
        str<VEC>(sp.eLq + rbase + j0[k], q0); str<VEC>(sp.eLp + rbase + j0[k], p0); str<VEC>(sp.eLg + rbase + j0[k], g0);
        str<VEC>(sp.eRq + rbase + j0[k], q0); str<VEC>(sp.eRp + rbase + j0[k], p0); str<VEC>(sp.eRg + rbase + j0[k], g0);
        str<VEC>(sp.qf + rbase + j0[k], x1);
        str<VEC>(sp.fp + rbase + j0[k], p1);
      }
    if (g == 0) {
      *reinterpret_cast<int4*>(iwp) = make_int4(ep_b, 0, 0, dirs2);
      *reinterpret_cast<int4*>(iwp + 4) = make_int4(eps_bits, cnt, SPS_RUN, 0);
      // acknowledge: the first record of the new epoch will be number cnt (everything before it is stale)
      st64_agent(reinterpret_cast<long long*>(sp.ack) + b, ((long long)ep_b << 32) | (long long)(uint32_t)cnt);
      atomicAdd(sp.dbg + SPD_RESTART, 1);
    }
    return;
  }
  if (state != SPS_RUN) return;  // tree exhausted (max_depth doublings speculated) or a transition end pending
  // Do not run further ahead of the bookkeeper than `lead` records (at most a ring): everything pushed past the
  // end of a transition is wasted, and the bookkeeper -- a full leaf of dependent arithmetic per record -- is not
  // much faster than this stream.  A waiting row pushes nothing; the callable re-evaluates the same qf.
  const int lead = sp.lead > 0 && sp.lead < sp.ring - 1 ? sp.lead : sp.ring - 1;
  if (cnt - cnt_b >= lead) {
    if (g == 0) atomicAdd(sp.dbg + SPD_STALL, 1);
    return;
  }
  const int dir = ((dirs >> depth) & 1) ? 1 : -1;
  const float deps = (float)dir * eps;
  const float h = deps * int_kick(nt);
  const float dd = deps * int_drift(nt);
  // ---- push (logp, gradient) of the leaf whose position the callable just evaluated
  const int slot = cnt & (sp.ring - 1);
  float* rg = sp.ring_g + ((int64_t)b * sp.ring + slot) * nt.D;
#pragma unroll
  for (int k = 0; k < NI; ++k)
    if (ok[k]) str<VEC>(rg + j0[k], G[k]);
  if (g == 0) {
    int* tp = sp.ring_tag + ((int64_t)b * sp.ring + slot) * BJX_NUTS_SPEC_TAG;
    *reinterpret_cast<int4*>(tp) = make_int4(ep, depth, s, __float_as_int(lp));
    *reinterpret_cast<int4*>(tp + 4) = make_int4(__float_as_int(X[0].v[0]), __float_as_int(X[0].v[1]),
                                                 __float_as_int(X[0].v[2]), __float_as_int(X[0].v[3]));
  }
  cnt += 1;
  // ---- closing kick of this leaf, then the opening half of the next one (async_leaf3_row's expressions)
#pragma unroll
  for (int k = 0; k < NI; ++k)
    if (ok[k]) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) P[k].v[e] = fmaf(h, G[k].v[e], P[k].v[e]);
    }
  const bool last = (s + 1) >= (1 << depth);
  if (!last) {
    s += 1;
#pragma unroll
    for (int k = 0; k < NI; ++k)
      if (ok[k]) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          P[k].v[e] = fmaf(h, G[k].v[e], P[k].v[e]);
          X[k].v[e] = fmaf(dd, M[k].v[e] * P[k].v[e], X[k].v[e]);
        }
        str<VEC>(sp.fp + rbase + j0[k], P[k]);
        str<VEC>(sp.qf + rbase + j0[k], X[k]);
      }
  } else if (depth + 1 >= nt.max_depth) {
    state = SPS_WAIT;  // nothing left to speculate: this transition ends here at the latest
  } else {
    const int dir2 = ((dirs >> (depth + 1)) & 1) ? 1 : -1;
    const float deps2 = (float)dir2 * eps;
    const float h2 = deps2 * int_kick(nt);
    const float dd2 = deps2 * int_drift(nt);
    if (dir2 == dir) {  // the end just reached keeps moving
#pragma unroll
      for (int k = 0; k < NI; ++k)
        if (ok[k]) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            P[k].v[e] = fmaf(h2, G[k].v[e], P[k].v[e]);
            X[k].v[e] = fmaf(dd2, M[k].v[e] * P[k].v[e], X[k].v[e]);
          }
          str<VEC>(sp.fp + rbase + j0[k], P[k]);
          str<VEC>(sp.qf + rbase + j0[k], X[k]);
        }
    } else {  // park this end, continue from the other one
      float* eq = (dir > 0 ? sp.eRq : sp.eLq) + rbase;
      float* eg = (dir > 0 ? sp.eRg : sp.eLg) + rbase;
      float* epp = (dir > 0 ? sp.eRp : sp.eLp) + rbase;
      const float* oq = (dir2 > 0 ? sp.eRq : sp.eLq) + rbase;
      const float* og = (dir2 > 0 ? sp.eRg : sp.eLg) + rbase;
      const float* op = (dir2 > 0 ? sp.eRp : sp.eLp) + rbase;
#pragma unroll
      for (int k = 0; k < NI; ++k)
        if (ok[k]) {
          Row<VEC> q2 = ldr<VEC>(oq + j0[k]);
          const Row<VEC> g2 = ldr<VEC>(og + j0[k]);
          Row<VEC> p2 = ldr<VEC>(op + j0[k]);
          str<VEC>(eq + j0[k], X[k]);
          str<VEC>(eg + j0[k], G[k]);
          str<VEC>(epp + j0[k], P[k]);
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            p2.v[e] = fmaf(h2, g2.v[e], p2.v[e]);
            q2.v[e] = fmaf(dd2, M[k].v[e] * p2.v[e], q2.v[e]);
          }
          str<VEC>(sp.fp + rbase + j0[k], p2);
          str<VEC>(sp.qf + rbase + j0[k], q2);
        }
    }
    depth += 1;
    s = 0;
  }
  if (g == 0) {
    *reinterpret_cast<int4*>(iwp) = make_int4(ep, depth, s, dirs);
    *reinterpret_cast<int4*>(iwp + 4) = make_int4(__float_as_int(eps), cnt, state, 0);
  }
}

template <int NI>
__global__ void __launch_bounds__(64)
k_nuts_spec_integrate(bjx_nuts_t nt, bjx_nuts_async_t ax, bjx_nuts_spec_t sp, const float* __restrict__ logp_f,
                      const float* __restrict__ gf, int bump) {
  const int64_t b = (int64_t)blockIdx.x;
  if (b < spec_n_rows(sp)) spec_integrate_row<NI>(nt, ax, sp, logp_f, gf, b);
  if (bump && b == 0 && threadIdx.x == 0) atomicAdd(sp.a_seq, 1);  // the last launch of a sequence of stream A
}

// Stream B: one wave per row, alive for one whole sequence of stream A (`target` = the value *a_seq reaches when that
// sequence has completed).  A row's wave: transition end pending -> async_end2_chain, publish the new epoch; leaf in
// flight -> wait for its record, then the tick kernel's leaf (k_nuts_async_tick3's phase-1 branch, cold path).
// Reading the ring.  A record is announced one integrate launch after the launch that wrote it, i.e. once that launch
// has COMPLETED and its stores are performed at agent scope; the wave only requests the record after the
// announcement has arrived (control dependence), and it reads the ring with agent-scope (sc1) loads, which are
// serviced at the coherence point instead of a possibly stale line of this XCD's L2 (a slot's previous lap).  So the
// per-record path needs no cache invalidation: the wave's own state (record words, momentum, pending position,
// momentum sums, checkpoints) stays in its L1 / L2.  (First form of this kernel: an agent-scope fence per record --
// L2 write-back + invalidate -- and three dependent round trips: 5.3 us per record against stream A's 6.5 us per
// leaf; the bookkeeper fell a ring behind and a quarter of the pushed leaves were wasted.)
template <int VEC>
__device__ __forceinline__ Row<VEC> ldr_agent(const float* p) {
  Row<VEC> r;
#pragma unroll
  for (int e = 0; e < VEC; ++e) r.v[e] = __int_as_float(ld_agent(reinterpret_cast<const int*>(p) + e));
  return r;
}

template <int NI>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2)))  // a wave per row, rows << CUs: registers are free
k_nuts_spec_book(bjx_nuts_t nt, bjx_nuts_async_t ax, bjx_nuts_spec_t sp, int target, long long timeout_ticks) {
  constexpr int VEC = 4;
  constexpr int GL = 64;
  const int g = threadIdx.x;
  const int64_t b = (int64_t)blockIdx.x;
  if (b >= spec_n_rows(sp)) return;
  const int64_t c = (int64_t)__builtin_amdgcn_readfirstlane(sp.rows[b]);
  int* recp = ax.rec + c * BJX_NUTS_REC_WORDS;
  int* bwp = sp.bw + b * BJX_NUTS_SPEC_IW;
  int my_ep = __builtin_amdgcn_readfirstlane(bwp[BWD_EP]);
  int cnt = __builtin_amdgcn_readfirstlane(bwp[BWD_CNT]);
  if (my_ep == -1) return;
  const int64_t base = c * nt.D, rbase = b * nt.D;
  const float* im = nt.imm + c * nt.imm_stride;
  const long long t0 = wall_clock64();
  long long busy = 0;
  int n_rec = 0, n_stale = 0, n_order = 0;
  int phase = __builtin_amdgcn_readfirstlane(ax.phase[c]);
  int n_avail = cnt;
  // has the integrator acknowledged this epoch?  (its acknowledgement names the first record of the epoch, so the
  // leaves it speculated past the end of the previous transition are skipped by count, not read one by one)
  const long long* ackp = reinterpret_cast<const long long*>(sp.ack) + b;
  bool need_ack = (int)(__builtin_amdgcn_readfirstlane((int)(ld64_agent(ackp) >> 32))) != my_ep;
  // the NEXT record's identity and gradient, requested while this one is worked on (an agent-scope load is ~2 us)
  int pf_cnt = -1;
  int pf_tg[BJX_NUTS_SPEC_TAG];
  Row<VEC> pf_G[NI];
  for (int it = 0;; ++it) {
    if (phase == 2) {
      if (g == 0) st_agent(bwp + BWD_EP, -1);
      break;
    }
    if (phase == 3 || phase == 0) {  // record, accept, momentum draw, tree start, first opening half -> qf_book / front_p
      if (it) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // this wave's own stores of the iteration before
      int w = recp[g & (BJX_NUTS_REC_WORDS - 1)];
      const int w_in = w;
      const bool pending = async_end2_chain<NI>(nt, ax, sp.qf_book, c, b, phase, w);
      if (g < BJX_NUTS_REC_WORDS && w != w_in) recp[g] = w;
      if (!pending) {
        if (g == 0) st_agent(bwp + BWD_EP, -1);
        break;
      }
      my_ep += 1;
      const int ik0 = rec_i(w, RW_IK), ik1 = rec_i(w, RW_IKB), epsb = rec_i(w, RW_EPS);
      if (g == 0) {
        st_agent(bwp + BWD_IK0, ik0);
        st_agent(bwp + BWD_IK1, ik1);
        st_agent(bwp + BWD_EPS, epsb);
      }
      __threadfence();  // release: chain state, first pending position / momentum, key and step size before the epoch
      if (g == 0) st_agent(bwp + BWD_EP, my_ep);
      phase = 1;
      need_ack = true;
      continue;
    }
    // ---- a leaf is in flight (phase 1): wait for a record (the announcement is only re-read when the records
    // known so far are used up: an agent-scope load is a trip to the coherence point, ~2 us)
    bool give_up = false;
    if (need_ack) {
      long long a = ld64_agent(ackp);
      while ((int)(a >> 32) != my_ep) {
        if (ld_agent(sp.a_seq) - target >= 0 || wall_clock64() - t0 > timeout_ticks) { give_up = true; break; }
        __builtin_amdgcn_s_sleep(8);
        a = ld64_agent(ackp);
      }
      if (give_up) break;
      const int first = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffll));
      n_stale += first - cnt;
      cnt = first;
      if (g == 0) st_agent(bwp + BWD_CNT, cnt);
      need_ack = false;
    }
    if (n_avail - cnt <= 0) n_avail = __builtin_amdgcn_readfirstlane(ld_agent(sp.avail + b));
    while (n_avail - cnt <= 0) {
      if (ld_agent(sp.a_seq) - target >= 0) { give_up = true; break; }  // stream A's sequence is over: the next launch carries on
      if (wall_clock64() - t0 > timeout_ticks) {
        if (g == 0) atomicAdd(sp.dbg + SPD_TIMEOUT, 1);
        give_up = true;
        break;
      }
      __builtin_amdgcn_s_sleep(8);
      n_avail = __builtin_amdgcn_readfirstlane(ld_agent(sp.avail + b));
    }
    if (give_up) break;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // compiler ordering only: no ring load before the announcement
    const long long tb = wall_clock64();
    if (it) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // this wave's own stores of the iteration before
    // one round trip: the record's identity and gradient (agent-scope loads) + this chain's own state
    const int slot = cnt & (sp.ring - 1);
    const int* tp = sp.ring_tag + ((int64_t)b * sp.ring + slot) * BJX_NUTS_SPEC_TAG;
    const float* rg = sp.ring_g + ((int64_t)b * sp.ring + slot) * nt.D;
    const bool pf_hit = pf_cnt == cnt;
    int tg[BJX_NUTS_SPEC_TAG];
#pragma unroll
    for (int k = 0; k < BJX_NUTS_SPEC_TAG; ++k) tg[k] = pf_hit ? pf_tg[k] : ld_agent(tp + k);
    int rw[kRecHot];
#pragma unroll
    for (int k = 0; k < kRecHot / 4; ++k) {
      const int4 t = *reinterpret_cast<const int4*>(recp + 4 * k);
      rw[4 * k] = t.x; rw[4 * k + 1] = t.y; rw[4 * k + 2] = t.z; rw[4 * k + 3] = t.w;
    }
    LeafRows<NI> R;
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const uint32_t j = ((uint32_t)g + (uint32_t)GL * k) * VEC;
      if (j < (uint32_t)nt.D) {
        R.G[k] = pf_hit ? pf_G[k] : ldr_agent<VEC>(rg + j);
        R.M[k] = ldr<VEC>(im + j);
        R.P[k] = ldr<VEC>(ax.front_p + base + j);
        R.X[k] = ldr<VEC>(sp.qf_book + rbase + j);
        R.S[k] = ldr<VEC>(nt.Smsum + base + j);
      }
    }
#pragma unroll
    for (int k = 0; k < kRecHot; ++k) rw[k] = __builtin_amdgcn_readfirstlane(rw[k]);
#pragma unroll
    for (int k = 0; k < 4; ++k) tg[k] = __builtin_amdgcn_readfirstlane(tg[k]);
    if (!(tg[TG_EP] == my_ep && tg[TG_DEPTH] == rw[RW_DEPTH] && tg[TG_S] == rw[RW_SUBN])) {
      // not the next leaf of this transition: a leaf speculated past the end of an earlier one.  Skip those
      // (identity words only) up to the first record of this epoch or the end of what is announced.
      if (tg[TG_EP] == my_ep) ++n_order; else ++n_stale;
      ++cnt;
      while (cnt < n_avail) {
        const int* tq = sp.ring_tag + ((int64_t)b * sp.ring + (cnt & (sp.ring - 1))) * BJX_NUTS_SPEC_TAG;
        const int e = __builtin_amdgcn_readfirstlane(ld_agent(tq + TG_EP));
        if (e == my_ep) break;
        ++n_stale;
        ++cnt;
      }
      if (g == 0) st_agent(bwp + BWD_CNT, cnt);
      busy += wall_clock64() - tb;
      continue;
    }
    {  // the integrator evaluated the gradient at ITS position: it must be this replica's, bit for bit
      const bool bad = g == 0 && (tg[TG_X0] != __float_as_int(R.X[0].v[0]) || tg[TG_X0 + 1] != __float_as_int(R.X[0].v[1]) ||
                                  tg[TG_X0 + 2] != __float_as_int(R.X[0].v[2]) || tg[TG_X0 + 3] != __float_as_int(R.X[0].v[3]));
      if (bad) atomicAdd(sp.dbg + SPD_MISMATCH, 1);
    }
    const float lp = __int_as_float(tg[TG_LP]);
    const bool done = async_leaf3_row<GL, NI>(nt, ax, sp.qf_book, lp, c, b, recp, rw, R);
    if (g == 0) {
#pragma unroll
      for (int k = 0; k < kRecHot / 4; ++k)
        *reinterpret_cast<int4*>(recp + 4 * k) = make_int4(rw[4 * k], rw[4 * k + 1], rw[4 * k + 2], rw[4 * k + 3]);
    }
    if (done) phase = 3;  // (async_leaf3_row has written ax.phase[c] = 3 as well)
    cnt += 1;
    if (!done && n_avail - cnt > 0) {  // request the next record now: it arrives behind this leaf's stores and the fence
      const int slot2 = cnt & (sp.ring - 1);
      const int* tp2 = sp.ring_tag + ((int64_t)b * sp.ring + slot2) * BJX_NUTS_SPEC_TAG;
      const float* rg2 = sp.ring_g + ((int64_t)b * sp.ring + slot2) * nt.D;
#pragma unroll
      for (int k = 0; k < BJX_NUTS_SPEC_TAG; ++k) pf_tg[k] = ld_agent(tp2 + k);
#pragma unroll
      for (int k = 0; k < NI; ++k) {
        const uint32_t j = ((uint32_t)g + (uint32_t)GL * k) * VEC;
        if (j < (uint32_t)nt.D) pf_G[k] = ldr_agent<VEC>(rg2 + j);
      }
      pf_cnt = cnt;
    }
    if (g == 0) st_agent(bwp + BWD_CNT, cnt);  // (the record's loads have returned: the leaf used them)
    busy += wall_clock64() - tb;
    n_rec += 1;
  }
  if (g == 0) {
    st_agent(bwp + BWD_CNT, cnt);
    if (n_stale) atomicAdd(sp.dbg + SPD_STALE, n_stale);
    if (n_order) atomicAdd(sp.dbg + SPD_ORDER, n_order);
    if (n_rec) {  // statistics: time spent on records (100 MHz ticks) and records consumed
      atomicAdd(sp.dbg + 6, (int)busy);
      atomicAdd(sp.dbg + 7, n_rec);
    }
  }
}


// Do two streams really run concurrently?  (Streams share a few hardware queues; two streams on one queue execute
// in order, and a bookkeeper queued in front of its sequence would then wait for its time-out.)  The wait kernel,
// launched FIRST, spins until the set kernel on the other stream has run, or gives up: flag2[1] = 1 / 2.
__global__ void k_stream_probe_wait(int* flag2, long long timeout_ticks) {
  const long long t0 = wall_clock64();
  int seen = 0;
  while (!(seen = ld_agent(flag2)) && wall_clock64() - t0 < timeout_ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) flag2[1] = seen ? 1 : 2;
}
__global__ void k_stream_probe_set(int* flag2) {
  if (threadIdx.x == 0) st_agent(flag2, 1);
}

// Hand-over from the one-stream tail: stream A's words, ends and momentum from the chain's record and arrays.
template <int NI>
__global__ void __launch_bounds__(64)
k_nuts_spec_enter(bjx_nuts_t nt, bjx_nuts_async_t ax, bjx_nuts_spec_t sp) {
  constexpr int VEC = 4;
  const int g = threadIdx.x;
  const int64_t b = (int64_t)blockIdx.x;
  if (b >= spec_n_rows(sp)) return;
  const int64_t c = (int64_t)__builtin_amdgcn_readfirstlane(sp.rows[b]);
  const int phase = __builtin_amdgcn_readfirstlane(ax.phase[c]);
  const int w = ax.rec[c * BJX_NUTS_REC_WORDS + (g & (BJX_NUTS_REC_WORDS - 1))];
  const int depth = rec_i(w, RW_DEPTH), s = rec_i(w, RW_SUBN), lazy = rec_i(w, RW_LAZY), dir = rec_i(w, RW_DIR);
  const int eps_bits = rec_i(w, RW_EPS), ik0 = rec_i(w, RW_IK), ik1 = rec_i(w, RW_IKB);
  const int dirs = spec_dirs(Key{(uint32_t)ik0, (uint32_t)ik1}, nt.max_depth);
  const int64_t base = c * nt.D, rbase = b * nt.D;
  const bool zl = (lazy & LZ_L) != 0, zr = (lazy & LZ_R) != 0;
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    const uint32_t j = ((uint32_t)g + 64u * k) * VEC;
    if (j < (uint32_t)nt.D) {
      str<VEC>(sp.qf_book + rbase + j, ldr<VEC>(sp.qf + rbase + j));
      str<VEC>(sp.fp + rbase + j, ldr<VEC>(ax.front_p + base + j));
      str<VEC>(sp.eLq + rbase + j, ldr<VEC>((zl ? nt.q0 : nt.Lq) + base + j));
      str<VEC>(sp.eLp + rbase + j, ldr<VEC>((zl ? nt.p0 : nt.Lp) + base + j));
      str<VEC>(sp.eLg + rbase + j, ldr<VEC>((zl ? nt.g0 : nt.Lg) + base + j));
      str<VEC>(sp.eRq + rbase + j, ldr<VEC>((zr ? nt.q0 : nt.Rq) + base + j));
      str<VEC>(sp.eRp + rbase + j, ldr<VEC>((zr ? nt.p0 : nt.Rp) + base + j));
      str<VEC>(sp.eRg + rbase + j, ldr<VEC>((zr ? nt.g0 : nt.Rg) + base + j));
    }
  }
  if (g == 0) {
    const bool run = phase == 1;
    int* iwp = sp.iw + b * BJX_NUTS_SPEC_IW;
    int* bwp = sp.bw + b * BJX_NUTS_SPEC_IW;
    *reinterpret_cast<int4*>(iwp) = make_int4(0, run ? depth : 0, run ? s : 0, dirs);
    *reinterpret_cast<int4*>(iwp + 4) = make_int4(eps_bits, 0, run ? SPS_RUN : SPS_WAIT, 0);
    *reinterpret_cast<int4*>(bwp) = make_int4(phase == 2 ? -1 : 0, 0, ik0, ik1);
    *reinterpret_cast<int4*>(bwp + 4) = make_int4(eps_bits, 0, 0, 0);
    sp.avail[b] = 0;
    reinterpret_cast<long long*>(sp.ack)[b] = 0;  // epoch 0 starts at record 0
    if (run && (((dirs >> depth) & 1) != (dir > 0 ? 1 : 0))) atomicAdd(sp.dbg + SPD_ORDER, 1);
  }
}

}  // namespace

extern "C" {

int bjx_stream_probe(void* stream_wait, void* stream_set, int32_t* flag2, int32_t timeout_us) {
  BJX_CHECK_ARG(flag2 && timeout_us > 0 && stream_wait != stream_set, "bjx_stream_probe: two streams, flag2 and a time-out needed");
  hipLaunchKernelGGL(k_stream_probe_wait, dim3(1), dim3(64), 0, (hipStream_t)stream_wait, flag2, (long long)timeout_us * 100);
  hipLaunchKernelGGL(k_stream_probe_set, dim3(1), dim3(64), 0, (hipStream_t)stream_set, flag2);
  return bjx_check_launch("bjx_stream_probe");
}

namespace {
int check_spec(const bjx_nuts_t* nuts, const bjx_nuts_async_t* run, const bjx_nuts_spec_t* sp, const char* what) {
  if (check_nuts(nuts, what)) return 1;
  const bool ok = run && sp && run->q && run->g && run->logp && run->p && run->t && run->phase && run->n_done &&
                  run->q == nuts->q0 && run->g == nuts->g0 && run->p == nuts->p0 && run->rec && run->front_p &&
                  !nuts->Mdense && !run->adapt_tab && !run->gemm_pc && run->target_kind == BJX_TARGET_NONE &&
                  run->int_stages <= 1 && nuts->max_depth >= 1 && nuts->max_depth <= 30 && nuts->D <= 1024 &&
                  sp->n_rows >= 1 && sp->n_rows <= nuts->N && sp->rows && sp->ring >= 8 &&
                  (sp->ring & (sp->ring - 1)) == 0 && sp->qf && sp->fp && sp->eLq && sp->eLp && sp->eLg && sp->eRq &&
                  sp->eRp && sp->eRg && sp->iw && sp->ring_g && sp->ring_tag && sp->avail && sp->ack && sp->qf_book && sp->bw &&
                  sp->a_seq && sp->dbg &&
                  nuts_vec4(nuts, sp->qf, sp->fp, sp->eLq, sp->eLp, sp->eLg, sp->eRq, sp->eRp, sp->eRg, sp->ring_g,
                            sp->qf_book, run->front_p, run->out_position) &&
                  ((uintptr_t)sp->iw & 15) == 0 && ((uintptr_t)sp->bw & 15) == 0 && ((uintptr_t)sp->ack & 7) == 0 && ((uintptr_t)sp->ring_tag & 15) == 0 &&
                  ((uintptr_t)run->rec & 15) == 0;
  if (!ok) {
    bjx_set_error("%s: the speculative tail serves a diagonal metric, D %% 4 == 0, D <= 1024, 16-byte aligned "
                  "buffers, one-gradient integrators, an external callable and no per-chain adaptation; every "
                  "bjx_nuts_spec_t buffer must be given and ring a power of two >= 8", what);
    return 1;
  }
  return 0;
}
#define BJX_SPEC_LAUNCH(KERNEL, stream, ...)                                                        \
  do {                                                                                              \
    const dim3 g1((unsigned)spec->n_rows);                                                          \
    hipStream_t s_ = (hipStream_t)stream;                                                           \
    if (nuts->D <= 256) hipLaunchKernelGGL((KERNEL<1>), g1, dim3(64), 0, s_, __VA_ARGS__);          \
    else if (nuts->D <= 512) hipLaunchKernelGGL((KERNEL<2>), g1, dim3(64), 0, s_, __VA_ARGS__);     \
    else if (nuts->D <= 768) hipLaunchKernelGGL((KERNEL<3>), g1, dim3(64), 0, s_, __VA_ARGS__);     \
    else hipLaunchKernelGGL((KERNEL<4>), g1, dim3(64), 0, s_, __VA_ARGS__);                         \
  } while (0)
}  // namespace

int bjx_nuts_spec_enter(void* stream, const bjx_nuts_t* nuts, const bjx_nuts_async_t* run, const bjx_nuts_spec_t* spec) {
  if (check_spec(nuts, run, spec, "bjx_nuts_spec_enter")) return 1;
  BJX_SPEC_LAUNCH(k_nuts_spec_enter, stream, *nuts, *run, *spec);
  return bjx_check_launch("bjx_nuts_spec_enter");
}

int bjx_nuts_spec_integrate(void* stream, const bjx_nuts_t* nuts, const bjx_nuts_async_t* run,
                            const bjx_nuts_spec_t* spec, const float* logp_f, const float* gf, int32_t bump) {
  if (check_spec(nuts, run, spec, "bjx_nuts_spec_integrate")) return 1;
  BJX_CHECK_ARG(logp_f && gf && bjx_vec4_ok(nuts->D, gf), "bjx_nuts_spec_integrate: logp_f / gf (16-byte aligned) needed");
  BJX_SPEC_LAUNCH(k_nuts_spec_integrate, stream, *nuts, *run, *spec, logp_f, gf, (int)bump);
  return bjx_check_launch("bjx_nuts_spec_integrate");
}

int bjx_nuts_spec_book(void* stream, const bjx_nuts_t* nuts, const bjx_nuts_async_t* run,
                       const bjx_nuts_spec_t* spec, int32_t target, int32_t timeout_us) {
  if (check_spec(nuts, run, spec, "bjx_nuts_spec_book")) return 1;
  BJX_CHECK_ARG(timeout_us >= 0, "bjx_nuts_spec_book: timeout_us must be >= 0");
  const long long ticks = (long long)timeout_us * 100;  // the wall clock counts at 100 MHz
  BJX_SPEC_LAUNCH(k_nuts_spec_book, stream, *nuts, *run, *spec, (int)target, ticks);
  return bjx_check_launch("bjx_nuts_spec_book");
}

}  // extern "C"
