// Microbenchmark of the free-running NUTS tick's MEMORY PATTERN (round 3, VERDICT r2 "next" #1a):
// one wave per chain row of 1 KB (D = 256), a dozen arrays, a dependent second round trip for half of
// the rows, a stretch of dependent arithmetic, then 3-6 row stores -- what k_nuts_async_tick2 does,
// without its arithmetic.  It answers, per launch form, how many TB/s that pattern can reach:
//   form 0: one row per 64-thread workgroup, grid = rows              (the round-2 product kernel)
//   form 1: persistent 64-thread workgroups looping over rows          (no prefetch)
//   form 2: persistent + the next row's first-round-trip loads issued before the current row's work
//   form 3: four rows per 256-thread workgroup, grid = rows / 4
// x occupancy (waves per CU, limited through the dynamic LDS size as the product's VGPR count does)
// x arithmetic length (dependent fp64 fma chain) x plain / nontemporal stores of the rows nobody
// re-reads inside the tick (proposal rows, checkpoints).
// Build: hipcc --offload-arch=gfx950 -O3 tools/tickbw.hip -o tools/tickbw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

typedef float f4 __attribute__((ext_vector_type(4)));
struct Args {
  const f4* gf; f4* qf; f4* fp; f4* sm; f4* ck; f4* sq; f4* sg; const f4* imm;
  int* rec; const int* phase;
  int n_rows, depth, fill, tick;
};
__device__ __forceinline__ unsigned hash2(unsigned b, unsigned t) {
  unsigned x = b * 2654435761u + t * 40503u + 12345u;
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
template <bool NT> __device__ __forceinline__ void st(f4* p, f4 v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}
struct Rows { f4 G, M, P, X, S; int w, ph; };

__device__ __forceinline__ Rows load_rows(const Args& a, int b) {
  const int lane = threadIdx.x & 63;
  Rows r;
  const size_t o = (size_t)b * 64 + lane;
  r.ph = a.phase[b];
  r.w = a.rec[(size_t)b * 32 + (lane & 31)];
  r.G = a.gf[o]; r.M = a.imm[lane]; r.P = a.fp[o]; r.X = a.qf[o]; r.S = a.sm[o];
  return r;
}
// second round trip of row b (issued BEFORE the next row's prefetch so it does not queue behind it)
__device__ __forceinline__ f4 issue2(const Args& a, int b) {
  const int lane = threadIdx.x & 63;
  const unsigned h = hash2((unsigned)b, (unsigned)a.tick);
  const int level = (int)((h >> 1) % (unsigned)a.depth);
  f4 C = {0, 0, 0, 0};
  if (h & 1u) C = a.ck[((size_t)b * a.depth + level) * 64 + lane];
  return C;
}
template <bool NT>
__device__ __forceinline__ void work(const Args& a, int b, Rows r, f4 C) {
  const int lane = threadIdx.x & 63;
  const size_t o = (size_t)b * 64 + lane;
  const unsigned h = hash2((unsigned)b, (unsigned)a.tick);
  const bool odd = h & 1u;
  const int level = (int)((h >> 1) % (unsigned)a.depth);
  const bool take = ((h >> 8) % 5u) < 2u;
  f4* ckrow = a.ck + ((size_t)b * a.depth + level) * 64 + lane;
  // pass 1 + reduction stand-in
  double acc = 0.0;
  for (int e = 0; e < 4; ++e) { r.P[e] = fmaf(0.05f, r.G[e], r.P[e]); acc += (double)(r.M[e] * r.P[e]) * (double)r.P[e]; }
  // dependent wave-uniform arithmetic (threefry + fp64 exp / log1p in the product)
  double x = acc + (double)__builtin_amdgcn_readfirstlane(r.w) * 1e-9 + (double)r.ph;
  for (int i = 0; i < a.fill; ++i) {
    x = fma(x, 0.999999, 1e-7); x = fma(x, 1.000001, -1e-7); x = fma(x, 0.999999, 1e-7); x = fma(x, 1.000001, -1e-7);
  }
  const float xf = (float)x * 1e-30f;
  for (int e = 0; e < 4; ++e) {
    r.S[e] = r.S[e] + r.P[e] + C[e] * 1e-20f + xf;
    r.P[e] = fmaf(0.05f, r.G[e], r.P[e]);
    r.X[e] = fmaf(0.1f, r.M[e] * r.P[e], r.X[e]);
  }
  if (!odd) st<NT>(ckrow, r.P);
  if (take) { st<NT>(a.sq + o, r.X); st<NT>(a.sg + o, r.G); }
  a.fp[o] = r.P; a.qf[o] = r.X; a.sm[o] = r.S;
  if (lane < 32) a.rec[(size_t)b * 32 + lane] = r.w + 1;
}

extern __shared__ char lds_pad[];

template <int FORM, bool NT>
__global__ void __launch_bounds__(FORM == 3 ? 256 : 64) k_tick(Args a) {
  if (threadIdx.x == 1023) lds_pad[0] = 0;  // keep the dynamic LDS allocation
  if constexpr (FORM == 0) {
    const int b = blockIdx.x;
    if (b < a.n_rows) { const Rows r = load_rows(a, b); work<NT>(a, b, r, issue2(a, b)); }
  } else if constexpr (FORM == 3) {
    const int b = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (b < a.n_rows) { const Rows r = load_rows(a, b); work<NT>(a, b, r, issue2(a, b)); }
  } else if constexpr (FORM == 1) {
    for (int b = blockIdx.x; b < a.n_rows; b += gridDim.x) { const Rows r = load_rows(a, b); work<NT>(a, b, r, issue2(a, b)); }
  } else {
    int b = blockIdx.x;
    if (b >= a.n_rows) return;
    Rows cur = load_rows(a, b);
    for (;;) {
      const int nb = b + (int)gridDim.x;
      const bool more = nb < a.n_rows;
      const f4 C = issue2(a, b);
      Rows nxt = cur;
      if (more) nxt = load_rows(a, nb);
      work<NT>(a, b, cur, C);
      if (!more) break;
      cur = nxt;
      b = nb;
    }
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 32768, depth = 10;
  const size_t row = 1024;
  f4 *gf, *qf, *fp, *sm, *ck, *sq, *sg, *imm; int *rec, *phase;
  CK(hipMalloc(&gf, N * row)); CK(hipMalloc(&qf, N * row)); CK(hipMalloc(&fp, N * row)); CK(hipMalloc(&sm, N * row));
  CK(hipMalloc(&ck, N * row * depth)); CK(hipMalloc(&sq, N * row)); CK(hipMalloc(&sg, N * row)); CK(hipMalloc(&imm, row));
  CK(hipMalloc(&rec, N * 128)); CK(hipMalloc(&phase, N * 4));
  CK(hipMemset(gf, 0x11, N * row)); CK(hipMemset(qf, 0x12, N * row)); CK(hipMemset(fp, 0x13, N * row)); CK(hipMemset(sm, 0x14, N * row));
  CK(hipMemset(ck, 0x15, N * row * depth)); CK(hipMemset(sq, 0, N * row)); CK(hipMemset(sg, 0, N * row)); CK(hipMemset(imm, 0x11, row));
  CK(hipMemset(rec, 0, N * 128)); CK(hipMemset(phase, 0, N * 4));
  // bytes one launch moves (same hash as the kernel)
  auto bytes_of = [&](int tick) {
    double by = 0;
    for (int b = 0; b < N; ++b) {
      unsigned x = (unsigned)b * 2654435761u + (unsigned)tick * 40503u + 12345u;
      x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
      const bool odd = x & 1u, take = ((x >> 8) % 5u) < 2u;
      by += 4 * 1024 + 128 + 4 /* trip 1 */ + 1024 /* ck read or write */ + (take ? 2048 : 0) + 3 * 1024 + 128;
      (void)odd;
    }
    return by;
  };
  const double bytes = bytes_of(0);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("{\"tool\": \"tools/tickbw.hip\", \"rows\": %d, \"row_bytes\": 1024, \"bytes_per_launch\": %.0f, \"results\": [\n", N, bytes);
  bool first = true;
  auto run = [&](int form, bool nt, int waves_per_cu, int fill) {
    Args a{gf, qf, fp, sm, ck, sq, sg, imm, rec, phase, N, depth, fill, 0};
    // occupancy limiter: workgroups per CU = 160 KiB / dynamic LDS per workgroup
    const int wg_per_cu = form == 3 ? waves_per_cu / 4 : waves_per_cu;
    size_t lds = (size_t)(160 * 1024) / wg_per_cu;
    lds = lds / 256 * 256;
    if (lds > 64 * 1024) lds = 64 * 1024;
    const int grid = form == 0 ? N : form == 3 ? N / 4 : 256 * waves_per_cu;
    auto launch = [&](int tick) {
      a.tick = tick;
      if (form == 0) { if (nt) k_tick<0, true><<<grid, 64, lds>>>(a); else k_tick<0, false><<<grid, 64, lds>>>(a); }
      if (form == 1) { if (nt) k_tick<1, true><<<grid, 64, lds>>>(a); else k_tick<1, false><<<grid, 64, lds>>>(a); }
      if (form == 2) { if (nt) k_tick<2, true><<<grid, 64, lds>>>(a); else k_tick<2, false><<<grid, 64, lds>>>(a); }
      if (form == 3) { if (nt) k_tick<3, true><<<grid, 256, lds>>>(a); else k_tick<3, false><<<grid, 256, lds>>>(a); }
    };
    for (int w = 0; w < 3; ++w) launch(w);
    const int reps = 20;
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) launch(r);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("%s  {\"form\": %d, \"nt_stores\": %d, \"waves_per_cu\": %d, \"fill\": %d, \"us\": %.1f, \"GBps\": %.0f}", first ? "" : ",\n",
           form, (int)nt, waves_per_cu, fill, ms * 1e3, bytes / ms / 1e6);
    first = false;
  };
  for (int fill : {0, 60, 150})
    for (int waves : {8, 12, 16, 24, 32})
      for (int form : {0, 1, 2, 3})
        for (int nt : {0, 1}) {
          if (nt && !(waves == 12 || waves == 16)) continue;
          run(form, nt != 0, waves, fill);
        }
  printf("\n]}\n");
  CK(hipDeviceSynchronize());
  return 0;
}
