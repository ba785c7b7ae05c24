// Microbenchmark: launch-geometry variants of the diag leapfrog + elementwise-gradient loop at the
// headline shape (rows of 1 024 floats), to see how far the product kernels are from what the
// memory system gives a plain flat sweep of the same bytes.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/lf_variants.hip -o tools/lf_variants
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
struct alignas(16) F4 { float x, y, z, w; };
#define D4 256  // float4 per row (D = 1024)

__device__ __forceinline__ void lf_math(F4& pp, const F4& gg, F4& qq, const F4& mm, float h, float ed) {
  pp.x = fmaf(h, gg.x, pp.x); pp.y = fmaf(h, gg.y, pp.y); pp.z = fmaf(h, gg.z, pp.z); pp.w = fmaf(h, gg.w, pp.w);
  pp.x = fmaf(h, gg.x, pp.x); pp.y = fmaf(h, gg.y, pp.y); pp.z = fmaf(h, gg.z, pp.z); pp.w = fmaf(h, gg.w, pp.w);
  qq.x = fmaf(ed, mm.x * pp.x, qq.x); qq.y = fmaf(ed, mm.y * pp.y, qq.y);
  qq.z = fmaf(ed, mm.z * pp.z, qq.z); qq.w = fmaf(ed, mm.w * pp.w, qq.w);
}

// A: one row per wave, 4 unrolled float4 per array (the product kernel's shape); REV sweeps last-to-first
template <bool REV>
__global__ void __launch_bounds__(256) lf_row(F4* q, F4* p, const F4* __restrict__ g, const F4* __restrict__ imm,
                                              size_t N, float h, float ed) {
  const int lane = threadIdx.x & 63;
  size_t r = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= N) return;
  if (REV) r = N - 1 - r;
  const size_t base = r * D4;
  F4 pp[4], gg[4], qq[4], mm[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) { pp[u] = p[base + lane + 64 * u]; gg[u] = g[base + lane + 64 * u]; qq[u] = q[base + lane + 64 * u]; mm[u] = imm[lane + 64 * u]; }
#pragma unroll
  for (int u = 0; u < 4; ++u) { lf_math(pp[u], gg[u], qq[u], mm[u], h, ed); p[base + lane + 64 * u] = pp[u]; q[base + lane + 64 * u] = qq[u]; }
}

// B: flat grid-stride over float4 elements, U independent elements in flight per thread
template <int U>
__global__ void __launch_bounds__(256) lf_flat(F4* q, F4* p, const F4* __restrict__ g, const F4* __restrict__ imm,
                                               size_t n4, float h, float ed) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x; i0 < n4; i0 += stride * U) {
    F4 pp[U], gg[U], qq[U], mm[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const size_t i = i0 + u * stride; if (i < n4) { pp[u] = p[i]; gg[u] = g[i]; qq[u] = q[i]; mm[u] = imm[i & (D4 - 1)]; } }
#pragma unroll
    for (int u = 0; u < U; ++u) { const size_t i = i0 + u * stride; if (i < n4) { lf_math(pp[u], gg[u], qq[u], mm[u], h, ed); p[i] = pp[u]; q[i] = qq[u]; } }
  }
}

// C: a block of 256 threads owns 4 consecutive rows as one 16 KB span; thread t takes float4 t, t+256, ...
__global__ void __launch_bounds__(256) lf_span(F4* q, F4* p, const F4* __restrict__ g, const F4* __restrict__ imm,
                                               size_t n4, float h, float ed) {
  const size_t base = (size_t)blockIdx.x * 1024;
  F4 pp[4], gg[4], qq[4], mm[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) { const size_t i = base + threadIdx.x + 256 * u; if (i < n4) { pp[u] = p[i]; gg[u] = g[i]; qq[u] = q[i]; mm[u] = imm[i & (D4 - 1)]; } }
#pragma unroll
  for (int u = 0; u < 4; ++u) { const size_t i = base + threadIdx.x + 256 * u; if (i < n4) { lf_math(pp[u], gg[u], qq[u], mm[u], h, ed); p[i] = pp[u]; q[i] = qq[u]; } }
}

// gradient of a diagonal Gaussian: g = -(q - mu) * prec (elementwise), one row per wave / flat
__global__ void __launch_bounds__(256) grad_row(const F4* __restrict__ q, F4* __restrict__ g, const F4* __restrict__ mu,
                                                const F4* __restrict__ pr, size_t N) {
  const int lane = threadIdx.x & 63;
  const size_t r = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= N) return;
  const size_t base = r * D4;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const F4 a = q[base + lane + 64 * u], m = mu[lane + 64 * u], s = pr[lane + 64 * u];
    g[base + lane + 64 * u] = F4{-(a.x - m.x) * s.x, -(a.y - m.y) * s.y, -(a.z - m.z) * s.z, -(a.w - m.w) * s.w};
  }
}
template <int U>
__global__ void __launch_bounds__(256) grad_flat(const F4* __restrict__ q, F4* __restrict__ g, const F4* __restrict__ mu,
                                                 const F4* __restrict__ pr, size_t n4) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x; i0 < n4; i0 += stride * U) {
    F4 a[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const size_t i = i0 + u * stride; if (i < n4) a[u] = q[i]; }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = i0 + u * stride;
      if (i < n4) { const F4 m = mu[i & (D4 - 1)], s = pr[i & (D4 - 1)];
        g[i] = F4{-(a[u].x - m.x) * s.x, -(a[u].y - m.y) * s.y, -(a[u].z - m.z) * s.z, -(a[u].w - m.w) * s.w}; }
    }
  }
}

// pseudo-random fill: all-zero buffers toggle no data lines and flatter a power-limited part
__global__ void fill_random(float* a, size_t n, unsigned seed, float lo, float hi) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    a[i] = lo + (hi - lo) * (float)(x >> 8) * (1.0f / 16777216.0f);
  }
}

int main(int argc, char** argv) {
  const int L = 50;
  const bool randomize = argc > 1 && atoi(argv[1]) != 0;
  F4 *q, *p, *g, *imm, *mu, *pr;
  const size_t maxN = 65536;
  hipMalloc(&q, maxN * D4 * 16); hipMalloc(&p, maxN * D4 * 16); hipMalloc(&g, maxN * D4 * 16);
  hipMalloc(&imm, D4 * 16); hipMalloc(&mu, D4 * 16); hipMalloc(&pr, D4 * 16);
  hipMemset(q, 0, maxN * D4 * 16); hipMemset(p, 0, maxN * D4 * 16); hipMemset(g, 0, maxN * D4 * 16);
  hipMemset(imm, 0, D4 * 16); hipMemset(mu, 0, D4 * 16); hipMemset(pr, 0, D4 * 16);
  if (randomize) {
    fill_random<<<4096, 256>>>((float*)q, maxN * 1024, 1u, -1.0f, 1.0f);
    fill_random<<<4096, 256>>>((float*)p, maxN * 1024, 2u, -1.0f, 1.0f);
    fill_random<<<4096, 256>>>((float*)g, maxN * 1024, 3u, -1.0f, 1.0f);
    fill_random<<<4, 256>>>((float*)imm, 1024, 4u, 0.5f, 2.0f);
    fill_random<<<4, 256>>>((float*)mu, 1024, 5u, -1.0f, 1.0f);
    fill_random<<<4, 256>>>((float*)pr, 1024, 6u, 0.5f, 2.0f);
  }
  printf("data: %s\n", randomize ? "pseudo-random" : "zeros");
  hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
  const float h = 0.125f, ed = 0.25f;
  for (size_t N : {(size_t)16384, (size_t)65536}) {
    const size_t n4 = N * D4;
    const unsigned rowgrid = (unsigned)(N / 4);
    for (int v = 0; v < 9; ++v) {
      auto lf = [&]() {
        switch (v) {
          case 0: lf_row<false><<<rowgrid, 256>>>(q, p, g, imm, N, h, ed); break;
          case 1: lf_row<true><<<rowgrid, 256>>>(q, p, g, imm, N, h, ed); break;
          case 2: lf_flat<4><<<(unsigned)(n4 / 1024), 256>>>(q, p, g, imm, n4, h, ed); break;
          case 3: lf_flat<4><<<2048, 256>>>(q, p, g, imm, n4, h, ed); break;
          case 4: lf_flat<2><<<4096, 256>>>(q, p, g, imm, n4, h, ed); break;
          case 5: lf_flat<1><<<4096, 256>>>(q, p, g, imm, n4, h, ed); break;
          case 6: lf_flat<1><<<(unsigned)(n4 / 256), 256>>>(q, p, g, imm, n4, h, ed); break;
          case 7: lf_span<<<(unsigned)(n4 / 1024), 256>>>(q, p, g, imm, n4, h, ed); break;
          case 8: lf_flat<8><<<1024, 256>>>(q, p, g, imm, n4, h, ed); break;
        }
      };
      auto gr = [&]() {
        if (v == 0 || v == 1 || v == 7) grad_row<<<rowgrid, 256>>>(q, g, mu, pr, N);
        else if (v == 6) grad_flat<1><<<(unsigned)(n4 / 256), 256>>>(q, g, mu, pr, n4);
        else grad_flat<4><<<(unsigned)(n4 / 1024), 256>>>(q, g, mu, pr, n4);
      };
      float best_loop = 1e30f, best_lf = 1e30f, best_gr = 1e30f;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        for (int s = 0; s < L; ++s) { lf(); gr(); }
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best_loop) best_loop = ms;
        // the kernels alone, back to back (same working set)
        hipEventRecord(e0);
        for (int s = 0; s < L; ++s) lf();
        hipEventRecord(e1);
        for (int s = 0; s < L; ++s) gr();
        hipEventRecord(e2);
        hipEventSynchronize(e2);
        float a, b; hipEventElapsedTime(&a, e0, e1); hipEventElapsedTime(&b, e1, e2);
        if (rep && a < best_lf) best_lf = a;
        if (rep && b < best_gr) best_gr = b;
      }
      const double bytes_lf = 20.0 * N * 1024, bytes_gr = 8.0 * N * 1024;
      printf("N %6zu v%d : loop %7.1f us/step (%5.2f TB/s of 28 B) | lf alone %6.1f us (%5.2f TB/s) | grad alone %6.1f us (%5.2f TB/s)\n",
             N, v, best_loop * 1e3 / L, (bytes_lf + bytes_gr) / (best_loop * 1e-3 / L) / 1e12, best_lf * 1e3 / L,
             bytes_lf / (best_lf * 1e-3 / L) / 1e12, best_gr * 1e3 / L, bytes_gr / (best_gr * 1e-3 / L) / 1e12);
    }
  }
  return 0;
}
