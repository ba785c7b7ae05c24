// Streaming-bandwidth yardstick in the PRODUCT's access pattern (round 3; VERDICT r2 "next" #2).
// tools/membw.hip measures grid-stride loops; the engine's kernels do not loop: every lane moves ONE
// 16-byte piece (k_leapfrog_diag_flat), workgroups may run last-to-first, and stores may be
// nontemporal.  This tool measures exactly those forms for the three traffic mixes of the engine
//   read  : read-only sum over one array (and over TWO arrays, the ChEES weights+colstats pass)
//   copy  : b[i] = a[i]                        (1 read : 1 write, the guide's 6.29 TB/s figure)
//   lf    : p += h g; q += e p  in place       (3 reads : 2 writes, the C2 leapfrog)
// on arrays of 256 MiB, 1 GiB and 2 GiB of pseudo-random data, and prints one JSON document.
// Build: hipcc --offload-arch=gfx950 -O3 tools/membw2.hip -o tools/membw2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

template <bool NT> __device__ __forceinline__ f4 ld(const f4* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}
template <bool NT> __device__ __forceinline__ void st(f4* p, f4 v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}
// one piece per lane; REV: workgroups sweep the array last-to-first
template <bool REV> __device__ __forceinline__ size_t piece(size_t nblocks) {
  const size_t blk = REV ? (nblocks - 1 - blockIdx.x) : blockIdx.x;
  return blk * 256ull + threadIdx.x;
}

template <bool REV, bool NTL>
__global__ void __launch_bounds__(256) k_read1(const f4* __restrict__ a, float* out, size_t nblocks) {
  const f4 v = ld<NTL>(a + piece<REV>(nblocks));
  const float s = v.x + v.y + v.z + v.w;
  if (s == 12345.678f) out[0] = s;
}
template <bool REV, bool NTL>
__global__ void __launch_bounds__(256) k_read2(const f4* __restrict__ a, const f4* __restrict__ b, float* out,
                                               size_t nblocks) {
  const size_t i = piece<REV>(nblocks);
  const f4 v = ld<NTL>(a + i), u = ld<NTL>(b + i);
  const float s = v.x * u.x + v.y * u.y + v.z * u.z + v.w * u.w;
  if (s == 12345.678f) out[0] = s;
}
template <bool REV, bool NTL, bool NTS>
__global__ void __launch_bounds__(256) k_copy1(const f4* __restrict__ a, f4* __restrict__ b, size_t nblocks) {
  const size_t i = piece<REV>(nblocks);
  st<NTS>(b + i, ld<NTL>(a + i));
}
template <bool REV, bool NTL, bool NTS>
__global__ void __launch_bounds__(256) k_lf1(f4* q, f4* p, const f4* __restrict__ g, size_t nblocks) {
  const size_t i = piece<REV>(nblocks);
  f4 pp = ld<NTL>(p + i);
  const f4 gg = ld<NTL>(g + i);
  f4 qq = ld<NTL>(q + i);
  pp.x = fmaf(0.1f, gg.x, pp.x); pp.y = fmaf(0.1f, gg.y, pp.y); pp.z = fmaf(0.1f, gg.z, pp.z); pp.w = fmaf(0.1f, gg.w, pp.w);
  qq.x = fmaf(0.2f, pp.x, qq.x); qq.y = fmaf(0.2f, pp.y, qq.y); qq.z = fmaf(0.2f, pp.z, qq.z); qq.w = fmaf(0.2f, pp.w, qq.w);
  st<NTS>(p + i, pp);
  st<NTS>(q + i, qq);
}
// grid-stride forms (what tools/membw.hip measures), for the comparison on the same box
__global__ void __launch_bounds__(256) k_read_gs(const f4* __restrict__ a, float* out, size_t n) {
  float s = 0;
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) { const f4 v = a[i]; s += v.x + v.y + v.z + v.w; }
  if (s == 12345.678f) out[0] = s;
}
__global__ void __launch_bounds__(256) k_copy_gs(const f4* __restrict__ a, f4* __restrict__ b, size_t n) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) b[i] = a[i];
}
__global__ void __launch_bounds__(256) k_lf_gs(f4* q, f4* p, const f4* __restrict__ g, size_t n) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) {
    f4 pp = p[i]; const f4 gg = g[i]; f4 qq = q[i];
    pp.x = fmaf(0.1f, gg.x, pp.x); pp.y = fmaf(0.1f, gg.y, pp.y); pp.z = fmaf(0.1f, gg.z, pp.z); pp.w = fmaf(0.1f, gg.w, pp.w);
    qq.x = fmaf(0.2f, pp.x, qq.x); qq.y = fmaf(0.2f, pp.y, qq.y); qq.z = fmaf(0.2f, pp.z, qq.z); qq.w = fmaf(0.2f, pp.w, qq.w);
    p[i] = pp; q[i] = qq;
  }
}
__global__ void fill_random(float* a, size_t n, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    a[i] = -1.0f + 2.0f * (float)(x >> 8) * (1.0f / 16777216.0f);
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Rec { std::string mix, form; size_t mib; double gbps, us; };

int main() {
  const size_t maxb = 2ull << 30;
  f4 *a, *b, *c; float* out;
  CK(hipMalloc(&a, maxb)); CK(hipMalloc(&b, maxb)); CK(hipMalloc(&c, maxb)); CK(hipMalloc(&out, 4));
  fill_random<<<8192, 256>>>((float*)a, maxb / 4, 1u);
  fill_random<<<8192, 256>>>((float*)b, maxb / 4, 2u);
  fill_random<<<8192, 256>>>((float*)c, maxb / 4, 3u);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<Rec> recs;
  auto timeit = [&](const char* mix, const char* form, size_t mib, double bytes_per_launch, auto launch) {
    const int reps = mib >= 1024 ? 12 : 30;
    for (int w = 0; w < 3; ++w) launch();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    recs.push_back({mix, form, mib, bytes_per_launch / ms / 1e6, ms * 1e3});
  };
  for (size_t mib : {256, 1024, 2048}) {
    const size_t bytes = mib << 20, n = bytes / 16, nb = n / 256;
    const dim3 g((unsigned)nb), t(256);
    timeit("read", "piece", mib, bytes, [&] { k_read1<false, false><<<g, t>>>(a, out, nb); });
    timeit("read", "piece_rev", mib, bytes, [&] { k_read1<true, false><<<g, t>>>(a, out, nb); });
    timeit("read", "piece_ntload", mib, bytes, [&] { k_read1<false, true><<<g, t>>>(a, out, nb); });
    timeit("read", "gridstride_16384", mib, bytes, [&] { k_read_gs<<<16384, t>>>(a, out, n); });
    timeit("read2", "piece", mib, 2.0 * bytes, [&] { k_read2<false, false><<<g, t>>>(a, b, out, nb); });
    timeit("read2", "piece_ntload", mib, 2.0 * bytes, [&] { k_read2<false, true><<<g, t>>>(a, b, out, nb); });
    timeit("copy", "piece", mib, 2.0 * bytes, [&] { k_copy1<false, false, false><<<g, t>>>(a, b, nb); });
    timeit("copy", "piece_rev", mib, 2.0 * bytes, [&] { k_copy1<true, false, false><<<g, t>>>(a, b, nb); });
    timeit("copy", "piece_ntstore", mib, 2.0 * bytes, [&] { k_copy1<false, false, true><<<g, t>>>(a, b, nb); });
    timeit("copy", "piece_ntload_ntstore", mib, 2.0 * bytes, [&] { k_copy1<false, true, true><<<g, t>>>(a, b, nb); });
    timeit("copy", "gridstride_16384", mib, 2.0 * bytes, [&] { k_copy_gs<<<16384, t>>>(a, b, n); });
    timeit("lf3r2w", "piece", mib, 5.0 * bytes, [&] { k_lf1<false, false, false><<<g, t>>>(a, b, c, nb); });
    timeit("lf3r2w", "piece_rev", mib, 5.0 * bytes, [&] { k_lf1<true, false, false><<<g, t>>>(a, b, c, nb); });
    // alternate forward / reversed launches: what back-to-back leapfrog launches do in the product
    // (the tail the previous launch wrote last is read first)
    {
      int flip = 0;
      timeit("lf3r2w", "piece_alternating", mib, 5.0 * bytes, [&] {
        if (flip ^= 1) k_lf1<false, false, false><<<g, t>>>(a, b, c, nb);
        else k_lf1<true, false, false><<<g, t>>>(a, b, c, nb);
      });
    }
    timeit("lf3r2w", "piece_ntstore", mib, 5.0 * bytes, [&] { k_lf1<false, false, true><<<g, t>>>(a, b, c, nb); });
    timeit("lf3r2w", "piece_ntload_ntstore", mib, 5.0 * bytes, [&] { k_lf1<false, true, true><<<g, t>>>(a, b, c, nb); });
    timeit("lf3r2w", "gridstride_16384", mib, 5.0 * bytes, [&] { k_lf_gs<<<16384, t>>>(a, b, c, n); });
  }
  CK(hipDeviceSynchronize());
  printf("{\"tool\": \"tools/membw2.hip\", \"data\": \"pseudo-random\", \"unit\": \"GB/s of algorithmic bytes\", \"results\": [\n");
  for (size_t i = 0; i < recs.size(); ++i)
    printf("  {\"mix\": \"%s\", \"form\": \"%s\", \"array_MiB\": %zu, \"GBps\": %.0f, \"us_per_launch\": %.1f}%s\n",
           recs[i].mix.c_str(), recs[i].form.c_str(), recs[i].mib, recs[i].gbps, recs[i].us, i + 1 < recs.size() ? "," : "");
  printf("]}\n");
  return 0;
}
