// What can a wave do while its SIMD partner streams MFMAs?  (round 6, C5 dense GEMM design question)
// One 512-thread workgroup per CU: waves w and w + 4 share a SIMD (tools/simd_probe.hip).  Wave 0 issues a stream of
// back-to-back fp32 MFMAs on two accumulators (or idles: baseline); wave 4 meanwhile repeats one of four small
// instruction groups and times every repetition with s_memtime:
//   0: 6 x ds_read_b128 + s_waitcnt lgkmcnt(0)      1: 2 x ds_write_b128 + wait
//   2: 4 x global_load_dwordx4 + s_waitcnt vmcnt(0)  3: 32 dependent v_fma_f32
// MFMA kinds: 0 = v_mfma_f32_32x32x2_f32 (64-cycle passes), 1 = v_mfma_f32_16x16x4_f32 (32-cycle), -1 = none.
// Output: median cycles per repetition for every (mfma kind, priority of wave 4, group).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_partner_probe tools/mfma_partner_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int REPS = 24;

template <int KIND, int PRIO, int GROUP, int ALLW>
__global__ void __launch_bounds__(512) k(const float* __restrict__ gsrc, float* __restrict__ sink, uint32_t* __restrict__ times,
                                          uint32_t* __restrict__ mfma_cycles) {
  __shared__ __attribute__((aligned(16))) float lds[10240];  // 40 KiB: two workgroups per CU fit, three do not... (160 KiB / 40)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = (float)i;
  __syncthreads();
  if (wave < ALLW * 3 + 1) {
    if constexpr (KIND >= 0) {
      f32x16 a0, a1;
      f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0}, c2 = {0, 0, 0, 0}, c3 = {0, 0, 0, 0};
      for (int r = 0; r < 16; ++r) { a0[r] = 0.0f; a1[r] = 0.0f; }
      const float x = 1.0f + lane, y = 0.5f;
      const uint64_t t0 = __builtin_amdgcn_s_memtime();
      for (int it = 0; it < 40; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if constexpr (KIND == 0) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
          } else {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c3, 0, 0, 0);
          }
        }
      }
      const uint64_t t1 = __builtin_amdgcn_s_memtime();
      float s = 0;
      for (int r = 0; r < 16; ++r) s += a0[r] + a1[r];
      s += c0[0] + c1[1] + c2[2] + c3[3];
      sink[(blockIdx.x * 8 + wave) * 64 + lane] = s;
      if (lane == 0 && wave == 0) mfma_cycles[blockIdx.x] = (uint32_t)(t1 - t0);
    }
  } else if (wave >= 4 && wave < 4 + ALLW * 3 + 1) {
    __builtin_amdgcn_s_sleep(20);  // let the partner's stream get going
    if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
    float acc = 0.0f;
    for (int rep = 0; rep < REPS; ++rep) {
      const uint64_t t0 = __builtin_amdgcn_s_memtime();
      if constexpr (GROUP == 0) {
        f32x4 v[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) v[i] = *reinterpret_cast<const f32x4*>(lds + ((lane * 20 + i * 1280 + rep * 4) & 8188 & ~3));
#pragma unroll
        for (int i = 0; i < 6; ++i) acc += v[i][0] + v[i][3];
      } else if constexpr (GROUP == 1) {
        *reinterpret_cast<f32x4*>(lds + ((lane * 20 + rep * 4) & 8188 & ~3)) = f32x4{acc, 1, 2, 3};
        *reinterpret_cast<f32x4*>(lds + ((lane * 20 + 4096 + rep * 4) & 8188 & ~3)) = f32x4{acc, 1, 2, 3};
        __builtin_amdgcn_s_waitcnt(0);
      } else if constexpr (GROUP == 2) {
        f32x4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const f32x4*>(gsrc + ((size_t)blockIdx.x * 4096 + (rep * 4 + i) * 256 + lane * 4));
#pragma unroll
        for (int i = 0; i < 4; ++i) acc += v[i][0];
      } else {
#pragma unroll
        for (int i = 0; i < 32; ++i) acc = __builtin_fmaf(acc, 1.0001f, 0.5f);
      }
      asm volatile("" :: "v"(acc));
      __builtin_amdgcn_s_waitcnt(0);
      const uint64_t t1 = __builtin_amdgcn_s_memtime();
      if (lane == 0 && wave == 4) times[blockIdx.x * REPS + rep] = (uint32_t)(t1 - t0);
    }
    sink[(blockIdx.x * 8 + wave) * 64 + lane + 262144] = acc;
  }
}

template <int KIND, int PRIO, int GROUP, int ALLW, int BLOCKS>
void run(const char* label, const float* gsrc, float* sink, uint32_t* times, uint32_t* mc) {
  const int blocks = BLOCKS;
  hipMemset(mc, 0, blocks * 4);
  hipLaunchKernelGGL((k<KIND, PRIO, GROUP, ALLW>), dim3(blocks), dim3(512), 0, 0, gsrc, sink, times, mc);
  hipDeviceSynchronize();
  std::vector<uint32_t> h(blocks * REPS), hm(blocks);
  hipMemcpy(h.data(), times, h.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(hm.data(), mc, hm.size() * 4, hipMemcpyDeviceToHost);
  std::vector<uint32_t> v;
  for (int b = 0; b < blocks; ++b)
    for (int r = 2; r < REPS - 2; ++r) v.push_back(h[b * REPS + r]);
  std::sort(v.begin(), v.end());
  std::sort(hm.begin(), hm.end());
  printf("  {\"case\": \"%s\", \"rep_cycles_p10\": %u, \"median\": %u, \"p90\": %u, \"mfma_stream_cycles_median\": %u},\n", label,
         v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10], hm[hm.size() / 2]);
}

int main() {
  float *gsrc, *sink;
  uint32_t *times, *mc;
  hipMalloc(&gsrc, 1024 * 4096 * 4 * 2);
  hipMemset(gsrc, 0, 1024 * 4096 * 4 * 2);
  hipMalloc(&sink, 1048576 * 4);
  hipMalloc(&times, 1024 * REPS * 4);
  hipMalloc(&mc, 1024 * 4);
  printf("{\"note\": \"wave 4 repeats a group while wave 0 (same SIMD) streams 640 MFMAs (32x32x2: 64 cycles each = 40960; 16x16x4: 1280 x 32 = 40960)\", \"cases\": [\n");
#define ALLG(K, P, W, B, L)                                                      \
  run<K, P, 0, W, B>(L " | 6 ds_read_b128", gsrc, sink, times, mc);              \
  run<K, P, 1, W, B>(L " | 2 ds_write_b128", gsrc, sink, times, mc);             \
  run<K, P, 2, W, B>(L " | 4 global_load_dwordx4", gsrc, sink, times, mc);
  ALLG(-1, 0, 0, 256, "no MFMA, waves 0+4 only, 1 WG/CU")
  ALLG(0, 0, 0, 256, "32x32x2, waves 0+4 only, 1 WG/CU")
  ALLG(0, 0, 1, 256, "32x32x2, waves 0-3 MFMA + 4-7 group, 1 WG/CU")
  ALLG(0, 0, 1, 512, "32x32x2, waves 0-3 MFMA + 4-7 group, 2 WG/CU")
  ALLG(-1, 0, 1, 512, "no MFMA, waves 4-7 group, 2 WG/CU")
  ALLG(0, 3, 1, 512, "32x32x2, all waves, 2 WG/CU, prio 3")
  printf("  {}]}\n");
  return 0;
}
