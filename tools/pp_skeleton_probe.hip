// Round 6: the ping-pong K-tile loop of the C5 dense GEMM reduced to its skeleton -- two waves per SIMD taking turns
// at 16 fp32 MFMAs (operands from LDS fragments, two accumulators) with a workgroup barrier between the phases -- to
// find what keeps the real loop at ~5 600 cycles per tile where 2 048 (one workgroup per CU) / 4 096 (two) is the floor.
// Variants add the real loop's ingredients one at a time:
//   V0 MFMA + fragment reads + barriers        V1 + LDS tile writes by the non-computing half
//   V2 + global loads feeding those writes (one tile ahead)      V3 = V2 with s_waitcnt vmcnt(0) before the LDS writes
// Output: cycles per K-tile (median over workgroups), 1 and 2 workgroups per CU.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/pp_skeleton_probe tools/pp_skeleton_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int LDK = 20, TILES = 64;

template <int V>
__global__ void __launch_bounds__(512, 4) k(const float* __restrict__ src, float* __restrict__ sink, uint32_t* __restrict__ cyc) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 128 * LDK * 2];  // [A | B] x 2 stages = 40 KiB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = __builtin_amdgcn_readfirstlane(wave >> 2), wn = wave & 3;
  for (int i = tid; i < 2 * 128 * LDK * 2; i += 512) lds[i] = 1.0f + (i & 7);
  __syncthreads();
  const int lm = lane & 31, lk = lane >> 5;
  const float* As0 = lds;
  const float* Bs0 = lds + 2 * 128 * LDK;
  const int a_off = (wm * 64 + lm) * LDK + lk * 8, b_off = (wn * 32 + lm) * LDK + lk * 8;
  const int ytid = tid & 255, s_row = ytid >> 2, s_k = (ytid & 3) * 4;
  float* wr = lds + (wm ? 0 : 2 * 128 * LDK) + s_row * LDK + s_k;
  const float* g = src + ((size_t)blockIdx.x * 128 + s_row) * 1024 + s_k;
  f32x16 acc0, acc1;
  for (int r = 0; r < 16; ++r) { acc0[r] = 0; acc1[r] = 0; }
  float fa0[8], fa1[8], fb[8];
  f32x4 r0 = {1, 2, 3, 4}, r1 = {1, 2, 3, 4};
  auto read_frags = [&](int buf) {
    const float* as = As0 + buf * 128 * LDK + a_off;
    const float* bs = Bs0 + buf * 128 * LDK + b_off;
    *reinterpret_cast<f32x4*>(fa0) = *reinterpret_cast<const f32x4*>(as);
    *reinterpret_cast<f32x4*>(fb) = *reinterpret_cast<const f32x4*>(bs);
    *reinterpret_cast<f32x4*>(fa1) = *reinterpret_cast<const f32x4*>(as + 32 * LDK);
    *reinterpret_cast<f32x4*>(fa0 + 4) = *reinterpret_cast<const f32x4*>(as + 4);
    *reinterpret_cast<f32x4*>(fb + 4) = *reinterpret_cast<const f32x4*>(bs + 4);
    *reinterpret_cast<f32x4*>(fa1 + 4) = *reinterpret_cast<const f32x4*>(as + 32 * LDK + 4);
  };
  auto mfma16 = [&]() {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[u], fb[u], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[u], fb[u], acc1, 0, 0, 0);
    }
  };
  auto stage = [&](int buf, int t) {
    if constexpr (V >= 1) {
      if constexpr (V == 3) __builtin_amdgcn_s_waitcnt(0x0F70 & 0xFFFF);  // placeholder: replaced below by inline asm
      *reinterpret_cast<f32x4*>(wr + buf * 128 * LDK) = r0;
      *reinterpret_cast<f32x4*>(wr + buf * 128 * LDK + 64 * LDK) = r1;
    }
    if constexpr (V >= 2) {
      r0 = *reinterpret_cast<const f32x4*>(g + (t & 31) * 16);
      r1 = *reinterpret_cast<const f32x4*>(g + 64 * 1024 + (t & 31) * 16);
    }
  };
  auto mfma8 = [&](int u0) {
#pragma unroll
    for (int u = u0; u < u0 + 4; ++u) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[u], fb[u], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[u], fb[u], acc1, 0, 0, 0);
    }
  };
  auto stage_all = [&](int buf, int t) {  // lockstep: every thread stages one 16-byte piece of A and of B
    float* w2 = lds + (tid >> 2) * LDK + (tid & 3) * 4 + buf * 128 * LDK;
    *reinterpret_cast<f32x4*>(w2) = r0;
    *reinterpret_cast<f32x4*>(w2 + 2 * 128 * LDK) = r1;
    r0 = *reinterpret_cast<const f32x4*>(g + (t & 31) * 16);
    r1 = *reinterpret_cast<const f32x4*>(g + 64 * 1024 + (t & 31) * 16);
  };
  if (V < 10 && !wm) read_frags(0);
  __syncthreads();
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  if constexpr (V < 10) {
    for (int t = 0; t < TILES; ++t) {
      const int buf = t & 1;
      if (!wm) mfma16();
      else { read_frags(buf); stage(buf ^ 1, t); }
      __syncthreads();
      if (wm) mfma16();
      else { read_frags(buf ^ 1); stage(buf, t); }
      __syncthreads();
    }
  } else if constexpr (V == 10) {  // the lockstep loop of k_dense_gemm_tn8: 16 MFMAs per wave and barrier
    for (int t = 0; t < TILES; ++t) {
      const int buf = t & 1;
      read_frags(buf);
      mfma8(0);
      __builtin_amdgcn_sched_barrier(0);
      stage_all(buf ^ 1, t);
      __builtin_amdgcn_sched_barrier(0);
      mfma8(4);
      __syncthreads();
    }
  } else {  // 32 MFMAs per wave and barrier (a 32-wide K-tile as two 16-wide halves; same LDS bytes per MFMA)
    for (int t = 0; t < TILES; t += 2) {
      const int buf = (t >> 1) & 1;
      read_frags(buf);
      mfma8(0);
      __builtin_amdgcn_sched_barrier(0);
      stage_all(buf ^ 1, t);
      __builtin_amdgcn_sched_barrier(0);
      mfma8(4);
      read_frags(buf);  // second half of the wide tile (same stage: the skeleton only needs the traffic)
      mfma8(0);
      __builtin_amdgcn_sched_barrier(0);
      stage_all(buf ^ 1, t + 1);
      __builtin_amdgcn_sched_barrier(0);
      mfma8(4);
      __syncthreads();
    }
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
  sink[(size_t)blockIdx.x * 512 + tid] = s + r0[0] + r1[1];
  if (tid == 0) cyc[blockIdx.x] = (uint32_t)(t1 - t0);
}

template <int V>
void run(const char* label, int blocks, const float* src, float* sink, uint32_t* cyc) {
  hipLaunchKernelGGL((k<V>), dim3(blocks), dim3(512), 0, 0, src, sink, cyc);
  hipDeviceSynchronize();
  std::vector<uint32_t> h(blocks);
  hipMemcpy(h.data(), cyc, blocks * 4, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  printf("  {\"variant\": \"%s\", \"workgroups\": %d, \"cycles_per_tile_p10\": %.0f, \"median\": %.0f, \"p90\": %.0f},\n", label, blocks,
         h[blocks / 10] / (double)TILES, h[blocks / 2] / (double)TILES, h[blocks * 9 / 10] / (double)TILES);
}

int main() {
  float *src, *sink;
  uint32_t* cyc;
  hipMalloc(&src, (size_t)512 * 128 * 1024 * 4);
  hipMemset(src, 0, (size_t)512 * 128 * 1024 * 4);
  hipMalloc(&sink, 512 * 512 * 4);
  hipMalloc(&cyc, 512 * 4);
  printf("{\"floor_cycles_per_tile\": {\"1 WG per CU\": 2048, \"2 WG per CU\": 4096}, \"cases\": [\n");
  for (int blocks : {256, 512}) {
    run<0>("V0 mfma + fragment reads + barriers", blocks, src, sink, cyc);
    run<1>("V1 + LDS tile writes", blocks, src, sink, cyc);
    run<2>("V2 + global loads one tile ahead", blocks, src, sink, cyc);
    run<10>("V10 lockstep: 16 MFMAs per wave and barrier (cycles per 16-wide tile)", blocks, src, sink, cyc);
    run<11>("V11 lockstep: 32 MFMAs per wave and barrier (cycles per 16-wide tile)", blocks, src, sink, cyc);
  }
  printf("  {}]}\n");
  return 0;
}
