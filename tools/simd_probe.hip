// Which SIMD does wave w of a 512-thread workgroup run on?  (HW_REG_HW_ID bits [5:4]; gfx950.)  Prints, for the first
// workgroups, the SIMD id of waves 0..7 and a histogram of "waves w and w + 4 share a SIMD".
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/simd_probe tools/simd_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__global__ void __launch_bounds__(512) k(uint32_t* out, int spin) {
  __shared__ float sink[40960 / 4];  // the dense kernel's LDS footprint: two workgroups per CU
  const int wave = threadIdx.x >> 6;
  float x = threadIdx.x;
  for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;  // keep the workgroup resident while the others arrive
  sink[threadIdx.x] = x;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
  if (sink[(threadIdx.x * 7) % 512] == 123.456f) out[0] = 0;
}
int main() {
  const int blocks = 512;
  uint32_t* d;
  hipMalloc(&d, blocks * 8 * 4);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, d, 20000);
  hipDeviceSynchronize();
  static uint32_t h[512 * 8];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int share4 = 0, share1 = 0, share2 = 0, total = 0;
  for (int b = 0; b < blocks; ++b) {
    int simd[8];
    for (int w = 0; w < 8; ++w) simd[w] = (h[b * 8 + w] >> 4) & 3;
    if (b < 12) {
      printf("block %3d cu %2u se %u: simd of waves 0..7 =", b, (h[b * 8] >> 8) & 15, (h[b * 8] >> 13) & 7);
      for (int w = 0; w < 8; ++w) printf(" %d", simd[w]);
      printf("\n");
    }
    for (int w = 0; w < 4; ++w) { share4 += simd[w] == simd[w + 4]; ++total; }
    for (int w = 0; w < 8; w += 2) share1 += simd[w] == simd[w + 1];
    for (int w = 0; w < 8; ++w) if ((w & 2) == 0) share2 += simd[w] == simd[w + 2];
  }
  printf("{\"pairs\": %d, \"w_and_w+4_share_a_simd\": %d, \"w_and_w+1\": %d, \"w_and_w+2\": %d}\n", total, share4, share1, share2);
  return 0;
}
