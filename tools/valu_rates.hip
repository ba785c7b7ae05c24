// Issue cost of the VALU instructions the normal draw is made of (threefry + erf_inv with an fp64 log1p), measured
// per SIMD on gfx950: every wave runs UNROLL x 8 independent chains of ONE instruction, `waves_per_simd` waves per SIMD
// on every CU, wall_clock around the loop.  Output: cycles per wave64 instruction per SIMD (shader clock from
// s_memtime).
//
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rates tools/valu_rates.hip ; run: ./valu_rates
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define CHECK(x)                                                        \
  do {                                                                  \
    hipError_t e_ = (x);                                                \
    if (e_ != hipSuccess) {                                             \
      printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); \
      return 1;                                                         \
    }                                                                   \
  } while (0)

constexpr int ITERS = 2000;

// 8 independent accumulators, each instruction repeated 4 times per iteration: 32 instructions per loop trip
#define BODY32_32(INS)                                                                        \
  for (int it = 0; it < ITERS; ++it) {                                                        \
    asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)                      \
                 INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)                      \
                 INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)                      \
                 INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)                      \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) \
                 : "v"(b), "v"(c) : "vcc", "s10", "s11");                                       \
  }

#define K32(NAME, INS)                                                                                   \
  __global__ void NAME(uint32_t* out, unsigned long long* cyc, uint32_t seed) {                          \
    uint32_t a[8], b = seed + threadIdx.x, c = seed * 3u + 1u;                                           \
    for (int i = 0; i < 8; ++i) a[i] = seed + i + threadIdx.x;                                           \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                          \
    BODY32_32(INS)                                                                                       \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                          \
    uint32_t s = 0;                                                                                      \
    for (int i = 0; i < 8; ++i) s ^= a[i];                                                               \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                      \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                     \
  }

#define K64(NAME, INS)                                                                                   \
  __global__ void NAME(uint32_t* out, unsigned long long* cyc, uint32_t seed) {                          \
    double a[8], b = 1.0 + 1e-9 * (seed + threadIdx.x), c = 1e-12 * seed;                                \
    for (int i = 0; i < 8; ++i) a[i] = 1.0 + 1e-6 * (seed + i + threadIdx.x);                            \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                          \
    BODY32_32(INS)                                                                                       \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                          \
    double s = 0;                                                                                        \
    for (int i = 0; i < 8; ++i) s += a[i];                                                               \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)__double2loint(s);                            \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                     \
  }

#define KF32(NAME, INS)                                                                                  \
  __global__ void NAME(uint32_t* out, unsigned long long* cyc, uint32_t seed) {                          \
    float a[8], b = 1.0f + 1e-7f * (seed + threadIdx.x), c = 1e-9f * seed;                               \
    for (int i = 0; i < 8; ++i) a[i] = 1.0f + 1e-6f * (seed + i + threadIdx.x);                          \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                          \
    BODY32_32(INS)                                                                                       \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                          \
    float s = 0;                                                                                         \
    for (int i = 0; i < 8; ++i) s += a[i];                                                               \
    out[blockIdx.x * blockDim.x + threadIdx.x] = __float_as_uint(s);                                     \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                     \
  }

// 64-bit pair of floats for packed ops
#define KPK(NAME, INS)                                                                                   \
  __global__ void NAME(uint32_t* out, unsigned long long* cyc, uint32_t seed) {                          \
    typedef float f2 __attribute__((ext_vector_type(2)));                                                \
    f2 a[8], b = {1.0f + 1e-7f * (seed + threadIdx.x), 1.0f}, c = {1e-9f * seed, 1e-9f};                 \
    for (int i = 0; i < 8; ++i) a[i] = f2{1.0f + 1e-6f * (seed + i + threadIdx.x), 1.0f};                \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                          \
    BODY32_32(INS)                                                                                       \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                          \
    float s = 0;                                                                                         \
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;                                                    \
    out[blockIdx.x * blockDim.x + threadIdx.x] = __float_as_uint(s);                                     \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                     \
  }

#define I_ADD(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define I_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define I_ALIGN(i) "v_alignbit_b32 %" #i ", %" #i ", %" #i ", 13\n"
#define I_XAD(i) "v_xad_u32 %" #i ", %" #i ", %8, %9\n"
#define I_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define I_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define I_CNDMASK64(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[10:11]\n"
#define I_CMPVCC(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n"
#define I_CMPSG(i) "v_cmp_lt_f32_e64 s[10:11], %" #i ", %8\n"
#define I_CMPCND(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n v_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"
#define I_FMAAK(i) "v_fmaak_f32 %" #i ", %" #i ", %8, 0x3fc00000\n"
#define I_LSHR(i) "v_lshrrev_b32 %" #i ", 9, %" #i "\n"
#define I_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define I_MAX32(i) "v_max_f32 %" #i ", %" #i ", %8\n"
#define I_MOV(i) "v_mov_b32 %" #i ", %8\n"
#define I_FMA32(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define I_FMAC32(i) "v_fmac_f32 %" #i ", %8, %9\n"
#define I_MUL32(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
#define I_RCP32(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define I_SQRT32(i) "v_sqrt_f32 %" #i ", %" #i "\n"
#define I_LOG32(i) "v_log_f32 %" #i ", %" #i "\n"
#define I_EXP32(i) "v_exp_f32 %" #i ", %" #i "\n"
#define I_PKFMA(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define I_PKMUL(i) "v_pk_mul_f32 %" #i ", %" #i ", %8\n"
#define I_PKADD(i) "v_pk_add_f32 %" #i ", %" #i ", %8\n"
#define I_FMA64(i) "v_fma_f64 %" #i ", %" #i ", %8, %9\n"
#define I_ADD64(i) "v_add_f64 %" #i ", %" #i ", %8\n"
#define I_MUL64(i) "v_mul_f64 %" #i ", %" #i ", %8\n"
#define I_RCP64(i) "v_rcp_f64 %" #i ", %" #i "\n"
#define I_RSQ64(i) "v_rsq_f64 %" #i ", %" #i "\n"
#define I_LDEXP64(i) "v_ldexp_f64 %" #i ", %" #i ", 1\n"
#define I_FREXPM64(i) "v_frexp_mant_f64 %" #i ", %" #i "\n"

K32(k_add_u32, I_ADD)
K32(k_xor_b32, I_XOR)
K32(k_alignbit, I_ALIGN)
K32(k_xad_u32, I_XAD)
K32(k_add3_u32, I_ADD3)
K32(k_cndmask, I_CNDMASK)
K32(k_cndmask64, I_CNDMASK64)
KF32(k_cmp_vcc, I_CMPVCC)
KF32(k_cmp_sgpr, I_CMPSG)
KF32(k_cmp_cnd_pair, I_CMPCND)
KF32(k_fmaak_f32, I_FMAAK)
K32(k_lshrrev, I_LSHR)
K32(k_and_b32, I_AND)
KF32(k_max_f32, I_MAX32)
K32(k_mov_b32, I_MOV)
KF32(k_fma_f32, I_FMA32)
KF32(k_fmac_f32, I_FMAC32)
KF32(k_mul_f32, I_MUL32)
KF32(k_rcp_f32, I_RCP32)
KF32(k_sqrt_f32, I_SQRT32)
KF32(k_log_f32, I_LOG32)
KF32(k_exp_f32, I_EXP32)
KPK(k_pk_fma_f32, I_PKFMA)
KPK(k_pk_mul_f32, I_PKMUL)
KPK(k_pk_add_f32, I_PKADD)
K64(k_fma_f64, I_FMA64)
K64(k_add_f64, I_ADD64)
K64(k_mul_f64, I_MUL64)
K64(k_rcp_f64, I_RCP64)
K64(k_rsq_f64, I_RSQ64)
K64(k_ldexp_f64, I_LDEXP64)
K64(k_frexp_mant_f64, I_FREXPM64)

// conversions: separate source / destination register classes
__global__ void k_cvt_f64_f32(uint32_t* out, unsigned long long* cyc, uint32_t seed) {
  float s[8];
  double d[8];
  for (int i = 0; i < 8; ++i) s[i] = 1.0f + 1e-6f * (seed + i + threadIdx.x);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; ++it) {
#define C1(i) "v_cvt_f64_f32 %" #i ", %" #i "+8\n"
    for (int r = 0; r < 4; ++r)
      asm volatile(
          "v_cvt_f64_f32 %0, %8\n v_cvt_f64_f32 %1, %9\n v_cvt_f64_f32 %2, %10\n v_cvt_f64_f32 %3, %11\n"
          "v_cvt_f64_f32 %4, %12\n v_cvt_f64_f32 %5, %13\n v_cvt_f64_f32 %6, %14\n v_cvt_f64_f32 %7, %15\n"
          : "=v"(d[0]), "=v"(d[1]), "=v"(d[2]), "=v"(d[3]), "=v"(d[4]), "=v"(d[5]), "=v"(d[6]), "=v"(d[7])
          : "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]), "v"(s[4]), "v"(s[5]), "v"(s[6]), "v"(s[7]));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double t = 0;
  for (int i = 0; i < 8; ++i) t += d[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)__double2loint(t);
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_cvt_f32_f64(uint32_t* out, unsigned long long* cyc, uint32_t seed) {
  float s[8];
  double d[8];
  for (int i = 0; i < 8; ++i) d[i] = 1.0 + 1e-6 * (seed + i + threadIdx.x);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; ++it) {
    for (int r = 0; r < 4; ++r)
      asm volatile(
          "v_cvt_f32_f64 %0, %8\n v_cvt_f32_f64 %1, %9\n v_cvt_f32_f64 %2, %10\n v_cvt_f32_f64 %3, %11\n"
          "v_cvt_f32_f64 %4, %12\n v_cvt_f32_f64 %5, %13\n v_cvt_f32_f64 %6, %14\n v_cvt_f32_f64 %7, %15\n"
          : "=v"(s[0]), "=v"(s[1]), "=v"(s[2]), "=v"(s[3]), "=v"(s[4]), "=v"(s[5]), "=v"(s[6]), "=v"(s[7])
          : "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]), "v"(d[4]), "v"(d[5]), "v"(d[6]), "v"(d[7]));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float t = 0;
  for (int i = 0; i < 8; ++i) t += s[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = __float_as_uint(t);
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_cvt_f64_i32(uint32_t* out, unsigned long long* cyc, uint32_t seed) {
  int s[8];
  double d[8];
  for (int i = 0; i < 8; ++i) s[i] = seed + i + threadIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; ++it) {
    for (int r = 0; r < 4; ++r)
      asm volatile(
          "v_cvt_f64_i32 %0, %8\n v_cvt_f64_i32 %1, %9\n v_cvt_f64_i32 %2, %10\n v_cvt_f64_i32 %3, %11\n"
          "v_cvt_f64_i32 %4, %12\n v_cvt_f64_i32 %5, %13\n v_cvt_f64_i32 %6, %14\n v_cvt_f64_i32 %7, %15\n"
          : "=v"(d[0]), "=v"(d[1]), "=v"(d[2]), "=v"(d[3]), "=v"(d[4]), "=v"(d[5]), "=v"(d[6]), "=v"(d[7])
          : "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]), "v"(s[4]), "v"(s[5]), "v"(s[6]), "v"(s[7]));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double t = 0;
  for (int i = 0; i < 8; ++i) t += d[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)__double2loint(t);
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

typedef void (*kern_t)(uint32_t*, unsigned long long*, uint32_t);

int main() {
  struct {
    const char* name;
    kern_t k;
  } ks[] = {{"v_add_u32", k_add_u32},       {"v_xor_b32", k_xor_b32},       {"v_alignbit_b32", k_alignbit},
            {"v_xad_u32", k_xad_u32},       {"v_add3_u32", k_add3_u32},     {"v_cndmask_b32", k_cndmask},
            {"v_cndmask_b32_e64(sgpr pair)", k_cndmask64}, {"v_cmp_lt_f32->vcc", k_cmp_vcc},
            {"v_cmp_lt_f32_e64->sgpr", k_cmp_sgpr}, {"v_cmp+v_cndmask(2 insts)", k_cmp_cnd_pair},
            {"v_fmaak_f32", k_fmaak_f32}, {"v_lshrrev_b32", k_lshrrev}, {"v_and_b32", k_and_b32},
            {"v_max_f32", k_max_f32}, {"v_mov_b32", k_mov_b32},
            {"v_fma_f32", k_fma_f32},       {"v_fmac_f32", k_fmac_f32},     {"v_mul_f32", k_mul_f32},
            {"v_rcp_f32", k_rcp_f32},       {"v_sqrt_f32", k_sqrt_f32},     {"v_log_f32", k_log_f32},
            {"v_exp_f32", k_exp_f32},       {"v_pk_fma_f32", k_pk_fma_f32}, {"v_pk_mul_f32", k_pk_mul_f32},
            {"v_pk_add_f32", k_pk_add_f32}, {"v_fma_f64", k_fma_f64},       {"v_add_f64", k_add_f64},
            {"v_mul_f64", k_mul_f64},       {"v_rcp_f64", k_rcp_f64},       {"v_rsq_f64", k_rsq_f64},
            {"v_ldexp_f64", k_ldexp_f64},   {"v_frexp_mant_f64", k_frexp_mant_f64},
            {"v_cvt_f64_f32", k_cvt_f64_f32}, {"v_cvt_f32_f64", k_cvt_f32_f64}, {"v_cvt_f64_i32", k_cvt_f64_i32}};
  const int n_cu = 256;
  uint32_t* out;
  unsigned long long* cyc;
  CHECK(hipMalloc(&out, (size_t)n_cu * 8 * 1024 * 4));
  CHECK(hipMalloc(&cyc, (size_t)n_cu * 8 * 8));
  printf("{\"iters\": %d, \"instructions_per_wave\": %d, \"rates\": {\n", ITERS, ITERS * 32);
  bool first = true;
  for (auto& e : ks) {
    for (int wps : {1, 2, 4}) {  // waves per SIMD: 256-thread blocks = one wave per SIMD; wps blocks per CU
      const int blocks = n_cu * wps;
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, cyc, 1u);  // warm
      CHECK(hipDeviceSynchronize());
      hipEvent_t a, b;
      CHECK(hipEventCreate(&a));
      CHECK(hipEventCreate(&b));
      CHECK(hipEventRecord(a));
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, cyc, 2u);
      CHECK(hipEventRecord(b));
      CHECK(hipDeviceSynchronize());
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, a, b));
      static unsigned long long h[256 * 8];
      CHECK(hipMemcpy(h, cyc, (size_t)blocks * 8, hipMemcpyDeviceToHost));
      double mean = 0;
      for (int i = 0; i < blocks; ++i) mean += (double)h[i];
      mean /= blocks;
      // s_memtime ticks at 100 MHz on this part; wall time is the trustworthy figure:
      // cycles per instruction per SIMD at 2.4 GHz nominal = ms * 2.4e6 / (ITERS*32*wps)
      printf("%s  \"%s@%d\": {\"kernel_us\": %.1f, \"cyc_per_inst_per_simd_at_2.4GHz\": %.2f, \"memtime_ticks\": %.0f}",
             first ? "" : ",\n", e.name, wps, ms * 1e3, ms * 2.4e6 / ((double)ITERS * 32 * wps), mean);
      first = false;
    }
  }
  printf("\n}}\n");
  return 0;
}
