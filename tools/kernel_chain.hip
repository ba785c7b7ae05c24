// What does a DEPENDENT small kernel cost on this part, launched in-stream versus replayed as a graph node?
// A chain of n launches, each one wave that reads a value the previous launch wrote (one dependent global
// round trip + a store), timed as a whole.  hipcc --offload-arch=gfx950 -O3 tools/kernel_chain.hip -o /tmp/kc
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x)                                                                         \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));                  \
      return 1;                                                                       \
    }                                                                                 \
  } while (0)

__global__ void __launch_bounds__(64) step(const float* __restrict__ in, float* __restrict__ out, int rows) {
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    const float v = in[r * 64 + threadIdx.x];
    out[r * 64 + threadIdx.x] = v * 1.0001f + 1.0f;
  }
}

// the product's funnel callable (one wave per row, D = 256) as a chain link: q -> g -> (as q) -> ...
#include "../blackjax_amd/csrc/bjx_targets_dev.h"
template <int VARIANT>
__global__ void __launch_bounds__(64) funnel_link(const float* __restrict__ in, float* __restrict__ out,
                                                  float* __restrict__ lp, int rows) {
  const int lane = threadIdx.x & 63;
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    bjx::F4 x[1];
    x[0] = bjx::ld4(in + r * 256 + lane * 4);
    if (VARIANT == 0) {
      bjx::funnel_row<1>(256, x, lp + r, out + r * 256);
    } else {  // the same traffic without the arithmetic
      bjx::st4(out + r * 256 + lane * 4, bjx::F4{x[0].x * 0.5f, x[0].y * 0.5f, x[0].z * 0.5f, x[0].w * 0.5f});
      if (lane == 0) lp[r] = x[0].x;
    }
  }
}

// a kernel whose EXECUTED code is large (a straight line of ~N dependent FMAs with distinct literals: 8 bytes
// each), for the instruction-cache question: does a small kernel get slower when it alternates with one?
#define F8(x, k) x = x * 1.0001f + (float)(k); x = x * 0.9999f + (float)(k + 1); x = x * 1.0002f + (float)(k + 2); \
  x = x * 0.9998f + (float)(k + 3); x = x * 1.0003f + (float)(k + 4); x = x * 0.9997f + (float)(k + 5);          \
  x = x * 1.0004f + (float)(k + 6); x = x * 0.9996f + (float)(k + 7);
#define F64(x, k) F8(x, k) F8(x, k + 8) F8(x, k + 16) F8(x, k + 24) F8(x, k + 32) F8(x, k + 40) F8(x, k + 48) F8(x, k + 56)
#define F512(x, k) F64(x, k) F64(x, k + 64) F64(x, k + 128) F64(x, k + 192) F64(x, k + 256) F64(x, k + 320) F64(x, k + 384) F64(x, k + 448)
template <int KB16>
__global__ void __launch_bounds__(64) big_code(const float* __restrict__ in, float* __restrict__ out, int rows) {
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    float v = in[r * 256 + threadIdx.x];
    F512(v, 1) F512(v, 600)                     // ~2 048 instructions with literals ~ 16 KB
    if (KB16 >= 2) { F512(v, 1300) F512(v, 1900) }
    if (KB16 >= 4) { F512(v, 2600) F512(v, 3200) F512(v, 3900) F512(v, 4500) }
    out[r * 256 + threadIdx.x] = v;
  }
}

template <typename F>
static int time_graph(hipStream_t s, int len, int total, const char* name, F&& body, bool last) {
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < len; ++i) body(i);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, s));
  CK(hipStreamSynchronize(s));
  const int reps = total / len > 0 ? total / len : 1;
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, s));
  CK(hipStreamSynchronize(s));
  auto t2 = std::chrono::steady_clock::now();
  printf("\"%s\": {\"us_per_kernel\": %.2f}%s", name,
         std::chrono::duration<double, std::micro>(t2 - t0).count() / (reps * len), last ? "}\n" : ", ");
  return 0;
}

int main() {
  const int rows = 32, n = 2000;
  float *a, *b;
  CK(hipMalloc(&a, rows * 64 * sizeof(float)));
  CK(hipMalloc(&b, rows * 64 * sizeof(float)));
  CK(hipMemset(a, 0, rows * 64 * sizeof(float)));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  auto chain = [&](int len) {
    for (int i = 0; i < len; ++i) hipLaunchKernelGGL(step, dim3(rows), dim3(64), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, rows);
  };
  chain(100);
  CK(hipStreamSynchronize(s));
  printf("{");
  for (int rep = 0; rep < 2; ++rep) {
    auto t0 = std::chrono::steady_clock::now();
    chain(n);
    auto t1 = std::chrono::steady_clock::now();
    CK(hipStreamSynchronize(s));
    auto t2 = std::chrono::steady_clock::now();
    printf("\"in_stream_rep%d\": {\"us_per_kernel\": %.2f, \"host_issue_us_per_kernel\": %.2f}, ", rep,
           std::chrono::duration<double, std::micro>(t2 - t0).count() / n,
           std::chrono::duration<double, std::micro>(t1 - t0).count() / n);
  }
  {
    float *fa, *fb, *flp;
    CK(hipMalloc(&fa, rows * 256 * sizeof(float)));
    CK(hipMalloc(&fb, rows * 256 * sizeof(float)));
    CK(hipMalloc(&flp, rows * sizeof(float)));
    CK(hipMemset(fa, 0, rows * 256 * sizeof(float)));
    if (time_graph(s, 512, 4096, "graph_of_512_funnel_callable_rows32", [&](int i) {
          hipLaunchKernelGGL(funnel_link<0>, dim3(rows), dim3(64), 0, s, (i & 1) ? fb : fa, (i & 1) ? fa : fb, flp, rows);
        }, false)) return 1;
    if (time_graph(s, 512, 4096, "graph_of_512_same_traffic_no_arithmetic", [&](int i) {
          hipLaunchKernelGGL(funnel_link<1>, dim3(rows), dim3(64), 0, s, (i & 1) ? fb : fa, (i & 1) ? fa : fb, flp, rows);
        }, false)) return 1;
    if (time_graph(s, 512, 4096, "graph_of_512_funnel_callable_1_row", [&](int i) {
          hipLaunchKernelGGL(funnel_link<0>, dim3(1), dim3(64), 0, s, (i & 1) ? fb : fa, (i & 1) ? fa : fb, flp, 1);
        }, false)) return 1;
  }
  {
    float *fa, *fb, *flp;
    CK(hipMalloc(&fa, rows * 256 * sizeof(float)));
    CK(hipMalloc(&fb, rows * 256 * sizeof(float)));
    CK(hipMalloc(&flp, rows * sizeof(float)));
    CK(hipMemset(fa, 0, rows * 256 * sizeof(float)));
#define BIG_ALONE(K_, NAME_)                                                                                       \
  if (time_graph(s, 512, 4096, NAME_, [&](int i) {                                                                \
        hipLaunchKernelGGL(big_code<K_>, dim3(rows), dim3(64), 0, s, (i & 1) ? fb : fa, (i & 1) ? fa : fb, rows);  \
      }, false)) return 1;
#define BIG_ALT(K_, NAME_)                                                                                         \
  if (time_graph(s, 512, 4096, NAME_, [&](int i) {                                                                \
        if (i & 1) hipLaunchKernelGGL(big_code<K_>, dim3(rows), dim3(64), 0, s, fb, fa, rows);                     \
        else hipLaunchKernelGGL(funnel_link<0>, dim3(rows), dim3(64), 0, s, fa, fb, flp, rows);                    \
      }, false)) return 1;
    BIG_ALONE(1, "big_code_16KB_alone") BIG_ALONE(2, "big_code_32KB_alone") BIG_ALONE(4, "big_code_64KB_alone")
    BIG_ALT(1, "alternating_funnel_and_16KB (per kernel)") BIG_ALT(2, "alternating_funnel_and_32KB (per kernel)")
    BIG_ALT(4, "alternating_funnel_and_64KB (per kernel)")
  }
  for (int len : {64, 512}) {
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    chain(len);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    const int reps = n / len > 0 ? n / len : 1;
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    auto t2 = std::chrono::steady_clock::now();
    printf("\"graph_of_%d\": {\"us_per_kernel\": %.2f}%s", len,
           std::chrono::duration<double, std::micro>(t2 - t0).count() / (reps * len), len == 512 ? "}\n" : ", ");
  }
  return 0;
}
