// Round 4 feasibility probe for a speculative NUTS tail: does a HIP graph on this part run a LONG kernel on a
// forked branch CONCURRENTLY with a dependent chain of short kernels, and what does the chain then cost per link?
// Chain: [c, p] x K per batch (two one-wave-per-row kernels, each reading what the previous one wrote -- the
// callable and a light position-update kernel); branch: one kernel per batch that spins for ~`book_us` (the
// bookkeeping of the previous batch), forked after the batch's first link and joined before the next batch's
// first link but one.  Reported: us per [c, p] pair without the branch, with it, and with the same long kernel
// IN the chain (serial).  hipcc --offload-arch=gfx950 -O3 tools/spec_probe.hip -o /tmp/spec_probe
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

__global__ void __launch_bounds__(64) link(const float* __restrict__ in, float* __restrict__ out, int rows) {
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    const float4 v = reinterpret_cast<const float4*>(in + r * 256)[threadIdx.x];
    reinterpret_cast<float4*>(out + r * 256)[threadIdx.x] = make_float4(v.x * 1.0001f + 1.0f, v.y, v.z, v.w);
  }
}

__global__ void __launch_bounds__(64) spin(const float* __restrict__ in, float* __restrict__ out, int rows,
                                           long long ticks) {  // s_memtime / wall_clock64 runs at 100 MHz
  const long long t0 = wall_clock64();
  float acc = in[blockIdx.x * 64 + threadIdx.x];
  while (wall_clock64() - t0 < ticks) acc = acc * 1.0001f + 1.0f;
  out[blockIdx.x * 64 + threadIdx.x] = acc;
}

static int run(int rows, int K, int batches, double book_us, int mode, const char* name, bool last) {
  // mode 0: chain only; 1: long kernel on a forked branch; 2: long kernel inside the chain
  hipStream_t s, side;
  CK(hipStreamCreate(&s));
  CK(hipStreamCreate(&side));
  float *a, *b, *c, *d;
  CK(hipMalloc(&a, rows * 256 * 4)); CK(hipMalloc(&b, rows * 256 * 4));
  CK(hipMalloc(&c, rows * 256 * 4)); CK(hipMalloc(&d, rows * 256 * 4));
  CK(hipMemset(a, 0, rows * 256 * 4)); CK(hipMemset(c, 0, rows * 256 * 4));
  std::vector<hipEvent_t> ev(4 * batches + 4);
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  const long long ticks = (long long)(book_us * 100.0);
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  int joined_upto = -1;
  for (int n = 0; n < batches; ++n) {
    for (int i = 0; i < K; ++i) {
      if (mode == 1 && i == 1 && n >= 1) {  // join the branch forked in batch n - 1 ... (one batch of slack)
        CK(hipStreamWaitEvent(s, ev[2 * (n - 1) + 1], 0));
        joined_upto = n - 1;
      }
      hipLaunchKernelGGL(link, dim3(rows), dim3(64), 0, s, a, b, rows);
      hipLaunchKernelGGL(link, dim3(rows), dim3(64), 0, s, b, a, rows);
      if (mode == 1 && i == 0) {  // fork: the branch depends on the chain up to here (and on the previous branch kernel)
        CK(hipEventRecord(ev[2 * n], s));
        CK(hipStreamWaitEvent(side, ev[2 * n], 0));
        hipLaunchKernelGGL(spin, dim3(rows), dim3(64), 0, side, c, d, rows, ticks);
        CK(hipEventRecord(ev[2 * n + 1], side));
      }
      if (mode == 2 && i == 0) hipLaunchKernelGGL(spin, dim3(rows), dim3(64), 0, s, c, d, rows, ticks);
    }
  }
  if (mode == 1) {
    for (int n = joined_upto + 1; n < batches; ++n) CK(hipStreamWaitEvent(s, ev[2 * n + 1], 0));
  }
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, s));
  CK(hipStreamSynchronize(s));
  const int reps = 20;
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, s));
  CK(hipStreamSynchronize(s));
  auto t1 = std::chrono::steady_clock::now();
  printf("\"%s\": {\"us_per_pair\": %.2f, \"us_per_batch\": %.1f}%s", name,
         std::chrono::duration<double, std::micro>(t1 - t0).count() / (reps * batches * K),
         std::chrono::duration<double, std::micro>(t1 - t0).count() / (reps * batches), last ? "" : ", ");
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(c)); CK(hipFree(d));
  return 0;
}

// mode 3: the chain of ONE batch is a graph replayed on stream s1 per batch; the long kernel of batch n is a plain
// launch on stream s2 that waits for batch n's first replay ... and batch n + 2's replay waits for it: the host
// issues ~5 API calls per batch (graph launch, 2 event records, 2 stream waits, 1 kernel launch)
static int run_two_streams(int rows, int K, int batches, double book_us, const char* name, bool last) {
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  float *a, *b, *c, *d;
  CK(hipMalloc(&a, rows * 256 * 4)); CK(hipMalloc(&b, rows * 256 * 4));
  CK(hipMalloc(&c, rows * 256 * 4)); CK(hipMalloc(&d, rows * 256 * 4));
  CK(hipMemset(a, 0, rows * 256 * 4)); CK(hipMemset(c, 0, rows * 256 * 4));
  const long long ticks = (long long)(book_us * 100.0);
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < K; ++i) {
    hipLaunchKernelGGL(link, dim3(rows), dim3(64), 0, s1, a, b, rows);
    hipLaunchKernelGGL(link, dim3(rows), dim3(64), 0, s1, b, a, rows);
  }
  CK(hipStreamEndCapture(s1, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  std::vector<hipEvent_t> e1(batches), e2(batches);
  for (int n = 0; n < batches; ++n) {
    CK(hipEventCreateWithFlags(&e1[n], hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&e2[n], hipEventDisableTiming));
  }
  auto once = [&]() -> int {
    for (int n = 0; n < batches; ++n) {
      if (n >= 2) CK(hipStreamWaitEvent(s1, e2[n - 2], 0));  // the bookkeeping of batch n - 2 gates batch n
      CK(hipGraphLaunch(ge, s1));
      CK(hipEventRecord(e1[n], s1));
      CK(hipStreamWaitEvent(s2, e1[n], 0));
      hipLaunchKernelGGL(spin, dim3(rows), dim3(64), 0, s2, c, d, rows, ticks);
      CK(hipEventRecord(e2[n], s2));
    }
    CK(hipStreamSynchronize(s1));
    CK(hipStreamSynchronize(s2));
    return 0;
  };
  if (once()) return 1;
  const int reps = 20;
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < reps; ++i)
    if (once()) return 1;
  auto t1 = std::chrono::steady_clock::now();
  printf("\"%s\": {\"us_per_pair\": %.2f, \"us_per_batch\": %.1f}%s", name,
         std::chrono::duration<double, std::micro>(t1 - t0).count() / (reps * batches * K),
         std::chrono::duration<double, std::micro>(t1 - t0).count() / (reps * batches), last ? "" : ", ");
  return 0;
}

int main() {
  printf("{");
  const int rows_list[3] = {4, 32, 512};
  for (int ri = 0; ri < 3; ++ri) {
    const int rows = rows_list[ri];
    char nm[128];
    for (int K : {8, 16}) {
      const double book = K * 3.6;  // the bookkeeping of K leaves
      snprintf(nm, sizeof nm, "rows%d_K%d_chain_only", rows, K);
      if (run(rows, K, 16, book, 0, nm, false)) return 1;
      snprintf(nm, sizeof nm, "rows%d_K%d_branch_%.0fus", rows, K, book);
      if (run(rows, K, 16, book, 1, nm, false)) return 1;
      snprintf(nm, sizeof nm, "rows%d_K%d_serial_%.0fus", rows, K, book);
      if (run(rows, K, 16, book, 2, nm, false)) return 1;
      snprintf(nm, sizeof nm, "rows%d_K%d_two_streams_%.0fus", rows, K, book);
      if (run_two_streams(rows, K, 16, book, nm, ri == 2 && K == 16)) return 1;
    }
  }
  printf("}\n");
  return 0;
}
