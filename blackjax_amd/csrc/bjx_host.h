// Host-side helpers for the C-ABI entry points (error reporting, launch geometry).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

// thread-local last-error message (defined in bjx_api.hip)
void bjx_set_error(const char* fmt, ...);

#define BJX_CHECK_ARG(cond, msg) \
  do {                           \
    if (!(cond)) {               \
      bjx_set_error("%s", msg);  \
      return 1;                  \
    }                            \
  } while (0)

static inline int bjx_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    bjx_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return 2;
  }
  return 0;
}

// Row-per-wave kernels: one row per wave (>> 256 CUs worth of workgroups); grid-stride only
// beyond 65536 workgroups (BJX_MAX_BLOCKS overrides).
unsigned bjx_row_grid(int64_t n_rows, int waves_per_block);

// 16-byte vector path is legal when D % 4 == 0 and every non-null pointer is 16-B aligned.
static inline bool bjx_vec4_ptr_ok(const void* p) { return p == nullptr || ((uintptr_t)p & 15u) == 0; }
template <typename... P>
static inline bool bjx_vec4_ok(int64_t D, P... ptrs) {
  return (D % 4 == 0) && (bjx_vec4_ptr_ok((const void*)ptrs) && ...);
}
