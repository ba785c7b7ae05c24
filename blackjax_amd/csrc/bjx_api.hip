// Library-level entry points: error string, ABI version, host key splitting, launch geometry.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/bjx_hip.h"
#include "bjx_host.h"

static thread_local char g_err[512] = "";

void bjx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

unsigned bjx_row_grid(int64_t n_rows, int waves_per_block) {
  static const int64_t cap = [] {
    const char* e = getenv("BJX_MAX_BLOCKS");
    int64_t v = e ? atoll(e) : 0;
    return v > 0 ? v : (int64_t)256 * 256;  // measured on MI355X: one row per wave (no grid-stride) is 2-3 % faster than a 4096-block cap
  }();
  int64_t blocks = (n_rows + waves_per_block - 1) / waves_per_block;
  if (blocks < 1) blocks = 1;
  if (blocks > cap) blocks = cap;
  return (unsigned)blocks;
}

namespace {
inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
// host copy of threefry2x32 (same published algorithm as bjx_device.h)
void threefry2x32_host(uint32_t k0, uint32_t k1, uint32_t& x0, uint32_t& x1) {
  static const int R[2][4] = {{13, 15, 26, 6}, {17, 29, 16, 24}};
  const uint32_t ks[3] = {k0, k1, k0 ^ k1 ^ 0x1BD11BDAu};
  x0 += ks[0];
  x1 += ks[1];
  for (int i = 0; i < 5; ++i) {
    for (int j = 0; j < 4; ++j) {
      x0 += x1;
      x1 = rotl32(x1, R[i & 1][j]);
      x1 ^= x0;
    }
    x0 += ks[(i + 1) % 3];
    x1 += ks[(i + 2) % 3] + (uint32_t)(i + 1);
  }
}
}  // namespace

extern "C" {

const char* bjx_last_error(void) { return g_err; }
int bjx_abi_version(void) { return BJX_ABI_VERSION; }

int bjx_keys_split(uint32_t key0, uint32_t key1, int64_t n, int64_t offset, uint32_t* out) {
  BJX_CHECK_ARG(n >= 0 && offset >= 0 && (n == 0 || out), "bjx_keys_split: bad arguments");
  for (int64_t i = 0; i < n; ++i) {
    const uint64_t c = (uint64_t)(offset + i);
    uint32_t x0 = (uint32_t)(c >> 32), x1 = (uint32_t)c;
    threefry2x32_host(key0, key1, x0, x1);
    out[2 * i] = x0;
    out[2 * i + 1] = x1;
  }
  return 0;
}

}  // extern "C"
