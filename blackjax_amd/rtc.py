"""Run-time compilation of user-written DEVICE code into the engine's kernels (hiprtc + the HIP module API).

The reference fuses the user's log-density into the sampler because both are traced into one XLA program.
The MI355X-native counterpart for the opt-in engine-resident path (``hmc(..., fuse_target=True)``): the user
writes the log-density and its gradient as a HIP device function (``blackjax_amd.targets.DeviceTarget``), and
hiprtc compiles it INTO the trajectory kernel of ``csrc/bjx_traj_dev.h`` -- the very header ``libbjxhip.so`` is
built from -- for gfx950, with the library's floating-point contract (``-ffp-contract=off``).  No hipcc
subprocess, no files: a second or so on first use, cached per source.

Nothing here is a fallback: without a GPU ``compile`` still works (hiprtc cross-compiles), loading a module and
launching need the device."""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
ARCH = "gfx950"

_rtc = None
_hip = None


def _librtc():
    global _rtc
    if _rtc is None:
        _rtc = ctypes.CDLL("libhiprtc.so")
        _rtc.hiprtcGetErrorString.restype = ctypes.c_char_p
    return _rtc


def _libhip():
    global _hip
    if _hip is None:
        _hip = ctypes.CDLL("libamdhip64.so")
        _hip.hipGetErrorString.restype = ctypes.c_char_p
    return _hip


class CompileError(RuntimeError):
    pass


_CODE_CACHE: dict = {}  # source text -> code object (a process-wide cache: the NUTS unit takes seconds)


def compile(source: str, name: str = "bjx_user.hip") -> bytes:  # noqa: A001  (mirrors hiprtcCompileProgram)
    """HIP source -> gfx950 code object.  The engine's device headers are on the include path."""
    hit = _CODE_CACHE.get(source)
    if hit is not None:
        return hit
    code = _compile(source, name)
    _CODE_CACHE[source] = code
    return code


def _compile(source: str, name: str) -> bytes:
    rtc = _librtc()
    prog = ctypes.c_void_p()
    rc = rtc.hiprtcCreateProgram(ctypes.byref(prog), source.encode(), name.encode(), 0, None, None)
    if rc != 0:
        raise CompileError(f"hiprtcCreateProgram: {rtc.hiprtcGetErrorString(rc).decode()}")
    try:
        opts = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-ffp-contract=off", f"-I{CSRC}"]
        arr = (ctypes.c_char_p * len(opts))(*[o.encode() for o in opts])
        rc = rtc.hiprtcCompileProgram(prog, len(opts), arr)
        n = ctypes.c_size_t()
        rtc.hiprtcGetProgramLogSize(prog, ctypes.byref(n))
        log = ctypes.create_string_buffer(max(n.value, 1))
        rtc.hiprtcGetProgramLog(prog, log)
        if rc != 0:
            raise CompileError("hiprtc could not compile the device source:\n" + log.value.decode(errors="replace"))
        rtc.hiprtcGetCodeSize(prog, ctypes.byref(n))
        code = ctypes.create_string_buffer(n.value)
        rtc.hiprtcGetCode(prog, code)
        return code.raw
    finally:
        rtc.hiprtcDestroyProgram(ctypes.byref(prog))


class Module:
    """A loaded code object; ``launch(name, grid, block, stream, *args)`` with ctypes arguments (a ctypes
    Structure is passed by value, as the kernel declares it)."""

    def __init__(self, code: bytes):
        hip = _libhip()
        self._code = ctypes.create_string_buffer(code, len(code))  # must outlive the module
        self._mod = ctypes.c_void_p()
        rc = hip.hipModuleLoadData(ctypes.byref(self._mod), self._code)
        if rc != 0:
            raise RuntimeError(f"hipModuleLoadData: {hip.hipGetErrorString(rc).decode()}")
        self._fns: dict = {}

    def function(self, name: str):
        f = self._fns.get(name)
        if f is None:
            hip = _libhip()
            f = ctypes.c_void_p()
            rc = hip.hipModuleGetFunction(ctypes.byref(f), self._mod, name.encode())
            if rc != 0:
                raise RuntimeError(f"hipModuleGetFunction({name}): {hip.hipGetErrorString(rc).decode()}")
            self._fns[name] = f
        return f

    def launch(self, name: str, grid: int, block: int, stream: int, *args):
        hip = _libhip()
        ptrs = (ctypes.c_void_p * len(args))(*[ctypes.cast(ctypes.pointer(a), ctypes.c_void_p) for a in args])
        rc = hip.hipModuleLaunchKernel(self.function(name), ctypes.c_uint(grid), 1, 1, ctypes.c_uint(block), 1, 1,
                                       0, ctypes.c_void_p(stream), ptrs, None)
        if rc != 0:
            raise RuntimeError(f"hipModuleLaunchKernel({name}): {hip.hipGetErrorString(rc).decode()}")


class TrajArgs(ctypes.Structure):
    """bjx::TrajArgs of csrc/bjx_traj_dev.h, field for field."""

    _fields_ = [("key0", ctypes.c_uint32), ("key1", ctypes.c_uint32),
                ("off", ctypes.c_int64), ("fold", ctypes.c_int64), ("N", ctypes.c_int64), ("D", ctypes.c_int64),
                ("L", ctypes.c_int64), ("eps_s", ctypes.c_float), ("eps_pc", ctypes.c_void_p),
                ("imm", ctypes.c_void_p), ("imm_stride", ctypes.c_int64), ("thr", ctypes.c_float),
                ("params", ctypes.c_void_p), ("q0", ctypes.c_void_p), ("logp0", ctypes.c_void_p),
                ("g0", ctypes.c_void_p), ("p0_out", ctypes.c_void_p), ("q1_out", ctypes.c_void_p),
                ("p_end_out", ctypes.c_void_p), ("logp1_out", ctypes.c_void_p), ("g1_out", ctypes.c_void_p),
                ("q_out", ctypes.c_void_p), ("logp_out", ctypes.c_void_p), ("g_out", ctypes.c_void_p),
                ("acc_rate_out", ctypes.c_void_p), ("energy_out", ctypes.c_void_p),
                ("is_acc_out", ctypes.c_void_p), ("is_div_out", ctypes.c_void_p)]


TARGET_TU = """#include "bjx_traj_dev.h"
using namespace bjx;
// ---- user source
%(source)s
// ---- kernels around it
#define BJX_RTC_KERNELS(NI_)                                                                              \\
  extern "C" __global__ void __launch_bounds__(256) bjx_rtc_traj_##NI_(TrajArgs a) {                      \\
    hmc_trajectory_rows<NI_, %(struct)s>(a);                                                              \\
  }                                                                                                       \\
  extern "C" __global__ void __launch_bounds__(256) bjx_rtc_eval_##NI_(long long N, long long D,          \\
                                                                       const float* params, const float* q, \\
                                                                       float* logp, float* grad) {        \\
    target_rows<NI_, %(struct)s>(N, D, params, q, logp, grad);                                            \\
  }
BJX_RTC_KERNELS(1)
BJX_RTC_KERNELS(2)
BJX_RTC_KERNELS(4)
"""


def ni_for(dim: int) -> int:
    return 1 if dim <= 256 else (2 if dim <= 512 else 4)


# The free-running NUTS multi-tick kernel (csrc/bjx_nuts_tick_dev.h: async_multi_tick_row) around a user target:
# hiprtc compiles the SAME source file the library is built from, device code only, with the user's struct as
# BJX_RTC_USER_TARGET.  One wave per workgroup and row, two waves per SIMD, as k_nuts_async_multi.
NUTS_TU = """#include "bjx_traj_dev.h"
using namespace bjx;
// ---- user source
%(source)s
// ---- the engine's NUTS device code around it
#define BJX_RTC_USER_TARGET %(struct)s
#include "bjx_nuts_tick_dev.h"
#define BJX_RTC_NUTS(NI_, FULL_, NAME_)                                                                      \\
  extern "C" __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2)))                   \\
  NAME_(bjx_nuts_t nt, bjx_nuts_async_t ax, float* qf, float* logp_f, float* gf) {                           \\
    const int64_t n_rows = async_n_rows(ax);                                                                 \\
    for (int64_t b = blockIdx.x; b < n_rows; b += gridDim.x)                                                 \\
      async_multi_tick_row<NI_, FULL_>(nt, ax, qf, logp_f, gf, b, ax.ticks_per_launch);                      \\
  }
BJX_RTC_NUTS(1, false, bjx_rtc_nuts_multi_1)
BJX_RTC_NUTS(1, true, bjx_rtc_nuts_multi_1_full)
BJX_RTC_NUTS(2, false, bjx_rtc_nuts_multi_2)
BJX_RTC_NUTS(2, true, bjx_rtc_nuts_multi_2_full)
"""


def nuts_kernel_name(dim: int) -> str:
    ni = 1 if dim <= 256 else 2
    return f"bjx_rtc_nuts_multi_{ni}" + ("_full" if dim == 256 * ni else "")
