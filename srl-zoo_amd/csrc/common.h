// common.h — shared device/host helpers for libsrlz_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/srlz.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// error.cpp
void srlz_set_error(const char* fmt, ...);
int srlz_hip_fail(hipError_t e, const char* what);

#define SRLZ_HIP(expr)                                   \
  do {                                                   \
    hipError_t _e = (expr);                              \
    if (_e != hipSuccess) return srlz_hip_fail(_e, #expr); \
  } while (0)

// Raise a kernel's dynamic-LDS limit once (per call site / template instantiation), not on every launch: the call is
// host overhead on a launch-bound path, and it is not a stream operation, so it must not happen while the stream is being
// captured into a hipGraph (models/learner.py::_graphStep) — after the warm-up launches every limit is already in place.
// (One host thread drives one GPU per process, so the per-site static needs no lock.)
#define SRLZ_MAX_LDS(fn, bytes)                                                                                         \
  do {                                                                                                                  \
    static int srlz_lds_set_ = -1;                                                                                      \
    if ((int)(bytes) > srlz_lds_set_) {                                                                                 \
      SRLZ_HIP(hipFuncSetAttribute((const void*)(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)));      \
      srlz_lds_set_ = (int)(bytes);                                                                                     \
    }                                                                                                                   \
  } while (0)

#define SRLZ_REQUIRE(cond, code, ...) \
  do {                                \
    if (!(cond)) {                    \
      srlz_set_error(__VA_ARGS__);    \
      return (code);                  \
    }                                 \
  } while (0)

#define SRLZ_LAUNCHED() SRLZ_HIP(hipGetLastError())

static inline hipStream_t as_stream(srlz_stream_t s) { return (hipStream_t)s; }

// Observed (not contractual) dispatch: block b runs on XCD b % 8.  Give every XCD a contiguous run of tiles so
// neighbouring tiles (which share halo rows) hit the same L2.  Bijective for any nb.
__device__ __forceinline__ int xcd_remap(int bid, int nb) {
  const int q = nb >> 3, r = nb & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// a * b + c with a < 2^24 and a wave-uniform 0 <= b < 2^24, as the one full-rate instruction it is (hipcc turns the intrinsic form
// __umul24(a, b) + c into a quarter-rate v_mad_u64_u32 whenever it cannot see the ranges, and plain 32-bit products are quarter-rate
// v_mul_lo_u32): the pixel-offset arithmetic of the stagings, where every issue slot is paid for in matrix time
__device__ __forceinline__ unsigned mad_u24(unsigned a, int b_uniform, unsigned c) {
  unsigned r;
  asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b_uniform), "v"(c));
  return r;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
