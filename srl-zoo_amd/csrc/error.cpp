// error.cpp — thread-local error text + misc host entry points of the C ABI.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void srlz_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int srlz_hip_fail(hipError_t e, const char* what) {
  srlz_set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
  return SRLZ_ERR_HIP;
}

extern "C" int srlz_version(void) { return SRLZ_ABI_VERSION; }

extern "C" const char* srlz_last_error(void) { return g_err; }

extern "C" int srlz_device_cus(void) {
  // queried once per process (one process drives one GPU): this sits on every persistent kernel's launch path
  static int cached = 0;
  if (cached > 0) return cached;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) return 256;
  cached = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
  return cached;
}

// ---- calibration micro-benchmark: back-to-back v_mfma_f32_32x32x2_f32 on random operands, no memory traffic ----
// Used by tools/kbench.py to measure the fp32-MFMA rate this chip actually sustains (clocks under load) so that
// kernel efficiencies can be read against both the datasheet peak and the sustained one.
__global__ __launch_bounds__(256) void mfma_peak_kernel(float* out, int iters, float seed) {
  const int lane = threadIdx.x;
  f32x16 a0, a1, a2, a3;
  for (int r = 0; r < 16; ++r) { a0[r] = seed * (lane + r); a1[r] = seed * (lane - r); a2[r] = seed * r; a3[r] = -seed * r; }
  float x = 0.37f + 0.001f * lane * seed, y = -0.41f + 0.002f * lane * seed;
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// Launches `blocks` workgroups of 256 threads, each wave issuing 4*iters MFMAs (4096 FLOP each).  out: blocks*256 floats.
extern "C" int srlz_debug_mfma_peak(float* out, int blocks, int iters, srlz_stream_t stream) {
  if (!out || blocks <= 0 || iters <= 0) { srlz_set_error("mfma_peak: bad arguments"); return SRLZ_ERR_BAD_DESC; }
  hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters, 1e-3f);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return srlz_hip_fail(e, "mfma_peak launch");
  return 0;
}

// ---- calibration: do VALU instructions hide behind fp32 MFMAs?  (tools/kbench.py "mfma valu") ----
// KV independent VALU instructions after every MFMA, in the same wave (SPLIT = false), or — SPLIT — waves 0-3 of a 512-thread
// workgroup issue only the MFMAs and waves 4-7 (the second wave of each SIMD) only the VALU instructions of the same count.
// KIND 0: v_fma_f32, 1: v_add_u32, 2: v_pk_fma_f32, 3: v_mov_b32.  asm volatile keeps count and place.
template <int KV, int KIND, bool SPLIT>
__global__ __launch_bounds__(SPLIT ? 512 : 256) void mfma_valu_kernel(float* out, int iters, float seed) {
  const int lane = threadIdx.x & 255;
  const bool do_mfma = !SPLIT || threadIdx.x < 256, do_valu = !SPLIT || threadIdx.x >= 256;
  f32x16 a0, a1, a2, a3;
  for (int r = 0; r < 16; ++r) { a0[r] = seed * (lane + r); a1[r] = seed * (lane - r); a2[r] = seed * r; a3[r] = -seed * r; }
  float x = 0.37f + 0.001f * lane * seed, y = -0.41f + 0.002f * lane * seed;
  float t[8];
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 p[8], px = {x, y}, py = {y, x};
  unsigned u[8];
  for (int i = 0; i < 8; ++i) { t[i] = seed * i; u[i] = lane + i; p[i] = f32x2{seed * i, seed}; }
  auto valu = [&](int base) {
#pragma unroll
    for (int k = 0; k < KV; ++k) {
      const int i = (base + k) & 7;
      if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(t[i]) : "v"(x), "v"(y));
      else if (KIND == 1) asm volatile("v_add_u32 %0, %1, %0" : "+v"(u[i]) : "v"(lane));
      else if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(px), "v"(py));
      else asm volatile("v_mov_b32 %0, %1" : "+v"(u[i]) : "v"(lane));
    }
  };
  for (int it = 0; it < iters; ++it) {
    if (do_mfma) a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
    if (do_valu) valu(0);
    if (do_mfma) a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
    if (do_valu) valu(KV);
    if (do_mfma) a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
    if (do_valu) valu(2 * KV);
    if (do_mfma) a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
    if (do_valu) valu(3 * KV);
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
  for (int i = 0; i < 8; ++i) s += t[i] + (float)u[i] + p[i][0] + p[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// blocks workgroups; every MFMA wave issues 4*iters MFMAs, each followed (same wave, or the SIMD's other wave when split) by
// valu_per_mfma instructions of `kind`.  valu_per_mfma in {0, 1, 2, 4, 8, 16}; out: blocks * 512 floats.
extern "C" int srlz_debug_mfma_valu(float* out, int blocks, int iters, int valu_per_mfma, int kind, int split, srlz_stream_t stream) {
  if (!out || blocks <= 0 || iters <= 0) { srlz_set_error("mfma_valu: bad arguments"); return SRLZ_ERR_BAD_DESC; }
  hipStream_t st = (hipStream_t)stream;
#define SRLZ_MV(KVV, KINDV)                                                                                        \
  if (valu_per_mfma == KVV && kind == KINDV) {                                                                     \
    if (split) hipLaunchKernelGGL((mfma_valu_kernel<KVV, KINDV, true>), dim3(blocks), dim3(512), 0, st, out, iters, 1e-3f); \
    else hipLaunchKernelGGL((mfma_valu_kernel<KVV, KINDV, false>), dim3(blocks), dim3(256), 0, st, out, iters, 1e-3f);      \
    launched = true;                                                                                               \
  }
#define SRLZ_MVK(KVV) SRLZ_MV(KVV, 0) SRLZ_MV(KVV, 1) SRLZ_MV(KVV, 2) SRLZ_MV(KVV, 3)
  bool launched = false;
  SRLZ_MVK(0) SRLZ_MVK(1) SRLZ_MVK(2) SRLZ_MVK(4) SRLZ_MVK(8) SRLZ_MVK(16)
#undef SRLZ_MVK
#undef SRLZ_MV
  if (!launched) { srlz_set_error("mfma_valu: valu_per_mfma must be 0, 1, 2, 4, 8 or 16 and kind 0..3"); return SRLZ_ERR_BAD_DESC; }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return srlz_hip_fail(e, "mfma_valu launch");
  return 0;
}

// ---- placement probe: which XCD / SE / CU does workgroup b land on, and when?  (tools/placement.py) ----
// out[b] = {xcc_id, hw_id, start clock (s_memtime low 32 bits), end clock}; every workgroup holds `lds_bytes` of LDS and
// spins for `spin` clock ticks so that later workgroups must wait for a free slot.
__global__ __launch_bounds__(256) void placement_kernel(unsigned* out, int spin) {
  extern __shared__ unsigned char smem[];
  if (threadIdx.x == 0) {
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    const unsigned long long t0 = __builtin_readcyclecounter();
    while ((long long)(__builtin_readcyclecounter() - t0) < spin) { smem[0] = (unsigned char)spin; }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[4 * blockIdx.x + 0] = xcc;
    out[4 * blockIdx.x + 1] = hwid;
    out[4 * blockIdx.x + 2] = (unsigned)t0;
    out[4 * blockIdx.x + 3] = (unsigned)t1;
  }
}

extern "C" int srlz_debug_placement(unsigned* out, int blocks, int lds_bytes, int spin, srlz_stream_t stream) {
  if (!out || blocks <= 0) { srlz_set_error("placement: bad arguments"); return SRLZ_ERR_BAD_DESC; }
  hipError_t e = hipFuncSetAttribute((const void*)placement_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (e != hipSuccess) return srlz_hip_fail(e, "placement attribute");
  hipLaunchKernelGGL(placement_kernel, dim3(blocks), dim3(256), lds_bytes, (hipStream_t)stream, out, spin);
  e = hipGetLastError();
  if (e != hipSuccess) return srlz_hip_fail(e, "placement launch");
  return 0;
}
