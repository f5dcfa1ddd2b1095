// wino.hip — the 3x3 stride-1 pad-1 64 -> 64 convolution of the encoder's second block (conv3x3(64, 64), /root/reference/models/models.py:54,
// 217-226) as Winograd F(2x2, 3x3) on the fp32 matrix cores (gfx950 only).
//
// Why: on this chip the fp32 MFMA rate IS the fp32 vector rate — the direct implicit GEMM (conv64_fwd_kernel) sits at
// 0.74 of a peak that cannot be raised, so the only way to make this layer faster is to multiply less.  F(2x2, 3x3) computes a 2 x 2
// output patch from a 4 x 4 input patch with 16 multiplications per (ci, co) instead of 36:
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A           (Lavin & Gray 2015; the form cuDNN — the reference's backend — runs for this layer)
// i.e. 16 independent [co 64] x [ci 64] x [patches] GEMMs (one per component (xi, nu) of the 4 x 4 transform domain) whose results are
// folded 16 -> 4 per patch.  The transforms are additions only (B, A have entries 0, +-1): ~3 % of the multiply-adds they replace.
//
// Numerics: every product and every accumulation is fp32 as before; the transforms add two roundings per operand, the accumulation chains
// are 2.25x shorter — against fp64 the outputs are CLOSER than the direct fp32 chain's (4.5e-7 against 1.2e-6 of the output scale at
// N = 512; tests/test_wino_gpu.py holds every element to 2e-5 and prints both); deterministic and position-independent (the arithmetic of
// a patch does not depend on which tile or launch it falls into), so batching, BatchNorm groups and batch size leave every bit unchanged.
//
// Forward / data gradient — conv64_wino_kernel<PSUM, FUSE> (256 threads, 64 KB of LDS: TWO workgroups per CU, persistent over
// XCD-contiguous tile runs; the other workgroup's matrix work covers this one's barriers, landings and epilogue):
//   tile  = 32 consecutive patches of a BatchNorm group's patch grid (n, a, b) — 128 outputs x 64 channels;
//   chunk = 16 input channels (a quarter of K).  During the matrix work of chunk c every thread transforms the raw 4 x 4 x 2-channel patch
//           it loaded during chunk c - 1 (zero padding = out-of-range buffer loads; FUSE: relu(batchnorm(raw)) applied as it lands) into
//           V[16 comps][32 patches][16 ci] of the OTHER LDS buffer (32 KB each, ONE barrier per chunk) and requests chunk c + 2;
//   matrix work: wave w owns output channels 16 w .. of all 32 patches (two 16 x 16 tiles per component: 128 accumulator registers for the
//           16 components), v_mfma_f32_16x16x4_f32; the transformed weights U never touch LDS — 1 KB per wave and component straight
//           from L2 into the MFMA's A operand, four components ahead; V as ds_read_b128 one component ahead (one read feeds four MFMAs);
//   epilogue: 16 -> 4 fold in registers, bias, 16-byte NHWC stores (a lane holds 4 consecutive channels of one patch), BatchNorm partial
//           sums (sum y, sum y^2) of the tile as one record; PSUM (conv2's data gradient = d pooled1): the pooled block's two
//           BatchNorm-BACKWARD sums instead (WinoPoolSum).
// Weight gradient — conv64_wino_wgrad_kernel: the transposed algorithm (below).
// LDS layouts are conflict-free against the hardware's lane groups: ds_read_b128 is served as {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31},
// ... (not 16 consecutive lanes), ds_write_b64 in runs of 16 lanes over 32 banks (MI355X_MICROARCH.md, LDS).
#include "common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int WN_THREADS = 256;
constexpr int WN_TP = 32;                   // patches per tile
constexpr int WN_CHUNK = 16 * 4 * 64 * 4;   // floats of one 16-channel chunk of the weights: U [comp][g][co 64][4]
constexpr int WN_VCHUNK = 16 * 4 * WN_TP * 4;  // ... and of a tile's transformed patches: V [comp][g][patch 32][4] (32 KB)
constexpr int WN_VCOMP = 4 * WN_TP * 4;      // floats of one component of V
constexpr unsigned WN_DROP = 0x80000000u;   // out of range for every buffer here (groups stay below 2 GB), and + any scalar offset does not wrap

struct WinoProg {
  int N, G;          // images per BatchNorm group, groups
  int H, W;          // spatial size (input = output), both even
  int PA, PB;        // patch grid: H / 2, W / 2
  int ppi, ppg, tpg; // patches per image / per group, tiles per group
  unsigned mPPI, mPB;
  int sPPI, sPB;
  long long gstride; // floats of one group's tensor
};

void wn_fastdiv_init(unsigned d, unsigned* m, int* sh) {  // (conv64.hip: Granlund-Montgomery for 31-bit dividends, d >= 2)
  int l = 0;
  while ((1u << l) < d) ++l;
  if (l == 0) l = 1;
  *m = (unsigned)((((unsigned long long)1 << (31 + l)) / d) + 1);
  *sh = l - 1;
}
__device__ __forceinline__ int wn_div(int q, unsigned m, int sh) { return (int)(__umulhi((unsigned)q, m) >> sh); }

__device__ __forceinline__ __amdgpu_buffer_rsrc_t wn_buffer(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

// ---------------------------------------------------------------------------------------------------------------
// U = G g G^T per (co, ci), G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], in the kernel's layout
//   upack[chunk p][comp = 4 xi + nu][g][m][e]   with contraction channel k = 16 p + 4 g + e,
// forward: m = co, k = ci, g[ky][kx] = w[co][ci][ky][kx]; data gradient: m = ci, k = co, g[ky][kx] = w[co][ci][2 - ky][2 - kx].
// ---------------------------------------------------------------------------------------------------------------
__global__ void conv64_wino_pack_kernel(const float* __restrict__ w_ref, float* __restrict__ uf, float* __restrict__ ub) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;  // (p, g, m, e) -> all 16 comps
  if (id >= 64 * 64) return;
  const int e = id & 3, m = (id >> 2) & 63, gq = (id >> 8) & 3, p = id >> 10;
  const int k = 16 * p + 4 * gq + e;
#pragma unroll
  for (int dir = 0; dir < 2; ++dir) {
    float* out = dir ? ub : uf;
    if (!out) continue;
    float gk[3][3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
        gk[ky][kx] = dir ? w_ref[((k * 64 + m) * 3 + (2 - ky)) * 3 + (2 - kx)] : w_ref[((m * 64 + k) * 3 + ky) * 3 + kx];
    float t[3][4];  // g G^T
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      t[ky][0] = gk[ky][0];
      t[ky][1] = 0.5f * (gk[ky][0] + gk[ky][1] + gk[ky][2]);
      t[ky][2] = 0.5f * (gk[ky][0] - gk[ky][1] + gk[ky][2]);
      t[ky][3] = gk[ky][2];
    }
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) {
      const float u0 = t[0][nu];
      const float u1 = 0.5f * (t[0][nu] + t[1][nu] + t[2][nu]);
      const float u2 = 0.5f * (t[0][nu] - t[1][nu] + t[2][nu]);
      const float u3 = t[2][nu];
      const float u[4] = {u0, u1, u2, u3};
#pragma unroll
      for (int xi = 0; xi < 4; ++xi) out[(size_t)p * WN_CHUNK + (((xi * 4 + nu) * 4 + gq) * 64 + m) * 4 + e] = u[xi];
    }
  }
}

// the raw 4 x 4 patch of two channels a thread transforms (FUSE: + scale / shift of the input-side BatchNorm for those two channels)
struct WinoRaw { f32x2 d[16]; f32x2 sc, sh; };

struct WinoPatch {  // a thread's patch for the transform role: byte offsets of input pixel (2a, 2b) (+ the thread's channel pair) in the
  unsigned vtop, vmid, vbot;  // group's tensor as seen by rows 2a - 1 / 2a, 2a + 1 / 2a + 2 (WN_DROP where the row or the patch does not exist)
  bool lef, rig;              // columns 2b - 1 / 2b + 2 inside the image
};

__device__ __forceinline__ void wn_patch(WinoPatch& wp, const WinoProg& P, int tile_in_group, int t_pt, int t_cp, int* ptab_slot) {
  const int pl = tile_in_group * WN_TP + t_pt;
  const bool ok = pl < P.ppg;
  const int n = wn_div(pl, P.mPPI, P.sPPI);
  const int rem = pl - n * P.ppi;
  const int a = wn_div(rem, P.mPB, P.sPB);
  const int b = rem - a * P.PB;
  const int pix = (n * P.H + 2 * a) * P.W + 2 * b;
  const unsigned vbase = (unsigned)pix * 256u + (unsigned)t_cp * 8u;
  wp.vmid = ok ? vbase : WN_DROP;
  wp.vtop = (ok && a > 0) ? vbase : WN_DROP;
  wp.vbot = (ok && a < P.PA - 1) ? vbase : WN_DROP;
  wp.lef = b > 0; wp.rig = b < P.PB - 1;
  if (t_cp == 0) *ptab_slot = ok ? pix : -1;
}

// request chunk `p` of the patch: 16 loads of 8 bytes; outside the image an out-of-range offset reads 0
template <bool FUSE = false>
__device__ __forceinline__ void wn_request(WinoRaw& rw, const WinoPatch& wp, __amdgpu_buffer_rsrc_t xb, int W, int p,
                                           const float* __restrict__ bnrec = nullptr, int t_cp = 0) {
  if constexpr (FUSE) {  // (travels with the pixels: the landing needs no idea of which group or chunk it is landing)
    rw.sc = *(const f32x2*)(bnrec + 128 + 16 * p + 2 * t_cp);
    rw.sh = *(const f32x2*)(bnrec + 192 + 16 * p + 2 * t_cp);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const unsigned vr = r == 0 ? wp.vtop : r == 3 ? wp.vbot : wp.vmid;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const unsigned off = c == 0 ? (wp.lef ? vr : WN_DROP) : c == 3 ? (wp.rig ? vr : WN_DROP) : vr;
      // the resource starts (W + 1) pixels in front of the tensor, so that the scalar offset of (r, c) is never negative
      // (column and chunk go into the instruction's immediate offset, the row into one of three scalar offsets)
      const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(xb, off + (unsigned)(c * 256 + p * 64), r * W * 256, 0);
      rw.d[r * 4 + c] = f32x2{__uint_as_float(v[0]), __uint_as_float(v[1])};
    }
  }
}

// V = B^T d B for the thread's two channels -> LDS (two columns of the transform domain at a time: 16 temporaries, not 32 — the landing
// is where the kernel's register demand peaks)
// FUSE: the tensor holds the RAW output of the previous convolution and the layer's input is relu(batchnorm(raw)) (cf. srlz_conv64_fwd's
// x_bnp): applied here, per loaded value — and the zero padding re-imposed behind it (an out-of-range load read 0, not relu(shift)).
template <bool FUSE = false>
__device__ __forceinline__ void wn_land(WinoRaw& rw, const WinoPatch& wp, float* __restrict__ Vw) {
  if constexpr (FUSE) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool rok = (r == 0 ? wp.vtop : r == 3 ? wp.vbot : wp.vmid) != WN_DROP;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const bool ok = rok && (c == 0 ? wp.lef : c == 3 ? wp.rig : true);
        f32x2 a = __builtin_elementwise_fma(rw.d[r * 4 + c], rw.sc, rw.sh);
        a[0] = ok ? __builtin_fmaxf(a[0], 0.f) : 0.f;
        a[1] = ok ? __builtin_fmaxf(a[1], 0.f) : 0.f;
        rw.d[r * 4 + c] = a;
      }
    }
  }
#pragma unroll
  for (int hv = 0; hv < 2; ++hv) {
    f32x2 e[4][2];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (hv == 0) {
        e[r][0] = rw.d[r * 4 + 0] - rw.d[r * 4 + 2];
        e[r][1] = rw.d[r * 4 + 1] + rw.d[r * 4 + 2];
      } else {
        e[r][0] = rw.d[r * 4 + 2] - rw.d[r * 4 + 1];
        e[r][1] = rw.d[r * 4 + 1] - rw.d[r * 4 + 3];
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int nu = 2 * hv + j;
      *(f32x2*)(Vw + (0 * 4 + nu) * WN_VCOMP) = e[0][j] - e[2][j];
      *(f32x2*)(Vw + (1 * 4 + nu) * WN_VCOMP) = e[1][j] + e[2][j];
      *(f32x2*)(Vw + (2 * 4 + nu) * WN_VCOMP) = e[2][j] - e[1][j];
      *(f32x2*)(Vw + (3 * 4 + nu) * WN_VCOMP) = e[1][j] - e[3][j];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

__device__ __forceinline__ f32x4 wn_load4(__amdgpu_buffer_rsrc_t b, unsigned voff, int soff) {
  const __attribute__((ext_vector_type(4))) unsigned v = __builtin_amdgcn_raw_buffer_load_b128(b, voff, soff, 0);
  return f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
}

constexpr int WN_AQ = 4;  // components the weight operand is requested ahead of its use (must divide the 64 steps of a tile)
static_assert(64 % WN_AQ == 0, "the ring of weight operands wraps with the tile");

__device__ __forceinline__ f32x4 wn_uload(__amdgpu_buffer_rsrc_t ub, unsigned uvoff, int step, int uo) {
  const __attribute__((ext_vector_type(4))) unsigned v = __builtin_amdgcn_raw_buffer_load_b128(ub, uvoff, (step & 63) * 4096 + uo, 0);
  return f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
}

// 16 components x 8 v_mfma_f32_16x16x4_f32 of chunk P: the weight operand (U, the MFMA's A) comes straight from L2 into registers,
// WN_AQ components ahead (one 1 KB wave-load per component: [g][16 co][4]); the patch operand (V) from LDS
template <int PH>
__device__ __forceinline__ void wn_mfma_phase(f32x4 (&acc)[16][2], f32x4 (&aq)[WN_AQ], const float* __restrict__ Bp,
                                              __amdgpu_buffer_rsrc_t ub, unsigned uvoff, int uo) {
  // the patch operand one component ahead of its use (hipcc left to itself reads it right in front of the MFMAs that need it)
  f32x4 b0n = *(const f32x4*)(Bp), b1n = *(const f32x4*)(Bp + 256);
#pragma unroll
  for (int comp = 0; comp < 16; ++comp) {
    const int step = PH * 16 + comp;
    const f32x4 a = aq[step % WN_AQ];
    const f32x4 b0 = b0n, b1 = b1n;
    if (comp < 15) {
      b0n = *(const f32x4*)(Bp + (comp + 1) * WN_VCOMP);
      b1n = *(const f32x4*)(Bp + (comp + 1) * WN_VCOMP + 256);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (PH == 0 && i == 0) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        acc[comp][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b0[i], z, 0, 0, 0);
        acc[comp][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b1[i], z, 0, 0, 0);
      } else {
        acc[comp][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b0[i], acc[comp][0], 0, 0, 0);
        acc[comp][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b1[i], acc[comp][1], 0, 0, 0);
      }
    }
    aq[step % WN_AQ] = wn_uload(ub, uvoff, step + WN_AQ, uo);
  }
  // the order of the phase, spelled out for the scheduler (which otherwise sinks every operand read to the MFMA that consumes it):
  // the reads of component c + 1, the eight MFMAs of component c, the weight request of component c + WN_AQ
  __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
  for (int comp = 0; comp < 16; ++comp) {
    if (comp < 15) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
}

// Epilogue option of the data-gradient launch whose output is the gradient of a POOLED map (conv2's data gradient = d pooled1; cf.
// conv64.hip's PoolSum / conv64_dgrad_poolsum_kernel): instead of BatchNorm-forward statistics the tile's record receives the two
// BatchNorm-BACKWARD sums of the block that produced the pooled map,  sum dz  and  sum dz * xhat  with dz = d pooled where pooled > 0
// (the gradient of max-pool + ReLU lives at the window's argmax, where the pooled value IS relu(bn(y)): xhat follows from it).
// y / argmax are only touched for channels whose BatchNorm scale is (almost) 0.
struct WinoPoolSum {
  const float* pooled;    // [N,H,W,64] like the launch's output; NULL = off
  const float* bnp;       // records of the pooled block's BatchNorm (256 floats per group)
  const float* y;         // raw convolution output under the pooling [N,YH,YW,64]
  const uint8_t* argmax;  // [N,H,W,64]
  long long y_gstride;    // floats between two groups' images in y
  int YH, YW, pad;
};

__device__ __forceinline__ float wn_row16_sum(float v) {  // sum over the 16 lanes of a row (every lane gets it): four rotating DPP adds
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x128, 0xf, 0xf, false));
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x124, 0xf, 0xf, false));
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x122, 0xf, 0xf, false));
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x121, 0xf, 0xf, false));
  return v;
}

// Companion of the pooled-block epilogue: the channels whose BatchNorm scale is (almost) 0 — exactly 0: relu(bn(.)) is the constant
// max(shift, 0) and every window's first position is the argmax; xhat cannot be recovered from the pooled value — summed from the
// convolution output under the recorded argmax, with the forward's own ReLU expression, into WN_ZBLOCKS records behind the main kernel's
// (which leaves 0 for them).  A group without such a channel (the normal case) costs one ~4 us launch that writes zero records; with one,
// this is a pass over (d pooled, argmax) and a gather from y: rare, and slow on purpose.
constexpr int WN_ZBLOCKS = 64;
__global__ __launch_bounds__(256) void conv64_wino_poolsum_zero_scale_kernel(const float* __restrict__ dx, const WinoPoolSum ps, float* __restrict__ partial,
                                                                            int N, int H, int W, long long gstride, int rows, int first_row) {
  const int g = blockIdx.y;
  const float* __restrict__ rec = ps.bnp + g * 256;
  const int c4 = threadIdx.x & 15;
  const f32x4 mean = *(const f32x4*)(rec + c4 * 4), pinv = *(const f32x4*)(rec + 64 + c4 * 4);
  const f32x4 psc = *(const f32x4*)(rec + 128 + c4 * 4), psh = *(const f32x4*)(rec + 192 + c4 * 4);
  unsigned zmask = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) zmask |= ((fabsf(psc[j]) <= 1e-3f * fabsf(psh[j]) || psc[j] == 0.f) ? 1u : 0u) << j;
  double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
  if (__syncthreads_or(zmask != 0)) {
    const long long pixels = (long long)N * H * W;
    for (long long pix = (long long)blockIdx.x * 16 + (threadIdx.x >> 4); pix < pixels; pix += (long long)gridDim.x * 16) {
      if (!zmask) continue;
      const int n = (int)(pix / (H * W));
      const int yy = (int)(pix - (long long)n * H * W) / W, xx = (int)(pix - ((long long)n * H + yy) * W);
      const uint32_t packed = *(const uint32_t*)(ps.argmax + (size_t)g * gstride + (size_t)pix * 64 + c4 * 4);
      const f32x4 v = *(const f32x4*)(dx + (size_t)g * gstride + (size_t)pix * 64 + c4 * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if ((zmask >> j) & 1u) {
          const int a = (packed >> (8 * j)) & 0xff;
          const int iy = yy * 2 - ps.pad + a / 3, ix = xx * 2 - ps.pad + a % 3;
          const float vy = ps.y[g * ps.y_gstride + ((size_t)(n * ps.YH + iy) * ps.YW + ix) * 64 + c4 * 4 + j];
          if (vy * psc[j] + psh[j] > 0.f) {
            s1[j] += (double)v[j];
            s2[j] += (double)(v[j] * ((vy - mean[j]) * pinv[j]));
          }
        }
    }
  }
  __shared__ double sm[16][128];
  const int prow = threadIdx.x >> 4;
#pragma unroll
  for (int j = 0; j < 4; ++j) { sm[prow][c4 * 4 + j] = s1[j]; sm[prow][64 + c4 * 4 + j] = s2[j]; }
  __syncthreads();
  if (threadIdx.x < 128) {
    double t = 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += sm[r][threadIdx.x];
    partial[((size_t)g * rows + first_row + blockIdx.x) * 128 + threadIdx.x] = (float)t;
  }
}

template <bool PSUM, bool FUSE>
__global__ __launch_bounds__(WN_THREADS, 2) void conv64_wino_kernel(const float* __restrict__ x_all, const float* __restrict__ upack,
                                                                    const float* __restrict__ bias, float* __restrict__ y_all,
                                                                    float* __restrict__ stats_partial, const WinoProg P, int ntiles,
                                                                    const WinoPoolSum ps, const float* __restrict__ x_bnp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* Vs = (float*)smem;                // [2 (chunk parity)][WN_VCHUNK]
  int* ptab = (int*)(Vs + 2 * WN_VCHUNK);  // [2 (tile parity)][32]: pixel index of output (2a, 2b) of a tile's patches, -1 = no such patch

  const int tid = threadIdx.x;
  int lane = tid & 63;
  asm volatile("" : "+v"(lane));
  const int wave = tid >> 6;
  // transform role: patch of the tile, channel pair of the chunk — 8 consecutive lanes load 64 contiguous bytes of a pixel
  const int t_pt = wave * 8 + (lane >> 3), t_cp = lane & 7;
  const int cb = wave;                                        // matrix role: 16 output channels x the tile's 32 patches
  const int l15 = lane & 15, g = lane >> 4;
  // V in LDS: [comp][patch][k-group ^ 2 ((patch >> 2) & 1)][4].  Conflict-free both ways: a ds_write_b64 is served in groups of 16
  // consecutive lanes over 32 banks (two patches x their 64 bytes = 128 contiguous bytes), a ds_read_b128 in the lane groups
  // {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... over 64 banks — where the XOR separates patches 0-3 / 12-15 of one k-group from
  // 4-7 / 8-11 of the next
  float* Vw = Vs + (t_pt * 4 + ((t_cp >> 1) ^ (((t_pt >> 2) & 1) << 1))) * 4 + 2 * (t_cp & 1);
  const float* Bp = Vs + (l15 * 4 + (g ^ (((l15 >> 2) & 1) << 1))) * 4;
  const unsigned uvoff = (unsigned)(g * 64 + 16 * cb + l15) * 16u;
  const __amdgpu_buffer_rsrc_t ub = wn_buffer(upack, 4u * WN_CHUNK * 4u);

  // XCD-contiguous runs of tiles (block b runs on XCD b % 8: neighbouring tiles share input rows in that XCD's L2)
  const int xcd = blockIdx.x & 7, wi = blockIdx.x >> 3, wpx = gridDim.x >> 3;
  const int tq = ntiles >> 3, tr = ntiles & 7;
  const int tbase = (xcd < tr) ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
  const int tcnt = tq + (xcd < tr ? 1 : 0);
  int k = wi;
  if (k >= tcnt) return;

  const unsigned xbytes = (unsigned)(P.gstride * 4) + (unsigned)(P.W + 1) * 256u;
  const unsigned ybytes = (unsigned)(P.gstride * 4);
  const __amdgpu_buffer_rsrc_t sbuf = wn_buffer(stats_partial, stats_partial ? (unsigned)(P.G * (P.tpg + (PSUM ? WN_ZBLOCKS : 0))) * 512u : 0u);

  // Pipeline (one barrier per chunk): during the matrix work of chunk c (from V[c & 1]) chunk c + 1 is transformed into V[(c + 1) & 1]
  // — at the top of the phase, from the registers its raw patch was loaded into during chunk c - 1 — and chunk c + 2 is requested.
  WinoRaw rw;
  WinoPatch wp;
  f32x4 aq[WN_AQ];
  int parity = 0;
  {
    const int tile = tbase + k;
    const int grp = tile / P.tpg;
    const __amdgpu_buffer_rsrc_t xb = wn_buffer(x_all + grp * P.gstride - (P.W + 1) * 64, xbytes);
    wn_patch(wp, P, tile - grp * P.tpg, t_pt, t_cp, ptab + t_pt);
    const float* bnrec = FUSE ? x_bnp + grp * 256 : nullptr;
    wn_request<FUSE>(rw, wp, xb, P.W, 0, bnrec, t_cp);
    wn_land<FUSE>(rw, wp, Vw);
    wn_request<FUSE>(rw, wp, xb, P.W, 1, bnrec, t_cp);
#pragma unroll
    for (int i = 0; i < WN_AQ; ++i) aq[i] = wn_uload(ub, uvoff, i, 0);
  }
  for (; k < tcnt; k += wpx, parity ^= 1) {
    const int tile = tbase + k;
    const int grp = tile / P.tpg;
    const int til = tile - grp * P.tpg;
    const __amdgpu_buffer_rsrc_t xb = wn_buffer(x_all + grp * P.gstride - (P.W + 1) * 64, xbytes);
    const int k2 = k + wpx;
    const bool more = k2 < tcnt;
    const int tile2 = tbase + (more ? k2 : k);  // (past the last tile: the same tile once more — chunks nobody multiplies)
    const int grp2 = tile2 / P.tpg;
    const __amdgpu_buffer_rsrc_t xb2 = wn_buffer(x_all + grp2 * P.gstride - (P.W + 1) * 64, xbytes);
    int uo = 0;  // (the weights are the same for every tile: without the opaque offset hipcc hoists their loads out of the tile loop)
    asm volatile("" : "+s"(uo));
    f32x4 acc[16][2];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      __syncthreads();  // chunk p has landed in V[p & 1] (all waves), and everybody is done multiplying chunk p - 1 out of V[(p + 1) & 1]
      wn_land<FUSE>(rw, wp, Vw + ((p + 1) & 1) * WN_VCHUNK);  // chunk p + 1 (p == 3: the next tile's chunk 0)
      if (p < 2) {
        wn_request<FUSE>(rw, wp, xb, P.W, p + 2, FUSE ? x_bnp + grp * 256 : nullptr, t_cp);
      } else {
        if (p == 2) wn_patch(wp, P, tile2 - grp2 * P.tpg, t_pt, t_cp, ptab + (parity ^ 1) * WN_TP + t_pt);
        wn_request<FUSE>(rw, wp, xb2, P.W, p - 2, FUSE ? x_bnp + grp2 * 256 : nullptr, t_cp);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (p == 0) wn_mfma_phase<0>(acc, aq, Bp, ub, uvoff, uo);
      if (p == 1) wn_mfma_phase<1>(acc, aq, Bp + WN_VCHUNK, ub, uvoff, uo);
      if (p == 2) wn_mfma_phase<2>(acc, aq, Bp, ub, uvoff, uo);
      if (p == 3) wn_mfma_phase<3>(acc, aq, Bp + WN_VCHUNK, ub, uvoff, uo);
    }
    // ---- epilogue: Y = A^T M A per patch, bias, stores, BatchNorm partial sums.  A lane holds, of the patches 32 pb + 16 s + l15,
    // the channels 16 cb + 4 g + {0..3}
    const __amdgpu_buffer_rsrc_t yb = wn_buffer(y_all + grp * P.gstride, ybytes);
    f32x4 s4 = {0.f, 0.f, 0.f, 0.f}, q4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (bias) bias4 = *(const f32x4*)(bias + 16 * cb + 4 * g);
    // PSUM: where the pooled value z is positive it is gamma * xhat + beta, so xhat = z * pA + pB (pA = invstd / scale,
    // pB = -(shift / scale + mean) * invstd); q4 collects sum dA * z (z is 0 where the ReLU is closed: no mask), s4 the masked sum of dA,
    // and sum dz * xhat = pA q4 + pB s4 at the end.  A channel whose scale is (almost) 0 contributes nothing here (threshold +inf,
    // pA = pB = 0) and is summed by the cold loop below from the convolution output under the recorded argmax.
    f32x4 pA = {0.f, 0.f, 0.f, 0.f}, pB = pA, pthr = pA;
    const __amdgpu_buffer_rsrc_t zb = wn_buffer(PSUM ? ps.pooled + grp * P.gstride : nullptr, PSUM ? ybytes : 0u);
    if constexpr (PSUM) {
      const float* __restrict__ rec = ps.bnp + grp * 256 + 16 * cb + 4 * g;
      const f32x4 mean = *(const f32x4*)rec, pinv = *(const f32x4*)(rec + 64), psc = *(const f32x4*)(rec + 128), psh = *(const f32x4*)(rec + 192);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool zero = fabsf(psc[e]) <= 1e-3f * fabsf(psh[e]) || psc[e] == 0.f;
        pthr[e] = zero ? __builtin_inff() : 0.f;
        const float isc = zero ? 0.f : 1.f / psc[e];
        pA[e] = pinv[e] * isc;
        pB[e] = zero ? 0.f : -(psh[e] * isc + mean[e]) * pinv[e];
      }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int pix = ptab[parity * WN_TP + 16 * s + l15];
      const bool ok = pix >= 0;
      const unsigned yoff = ok ? (unsigned)pix * 256u + (unsigned)(16 * cb + 4 * g) * 4u : WN_DROP;
      f32x4 z00, z01, z10, z11;
      if constexpr (PSUM) {  // the pooled values of the patch's four pixels (outside: 0 — an out-of-range offset)
        z00 = wn_load4(zb, yoff, 0); z01 = wn_load4(zb, yoff, 256);
        z10 = wn_load4(zb, yoff, P.W * 256); z11 = wn_load4(zb, yoff, P.W * 256 + 256);
      }
      const f32x4 bz = {ok ? bias4[0] : 0.f, ok ? bias4[1] : 0.f, ok ? bias4[2] : 0.f, ok ? bias4[3] : 0.f};
      f32x4 y00 = bz, y01 = bz, y10 = bz, y11 = bz;
#pragma unroll
      for (int nu = 0; nu < 4; ++nu) {  // column nu of M: t = A^T M[:, nu], then its row of A
        const f32x4 t0 = acc[0 + nu][s] + acc[4 + nu][s] + acc[8 + nu][s];
        const f32x4 t1 = acc[4 + nu][s] - acc[8 + nu][s] - acc[12 + nu][s];
        if (nu < 3) { y00 += t0; y10 += t1; }
        if (nu == 1) { y01 += t0; y11 += t1; }
        if (nu >= 2) { y01 -= t0; y11 -= t1; }
      }
      __builtin_amdgcn_raw_buffer_store_b128(y00, yb, yoff, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(y01, yb, yoff, 256, 0);
      __builtin_amdgcn_raw_buffer_store_b128(y10, yb, yoff, P.W * 256, 0);
      __builtin_amdgcn_raw_buffer_store_b128(y11, yb, yoff, P.W * 256 + 256, 0);
      if constexpr (PSUM) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s4[e] += ((z00[e] > pthr[e] ? y00[e] : 0.f) + (z01[e] > pthr[e] ? y01[e] : 0.f)) +
                   ((z10[e] > pthr[e] ? y10[e] : 0.f) + (z11[e] > pthr[e] ? y11[e] : 0.f));
        }
        q4 += (y00 * z00 + y01 * z01) + (y10 * z10 + y11 * z11);
      } else if constexpr (FUSE) {  // (a patch that does not exist has transformed relu(shift), not zeros)
        const f32x4 ts = (y00 + y01) + (y10 + y11), tq = (y00 * y00 + y01 * y01) + (y10 * y10 + y11 * y11);
#pragma unroll
        for (int e = 0; e < 4; ++e) { s4[e] += ok ? ts[e] : 0.f; q4[e] += ok ? tq[e] : 0.f; }
      } else {
        s4 += (y00 + y01) + (y10 + y11);
        q4 += (y00 * y00 + y01 * y01) + (y10 * y10 + y11 * y11);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (PSUM) {
#pragma unroll
      for (int e = 0; e < 4; ++e) q4[e] = __builtin_fmaf(pA[e], q4[e], pB[e] * s4[e]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { s4[e] = wn_row16_sum(s4[e]); q4[e] = wn_row16_sum(q4[e]); }
    const unsigned rec = (unsigned)(grp * (P.tpg + (PSUM ? WN_ZBLOCKS : 0)) + til) * 512u + (unsigned)(16 * cb + 4 * g) * 4u;
    __builtin_amdgcn_raw_buffer_store_b128(s4, sbuf, l15 == 0 ? rec : WN_DROP, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(q4, sbuf, l15 == 0 ? rec + 256u : WN_DROP, 0, 0);
  }
}

int wino_program(WinoProg* P, const srlz_conv64_desc* d) {
  if (!d || d->transposed || d->ksize != 3 || d->stride != 1 || d->pad != 1) return 1;
  if (d->hi != d->ho || d->wi != d->wo || (d->hi & 1) || (d->wi & 1) || d->hi < 4 || d->wi < 4 || d->n <= 0) return 1;
  const int G = d->groups > 1 ? d->groups : 1;
  if (d->n % G) return 1;
  P->G = G; P->N = d->n / G; P->H = d->hi; P->W = d->wi; P->PA = d->hi / 2; P->PB = d->wi / 2;
  const long long ppi = (long long)P->PA * P->PB, ppg = ppi * P->N;
  const long long gfl = (long long)P->N * P->H * P->W * 64;
  // 32-bit byte offsets inside a group's buffer (+ the (W + 1)-pixel lead of the input resource), 31-bit patch indices
  if (gfl * 4 + (long long)(P->W + 1) * 256 >= 0x7FFF0000LL || ppg + WN_TP >= (1LL << 31) || ppi < 2) return 1;
  P->ppi = (int)ppi; P->ppg = (int)ppg; P->tpg = (int)((ppg + WN_TP - 1) / WN_TP);
  P->gstride = gfl;
  wn_fastdiv_init((unsigned)P->ppi, &P->mPPI, &P->sPPI);
  wn_fastdiv_init((unsigned)P->PB, &P->mPB, &P->sPB);
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// The weight gradient of the same layer, by the transposed algorithm:
//     dW = G^T [ sum over patches  (A dY A^T) .* (B^T d B) ] G
// — per transform-domain component (xi, nu) ONE [co 64] x [ci 64] contraction over all patches, 16 / 36 of the direct weight gradient's
// multiply-adds; V = B^T d B is the forward's patch transform, Z = A dY A^T (A 4 x 2: entries 0, +-1) turns the patch's 2 x 2 output
// gradients into 16 values with 12 additions.  G^T . G (4 x 4 -> 3 x 3) runs once, in the second stage of the split-K reduction.
//
// Workgroup (256 threads, two per CU): HALF of the components — nu in {2 S, 2 S + 1}, S = workgroup parity — over a contiguous run of
// 8-patch stages; both operands of a stage ([8 comps][8 patches][64 channels], 16 KB each) live in LDS, double-buffered, one barrier per
// stage, the raw pixels of stage s + 2 in flight during the matrix work of stage s (the forward kernel's pipeline).  A thread transforms
// one patch x one channel PAIR of both operands (12 + 4 loads of 8 bytes: the half needs three of the patch's four columns).
// Wave (ch = wave & 1, cq = wave >> 1): output channels 32 ch .., components 4 cq .. of the half, all 64 input channels:
// 4 comps x 2 blocks of 32 x 32 = 128 accumulator registers, v_mfma_f32_32x32x2_f32 (k = 2 patches per instruction), operands as
// ds_read_b64 (two k-steps per read) from [comp][j][h][channel][e] with patch-in-stage = 4 j + 2 h + e... any bijection serves as long as
// Z and V share it; this one makes both the reads (32 lanes = 256 contiguous bytes) and the transform's writes conflict-free.
// ---------------------------------------------------------------------------------------------------------------
constexpr int WG_SP = 8;                    // patches per stage
constexpr int WG_OP = 8 * WG_SP * 64;       // floats of one operand of a stage (8 comps): 16 KB
constexpr int WG_COMP = WG_SP * 64;         // floats of one component of an operand: [j 2][h 2][channel 64][e 2]

struct WinoWgradRaw { f32x2 d[12]; f32x2 g[4]; };

struct WinoWgradProg {
  int N, H, W, PA, PB, ppi;
  int total;      // patches of all images
  int stages;     // ceil(total / 8)
  int spw;        // stages per workgroup pair
  unsigned mPPI, mPB;
  int sPPI, sPB;
};

struct WinoWgradPatch { unsigned vtop, vmid, vbot, vdy; bool lef, rig; };

__device__ __forceinline__ void wg_patch(WinoWgradPatch& wp, const WinoWgradProg& P, int patch, int chan_pair) {
  const bool ok = patch < P.total;
  const int n = wn_div(patch, P.mPPI, P.sPPI);
  const int rem = patch - n * P.ppi;
  const int a = wn_div(rem, P.mPB, P.sPB);
  const int b = rem - a * P.PB;
  const int pix = (n * P.H + 2 * a) * P.W + 2 * b;
  const unsigned vbase = (unsigned)pix * 256u + (unsigned)chan_pair * 8u;
  wp.vmid = ok ? vbase : WN_DROP;
  wp.vtop = (ok && a > 0) ? vbase : WN_DROP;
  wp.vbot = (ok && a < P.PA - 1) ? vbase : WN_DROP;
  wp.vdy = wp.vmid;
  wp.lef = b > 0; wp.rig = b < P.PB - 1;
}

// S = 0: columns 0..2 of the 4 x 4 input patch, S = 1: columns 1..3; and the 2 x 2 output gradients
template <int S>
__device__ __forceinline__ void wg_request(WinoWgradRaw& rw, const WinoWgradPatch& wp, __amdgpu_buffer_rsrc_t xb, __amdgpu_buffer_rsrc_t gb, int W) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const unsigned vr = r == 0 ? wp.vtop : r == 3 ? wp.vbot : wp.vmid;
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
      const int c = cc + S;
      const unsigned off = c == 0 ? (wp.lef ? vr : WN_DROP) : c == 3 ? (wp.rig ? vr : WN_DROP) : vr;
      const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(xb, off + (unsigned)(c * 256), r * W * 256, 0);
      rw.d[r * 3 + cc] = f32x2{__uint_as_float(v[0]), __uint_as_float(v[1])};
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(gb, wp.vdy + (unsigned)(j * 256), i * W * 256, 0);
      rw.g[i * 2 + j] = f32x2{__uint_as_float(v[0]), __uint_as_float(v[1])};
    }
}

// the thread's patch x channel pair of both operands -> LDS: V[xi][nu] = (B^T d B)[xi][nu], Z[xi][nu] = (A dY A^T)[xi][nu], nu in the half
template <int S>
__device__ __forceinline__ void wg_land(const WinoWgradRaw& rw, float* __restrict__ Zw, float* __restrict__ Vw) {
  {
    f32x2 e[4][2];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const f32x2 d0 = rw.d[r * 3], d1 = rw.d[r * 3 + 1], d2 = rw.d[r * 3 + 2];
      if (S == 0) { e[r][0] = d0 - d2; e[r][1] = d1 + d2; }   // columns 0, 1, 2: nu = 0, 1
      else        { e[r][0] = d1 - d0; e[r][1] = d0 - d2; }   // columns 1, 2, 3: nu = 2, 3
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const f32x2 v0 = e[0][j] - e[2][j], v1 = e[1][j] + e[2][j], v2 = e[2][j] - e[1][j], v3 = e[1][j] - e[3][j];
      const f32x2 v[4] = {v0, v1, v2, v3};
#pragma unroll
      for (int xi = 0; xi < 4; ++xi) {  // the pair's two channels are two rows of the operand: 8 bytes apart
        Vw[(xi * 2 + j) * WG_COMP] = v[xi][0];
        Vw[(xi * 2 + j) * WG_COMP + 2] = v[xi][1];
      }
    }
  }
  {
    f32x2 t[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (S == 0) { t[i][0] = rw.g[i * 2]; t[i][1] = rw.g[i * 2] + rw.g[i * 2 + 1]; }
      else        { t[i][0] = rw.g[i * 2] - rw.g[i * 2 + 1]; t[i][1] = -rw.g[i * 2 + 1]; }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const f32x2 z0 = t[0][j], z1 = t[0][j] + t[1][j], z2 = t[0][j] - t[1][j], z3 = -t[1][j];
      const f32x2 z[4] = {z0, z1, z2, z3};
#pragma unroll
      for (int xi = 0; xi < 4; ++xi) {
        Zw[(xi * 2 + j) * WG_COMP] = z[xi][0];
        Zw[(xi * 2 + j) * WG_COMP + 2] = z[xi][1];
      }
    }
  }
}

// one stage: 4 comps x 2 patch quads x 2 k-steps x 2 input-channel blocks = 32 MFMAs per wave; the operands of component c + 1 are read
// under the MFMAs of component c (spelled out for the scheduler, which otherwise sinks every read to the MFMA that consumes it)
__device__ __forceinline__ void wg_mfma_stage(f32x16 (&acc)[4][2], const float* __restrict__ Zp, const float* __restrict__ Vp) {
  f32x2 an[2], bn[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    an[j] = *(const f32x2*)(Zp + j * 256);
    bn[j][0] = *(const f32x2*)(Vp + j * 256);
    bn[j][1] = *(const f32x2*)(Vp + j * 256 + 64);
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    f32x2 a[2], b[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) { a[j] = an[j]; b[j][0] = bn[j][0]; b[j][1] = bn[j][1]; }
    if (c < 3) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        an[j] = *(const f32x2*)(Zp + (c + 1) * WG_COMP + j * 256);
        bn[j][0] = *(const f32x2*)(Vp + (c + 1) * WG_COMP + j * 256);
        bn[j][1] = *(const f32x2*)(Vp + (c + 1) * WG_COMP + j * 256 + 64);
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        acc[c][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j][e], b[j][0][e], acc[c][0], 0, 0, 0);
        acc[c][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j][e], b[j][1][e], acc[c][1], 0, 0, 0);
      }
  }
  __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    if (c < 3) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
}

template <int S>
__device__ __forceinline__ void wg_body(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ wpart,
                                        const WinoWgradProg& P, float* __restrict__ smem, int slice) {
  const int tid = threadIdx.x;
  int lane = tid & 63;
  asm volatile("" : "+v"(lane));
  const int wave = tid >> 6;
  // transform role: patch q of the stage (LDS coordinates e = q & 1, h = (q >> 1) & 1, j = q >> 2), channel pair 8 wave + (lane & 7)
  const int q = lane >> 3, cpair = wave * 8 + (lane & 7);
  const int tw = (((q >> 2) * 2 + ((q >> 1) & 1)) * 64 + 2 * cpair) * 2 + (q & 1);
  // matrix role
  const int ch = wave & 1, cq = wave >> 1, l31 = lane & 31, h = lane >> 5;
  const int tz = cq * 4 * WG_COMP + (h * 64 + 32 * ch + l31) * 2, tv = cq * 4 * WG_COMP + (h * 64 + l31) * 2;

  const long long tfl = (long long)P.N * P.H * P.W * 64;
  const __amdgpu_buffer_rsrc_t xb = wn_buffer(x - (P.W + 1) * 64, (unsigned)(tfl * 4) + (unsigned)(P.W + 1) * 256u);
  const __amdgpu_buffer_rsrc_t gb = wn_buffer(dy, (unsigned)(tfl * 4));

  f32x16 acc[4][2];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][b][r] = 0.f;

  const int s0 = slice * P.spw;
  int s1 = s0 + P.spw;
  if (s1 > P.stages) s1 = P.stages;
  WinoWgradRaw rw;
  WinoWgradPatch wp;
  if (s0 < s1) {
    // stage s0 -> buffer 0; stage s0 + 1 requested
    wg_patch(wp, P, s0 * WG_SP + q, cpair);
    wg_request<S>(rw, wp, xb, gb, P.W);
    wg_land<S>(rw, smem + tw, smem + WG_OP + tw);
    wg_patch(wp, P, (s0 + 1) * WG_SP + q, cpair);   // (past the run: stages nobody multiplies; past the tensor: zeros)
    wg_request<S>(rw, wp, xb, gb, P.W);
    for (int s = s0; s < s1; ++s) {
      const int par = (s - s0) & 1;
      float* cur = smem + par * 2 * WG_OP;
      float* nxt = smem + (par ^ 1) * 2 * WG_OP;
      __syncthreads();  // stage s has landed in `cur` (all waves), and everybody is done multiplying stage s - 1 out of `nxt`
      wg_land<S>(rw, nxt + tw, nxt + WG_OP + tw);
      wg_patch(wp, P, (s + 2) * WG_SP + q, cpair);
      wg_request<S>(rw, wp, xb, gb, P.W);
      __builtin_amdgcn_sched_barrier(0);
      wg_mfma_stage(acc, cur + tz, cur + WG_OP + tv);
    }
  }
  // the workgroup's partial: [comp 8 (xi * 2 + nu & 1)][co 64][ci 64]
  float* out = wpart + (size_t)blockIdx.x * (8 * 4096);
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = 32 * ch + (r & 3) + 8 * (r >> 2) + 4 * h;
        out[((cq * 4 + c) * 64 + co) * 64 + 32 * b + l31] = acc[c][b][r];
      }
}

__global__ __launch_bounds__(256, 2) void conv64_wino_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                   float* __restrict__ wpart, const WinoWgradProg P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* smem = (float*)smem_raw;  // [2 (stage parity)][Z | V][WG_OP]
  // workgroup b: XCD b & 7; within the XCD, consecutive workgroups are the two halves of one slice (they read the same pixels)
  const int xcd = blockIdx.x & 7, wi = blockIdx.x >> 3, wpx = gridDim.x >> 3;
  const int slice = xcd * (wpx >> 1) + (wi >> 1);
  if (wi & 1) wg_body<1>(x, dy, wpart, P, smem, slice);
  else wg_body<0>(x, dy, wpart, P, smem, slice);
}

// second stage, part one: the partials of `per` workgroups summed in fp64, fixed order -> mid[chunk][comp 16][co][ci] (double)
__global__ __launch_bounds__(256) void conv64_wino_wgrad_reduce_a(const float* __restrict__ wpart, double* __restrict__ mid, int nwg, int per) {
  const int id = blockIdx.x * 256 + threadIdx.x;  // (comp 16, co, ci)
  const int chunk = blockIdx.y;
  const int comp = id >> 12, rest = id & 4095;
  const int xi = comp >> 2, nu = comp & 3;
  const int half = nu >> 1, local = xi * 2 + (nu & 1);
  // workgroup b holds half (b >> 3) & 1
  double t = 0.0;
  const int npair = nwg >> 1;
  for (int k = chunk * per; k < (chunk + 1) * per && k < npair; ++k) {
    const int b = ((k >> 3) * 2 + half) * 8 + (k & 7);  // the k-th workgroup of this half: pairs are (wi = 2 m, 2 m + 1) on XCD b & 7
    t += (double)wpart[(size_t)b * (8 * 4096) + local * 4096 + rest];
  }
  mid[((size_t)chunk * 16 + comp) * 4096 + rest] = t;
}

// ... part two: the chunks summed (one thread per component and (co, ci): 16 x 4096 threads), then dW = G^T M G per (co, ci) through LDS
// -> the reference layout [co][ci][3][3]
__global__ __launch_bounds__(256) void conv64_wino_wgrad_reduce_b(const double* __restrict__ mid, int chunks, float* __restrict__ dw_ref) {
  __shared__ double m[16][17];
  const int comp = threadIdx.x >> 4, pr = threadIdx.x & 15;
  const int id = blockIdx.x * 16 + pr;  // (co, ci)
  double t = 0.0;
  for (int k = 0; k < chunks; ++k) t += mid[((size_t)k * 16 + comp) * 4096 + id];
  m[pr][comp] = t;
  __syncthreads();
  if (threadIdx.x < 144) {  // (pair, ky, kx);  G^T (3 x 4) = [[1, .5, .5, 0], [0, .5, -.5, 0], [0, .5, .5, 1]]
    const int p2 = threadIdx.x / 9, tap = threadIdx.x - p2 * 9, ky = tap / 3, kx = tap - ky * 3;
    double r[4];
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) {
      const double m0 = m[p2][nu], m1 = m[p2][4 + nu], m2 = m[p2][8 + nu], m3 = m[p2][12 + nu];
      r[nu] = ky == 0 ? m0 + 0.5 * (m1 + m2) : ky == 1 ? 0.5 * (m1 - m2) : 0.5 * (m1 + m2) + m3;
    }
    const double v = kx == 0 ? r[0] + 0.5 * (r[1] + r[2]) : kx == 1 ? 0.5 * (r[1] - r[2]) : 0.5 * (r[1] + r[2]) + r[3];
    dw_ref[(blockIdx.x * 16 + p2) * 9 + tap] = (float)v;
  }
}

constexpr int WG_CHUNKS = 8;

int wino_wgrad_program(WinoWgradProg* P, const srlz_conv64_desc* d, int* grid) {
  WinoProg F;
  if (wino_program(&F, d)) return 1;
  const long long total = (long long)d->n * F.ppi;
  if ((long long)d->n * d->hi * d->wi * 256 + (long long)(d->wi + 1) * 256 >= 0x7FFF0000LL || total + 64 >= (1LL << 31)) return 1;
  P->N = d->n; P->H = d->hi; P->W = d->wi; P->PA = F.PA; P->PB = F.PB; P->ppi = F.ppi;
  P->total = (int)total;
  P->stages = (int)((total + WG_SP - 1) / WG_SP);
  P->mPPI = F.mPPI; P->sPPI = F.sPPI; P->mPB = F.mPB; P->sPB = F.sPB;
  int g = 2 * srlz_device_cus();          // two workgroups per CU: pairs = the two component halves of one slice
  g = (g + 15) & ~15;
  while (g > 16 && (g >> 1) * 4 > P->stages) g -= 16;  // (small launches: at least four stages per slice)
  const int slices = g >> 1;
  P->spw = (P->stages + slices - 1) / slices;
  *grid = g;
  return 0;
}

}  // namespace

extern "C" int srlz_conv64_wino_supported(const srlz_conv64_desc* d) {
  WinoProg P;
  return wino_program(&P, d) ? 0 : 1;
}

extern "C" size_t srlz_conv64_wino_packed_floats(void) { return (size_t)4 * WN_CHUNK; }

extern "C" int srlz_conv64_wino_pack_weights(const float* w_ref, float* upack_fwd, float* upack_bwd, srlz_stream_t stream) {
  SRLZ_REQUIRE(w_ref && (upack_fwd || upack_bwd), SRLZ_ERR_NULL, "conv64_wino_pack_weights: null pointer");
  hipLaunchKernelGGL(conv64_wino_pack_kernel, dim3(16), dim3(256), 0, as_stream(stream), w_ref, upack_fwd, upack_bwd);
  SRLZ_LAUNCHED();
  return 0;
}

extern "C" int srlz_conv64_wino_tiles(const srlz_conv64_desc* d) {
  WinoProg P;
  if (wino_program(&P, d)) return -1;
  return P.G * P.tpg;
}

static int wino_launch(const float* x, const float* upack, const float* bias, float* y, float* partial, const WinoProg& P,
                       const WinoPoolSum* ps, const float* x_bnp, srlz_stream_t stream) {
  const int ntiles = P.G * P.tpg;
  int grid = 2 * srlz_device_cus();  // two workgroups per CU (64 KB of LDS, 256 registers each): one's barriers, landings and epilogue
  if (ntiles < grid) grid = ntiles;  // run under the other's matrix work
  grid = (grid + 7) & ~7;
  const size_t lds = (size_t)2 * WN_VCHUNK * 4 + 2 * WN_TP * 4;
  const WinoPoolSum none = {nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0};
  hipStream_t st = as_stream(stream);
  if (ps) {
    SRLZ_MAX_LDS((conv64_wino_kernel<true, false>), lds);
    hipLaunchKernelGGL((conv64_wino_kernel<true, false>), dim3(grid), dim3(WN_THREADS), lds, st, x, upack, bias, y, partial, P, ntiles, *ps, x_bnp);
  } else if (x_bnp) {
    SRLZ_MAX_LDS((conv64_wino_kernel<false, true>), lds);
    hipLaunchKernelGGL((conv64_wino_kernel<false, true>), dim3(grid), dim3(WN_THREADS), lds, st, x, upack, bias, y, partial, P, ntiles, none, x_bnp);
  } else {
    SRLZ_MAX_LDS((conv64_wino_kernel<false, false>), lds);
    hipLaunchKernelGGL((conv64_wino_kernel<false, false>), dim3(grid), dim3(WN_THREADS), lds, st, x, upack, bias, y, partial, P, ntiles, none, x_bnp);
  }
  SRLZ_LAUNCHED();
  if (ps) {  // the records of the channels the main kernel cannot sum from the pooled value (normally: zeros)
    hipLaunchKernelGGL(conv64_wino_poolsum_zero_scale_kernel, dim3(WN_ZBLOCKS, P.G), dim3(256), 0, st, (const float*)y, *ps, partial,
                       P.N, P.H, P.W, P.gstride, P.tpg + WN_ZBLOCKS, P.tpg);
    SRLZ_LAUNCHED();
  }
  return 0;
}

extern "C" int srlz_conv64_wino_bwd_data_rows(const srlz_conv64_desc* d) {
  WinoProg P;
  if (wino_program(&P, d)) return -1;
  return P.G * (P.tpg + WN_ZBLOCKS);
}

extern "C" int srlz_conv64_wino_fwd(const float* x, const float* upack, const float* bias, float* y, float* stats_partial,
                                    const float* x_bnp, const srlz_conv64_desc* d, srlz_stream_t stream) {
  SRLZ_REQUIRE(x && upack && y, SRLZ_ERR_NULL, "conv64_wino_fwd: null pointer");
  WinoProg P;
  SRLZ_REQUIRE(wino_program(&P, d) == 0, SRLZ_ERR_BAD_DESC,
               "conv64_wino_fwd: 3x3 stride 1 pad 1 on even sizes only (ask srlz_conv64_wino_supported)");
  return wino_launch(x, upack, bias, y, stats_partial, P, nullptr, x_bnp, stream);
}

extern "C" int srlz_conv64_wino_bwd_data(const float* dy, const float* upack_bwd, float* dx, const srlz_conv64_desc* d, srlz_stream_t stream) {
  SRLZ_REQUIRE(dy && upack_bwd && dx, SRLZ_ERR_NULL, "conv64_wino_bwd_data: null pointer");
  WinoProg P;
  SRLZ_REQUIRE(wino_program(&P, d) == 0, SRLZ_ERR_BAD_DESC,
               "conv64_wino_bwd_data: 3x3 stride 1 pad 1 on even sizes only (ask srlz_conv64_wino_supported)");
  return wino_launch(dy, upack_bwd, nullptr, dx, nullptr, P, nullptr, nullptr, stream);
}

extern "C" int srlz_conv64_wino_bwd_data_pool_sums(const float* dy, const float* upack_bwd, float* dx, const float* pooled,
                                                   const float* pool_bnp, const float* pool_y, const uint8_t* pool_argmax,
                                                   const srlz_pool_desc* pd, float* bn_bwd_partial, const srlz_conv64_desc* d,
                                                   srlz_stream_t stream) {
  SRLZ_REQUIRE(dy && upack_bwd && dx && pooled && pool_bnp && pool_y && pool_argmax && pd && bn_bwd_partial, SRLZ_ERR_NULL,
               "conv64_wino_bwd_data_pool_sums: null pointer");
  WinoProg P;
  SRLZ_REQUIRE(wino_program(&P, d) == 0, SRLZ_ERR_BAD_DESC,
               "conv64_wino_bwd_data_pool_sums: 3x3 stride 1 pad 1 on even sizes only (ask srlz_conv64_wino_supported)");
  // dx (this layer's input gradient) is the gradient of the pooled map pd describes: same images, same spatial size, NHWC
  SRLZ_REQUIRE(pd->n == d->n && pd->hp == d->hi && pd->wp == d->wi && !pd->out_nchw && (pd->groups > 1 ? pd->groups : 1) == P.G,
               SRLZ_ERR_BAD_DESC, "conv64_wino_bwd_data_pool_sums: the pooled map [%d,%d,%d] is not this layer's input [%d,%d,%d]", pd->n, pd->hp,
               pd->wp, d->n, d->hi, d->wi);
  const WinoPoolSum ps = {pooled, pool_bnp, pool_y, pool_argmax, (long long)(pd->n / P.G) * pd->h * pd->w * 64, pd->h, pd->w, pd->pool_pad};
  return wino_launch(dy, upack_bwd, nullptr, dx, bn_bwd_partial, P, &ps, nullptr, stream);
}

extern "C" size_t srlz_conv64_wino_bwd_weight_workspace(const srlz_conv64_desc* d) {
  WinoWgradProg P;
  int grid;
  if (wino_wgrad_program(&P, d, &grid)) return 0;
  return (size_t)grid * 8 * 4096 * sizeof(float) + (size_t)WG_CHUNKS * 16 * 4096 * sizeof(double);
}

extern "C" int srlz_conv64_wino_bwd_weight(const float* x, const float* dy, float* dw_ref, void* ws, size_t ws_bytes,
                                           const srlz_conv64_desc* d, srlz_stream_t stream) {
  SRLZ_REQUIRE(x && dy && dw_ref && ws, SRLZ_ERR_NULL, "conv64_wino_bwd_weight: null pointer");
  WinoWgradProg P;
  int grid;
  SRLZ_REQUIRE(wino_wgrad_program(&P, d, &grid) == 0, SRLZ_ERR_BAD_DESC,
               "conv64_wino_bwd_weight: 3x3 stride 1 pad 1 on even sizes only (ask srlz_conv64_wino_supported)");
  const size_t part_bytes = (size_t)grid * 8 * 4096 * sizeof(float);
  SRLZ_REQUIRE(ws_bytes >= part_bytes + (size_t)WG_CHUNKS * 16 * 4096 * sizeof(double), SRLZ_ERR_WORKSPACE,
               "conv64_wino_bwd_weight: workspace too small (%zu bytes)", ws_bytes);
  hipStream_t st = as_stream(stream);
  const size_t lds = (size_t)4 * WG_OP * 4;
  SRLZ_MAX_LDS(conv64_wino_wgrad_kernel, lds);
  hipLaunchKernelGGL(conv64_wino_wgrad_kernel, dim3(grid), dim3(256), lds, st, x, dy, (float*)ws, P);
  SRLZ_LAUNCHED();
  double* mid = (double*)((char*)ws + part_bytes);
  const int npair = grid >> 1, per = (npair + WG_CHUNKS - 1) / WG_CHUNKS;
  hipLaunchKernelGGL(conv64_wino_wgrad_reduce_a, dim3(65536 / 256, WG_CHUNKS), dim3(256), 0, st, (const float*)ws, mid, grid, per);
  SRLZ_LAUNCHED();
  hipLaunchKernelGGL(conv64_wino_wgrad_reduce_b, dim3(256), dim3(256), 0, st, (const double*)mid, WG_CHUNKS, dw_ref);
  SRLZ_LAUNCHED();
  return 0;
}
