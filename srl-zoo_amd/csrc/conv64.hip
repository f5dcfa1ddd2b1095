// conv64.hip — 3x3, 64->64 channel convolutions and transposed convolutions (stride 1 / 2) as fp32-MFMA implicit GEMM.
//
// Replaces (forward + autograd backward) nn.Conv2d / nn.ConvTranspose2d of /root/reference/models/models.py:54,59
// (conv3x3 s1/s2) and :66,70,74,78 (ConvTranspose2d(64,64,3,stride=2) x4).
//
// One formulation covers every case ("virtual-grid program"):
//   * a virtual grid of PH x PW positions per image, flattened to q = (n*PH + a)*PW + b;
//   * source class c = (cy,cx):  S_c(q) = src[n, a*ss+cy, b*ss+cx, :]  (64 floats) if in bounds, else 0;
//   * dest   class d = (dy,dx):  D_d(q) -> dst[n, a*ds+dy, b*ds+dx, :] if in bounds, else discarded;
//   * D_d(q) = sum over taps t of  S_{c_t}(q + off_t) . W[w_t]   (W[w] is a 64x64 slab of the 3x3 kernel).
// PH/PW carry one spare (zero) row/column so that a flat offset never wraps onto real data (see build_program).
//   conv s1         : 1 src class, 1 dst class, 9 taps          (conv2 fwd, conv2 dgrad)
//   conv s2 (gather): 4 src classes (input parity), 1 dst class  (conv3 fwd, ConvT dgrad)
//   convT s2 (scatter): 1 src class, 4 dst classes (output parity, 4/2/2/1 taps) (ConvT fwd, conv3 dgrad)
// GEMM view per tile: M = 128 grid positions, N = 64 output channels, K = 64 input channels per tap.
//
// Kernel structure (fwd): 256 threads = 4 waves, wave w owns rows [32w,32w+32) x 64 columns = two 32x32
// v_mfma_f32_32x32x2_f32 accumulators.  The source rows the tile touches (128 + halo) are staged ONCE in LDS and
// re-used by all taps; the weight slab of the current tap (16 KB) is staged per tap.  Both operands are read with
// ds_read_b128: lanes 0-31 take k = 8c..8c+3 and lanes 32-63 k = 8c+4..8c+7 of a chunk (MFMA k-order is free as
// long as A and B agree), so one 16-byte read feeds four MFMAs.  Rows are 256 B; the 16-byte slot index is XORed
// with (row & 15) so a 16-lane ds_read_b128 group (consecutive rows, same k) covers all 64 banks.
// LDS: (128+116)*256 + 16384 + 1024 + 512 + 1024 = 81408 B for conv2 (source rows, slab, the tile's row table, one word per output
// position, fused-operand coefficients) -> 2 workgroups per CU (81920 B each), which is what hides the staging.
// Epilogue: accumulators leave through a wave-private 4 KB transposition buffer (the wave's own quarter of the idle slab), so
// that every global store is 16 bytes per lane (flush16).  See DESIGN.md 5.2 for what the generated ISA taught about this file.
#include "common.h"
#include <stdlib.h>

namespace {

#ifndef SRLZ_BATCH_FWD
#define SRLZ_BATCH_FWD 16
#endif
#ifndef SRLZ_BATCH_BWD
#define SRLZ_BATCH_BWD 12
#endif
constexpr int TM = 128;       // grid positions per forward tile
constexpr int BATCH_FWD = SRLZ_BATCH_FWD;  // rows (of 16 lanes) a thread requests per round trip of a plain / forward-fused staging
constexpr int NTAPS = 9;

struct ConvProg {
  int N, PH, PW, PHW, total_q;  // N / total_q: images / grid positions of ONE BatchNorm group
  int G, tpg;                   // groups batched along the image axis (images [g*N, (g+1)*N)); forward tiles per group
  long long src_gstride, dst_gstride;  // floats between two groups' images in src / dst
  int ss, Hs, Ws;  // source: stride, image dims
  int ds, Hd, Wd;  // dest
  int tsrc[NTAPS], tdst[NTAPS], toff[NTAPS], tw[NTAPS];
  int tp[NTAPS + 2];  // the same per tap in one word: toff (low 16 bits, signed) | tw << 16 | tsrc << 20 | tdst << 22; two zero words
                      // behind the last tap (conv64_fwd_kernel fetches two taps ahead)
  int min_off, span;
  int s2;          // 1 if taps are grouped {4,2,2,1} by class, 0 if a single group of 9
  int dbg;         // ablation switches for tools/kbench.py (env SRLZ_ABLATE): 1 skip A staging, 2 skip epilogue
  unsigned mPHW, mPW;  // q / PHW and r / PW for 0 <= q, r < 2^31 as (__umulhi(q, m) >> sh): a run-time integer division is ~20
  int sPHW, sPW;       // VALU instructions and a reciprocal the compiler keeps in a register for the whole kernel (fastdiv)
};

// Granlund-Montgomery division by an invariant for 31-bit dividends: l = ceil(log2 d), m = floor(2^(31+l) / d) + 1 (< 2^32),
// q / d == umulhi(q, m) >> (l - 1) for every 0 <= q < 2^31 (d >= 2).
static void fastdiv_init(unsigned d, unsigned* m, int* sh) {
  int l = 0;
  while ((1u << l) < d) ++l;
  if (l == 0) l = 1;  // d == 1: m = 2^32 does not fit; callers have d >= 2 (checked in build_program)
  *m = (unsigned)((((unsigned long long)1 << (31 + l)) / d) + 1);
  *sh = l - 1;
}
__device__ __forceinline__ int fastdiv(int q, unsigned m, int sh) { return (int)(__umulhi((unsigned)q, m) >> sh); }

struct Axis {
  int cls[3], d[3];  // per kernel index ky: class (parity) and grid offset
};

// GATHER: out[o] = sum_k in[o*s - p + k];  SCATTER: out[o] = sum_{k: (o+p-k)%s==0} in[(o+p-k)/s]
static int floordiv(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }

// Build the program.  gather != 0: conv-like (dst is the low-res / same-res side); else convT-like.
static int build_program(ConvProg* P, int gather, int stride, int pad, int N, int Hs, int Ws, int Hd, int Wd, int G = 1) {
  if (stride != 1 && stride != 2) return -1;
  if (G < 1 || N % G != 0) return -1;
  N /= G;  // the virtual grid covers ONE group; a tile / chunk never straddles two groups (see conv64_fwd_kernel)
  P->N = N; P->Hs = Hs; P->Ws = Ws; P->Hd = Hd; P->Wd = Wd;
  P->G = G;
  P->src_gstride = (long long)N * Hs * Ws * 64;
  P->dst_gstride = (long long)N * Hd * Wd * 64;
  int kcls[3], kd[3];  // identical for both axes (square kernel, same stride/pad)
  if (gather) {
    P->ss = stride; P->ds = 1;
    for (int k = 0; k < 3; ++k) {
      int t = k - pad;
      int c = ((t % stride) + stride) % stride;
      kcls[k] = c; kd[k] = (t - c) / stride;
    }
  } else {
    P->ss = 1; P->ds = stride;
    // for dest class py: k valid iff (py + pad - k) % stride == 0; exactly one py per k
    for (int k = 0; k < 3; ++k) {
      int py = (((k - pad) % stride) + stride) % stride;
      kcls[k] = py; kd[k] = floordiv(py + pad - k, stride);
      if ((py + pad - k) != kd[k] * stride) return -1;
    }
  }
  // grid extents per axis
  auto extent = [&](int Hsrc, int Hdst) {
    int mx = 0, mn = 0;
    for (int k = 0; k < 3; ++k) { if (kd[k] > mx) mx = kd[k]; if (kd[k] < mn) mn = kd[k]; }
    int n_out = (Hdst + P->ds - 1) / P->ds;
    int n_src = (Hsrc + P->ss - 1) / P->ss;
    int ph = n_out + mx; if (n_src > ph) ph = n_src;
    if (mn < 0) {
      // the last row must read as zero for every source class (it is what offset -1 wraps onto)
      bool nonzero = false;
      for (int c = 0; c < P->ss; ++c) if (P->ss * (ph - 1) + c < Hsrc) nonzero = true;
      if (nonzero) ph += 1;
      if (mn < -1) return -1;
    }
    return ph;
  };
  P->PH = extent(Hs, Hd);
  P->PW = extent(Ws, Wd);
  if (P->PH <= 0 || P->PW <= 0) return -1;
  P->PHW = P->PH * P->PW;
  if (P->PW < 2) return -1;
  fastdiv_init((unsigned)P->PHW, &P->mPHW, &P->sPHW);
  fastdiv_init((unsigned)P->PW, &P->mPW, &P->sPW);
  long long tq = (long long)N * P->PHW;
  if (tq > 0x7fffff00LL) return -1;
  P->total_q = (int)tq;
  P->tpg = (P->total_q + TM - 1) / TM;
  // taps, grouped by class, groups in descending size (4,2,2,1 for stride 2)
  int nclass = stride * stride;
  int order[4] = {0, 1, 2, 3}, cnt[4] = {0, 0, 0, 0};
  for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx) cnt[kcls[ky] * stride + kcls[kx]]++;
  for (int i = 0; i < nclass; ++i) for (int j = i + 1; j < nclass; ++j)
    if (cnt[order[j]] > cnt[order[i]]) { int t = order[i]; order[i] = order[j]; order[j] = t; }
  if (nclass == 4 && cnt[order[1]] == cnt[order[2]]) {
    // Of the two 2-tap classes the one whose taps reach LESS far ahead comes first (round 6).  conv64_bwd_fused_kernel keeps the
    // classes of a tile in two alternating LDS buffers (classes 0 / 2 in one, 1 / 3 in the other): with the {0, +1} class second
    // and the 1-tap class fourth the second buffer needs TM + 1 rows instead of TM + PW, which is what makes room for a second
    // weight slab (one barrier per tap instead of two).  Every kernel reads the order from the program: results differ from the
    // other order by the summation order of the taps only.
    int reach[4] = {0, 0, 0, 0};
    for (int c = 0; c < 4; ++c)
      for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx)
        if (kcls[ky] * stride + kcls[kx] == c) { const int o = kd[ky] * P->PW + kd[kx]; if (o > reach[c]) reach[c] = o; }
    if (reach[order[2]] < reach[order[1]]) { int t = order[1]; order[1] = order[2]; order[2] = t; }
  }
  int nt = 0;
  P->min_off = 0; int max_off = 0;
  for (int g = 0; g < nclass; ++g) {
    int cy = order[g] / stride, cx = order[g] % stride;
    for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx) {
      if (kcls[ky] != cy || kcls[kx] != cx) continue;
      int cls2 = cy * 2 + cx;  // class encoding used by the kernels: (cy<<1)|cx
      P->tsrc[nt] = gather ? cls2 : 0;
      P->tdst[nt] = gather ? 0 : cls2;
      P->toff[nt] = kd[ky] * P->PW + kd[kx];
      P->tw[nt] = ky * 3 + kx;
      if (P->toff[nt] < P->min_off) P->min_off = P->toff[nt];
      if (P->toff[nt] > max_off) max_off = P->toff[nt];
      nt++;
    }
  }
  if (nt != NTAPS) return -1;
  for (int t = 0; t < NTAPS; ++t) {
    if (P->toff[t] < -32768 || P->toff[t] > 32767) return -1;
    P->tp[t] = (P->toff[t] & 0xffff) | (P->tw[t] << 16) | (P->tsrc[t] << 20) | (P->tdst[t] << 22);
  }
  P->tp[NTAPS] = P->tp[NTAPS + 1] = 0;
  P->span = max_off - P->min_off;
  P->s2 = (stride == 2);
  { const char* e = getenv("SRLZ_ABLATE"); P->dbg = e ? atoi(e) : 0; }
  if (P->s2 && !(cnt[order[0]] == 4 && cnt[order[1]] == 2 && cnt[order[2]] == 2 && cnt[order[3]] == 1)) return -1;
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Tile staging: rows [qstart, qstart+nrows) of class `cls` of an NHWC/64 tensor -> LDS (256 B per row).
// 16 lanes fetch one row (256 contiguous bytes); out-of-range rows are zero-filled.
// ---------------------------------------------------------------------------------------------------------------
// How raw rows become the operand (OpFuse):
//  * bnp != NULL, y == NULL: the source is the RAW output of a convolution and the consumer wants relu(batchnorm(.)):
//    the affine (scale = bnp[128..], shift = bnp[192..]) and the ReLU are applied to in-bounds rows on the way into
//    LDS, so the activated tensor is never materialised in HBM (padding rows stay exactly zero).
//  * y != NULL: the source is dA = d(loss)/d(relu(bn(y))) and the consumer wants dy = d(loss)/dy, the BatchNorm + ReLU
//    BACKWARD: dy = scale*(dA*[bn(y)>0] - S1/count - xhat*S2/count) = scale*dz - (c0 + c1*y), rebuilt from (dA, y) and
//    the two per-channel sums of srlz_bn_relu_bwd_sums, so dy is never materialised either.
//    With dy_out != NULL every rebuilt element whose grid position lies in this tile's own range [core_lo, core_lo+TM)
//    is also written to dy_out (each element exactly once across the launch): the data-gradient kernel materialises
//    the tensor for the weight-gradient kernel as a by-product of its staging, replacing the separate apply pass.
struct OpFuse {
  const float* bnp;
  const float* y;
  const float* sums;
  float inv_count;
  int training;
  float* dy_out;
};
#define SRLZ_NO_FUSE OpFuse{nullptr, nullptr, nullptr, 0.f, 0, nullptr}

// Epilogue option of the data-gradient launches whose output is the gradient of a POOLED map (conv2 -> d pooled1, conv3 -> d pooled2;
// conv64_dgrad_poolsum_kernel<1 / 2>): instead of BatchNorm-forward statistics the tile's partial record receives the two
// BatchNorm-BACKWARD sums of the block that produced the pooled map,  sum dz  and  sum dz*xhat  with dz = d pooled where pooled > 0
// (the gradient lives at the window's argmax, and the pooled value IS relu(bn(y)) there: xhat follows from it) — what
// bn_relu_pool_bwd_reduce computes in a pass of its own over (d pooled, pooled, argmax).  y / argmax are only touched for channels
// whose BatchNorm scale is exactly 0 (xhat cannot be recovered from the pooled value there).
struct PoolSum {
  const float* pooled;    // [N,Hd,Wd,64] like the launch's dst; NULL = off
  const float* bnp;       // records of the pooled block's BatchNorm (256 floats per group)
  const float* y;         // raw convolution output under the pooling [N,H,W,64]
  const uint8_t* argmax;  // [N,Hd,Wd,64]
  long long y_gstride;    // floats between two groups' images in y
  int H, W, pad;
};
#define SRLZ_NO_POOLSUM PoolSum{nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0}

// The records of BatchNorm group `grp` (bnp: 256 floats per group, sums: 128) and the group's slice of the tensors that are
// indexed like the staged source (y, dy_out): `goff` floats further.
__device__ __forceinline__ OpFuse fuse_for_group(OpFuse f, int grp, long long goff) {
  if (f.bnp) f.bnp += grp * 256;
  if (f.sums) f.sums += grp * 128;
  if (f.y) f.y += goff;
  if (f.dy_out) f.dy_out += goff;
  return f;
}

// Reciprocals of the virtual grid's PH * PW and PW (fastdiv) for rows64_load.
struct GridDiv { unsigned mPHW, mPW; int sPHW, sPW; };
__device__ __forceinline__ GridDiv grid_div(const ConvProg& P) { return GridDiv{P.mPHW, P.mPW, P.sPHW, P.sPW}; }

template <bool SWZ, int BATCH = 8, int NTHREADS = 256, bool BWD = false>
__device__ __forceinline__ void stage_rows(float* __restrict__ lds, const float* __restrict__ src, int H, int W,
                                           int stride, int cls, int PW, int PH, int total_q, int qstart,
                                           int nrows, const OpFuse f = SRLZ_NO_FUSE, int core_lo = 0, int core_n = 0,
                                           int cstride = 64, int coff = 0, const float* __restrict__ lrec = nullptr) {
  // lrec != NULL: the per-channel coefficients of the fused operand (scale, shift, c0, c1: 4 x 64 floats) have been put in LDS by
  // the caller, once per workgroup — read from the global records at every call they cost two to three dependent L2 round trips in
  // front of each staging (four stagings per tile for the stride-2 gather programs)
  // cstride / coff: the tensor has `cstride` channels per pixel and this call stages channels [coff, coff + 64) (convN_*)
  const float* __restrict__ bnp = f.bnp;
  // Loads are issued in batches of 8 rows per thread before any LDS store, so the HBM/L2 latency is paid once per
  // batch instead of once per row; (n, a, b) of a thread's rows are advanced incrementally (rows are NTHREADS/16 apart), the
  // only integer divisions are the two for its first row.
  const int t = threadIdx.x;
  const int slot = t & 15;
  const int cy = cls >> 1, cx = cls & 1;
  const int PHW = PH * PW;
  constexpr int RP = NTHREADS / 16;  // rows per pass
  // (true divisions: with the uniform divisors hipcc keeps one reciprocal per kernel, and fastdiv here measured 0.5 % SLOWER in
  // conv64_fwd_kernel — unlike in rows64_load, where it is worth 1.5 % of the weight-gradient ring)
  const int sa = RP / PW, sb = RP - sa * PW;
  // shift by one image so the first rows of the first tile (negative q) stay non-negative: n1 = n + 1
  const int qq = qstart + (t >> 4) + PHW;
  int n1 = qq / PHW;
  int rem = qq - n1 * PHW;
  int a = rem / PW;
  int b = rem - a * PW;
  const int N1max = total_q / PHW;  // images
  f32x4 sc4 = {1.f, 1.f, 1.f, 1.f}, sh4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
  if (lrec) {
    sc4 = *(const f32x4*)(lrec + slot * 4); sh4 = *(const f32x4*)(lrec + 64 + slot * 4);
    if (BWD) { c0 = *(const f32x4*)(lrec + 128 + slot * 4); c1 = *(const f32x4*)(lrec + 192 + slot * 4); }
  } else if (bnp) { sc4 = *(const f32x4*)(bnp + 128 + slot * 4); sh4 = *(const f32x4*)(bnp + 192 + slot * 4); }
  if (!lrec && BWD && f.training) {
    const f32x4 mean = *(const f32x4*)(bnp + slot * 4), invstd = *(const f32x4*)(bnp + 64 + slot * 4);
    const f32x4 m1 = *(const f32x4*)(f.sums + slot * 4), m2 = *(const f32x4*)(f.sums + 64 + slot * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      c1[e] = sc4[e] * invstd[e] * m2[e] * f.inv_count;
      c0[e] = sc4[e] * m1[e] * f.inv_count - c1[e] * mean[e];
    }
  }
  for (int base = t >> 4; base < nrows; base += RP * BATCH) {
    f32x4 v[BATCH], yv[BWD ? BATCH : 1];
    unsigned offs[BWD ? BATCH : 1];  // float offsets of the rows (a group's tensor has < 2^32 floats: checked by the host)
    unsigned okmask = 0;
#pragma unroll
    for (int j = 0; j < BATCH; ++j) {
      const int y = a * stride + cy, x = b * stride + cx;
      const bool ok = base + RP * j < nrows && n1 >= 1 && n1 <= N1max && y < H && x < W;
      okmask |= (ok ? 1u : 0u) << j;
      // Branch-free: a padding row reads pixel 0 (always valid) and is zeroed when it is consumed.  With a load inside a branch the
      // compiler cannot tell, after the join, which loads are still in flight; every later first write of a register such a load
      // once targeted then gets "s_waitcnt vmcnt(0)" — in the caller that was the first MFMA of each tap, i.e. the prefetch of the
      // next weight slab was waited for before the MFMAs it is meant to hide behind.
      const size_t off = (ok ? ((size_t)((n1 - 1) * H + y) * W + x) * cstride : (size_t)0) + coff + slot * 4;
      v[j] = *(const f32x4*)(src + off);
      if (BWD) { yv[j] = *(const f32x4*)(f.y + off); offs[j] = (unsigned)off; }
      b += sb; a += sa;
      if (b >= PW) { b -= PW; ++a; }
      if (a >= PH) { a -= PH; ++n1; }
    }
    // All loads of the batch are waited for HERE, once, in straight-line code: the rows below are consumed inside branches, and
    // after a branch join the compiler no longer knows which loads have landed.  It then protects every later re-use of one of
    // these registers with "s_waitcnt vmcnt(0)": in the fused data-gradient that wait sat behind every dy_out store (one HBM
    // round trip per store), in the caller's tap loop in front of the first MFMA of every tap (defeating the slab prefetch).
#pragma unroll
    for (int j = 0; j < BATCH; ++j) {
      asm volatile("" : "+v"(v[j]));
      if (BWD) asm volatile("" : "+v"(yv[BWD ? j : 0]));
    }
#pragma unroll
    for (int j = 0; j < BATCH; ++j) {
      const int R = base + RP * j;
      if (R < nrows) {
        if (!((okmask >> j) & 1u)) v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        else if (bnp) {
          if (BWD) {
            // (explicit fused multiply-adds: the same roundings in every kernel that rebuilds dy — as whole-vector operations, which
            // hipcc issues as v_pk_fma_f32, two elements per instruction slot: every instruction next to the MFMAs costs matrix time)
            const f32x4 z4 = __builtin_elementwise_fma(yv[j], sc4, sh4), t4 = __builtin_elementwise_fma(c1, yv[j], c0);
            f32x4 dz4;
#pragma unroll
            for (int e = 0; e < 4; ++e) dz4[e] = z4[e] > 0.f ? v[j][e] : 0.f;
            v[j] = __builtin_elementwise_fma(sc4, dz4, -t4);
            if (f.dy_out && (unsigned)(R - core_lo) < (unsigned)core_n) *(f32x4*)(f.dy_out + (size_t)offs[j]) = v[j];
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float z = v[j][e] * sc4[e] + sh4[e]; v[j][e] = z > 0.f ? z : 0.f; }
          }
        }
        const int sl = SWZ ? (slot ^ (R & 15)) : slot;
        *(f32x4*)(lds + R * 64 + sl * 4) = v[j];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The row table of a forward tile.  stage_rows walks (image, row, column) per staged row and thread: with the bounds tests, the
// pixel offset and the 64-bit address that is ~30 vector-ALU instructions per row (two of them quarter-rate 64-bit multiply-adds),
// 16 rows per thread and tile — a sixth of everything conv64_fwd_kernel issues next to its MFMAs, each costing matrix time
// (DESIGN.md 5.3).  Here every thread decomposes ONE row of the tile's range [qstart, qstart + nrows) into a table in LDS, once per
// tile:   entry = pixel index of the row's class-(0,0) source pixel << 4 | bit k: source class k = (cy << 1) | cx is inside the image
// (0: no class is), and a staging is then, per row: one quarter of a ds_read_b128, a bit-field extract, an add-shift, two ANDs and the
// address add.  The four stagings of a stride-2 gather tile share the table.
// Layout: entry of row R at (R & 15) * tpa + (R >> 4): the rows of thread t (R = (t >> 4) + 16 j) are consecutive words.
// tpa = passes over the tile's rows rounded up to the batch (entries past nrows are 0: such a row reads pixel 0 and is dropped).
// ---------------------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ int rowtab_passes(int nrows) { return ((nrows + 15) / 16 + SRLZ_BATCH_FWD - 1) / SRLZ_BATCH_FWD * SRLZ_BATCH_FWD; }

__device__ __forceinline__ void rowtab_build(unsigned* __restrict__ tab, int tpa, const ConvProg& P, int qstart, int nrows) {
  for (int R = threadIdx.x; R < 16 * tpa; R += blockDim.x) {
    unsigned e = 0;
    if (R < nrows) {
      const int qq = qstart + R + P.PHW;  // shifted by one image: the first rows of the first tile (negative q) stay non-negative
      const int n1 = fastdiv(qq, P.mPHW, P.sPHW);
      const int rem = qq - n1 * P.PHW;
      const int a = fastdiv(rem, P.mPW, P.sPW);
      const int y0 = a * P.ss, x0 = (rem - a * P.PW) * P.ss;
      if ((unsigned)(n1 - 1) < (unsigned)P.N) {
        unsigned f = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) f |= (y0 + (k >> 1) < P.Hs && x0 + (k & 1) < P.Ws) ? 1u << k : 0u;
        e = ((unsigned)(((n1 - 1) * P.Hs + y0) * P.Ws + x0) << 4) | f;
      }
    }
    tab[(R & 15) * tpa + (R >> 4)] = e;
  }
}

// Rows of source class `cls` -> LDS (swizzled), through the table.  lrec != NULL: relu(batchnorm(.)) on the way in (scale, shift in LDS).
// cshift / coff: the tensor has 2^cshift channels per pixel and channels [coff, coff + 64) are staged (convN_fwd_kernel; the 64-channel
// kernels pass the defaults, which fold to the constants they had)
template <int BATCH>
__device__ __forceinline__ void stage_rows_tab(float* __restrict__ lds, const float* __restrict__ src,
                                               const unsigned* __restrict__ tab, int tpa, int cls, int W, int nrows,
                                               const float* __restrict__ lrec, int cshift = 6, int coff = 0) {
  const int t = threadIdx.x;
  const int slot = t & 15, r = t >> 4;
  const unsigned delta = (unsigned)((cls >> 1) * W + (cls & 1));
  const float* __restrict__ base = src + coff + slot * 4;
  f32x4 sc4 = {1.f, 1.f, 1.f, 1.f}, sh4 = {0.f, 0.f, 0.f, 0.f};
  if (lrec) { sc4 = *(const f32x4*)(lrec + slot * 4); sh4 = *(const f32x4*)(lrec + 64 + slot * 4); }
  for (int j0 = 0; j0 < tpa; j0 += BATCH) {
    f32x4 v[BATCH];
    unsigned okmask = 0;
#pragma unroll
    for (int jj = 0; jj < BATCH; jj += 4) {  // (four entries at a time: all sixteen up front cost 12 registers the pooled-block kernel lacks)
      const uint4 q = *(const uint4*)(tab + r * tpa + j0 + jj);
      const unsigned e[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int j = jj + i;
        // branch-free (see stage_rows): m = all ones where the row's pixel of this class exists; any other row reads pixel 0
        const unsigned m = (unsigned)__builtin_amdgcn_sbfe((int)e[i], (unsigned)cls, 1u);
        const unsigned off = (((e[i] >> 4) + delta) << cshift) & m;  // floats (a group's tensor has < 2^32: checked by the host)
        v[j] = *(const f32x4*)(base + off);
        okmask |= m & (1u << j);
      }
    }
    // all loads of the batch are waited for here, once, in straight-line code (see stage_rows)
#pragma unroll
    for (int j = 0; j < BATCH; ++j) asm volatile("" : "+v"(v[j]));
#pragma unroll
    for (int j = 0; j < BATCH; ++j) {
      const int R = r + 16 * (j0 + j);
      if (R < nrows) {
        if (!((okmask >> j) & 1u)) v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        else if (lrec) {
#pragma unroll
          for (int e2 = 0; e2 < 4; ++e2) { const float z = v[j][e2] * sc4[e2] + sh4[e2]; v[j][e2] = z > 0.f ? z : 0.f; }
        }
        *(f32x4*)(lds + R * 64 + ((slot ^ (R & 15)) << 2)) = v[j];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Forward / data-gradient kernel.  NW = waves per workgroup: 4 (wave = 32 rows x 64 columns, two accumulators) or
// 8 (wave = 32 rows x 32 columns, one accumulator; twice the waves per SIMD to hide barriers, LDS and staging latency).
// ---------------------------------------------------------------------------------------------------------------
// BWD = true is the data-gradient launch whose operand is rebuilt from (dA, y) by the fused BatchNorm+ReLU backward (and
// optionally stored): a separate instantiation, so the plain kernel keeps its register budget and shows up under its own
// name in rocprof.
// PSUM: 0 = statistics of a forward launch; 1 / 2 = the pooled-block epilogue (PoolSum) of a data gradient with ONE destination class
// (its pooled values are requested before the first tap and travel under the whole tile) / with several (requested inside every
// flush: the small stride-2 layer).
// (The body is shared by two kernel symbols: conv64_fwd_kernel<NW, BWD> — PSUM = 0, what rocprof has listed since round 1 — and
// conv64_dgrad_poolsum_kernel<PSUM>.)
template <int NW, bool BWD, int PSUM>
__device__ __forceinline__ void conv64_fwd_body(const float* __restrict__ src, const float* __restrict__ wpack,
                                                const float* __restrict__ bias, float* __restrict__ dst,
                                                float* __restrict__ stats_partial, const ConvProg& P, int ntiles,
                                                const OpFuse& src_fuse_all, const PoolSum& ps) {
  static_assert(NW == 4, "8 waves of 32 x 32 measured within +-3 % in rounds 1-3 and are not kept");
  static_assert(PSUM == 0 || !BWD, "the pooled-block epilogue exists for the plain kernel");
  constexpr int NT = NW * 64;      // threads
  constexpr int NACC = 8 / NW;     // 32-column tiles per wave
  // rows (of 16 lanes) a thread requests per HBM round trip of the source staging: a stride-1 tile (244 rows = 15.25 passes) or a
  // gather class (185 rows = 11.6 passes) in ONE batch instead of two
  constexpr int BATCH_BWD = SRLZ_BATCH_BWD;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* As = (float*)smem;                 // (TM + span) x 64, swizzled
  float* Bs = As + (TM + P.span) * 64;      // 64 x 64 weight slab of the current tap, pre-swizzled in global
  const int tpa = rowtab_passes(TM + P.span);
  unsigned* rowtab = (unsigned*)(Bs + 4096);  // [16][tpa]: the source-side row table (rowtab_build)
  int* rowinfo = (int*)(rowtab + 16 * tpa);   // [TM]: the destination side of a grid position — pixel index of its class-(0,0) output
                                              // << 2 | bit 0: the row below exists | bit 1: the column to the right exists; -1 = outside

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wrow = wave & 3, wcol = wave >> 2;  // row group (32 rows), first column tile
  const int h = lane >> 5, l31 = lane & 31;
  // BatchNorm groups (P.G > 1: `obs || next_obs` of one training step batched along n): tiles [g*tpg, (g+1)*tpg) cover group g's
  // own virtual grid, so a tile never mixes two groups — its statistics partial belongs to one group, its fused operand
  // uses one group's BatchNorm record — and the launch is exactly the union of the G per-group launches.
  const int tile = xcd_remap(blockIdx.x, ntiles);
  const int grp = (P.G > 1) ? tile / P.tpg : 0;
  const int q0 = (tile - grp * P.tpg) * TM;
  src += grp * P.src_gstride;
  dst += grp * P.dst_gstride;
  const OpFuse src_fuse = fuse_for_group(src_fuse_all, grp, grp * P.src_gstride);

  if (tid < TM) {
    const int q = q0 + tid;
    int ri = -1;
    if (q < P.total_q) {
      const int n = fastdiv(q, P.mPHW, P.sPHW);
      const int rem = q - n * P.PHW;
      const int a = fastdiv(rem, P.mPW, P.sPW);
      const int ya = a * P.ds, xb = (rem - a * P.PW) * P.ds;
      if (ya < P.Hd && xb < P.Wd) ri = (((n * P.Hd + ya) * P.Wd + xb) << 2) | (ya + 1 < P.Hd ? 1 : 0) | (xb + 1 < P.Wd ? 2 : 0);
    }
    rowinfo[tid] = ri;
  }
  if constexpr (!BWD) rowtab_build(rowtab, tpa, P, q0 + P.min_off, TM + P.span);
  // coefficients of a fused operand, once per workgroup (visible after the first tap's barrier, which precedes the first staging)
  float* frec = (float*)(rowinfo + TM);  // [4][64]: scale, shift, c0, c1
  if (src_fuse.bnp && tid >= NT - 64) {
    const int c = tid - (NT - 64);
    const float sc = src_fuse.bnp[128 + c], sh = src_fuse.bnp[192 + c];
    float c0 = 0.f, c1 = 0.f;
    if (BWD && src_fuse.training) {
      c1 = sc * src_fuse.bnp[64 + c] * src_fuse.sums[64 + c] * src_fuse.inv_count;
      c0 = sc * src_fuse.sums[c] * src_fuse.inv_count - c1 * src_fuse.bnp[c];
    }
    frec[c] = sc; frec[64 + c] = sh; frec[128 + c] = c0; frec[192 + c] = c1;
  }
  const float* lrec = src_fuse.bnp ? frec : nullptr;

  f32x16 acc[NACC];
  float bcol[NACC];
#pragma unroll
  for (int j = 0; j < NACC; ++j) bcol[j] = bias ? bias[(wcol * NACC + j) * 32 + l31] : 0.f;

  int cur_src = -1, cur_dst = -1;
  bool pz_new = false;  // (PSUM == 2) a destination class opened at this tap: its pooled values are still to be requested

  // PSUM: the pooled value z = relu(gamma * xhat + beta), so where it is positive xhat = (z - beta) / gamma = z * pA + pB with
  // pA = invstd / scale, pB = -(shift / scale + mean) * invstd — one fused multiply-add per element, for this lane's four channels of
  // the flush layout (4 * (lane & 15) ..); channels with scale == 0 (pz_zero, pthr = +inf) are summed by the cold loop of the flush
  f32x4 pA = {0.f, 0.f, 0.f, 0.f}, pB = pA, pthr = pA;
  unsigned pz_zero = 0;
  const float* __restrict__ ppool = nullptr;
  if constexpr (PSUM) {
    const float* __restrict__ rec = ps.bnp + grp * 256 + (lane & 15) * 4;
    const f32x4 mean = *(const f32x4*)rec;
    const f32x4 pinv = *(const f32x4*)(rec + 64), psh = *(const f32x4*)(rec + 192);
    const f32x4 psc = *(const f32x4*)(rec + 128);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      // |scale| tiny against |shift| (exactly 0 included): z - shift cancels — xhat = (z - shift) / scale ... would carry an error of
      // ~ eps * |shift| / |scale| — so such a channel takes the cold loop, which computes xhat from the convolution output itself
      const bool zero = fabsf(psc[e]) <= 1e-3f * fabsf(psh[e]) || psc[e] == 0.f;
      pz_zero |= (zero ? 1u : 0u) << e;
      pthr[e] = zero ? __builtin_inff() : 0.f;
      const float isc = zero ? 0.f : 1.f / psc[e];
      pA[e] = pinv[e] * isc;
      pB[e] = -(psh[e] * isc + mean[e]) * pinv[e];
    }
    ppool = ps.pooled + grp * P.dst_gstride;
  }

  // NW == 4: the accumulators leave through a 4 KB wave-private transposition buffer (this wave's part of the idle weight slab),
  // 16 tile rows at a time, so that every global store is a 16-byte one — lane (g = lane >> 4, slot = lane & 15) stores channels
  // [4*slot, 4*slot+4) of rows g, g+4, g+8, g+12 — instead of 32 dword-per-lane stores per flush (the dword path sustains ~5 B per
  // cycle and CU, which bounds the scatter programs: a ConvTranspose forward tile writes 128 KB).  The BatchNorm partials are taken
  // in the same layout (s4 / q4).  Rows 4..7 and 12..15 of the buffer hold their two 32-channel halves swapped, so that the lanes
  // h = 0 / 1 of one ds_write_b32 (rows 4 apart, same column) hit different banks.
  f32x4 s4 = {0.f, 0.f, 0.f, 0.f}, q4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 pz[PSUM ? 8 : 1];  // PSUM: the pooled values of this lane's eight rows of a flush (rows 16*(hk >> 2) + (lane >> 4) + 4*(hk & 3))
  auto pz_request = [&](int d) {  // branch-free, clamped: a row outside the tensor reads pixel 0 and is never used
    const int need = d >> 1 | (d & 1) << 1;  // destination class (dy, dx): the rowinfo bits it needs
    const unsigned ddelta = (unsigned)((d >> 1) * P.Wd + (d & 1));
#pragma unroll
    for (int hk = 0; hk < 8; ++hk) {  // (the flush works the row's position out again: one LDS word and four instructions, against
      // nine registers for the whole tile in a kernel that has none to spare)
      const int row = wrow * 32 + 16 * (hk >> 2) + (lane >> 4) + 4 * (hk & 3);
      const int ri = rowinfo[row];
      const bool ok = ri >= 0 && (ri & need) == need;
      const unsigned pix = ok ? (unsigned)(ri >> 2) + ddelta : 0u;
      pz[PSUM ? hk : 0] = *(const f32x4*)(ppool + (size_t)pix * 64 + (lane & 15) * 4);
    }
  };
  auto flush16 = [&](int d) {
    if constexpr (PSUM == 0) {  // (the ablation switches belong to the plain kernel)
      if (P.dbg & 2) {
        if (acc[0][0] + acc[NACC - 1][5] == 123.456f) dst[tid] = acc[0][1];
        return;
      }
    }
    const int need = d >> 1 | (d & 1) << 1;
    const unsigned ddelta = (unsigned)((d >> 1) * P.Wd + (d & 1));
    float* S = Bs + wave * 1024;
    const int eg = lane >> 4, eslot = lane & 15;
    // The pooled values were requested before the first tap of this destination class (PSUM == 1: the tile's only one) and have
    // travelled under its MFMAs; they are waited for once, in straight-line code, before the first conditional store
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        const int rowl = (rr & 3) + 8 * (rr >> 2) + 4 * h;
        const int swz = (rowl & 4) << 3;  // 32 for rows 4..7, 12..15
#pragma unroll
        for (int j = 0; j < NACC; ++j) S[rowl * 64 + ((32 * j + l31) ^ swz)] = acc[j][8 * half + rr] + bcol[j];
      }
      // (wave-private: the compiler's lgkmcnt wait orders the writes above before the reads below)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int rowl = eg + 4 * k;
        const int row = wrow * 32 + 16 * half + rowl;
        const f32x4 v = *(const f32x4*)(S + rowl * 64 + ((eslot ^ ((rowl & 4) << 1)) << 2));
        if constexpr (PSUM) {
          if (half == 0 && k == 0) {
#pragma unroll
            for (int hk = 0; hk < 8; ++hk) asm volatile("" : "+v"(pz[hk]));
          }
        }
        const int ri = rowinfo[row];
        const bool inside = ri >= 0 && (ri & need) == need;
        const size_t pixel = (unsigned)(ri >> 2) + ddelta;
        if (inside) {
          *(f32x4*)(dst + pixel * 64 + eslot * 4) = v;
          if constexpr (PSUM) {
            const f32x4 z = pz[half * 4 + k];
            f32x4 xh;
#pragma unroll
            for (int e = 0; e < 4; ++e) xh[e] = __builtin_fmaf(z[e], pA[e], pB[e]);
            // (pthr = 0, or +inf for a channel whose scale is 0: those are summed by the cold loop behind the stores — a load inside THIS
            // loop's branches would put a full vmcnt wait behind every store)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float dz = z[e] > pthr[e] ? v[e] : 0.f;
              s4[e] += dz; q4[e] += dz * xh[e];
            }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) { s4[e] += v[e]; q4[e] += v[e] * v[e]; }
          }
        }
      }
    }
    if constexpr (PSUM != 0) {
      if (pz_zero) {
        // BatchNorm scale (almost) 0 in one of this lane's channels — exactly 0: relu(bn(.)) is the constant max(shift, 0), every
        // window's first position is the argmax — xhat cannot be recovered from the pooled value: it is taken from the convolution
        // output under the recorded argmax, and the ReLU decision is the forward's own expression on that output.  Rare and slow on
        // purpose: d(pooled) is read back from where this lane has just stored it.
        const float* __restrict__ rec = ps.bnp + grp * 256 + eslot * 4;  // (the cold loop re-reads what it needs of the record)
        const f32x4 mean = *(const f32x4*)rec, pinv = *(const f32x4*)(rec + 64), psc = *(const f32x4*)(rec + 128),
                    psh = *(const f32x4*)(rec + 192);
#pragma unroll 1
        for (int hk = 0; hk < 8; ++hk) {
          const int ri = rowinfo[wrow * 32 + 16 * (hk >> 2) + eg + 4 * (hk & 3)];
          if (!(ri >= 0 && (ri & need) == need)) continue;
          const size_t pixel = (unsigned)(ri >> 2) + ddelta;  // (n * Hd + y) * Wd + x of the output this lane stored
          const int n = (int)pixel / (P.Hd * P.Wd);
          const int y = ((int)pixel - n * P.Hd * P.Wd) / P.Wd, x = (int)pixel - (n * P.Hd + y) * P.Wd;
          const uint32_t packed = *(const uint32_t*)(ps.argmax + (size_t)grp * P.dst_gstride + pixel * 64 + eslot * 4);
          const f32x4 v = *(const f32x4*)(dst + pixel * 64 + eslot * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if ((pz_zero >> e) & 1u) {
              const int a = (packed >> (8 * e)) & 0xff;
              const int iy = y * 2 - ps.pad + a / 3, ix = x * 2 - ps.pad + a % 3;
              const float vy = ps.y[grp * ps.y_gstride + ((size_t)(n * ps.H + iy) * ps.W + ix) * 64 + eslot * 4 + e];
              const float dz = vy * psc[e] + psh[e] > 0.f ? v[e] : 0.f;
              s4[e] += dz; q4[e] += dz * ((vy - mean[e]) * pinv[e]);
            }
        }
      }
    }
  };

  // The weight slab of tap t+1 is fetched into registers while tap t's MFMAs run, and written to LDS between the two
  // barriers of the next iteration: the L2 latency of the slab never sits between barriers.
  // Wave w moves the contiguous part [w, w+1) * 16 KB / NW of the slab (and nobody else's): between the barrier that ends a tap and
  // its own slab write a wave may therefore use that part of Bs as private scratch (flush16).
  constexpr int BV = 1024 / NT;  // 16-byte vectors of the 16 KB slab per thread
  const int bslot = wave * (BV * 64) + lane;
  f32x4 breg[BV];
  {
    const f32x4* wsrc = (const f32x4*)(wpack + (size_t)P.tw[0] * 4096);
#pragma unroll
    for (int i = 0; i < BV; ++i) breg[i] = wsrc[bslot + i * 64];
  }
  // The per-tap program words travel two taps ahead of their use (w0 = this tap, w1 = the next one, whose slab is requested
  // right after this tap's second barrier): fetched at their point of use, each costs a scalar-memory round trip in front of the
  // slab requests / the first LDS reads of every tap.
  int w0 = P.tp[0], w1 = P.tp[1];
  if constexpr (PSUM == 1) {  // the tile's one flush is known now: its pooled values travel under the whole tile
    __syncthreads();          // (rowinfo)
    pz_request(P.tdst[0]);
  }
  // Three taps per trip of the loop (the plain kernels; hipcc chose this by itself while the body was larger): 20 KB of code instead of
  // 50 KB unrolled nine times — the instruction cache is 64 KB for two CUs.  (Everything tap-dependent comes from the program words.)
  constexpr int TAP_UNROLL = BWD ? NTAPS : 3;
#pragma unroll TAP_UNROLL
  for (int ti = 0; ti < NTAPS; ++ti) {
    const int w2 = P.tp[ti + 2];
    const int tsrc = (w0 >> 20) & 3, tdst = (w0 >> 22) & 3;
    __syncthreads();  // all waves are done with the previous tap's Bs (and with As if it is about to be replaced)
    if (tdst != cur_dst) {
      if (cur_dst >= 0) flush16(cur_dst);
#pragma unroll
      for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
      if constexpr (PSUM == 2) pz_new = true;
      cur_dst = tdst;
    }
    if (tsrc != cur_src) {
      if (!(P.dbg & 1)) {
        if (BWD)
          stage_rows<true, (NW == 4 ? BATCH_BWD : 2), NT, true>(As, src, P.Hs, P.Ws, P.ss, tsrc, P.PW, P.PH, P.total_q,
                                                        q0 + P.min_off, TM + P.span, src_fuse, -P.min_off, TM, 64, 0, lrec);
        else
          stage_rows_tab<BATCH_FWD>(As, src, rowtab, tpa, tsrc, P.Ws, TM + P.span, lrec);
      }
      cur_src = tsrc;
    }
    {
      f32x4* wdst = (f32x4*)Bs;
#pragma unroll
      for (int i = 0; i < BV; ++i) wdst[bslot + i * 64] = breg[i];
    }
    if constexpr (PSUM == 2) {
      // Several destination classes: the pooled values of the class that opens at this tap, for the flush that will close it.
      // Requested inside that flush they were one exposed HBM round trip per class and tile; requested here — BEHIND the slab write,
      // whose wait for the slab registers would otherwise wait for these younger loads too — they travel under the class's MFMAs.
      if (pz_new) { pz_request(cur_dst); pz_new = false; }
    }
    __syncthreads();
    if (BWD ? ti + 1 < NTAPS : true) {  // (never a run-time condition: behind the join of a branch with loads in it hipcc waits for
      // everything at the next re-use of their registers — in the flush that was one full wait per store.  The plain kernels, whose
      // loop is not unrolled nine times, therefore request a slab behind the last tap as well: word NTAPS of the program is 0 = slab 0)
      const f32x4* wsrc = (const f32x4*)(wpack + (size_t)((w1 >> 16) & 15) * 4096);
#pragma unroll
      for (int i = 0; i < BV; ++i) breg[i] = wsrc[bslot + i * 64];
    }
    __builtin_amdgcn_sched_barrier(0);  // the slab requests go out HERE, ahead of the tap's MFMAs (the scheduler sinks them otherwise)
    const int R = wrow * 32 + l31 + (int)(short)(w0 & 0xffff) - P.min_off;
    // offset of this lane's first 16-byte slot of row R; slot (2kc + h) ^ (R & 15) of the swizzled row is that offset XOR
    // (kc << 5) bytes — R*256 has no bits below 8, 2kc + h = 2kc ^ h — so each k-chunk costs ONE v_xor with a literal
    int abase = (R * 64 + ((h ^ (R & 15)) << 2)) * 4;  // in BYTES (the XOR then is the whole address computation)
    asm volatile("" : "+v"(abase));                     // one opaque value: the compiler otherwise re-splits it into its parts
    const float* brow = Bs + (wcol * NACC * 32 + l31) * 64;  // column tile j is 32 rows (2048 floats) further
    const int bkey = lane & 15;
    f32x4 a = *(const f32x4*)((const char*)As + abase);
    f32x4 b[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j) b[j] = *(const f32x4*)(brow + j * 2048 + ((h ^ bkey) << 2));
#pragma unroll
    for (int kc = 0; kc < 8; ++kc) {
      f32x4 an = a, bn[NACC];
#pragma unroll
      for (int j = 0; j < NACC; ++j) bn[j] = b[j];
      if (kc < 7) {
        const int slot = (kc + 1) * 2 + h;
        an = *(const f32x4*)((const char*)As + (abase ^ ((kc + 1) << 5)));
#pragma unroll
        for (int j = 0; j < NACC; ++j) bn[j] = *(const f32x4*)(brow + j * 2048 + ((slot ^ bkey) << 2));
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the prefetch reads ahead of this chunk's MFMAs (hipcc sinks them otherwise)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], b[j][r], acc[j], 0, 0, 0);
      a = an;
#pragma unroll
      for (int j = 0; j < NACC; ++j) b[j] = bn[j];
    }
    w0 = w1; w1 = w2;
  }
  __syncthreads();  // every wave is done with the last tap's slab: Bs becomes scratch
  flush16(cur_dst);

  if (stats_partial && !(P.dbg & 2)) {
    // the 4 row groups g of a wave hold the same columns; then the 4 row groups of waves are combined through LDS
    __syncthreads();
    float* red = Bs;  // [4 row groups][128]
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s4[e] += __shfl_xor(s4[e], 16, 64); s4[e] += __shfl_xor(s4[e], 32, 64);
      q4[e] += __shfl_xor(q4[e], 16, 64); q4[e] += __shfl_xor(q4[e], 32, 64);
    }
    if (lane < 16) {
      *(f32x4*)(red + wrow * 128 + lane * 4) = s4;
      *(f32x4*)(red + wrow * 128 + 64 + lane * 4) = q4;
    }
    __syncthreads();
    if (tid < 128) {
      const float v = red[tid] + red[128 + tid] + red[256 + tid] + red[384 + tid];
      stats_partial[(size_t)tile * 128 + tid] = v;  // [0,64): sum, [64,128): sum of squares
    }
  }
}

template <int NW, bool BWD = false>
__global__ __launch_bounds__(NW * 64, NW / 2) void conv64_fwd_kernel(const float* __restrict__ src,
                                                                    const float* __restrict__ wpack,
                                                                    const float* __restrict__ bias,
                                                                    float* __restrict__ dst,
                                                                    float* __restrict__ stats_partial,
                                                                    const ConvProg P, int ntiles, const OpFuse src_fuse_all) {
  conv64_fwd_body<NW, BWD, 0>(src, wpack, bias, dst, stats_partial, P, ntiles, src_fuse_all, SRLZ_NO_POOLSUM);
}

// The data gradient of a convolution whose input was a pooled map, with the pooled block's BatchNorm-backward sums in its epilogue
// (PoolSum; PSUM = 1: one destination class, 2: several).
template <int PSUM>
__global__ __launch_bounds__(256, 2) void conv64_dgrad_poolsum_kernel(const float* __restrict__ src,
                                                                     const float* __restrict__ wpack,
                                                                     float* __restrict__ dst,
                                                                     float* __restrict__ bn_bwd_partial, const ConvProg P,
                                                                     int ntiles, const PoolSum ps) {
  conv64_fwd_body<4, false, PSUM>(src, wpack, nullptr, dst, bn_bwd_partial, P, ntiles, SRLZ_NO_FUSE, ps);
}

// ---------------------------------------------------------------------------------------------------------------
// Software-pipelined, persistent kernels of the stride-2 gather programs (conv64_bwd_fused_kernel: the whole backward of a
// ConvTranspose block; conv64_gather_pipe_kernel: plain operands).  A stride-2 gather tile stages its source FOUR times (one class of
// 128 + span rows per tap group 4 / 2 / 2 / 1), and in conv64_fwd_kernel every one of those stagings is a synchronous HBM round trip
// between two barriers.  Here
//  * the rows of class c+1 are REQUESTED into registers right after the barrier that opens the first tap of class c and LAND
//    in LDS after the barrier that closes its last tap: they travel under 4 / 2 / 2 taps of MFMAs; the requests are branch-free
//    (clamped addresses, masks applied at the landing), so hipcc's wait insertion keeps them in flight;
//  * workgroups are persistent and walk a contiguous run of their XCD's tiles, so class 0 of the NEXT tile travels under the single
//    tap of class 3 and the epilogue, and the weight slab of tap 0 under tap 8;
//  * the tap structure is compile-time (groups {0..3}, {4, 5}, {6, 7}, {8}): no run-time class switch inside the pipeline.
// (Round 3's conv64_dgrad_pipe_kernel — the fused data gradient alone, with the rebuilt gradient stored for a separate weight-gradient
// launch — was the first of this family; conv64_bwd_fused_kernel took over every shape it served and it was removed in round 5.)
// ---------------------------------------------------------------------------------------------------------------
constexpr int GP_THREADS = 512;                 // 8 waves: wave = 32 rows x 32 columns, one accumulator (the rows of a class in
constexpr int GP_RP = GP_THREADS / 16;          // flight cost 376 bytes per thread at 256 threads — with the 4-wave kernel's 64
constexpr int GP_BATCH = 6;                     // accumulator registers on top, that spills; at 512 threads it is 48 + 16)
constexpr int GP_CORE = TM / GP_RP;             // a tile's own rows are its first TM (these programs have min_off == 0)
                                                // rows per pass / rows (of 16 lanes) per thread and class: 128 + span <= 192

struct GatherRows {
  f32x4 v[GP_BATCH], yv[GP_BATCH];
  unsigned offs[GP_CORE];   // float offset of the rows that can lie in the tile's own range (dy_out is indexed like y)
  unsigned ok;              // bit j: row j lies inside the tensor
};

// The source-side row table of a gather tile (cf. rowtab_build): the walk over (image, row, column), the bounds tests and the pixel
// offset of a tile's rows were redone by every thread for each of the tile's four classes (~22 vector-ALU instructions per row and
// class); here the first 192 threads decompose one row each, once per tile:
//   entry = pixel index of the row's class-(0,0) source pixel << 4 | bit k: class k = (cy << 1) | cx is inside the image  (0 = no row)
// Layout: row R at (R & 31) * GT_P + (R >> 5): the six rows of thread t (R = (t >> 4) + 32 j) are consecutive words.
// The tile starts at grid position q0 >= 0 (these programs have min_off == 0).
constexpr int GT_P = 8;
constexpr int GT_WORDS = GP_RP * GT_P;
__device__ __forceinline__ void gtab_build(unsigned* __restrict__ tab, const ConvProg& P, int q0, int nrows) {
  int R = threadIdx.x;
  asm volatile("" : "+v"(R));  // opaque (as in gather_request): the word's address is not worth a register across the tile loop
  if (R < GT_WORDS) {
    unsigned e = 0;
    if (R < nrows) {
      const int qq = q0 + R;
      const int n = fastdiv(qq, P.mPHW, P.sPHW);
      const int rem = qq - n * P.PHW;
      const int a = fastdiv(rem, P.mPW, P.sPW);
      const int y0 = 2 * a, x0 = 2 * (rem - a * P.PW);
      if (n < P.N) {
        unsigned f = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) f |= (y0 + (k >> 1) < P.Hs && x0 + (k & 1) < P.Ws) ? 1u << k : 0u;
        e = ((unsigned)((n * P.Hs + y0) * P.Ws + x0) << 4) | f;
      }
    }
    tab[(R & (GP_RP - 1)) * GT_P + (R >> 5)] = e;
  }
}

// NJ: how many of the thread's six rows (32 j + (t >> 4)) this class needs — a class whose taps reach at most `off` positions ahead
// reads rows [0, TM + off) of its buffer, so the 2-tap class with offsets {0, 1} needs 129 rows (NJ = 5) and the 1-tap class 128
// (NJ = 4): the rows beyond were requested, rebuilt and landed for nobody (3 of a tile's 24 row slots per thread; round 6).
template <int NJ = GP_BATCH>
__device__ __forceinline__ void gather_request(GatherRows& r, const float* __restrict__ src, const float* __restrict__ y,
                                               const unsigned* __restrict__ tab, int cls, int W) {
  int t = threadIdx.x;
  asm volatile("" : "+v"(t));  // opaque: nothing derived from the thread index here is worth a register across the tile loop
  const int slot = t & 15;
  const unsigned delta = (unsigned)((cls >> 1) * W + (cls & 1));
  const unsigned* __restrict__ tp = tab + (t >> 4) * GT_P;
  const uint4 e03 = *(const uint4*)tp;
  uint2 e45 = {0u, 0u};
  if constexpr (NJ > 4) e45 = *(const uint2*)(tp + 4);
  const unsigned e[GP_BATCH] = {e03.x, e03.y, e03.z, e03.w, e45.x, e45.y};
  static_assert(GP_BATCH == 6 && NJ >= GP_CORE && NJ <= GP_BATCH, "the table read above takes six rows");
  unsigned okmask = 0;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    // branch-free: m = all ones where the row's pixel of this class exists; any other row reads pixel 0 and is zeroed when it lands
    const unsigned m = (unsigned)__builtin_amdgcn_sbfe((int)e[j], (unsigned)cls, 1u);
    const unsigned off = ((((e[j] >> 4) + delta) << 6) & m) + slot * 4;
    r.v[j] = *(const f32x4*)(src + off);
    r.yv[j] = *(const f32x4*)(y + off);
    if (j < GP_CORE) r.offs[j] = off;
    okmask |= m & (1u << j);
  }
  r.ok = okmask;
}

// A buffer resource over `bytes` bytes at `p` (raw, unstrided): buffer stores whose offset lies outside are DROPPED by the hardware,
// which makes a conditional store branch-free (offset = GP_DROP for a lane that must not write).  That matters beyond the branch
// itself: vmcnt counts stores too and is in-order, and behind a store inside a branch hipcc can only wait for everything, so
// every later "wait for the weight slab" would also wait for these stores' HBM round trip.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t gp_buffer(float* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(p, 0, (int)bytes, 0x00020000);
}
constexpr unsigned GP_DROP = 0xFFFFFF00u;

// ---------------------------------------------------------------------------------------------------------------
// The stride-2 gather programs with a PLAIN operand — conv3's forward (27x27 -> 14x14, with the BatchNorm statistics of its output)
// and the data gradient of the decoder's first ConvTranspose — software-pipelined like conv64_bwd_fused_kernel's data-gradient half (round 5).
// In conv64_fwd_kernel<4, false> such a tile stages its four source classes in four synchronous HBM round trips between barriers,
// and conv3 has only 900 tiles of them for 512 workgroup slots: 102 us for 47 us of matrix work.  Here, as in the fused kernel above:
// persistent workgroups (2 per CU) walk their XCD's tiles, class c + 1 is requested behind the barrier that opens class c and lands
// behind its last tap, class 0 of the next tile travels under the single tap of class 3 and the epilogue; branch-free requests and
// stores.  Differences: no BatchNorm-backward rebuild, no dy_out; the programs of a convolution with padding start their staged range
// at a NEGATIVE grid offset (min_off < 0: the row table is built with rowtab_build's one-image shift); the epilogue takes the
// per-tile BatchNorm partial sums (sum y, sum y^2 over the tile's valid rows) like conv64_fwd_kernel's, in this kernel's own
// (fixed) summation order.  Same tiles, same accumulation order of the contraction: y is bit-identical to conv64_fwd_kernel's.
// ---------------------------------------------------------------------------------------------------------------
struct PlainRows {
  f32x4 v[GP_BATCH];
  unsigned ok;
};

// gtab_build for a staged range [qstart, qstart + nrows) that may begin before grid position 0 (qstart >= -PHW)
__device__ __forceinline__ void gtab_build_shifted(unsigned* __restrict__ tab, const ConvProg& P, int qstart, int nrows) {
  int R = threadIdx.x;
  asm volatile("" : "+v"(R));
  if (R < GT_WORDS) {
    unsigned e = 0;
    if (R < nrows) {
      const int qq = qstart + R + P.PHW;  // shifted by one image: stays non-negative
      const int n1 = fastdiv(qq, P.mPHW, P.sPHW);
      const int rem = qq - n1 * P.PHW;
      const int a = fastdiv(rem, P.mPW, P.sPW);
      const int y0 = 2 * a, x0 = 2 * (rem - a * P.PW);
      if ((unsigned)(n1 - 1) < (unsigned)P.N) {
        unsigned f = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) f |= (y0 + (k >> 1) < P.Hs && x0 + (k & 1) < P.Ws) ? 1u << k : 0u;
        e = ((unsigned)(((n1 - 1) * P.Hs + y0) * P.Ws + x0) << 4) | f;
      }
    }
    tab[(R & (GP_RP - 1)) * GT_P + (R >> 5)] = e;
  }
}

__device__ __forceinline__ void plain_request(PlainRows& r, const float* __restrict__ src, const unsigned* __restrict__ tab, int cls,
                                              int W) {
  int t = threadIdx.x;
  asm volatile("" : "+v"(t));
  const int slot = t & 15;
  const unsigned delta = (unsigned)((cls >> 1) * W + (cls & 1));
  const unsigned* __restrict__ tp = tab + (t >> 4) * GT_P;
  const uint4 e03 = *(const uint4*)tp;
  const uint2 e45 = *(const uint2*)(tp + 4);
  const unsigned e[GP_BATCH] = {e03.x, e03.y, e03.z, e03.w, e45.x, e45.y};
  unsigned okmask = 0;
#pragma unroll
  for (int j = 0; j < GP_BATCH; ++j) {
    const unsigned m = (unsigned)__builtin_amdgcn_sbfe((int)e[j], (unsigned)cls, 1u);  // all ones where the row's pixel of this class exists
    const unsigned off = ((((e[j] >> 4) + delta) << 6) & m) + slot * 4;                // (any other row reads pixel 0; zeroed at the landing)
    r.v[j] = *(const f32x4*)(src + off);
    okmask |= m & (1u << j);
  }
  r.ok = okmask;
}

__device__ __forceinline__ void plain_land(float* __restrict__ lds, PlainRows& r, int nrows) {
  int t = threadIdx.x;
  asm volatile("" : "+v"(t));
  const int slot = t & 15;
#pragma unroll
  for (int j = 0; j < GP_BATCH; ++j) {
    const int R = (t >> 4) + GP_RP * j;
    const f32x4 v = ((r.ok >> j) & 1u) ? r.v[j] : f32x4{0.f, 0.f, 0.f, 0.f};
    if (R < nrows) *(f32x4*)(lds + R * 64 + ((slot ^ (R & 15)) << 2)) = v;
  }
}

__global__ __launch_bounds__(GP_THREADS, 4) void conv64_gather_pipe_kernel(const float* __restrict__ src_all,
                                                                          const float* __restrict__ wpack,
                                                                          float* __restrict__ dst_all,
                                                                          float* __restrict__ stats_partial, const ConvProg P,
                                                                          int ntiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* As = (float*)smem;                 // (TM + span) x 64, swizzled: the rows of the current class
  float* Bs = As + (TM + P.span) * 64;      // 64 x 64 weight slab of the current tap
  int* rowinfo = (int*)(Bs + 4096);         // [2 (tile parity)][3][TM]: image index (or -1), a*ds, b*ds
  unsigned* gtab = (unsigned*)(rowinfo + 6 * TM);  // the source-side row table of the tile whose rows are being requested
  float* red = (float*)(gtab + GT_WORDS);   // [8 waves][sum 32 | sum of squares 32]: the tile's BatchNorm partials on their way out

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int nrows = TM + P.span;
  const int cls0 = P.tsrc[0], cls1 = P.tsrc[4], cls2 = P.tsrc[6], cls3 = P.tsrc[8];

  const int xcd = blockIdx.x & 7, wi = blockIdx.x >> 3, wpx = gridDim.x >> 3;
  const int tq = ntiles >> 3, tr = ntiles & 7;
  const int tbase = (xcd < tr) ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
  const int tcnt = tq + (xcd < tr ? 1 : 0);

  constexpr int BV = 1024 / GP_THREADS;
  const int bslot = wave * (BV * 64) + lane;
  f32x4 breg[BV];
  {
    const f32x4* wsrc = (const f32x4*)(wpack + (size_t)P.tw[0] * 4096);
#pragma unroll
    for (int i = 0; i < BV; ++i) breg[i] = wsrc[bslot + i * 64];
  }
  const unsigned dst_bytes = (unsigned)P.dst_gstride * 4u;

  PlainRows rr;
  int k = wi;
  int parity = 0;
  if (k < tcnt) {  // the first tile's class 0 is staged the plain way
    const int tile = tbase + k;
    const int grp = (P.G > 1 && tile >= P.tpg) ? 1 : 0;  // (G <= 2, checked by the host)
    const int q0 = (tile - grp * P.tpg) * TM;
    gtab_build_shifted(gtab, P, q0 + P.min_off, nrows);
    __syncthreads();
    plain_request(rr, src_all + grp * P.src_gstride, gtab, cls0, P.Ws);
    plain_land(As, rr, nrows);
  }
  for (; k < tcnt; k += wpx, parity ^= 1) {
    const int tile = tbase + k;
    const int grp = (P.G > 1 && tile >= P.tpg) ? 1 : 0;
    const int q0 = (tile - grp * P.tpg) * TM;
    const float* __restrict__ src = src_all + grp * P.src_gstride;
    const __amdgpu_buffer_rsrc_t dst = gp_buffer(dst_all + grp * P.dst_gstride, dst_bytes);
    const int k2 = k + wpx;  // this workgroup's next tile (past the end: this one again, with no rows)
    const int tile2 = tbase + (k2 < tcnt ? k2 : k);
    const int grp2 = (P.G > 1 && tile2 >= P.tpg) ? 1 : 0;
    const int q02 = (tile2 - grp2 * P.tpg) * TM;
    int* ri = rowinfo + parity * (3 * TM);
    if (tid < TM) {  // (the other parity's copy may still be read by a wave that is flushing the previous tile)
      const int q = q0 + tid;
      int n = -1, ya = 0, xb = 0;
      if (q < P.total_q) {
        n = fastdiv(q, P.mPHW, P.sPHW);
        const int rem = q - n * P.PHW;
        const int a = fastdiv(rem, P.mPW, P.sPW);
        ya = a * P.ds;
        xb = (rem - a * P.PW) * P.ds;
      }
      ri[tid] = n; ri[TM + tid] = ya; ri[2 * TM + tid] = xb;
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    int tid_t = tid;  // (one opaque copy of the thread index per tile: what the nine taps derive from it must not live in registers
                      // across the tile loop, or the rows in flight are pushed into scratch)
    asm volatile("" : "+v"(tid_t));
    const int lane_t = tid_t & 63, wave_t = tid_t >> 6;
    const int wrow_t = wave_t & 3, wcol_t = wave_t >> 2;
    const int h_t = lane_t >> 5, l31_t = lane_t & 31;
    const int arow0 = wrow_t * 32 + l31_t - P.min_off;
    const float* brow = Bs + (wcol_t * 32 + l31_t) * 64;
    const int bkey = lane_t & 15;
    const int bslot_t = wave_t * (BV * 64) + lane_t;

#pragma unroll
    for (int ti = 0; ti < NTAPS; ++ti) {
      __syncthreads();  // all waves are done with the previous tap's Bs — and with As when this tap opens a new class
      {
        f32x4* wdst = (f32x4*)Bs;
#pragma unroll
        for (int i = 0; i < BV; ++i) wdst[bslot_t + i * 64] = breg[i];
      }
      {  // the next tap's slab (tap 0 of the next tile behind tap 8: same weights)
        const f32x4* wsrc = (const f32x4*)(wpack + (size_t)P.tw[(ti + 1) % NTAPS] * 4096);
#pragma unroll
        for (int i = 0; i < BV; ++i) breg[i] = wsrc[bslot_t + i * 64];
      }
      if (ti == 4 || ti == 6 || ti == 8) plain_land(As, rr, nrows);
      if (ti == 7) gtab_build_shifted(gtab, P, q02 + P.min_off, k2 < tcnt ? nrows : 0);
      __syncthreads();
      if (ti == 0) plain_request(rr, src, gtab, cls1, P.Ws);
      if (ti == 4) plain_request(rr, src, gtab, cls2, P.Ws);
      if (ti == 6) plain_request(rr, src, gtab, cls3, P.Ws);
      if (ti == 8) plain_request(rr, src_all + grp2 * P.src_gstride, gtab, cls0, P.Ws);  // class 0 of this workgroup's NEXT tile
      __builtin_amdgcn_sched_barrier(0);  // every request goes out HERE, ahead of the tap's MFMAs
      const int R = arow0 + P.toff[ti];
      int abase = (R * 64 + ((h_t ^ (R & 15)) << 2)) * 4;  // bytes; slot (2kc + h) ^ (R & 15) is this XOR (kc << 5)
      asm volatile("" : "+v"(abase));
#pragma unroll
      for (int kc = 0; kc < 8; ++kc) {
        const f32x4 a = *(const f32x4*)((const char*)As + (abase ^ (kc << 5)));
        const f32x4 b = *(const f32x4*)(brow + (((kc * 2 + h_t) ^ bkey) << 2));
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], b[r], acc, 0, 0, 0);
      }
    }
    __syncthreads();  // every wave is done with the last tap's slab and with As: Bs becomes scratch, As takes the next tile
    if (k + wpx < tcnt) plain_land(As, rr, nrows);  // (the landing first: it waits for its rows only)
    {  // flush through this wave's own 2 KB of the idle slab, 16 tile rows x 32 columns at a time: 16-byte stores, branch-free
      int tid_f = tid;
      asm volatile("" : "+v"(tid_f));
      const int lane_f = tid_f & 63, wave_f = tid_f >> 6;
      const int wrow_f = wave_f & 3, wcol_f = wave_f >> 2, h_f = lane_f >> 5, l31_f = lane_f & 31;
      float* S = Bs + wave_f * 512;
      const int eg = lane_f >> 3, eslot = lane_f & 7;
      f32x4 s4 = {0.f, 0.f, 0.f, 0.f}, q4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int rq = 0; rq < 8; ++rq) {
          const int rowl = (rq & 3) + 8 * (rq >> 2) + 4 * h_f;
          S[rowl * 32 + l31_f] = acc[8 * half + rq];
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const int rowl = eg + 8 * kk;
          const int row = wrow_f * 32 + 16 * half + rowl;
          const f32x4 v = *(const f32x4*)(S + rowl * 32 + eslot * 4);
          const int n = ri[row];
          const int y = ri[TM + row], x = ri[2 * TM + row];
          const bool inside = n >= 0 && y < P.Hd && x < P.Wd;
          __builtin_amdgcn_raw_buffer_store_b128(v, dst, inside ? (unsigned)((n * P.Hd + y) * P.Wd + x) * 256u + wcol_f * 128 + eslot * 16 : GP_DROP,
                                                 0, 0);
          const f32x4 vv = inside ? v : f32x4{0.f, 0.f, 0.f, 0.f};
          s4 += vv;
          q4 += vv * vv;
        }
      }
      if (stats_partial) {
        // this lane's four channels (32 wcol + 4 eslot ..) over its four rows; the eight row groups of the wave (lane bits 3-5), then
        // the four row-waves of a column half through LDS; fixed order
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s4[e] += __shfl_xor(s4[e], 8, 64); s4[e] += __shfl_xor(s4[e], 16, 64); s4[e] += __shfl_xor(s4[e], 32, 64);
          q4[e] += __shfl_xor(q4[e], 8, 64); q4[e] += __shfl_xor(q4[e], 16, 64); q4[e] += __shfl_xor(q4[e], 32, 64);
        }
        if (lane_f < 8) {
          *(f32x4*)(red + wave_f * 64 + lane_f * 4) = s4;
          *(f32x4*)(red + wave_f * 64 + 32 + lane_f * 4) = q4;
        }
        __syncthreads();
        if (tid_f < 128) {
          const int c = tid_f & 63, which = tid_f >> 6;  // [0, 64): sum, [64, 128): sum of squares
          const float* base = red + ((c >> 5) * 4) * 64 + which * 32 + (c & 31);  // waves 4 wcol + wrow
          stats_partial[(size_t)tile * 128 + tid_f] = (base[0] + base[64]) + (base[128] + base[192]);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The WHOLE backward of a decoder block's ConvTranspose2d(64, 64, 3, stride 2) in one launch: data gradient, weight gradient and
// bias gradient from ONE staging of the rebuilt d(loss)/dy.
// A separate fused data-gradient launch reads (dA, y) = 3.2 GB at the 111x111 layer, rebuilds dy and — only so that the weight-gradient
// kernel can read it back — stores it (1.6 GB written, 1.6 GB read again).  Both contractions consume the same operand:
//     da(p)  = sum_t dy_{c_t}(p + off_t) . Wb[t]          (M = positions, N = ci, K = co)
//     dW[t]  = sum_p a(p)^T . dy_{c_t}(p + off_t)         (M = ci, N = co, K = positions),    a = relu(bn(y_prev)),
// so here a tile (128 positions p of the low-resolution grid) stages each class of dy rows once in LDS, runs the data-gradient
// taps on it as the pipelined kernel does, and ALSO multiplies it with the tile's 128 rows of a.  dy never leaves the chip:
// 3.2 GB of the pair's 7.2 GB disappear, and the MFMA work per staged byte doubles.
//  * 512 threads, ONE workgroup per CU (256 registers per lane, 150 KB of LDS): the class rows are double-buffered in LDS, so a
//    class lands two taps after it was requested — where the in-order vmcnt completes its loads anyway — while the previous
//    class is still being read; the a-tile of the next tile lands at the tile boundary.
//  * data gradient: wave = 32 positions x 32 channels (as conv64_gather_pipe_kernel).  Weight gradient: the 36 blocks
//    (9 taps x 2x2 quadrants of 32x32) are spread over the 8 waves per CLASS so that every wave has 32 weight-gradient MFMAs
//    in every tap period: class of 4 taps — wave w owns tap (w >> 1), quadrants (w & 1, {0, 1}), a quarter of the positions per
//    period; classes of 2 taps — tap (w >> 2), quadrant (w & 1, (w >> 1) & 1), half of the positions per period; the 1-tap
//    class — quadrant as before, position half (w >> 2), the two halves added through LDS at the end.  80 accumulator
//    registers per lane instead of 144.
//  * positions are visited as p = 16 i + jj + 8 h (h = the MFMA's two k-lanes): for fixed (jj, h) the rows 16 i apart share the
//    XOR key of the swizzled dy rows, so a lane reaches its eight k-steps from ONE address with immediate offsets
//    (ds_read2st64_b32), for the dy operand as for the (unswizzled) a-tile.
// Results: dx bit-identical to conv64_fwd_kernel<4, true>'s; dW / db differ from the two-kernel path by summation order only
// (per-workgroup partials, fixed-order fp64 second stage: deterministic).
// ---------------------------------------------------------------------------------------------------------------
// which channels of a BatchNorm record cannot give xhat back from the activation (shared by the fused kernel and its companion)
__host__ __device__ __forceinline__ bool bnpart_zero_scale(float scale, float shift) {
  return fabsf(scale) <= 1e-3f * fabsf(shift) || scale == 0.f;
}

// Folding one lane bit of TWO per-lane values with one add (gfx950): fold32(a, b) = { a[l] + a[l + 32] in lanes l < 32, b[l - 32] + b[l]
// in lanes l >= 32 };  fold16(a, b) = { a's rows (16 lanes) 0 + 1 in row 0, b's rows 0 + 1 in row 1, a's 2 + 3 in row 2, b's 2 + 3 in row 3 }.
__device__ __forceinline__ float fold32(float a, float b) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float fold16(float a, float b) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

struct FusedBwd {
  const float* x;        // raw y_prev [N, Hd, Wd, 64] (the ConvTranspose's input before BatchNorm + ReLU)
  const float* x_bnp;    // its BatchNorm record(s): a = relu(bn(x))
  float* wpartial;       // [workgroups][9 * 4096 + 64]
  // Round 6: the data gradient this kernel writes is dA of the PREVIOUS layer's BatchNorm + ReLU (x is that layer's raw output), and the
  // tile that writes it holds relu(bn(x)) of the same 128 positions in LDS (the weight gradient's operand) — so the flush also leaves
  // the two BatchNorm-backward sums of that layer,  sum dz  and  sum dz * xhat  (dz = dA where relu(bn(x)) > 0; there the activation
  // IS gamma * xhat + beta, so xhat follows from it), as partial records for srlz_bn_bwd_finalize_partials: the separate pass of
  // srlz_bn_relu_bwd_sums over (x, dA) — 0.8 GB at the 55 x 55 layer — disappears.  bnpart: [groups][bn_rows][128] floats, rows
  // 4 * tile + (wave & 3) of a group from this kernel (channels whose BatchNorm scale is (almost) 0 contribute 0 here: xhat cannot be
  // recovered from the activation — the rows behind them come from conv64_bnpart_zero_scale_kernel); NULL = off.
  float* bnpart;
  int bn_rows;           // records per BatchNorm group
};

// rows per thread the second / fourth class of a tile need (gather_request<NJ>): their taps reach at most 1 / 0 positions ahead
// (fused_bwd_ok checks it: build_program puts the 2-tap class {0, +1} of a stride-2 ConvTranspose's data gradient second, the 1-tap class {0} last)
constexpr int FB_NJ1 = 5, FB_NJ3 = 4;
constexpr int FB_REACH1 = FB_NJ1 * GP_RP - TM, FB_REACH3 = FB_NJ3 * GP_RP - TM;  // 32 and 0 positions

struct YRows { f32x4 v[4]; unsigned ok; };  // a thread's share of the tile's 128 rows of y_prev (ytile_row)

// The a-tile is WAVE-PRIVATE between its landing and the flush that reads it back (round 6): wave (wrow = wave & 3, wcol = wave >> 2)
// requests, lands and — in the flush, for the BatchNorm-backward sums of the layer that produced y_prev — re-reads rows
// 32 wrow + 8 j + (lane >> 3), j = 0..3, channels 32 wcol + 4 (lane & 7) ..: exactly the block of the data gradient it flushes.  So a
// wave may read its block back BEHIND the tile's closing barrier and land the next tile's block over it without another barrier
// (every other reader of the a-tile — the weight-gradient steps of all waves — sits between the tap barriers).
// ... and the table of the tile's own 128 positions in the low-resolution tensor (y_prev): entry = pixel index << 1 | 1, 0 = outside;
// row R at ((R >> 5) * 8 + (R & 7)) * 4 + ((R >> 3) & 3): one 16-byte read per thread.  Built by threads [256, 384).
constexpr int YT_WORDS = GP_RP * 4;
__device__ __forceinline__ void ytab_build(unsigned* __restrict__ tab, const ConvProg& P, int q0, bool live) {
  int R = (int)threadIdx.x - 256;
  asm volatile("" : "+v"(R));
  if ((unsigned)R < (unsigned)YT_WORDS) {
    const int qq = q0 + R;
    const int n = fastdiv(qq, P.mPHW, P.sPHW);
    const int rem = qq - n * P.PHW;
    const int a = fastdiv(rem, P.mPW, P.sPW);
    const int b = rem - a * P.PW;
    const bool ok = live && n < P.N && a < P.Hd && b < P.Wd;
    tab[((R >> 5) * 8 + (R & 7)) * 4 + ((R >> 3) & 3)] = ok ? ((unsigned)((n * P.Hd + a) * P.Wd + b) << 1) | 1u : 0u;
  }
}

__device__ __forceinline__ void ytile_request(YRows& r, const float* __restrict__ x, const unsigned* __restrict__ tab) {
  int t = threadIdx.x;
  asm volatile("" : "+v"(t));
  const int lane = t & 63, wave = t >> 6;
  const int col = (wave >> 2) * 32 + (lane & 7) * 4;
  const uint4 q = *(const uint4*)(tab + ((wave & 3) * 8 + (lane >> 3)) * 4);
  const unsigned e[4] = {q.x, q.y, q.z, q.w};
  unsigned okmask = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    r.v[j] = *(const f32x4*)(x + ((e[j] >> 1) << 6) + col);  // (a row outside reads pixel 0; zeroed when it lands)
    okmask |= (e[j] & 1u) << j;
  }
  r.ok = okmask;
}

// xrec: [2][64] in LDS — scale, shift of the previous layer's BatchNorm for the tile's group
__device__ __forceinline__ void ytile_land(float* __restrict__ Ys, YRows& r, const float* __restrict__ xrec) {
  int t = threadIdx.x;
  asm volatile("" : "+v"(t));
  const int lane = t & 63, wave = t >> 6;
  const int col = (wave >> 2) * 32 + (lane & 7) * 4;
  const int row0 = (wave & 3) * 32 + (lane >> 3);
  const f32x4 sc4 = *(const f32x4*)(xrec + col), sh4 = *(const f32x4*)(xrec + 64 + col);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const bool ok = (r.ok >> j) & 1u;
    f32x4 v = r.v[j];
    {
      const f32x4 z4 = __builtin_elementwise_fma(v, sc4, sh4);  // (v_pk_fma_f32; each element the same IEEE fma as before)
      const float hi = ok ? __builtin_inff() : 0.f;              // relu, and 0 for a row outside the tensor: one v_med3 per element
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_fmed3f(z4[e], 0.f, hi);
    }
    *(f32x4*)(Ys + (row0 + 8 * j) * 64 + col) = v;
  }
}

// the landing of a class's rows: BatchNorm + ReLU backward rebuilt from (dA, y), zero outside the tensor; bs4 += the tile's own rows (every dy element belongs to exactly one tile's range)
template <int NJ = GP_BATCH>
__device__ __forceinline__ void gather_land_sum(float* __restrict__ lds, GatherRows& r, int nrows, const float* __restrict__ lrec,
                                                f32x4& bs4) {
  int t = threadIdx.x;
  asm volatile("" : "+v"(t));
  const int slot = t & 15;
  const f32x4 sc4 = *(const f32x4*)(lrec + slot * 4), sh4 = *(const f32x4*)(lrec + 64 + slot * 4);
  const f32x4 c0 = *(const f32x4*)(lrec + 128 + slot * 4), c1 = *(const f32x4*)(lrec + 192 + slot * 4);
#pragma unroll
  for (int j = 0; j < NJ; ++j) {  // (rows 32 NJ .. of the buffer keep an earlier class's values: this class's taps never read them)
    const int R = (t >> 4) + GP_RP * j;
    const bool ok = (r.ok >> j) & 1u;
    f32x4 v = r.v[j];
    const f32x4 yy = r.yv[j];
#pragma unroll
    for (int e = 0; e < 4; ++e) {  // (scalar on purpose: the packed form needs aligned register pairs, and conv64_bwd_fused_kernel — at
      // its 256-register limit — spills 14 registers with it instead of 4)
      const float z = __builtin_fmaf(yy[e], sc4[e], sh4[e]);
      const float dz = z > 0.f ? v[e] : 0.f;
      v[e] = ok ? __builtin_fmaf(sc4[e], dz, -__builtin_fmaf(c1[e], yy[e], c0[e])) : 0.f;
    }
    if (j < GP_CORE) bs4 += v;  // (min_off == 0: rows [0, TM) are the tile's own; rows outside the tensor are zero)
    if (R < nrows) *(f32x4*)(lds + R * 64 + ((slot ^ (R & 15)) << 2)) = v;
  }
}

// NJJ k-groups (jj0 .. jj0 + NJJ - 1) of one tap's weight gradient for NB (1 or 2) column quadrants sharing the row quadrant mi:
// acc[b] += a-tile[p][32 mi ..]^T . dy[p + off][32 (nj0 + b) ..]  over p = 16 i + jj + 8 h.
template <int NB, int NJJ>
__device__ __forceinline__ void wgrad_steps(f32x16 (&acc)[NB], const float* __restrict__ Ys, const float* __restrict__ Ac, int off,
                                            int mi, int nj0, int jj0, int h, int l31) {
#pragma unroll
  for (int q = 0; q < NJJ; ++q) {
    const int jj = jj0 + q;
    const float* ap = Ys + (jj + 8 * h) * 64 + mi * 32 + l31;
    const int Rj = jj + 8 * h + off, key = Rj & 15;
    const float* bp[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int co = (nj0 + b) * 32 + l31;
      bp[b] = Ac + Rj * 64 + ((((co >> 2) ^ key) << 2) | (co & 3));
    }
#pragma unroll
    for (int i0 = 0; i0 < 8; i0 += 4) {  // (four k-steps at a time: the fragments of eight would cost 12 more registers)
      float av[4], bv[NB][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        av[i] = ap[(i0 + i) * 1024];
#pragma unroll
        for (int b = 0; b < NB; ++b) bv[b][i] = bp[b][(i0 + i) * 1024];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[b][i], acc[b], 0, 0, 0);
    }
  }
}

__global__ __launch_bounds__(GP_THREADS, 2) void conv64_bwd_fused_kernel(const float* __restrict__ src_all,
                                                                        const float* __restrict__ wpack,
                                                                        float* __restrict__ dst_all, const ConvProg P, int ntiles,
                                                                        const OpFuse fuse_all, const FusedBwd fb) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int nrows = TM + P.span;
  constexpr int nrows1 = TM + FB_REACH1;    // what the short-reach classes (second, fourth) can touch of their buffer
  float* As0 = (float*)smem;                // class rows, double-buffered: classes 0 / 2 here (TM + span rows),
  float* As1 = As0 + nrows * 64;            // classes 1 / 3 here (TM + 32 rows)
  float* Ys = As1 + nrows1 * 64;            // [TM][64]: relu(bn(y_prev)) of the tile's positions
  float* Bs0 = Ys + TM * 64;                // 2 x (64 x 64): the weight slabs of the current tap and of the next one (round 6: the slab of
                                            // tap t + 1 is written WHILE tap t runs, so a tap needs one barrier, not two)
  int* rowinfo = (int*)(Bs0 + 2 * 4096);    // [2 (tile parity)][3][TM]
  float* frec = (float*)(rowinfo + 6 * TM); // [G <= 2][4][64]: scale, shift, c0, c1 of this layer's BatchNorm backward
  float* xrec = frec + 512;                 // [G <= 2][2][64]: scale, shift of the previous layer's BatchNorm
  unsigned* gtab = (unsigned*)(xrec + 256); // row tables of the tile whose rows are being requested: source side (gtab_build)
  unsigned* ytab = gtab + GT_WORDS;         // ... and its own 128 positions in y_prev (ytab_build)
  float* prec = (float*)(ytab + YT_WORDS);  // [G <= 2][3][64]: pA, pB, threshold — xhat = a * pA + pB where a = relu(bn(x)) > threshold

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wrow = wave & 3, wcol = wave >> 2;
  const int cls0 = P.tsrc[0], cls1 = P.tsrc[4], cls2 = P.tsrc[6], cls3 = P.tsrc[8];
  // weight-gradient assignment of this wave (wave-uniform)
  const int wmi = wave & 1, wnj = (wave >> 1) & 1, wk = wave >> 2;
  const int tap_c0 = wave >> 1, tap_c1 = 4 + wk, tap_c2 = 6 + wk;
  const int off_c0 = P.toff[tap_c0], off_c1 = P.toff[tap_c1], off_c2 = P.toff[tap_c2], off_c3 = P.toff[8];

  const int xcd = blockIdx.x & 7, wi = blockIdx.x >> 3, wpx = gridDim.x >> 3;
  const int tq = ntiles >> 3, tr = ntiles & 7;
  const int tbase = (xcd < tr) ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
  const int tcnt = tq + (xcd < tr ? 1 : 0);

  if (tid < 64 * P.G) {
    const int g = tid >> 6, c = tid & 63;
    const float* bnp = fuse_all.bnp + g * 256;
    const float* sums = fuse_all.sums + g * 128;
    const float sc = bnp[128 + c], sh = bnp[192 + c];
    float c0 = 0.f, c1 = 0.f;
    if (fuse_all.training) {
      c1 = sc * bnp[64 + c] * sums[64 + c] * fuse_all.inv_count;
      c0 = sc * sums[c] * fuse_all.inv_count - c1 * bnp[c];
    }
    float* fr = frec + g * 256;
    fr[c] = sc; fr[64 + c] = sh; fr[128 + c] = c0; fr[192 + c] = c1;
    xrec[g * 128 + c] = fb.x_bnp[g * 256 + 128 + c];
    xrec[g * 128 + 64 + c] = fb.x_bnp[g * 256 + 192 + c];
    {  // a = scale * x + shift = gamma * xhat + beta  =>  xhat = a * (invstd / scale) - (shift / scale + mean) * invstd; a channel whose
       // |scale| is tiny against |shift| (exactly 0 included) would lose xhat in the cancellation: it contributes nothing here
       // (threshold +inf) and is summed by conv64_bnpart_zero_scale_kernel from x itself (cf. the pooled-block epilogue, PSUM)
      const float xmean = fb.x_bnp[g * 256 + c], xinv = fb.x_bnp[g * 256 + 64 + c];
      const float xsc = fb.x_bnp[g * 256 + 128 + c], xsh = fb.x_bnp[g * 256 + 192 + c];
      const bool zero = bnpart_zero_scale(xsc, xsh);
      const float isc = zero ? 0.f : 1.f / xsc;
      prec[g * 192 + c] = xinv * isc;
      prec[g * 192 + 64 + c] = -(xsh * isc + xmean) * xinv;
      prec[g * 192 + 128 + c] = zero ? __builtin_inff() : 0.f;
    }
  }

  constexpr int BV = 1024 / GP_THREADS;
  f32x4 breg[BV];
  {
    const f32x4* wsrc = (const f32x4*)(wpack + (size_t)P.tw[0] * 4096);
#pragma unroll
    for (int i = 0; i < BV; ++i) breg[i] = wsrc[wave * (BV * 64) + lane + i * 64];
  }
  const unsigned dst_bytes = (unsigned)P.dst_gstride * 4u;
  const __amdgpu_buffer_rsrc_t bnbuf = gp_buffer(fb.bnpart, fb.bnpart ? (unsigned)(P.G * fb.bn_rows) * 512u : 0u);

  f32x16 aw0[2], aw1[1], aw2[1], aw3[1];  // weight-gradient accumulators of the four classes
#pragma unroll
  for (int r = 0; r < 16; ++r) { aw0[0][r] = 0.f; aw0[1][r] = 0.f; aw1[0][r] = 0.f; aw2[0][r] = 0.f; aw3[0][r] = 0.f; }
  f32x4 bs4 = {0.f, 0.f, 0.f, 0.f};  // bias gradient: column sums of the tile's own dy rows (channels 4 slot .. of this thread's rows)

  GatherRows rr;
  YRows yr;
  int k = wi;
  int parity = 0;
  if (k < tcnt) {  // the first tile's class 0 and a-tile are staged the plain way
    const int tile = tbase + k;
    const int grp = (P.G > 1 && tile >= P.tpg) ? 1 : 0;
    const int q0 = (tile - grp * P.tpg) * TM;
    gtab_build(gtab, P, q0, nrows);
    ytab_build(ytab, P, q0, true);
    __syncthreads();  // the tables, frec and xrec are complete
    gather_request(rr, src_all + grp * P.src_gstride, fuse_all.y + grp * P.src_gstride, gtab, cls0, P.Ws);
    ytile_request(yr, fb.x + grp * P.dst_gstride, ytab);
    gather_land_sum(As0, rr, nrows, frec + grp * 256, bs4);
    ytile_land(Ys, yr, xrec + grp * 128);
  }
  // The slab of the first tile's tap 0 goes to slab buffer 0 now (published by that tap's barrier) and tap 1's is requested: from here
  // on tap t writes the slab of tap t + 1 into the buffer tap t - 1 read, so a tap needs ONE barrier — the one that says "everybody is
  // done with tap t - 1" — instead of two (until round 6: slab write and landings sat between two barriers, with every matrix pipe idle)
  // Tap t reads slab buffer t & 1 (compile-time); a tile has nine taps, so its last tap and the next tile's first both read buffer 0:
  // tap 8 writes no slab, the next tile's first slab is written behind the tile's closing barrier instead.
  {
    f32x4* wdst = (f32x4*)Bs0;
#pragma unroll
    for (int i = 0; i < BV; ++i) wdst[wave * (BV * 64) + lane + i * 64] = breg[i];
    const f32x4* wsrc = (const f32x4*)(wpack + (size_t)P.tw[1] * 4096);
#pragma unroll
    for (int i = 0; i < BV; ++i) breg[i] = wsrc[wave * (BV * 64) + lane + i * 64];
  }
  for (; k < tcnt; k += wpx, parity ^= 1) {
    const int tile = tbase + k;
    const int grp = (P.G > 1 && tile >= P.tpg) ? 1 : 0;
    const int q0 = (tile - grp * P.tpg) * TM;
    const float* __restrict__ src = src_all + grp * P.src_gstride;
    const float* __restrict__ ysrc = fuse_all.y + grp * P.src_gstride;
    const __amdgpu_buffer_rsrc_t dst = gp_buffer(dst_all + grp * P.dst_gstride, dst_bytes);
    const float* lrec = frec + grp * 256;
    const int k2 = k + wpx;
    const bool more = k2 < tcnt;
    const int tile2 = tbase + (more ? k2 : k);
    const int grp2 = (P.G > 1 && tile2 >= P.tpg) ? 1 : 0;
    const int q02 = (tile2 - grp2 * P.tpg) * TM;
    int* ri = rowinfo + parity * (3 * TM);
    if (tid < TM) {
      const int q = q0 + tid;
      int n = -1, ya = 0, xb = 0;
      if (q < P.total_q) {
        n = fastdiv(q, P.mPHW, P.sPHW);
        const int rem = q - n * P.PHW;
        const int a = fastdiv(rem, P.mPW, P.sPW);
        ya = a * P.ds;
        xb = (rem - a * P.PW) * P.ds;
      }
      ri[tid] = n; ri[TM + tid] = ya; ri[2 * TM + tid] = xb;
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // (one opaque copy of the lane index per tile: see conv64_gather_pipe_kernel)
    int lane_t = lane;
    asm volatile("" : "+v"(lane_t));
    const int h_t = lane_t >> 5, l31_t = lane_t & 31;
    const int arow0 = wrow * 32 + l31_t;
    const int brow_off = (wcol * 32 + l31_t) * 64;
    const int bkey = lane_t & 15;
    const int bslot_t = wave * (BV * 64) + lane_t;

#pragma unroll
    for (int ti = 0; ti < NTAPS; ++ti) {
      // class of this tap and the LDS buffer that holds it (compile-time after unrolling)
      const float* Ac = (ti < 4 || ti == 6 || ti == 7) ? As0 : As1;
      const float* Bcur = Bs0 + (ti & 1) * 4096;
      float* Bnext = Bs0 + ((ti + 1) & 1) * 4096;
      // ONE barrier per tap: every wave has finished tap ti - 1, so the slab buffer that tap read (Bnext) and the class buffer that is
      // about to be refilled are free, and this tap's slab (written during tap ti - 1) and the rows landed meanwhile are visible.
      // Everything up to the MFMAs below runs per wave, un-synchronised: a wave that is done landing starts its matrix work while its
      // neighbours are still landing.
      __syncthreads();
      if (ti < NTAPS - 1) {
        f32x4* wdst = (f32x4*)Bnext;  // the slab of tap ti + 1
#pragma unroll
        for (int i = 0; i < BV; ++i) wdst[bslot_t + i * 64] = breg[i];
        // ... and the request for the one after it (behind tap 7: tap 0 of the next tile — same weights — which is written behind the
        // tile's closing barrier)
        const f32x4* wsrc = (const f32x4*)(wpack + (size_t)P.tw[(ti + 2) % NTAPS] * 4096);
#pragma unroll
        for (int i = 0; i < BV; ++i) breg[i] = wsrc[bslot_t + i * 64];
      }
      // rows requested two taps ago land now (the in-order vmcnt has completed them with the slab just written): class 1 -> As1
      // while taps 2, 3 still read class 0 in As0; class 2 -> As0 once class 0 is done; class 3 -> As1; the next tile's class 0 -> As0.
      // (Two barriers — those of taps ti + 1 and ti + 2 — lie between a landing and the first tap that reads it.)
      if (ti == 2) gather_land_sum<FB_NJ1>(As1, rr, nrows1, lrec, bs4);
      if (ti == 4) gather_land_sum(As0, rr, nrows, lrec, bs4);
      if (ti == 6) gather_land_sum<FB_NJ3>(As1, rr, nrows1, lrec, bs4);
      if (ti == 8) gather_land_sum(As0, rr, nrows, frec + grp2 * 256, bs4);  // (past the last tile: every row masked off -> zeros; no
                                                                             // run-time branch around a landing, or its join costs a full vmcnt(0))
      // the NEXT tile's row tables, between the last request of this tile (tap 4) and the first of the next (tap 6) — the barriers of
      // taps 5 and 6 fence both sides; past the end: no rows -> every entry 0 -> every row reads pixel 0 and is dropped
      if (ti == 5) { gtab_build(gtab, P, q02, more ? nrows : 0); ytab_build(ytab, P, q02, more); }
      if (ti == 0) gather_request<FB_NJ1>(rr, src, ysrc, gtab, cls1, P.Ws);
      if (ti == 2) gather_request(rr, src, ysrc, gtab, cls2, P.Ws);
      if (ti == 4) gather_request<FB_NJ3>(rr, src, ysrc, gtab, cls3, P.Ws);
      if (ti == 6) gather_request(rr, src_all + grp2 * P.src_gstride, fuse_all.y + grp2 * P.src_gstride, gtab, cls0, P.Ws);
      if (ti == 8) ytile_request(yr, fb.x + grp2 * P.dst_gstride, ytab);
      __builtin_amdgcn_sched_barrier(0);
      {  // ---- data gradient: 32 positions x 32 channels of this wave
        const int R = arow0 + P.toff[ti];
        int abase = (R * 64 + ((h_t ^ (R & 15)) << 2)) * 4;
        asm volatile("" : "+v"(abase));
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) {
          const f32x4 a = *(const f32x4*)((const char*)Ac + (abase ^ (kc << 5)));
          const f32x4 b = *(const f32x4*)(Bcur + brow_off + (((kc * 2 + h_t) ^ bkey) << 2));
#pragma unroll
          for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], b[r], acc, 0, 0, 0);
        }
      }
      // ---- weight gradient: this wave's share of the class that is resident during this tap period
      // (the lane index opaque once more, per tap: the operand addresses of all nine taps are tile-invariant functions of it, and
      // computed early and kept they cost more registers than there are)
      int lane_p = lane_t;
      asm volatile("" : "+v"(lane_p));
      const int h_p = lane_p >> 5, l31_p = lane_p & 31;
      if (ti < 4) wgrad_steps<2, 2>(aw0, Ys, As0, off_c0, wmi, 0, 2 * ti, h_p, l31_p);
      else if (ti < 6) wgrad_steps<1, 4>(aw1, Ys, As1, off_c1, wmi, wnj, 4 * (ti - 4), h_p, l31_p);
      else if (ti < 8) wgrad_steps<1, 4>(aw2, Ys, As0, off_c2, wmi, wnj, 4 * (ti - 6), h_p, l31_p);
      else wgrad_steps<1, 4>(aw3, Ys, As1, off_c3, wmi, wnj, 4 * wk, h_p, l31_p);
    }
    __syncthreads();  // every wave is done with the last tap's slab (buffer 0), with class 3 and with the a-tile
    {  // the next tile's first slab -> buffer 0, its second requested (cf. the prologue)
      f32x4* wdst = (f32x4*)Bs0;
#pragma unroll
      for (int i = 0; i < BV; ++i) wdst[bslot_t + i * 64] = breg[i];
      const f32x4* wsrc = (const f32x4*)(wpack + (size_t)P.tw[1] * 4096);
#pragma unroll
      for (int i = 0; i < BV; ++i) breg[i] = wsrc[bslot_t + i * 64];
    }
    {  // flush of the data gradient (see conv64_gather_pipe_kernel) through this wave's own 2 KB of slab buffer 1 (tap 7 was its last
       // reader).  Round 6: the lane that stores four channels of a position also reads relu(bn(x)) of the same four out of the a-tile —
       // this wave's own block of it (ytile_request), so no barrier is needed before the next tile's block lands over it below — for the
       // BatchNorm-backward sums of the layer that produced x (FusedBwd::bnpart): s = sum of dA where a > 0, q = sum of dA * a (a is 0
       // where the ReLU is closed and outside the tensor, so q needs no mask); sum dz * xhat = pA q + pB s per lane.
      float* S = Bs0 + 4096 + wave * 512;
      const int eg = lane_t >> 3, eslot = lane_t & 7;
      const int cbase = wcol * 32 + eslot * 4;  // this lane's four channels
      const f32x4 pT = *(const f32x4*)(prec + grp * 192 + 128 + cbase);
      f32x4 s4 = {0.f, 0.f, 0.f, 0.f}, q4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int rq = 0; rq < 8; ++rq) {
          const int rowl = (rq & 3) + 8 * (rq >> 2) + 4 * h_t;
          S[rowl * 32 + l31_t] = acc[8 * half + rq];
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const int rowl = eg + 8 * kk;
          const int row = wrow * 32 + 16 * half + rowl;
          const f32x4 v = *(const f32x4*)(S + rowl * 32 + eslot * 4);
          const f32x4 av = *(const f32x4*)(Ys + row * 64 + cbase);  // relu(bn(x)) of the position; 0 outside the tensor
          const int n = ri[row];
          const int y = ri[TM + row], x = ri[2 * TM + row];
          const bool inside = n >= 0 && y < P.Hd && x < P.Wd;
          __builtin_amdgcn_raw_buffer_store_b128(v, dst, inside ? (unsigned)((n * P.Hd + y) * P.Wd + x) * 256u + wcol * 128 + eslot * 16 : GP_DROP,
                                                 0, 0);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            s4[e] += av[e] > pT[e] ? v[e] : 0.f;
            q4[e] = __builtin_fmaf(v[e], av[e], q4[e]);
          }
        }
      }
      {
        const f32x4 pA = *(const f32x4*)(prec + grp * 192 + cbase), pB = *(const f32x4*)(prec + grp * 192 + 64 + cbase);
#pragma unroll
        for (int e = 0; e < 4; ++e) q4[e] = __builtin_fmaf(pA[e], q4[e], pB[e] * s4[e]);
      }
      // this wave's 32 rows = the eight row groups (lane bits 3-5) of the eight sums, as a transposing butterfly (gfx950's
      // v_permlane32_swap / v_permlane16_swap: one add folds a lane bit of TWO values): 14 vector instructions instead of 24 ds_bpermute + 24
      // adds, fixed order.  Afterwards lane L holds channel 4 eslot + {0, 2, 1, 3}[L >> 4] of s (in s4[0]) and of q (in q4[0]); the lanes
      // with bit 3 clear leave the wave's half of record 4 * tile + wrow (branch-free: the other lanes — and every lane when bnpart is
      // NULL, a zero-sized buffer — store out of range)
      {
        const float u0 = fold32(s4[0], s4[1]), u1 = fold32(s4[2], s4[3]), u2 = fold32(q4[0], q4[1]), u3 = fold32(q4[2], q4[3]);
        float w0 = fold16(u0, u1), w1 = fold16(u2, u3);
        w0 += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(w0), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
        w1 += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(w1), 0x128, 0xf, 0xf, false));
        const int erow = lane_t >> 4;
        const unsigned rec = (unsigned)(grp * fb.bn_rows + 4 * (tile - grp * P.tpg) + wrow) * 512u +
                             (unsigned)(cbase + ((erow & 1) << 1) + (erow >> 1)) * 4u;
        const bool mine = (lane_t & 8) == 0;
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(w0), bnbuf, mine ? rec : GP_DROP, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(w1), bnbuf, mine ? rec + 256u : GP_DROP, 0, 0);
      }
    }
    ytile_land(Ys, yr, xrec + grp2 * 128);  // (past the last tile: zeros)
  }

  // ---- the workgroup's weight-gradient partial [9 (reference tap index)][64 ci][64 co] and bias partial [64]
  __syncthreads();  // (everything in LDS is dead from here on)
  {
    const int h = lane >> 5, l31 = lane & 31;
    float* out = fb.wpartial + (size_t)blockIdx.x * (NTAPS * 4096 + 64);
    auto put = [&](const f32x16& a, int tap, int mi, int nj) {
      float* o = out + (size_t)P.tw[tap] * 4096;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[(mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 64 + nj * 32 + l31] = a[r];
    };
    put(aw0[0], tap_c0, wmi, 0);
    put(aw0[1], tap_c0, wmi, 1);
    put(aw1[0], tap_c1, wmi, wnj);
    put(aw2[0], tap_c2, wmi, wnj);
    // tap 8: the two position halves (waves w and w + 4) are added through LDS
    float* X = As0;  // [4 quadrants][16 regs][64 lanes]
    if (wk == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) X[((wave & 3) * 16 + r) * 64 + lane] = aw3[0][r];
    }
    __syncthreads();
    if (wk == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) aw3[0][r] += X[((wave & 3) * 16 + r) * 64 + lane];
      put(aw3[0], 8, wmi, wnj);
    }
    __syncthreads();
    float* red = As0;  // [32 row groups][64 channels]
    *(f32x4*)(red + (tid >> 4) * 64 + (tid & 15) * 4) = bs4;
    __syncthreads();
    if (tid < 64) {
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < GP_RP; ++r) t += red[r * 64 + tid];
      out[NTAPS * 4096 + tid] = t;
    }
  }
}

// Companion of conv64_bwd_fused_kernel's BatchNorm-backward records (FusedBwd::bnpart): the channels whose BatchNorm scale is (almost)
// 0 — xhat cannot be recovered from relu(bn(x)) there — summed from x itself, as srlz_bn_relu_bwd_sums does for every channel, into
// the BNZ_BLOCKS records behind the fused kernel's.  A group without such a channel (the normal case) costs one ~4 us launch that
// writes zero records; with one, this is a pass over (x, dA): rare, and slow on purpose.
constexpr int BNZ_BLOCKS = 64;
__global__ __launch_bounds__(256) void conv64_bnpart_zero_scale_kernel(const float* __restrict__ x, const float* __restrict__ x_bnp,
                                                                      const float* __restrict__ da, float* __restrict__ bnpart,
                                                                      long long pixels, int bn_rows, int first_row) {
  const int g = blockIdx.y;  // BatchNorm group; pixels = positions of ONE group
  x_bnp += g * 256;
  x += (size_t)g * pixels * 64;
  da += (size_t)g * pixels * 64;
  const int c4 = threadIdx.x & 15;
  const f32x4 mean = *(const f32x4*)(x_bnp + c4 * 4), invstd = *(const f32x4*)(x_bnp + 64 + c4 * 4);
  const f32x4 sc = *(const f32x4*)(x_bnp + 128 + c4 * 4), sh = *(const f32x4*)(x_bnp + 192 + c4 * 4);
  unsigned zmask = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) zmask |= (bnpart_zero_scale(sc[j], sh[j]) ? 1u : 0u) << j;
  double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
  if (__syncthreads_or(zmask != 0)) {
    for (long long pix = (long long)blockIdx.x * 16 + (threadIdx.x >> 4); pix < pixels; pix += (long long)gridDim.x * 16) {
      const f32x4 v = *(const f32x4*)(x + pix * 64 + c4 * 4);
      const f32x4 d = *(const f32x4*)(da + pix * 64 + c4 * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (((zmask >> j) & 1u) && v[j] * sc[j] + sh[j] > 0.f) {
          s1[j] += (double)d[j];
          s2[j] += (double)(d[j] * ((v[j] - mean[j]) * invstd[j]));
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
    bnpart[((size_t)g * bn_rows + first_row + blockIdx.x) * 128 + threadIdx.x] = (float)t;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Weight-gradient kernel: dW[w][ci][co] = sum_q S_c(q+off)[ci] * G_d(q)[co]   (G = dy at dest class d).
// GEMM view: M = ci (64), N = co (64), K = grid positions.  4 waves = 4 quadrants of 32x32, each holding all 9 taps
// (144 accumulator registers); persistent over K-chunks of 64 positions; per-workgroup partials are reduced by
// conv64_wgrad_reduce in a fixed order (deterministic).
// ---------------------------------------------------------------------------------------------------------------
template <bool S2, int TK>
__global__ __launch_bounds__(256, 2) void conv64_wgrad_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ g,
                                                             float* __restrict__ partial, const ConvProg P,
                                                             int nchunks, const OpFuse x_fuse, const OpFuse g_fuse) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* Ss = (float*)smem;              // (TK + span) x 64
  float* Gs = Ss + (TK + P.span) * 64;   // TK x 64

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int mi = wave & 1, nj = wave >> 1;

  f32x16 acc[NTAPS];
#pragma unroll
  for (int t = 0; t < NTAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;  // column sum of dy (bias gradient): thread (col = tid&63, part = tid>>6)

  constexpr int NG = S2 ? 4 : 1;
  constexpr int GSTART[5] = {0, S2 ? 4 : 9, 6, 8, 9};

  // nchunks = P.G * cpg: chunk -> (BatchNorm group, chunk of that group's grid); a chunk never straddles two groups
  const int cpg = nchunks / P.G;
  for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    const int grp = (P.G > 1) ? chunk / cpg : 0;
    const int q0 = (chunk - grp * cpg) * TK;
    const float* __restrict__ xg = x + grp * P.src_gstride;
    const float* __restrict__ gg = g + grp * P.dst_gstride;
    const OpFuse xf = fuse_for_group(x_fuse, grp, grp * P.src_gstride);
    const OpFuse gf = fuse_for_group(g_fuse, grp, grp * P.dst_gstride);
    int cur_s = -1, cur_g = -1;
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      const int t0 = GSTART[gi], t1 = GSTART[gi + 1];
      const int cs = P.tsrc[t0], cd = P.tdst[t0];
      __syncthreads();
      if (cs != cur_s) {
        stage_rows<false, 4>(Ss, xg, P.Hs, P.Ws, P.ss, cs, P.PW, P.PH, P.total_q, q0 + P.min_off, TK + P.span, xf);
        cur_s = cs;
      }
      const bool newg = (cd != cur_g);
      if (newg) {
        if (g_fuse.y) stage_rows<false, 2, 256, true>(Gs, gg, P.Hd, P.Wd, P.ds, cd, P.PW, P.PH, P.total_q, q0, TK, gf);
        else stage_rows<false, 4>(Gs, gg, P.Hd, P.Wd, P.ds, cd, P.PW, P.PH, P.total_q, q0, TK);
        cur_g = cd;
      }
      __syncthreads();
      if (newg) {
        const int col = tid & 63, part = tid >> 6;
#pragma unroll
        for (int r = 0; r < TK / 4; ++r) bsum += Gs[(part * (TK / 4) + r) * 64 + col];
      }
      // blocks of 4 k-steps (rows 8b + 2i + h): one address per operand column and block, the 4 rows as immediate offsets
      const float* gcol = Gs + h * 64 + nj * 32 + l31;
      const float* scol = Ss + h * 64 + mi * 32 + l31;
#pragma unroll 2
      for (int b = 0; b < TK / 8; ++b) {
        float bf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) bf[i] = gcol[(8 * b + 2 * i) * 64];
#pragma unroll
        for (int t = t0; t < t1; ++t) {
          const float* ap = scol + (8 * b + P.toff[t] - P.min_off) * 64;
          float af[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) af[i] = ap[2 * i * 64];
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[i], acc[t], 0, 0, 0);
        }
      }
    }
  }
  // partial[wg][9 (reference tap index)][64 ci][64 co] + [wg][64] bias sums after all workgroups' tap blocks
  float* out = partial + (size_t)blockIdx.x * (NTAPS * 4096);
#pragma unroll
  for (int t = 0; t < NTAPS; ++t) {
    float* o = out + (size_t)P.tw[t] * 4096;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      o[row * 64 + nj * 32 + l31] = acc[t][r];
    }
  }
  __syncthreads();
  float* red = Ss;
  red[tid] = bsum;
  __syncthreads();
  if (tid < 64) {
    float* bout = partial + (size_t)gridDim.x * (NTAPS * 4096) + (size_t)blockIdx.x * 64;
    bout[tid] = red[tid] + red[64 + tid] + red[128 + tid] + red[192 + tid];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Weight-gradient kernel of the stride-2 GATHER programs (conv3: four source classes in tap groups {4, 2, 2, 1}, one destination
// class), software-pipelined.  conv64_wgrad_kernel<true> stages each class synchronously between two barriers — four HBM round
// trips per 64-position chunk in front of 128 / 64 / 64 / 32 MFMAs per wave — and walks (image, row, column) for every staged row
// (~30 vector-ALU instructions per row, 24 rows per thread and chunk against 288 MFMAs).  Here
//  * the rows of the NEXT group's class (the next chunk's class 0 and gradient rows behind the last group) are requested into
//    registers right after the barrier that opens a group's MFMA loop and land in LDS behind the barrier that closes it;
//  * a chunk's rows are decomposed once, into two small tables (source side: rowtab_build; gradient side below), rebuilt for the next
//    chunk in the inter-barrier section of the last group, when nobody reads them.
// Same chunks per workgroup, same MFMA order, same partial layout as conv64_wgrad_kernel<true, 64>: results are bit-identical.
// ---------------------------------------------------------------------------------------------------------------
constexpr int WG_TK = 64;           // positions per chunk
constexpr int WG_SROWS = 8;         // source rows per thread: WG_TK + span <= 128
constexpr int WG_SWORDS = 16 * WG_SROWS, WG_GWORDS = 16 * 4;

// gradient side: entry of row R of the chunk at (R & 15) * 4 + (R >> 4) = pixel index << 1 | 1 (0: outside the tensor)
__device__ __forceinline__ void wg_gtab_build(unsigned* __restrict__ tab, const ConvProg& P, int q0) {
  int R = (int)threadIdx.x - 128;  // (threads 128 .. 191; the source table is built by threads 0 .. 127)
  asm volatile("" : "+v"(R));
  if ((unsigned)R < (unsigned)WG_GWORDS) {
    const int q = q0 + R;
    unsigned e = 0;
    if (q < P.total_q) {
      const int n = fastdiv(q, P.mPHW, P.sPHW);
      const int rem = q - n * P.PHW;
      const int a = fastdiv(rem, P.mPW, P.sPW);
      const int ya = a * P.ds, xb = (rem - a * P.PW) * P.ds;
      if (ya < P.Hd && xb < P.Wd) e = ((unsigned)((n * P.Hd + ya) * P.Wd + xb) << 1) | 1u;
    }
    tab[(R & 15) * 4 + (R >> 4)] = e;
  }
}

__global__ __launch_bounds__(256, 2) void conv64_wgrad_gather_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                                    float* __restrict__ partial, const ConvProg P, int nchunks) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int srows = WG_TK + P.span;
  float* Ss = (float*)smem;                    // (WG_TK + span) x 64: the rows of the current source class
  float* Gs = Ss + srows * 64;                 // WG_TK x 64: the chunk's gradient rows
  unsigned* stab = (unsigned*)(Gs + WG_TK * 64);  // [16][WG_SROWS]
  unsigned* gtab = stab + WG_SWORDS;              // [16][4]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int mi = wave & 1, nj = wave >> 1;

  f32x16 acc[NTAPS];
#pragma unroll
  for (int t = 0; t < NTAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;

  constexpr int GSTART[5] = {0, 4, 6, 8, 9};
  const int cpg = nchunks / P.G;
  f32x4 sv[WG_SROWS], gv[4];
  unsigned sok = 0, gok = 0;

  // rows of source class `cls` of the chunk whose table is in stab -> registers (branch-free; masks applied at the landing)
  auto s_request = [&](const float* __restrict__ xg, int cls) {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    const int slot = t & 15;
    const unsigned delta = (unsigned)((cls >> 1) * P.Ws + (cls & 1));
    const unsigned* __restrict__ tp = stab + (t >> 4) * WG_SROWS;
    const uint4 e0 = *(const uint4*)tp, e1 = *(const uint4*)(tp + 4);
    const unsigned e[WG_SROWS] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
    sok = 0;
#pragma unroll
    for (int j = 0; j < WG_SROWS; ++j) {
      const unsigned m = (unsigned)__builtin_amdgcn_sbfe((int)e[j], (unsigned)cls, 1u);
      sv[j] = *(const f32x4*)(xg + (((((e[j] >> 4) + delta) << 6) & m) + slot * 4));
      sok |= m & (1u << j);
    }
  };
  auto g_request = [&](const float* __restrict__ gg, bool live) {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    const int slot = t & 15;
    const uint4 q = *(const uint4*)(gtab + (t >> 4) * 4);
    const unsigned e[4] = {q.x, q.y, q.z, q.w};
    gok = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      gv[j] = *(const f32x4*)(gg + (((e[j] >> 1) << 6) + slot * 4));
      gok |= (live ? (e[j] & 1u) : 0u) << j;
    }
  };
  auto s_land = [&]() {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    const int slot = t & 15, r = t >> 4;
#pragma unroll
    for (int j = 0; j < WG_SROWS; ++j) {
      const int R = r + 16 * j;
      const f32x4 v = ((sok >> j) & 1u) ? sv[j] : f32x4{0.f, 0.f, 0.f, 0.f};
      if (R < srows) *(f32x4*)(Ss + R * 64 + slot * 4) = v;  // (an LDS write only: no vector-memory operation in a branch)
    }
  };
  auto g_land = [&]() {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    const int slot = t & 15, r = t >> 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) *(f32x4*)(Gs + (r + 16 * j) * 64 + slot * 4) = ((gok >> j) & 1u) ? gv[j] : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto tables = [&](int q0, bool live) {
    // (no rows for a chunk past the end: every entry 0 -> every row reads pixel 0 and lands as zeros)
    if (tid < 128) rowtab_build(stab, WG_SROWS, P, q0 + P.min_off, live ? srows : 0);
    wg_gtab_build(gtab, P, live ? q0 : P.total_q);
  };

  int chunk = blockIdx.x;
  if (chunk < nchunks) {  // the first chunk's class 0 and gradient rows are staged the plain way
    const int grp = (P.G > 1) ? chunk / cpg : 0;
    tables((chunk - grp * cpg) * WG_TK, true);
    __syncthreads();
    s_request(x + grp * P.src_gstride, P.tsrc[0]);
    g_request(g + grp * P.dst_gstride, true);
  }
  for (; chunk < nchunks; chunk += gridDim.x) {
    const int grp = (P.G > 1) ? chunk / cpg : 0;
    const float* __restrict__ xg = x + grp * P.src_gstride;
    const int chunk2 = chunk + (int)gridDim.x;
    const bool more = chunk2 < nchunks;
    const int grp2 = (P.G > 1 && more) ? chunk2 / cpg : grp;
    const int q02 = ((more ? chunk2 : chunk) - grp2 * cpg) * WG_TK;
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) {
      const int t0 = GSTART[gi], t1 = GSTART[gi + 1];
      __syncthreads();  // every wave is done with the previous group's Ss (and, at gi == 0, with the previous chunk's Gs)
      s_land();
      if (gi == 0) g_land();
      if (gi == 3) tables(q02, more);  // (the last request through this chunk's tables went out behind the previous barrier)
      __syncthreads();
      if (gi < 3) s_request(xg, P.tsrc[GSTART[gi + 1]]);
      else {
        s_request(x + grp2 * P.src_gstride, P.tsrc[0]);
        g_request(g + grp2 * P.dst_gstride, more);
      }
      __builtin_amdgcn_sched_barrier(0);  // the requests go out HERE, ahead of the group's MFMAs
      if (gi == 0) {
        const int col = tid & 63, part = tid >> 6;
#pragma unroll
        for (int r = 0; r < WG_TK / 4; ++r) bsum += Gs[(part * (WG_TK / 4) + r) * 64 + col];
      }
      const float* gcol = Gs + h * 64 + nj * 32 + l31;
      const float* scol = Ss + h * 64 + mi * 32 + l31;
#pragma unroll 2
      for (int b = 0; b < WG_TK / 8; ++b) {
        float bf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) bf[i] = gcol[(8 * b + 2 * i) * 64];
#pragma unroll
        for (int t = t0; t < t1; ++t) {
          const float* ap = scol + (8 * b + P.toff[t] - P.min_off) * 64;
          float af[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) af[i] = ap[2 * i * 64];
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[i], acc[t], 0, 0, 0);
        }
      }
    }
  }
  float* out = partial + (size_t)blockIdx.x * (NTAPS * 4096);
#pragma unroll
  for (int t = 0; t < NTAPS; ++t) {
    float* o = out + (size_t)P.tw[t] * 4096;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      o[row * 64 + nj * 32 + l31] = acc[t][r];
    }
  }
  __syncthreads();
  float* red = Ss;
  red[tid] = bsum;
  __syncthreads();
  if (tid < 64) {
    float* bout = partial + (size_t)gridDim.x * (NTAPS * 4096) + (size_t)blockIdx.x * 64;
    bout[tid] = red[tid] + red[64 + tid] + red[128 + tid] + red[192 + tid];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Weight-gradient kernel, ring version (programs with ONE source class: stride-1 conv, transposed conv).
// Same GEMM as above, but
//  * a workgroup walks a CONTIGUOUS range of chunks and keeps the source rows in a 256-row LDS ring: consecutive
//    chunks share TK+span-64 of their TK+span rows (span = 116 for conv2), so each chunk only fetches 64 new source rows
//    instead of 180 (121 -> 64 for the transposed convolutions);
//  * the new source rows and the next gradient rows are requested into registers BEFORE the chunk's MFMA loop and
//    written to LDS after it: their latency sits behind the matrix work (two barriers per group remain).
// ---------------------------------------------------------------------------------------------------------------
// The source ring of conv64_wgrad_ring_kernel: row q of the virtual grid lives at slot q mod RING_ROWS; the first RING_MIRROR
// slots are kept a second time behind the ring (slots RING_ROWS .. RING_ROWS + RING_MIRROR), so that a reader that starts at
// any slot can go on for RING_MIRROR rows without wrapping — the MFMA loop wraps ONE wave-uniform (scalar) row index per tap
// and 4 k-steps and reaches its 4 rows through the immediate offsets of two ds_read2st64_b32.  (The previous layout — 256
// slots, "& 255" on every address — cost three VALU instructions and one ds_read_b32 per MFMA, and that instruction stream,
// not the matrix pipe, bounded the loop: 113 TF with every load and barrier removed.)
// RING_ROWS >= TK + span + TK (the prefetched TK rows are written only after the readers' barrier).
template <int V> struct IntC { static constexpr int value = V; };
constexpr int RING_ROWS = 246, RING_MIRROR = 8, RING = RING_ROWS + RING_MIRROR;  // (246 + 8 + 64 rows + 2 rows of tables = 80 KB)
// the stride-2 (ConvTranspose) kernel walks 32-position chunks: RING_ROWS_S2 >= 32 + span + 32, and 4 x 32 gradient rows next to it
constexpr int RING_ROWS_S2 = 184, RING_S2 = RING_ROWS_S2 + RING_MIRROR;
template <int ROWS = RING_ROWS>
__device__ __forceinline__ int ring_slot(int q) { return (q + 4 * ROWS) % ROWS; }  // q >= -4 * ROWS

// rows [qstart, qstart+64) of class `cls`: 4 rows per thread (16 apart) into registers; okmask bit j = row j in bounds.
// <J0, NJ>: only this thread's rows J0 .. J0+NJ-1 (v[j - J0]); the other bits of okmask are left alone.
template <int J0 = 0, int NJ = 4>
__device__ __forceinline__ void rows64_load(f32x4 (&v)[NJ], unsigned& okmask, const float* __restrict__ src, int H, int W,
                                            int stride, int cls, int PW, int PH, int total_q, int qstart, const GridDiv gd) {
  int t = threadIdx.x;
  asm volatile("" : "+v"(t));  // opaque: nothing derived from the thread index here is worth a register across the caller's loops
  const int slot = t & 15;
  const int cy = cls >> 1, cx = cls & 1;
  const int PHW = PH * PW;
  // (fastdiv: three true divisions here were ~120 VALU instructions per call — per 64-position chunk and thread)
  const int sa = fastdiv(16, gd.mPW, gd.sPW), sb = 16 - sa * PW;
  const int qq = qstart + (t >> 4) + PHW;  // shifted by one image: non-negative for the first rows of the first chunk
  int n1 = fastdiv(qq, gd.mPHW, gd.sPHW);
  int rem = qq - n1 * PHW;
  int a = fastdiv(rem, gd.mPW, gd.sPW);
  int b = rem - a * PW;
  const int N1max = total_q / PHW;
  if (J0 == 0) okmask = 0;
#pragma unroll
  for (int j = 0; j < J0 + NJ; ++j) {
    if (j >= J0) {
      v[j - J0] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int y = (a << (stride - 1)) + cy, x = (b << (stride - 1)) + cx;  // (stride is 1 or 2: a shift-add, not a quarter-rate multiply)
      const bool ok = (unsigned)(n1 - 1) < (unsigned)N1max && y < H && x < W;
      okmask |= (ok ? 1u : 0u) << j;
      if (ok) v[j - J0] = *(const f32x4*)(src + (mad_u24(mad_u24((unsigned)(n1 - 1), H, (unsigned)y), W, (unsigned)x) * 64u + (unsigned)(slot * 4)));
    }
    b += sb; a += sa;
    if (b >= PW) { b -= PW; ++a; }
    if (a >= PH) { a -= PH; ++n1; }
  }
}

// registers -> LDS rows (row index of this thread's j-th row = rbase + 16*j; RINGED: its ring slot, plus the mirror copy);
// bnp != NULL: relu(batchnorm(.)) applied to in-bounds rows on the way (OpFuse forward fusion)
// The scale / shift of a fused operand for this thread's four channels (identity when bnp == NULL): loaded ONCE by the caller — at
// every landing they would be an L2 round trip in front of the LDS writes, once per 32- or 64-position chunk.
struct BnQuad { f32x4 sc, sh; bool on; };
__device__ __forceinline__ BnQuad bn_quad(const float* __restrict__ bnp) {
  BnQuad q = {f32x4{1.f, 1.f, 1.f, 1.f}, f32x4{0.f, 0.f, 0.f, 0.f}, bnp != nullptr};
  const int slot = threadIdx.x & 15;
  if (bnp) { q.sc = *(const f32x4*)(bnp + 128 + slot * 4); q.sh = *(const f32x4*)(bnp + 192 + slot * 4); }
  return q;
}

template <bool RINGED, int J0 = 0, int NJ = 4, int ROWS = RING_ROWS>
__device__ __forceinline__ void rows64_store(float* __restrict__ lds, int rbase, f32x4 (&v)[NJ], unsigned okmask,
                                             const BnQuad& bq) {
  int t = threadIdx.x;
  asm volatile("" : "+v"(t));  // (see rows64_load)
  const int slot = t & 15;
  const f32x4 sc4 = bq.sc, sh4 = bq.sh;
  const bool bnp = bq.on;
#pragma unroll
  for (int j = J0; j < J0 + NJ; ++j) {
    if (bnp && ((okmask >> j) & 1u)) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float z = v[j - J0][e] * sc4[e] + sh4[e]; v[j - J0][e] = z > 0.f ? z : 0.f; }
    }
    int R = rbase + (t >> 4) + 16 * j;
    if (RINGED) R = ring_slot<ROWS>(R);
    *(f32x4*)(lds + R * 64 + slot * 4) = v[j - J0];
    if (RINGED && R < RING_MIRROR) *(f32x4*)(lds + (R + ROWS) * 64 + slot * 4) = v[j - J0];
  }
}

// The 64 rows a chunk of conv64_wgrad_ring_kernel requests per operand, decomposed ONCE (rows64_load does it per row and thread:
// ~22 vector-ALU instructions per row, 8 rows per thread and chunk next to 288 MFMAs): entry of row R at (R & 15) * 4 + (R >> 4) =
// pixel index << 1 | 1, 0 = outside the tensor.  One wave builds one table (lane = row).
__device__ __forceinline__ void ring_tab_build(unsigned* __restrict__ tab, const ConvProg& P, int H, int W, int stride, int cls,
                                               int qstart, int R) {
  const int qq = qstart + R + P.PHW;  // shifted by one image: non-negative for the first rows of the first chunk
  const int n1 = fastdiv(qq, P.mPHW, P.sPHW);
  const int rem = qq - n1 * P.PHW;
  const int a = fastdiv(rem, P.mPW, P.sPW);
  const int y = (a << (stride - 1)) + (cls >> 1), x = ((rem - a * P.PW) << (stride - 1)) + (cls & 1);
  const bool ok = (unsigned)(n1 - 1) < (unsigned)P.N && y < H && x < W;
  tab[(R & 15) * 4 + (R >> 4)] = ok ? ((unsigned)(((n1 - 1) * H + y) * W + x) << 1) | 1u : 0u;
}

// rows64_load through such a table (same loads, same zeros for rows outside)
__device__ __forceinline__ void rows64_load_tab(f32x4 (&v)[4], unsigned& okmask, const float* __restrict__ src,
                                                const unsigned* __restrict__ tab) {
  int t = threadIdx.x;
  asm volatile("" : "+v"(t));
  const float* __restrict__ base = src + (t & 15) * 4;
  const uint4 q = *(const uint4*)(tab + (t >> 4) * 4);
  const unsigned e[4] = {q.x, q.y, q.z, q.w};
  okmask = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    okmask |= (e[j] & 1u) << j;
    if (e[j] & 1u) v[j] = *(const f32x4*)(base + ((e[j] & ~1u) << 5));  // (pixel index * 64 floats)
  }
}

// (Stride-1 programs only: the stride-2 form of this kernel, which walked a 64-position chunk class by class, was superseded by
// conv64_wgrad_ring_s2_kernel in round 2 and removed in round 5 — every program it took, span <= 118, the s2 kernel takes too.)
__global__ __launch_bounds__(256, 2) void conv64_wgrad_ring_kernel(const float* __restrict__ x,
                                                                  const float* __restrict__ g,
                                                                  float* __restrict__ partial, const ConvProg P,
                                                                  int nchunks, int chunks_per_wg, int wgs_per_group,
                                                                  const float* __restrict__ x_bnp) {
  // nchunks / chunks_per_wg describe ONE BatchNorm group; workgroups [g*wgs_per_group, (g+1)*wgs_per_group) walk group g
  constexpr int TK = 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* Ss = (float*)smem;        // ring: source row q lives at slot (q & 255)
  float* Gs = Ss + RING * 64;      // TK x 64: gradient rows of the current (chunk, destination class)
  unsigned* tabx = (unsigned*)(Gs + TK * 64);  // the 64 new source rows / the 64 gradient rows the current chunk requests
  unsigned* tabg = tabx + 64;                  //       (ring_tab_build)

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int mi = wave & 1, nj = wave >> 1;
  // the tables of the requests chunk `c` makes: its successor's 64 new source rows and 64 gradient rows; waves 0 / 1 build one each
  auto next_tables = [&](int c) {
    const int q0 = c * TK;
    if (wave == 0) ring_tab_build(tabx, P, P.Hs, P.Ws, P.ss, P.tsrc[0], q0 + P.min_off + TK + P.span, lane);
    if (wave == 1) ring_tab_build(tabg, P, P.Hd, P.Wd, P.ds, P.tdst[0], q0 + TK, lane);
  };

  f32x16 acc[NTAPS];
#pragma unroll
  for (int t = 0; t < NTAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  // bias gradient (column sums of the gradient rows): every gradient row passes through exactly one thread's registers on its
  // way into Gs, so the sums are taken there (channels [4*slot, 4*slot+4) of this thread's rows) instead of re-reading Gs
  f32x4 bs4 = {0.f, 0.f, 0.f, 0.f};

  const int cs = P.tsrc[0];  // the single source class (and a single destination class: one tap group of nine per chunk)

  const int grp = (P.G > 1) ? blockIdx.x / wgs_per_group : 0;
  x += grp * P.src_gstride;
  g += grp * P.dst_gstride;
  if (x_bnp) x_bnp += grp * 256;
  const BnQuad xq = bn_quad(x_bnp), noq = bn_quad(nullptr);
  const int c_begin = (blockIdx.x - grp * wgs_per_group) * chunks_per_wg;
  const int c_end = (c_begin + chunks_per_wg < nchunks) ? c_begin + chunks_per_wg : nchunks;
  if (c_begin < c_end) {
    // prologue: source rows [q0+min_off, q0+min_off+TK+span) of the first chunk, gradient rows of its first class
    const int q0 = c_begin * TK;
    f32x4 v[4];
    unsigned ok;
    for (int r0 = 0; r0 < TK + P.span; r0 += 64) {
      rows64_load(v, ok, x, P.Hs, P.Ws, P.ss, cs, P.PW, P.PH, P.total_q, q0 + P.min_off + r0, grid_div(P));
      rows64_store<true>(Ss, q0 + P.min_off + r0, v, ok, xq);
    }
    rows64_load(v, ok, g, P.Hd, P.Wd, P.ds, P.tdst[0], P.PW, P.PH, P.total_q, q0, grid_div(P));
    rows64_store<false>(Gs, 0, v, ok, noq);
    bs4 += (v[0] + v[1]) + (v[2] + v[3]);  // (rows outside the tensor are zero)
    next_tables(c_begin);
  }
  __syncthreads();

  unsigned oks = 0;
  for (int chunk = c_begin; chunk < c_end; ++chunk) {
    const int q0 = chunk * TK;
    const bool last_chunk = chunk + 1 >= c_end;
    {
      constexpr int t0 = 0, t1 = NTAPS;
      // ---- requests for what the NEXT chunk needs: its gradient rows and its 64 new source rows (they overwrite ring slots nobody
      //      reads after this chunk: RING_ROWS >= TK + span + TK keeps them outside the window the chunk still reads)
      f32x4 pg[4], ps[4];
      unsigned okg = 0;
      const bool want_g = !last_chunk, want_s = !last_chunk;
      if (want_g) rows64_load_tab(pg, okg, g, tabg);  // (rows q0 + TK ..: what next_tables(chunk) decomposed)
      if (want_s) rows64_load_tab(ps, oks, x, tabx);
      // ---- this group's work
      // 8 blocks of 4 k-steps; k-step i of block b multiplies grid rows q0 + 8b + 2i + h.  Per tap the ring slot of row
      // q0 + toff + 8b is wave-uniform (u[t], wrapped with scalar instructions); the 4 rows of a lane are u + h + {0,2,4,6}
      // — never past the mirror — i.e. one address and two ds_read2st64_b32 per tap and block.
      const float* gcol = Gs + (h * 64 + nj * 32 + l31);
      const float* scol = Ss + (h * 64 + mi * 32 + l31);
      int u[NTAPS];
#pragma unroll
      for (int t = t0; t < t1; ++t) u[t] = ring_slot(q0 + P.toff[t]);
#pragma unroll 1
      for (int b = 0; b < TK / 8; ++b) {
        float bf[4];
        int go = b * (8 * 64);
        asm volatile("" : "+v"(go));  // one address; the 4 rows are immediate offsets (Gs sits 64 KB into LDS: left to fold that
                                      // constant itself, the compiler needs one add and one ds_read_b32 per row).  The opaque value
                                      // is the OFFSET: laundering the pointer would lose the LDS address space (flat loads).
#pragma unroll
        for (int i = 0; i < 4; ++i) bf[i] = gcol[go + 2 * i * 64];
#pragma unroll
        for (int t = t0; t < t1; ++t) {
          const float* ap = scol + u[t] * 64;
          float af[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) af[i] = ap[2 * i * 64];
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[i], acc[t], 0, 0, 0);
          u[t] += 8;
          if (u[t] >= RING_ROWS) u[t] -= RING_ROWS;
        }
      }
      // ---- land the prefetched rows
      __syncthreads();
      if (want_g) {
        rows64_store<false>(Gs, 0, pg, okg, noq);
        bs4 += (pg[0] + pg[1]) + (pg[2] + pg[3]);
      }
      if (want_s) rows64_store<true, 0, 4>(Ss, q0 + P.min_off + TK + P.span, ps, oks, xq);
      if (!last_chunk) next_tables(chunk + 1);  // (this chunk's requests have been issued — and their table reads returned — long ago)
      __syncthreads();
    }
  }
  // partial[wg][9 (reference tap index)][64 ci][64 co] + [wg][64] bias sums after all workgroups' tap blocks
  float* out = partial + (size_t)blockIdx.x * (NTAPS * 4096);
#pragma unroll
  for (int t = 0; t < NTAPS; ++t) {
    float* o = out + (size_t)P.tw[t] * 4096;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      o[row * 64 + nj * 32 + l31] = acc[t][r];
    }
  }
  __syncthreads();  // (every wave is done with the ring)
  float* red = Ss;  // [16 row groups][64 channels]
  *(f32x4*)(red + (tid >> 4) * 64 + (tid & 15) * 4) = bs4;
  __syncthreads();
  if (tid < 64) {
    float* bout = partial + (size_t)gridDim.x * (NTAPS * 4096) + (size_t)blockIdx.x * 64;
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += red[r * 64 + tid];
    bout[tid] = t;
  }
}


// ---------------------------------------------------------------------------------------------------------------
// The same for stride-2 scatter programs (ConvTranspose weight gradients: one source class, four destination classes with
// 4 / 2 / 2 / 1 taps).  (Its predecessor walked a 64-position chunk class by class — four barrier pairs per chunk,
// the last of them around 32 MFMAs per wave.  Here a chunk is 32 positions and carries the gradient rows of ALL four classes
// (4 x 8 KB) next to a 184 + 8 row source ring (48 KB): one barrier pair per 144 MFMAs, every tap of a k-block shares the block's
// loads, and the prefetch (8 gradient + 2 source float4 per thread) travels under a whole chunk.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void conv64_wgrad_ring_s2_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                                     float* __restrict__ partial, const ConvProg P, int nchunks,
                                                                     int chunks_per_wg, int wgs_per_group,
                                                                     const float* __restrict__ x_bnp) {
  constexpr int TK = 32;
  constexpr int GRP[NTAPS] = {0, 0, 0, 0, 1, 1, 2, 2, 3};  // destination-class group of tap t (taps are sorted 4/2/2/1)
  constexpr int GFIRST[4] = {0, 4, 6, 8};
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* Ss = (float*)smem;          // ring: source row q at slot q mod RING_ROWS_S2 (+ mirror)
  float* Gs = Ss + RING_S2 * 64;     // [4 classes][TK rows][64]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int mi = wave & 1, nj = wave >> 1;

  f32x16 acc[NTAPS];
#pragma unroll
  for (int t = 0; t < NTAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  f32x4 bs4 = {0.f, 0.f, 0.f, 0.f};

  const int cs = P.tsrc[0];
  const int grp = (P.G > 1) ? blockIdx.x / wgs_per_group : 0;
  x += grp * P.src_gstride;
  g += grp * P.dst_gstride;
  if (x_bnp) x_bnp += grp * 256;
  const BnQuad xq = bn_quad(x_bnp), noq = bn_quad(nullptr);
  const int c_begin = (blockIdx.x - grp * wgs_per_group) * chunks_per_wg;
  const int c_end = (c_begin + chunks_per_wg < nchunks) ? c_begin + chunks_per_wg : nchunks;

  f32x4 pg[4][2], ps[2];
  unsigned okg[4] = {0, 0, 0, 0}, oks = 0;
  auto g_request = [&](int q0_) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
      rows64_load<0, 2>(pg[c], okg[c], g, P.Hd, P.Wd, P.ds, P.tdst[GFIRST[c]], P.PW, P.PH, P.total_q, q0_, grid_div(P));
  };
  auto g_land = [&]() {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      rows64_store<false, 0, 2>(Gs + c * TK * 64, 0, pg[c], okg[c], noq);
      bs4 += pg[c][0] + pg[c][1];  // (rows outside the tensor are zero)
    }
  };
  if (c_begin < c_end) {
    const int q0 = c_begin * TK;
    for (int r0 = 0; r0 < TK + P.span; r0 += 32) {
      rows64_load<0, 2>(ps, oks, x, P.Hs, P.Ws, P.ss, cs, P.PW, P.PH, P.total_q, q0 + P.min_off + r0, grid_div(P));
      rows64_store<true, 0, 2, RING_ROWS_S2>(Ss, q0 + P.min_off + r0, ps, oks, xq);
    }
    g_request(q0);
    g_land();
  }
  __syncthreads();

  const float* gcol = Gs + (h * 64 + nj * 32 + l31);
  const float* scol = Ss + (h * 64 + mi * 32 + l31);
  for (int chunk = c_begin; chunk < c_end; ++chunk) {
    const int q0 = chunk * TK;
    const bool more = chunk + 1 < c_end;
    if (more) {
      g_request(q0 + TK);
      rows64_load<0, 2>(ps, oks, x, P.Hs, P.Ws, P.ss, cs, P.PW, P.PH, P.total_q, q0 + P.min_off + TK + P.span, grid_div(P));
    }
    int u[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) u[t] = ring_slot<RING_ROWS_S2>(q0 + P.toff[t]);
#pragma unroll 1
    for (int b = 0; b < TK / 8; ++b) {
      int go = b * (8 * 64);
      asm volatile("" : "+v"(go));  // (one address per class; see conv64_wgrad_ring_kernel)
      float bf[4][4];
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i) bf[c][i] = gcol[c * TK * 64 + go + 2 * i * 64];
#pragma unroll
      for (int t = 0; t < NTAPS; ++t) {
        const float* ap = scol + u[t] * 64;
        float af[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = ap[2 * i * 64];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[GRP[t]][i], acc[t], 0, 0, 0);
        u[t] += 8;
        if (u[t] >= RING_ROWS_S2) u[t] -= RING_ROWS_S2;
      }
    }
    __syncthreads();  // every wave is done with this chunk's gradient rows (and with the ring slots the new rows replace)
    if (more) {
      g_land();
      rows64_store<true, 0, 2, RING_ROWS_S2>(Ss, q0 + P.min_off + TK + P.span, ps, oks, xq);
    }
    __syncthreads();
  }
  float* out = partial + (size_t)blockIdx.x * (NTAPS * 4096);
#pragma unroll
  for (int t = 0; t < NTAPS; ++t) {
    float* o = out + (size_t)P.tw[t] * 4096;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      o[row * 64 + nj * 32 + l31] = acc[t][r];
    }
  }
  __syncthreads();
  float* red = Ss;  // [16 row groups][64 channels]
  *(f32x4*)(red + (tid >> 4) * 64 + (tid & 15) * 4) = bs4;
  __syncthreads();
  if (tid < 64) {
    float* bout = partial + (size_t)gridDim.x * (NTAPS * 4096) + (size_t)blockIdx.x * 64;
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += red[r * 64 + tid];
    bout[tid] = t;
  }
}

// dw_ref[...] = sum over workgroups (fixed order); layout: conv [co][ci][3][3], convT [ci][co][3][3].
// 1024 threads per block: 256 outputs x 4 slices of the workgroup range, 4 loads in flight per thread, fp64 combine.
__global__ __launch_bounds__(1024) void conv64_wgrad_reduce(const float* __restrict__ partial, int nwg,
                                                           float* __restrict__ dw_ref, float* __restrict__ dbias,
                                                           int transposed, int interleaved) {
  const int o = threadIdx.x & 255, part = threadIdx.x >> 8;
  const int id = blockIdx.x * 256 + o;
  constexpr int TOT = NTAPS * 4096;
  const int per = (nwg + 3) / 4;
  const int w0 = part * per, w1 = (w0 + per < nwg) ? w0 + per : nwg;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (id < TOT + 64) {
    // bias partials live after all workgroups' tap blocks: [nwg][64] — or, interleaved (= the record size), behind each workgroup's own
    const float* base = interleaved ? partial + id : (id < TOT) ? partial + id : partial + (size_t)nwg * TOT + (id - TOT);
    const size_t stride = interleaved ? (size_t)interleaved : (id < TOT) ? (size_t)TOT : 64;
    int w = w0;
    for (; w + 3 < w1; w += 4) {
      s0 += (double)base[(size_t)w * stride];
      s1 += (double)base[(size_t)(w + 1) * stride];
      s2 += (double)base[(size_t)(w + 2) * stride];
      s3 += (double)base[(size_t)(w + 3) * stride];
    }
    for (; w < w1; ++w) s0 += (double)base[(size_t)w * stride];
  }
  __shared__ double sm[4][256];
  sm[part][o] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (part == 0 && id < TOT + 64) {
    const double s = (sm[0][o] + sm[1][o]) + (sm[2][o] + sm[3][o]);
    if (id < TOT) {
      const int tap = id >> 12, ci = (id >> 6) & 63, co = id & 63;
      dw_ref[transposed ? ((ci * 64 + co) * 9 + tap) : ((co * 64 + ci) * 9 + tap)] = (float)s;
    } else if (dbias) {
      dbias[id - TOT] = (float)s;
    }
  }
}

// w_ref -> packed [tap][n_out][slot ^ (n_out&15)][4]; fwd: (n=co,k=ci), bwd: (n=ci,k=co)
__global__ void conv64_pack_kernel(const float* __restrict__ w_ref, float* __restrict__ pf, float* __restrict__ pb,
                                   int transposed) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= NTAPS * 4096) return;
  const int tap = id >> 12, n = (id >> 6) & 63, k = id & 63;
  const int o = (tap * 64 + n) * 64 + ((((k >> 2) ^ (n & 15)) << 2) | (k & 3));
  // reference element [A][B][tap]: conv A=co,B=ci ; convT A=ci,B=co
  const int fwd_idx = transposed ? ((k * 64 + n) * 9 + tap) : ((n * 64 + k) * 9 + tap);
  const int bwd_idx = transposed ? ((n * 64 + k) * 9 + tap) : ((k * 64 + n) * 9 + tap);
  if (pf) pf[o] = w_ref[fwd_idx];
  if (pb) pb[o] = w_ref[bwd_idx];
}

static int check_desc(const srlz_conv64_desc* d) {
  SRLZ_REQUIRE(d != nullptr, SRLZ_ERR_NULL, "conv64: null descriptor");
  SRLZ_REQUIRE(d->ksize == 3 && (d->stride == 1 || d->stride == 2) && d->n > 0, SRLZ_ERR_BAD_DESC,
               "conv64: only 3x3 stride 1/2 supported (k=%d s=%d)", d->ksize, d->stride);
  SRLZ_REQUIRE(d->groups >= 0 && (d->groups <= 1 || d->n % d->groups == 0), SRLZ_ERR_BAD_DESC,
               "conv64: n = %d is not a multiple of groups = %d", d->n, d->groups);
  int eho, ewo;
  if (!d->transposed) {
    eho = (d->hi + 2 * d->pad - 3) / d->stride + 1;
    ewo = (d->wi + 2 * d->pad - 3) / d->stride + 1;
  } else {
    eho = (d->hi - 1) * d->stride - 2 * d->pad + 3;
    ewo = (d->wi - 1) * d->stride - 2 * d->pad + 3;
  }
  SRLZ_REQUIRE(eho == d->ho && ewo == d->wo, SRLZ_ERR_BAD_DESC, "conv64: output size %dx%d inconsistent (expected %dx%d)",
               d->ho, d->wo, eho, ewo);
  return 0;
}

static int program_for(ConvProg* P, const srlz_conv64_desc* d, int backward_data) {
  int rc;
  const int G = d->groups > 1 ? d->groups : 1;
  if (!backward_data)
    rc = build_program(P, !d->transposed, d->stride, d->pad, d->n, d->hi, d->wi, d->ho, d->wo, G);
  else
    rc = build_program(P, d->transposed, d->stride, d->pad, d->n, d->ho, d->wo, d->hi, d->wi, G);
  SRLZ_REQUIRE(rc == 0, SRLZ_ERR_BAD_DESC, "conv64: cannot build a grid program for this descriptor");
  return 0;
}

static size_t fwd_lds_bytes(const ConvProg& P) {  // source rows, slab, row table, rowinfo, fused-operand coefficients
  return (size_t)(TM + P.span) * 256 + 16384 + (size_t)64 * rowtab_passes(TM + P.span) + TM * 4 + 256 * 4;
}
// source rows, slab, row table, rowinfo, (scale, shift) of up to 8 input-channel blocks
static size_t convn_lds_bytes(const ConvProg& P) {
  return (size_t)(TM + P.span) * 256 + 16384 + (size_t)64 * rowtab_passes(TM + P.span) + TM * 4 + 8 * 128 * 4;
}
static size_t wgrad_lds_bytes(const ConvProg& P, int tk) { return (size_t)(tk + P.span + tk) * 256; }
static int wgrad_tk(const ConvProg& P) { (void)P; return 64; }  // 128 was tried: the extra staging registers spill

// conv64_gather_pipe_kernel takes a program when it is a stride-2 gather with the tap groups {4, 2, 2, 1}, at most two BatchNorm groups,
// a staged class of at most 192 rows (PW <= 63), 32-bit byte offsets — and at least 256 tiles PER BatchNorm GROUP: the pipeline pays
// when a workgroup walks several tiles (conv3 forward at bs = 256: 450 tiles per group, 102 -> 92 us); with about one tile per
// workgroup the 4-wave synchronous kernel is faster (bs = 32: 33 us against 51).  Per group, not per launch: one group alone and the
// batched pair of a step must take the same kernel (their statistics are compared bit for bit, and the two kernels sum a tile's
// partial in different orders).
constexpr int GATHER_PIPE_MIN_TILES_PER_GROUP = 256;
static bool gather_pipe_ok(const ConvProg& P) {
  if (P.tpg < GATHER_PIPE_MIN_TILES_PER_GROUP) return false;
  bool grouped = P.s2 && P.ss == 2 && P.G <= 2 && !P.dbg;
  for (int t = 0; t < NTAPS; ++t) grouped = grouped && P.tsrc[t] == P.tsrc[t < 4 ? 0 : t < 6 ? 4 : t < 8 ? 6 : 8] && P.tdst[t] == 0;
  const bool fits32 = P.src_gstride * 4 < (1LL << 32) - 65536 && P.dst_gstride * 4 < (1LL << 32) - 65536;
  return grouped && fits32 && TM + P.span <= GP_RP * GP_BATCH && GP_RP <= P.PHW && -P.min_off <= P.PHW;
}

static int launch_fwd(const float* src, const float* wpack, const float* bias, float* dst, float* stats,
                      const ConvProg& P, hipStream_t st, const OpFuse src_fuse = SRLZ_NO_FUSE, const PoolSum* psum = nullptr) {
  const int ntiles = P.G * P.tpg;
  const size_t lds = fwd_lds_bytes(P);
  SRLZ_REQUIRE(lds <= 160 * 1024, SRLZ_ERR_BAD_DESC, "conv64: tile needs %zu bytes of LDS", lds);
  // the row table keeps pixel indices in 28 bits and the staging 32-bit float offsets; the fast divisions take dividends below 2^31
  SRLZ_REQUIRE((long long)P.N * P.Hs * P.Ws * 64 < (1LL << 32) && (long long)P.N * P.Hd * P.Wd < (1LL << 29) &&
                   (long long)P.total_q + P.PHW + TM + P.span < (1LL << 31),
               SRLZ_ERR_BAD_DESC, "conv64: a group of %d images of %d x %d -> %d x %d is beyond the tile tables' 32-bit offsets", P.N, P.Hs,
               P.Ws, P.Hd, P.Wd);
  if (psum) {  // data gradient whose epilogue takes the pooled block's BatchNorm-backward sums (its own instantiation)
    SRLZ_REQUIRE(!src_fuse.y && !src_fuse.bnp && stats && !bias, SRLZ_ERR_BAD_DESC, "conv64: the pooled-block epilogue takes a plain operand");
    bool one_class = true;
    for (int t = 1; t < NTAPS; ++t) one_class = one_class && P.tdst[t] == P.tdst[0];
    if (one_class) {
      SRLZ_MAX_LDS(conv64_dgrad_poolsum_kernel<1>, lds);
      hipLaunchKernelGGL(conv64_dgrad_poolsum_kernel<1>, dim3(ntiles), dim3(256), lds, st, src, wpack, dst, stats, P, ntiles, *psum);
    } else {
      SRLZ_MAX_LDS(conv64_dgrad_poolsum_kernel<2>, lds);
      hipLaunchKernelGGL(conv64_dgrad_poolsum_kernel<2>, dim3(ntiles), dim3(256), lds, st, src, wpack, dst, stats, P, ntiles, *psum);
    }
    SRLZ_LAUNCHED();
    return 0;
  }
  // 4 waves (32x64 per wave); 8 waves of 32x32 were measured within +-3 % (the kernel is bound by the power-limited matrix
  // rate, not by latency hiding) and are not instantiated any more
#define SRLZ_FWD_LAUNCH(NWV, BWDV)                                                                                          \
  do {                                                                                                                     \
    SRLZ_MAX_LDS((conv64_fwd_kernel<NWV, BWDV>), lds);                                                                      \
    hipLaunchKernelGGL((conv64_fwd_kernel<NWV, BWDV>), dim3(ntiles), dim3(NWV * 64), lds, st, src, wpack, bias, dst, stats, \
                       P, ntiles, src_fuse);                                                                               \
  } while (0)
  if (src_fuse.y) {
    SRLZ_REQUIRE((long long)P.N * P.Hs * P.Ws * 64 < (1LL << 32), SRLZ_ERR_BAD_DESC,
                 "conv64: a group's operand has %lld floats (the fused staging keeps 32-bit row offsets)", (long long)P.N * P.Hs * P.Ws * 64);
    // (a fused BatchNorm-backward operand that must also be STORED for a separate weight-gradient launch: the shapes
    // conv64_bwd_fused_kernel does not take — fewer than 8 tiles, a low-resolution grid wider than 63)
    SRLZ_FWD_LAUNCH(4, true);
  } else {
    // plain stride-2 gather programs with many tiles (conv3 forward at training batch sizes): the software-pipelined persistent
    // kernel (gather_pipe_ok: a per-group criterion, so that one BatchNorm group alone and the batched pair take the same kernel)
    if (!src_fuse.bnp && !bias && gather_pipe_ok(P)) {
      int pgrid = 2 * srlz_device_cus();
      if (pgrid > ntiles) pgrid = ntiles;
      pgrid &= ~7;
      if (pgrid < 8) pgrid = 8;  // (the XCD walk wants a multiple of 8 workgroups; those without a tile leave at once)
      const size_t plds = (size_t)(TM + P.span) * 256 + 16384 + 6 * TM * 4 + GT_WORDS * 4 + 8 * 64 * 4;
      SRLZ_MAX_LDS(conv64_gather_pipe_kernel, plds);
      hipLaunchKernelGGL(conv64_gather_pipe_kernel, dim3(pgrid), dim3(GP_THREADS), plds, st, src, wpack, dst, stats, P, ntiles);
      SRLZ_LAUNCHED();
      return 0;
    }
    SRLZ_FWD_LAUNCH(4, false);
  }
#undef SRLZ_FWD_LAUNCH
  SRLZ_LAUNCHED();
  return 0;
}

// host view of the fused BatchNorm-backward operand (include/srlz.h: srlz_bn_bwd_operand)
static int make_bwd_fuse(OpFuse* f, const srlz_bn_bwd_operand* o, const char* who) {
  *f = SRLZ_NO_FUSE;
  if (!o) return 0;
  SRLZ_REQUIRE(o->y && o->bnp && o->sums && o->count > 0, SRLZ_ERR_NULL, "%s: incomplete srlz_bn_bwd_operand", who);
  f->bnp = o->bnp; f->y = o->y; f->sums = o->sums; f->training = o->training;
  f->inv_count = 1.0f / (float)(double)o->count;
  f->dy_out = o->dy_out;
  return 0;
}

// workgroups of the weight-gradient kernels (all groups together); a multiple of P.G
// conv64_wgrad_gather_kernel: four source classes in tap groups {4, 2, 2, 1}, one destination class, chunk + halo within its registers,
// 32-bit offsets
static bool wgrad_gather_ok(const ConvProg& P) {
  bool grouped = P.s2 && P.ss == 2 && !P.dbg;
  for (int t = 0; t < NTAPS; ++t) grouped = grouped && P.tsrc[t] == P.tsrc[t < 4 ? 0 : t < 6 ? 4 : t < 8 ? 6 : 8] && P.tdst[t] == 0;
  return grouped && WG_TK + P.span <= 16 * WG_SROWS && (long long)P.N * P.Hs * P.Ws * 64 < (1LL << 32) &&
         (long long)P.N * P.Hd * P.Wd * 64 < (1LL << 32) && (long long)P.total_q + P.PHW + WG_TK + P.span < (1LL << 31);
}

static int wgrad_grid(const ConvProg& P) {
  const int tk = wgrad_tk(P);
  const int nchunks = (P.total_q + tk - 1) / tk;  // per group
  int g = 2 * srlz_device_cus() / P.G;           // two persistent workgroups per CU; per group
  // (Fewer, longer-running workgroups on the small layers — at least 8 chunks each, to halve the 147 KB partial every workgroup
  // leaves for the second stage — were measured in round 4 and lost: conv3's weight gradient 121 -> 172 us, ConvT1's 31 -> 102 us,
  // the bs = 32 step 2.49 -> 2.70 ms.  These launches are bound by how many CUs work, not by the partials' traffic.)
  if (g > nchunks) g = nchunks;
  if (g < 1) g = 1;
  return g * P.G;
}


// ---------------------------------------------------------------------------------------------------------------
// convN: the same virtual-grid implicit GEMM for Cin, Cout in {64, 128, 256, 512} — FORWARD ONLY.  It exists for the frozen
// ResNet-18 trunk of EmbeddingNet (/root/reference/models/triplet.py:6-39 -> torchvision resnet18: 3x3 convolutions of
// stride 1 / 2 and 1x1 stride-2 downsample convolutions, all without bias); nothing is ever back-propagated through it.
// blockIdx.y = block of 64 output channels; the input channels are walked in blocks of 64 around the nine taps with the
// accumulators kept; weights are packed [cout block][cin block][tap][64][64].  A 1x1 stride-2 convolution runs as the 3x3
// stride-2 pad-1 program with only the centre tap non-zero (same output size, same sampled pixels; its 8 zero taps cost
// ~4 % of the trunk's FLOP).  x_bnp: one 256-float BatchNorm record per block of 64 input channels.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void convN_fwd_kernel(const float* __restrict__ src, const float* __restrict__ wpack,
                                                          float* __restrict__ dst, float* __restrict__ stats_partial,
                                                          const ConvProg P, int ntiles, int nci, int nco,
                                                          const float* __restrict__ src_bnp, int only_tap, int cshift) {
  // Round 5: rebuilt on the 64-channel family's machinery (conv64_fwd_body) — the tile's row table in LDS (one decomposition per row
  // and tile instead of one per row, thread, class AND input-channel block), the BatchNorm coefficients of every input-channel block
  // in LDS once per workgroup, operand fragments double-buffered in registers, the epilogue through a wave-private LDS transpose with
  // 16-byte stores.  Same tiles and the same accumulation order as the round-2 kernel: outputs bit-identical to it.
  // only_tap >= 0: the program's tap index of the ONE tap whose weights are not zero — a 1x1 stride-2 convolution (ResNet's downsample
  // branch) is the 3x3 stride-2 pad-1 program's centre tap; the other eight used to be multiplied through as zeros (4 % of the trunk).
  // cshift = log2(input channels).
  constexpr int NT = 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* As = (float*)smem;
  float* Bs = As + (TM + P.span) * 64;
  const int tpa = rowtab_passes(TM + P.span);
  unsigned* rowtab = (unsigned*)(Bs + 4096);
  int* rowinfo = (int*)(rowtab + 16 * tpa);      // [TM]: output pixel index, or -1
  float* frec = (float*)(rowinfo + TM);          // [nci][128]: scale, shift of every input-channel block (fused relu(bn(.)) operand)

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int tile = xcd_remap(blockIdx.x, ntiles);
  const int co = blockIdx.y;
  const int cout = nco * 64;
  // BatchNorm groups (round 6: the six views of a time-contrastive step batched along n — models/learner.py:383-391 calls the trunk
  // once per view): tiles [g * tpg, (g + 1) * tpg) cover group g's own virtual grid, exactly as in conv64_fwd_body — a tile's
  // statistics partial belongs to one group, its fused operand uses that group's records ([group][input-channel block][256])
  const int grp = (P.G > 1) ? tile / P.tpg : 0;
  const int q0 = (tile - grp * P.tpg) * TM;
  src += (size_t)grp * P.src_gstride * nci;
  dst += (size_t)grp * P.dst_gstride * nco;
  if (src_bnp) src_bnp += (size_t)grp * nci * 256;

  if (tid < TM) {
    const int q = q0 + tid;
    int ri = -1;
    if (q < P.total_q) {
      const int n = fastdiv(q, P.mPHW, P.sPHW);
      const int rem = q - n * P.PHW;
      const int a = fastdiv(rem, P.mPW, P.sPW);
      const int ya = a * P.ds, xb = (rem - a * P.PW) * P.ds;
      if (ya < P.Hd && xb < P.Wd) ri = (n * P.Hd + ya) * P.Wd + xb;
    }
    rowinfo[tid] = ri;
  }
  rowtab_build(rowtab, tpa, P, q0 + P.min_off, TM + P.span);
  if (src_bnp)
    for (int i = tid; i < nci * 128; i += NT) frec[i] = src_bnp[(i >> 7) * 256 + 128 + (i & 127)];

  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  constexpr int BV = 1024 / NT;
  const int bslot = wave * (BV * 64) + lane;  // wave w moves (and may later scribble on) its own 4 KB of the slab
  f32x4 breg[BV];
  const float* wbase = wpack + (size_t)co * nci * NTAPS * 4096;
  const int t_first = only_tap >= 0 ? only_tap : 0, t_count = only_tap >= 0 ? 1 : NTAPS;
  {
    const f32x4* wsrc = (const f32x4*)(wbase + (size_t)P.tw[t_first] * 4096);
#pragma unroll
    for (int i = 0; i < BV; ++i) breg[i] = wsrc[bslot + i * 64];
  }
  const int nsteps = nci * t_count;
  int ci = 0, tk = 0;       // the step's input-channel block and its tap (tk-th of the block's t_count)
  int cur_src = -1;
  for (int step = 0; step < nsteps; ++step) {
    const int ti = t_first + tk;
    const int tsrc = P.tsrc[ti], toff = P.toff[ti];
    // the next step's (block, tap): its weight slab is requested behind this step's second barrier
    int tk2 = tk + 1, ci2 = ci;
    if (tk2 == t_count) { tk2 = 0; ci2 = ci + 1; }
    __syncthreads();  // all waves are done with the previous step's Bs (and with As if it is about to be replaced)
    if (tk == 0 || tsrc != cur_src) {
      stage_rows_tab<BATCH_FWD>(As, src, rowtab, tpa, tsrc, P.Ws, TM + P.span, src_bnp ? frec + ci * 128 : nullptr, cshift, ci * 64);
      cur_src = tsrc;
    }
    {
      f32x4* wdst = (f32x4*)Bs;
#pragma unroll
      for (int i = 0; i < BV; ++i) wdst[bslot + i * 64] = breg[i];
    }
    __syncthreads();
    {  // (past the last step: block 0, first tap again — never a run-time condition around loads, see conv64_fwd_body)
      const int cn = ci2 < nci ? ci2 : 0;
      const f32x4* wsrc = (const f32x4*)(wbase + ((size_t)cn * NTAPS + P.tw[t_first + tk2]) * 4096);
#pragma unroll
      for (int i = 0; i < BV; ++i) breg[i] = wsrc[bslot + i * 64];
    }
    __builtin_amdgcn_sched_barrier(0);  // the slab requests go out HERE, ahead of the step's MFMAs
    const int R = wave * 32 + l31 + toff - P.min_off;
    int abase = (R * 64 + ((h ^ (R & 15)) << 2)) * 4;  // bytes; slot (2kc + h) ^ (R & 15) is this XOR (kc << 5)
    asm volatile("" : "+v"(abase));
    const float* brow = Bs + l31 * 64;
    const int bkey = lane & 15;
    f32x4 a = *(const f32x4*)((const char*)As + abase);
    f32x4 b[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) b[j] = *(const f32x4*)(brow + j * 2048 + ((h ^ bkey) << 2));
#pragma unroll
    for (int kc = 0; kc < 8; ++kc) {
      f32x4 an = a, bn[2] = {b[0], b[1]};
      if (kc < 7) {
        const int slot = (kc + 1) * 2 + h;
        an = *(const f32x4*)((const char*)As + (abase ^ ((kc + 1) << 5)));
#pragma unroll
        for (int j = 0; j < 2; ++j) bn[j] = *(const f32x4*)(brow + j * 2048 + ((slot ^ bkey) << 2));
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the prefetch reads ahead of this chunk's MFMAs
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], b[j][r], acc[j], 0, 0, 0);
      a = an; b[0] = bn[0]; b[1] = bn[1];
    }
    tk = tk2; ci = ci2;
  }
  __syncthreads();  // every wave is done with the last slab: Bs becomes scratch
  // epilogue: 16 tile rows at a time through this wave's 4 KB of the idle slab, so that every global store is 16 bytes per lane
  // (lane = (row group eg = lane >> 4, channels 4 eslot ..)); the BatchNorm partials in the same layout (see conv64_fwd_body::flush16)
  f32x4 s4 = {0.f, 0.f, 0.f, 0.f}, q4 = {0.f, 0.f, 0.f, 0.f};
  {
    float* S = Bs + wave * 1024;
    const int eg = lane >> 4, eslot = lane & 15;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        const int rowl = (rr & 3) + 8 * (rr >> 2) + 4 * h;
        const int swz = (rowl & 4) << 3;
#pragma unroll
        for (int j = 0; j < 2; ++j) S[rowl * 64 + ((32 * j + l31) ^ swz)] = acc[j][8 * half + rr];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int rowl = eg + 4 * k;
        const int row = wave * 32 + 16 * half + rowl;
        const f32x4 v = *(const f32x4*)(S + rowl * 64 + ((eslot ^ ((rowl & 4) << 1)) << 2));
        const int ri = rowinfo[row];
        if (ri >= 0) {
          *(f32x4*)(dst + (size_t)ri * cout + co * 64 + eslot * 4) = v;
#pragma unroll
          for (int e = 0; e < 4; ++e) { s4[e] += v[e]; q4[e] += v[e] * v[e]; }
        }
      }
    }
  }
  if (stats_partial) {
    __syncthreads();
    float* red = Bs;  // [4 waves][128]
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s4[e] += __shfl_xor(s4[e], 16, 64); s4[e] += __shfl_xor(s4[e], 32, 64);
      q4[e] += __shfl_xor(q4[e], 16, 64); q4[e] += __shfl_xor(q4[e], 32, 64);
    }
    if (lane < 16) {
      *(f32x4*)(red + wave * 128 + lane * 4) = s4;
      *(f32x4*)(red + wave * 128 + 64 + lane * 4) = q4;
    }
    __syncthreads();
    // chunk-major: the partial records of one 64-channel block are contiguous (what srlz_bn_finalize_chunks reduces)
    if (tid < 128) stats_partial[((size_t)co * ntiles + tile) * 128 + tid] = red[tid] + red[128 + tid] + red[256 + tid] + red[384 + tid];
  }
}

// w_ref [Cout][Cin][k][k] (k = 3 or 1) -> packed [cout block][cin block][tap][n][swizzled k]; a 1x1 kernel becomes the centre tap
__global__ void convN_pack_kernel(const float* __restrict__ w_ref, float* __restrict__ pf, int nci, int nco, int ksize) {
  const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)nco * nci * NTAPS * 4096;
  if (id >= total) return;
  const int k = (int)(id & 63), n = (int)((id >> 6) & 63);
  const long long blk = id >> 12;
  const int tap = (int)(blk % NTAPS);
  const int ci = (int)((blk / NTAPS) % nci), co = (int)(blk / NTAPS / nci);
  const int cin = nci * 64;
  float v;
  if (ksize == 3) v = w_ref[((size_t)(co * 64 + n) * cin + ci * 64 + k) * 9 + tap];
  else v = (tap == 4) ? w_ref[(size_t)(co * 64 + n) * cin + ci * 64 + k] : 0.f;
  pf[blk * 4096 + n * 64 + ((((k >> 2) ^ (n & 15)) << 2) | (k & 3))] = v;
}

static int check_convn(const srlz_convn_desc* d) {
  SRLZ_REQUIRE(d != nullptr, SRLZ_ERR_NULL, "convn: null descriptor");
  SRLZ_REQUIRE(d->n > 0 && d->cin > 0 && d->cout > 0 && d->cin % 64 == 0 && d->cout % 64 == 0, SRLZ_ERR_BAD_DESC,
               "convn: channels must be multiples of 64 (cin=%d cout=%d)", d->cin, d->cout);
  SRLZ_REQUIRE(d->groups >= 0 && (d->groups <= 1 || d->n % d->groups == 0), SRLZ_ERR_BAD_DESC,
               "convn: n = %d is not a multiple of groups = %d", d->n, d->groups);
  const bool k3 = d->ksize == 3 && d->pad == 1 && (d->stride == 1 || d->stride == 2);
  const bool k1 = d->ksize == 1 && d->pad == 0 && d->stride == 2;
  SRLZ_REQUIRE(k3 || k1, SRLZ_ERR_BAD_DESC, "convn: 3x3 pad 1 stride 1/2 or 1x1 stride 2 only (k=%d s=%d p=%d)", d->ksize, d->stride,
               d->pad);
  const int eho = (d->hi + 2 * d->pad - d->ksize) / d->stride + 1, ewo = (d->wi + 2 * d->pad - d->ksize) / d->stride + 1;
  SRLZ_REQUIRE(eho == d->ho && ewo == d->wo, SRLZ_ERR_BAD_DESC, "convn: output size %dx%d inconsistent (expected %dx%d)", d->ho,
               d->wo, eho, ewo);
  return 0;
}

static int convn_program(ConvProg* P, const srlz_convn_desc* d) {
  // (a 1x1 stride-2 pad-0 convolution samples exactly the centre-tap pixels of the 3x3 stride-2 pad-1 program)
  const int rc = build_program(P, 1, d->stride, 1, d->n, d->hi, d->wi, d->ho, d->wo, d->groups > 1 ? d->groups : 1);
  SRLZ_REQUIRE(rc == 0, SRLZ_ERR_BAD_DESC, "convn: cannot build a grid program for this descriptor");
  return 0;
}

}  // namespace

extern "C" size_t srlz_convn_packed_floats(const srlz_convn_desc* d) {
  if (check_convn(d)) return 0;
  return (size_t)(d->cin / 64) * (d->cout / 64) * NTAPS * 4096;
}

extern "C" int srlz_convn_pack_weights(const float* w_ref, float* wpack, const srlz_convn_desc* d, srlz_stream_t stream) {
  if (int rc = check_convn(d)) return rc;
  SRLZ_REQUIRE(w_ref && wpack, SRLZ_ERR_NULL, "convn_pack: null pointer");
  const long long total = (long long)(d->cin / 64) * (d->cout / 64) * NTAPS * 4096;
  hipLaunchKernelGGL(convN_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), w_ref, wpack,
                     d->cin / 64, d->cout / 64, d->ksize);
  SRLZ_LAUNCHED();
  return 0;
}

extern "C" int srlz_convn_fwd_tiles(const srlz_convn_desc* d) {
  if (check_convn(d)) return -1;
  ConvProg P;
  if (convn_program(&P, d)) return -1;
  return P.G * P.tpg;
}

extern "C" int srlz_convn_fwd(const float* x, const float* wpack, float* y, float* stats_partial, const float* x_bnp,
                              const srlz_convn_desc* d, srlz_stream_t stream) {
  if (int rc = check_convn(d)) return rc;
  SRLZ_REQUIRE(x && wpack && y, SRLZ_ERR_NULL, "convn_fwd: null pointer");
  ConvProg P;
  if (int rc = convn_program(&P, d)) return rc;
  const int ntiles = P.G * P.tpg;
  const size_t lds = convn_lds_bytes(P);
  SRLZ_REQUIRE(lds <= 160 * 1024, SRLZ_ERR_BAD_DESC, "convn: tile needs %zu bytes of LDS", lds);
  SRLZ_MAX_LDS(convN_fwd_kernel, lds);
  int cshift = 6;
  while ((1 << cshift) < d->cin) ++cshift;
  SRLZ_REQUIRE((1 << cshift) == d->cin && d->cin <= 512, SRLZ_ERR_BAD_DESC, "convn: %d input channels (a power of two from 64 to 512)", d->cin);
  // (the row table keeps pixel indices in 28 bits, the staging 32-bit float offsets)
  // (per BatchNorm group: P.N images)
  SRLZ_REQUIRE((long long)P.N * d->hi * d->wi * d->cin < (1LL << 32) && (long long)P.N * d->hi * d->wi < (1LL << 28), SRLZ_ERR_BAD_DESC,
               "convn: a group of %d images of %d x %d x %d is beyond the tile tables' 32-bit offsets", P.N, d->hi, d->wi, d->cin);
  int only_tap = -1;
  if (d->ksize == 1) {  // the tap of the 3x3 program that carries the 1x1 kernel: weight slab 4 = (ky, kx) = (1, 1)
    for (int t = 0; t < NTAPS; ++t)
      if (P.tw[t] == 4) only_tap = t;
    SRLZ_REQUIRE(only_tap >= 0, SRLZ_ERR_BAD_DESC, "convn: no centre tap in the program of a 1x1 convolution");
  }
  hipLaunchKernelGGL(convN_fwd_kernel, dim3(ntiles, d->cout / 64), dim3(256), lds, as_stream(stream), x, wpack, y, stats_partial, P,
                     ntiles, d->cin / 64, d->cout / 64, x_bnp, only_tap, cshift);
  SRLZ_LAUNCHED();
  return 0;
}

extern "C" size_t srlz_conv64_packed_floats(void) { return (size_t)NTAPS * 4096; }

extern "C" int srlz_conv64_pack_weights(const float* w_ref, float* wpack_fwd, float* wpack_bwd,
                                        const srlz_conv64_desc* d, srlz_stream_t stream) {
  if (int rc = check_desc(d)) return rc;
  SRLZ_REQUIRE(w_ref, SRLZ_ERR_NULL, "conv64_pack: null weights");
  hipLaunchKernelGGL(conv64_pack_kernel, dim3((NTAPS * 4096 + 255) / 256), dim3(256), 0, as_stream(stream), w_ref,
                     wpack_fwd, wpack_bwd, d->transposed);
  SRLZ_LAUNCHED();
  return 0;
}

extern "C" int srlz_conv64_fwd_tiles(const srlz_conv64_desc* d) {
  if (check_desc(d)) return -1;
  ConvProg P;
  if (program_for(&P, d, 0)) return -1;
  return P.G * P.tpg;
}

extern "C" int srlz_conv64_fwd(const float* x, const float* wpack_fwd, const float* bias, float* y,
                               float* stats_partial, const float* x_bnp, const srlz_conv64_desc* d,
                               srlz_stream_t stream) {
  if (int rc = check_desc(d)) return rc;
  SRLZ_REQUIRE(x && wpack_fwd && y, SRLZ_ERR_NULL, "conv64_fwd: null pointer");
  ConvProg P;
  if (int rc = program_for(&P, d, 0)) return rc;
  return launch_fwd(x, wpack_fwd, bias, y, stats_partial, P, as_stream(stream), OpFuse{x_bnp, nullptr, nullptr, 0.f, 0, nullptr});
}

extern "C" int srlz_conv64_bwd_data(const float* dy, const float* wpack_bwd, float* dx, const srlz_bn_bwd_operand* dy_bn,
                                    const srlz_conv64_desc* d, srlz_stream_t stream) {
  if (int rc = check_desc(d)) return rc;
  SRLZ_REQUIRE(dy && wpack_bwd && dx, SRLZ_ERR_NULL, "conv64_bwd_data: null pointer");
  ConvProg P;
  if (int rc = program_for(&P, d, 1)) return rc;
  OpFuse gf;
  if (int rc = make_bwd_fuse(&gf, dy_bn, "conv64_bwd_data")) return rc;
  return launch_fwd(dy, wpack_bwd, nullptr, dx, nullptr, P, as_stream(stream), gf);
}

extern "C" int srlz_conv64_bwd_data_tiles(const srlz_conv64_desc* d) {
  if (check_desc(d)) return -1;
  ConvProg P;
  if (program_for(&P, d, 1)) return -1;
  return P.G * P.tpg;
}

extern "C" int srlz_conv64_bwd_data_pool_sums(const float* dy, const float* wpack_bwd, float* dx, const float* pooled,
                                              const float* pool_bnp, const float* pool_y, const uint8_t* pool_argmax,
                                              const srlz_pool_desc* pd, float* bn_bwd_partial, const srlz_conv64_desc* d,
                                              srlz_stream_t stream) {
  if (int rc = check_desc(d)) return rc;
  SRLZ_REQUIRE(dy && wpack_bwd && dx && pooled && pool_bnp && pool_y && pool_argmax && pd && bn_bwd_partial, SRLZ_ERR_NULL,
               "conv64_bwd_data_pool_sums: null pointer");
  // dx (this layer's input gradient) is the gradient of the pooled map pd describes: same images, same spatial size, NHWC
  SRLZ_REQUIRE(pd->n == d->n && pd->hp == d->hi && pd->wp == d->wi && !pd->out_nchw &&
               (pd->groups > 1 ? pd->groups : 1) == (d->groups > 1 ? d->groups : 1), SRLZ_ERR_BAD_DESC,
               "conv64_bwd_data_pool_sums: the pooled map [%d,%d,%d] is not this layer's input [%d,%d,%d]", pd->n, pd->hp, pd->wp, d->n,
               d->hi, d->wi);
  ConvProg P;
  if (int rc = program_for(&P, d, 1)) return rc;
  const int G = d->groups > 1 ? d->groups : 1;
  PoolSum ps = {pooled, pool_bnp, pool_y, pool_argmax, (long long)(pd->n / G) * pd->h * pd->w * 64, pd->h, pd->w, pd->pool_pad};
  return launch_fwd(dy, wpack_bwd, nullptr, dx, bn_bwd_partial, P, as_stream(stream), SRLZ_NO_FUSE, &ps);
}

extern "C" size_t srlz_conv64_bwd_weight_workspace(const srlz_conv64_desc* d) {
  if (check_desc(d)) return 0;
  ConvProg P;
  if (program_for(&P, d, 0)) return 0;
  return (size_t)wgrad_grid(P) * (NTAPS * 4096 + 64) * sizeof(float);
}

extern "C" int srlz_conv64_bwd_weight(const float* x, const float* dy, float* dw_ref, float* dbias, const float* x_bnp,
                                      const srlz_bn_bwd_operand* dy_bn, void* ws, size_t ws_bytes,
                                      const srlz_conv64_desc* d, srlz_stream_t stream) {
  if (int rc = check_desc(d)) return rc;
  OpFuse gf;
  if (int rc = make_bwd_fuse(&gf, dy_bn, "conv64_bwd_weight")) return rc;
  SRLZ_REQUIRE(gf.dy_out == nullptr, SRLZ_ERR_BAD_DESC, "conv64_bwd_weight: dy_out is only produced by srlz_conv64_bwd_data");
  const OpFuse xf = OpFuse{x_bnp, nullptr, nullptr, 0.f, 0, nullptr};
  SRLZ_REQUIRE(x && dy && dw_ref && ws, SRLZ_ERR_NULL, "conv64_bwd_weight: null pointer");
  ConvProg P;
  if (int rc = program_for(&P, d, 0)) return rc;
  const int grid = wgrad_grid(P);
  SRLZ_REQUIRE(ws_bytes >= (size_t)grid * (NTAPS * 4096 + 64) * sizeof(float), SRLZ_ERR_WORKSPACE,
               "conv64_bwd_weight: workspace too small (%zu bytes)", ws_bytes);
  const int tk = wgrad_tk(P);
  const int nchunks = (P.total_q + tk - 1) / tk;  // per BatchNorm group
  hipStream_t st = as_stream(stream);
  float* partial = (float*)ws;
  bool single_src = true;
  for (int t = 1; t < NTAPS; ++t) single_src = single_src && P.tsrc[t] == P.tsrc[0];
  int launched_grid = grid;
  // The chain below is a chain of SHAPE fallbacks (no environment switches): ConvTranspose programs whose span fits the 184-row ring
  // (PW <= 119) -> conv64_wgrad_ring_s2_kernel; stride-1 programs whose span fits the 246-row ring (PW <= 58) with 32-bit offsets ->
  // conv64_wgrad_ring_kernel; stride-2 gather programs (conv3) -> conv64_wgrad_gather_kernel; anything else (wider images, operands
  // rebuilt from (dA, y)) -> the chunk-at-a-time conv64_wgrad_kernel.
  if (single_src && gf.y == nullptr && P.s2 && 32 + P.span + 32 <= RING_ROWS_S2) {
    // 32-position chunks carrying all four destination classes (conv64_wgrad_ring_s2_kernel)
    const int nch = (P.total_q + 31) / 32;
    const int gpg = grid / P.G;
    const int cpw = (nch + gpg - 1) / gpg;
    const int wpg = (nch + cpw - 1) / cpw;
    launched_grid = wpg * P.G;
    const size_t lds = (size_t)(RING_S2 + 4 * 32) * 256;
    SRLZ_MAX_LDS(conv64_wgrad_ring_s2_kernel, lds);
    hipLaunchKernelGGL(conv64_wgrad_ring_s2_kernel, dim3(launched_grid), dim3(256), lds, st, x, dy, partial, P, nch, cpw, wpg, x_bnp);
  } else if (single_src && gf.y == nullptr && !P.s2 && tk + P.span + tk <= RING_ROWS && tk == 64 &&
             (long long)P.N * P.Hs * P.Ws * 64 < (1LL << 32) && (long long)P.N * P.Hd * P.Wd * 64 < (1LL << 32) &&
             (long long)P.total_q + 2 * P.PHW < (1LL << 31)) {  // (the row tables of the stride-1 kernel: 32-bit offsets)
    // contiguous chunk ranges per workgroup (ring re-use of the source rows), group by group
    const int gpg = grid / P.G;
    const int cpw = (nchunks + gpg - 1) / gpg;
    const int wpg = (nchunks + cpw - 1) / cpw;
    launched_grid = wpg * P.G;
    const size_t lds = (size_t)(RING + 64) * 256 + 2 * 64 * 4;  // ring + gradient rows + the two row tables = 80 KB
    SRLZ_MAX_LDS(conv64_wgrad_ring_kernel, lds);
    hipLaunchKernelGGL(conv64_wgrad_ring_kernel, dim3(launched_grid), dim3(256), lds, st, x, dy, partial, P, nchunks, cpw, wpg, x_bnp);
  } else if (wgrad_gather_ok(P) && x_bnp == nullptr && gf.y == nullptr && tk == WG_TK) {
    // stride-2 gather programs (conv3): the software-pipelined kernel; same grid, same partials as conv64_wgrad_kernel<true, 64>
    const size_t lds = (size_t)(WG_TK + P.span + WG_TK) * 256 + (WG_SWORDS + WG_GWORDS) * 4;
    SRLZ_MAX_LDS(conv64_wgrad_gather_kernel, lds);
    hipLaunchKernelGGL(conv64_wgrad_gather_kernel, dim3(grid), dim3(256), lds, st, x, dy, partial, P, nchunks * P.G);
  } else {
    const size_t lds = wgrad_lds_bytes(P, tk);
    SRLZ_REQUIRE(lds <= 160 * 1024, SRLZ_ERR_BAD_DESC, "conv64 wgrad: chunk needs %zu bytes of LDS", lds);
#define SRLZ_WGRAD_LAUNCH(S2V, TKV)                                                                                        \
  do {                                                                                                                     \
    SRLZ_MAX_LDS((conv64_wgrad_kernel<S2V, TKV>), lds);                                                                     \
    hipLaunchKernelGGL((conv64_wgrad_kernel<S2V, TKV>), dim3(grid), dim3(256), lds, st, x, dy, partial, P, nchunks * P.G, xf, gf); \
  } while (0)
    if (P.s2) SRLZ_WGRAD_LAUNCH(true, 64);
    else SRLZ_WGRAD_LAUNCH(false, 64);
#undef SRLZ_WGRAD_LAUNCH
  }
  SRLZ_LAUNCHED();
  hipLaunchKernelGGL(conv64_wgrad_reduce, dim3((NTAPS * 4096 + 64 + 255) / 256), dim3(1024), 0, st, partial, launched_grid,
                     dw_ref, dbias, d->transposed, 0);
  SRLZ_LAUNCHED();
  return 0;
}

// ---- the fused backward of a decoder block's ConvTranspose (conv64_bwd_fused_kernel) ----
static int fused_bwd_grid(const ConvProg& P) {
  int g = srlz_device_cus() & ~7;  // ONE workgroup per CU (150 KB of LDS, 256 registers per lane); a multiple of 8 for the XCD walk
  const int ntiles = P.G * P.tpg;
  // fewer tiles than CUs: rounded UP to the multiple of 8 (a workgroup without a tile leaves a zero partial) — rounded down, 98 tiles
  // (ConvT1's backward at bs = 32) ran on 96 workgroups, two of which took a second tile: 99 us for 50 us of work
  if (g > ntiles) g = (ntiles + 7) & ~7;
  return g;
}

static bool fused_bwd_ok(const ConvProg& P) {
  bool grouped = P.s2 && P.ss == 2 && P.G <= 2 && P.min_off == 0 && !P.dbg;
  for (int t = 0; t < NTAPS; ++t) grouped = grouped && P.tsrc[t] == P.tsrc[t < 4 ? 0 : t < 6 ? 4 : t < 8 ? 6 : 8] && P.tdst[t] == 0;
  const bool fits32 = P.src_gstride * 4 < (1LL << 32) - 65536 && P.dst_gstride * 4 < (1LL << 32) - 65536;
  // the second and fourth class are requested and landed for the rows their taps can reach only (FB_NJ1 / FB_NJ3); they share the
  // short LDS buffer As1 (TM + FB_REACH1 rows)
  const bool reach = P.toff[4] >= 0 && P.toff[5] >= 0 && P.toff[4] < FB_REACH1 && P.toff[5] < FB_REACH1 && P.toff[8] >= 0 &&
                     P.toff[8] <= FB_REACH3;
  return grouped && fits32 && reach && TM + P.span <= GP_RP * GP_BATCH && GP_RP <= P.PHW && P.G * P.tpg >= 8;
}

extern "C" int srlz_conv64_gather_pipe_supported(const srlz_conv64_desc* d, int backward_data) {
  if (check_desc(d)) return 0;
  ConvProg P;
  if (program_for(&P, d, backward_data)) return 0;
  return gather_pipe_ok(P) ? 1 : 0;
}

extern "C" int srlz_conv64_bwd_fused_supported(const srlz_conv64_desc* d) {
  if (check_desc(d) || !d->transposed || d->stride != 2) return 0;
  ConvProg P;
  if (program_for(&P, d, 1)) return 0;
  return fused_bwd_ok(P) ? 1 : 0;
}

extern "C" int srlz_conv64_bwd_fused_bn_rows(const srlz_conv64_desc* d) {
  if (check_desc(d)) return -1;
  ConvProg P;
  if (program_for(&P, d, 1)) return -1;
  return P.G * (4 * P.tpg + BNZ_BLOCKS);
}

extern "C" size_t srlz_conv64_bwd_fused_workspace(const srlz_conv64_desc* d) {
  if (check_desc(d)) return 0;
  ConvProg P;
  if (program_for(&P, d, 1)) return 0;
  return (size_t)fused_bwd_grid(P) * (NTAPS * 4096 + 64) * sizeof(float);
}

extern "C" int srlz_conv64_bwd_fused(const float* x, const float* x_bnp, const float* dy, const srlz_bn_bwd_operand* dy_bn,
                                     const float* wpack_bwd, float* dx, float* dw_ref, float* dbias, float* x_bn_bwd_partial,
                                     void* ws, size_t ws_bytes, const srlz_conv64_desc* d, srlz_stream_t stream) {
  if (int rc = check_desc(d)) return rc;
  SRLZ_REQUIRE(d->transposed && d->stride == 2, SRLZ_ERR_BAD_DESC, "conv64_bwd_fused: ConvTranspose2d(64, 64, 3, stride 2) only");
  SRLZ_REQUIRE(x && x_bnp && dy && dy_bn && wpack_bwd && dx && dw_ref && ws, SRLZ_ERR_NULL, "conv64_bwd_fused: null pointer");
  OpFuse gf;
  if (int rc = make_bwd_fuse(&gf, dy_bn, "conv64_bwd_fused")) return rc;
  SRLZ_REQUIRE(gf.dy_out == nullptr, SRLZ_ERR_BAD_DESC, "conv64_bwd_fused: d(loss)/dy is not materialised by this entry point");
  ConvProg P;
  if (int rc = program_for(&P, d, 1)) return rc;
  SRLZ_REQUIRE(fused_bwd_ok(P), SRLZ_ERR_BAD_DESC, "conv64_bwd_fused: shape not supported (ask srlz_conv64_bwd_fused_supported)");
  const int grid = fused_bwd_grid(P);
  SRLZ_REQUIRE(ws_bytes >= (size_t)grid * (NTAPS * 4096 + 64) * sizeof(float), SRLZ_ERR_WORKSPACE,
               "conv64_bwd_fused: workspace too small (%zu bytes)", ws_bytes);
  hipStream_t st = as_stream(stream);
  // class rows (TM + span; TM + 32 for the short-reach classes), the a-tile, two weight slabs, rowinfo, records, row tables
  const size_t lds = (size_t)(TM + P.span) * 256 + (size_t)(TM + FB_REACH1) * 256 + (size_t)TM * 256 + 2 * 16384 + 6 * TM * 4 + 512 * 4 +
                     256 * 4 + (GT_WORDS + YT_WORDS) * 4 + 2 * 192 * 4;
  SRLZ_REQUIRE(lds <= 160 * 1024, SRLZ_ERR_BAD_DESC, "conv64_bwd_fused: tile needs %zu bytes of LDS", lds);
  FusedBwd fb;
  fb.x = x; fb.x_bnp = x_bnp; fb.wpartial = (float*)ws;
  fb.bnpart = x_bn_bwd_partial; fb.bn_rows = 4 * P.tpg + BNZ_BLOCKS;
  SRLZ_MAX_LDS(conv64_bwd_fused_kernel, lds);
  hipLaunchKernelGGL(conv64_bwd_fused_kernel, dim3(grid), dim3(GP_THREADS), lds, st, dy, wpack_bwd, dx, P, P.G * P.tpg, gf, fb);
  SRLZ_LAUNCHED();
  // second stage: fixed-order fp64 sum over the workgroups (the partial's bias block sits behind EACH workgroup's taps here)
  hipLaunchKernelGGL(conv64_wgrad_reduce, dim3((NTAPS * 4096 + 64 + 255) / 256), dim3(1024), 0, st, (const float*)ws, grid, dw_ref, dbias,
                     1, NTAPS * 4096 + 64);
  SRLZ_LAUNCHED();
  if (x_bn_bwd_partial) {  // the records of the channels the fused kernel cannot sum from the activation (normally: zeros)
    hipLaunchKernelGGL(conv64_bnpart_zero_scale_kernel, dim3(BNZ_BLOCKS, P.G), dim3(256), 0, st, x, x_bnp, (const float*)dx,
                       x_bn_bwd_partial, (long long)P.N * P.Hd * P.Wd, fb.bn_rows, 4 * P.tpg);
    SRLZ_LAUNCHED();
  }
  return 0;
}

// Debug/test hook (host only, no GPU needed): dump the grid program so tests can interpret it on the CPU.
// out[0..]: N,PH,PW,ss,Hs,Ws,ds,Hd,Wd,min_off,span,s2, then 9 x {src,dst,off,w}.  Returns number of ints or <0.
extern "C" int srlz_conv64_debug_program(const srlz_conv64_desc* d, int backward_data, int* out, int cap) {
  if (int rc = check_desc(d)) return rc;
  ConvProg P;
  if (int rc = program_for(&P, d, backward_data)) return rc;
  if (cap < 12 + 4 * NTAPS) return SRLZ_ERR_WORKSPACE;
  int i = 0;
  out[i++] = P.N; out[i++] = P.PH; out[i++] = P.PW; out[i++] = P.ss; out[i++] = P.Hs; out[i++] = P.Ws;
  out[i++] = P.ds; out[i++] = P.Hd; out[i++] = P.Wd; out[i++] = P.min_off; out[i++] = P.span; out[i++] = P.s2;
  for (int t = 0; t < NTAPS; ++t) { out[i++] = P.tsrc[t]; out[i++] = P.tdst[t]; out[i++] = P.toff[t]; out[i++] = P.tw[t]; }
  return i;
}
