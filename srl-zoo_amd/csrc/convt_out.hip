// convt_out.hip — the decoder's last layer, nn.ConvTranspose2d(64, C, 4, stride=2) (/root/reference/models/models.py:82): forward
// (+ the step's reconstruction / generation loss, /root/reference/losses/losses.py:172-214) and its whole backward, both
// OUTPUT-STATIONARY and WAVE-PRIVATE (round 4; the round-1..3 kernels — a pixel-GEMM into 48 products per pixel followed by a col2im
// gather through LDS, and 8x16-position tiles for the backward — spent 5-7 VALU instructions per MFMA on gathers, transposes and
// window bookkeeping and were the only kernels of the step with LDS bank conflicts).
//
// Geometry.  out[co, 2a+py, 2b+px] = bias[co] + sum_{dy,dx in {0,1}} sum_ci  in[a-dy, b-dx, ci] * W[ci, co, py+2dy, px+2dx]
// for block positions (a, b) in [0, HF] x [0, WF]: every one of the 16 taps is used exactly once per 2x2 output block, nothing is
// scattered and nothing gathered — the accumulators ARE output pixels.
//
// Work unit = one WAVE walking down a column strip: 16 block positions wide (one MFMA N-tile), `R` rows high.  A wave keeps what
// it needs of the rows above in its own LDS (forward: the previous input row; backward: an 8-row ring of the error image) and
// never synchronises with another wave: no __syncthreads in the main loops, the compiler's lgkmcnt waits order a wave's own LDS
// traffic.  Per row the loads of the NEXT row are issued before the MFMAs of this one.
//
// MFMA: v_mfma_f32_16x16x4_f32 (16 x 16 outputs, 4 k per instruction; lane l supplies A[i = l & 15][k = l >> 4] and
// B[k = l >> 4][j = l & 15], and receives D[i = 4 (l >> 4) + r][j = l & 15] in register r).
//   forward   D[comp][pos]     = sum_k W[comp][k] * in[k][pos]      comp = co*4 + py*2 + px (12 of 16 rows used), k = (dy,dx,ci): 256
//   data grad D[ci][pos]       = sum_k W[ci][k] * err[k][pos]       k = (co,ky,kx): 48 per channel group, exact
//   wgt grad  D[ci][(ky,kx)]   = sum_pos act[ci][pos] * err[pos][co,ky,kx]   per co, exact
// so a lane's four accumulator registers are: forward — the 2x2 output block of ONE channel co = lane >> 4 at position lane & 15
// (two 8-byte NCHW stores, the loss taken in registers); data gradient — four consecutive channels of one position (16-byte NHWC
// stores, the same layout the BatchNorm-backward sums read y in).
#include "common.h"

namespace {

__device__ __forceinline__ int xcd_wg(int b, int nb) { return xcd_remap(b, nb); }

// Branch-free conditional stores (as conv64.hip's gp_buffer): a buffer resource over one image; a lane that must not write passes
// OS_DROP and the hardware drops its store.  A store inside a branch makes hipcc's wait insertion give up at the next loop header
// (everything is waited for: vmcnt is in-order and the path through the branch has an unknown number of stores in flight) — which
// would empty the rows-in-flight queue of these kernels once per trip.  A NULL tensor becomes a zero-sized resource: all dropped.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t os_buffer(float* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(p, 0, (int)(p ? bytes : 0u), 0x00020000);
}
constexpr unsigned OS_DROP = 0xFFFFFF00u;
__device__ __forceinline__ void os_store2(__amdgpu_buffer_rsrc_t r, unsigned byte_off, float a, float b) {
  __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(a), __float_as_uint(b)}, r, byte_off, 0, 0);
}

constexpr int OSP = 68;            // LDS pitch (floats) of one staged 64-channel pixel in the BACKWARD kernel's activation tile
// Forward: a staged input row (16 positions + the left halo pixel + one dummy pixel for the branch-free tail of the landing) lives in
// TWO PLANES of 32 channels — plane 0: channel blocks kq = 0, 2 (channels 0-15, 32-47), plane 1: kq = 1, 3 — with a pixel pitch of
// 36 floats (9 sixteen-byte slots, odd) and the planes 704 floats (a multiple of the 256-byte bank row) apart.  Why: ds_read_b128 is
// serviced in four NON-contiguous groups of 16 lanes ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...: MI355X_MICROARCH.md, LDS) —
// each group pairs positions {0-3, 12-15} of one k-quarter kq with positions {4-11} of kq ^ 1.  With one plane of 68-float pixels
// (rounds 4) the two halves of a group collided on four of the sixteen bank quads (SQ_LDS_BANK_CONFLICT: 8 % of the kernel's CU
// cycles); here a group's sixteen lanes read slots 9 p + const for sixteen different p: all sixteen bank quads, once.  The landing
// writes stay conflict-free because lanes 0-7 / 8-15 of a pixel's sixteen loaders take the channels of plane 0 / plane 1
// (ds_write_b128: contiguous 8-lane groups, 32 banks).
constexpr int FPP = 36;            // forward: pixel pitch inside a plane (floats)
constexpr int FPLANE = 704;        // forward: distance of plane 1 from plane 0 (floats; >= 18 * FPP, = 0 mod 64)
constexpr int OSROW = 1408;        // forward: one row slot = two planes (floats; = 0 mod 64)

// ------------------------------------------------------------------------------------------------------------------
// forward.  NCG = C / 3 channel groups share one staging of the input row: NCG == 1 keeps the 64 A fragments (weights) of a lane in
// registers, NCG > 1 reads them per group from LDS (16 x ds_read_b128 per group and row).
// LOSS: target != NULL; img receives dec - target, dec_out (optional) the reconstruction; loss_partial[2][workgroups] (fp64).
// ------------------------------------------------------------------------------------------------------------------
#ifndef SRLZ_OS_DEPTH
#define SRLZ_OS_DEPTH 2
#endif
constexpr int DEPTH = SRLZ_OS_DEPTH;  // rows in flight per wave of the forward kernel

template <int NCG, bool LOSS, typename TT, bool DEC = false>
__global__ __launch_bounds__(256, 2) void convT_out_os_kernel(const float* __restrict__ feat, const float* __restrict__ w_ref,
                                                             const float* __restrict__ bias, float* __restrict__ img, int N, int H,
                                                             int W, int HF, int WF, const float* __restrict__ feat_bnp, int npg,
                                                             const TT* __restrict__ target, float* __restrict__ dec_out,
                                                             double* __restrict__ loss_partial, int lpg,
                                                             const float* __restrict__ lut, int R, int nseg, int nchunk) {
  constexpr bool U8 = sizeof(TT) == 1;
  constexpr int C = 3 * NCG;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* S = (float*)smem + wave * (2 * OSROW);         // this wave's two row slots
  float* Wl = (float*)smem + 4 * 2 * OSROW;             // NCG > 1: A fragments [cg][t][jj][lane][4]
  float* L = Wl + (NCG > 1 ? NCG * 4 * 4 * 64 * 4 : 0);  // U8: the normalisation table
  const int p = lane & 15, kq = lane >> 4;              // compute roles: position / k-quarter (forward epilogue: kq = co)
  const int pq = lane >> 4;                             // staging roles: pixel pq + 4 i, and four channels of it:
  // loader u = lane & 15 takes channels c4 * 4 .. + 3 with c4 = 4 kq + (u & 3), kq = 2 ((u >> 2) & 1) + (u >> 3): loaders 0-7 the
  // channel blocks of plane 0 (kq = 0, 2), loaders 8-15 those of plane 1 (kq = 1, 3); the sixteen still cover the pixel's 256 bytes
  const int lu = lane & 15;
  const int c4 = 4 * (2 * ((lu >> 2) & 1) + (lu >> 3)) + (lu & 3);
  const int lofs = (lu >> 3) * FPLANE + 4 * (lu & 7);   // where this loader's four channels sit inside a staged pixel

  // ---- A fragments: lane (m = comp, kq) holds W[ci = 16 kq + j][co][py + 2 dy][px + 2 dx] for t = (dy, dx), j = 0..15
  const int co_m = p >> 2, py_m = (p >> 1) & 1, px_m = p & 1;
  float wr[NCG == 1 ? 4 : 1][16];
  if constexpr (NCG == 1) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int j = 0; j < 16; ++j)
        wr[t][j] = w_ref[((size_t)(16 * kq + j) * C + (p < 12 ? co_m : 0)) * 16 + (py_m + 2 * (t >> 1)) * 4 + (px_m + 2 * (t & 1))];
    if (p >= 12) {  // rows 12..15 of the 16 x 16 output tile do not exist
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) wr[t][j] = 0.f;
    }
  } else {
    for (int idx = tid; idx < NCG * 4 * 4 * 64 * 4; idx += 256) {
      const int e = idx & 3, ln = (idx >> 2) & 63, jj = (idx >> 8) & 3, t = (idx >> 10) & 3, cg = idx >> 12;
      const int m = ln & 15, q = ln >> 4;
      Wl[idx] = m < 12 ? w_ref[((size_t)(16 * q + 4 * jj + e) * C + cg * 3 + (m >> 2)) * 16 + (((m >> 1) & 1) + 2 * (t >> 1)) * 4 +
                               ((m & 1) + 2 * (t & 1))]
                       : 0.f;
    }
  }
  if constexpr (U8) {
    for (int i = tid; i < 768; i += 256) L[i] = lut[i];  // (per channel of a group of 3, as image_land<U8>)
  }
  if constexpr (NCG > 1 || U8) __syncthreads();
  float bs[NCG];
#pragma unroll
  for (int cg = 0; cg < NCG; ++cg) bs[cg] = (bias && kq < 3) ? bias[cg * 3 + kq] : 0.f;

  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};  // (no record: z = 1 * y + 0 = y exactly)
  const float act_lo = feat_bnp ? 0.f : -__builtin_inff();
  int cur_grp = -1;
  double lacc0 = 0.0, lacc1 = 0.0;
  const int njobs = N * nchunk * nseg;
  const int wg = xcd_wg(blockIdx.x, gridDim.x);

  for (int job = wg * 4 + wave; job < njobs; job += gridDim.x * 4) {
    const int seg = job % nseg, jt = job / nseg;
    const int chunk = jt % nchunk, n = jt / nchunk;
    const int a0 = chunk * R, a1 = min(a0 + R, HF + 1), b0 = seg * 16;
    if (feat_bnp && n / npg != cur_grp) {
      cur_grp = n / npg;
      const float* __restrict__ rec = feat_bnp + cur_grp * 256;
      sc = *(const f32x4*)(rec + 128 + 4 * c4);
      sh = *(const f32x4*)(rec + 192 + 4 * c4);
    }
    const float* __restrict__ fimg = feat + (size_t)n * HF * WF * 64 + 4 * c4;
    const __amdgpu_buffer_rsrc_t img_rs = os_buffer(img + (size_t)n * C * H * W, (unsigned)(C * H * W * 4));
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t dec_rs = os_buffer(dec_out ? dec_out + (size_t)n * C * H * W : nullptr,
                                                                      (unsigned)(C * H * W * 4));
    // Rows travel through a queue of DEPTH register sets: row r is requested DEPTH row-steps before it lands.  With ONE row in
    // flight per wave a step waited ~1 us for HBM (563 us per launch at N = 512); with two the kernel is bound by instruction
    // issue instead (DEPTH 2 and 3 measure the same, 567 / 574 us).  The queue slot of row r is
    // (r - (a0 - 1)) % DEPTH; the row loop is unrolled by DEPTH so that every slot index is static.
    f32x4 ld[DEPTH][5];
    unsigned ldok[DEPTH];
    // row r of the input strip (pixels b0-1 .. b0+15) -> registers; a pixel outside the map reads a clamped address and lands as 0
    auto request = [&](int r, f32x4 (&q)[5], unsigned& qok) {
      const bool rowok = (unsigned)r < (unsigned)HF;
      qok = 0;
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const int fx = b0 - 1 + pq + 4 * i;
        const bool ok = rowok && (i < 4 || pq == 0) && (unsigned)fx < (unsigned)WF;
        const unsigned off = ok ? (unsigned)((r * WF + fx) * 64) : 0u;
        q[i] = *(const f32x4*)(fimg + off);
        qok |= (ok ? 1u : 0u) << i;
      }
    };
    // landing: relu(bn(.)) and the zero of a pixel outside the map are ONE v_med3 per element — med3(z, lo, hi) with (lo, hi) =
    // (0, +inf) for a live pixel of a BatchNorm+ReLU operand, (-inf, +inf) when there is no BatchNorm record, (0, 0) outside
    auto land = [&](int r, const f32x4 (&q)[5], unsigned qok) {
      float* dst = S + (r & 1) * OSROW + lofs;
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const bool ok = (qok >> i) & 1u;
        const float lo = ok ? act_lo : 0.f, hi = ok ? __builtin_inff() : 0.f;
        f32x4 v = q[i] * sc + sh;  // (a vector expression: two v_pk_fma_f32; each element one IEEE fma as before)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], lo, hi);
        const int pl = (i < 4 || pq == 0) ? pq + 4 * i : 17;  // (pixel 17: the dummy the idle lanes of the fifth load write)
        *(f32x4*)(dst + pl * FPP) = v;
      }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) request(a0 - 1 + d, ld[d], ldok[d]);
    land(a0 - 1, ld[0], ldok[0]);
    request(a0 - 1 + DEPTH, ld[0], ldok[0]);
    for (int ab = a0; ab < a1; ab += DEPTH) {
#pragma unroll
     for (int dd = 0; dd < DEPTH; ++dd) {
      // (no exit in the middle of the unrolled body: with one, hipcc's wait insertion drains the whole queue at the loop header.
      // A row past the strip's end — at most DEPTH - 1 per strip — is computed and dropped: its stores carry OS_DROP.)
      const int a = ab + dd;
      const bool arow = a < a1;
      f32x4 (&slot)[5] = ld[(dd + 1) % DEPTH];
      unsigned& slot_ok = ldok[(dd + 1) % DEPTH];
      land(a, slot, slot_ok);
      // ---- targets of this row's 2x2 blocks (LOSS): requested BEFORE the next row — the memory counter is in-order, so waiting for
      // them in the epilogue must not also wait for the row that is meant to stay in flight until the next iteration
      const bool live = arow && kq < 3 && b0 + p <= WF;
      // (raw loads only — a byte pair is unpacked in the epilogue: an instruction that consumes the loaded value HERE would wait for
      // it here, and, the counter being in-order, would not need to wait for more, but would sit in front of the row request)
      [[maybe_unused]] float tg[NCG][4];
      [[maybe_unused]] unsigned tgraw[NCG][2];
      if constexpr (LOSS) {
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg)
#pragma unroll
          for (int py = 0; py < 2; ++py) {
            const size_t o = live ? ((size_t)(n * C + cg * 3 + kq) * H + 2 * a + py) * W + 2 * (b0 + p) : (size_t)0;
            if constexpr (U8) {
              // the aligned dword that holds the two bytes (o is even; a 16-bit load would be zero-extended by an instruction that
              // consumes it on the spot); which half is decided in the epilogue
              tgraw[cg][py] = *(const unsigned*)((const uint8_t*)target + (o & ~(size_t)3));
            } else {
              const float2 raw = *(const float2*)((const float*)target + o);
              tg[cg][2 * py] = raw.x;
              tg[cg][2 * py + 1] = raw.y;
            }
          }
        __builtin_amdgcn_sched_barrier(0);
      }
      request(a + DEPTH, slot, slot_ok);  // (past the strip's last row: clamped addresses, never landed)
      __builtin_amdgcn_sched_barrier(0);  // (the scheduler would sink the requests below the MFMAs, next to their first use)
      float lsum = 0.f;
#pragma unroll
      for (int cg = 0; cg < NCG; ++cg) {
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        // B fragments: the 4 x 16 bytes of shift t + 1 are read while the 16 MFMAs of shift t run
        f32x4 bb[2][4];
        auto read_b = [&](int t, f32x4 (&b)[4]) {
          const float* bp = S + ((a - (t >> 1)) & 1) * OSROW + (kq & 1) * FPLANE + (p + 1 - (t & 1)) * FPP + 16 * (kq >> 1);
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) b[jj] = *(const f32x4*)(bp + 4 * jj);
        };
        read_b(0, bb[0]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (t < 3) read_b(t + 1, bb[(t + 1) & 1]);
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const f32x4 b = bb[t & 1][jj];
            f32x4 wv;
            if constexpr (NCG == 1) wv = f32x4{wr[t][4 * jj], wr[t][4 * jj + 1], wr[t][4 * jj + 2], wr[t][4 * jj + 3]};
            else wv = *(const f32x4*)(Wl + (((cg * 4 + t) * 4 + jj) * 64 + lane) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (t < 2) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[e], b[e], acc0, 0, 0, 0);
              else acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[e], b[e], acc1, 0, 0, 0);
            }
          }
        }
        // (pin the order the source states: hipcc's scheduler otherwise pairs every two reads with the eight MFMAs that use them)
        if constexpr (NCG == 1) {
          __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
          for (int t = 0; t < 3; ++t) {
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
            if (t < 2) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
        }
        // ---- epilogue: this lane holds out[co = kq][2a + py][2(b0 + p) + px] in register py*2 + px
        const f32x4 v = (acc0 + acc1) + f32x4{bs[cg], bs[cg], bs[cg], bs[cg]};  // (vector expressions: v_pk_add_f32)
        if constexpr (LOSS) {
          if constexpr (U8) { asm volatile("" : "+v"(tgraw[cg][0]), "+v"(tgraw[cg][1])); }
          else {
#pragma unroll
            for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(tg[cg][r]));
          }
          if constexpr (U8) {
#pragma unroll
            for (int py = 0; py < 2; ++py) {
              const float* Lc = L + (kq < 3 ? kq : 0) * 256;
              // which half of the dword: bit 1 of the byte offset o = row * W + 2 (b0 + p), row = (plane * H + 2a + py).  H and W are
              // even, so bit 1 of row * W is (row & 1) & (W / 2 & 1) = py & (W / 2 & 1); no carry reaches it from bit 0
              const unsigned two = tgraw[cg][py] >> ((((py & (W >> 1)) ^ (b0 + p)) & 1) * 16);
              tg[cg][2 * py] = Lc[two & 0xffu];
              tg[cg][2 * py + 1] = Lc[(two >> 8) & 0xffu];
            }
          }
        }
        // (offsets inside image n: 32-bit; a dead lane's stores are dropped by the buffer hardware)
        const unsigned ob = live ? (unsigned)((((cg * 3 + kq) * H + 2 * a) * W + 2 * (b0 + p)) * 4) : OS_DROP;
        const unsigned ob2 = live ? ob + (unsigned)(W * 4) : OS_DROP;
        if constexpr (LOSS) {
          const f32x4 d = v - f32x4{tg[cg][0], tg[cg][1], tg[cg][2], tg[cg][3]};
          // (explicit fmas in a fixed order: every instantiation rounds the partial sums alike; a dead lane adds nothing)
          const float sq = __builtin_fmaf(d[3], d[3], __builtin_fmaf(d[2], d[2], __builtin_fmaf(d[1], d[1], d[0] * d[0])));
          lsum += live ? sq : 0.f;
          os_store2(img_rs, ob, d[0], d[1]);
          os_store2(img_rs, ob2, d[2], d[3]);
          if constexpr (DEC) {  // (the reconstruction itself: not on the training path)
            os_store2(dec_rs, ob, v[0], v[1]);
            os_store2(dec_rs, ob2, v[2], v[3]);
          }
        } else {
          os_store2(img_rs, ob, v[0], v[1]);
          os_store2(img_rs, ob2, v[2], v[3]);
        }
      }
      if constexpr (LOSS) { if (n / lpg == 0) lacc0 += (double)lsum; else lacc1 += (double)lsum; }
     }
    }
  }
  if constexpr (LOSS) {
    // [2 loss groups][workgroups of the launch] — every workgroup writes both slots (zeros included), fixed-order final sum
    __syncthreads();
    double* lred = (double*)smem;  // [2][4 waves]
    const double w0 = wave_sum_d(lacc0), w1 = wave_sum_d(lacc1);
    if (lane == 0) { lred[wave] = w0; lred[4 + wave] = w1; }
    __syncthreads();
    if (tid < 2) loss_partial[(size_t)tid * gridDim.x + blockIdx.x] = (lred[tid * 4] + lred[tid * 4 + 1]) + (lred[tid * 4 + 2] + lred[tid * 4 + 3]);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// backward: data gradient dA = d(loss)/d(relu(bn(y))), the two BatchNorm-backward sums of every strip (one record per job),
// weight gradient and bias gradient from ONE read of err (= dy, or the stored reconstruction error times `gain`) and ONE read of y.
//   err ring: [3 NCG channels][8 rows][RP] per wave — rows 2a .. 2a+3 serve position row a; two new rows per step.
//   F:        relu(bn(y)) of the strip's current row, [16 positions][OSP] — the weight gradient's A operand.
// ------------------------------------------------------------------------------------------------------------------
constexpr int RP = 40;  // ring row pitch (34 columns used)

template <int NCG>
// (err and y_raw are deliberately NOT __restrict__: hipcc treats loads through a `const __restrict__` kernel argument as invariant —
// free of every ordering constraint — and sinks the requests of the software pipeline below the MFMAs they are meant to travel
// under; as ordinary loads they stay in front of the memory clobber that follows them)
__global__ __launch_bounds__(256, 2) void convT_out_os_bwd_kernel(const float* err, const float* __restrict__ w_ref,
                                                                 float* __restrict__ dA, float* __restrict__ stats_partial,
                                                                 float* __restrict__ wpartial, int N, int H, int W, int HF, int WF,
                                                                 const float* y_raw, const float* __restrict__ y_bnp,
                                                                 int npg, double* __restrict__ bias_partial,
                                                                 const float* __restrict__ gain_dev, float gain_div, float gain_coef,
                                                                 int R, int nseg, int nchunk) {
  constexpr int C = 3 * NCG, RING = C * 8 * RP, FSZ = 16 * OSP, PERW = RING + FSZ + 192;
  constexpr int NE = C * 34;                  // float2 elements of a 2-row err request: (channel, row of the pair, column pair)
  constexpr int NSLOT = (NE + 63) / 64;
  const float gain = gain_dev ? (gain_dev[0] / gain_div) * gain_coef : 1.f;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* ring = (float*)smem + wave * PERW;
  float* F = ring + RING;
  float* Bn = F + FSZ;                        // the BatchNorm record of the strip's group: (unused), scale[64], shift[64]
  float* Wl = (float*)smem + 4 * PERW;        // NCG > 1: data-gradient A fragments [cg][mt][s][lane]
  const int p = lane & 15, kq = lane >> 4;
  const int ky_n = p >> 2, kx_n = p & 3;      // weight gradient: this lane's output column n = (ky, kx)

  // ---- data-gradient A fragments: lane (m, kq = kx) holds W[ci = 16 mt + m][co][ky][kx] for step s = co*4 + ky
  float wA[NCG == 1 ? 4 : 1][12];
  if constexpr (NCG == 1) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int s = 0; s < 12; ++s) wA[mt][s] = w_ref[((size_t)(16 * mt + p) * C + (s >> 2)) * 16 + (s & 3) * 4 + kq];
  } else {
    for (int idx = tid; idx < NCG * 4 * 12 * 64; idx += 256) {
      const int ln = idx & 63, s = (idx >> 6) % 12, mt = ((idx >> 6) / 12) & 3, cg = (idx >> 6) / 48;
      Wl[idx] = w_ref[((size_t)(16 * mt + (ln & 15)) * C + cg * 3 + (s >> 2)) * 16 + (s & 3) * 4 + (ln >> 4)];
    }
    __syncthreads();
  }
  f32x4 accw[NCG][4][3];
#pragma unroll
  for (int cg = 0; cg < NCG; ++cg)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int co = 0; co < 3; ++co) accw[cg][mt][co] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- err request slots: element e = lane + 64 k = (channel ch, row rs of the pair, column pair cp), packed in one register
  int e_pk[NSLOT];  // (ch << 8) | (rs << 5) | cp, or -1 past the end of the request
  float bsum[NSLOT];
#pragma unroll
  for (int k = 0; k < NSLOT; ++k) {
    const int e = lane + 64 * k;
    e_pk[k] = e < NE ? ((e / 34) << 8) | (((e % 34) / 17) << 5) | (e % 17) : -1;
    bsum[k] = 0.f;
  }
  int cur_grp = -1;
  const int njobs = N * nchunk * nseg;
  const int wg = xcd_wg(blockIdx.x, gridDim.x);

  for (int job = wg * 4 + wave; job < njobs; job += gridDim.x * 4) {
    const int seg = job % nseg, jt = job / nseg;
    const int chunk = jt % nchunk, n = jt / nchunk;
    const int a0 = chunk * R, a1 = min(a0 + R, HF), b0 = seg * 16;
    if (n / npg != cur_grp) {
      cur_grp = n / npg;
      const float* __restrict__ rec = y_bnp + cur_grp * 256;
      Bn[64 + lane] = rec[128 + lane];
      Bn[128 + lane] = rec[192 + lane];
    }
    const bool last_seg = seg == nseg - 1, last_chunk = a1 == HF;
    const float* eimg = err + (size_t)n * C * H * W;
    const float* yimg = y_raw + (size_t)n * HF * WF * 64 + 4 * kq;
    const __amdgpu_buffer_rsrc_t dA_rs = os_buffer(dA + (size_t)n * HF * WF * 64, (unsigned)(HF * WF * 256));
    // ---- per strip: where each request slot reads (element offset of row 0) and lands, which slots exist, which are owned
    // (H and W are even and requests start at even rows / columns, so "row inside the image" and "row owned" are wave-uniform)
    unsigned e_off[NSLOT];
    int r_off[NSLOT];
    unsigned livemask = 0, ownmask = 0;
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) {
      const int ch = e_pk[k] >> 8, rs = (e_pk[k] >> 5) & 1, cp = e_pk[k] & 31;
      const int xx = 2 * b0 + 2 * cp;
      const bool lv = e_pk[k] >= 0 && xx < W;
      e_off[k] = lv ? (unsigned)((ch * H + rs) * W + xx) : 0u;
      // (elements past the end of the request land in the spare columns 36, 37 of channel 0)
      r_off[k] = e_pk[k] >= 0 ? (ch * 8 + rs) * RP + 2 * cp : rs * RP + 36;
      livemask |= (lv ? 1u : 0u) << k;
      // bias gradient: a strip OWNS rows [2 a0, 2 a1) (the last chunk also the two tail rows) and column pairs 0..15 (the last
      // segment also pair 16) of what it stages, so every pixel of the error image is counted exactly once
      ownmask |= ((lv && (cp < 16 || last_seg)) ? 1u : 0u) << k;
    }
    float2 ev[NSLOT];
    unsigned evin = 0, evown = 0;
    // err rows y0, y0 + 1 (y0 even; columns 2 b0 .. 2 b0 + 33) -> registers
    auto req_err = [&](int y0, bool any) {
      const bool rows = any && y0 < H;
      evin = rows ? livemask : 0u;
      evown = (rows && (y0 < 2 * a1 || last_chunk)) ? ownmask : 0u;
      const unsigned rowoff = (unsigned)(y0 * W);
#pragma unroll
      for (int k = 0; k < NSLOT; ++k) ev[k] = *(const float2*)(eimg + (((evin >> k) & 1u) ? e_off[k] + rowoff : 0u));
    };
    auto land_err = [&](int y0) {
      float* dst = ring + (y0 & 7) * RP;
#pragma unroll
      for (int k = 0; k < NSLOT; ++k) {
        const bool in = (evin >> k) & 1u;
        const float2 g = float2{in ? ev[k].x * gain : 0.f, in ? ev[k].y * gain : 0.f};
        bsum[k] += ((evown >> k) & 1u) ? g.x + g.y : 0.f;
        *(float2*)(dst + r_off[k]) = g;
      }
    };
    const bool pvalid = b0 + p < WF;
    const float act_hi = pvalid ? __builtin_inff() : 0.f;
    const unsigned ypos = (unsigned)((b0 + p) * 64);
    f32x4 yv[4];
    auto req_y = [&](int a, bool any) {
      const unsigned off = (any && pvalid) ? (unsigned)(a * WF * 64) + ypos : 0u;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) yv[mt] = *(const f32x4*)(yimg + off + 16 * mt);
    };
    // Software pipeline of a strip (loop rotated by hand so that no loaded-but-not-yet-used register is live across the back edge —
    // hipcc copies such registers at the loop header, which waits for them there):
    //   iteration a:  request y(a) and the two err rows step a + 1 adds   |  weight gradient of step a - 1  |  data gradient of
    //                 step a  |  epilogue (consumes y)  |  land the err rows
    // so everything requested at the top of an iteration has the two MFMA phases to arrive (requested and landed around ONE
    // phase, the err rows had ~1 us of the ~2.5 us an HBM round trip takes at this load).  The ring holds 8 rows and a step uses
    // 4, so rows can land a step early.  The first iteration has no weight gradient, the last weight gradient follows the loop.
    f32x4 s1[4], s2[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) s1[mt] = s2[mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto wgrad = [&](int a) {
      // D[ci][(ky,kx)] per co, K = the 16 positions of the row (k-step s, quarter kq -> position 4 s + kq)
      const int rrow = ((2 * a + ky_n) & 7) * RP + 2 * kq + kx_n;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float af[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) af[mt] = F[(4 * s + kq) * OSP + 16 * mt + p];
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg)
#pragma unroll
          for (int co = 0; co < 3; ++co) {
            const float b = ring[(cg * 3 + co) * 8 * RP + rrow + 8 * s];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) accw[cg][mt][co] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt], b, accw[cg][mt][co], 0, 0, 0);
          }
      }
    };
    auto dgrad_epilogue = [&](int a) {
      // ---- data gradient: D[ci][pos], K = (cg, co, ky, kx)
      f32x4 acc[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int cg = 0; cg < NCG; ++cg)
#pragma unroll
        for (int s = 0; s < 12; ++s) {
          const int co = s >> 2, ky = s & 3;
          const float b = ring[((cg * 3 + co) * 8 + ((2 * a + ky) & 7)) * RP + 2 * p + kq];
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            float w;
            if constexpr (NCG == 1) w = wA[mt][s]; else w = Wl[((cg * 4 + mt) * 12 + s) * 64 + lane];
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, b, acc[mt], 0, 0, 0);
          }
        }
      // ---- epilogue: this lane holds dA[pos p][ci = 16 mt + 4 kq + e]; y of the same (pos, channels) is in yv
      // The wait for y belongs HERE, behind both MFMA phases.  The MFMAs are pure operations: nothing but a data dependence keeps
      // them in front of an `asm volatile`, so the statement that "touches" y (and makes hipcc wait for it once, outside what
      // follows) also names the accumulators of both phases.
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" : "+v"(yv[0]), "+v"(yv[1]), "+v"(yv[2]), "+v"(yv[3]), "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
#pragma unroll
      for (int cg = 0; cg < NCG; ++cg)
        asm volatile("" : "+v"(accw[cg][0][0]), "+v"(accw[cg][0][1]), "+v"(accw[cg][0][2]), "+v"(accw[cg][1][0]), "+v"(accw[cg][1][1]),
                     "+v"(accw[cg][1][2]), "+v"(accw[cg][2][0]), "+v"(accw[cg][2][1]), "+v"(accw[cg][2][2]), "+v"(accw[cg][3][0]),
                     "+v"(accw[cg][3][1]), "+v"(accw[cg][3][2]), "+v"(yv[0]));
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const f32x4 bsc = *(const f32x4*)(Bn + 64 + 16 * mt + 4 * kq), bsh = *(const f32x4*)(Bn + 128 + 16 * mt + 4 * kq);
        // (whole-vector expressions: hipcc turns them into v_pk_fma_f32 / v_pk_add_f32, two elements per issue slot — every
        // instruction next to the MFMAs costs ~4.7 cycles of matrix time; each element is still one IEEE fma, the
        // rounding of the forward kernel's relu(bn(y)))
        const f32x4 z4 = yv[mt] * bsc + bsh;
        f32x4 act, v4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          act[e] = __builtin_amdgcn_fmed3f(z4[e], 0.f, act_hi);  // relu, and 0 for a position outside the map
          v4[e] = (z4[e] > 0.f && pvalid) ? acc[mt][e] : 0.f;
        }
        s1[mt] += v4;
        s2[mt] = v4 * yv[mt] + s2[mt];  // (centred once per strip: sum dz (y - mean) = sum dz y - mean sum dz, in fp64)
        *(f32x4*)(F + p * OSP + 16 * mt + 4 * kq) = act;
      }
      const unsigned ob = ((unsigned)(a * WF * 64) + ypos + 4 * kq) * 4u;  // (branch-free, see os_buffer)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) __builtin_amdgcn_raw_buffer_store_b128(acc[mt], dA_rs, pvalid ? ob + 64u * mt : OS_DROP, 0, 0);
    };
    // requests of iteration a: the err rows step a + 1 adds (2a + 4, 2a + 5) and y(a); pinned in front of the MFMAs that follow
    // (hipcc sinks a load towards its first use; a memory clobber is a point no load may be moved across)
    auto requests = [&](int a) {
      req_err(2 * a + 4, a + 1 < a1);
      req_y(a, true);
      asm volatile("" ::: "memory");
    };
    req_err(2 * a0, true);
    land_err(2 * a0);
    req_err(2 * a0 + 2, true);
    land_err(2 * a0 + 2);
    requests(a0);
    dgrad_epilogue(a0);
    land_err(2 * a0 + 4);
    for (int a = a0 + 1; a < a1; ++a) {
      requests(a);
      wgrad(a - 1);
      dgrad_epilogue(a);
      land_err(2 * a + 4);  // (zeros past the strip's last step: evin = 0)
    }
    wgrad(a1 - 1);
    // ---- the strip's BatchNorm-backward record: [sum dz (64)] [sum dz * xhat (64)], summed over the 16 positions (lanes p)
    {
      const float* __restrict__ rec = y_bnp + cur_grp * 256;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const f32x4 inv = *(const f32x4*)(rec + 64 + 16 * mt + 4 * kq), mean = *(const f32x4*)(rec + 16 * mt + 4 * kq);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float u = s1[mt][e], v = s2[mt][e];
#pragma unroll
          for (int o = 1; o < 16; o <<= 1) { u += __shfl_xor(u, o, 64); v += __shfl_xor(v, o, 64); }
          s1[mt][e] = u;
          // sum dz * xhat = (sum dz y - mean sum dz) * invstd: the subtraction in fp64 (both sums are fp32 partials of <= 448 terms)
          s2[mt][e] = (float)(((double)v - (double)mean[e] * (double)u) * (double)inv[e]);
        }
      }
      if (p == 0) {
        float* out = stats_partial + (size_t)job * 128 + 4 * kq;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          *(f32x4*)(out + 16 * mt) = s1[mt];
          *(f32x4*)(out + 64 + 16 * mt) = s2[mt];
        }
      }
    }
  }

  // ---- weight-gradient partial of the workgroup: the four waves' accumulators summed through LDS -> wpartial[cg][wg][ci][48]
  __syncthreads();
  float* red = (float*)smem;  // [4 waves][64 ci][48]
#pragma unroll
  for (int cg = 0; cg < NCG; ++cg) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int co = 0; co < 3; ++co)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(wave * 64 + 16 * mt + 4 * kq + r) * 48 + co * 16 + p] = accw[cg][mt][co][r];
    __syncthreads();
    float* out = wpartial + ((size_t)cg * gridDim.x + blockIdx.x) * (64 * 48);
    for (int i = tid; i < 64 * 48; i += 256) out[i] = (red[i] + red[64 * 48 + i]) + (red[2 * 64 * 48 + i] + red[3 * 64 * 48 + i]);
    __syncthreads();
  }
  if (bias_partial) {  // [C][gridDim.x] fp64, summed over workgroups in a fixed order by chan_sum_final
    double* bred = (double*)smem;  // [C][4 waves]
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
      float cs = 0.f;
#pragma unroll
      for (int k = 0; k < NSLOT; ++k) cs += (e_pk[k] >= 0 && (e_pk[k] >> 8) == cc) ? bsum[k] : 0.f;
      const double dsum = wave_sum_d((double)cs);
      if (lane == 0) bred[cc * 4 + wave] = dsum;
    }
    __syncthreads();
    if (tid < C) bias_partial[(size_t)tid * gridDim.x + blockIdx.x] = (bred[tid * 4] + bred[tid * 4 + 1]) + (bred[tid * 4 + 2] + bred[tid * 4 + 3]);
  }
}


// ---- host side ---------------------------------------------------------------------------------------------------------
// Rows per strip.  Forward: a multiple of DEPTH (the row loop is unrolled by DEPTH; a ragged strip computes and drops up to DEPTH - 1
// rows) — 112 block rows = 4 x 28; backward: 111 = 28 + 28 + 28 + 27.  Measured at N = 512 (tools/kb_convt_out.py, strips of 8 /
// 16 / 28 / 56 rows): forward 569 / 563 / 545 / 584 us, backward 879 / 853 / 833 / 894 us — short strips pay their prologue (the
// halo row, the weight fragments) more often, long ones leave the persistent waves unevenly loaded.
constexpr int OS_FWD_ROWS = 28, OS_BWD_ROWS = 28;
static_assert(OS_FWD_ROWS % DEPTH == 0, "forward strips are walked DEPTH rows at a time");
int fwd_rows() { return OS_FWD_ROWS; }
int bwd_rows() { return OS_BWD_ROWS; }

int os_check(const srlz_skinny_desc* d, const char* who) {
  SRLZ_REQUIRE(d != nullptr, SRLZ_ERR_NULL, "%s: null descriptor", who);
  SRLZ_REQUIRE(d->kind == 1, SRLZ_ERR_BAD_DESC, "%s: descriptor kind must be 1", who);
  SRLZ_REQUIRE(d->n > 0 && d->c > 0 && d->c % 3 == 0 && d->c <= 9, SRLZ_ERR_BAD_DESC, "%s: C must be 3, 6 or 9 (got %d)", who, d->c);
  SRLZ_REQUIRE(d->himg == (d->hf - 1) * 2 + 4 && d->wimg == (d->wf - 1) * 2 + 4, SRLZ_ERR_BAD_DESC,
               "%s: image %dx%d inconsistent with feature map %dx%d", who, d->himg, d->wimg, d->hf, d->wf);
  SRLZ_REQUIRE(d->groups >= 0 && (d->groups <= 1 || d->n % d->groups == 0), SRLZ_ERR_BAD_DESC,
               "%s: n = %d is not a multiple of groups = %d", who, d->n, d->groups);
  // offsets inside one image are 32-bit in the kernels — as BYTES in the buffer stores (a buffer resource spans one image)
  SRLZ_REQUIRE((long long)d->hf * d->wf * 256 < (1LL << 31) && (long long)d->c * d->himg * d->wimg * 4 < (1LL << 31), SRLZ_ERR_BAD_DESC,
               "%s: one image of %dx%d exceeds the 32-bit per-image byte offsets", who, d->himg, d->wimg);
  return 0;
}

struct OsGeo { int nseg, nchunk, njobs, grid, rows; };
OsGeo fwd_geo(const srlz_skinny_desc* d) {
  OsGeo g;
  g.rows = fwd_rows();
  g.nseg = (d->wf + 1 + 15) / 16;
  g.nchunk = (d->hf + 1 + g.rows - 1) / g.rows;
  g.njobs = d->n * g.nchunk * g.nseg;
  const int want = (g.njobs + 3) / 4, cap = 2 * srlz_device_cus();
  g.grid = want < cap ? want : cap;
  return g;
}
OsGeo bwd_geo(const srlz_skinny_desc* d) {
  OsGeo g;
  g.rows = bwd_rows();
  g.nseg = (d->wf + 15) / 16;
  g.nchunk = (d->hf + g.rows - 1) / g.rows;
  g.njobs = d->n * g.nchunk * g.nseg;
  const int want = (g.njobs + 3) / 4, cap = 2 * srlz_device_cus();
  g.grid = want < cap ? want : cap;
  return g;
}
int os_npg(const srlz_skinny_desc* d) { return d->n / (d->groups > 1 ? d->groups : 1); }

template <int NCG, bool LOSS, typename TT>
int os_fwd_launch(const float* x, const float* w, const float* bias, float* out, const float* bnp, const TT* target, float* dec,
                  double* loss_partial, const float* lut, const srlz_skinny_desc* d, hipStream_t st) {
  const OsGeo g = fwd_geo(d);
  const size_t lds = (size_t)(4 * 2 * OSROW + (NCG > 1 ? NCG * 4096 : 0) + (sizeof(TT) == 1 ? 768 : 0)) * 4;
#define SRLZ_OS_FWD(DECV)                                                                                                       \
  do {                                                                                                                        \
    SRLZ_MAX_LDS((convT_out_os_kernel<NCG, LOSS, TT, DECV>), lds);                                                            \
    hipLaunchKernelGGL((convT_out_os_kernel<NCG, LOSS, TT, DECV>), dim3(g.grid), dim3(256), lds, st, x, w, bias, out, d->n, d->himg, \
                       d->wimg, d->hf, d->wf, bnp, os_npg(d), target, dec, loss_partial, d->n / 2 > 0 ? d->n / 2 : 1, lut, g.rows, \
                       g.nseg, g.nchunk);                                                                                     \
  } while (0)
  if constexpr (LOSS) { if (dec) SRLZ_OS_FWD(true); else SRLZ_OS_FWD(false); }
  else SRLZ_OS_FWD(false);
#undef SRLZ_OS_FWD
  SRLZ_LAUNCHED();
  return 0;
}

template <bool LOSS, typename TT>
int os_fwd(const float* x, const float* w, const float* bias, float* out, const float* bnp, const TT* target, float* dec,
           double* loss_partial, const float* lut, const srlz_skinny_desc* d, hipStream_t st) {
  switch (d->c) {
    case 3: return os_fwd_launch<1, LOSS, TT>(x, w, bias, out, bnp, target, dec, loss_partial, lut, d, st);
    case 6: return os_fwd_launch<2, LOSS, TT>(x, w, bias, out, bnp, target, dec, loss_partial, lut, d, st);
    default: return os_fwd_launch<3, LOSS, TT>(x, w, bias, out, bnp, target, dec, loss_partial, lut, d, st);
  }
}

__global__ void os_chan_sum_final(const double* __restrict__ partial, int n, float* __restrict__ out) {
  const int c = blockIdx.x;
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) s += partial[(size_t)c * n + i];
  s = wave_sum_d(s);
  if (threadIdx.x == 0) out[c] = (float)s;
}

// dw_ref[ci][cg*3+co][ky][kx] = sum over workgroups of partial[cg][wg][ci][co*16 + ky*4 + kx]; fp64, fixed order
__global__ __launch_bounds__(1024) void os_wgrad_reduce(const float* __restrict__ partial, int nwg, int C, float* __restrict__ dw_ref) {
  constexpr int OUTS = 64, PARTS = 16;
  const int o = threadIdx.x & (OUTS - 1), part = threadIdx.x / OUTS;
  const int id = blockIdx.x * OUTS + o;  // (cg, ci, k) with k fastest; the grid covers NCG * 64 * 48 exactly
  const int cg = id / (64 * 48), rem = id - cg * (64 * 48);
  const float* base = partial + (size_t)cg * nwg * (64 * 48) + rem;
  const int per = (nwg + PARTS - 1) / PARTS;
  const int w0 = part * per, w1 = (w0 + per < nwg) ? w0 + per : nwg;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int w = w0;
  for (; w + 3 < w1; w += 4) {
    s0 += (double)base[(size_t)w * (64 * 48)];
    s1 += (double)base[(size_t)(w + 1) * (64 * 48)];
    s2 += (double)base[(size_t)(w + 2) * (64 * 48)];
    s3 += (double)base[(size_t)(w + 3) * (64 * 48)];
  }
  for (; w < w1; ++w) s0 += (double)base[(size_t)w * (64 * 48)];
  __shared__ double sm[PARTS][OUTS];
  sm[part][o] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (part == 0) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < PARTS; ++q) t += sm[q][o];
    const int ci = rem / 48, k = rem - ci * 48;
    dw_ref[((size_t)ci * C + cg * 3) * 16 + k] = (float)t;
  }
}

}  // namespace

// ---- C ABI (declared in include/srlz.h) ---------------------------------------------------------------------------------
extern "C" int srlz_convT_out_fwd(const float* x_nhwc, const float* w_ref, const float* bias, float* y_nchw, const float* x_bnp,
                                  const srlz_skinny_desc* d, srlz_stream_t stream) {
  if (int rc = os_check(d, "convT_out_fwd")) return rc;
  SRLZ_REQUIRE(x_nhwc && w_ref && y_nchw, SRLZ_ERR_NULL, "convT_out_fwd: null pointer");
  return os_fwd<false, float>(x_nhwc, w_ref, bias, y_nchw, x_bnp, nullptr, nullptr, nullptr, nullptr, d, as_stream(stream));
}

extern "C" int srlz_convT_out_fwd_loss_workgroups(const srlz_skinny_desc* d) {
  if (os_check(d, "convT_out_fwd_loss_workgroups")) return -1;
  return fwd_geo(d).grid;
}

template <typename TT>
static int os_fwd_loss(const float* x_nhwc, const float* w_ref, const float* bias, const TT* target, const float* lut, float* err_nchw,
                       float* dec_nchw, const float* x_bnp, double* loss_partial, const srlz_skinny_desc* d, srlz_stream_t stream) {
  if (int rc = os_check(d, "convT_out_fwd_loss")) return rc;
  SRLZ_REQUIRE(x_nhwc && w_ref && target && err_nchw && loss_partial, SRLZ_ERR_NULL, "convT_out_fwd_loss: null pointer");
  SRLZ_REQUIRE(d->n % 2 == 0, SRLZ_ERR_BAD_DESC, "convT_out_fwd_loss: the batch is the two frames of a step (n = %d is odd)", d->n);
  SRLZ_REQUIRE((((uintptr_t)loss_partial) & 7) == 0, SRLZ_ERR_BAD_DESC, "convT_out_fwd_loss: unaligned partial buffer");
  return os_fwd<true, TT>(x_nhwc, w_ref, bias, err_nchw, x_bnp, target, dec_nchw, loss_partial, lut, d, as_stream(stream));
}

extern "C" int srlz_convT_out_fwd_loss(const float* x_nhwc, const float* w_ref, const float* bias, const float* target_nchw,
                                       float* err_nchw, float* dec_nchw, const float* x_bnp, double* loss_partial,
                                       const srlz_skinny_desc* d, srlz_stream_t stream) {
  return os_fwd_loss<float>(x_nhwc, w_ref, bias, target_nchw, nullptr, err_nchw, dec_nchw, x_bnp, loss_partial, d, stream);
}

extern "C" int srlz_convT_out_fwd_loss_u8(const float* x_nhwc, const float* w_ref, const float* bias, const uint8_t* target_u8,
                                          const float* norm_lut, float* err_nchw, float* dec_nchw, const float* x_bnp,
                                          double* loss_partial, const srlz_skinny_desc* d, srlz_stream_t stream) {
  SRLZ_REQUIRE(norm_lut, SRLZ_ERR_NULL, "convT_out_fwd_loss_u8: null normalisation table");
  return os_fwd_loss<uint8_t>(x_nhwc, w_ref, bias, target_u8, norm_lut, err_nchw, dec_nchw, x_bnp, loss_partial, d, stream);
}

extern "C" int srlz_convT_out_bwd_fused_supported(const srlz_skinny_desc* d) {
  return d && d->kind == 1 && (d->c == 3 || d->c == 6) ? 1 : 0;
}

extern "C" int srlz_convT_out_bwd_fused_tiles(const srlz_skinny_desc* d) {
  if (os_check(d, "convT_out_bwd_fused_tiles")) return -1;
  return bwd_geo(d).njobs;
}

extern "C" size_t srlz_convT_out_bwd_fused_workspace(const srlz_skinny_desc* d) {
  if (os_check(d, "convT_out_bwd_fused_workspace")) return 0;
  const OsGeo g = bwd_geo(d);
  return (size_t)g.grid * (d->c / 3) * 64 * 48 * sizeof(float) + (size_t)g.grid * d->c * sizeof(double);
}

extern "C" int srlz_convT_out_bwd_fused(const float* dy_nchw, const float* w_ref, float* dx_nhwc, const float* x_raw,
                                        const float* x_bnp, float* bn_bwd_partial, float* dw_ref, float* dbias, void* ws,
                                        size_t ws_bytes, const float* dy_gain_dev, float dy_gain_div, float dy_gain_coef,
                                        const srlz_skinny_desc* d, srlz_stream_t stream) {
  if (int rc = os_check(d, "convT_out_bwd_fused")) return rc;
  SRLZ_REQUIRE(d->c == 3 || d->c == 6, SRLZ_ERR_BAD_DESC, "convT_out_bwd_fused: 3 or 6 image channels (got %d)", d->c);
  SRLZ_REQUIRE(dy_nchw && w_ref && dx_nhwc && x_raw && x_bnp && bn_bwd_partial && dw_ref && ws, SRLZ_ERR_NULL,
               "convT_out_bwd_fused: null pointer");
  SRLZ_REQUIRE(ws_bytes >= srlz_convT_out_bwd_fused_workspace(d), SRLZ_ERR_WORKSPACE,
               "convT_out_bwd_fused: workspace too small (%zu)", ws_bytes);
  SRLZ_REQUIRE(dy_gain_dev == nullptr || dy_gain_div != 0.f, SRLZ_ERR_BAD_DESC, "convT_out_bwd_fused: dy_gain_div is zero");
  hipStream_t st = as_stream(stream);
  const OsGeo g = bwd_geo(d);
  const int ncg = d->c / 3;
  float* partial = (float*)ws;
  double* bias_part = dbias ? (double*)((char*)ws + (size_t)g.grid * ncg * 64 * 48 * sizeof(float)) : nullptr;  // [C][grid]
  SRLZ_REQUIRE((((uintptr_t)bias_part) & 7) == 0, SRLZ_ERR_BAD_DESC, "convT_out_bwd_fused: unaligned workspace");
#define SRLZ_OS_BWD(NCG)                                                                                                      \
  do {                                                                                                                        \
    size_t fl = (size_t)4 * ((NCG) * 3 * 8 * RP + 16 * OSP + 192) + ((NCG) > 1 ? (NCG) * 4 * 12 * 64 : 0);                          \
    if (fl < 4 * 64 * 48) fl = 4 * 64 * 48;                                                                                   \
    SRLZ_MAX_LDS((convT_out_os_bwd_kernel<NCG>), fl * 4);                                                                     \
    hipLaunchKernelGGL((convT_out_os_bwd_kernel<NCG>), dim3(g.grid), dim3(256), fl * 4, st, dy_nchw, w_ref, dx_nhwc, bn_bwd_partial, \
                       partial, d->n, d->himg, d->wimg, d->hf, d->wf, x_raw, x_bnp, os_npg(d), bias_part, dy_gain_dev,        \
                       dy_gain_div, dy_gain_coef, g.rows, g.nseg, g.nchunk);                                                  \
  } while (0)
  if (ncg == 1) SRLZ_OS_BWD(1); else SRLZ_OS_BWD(2);
#undef SRLZ_OS_BWD
  SRLZ_LAUNCHED();
  hipLaunchKernelGGL(os_wgrad_reduce, dim3(ncg * 64 * 48 / 64), dim3(1024), 0, st, partial, g.grid, d->c, dw_ref);
  SRLZ_LAUNCHED();
  if (dbias) {
    hipLaunchKernelGGL(os_chan_sum_final, dim3(d->c), dim3(64), 0, st, bias_part, g.grid, dbias);
    SRLZ_LAUNCHED();
  }
  return 0;
}
