// bn_pool.hip — BatchNorm2d(64) statistics, fused BN-apply + ReLU (+ MaxPool 3x3 s2) forward/backward, layout seams.
// Replaces nn.BatchNorm2d / nn.ReLU / nn.MaxPool2d of /root/reference/models/models.py:50-52,55-57,60-62,67-80.
// All kernels are HBM-bound: NHWC rows are 256 B, one thread handles 4 channels (float4), 16 threads one pixel.
// Reductions: per-block partials (fp32 inside a block, fp64 across blocks) in a fixed order -> deterministic.
#include "common.h"

namespace {

constexpr int STAGE_ROWS = 128;  // rows of the intermediate [STAGE_ROWS][128] fp64 buffer

// Stage A of every per-channel reduction: rows[nrows][128] (fp32 or fp64) -> out[gridDim.x][128] fp64.
// Block b sums rows b*2+half, stepping 2*gridDim.x; fixed assignment -> deterministic.
// blockIdx.y = BatchNorm group: its rows are rows[g*nrows .. (g+1)*nrows), its output out[g*gridDim.x .. ][128].
template <typename T>
__global__ __launch_bounds__(256) void reduce_rows_kernel(const T* __restrict__ rows, int nrows, double* __restrict__ out) {
  rows += (size_t)blockIdx.y * nrows * 128;
  out += (size_t)blockIdx.y * gridDim.x * 128;
  const int col = threadIdx.x & 127, half = threadIdx.x >> 7;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  const int step = 2 * gridDim.x;
  int r = blockIdx.x * 2 + half;
  for (; r + 3 * step < nrows; r += 4 * step) {
    a0 += (double)rows[(size_t)r * 128 + col];
    a1 += (double)rows[(size_t)(r + step) * 128 + col];
    a2 += (double)rows[(size_t)(r + 2 * step) * 128 + col];
    a3 += (double)rows[(size_t)(r + 3 * step) * 128 + col];
  }
  for (; r < nrows; r += step) a0 += (double)rows[(size_t)r * 128 + col];
  __shared__ double sm[128];
  const double s = (a0 + a1) + (a2 + a3);
  if (half) sm[col] = s;
  __syncthreads();
  if (!half) out[(size_t)blockIdx.x * 128 + col] = s + sm[col];
}

static int stage_blocks(int nrows) {
  int g = (nrows + 7) / 8;
  if (g > STAGE_ROWS) g = STAGE_ROWS;
  return g < 1 ? 1 : g;
}

// bnp record: [0,64) mean, [64,128) invstd, [128,192) scale, [192,256) shift
// `groups` independent BatchNorm calls (one per group of images) are finalised one after the other by the same block:
// group g's staged partials are partial[g*n_partials ..], its records bnp[g*256 ..] / batch_stat[g*128 ..], and the running
// statistics receive the groups' momentum updates IN ORDER (obs, then next_obs — models/learner.py:392-393).
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const double* __restrict__ partial, int n_partials, double count,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                   float momentum, int repeat, float* __restrict__ running_mean,
                                   float* __restrict__ running_var, float* __restrict__ bnp,
                                   float* __restrict__ batch_stat, int groups, long long* __restrict__ tick) {
  // tick (may be NULL): nn.BatchNorm2d.num_batches_tracked, advanced by the number of calls this launch stands for
  if (tick && blockIdx.x == 0 && threadIdx.x == 0) tick[0] += groups;
  // blockIdx.x > 0 only in CHANNEL-BLOCK mode (srlz_bn_finalize_chunks: a C-channel BatchNorm handled as C/64 independent
  // 64-channel layers, one block each): everything per-channel moves on by 64, the staged partials by one block's rows
  // (with `groups` > 1 in channel-block mode — the views of a triplet step batched along n — a block's staged partials are
  // [group][n_partials][128] and the records come out group-major, bnp[(g * blocks + block) * 256 ..]: every group's records are one
  // contiguous [blocks][256] run, which is what a single-group consumer expects)
  gamma += blockIdx.x * 64; beta += blockIdx.x * 64;
  if (running_mean) running_mean += blockIdx.x * 64;
  if (running_var) running_var += blockIdx.x * 64;
  bnp += blockIdx.x * 256;
  partial += (size_t)blockIdx.x * groups * n_partials * 128;
  const int rec_stride = 256 * (int)gridDim.x;
  // one block of 1024 threads: thread (c = tid & 63, part = tid >> 6) sums a strided sixteenth of the partial records
  // (the chain of dependent loads, not bandwidth, is what this kernel waits for)
  constexpr int NP = 16;
  const int c = threadIdx.x & 63, part = threadIdx.x >> 6;
  __shared__ double sm[2][NP][64];
  float rm = 0.f, rv = 0.f;
  const bool upd = running_mean && running_var;
  if (part == 0 && upd) { rm = running_mean[c]; rv = running_var[c]; }
  for (int g = 0; g < groups; ++g) {
    const double* pg = partial + (size_t)g * n_partials * 128;
    double s = 0.0, q = 0.0;
    for (int i = part; i < n_partials; i += NP) {
      s += pg[(size_t)i * 128 + c];
      q += pg[(size_t)i * 128 + 64 + c];
    }
    __syncthreads();  // (the previous group's readers are done with sm)
    sm[0][part][c] = s; sm[1][part][c] = q;
    __syncthreads();
    if (part == 0) {
      s = 0.0; q = 0.0;
#pragma unroll
      for (int k = 0; k < NP; ++k) { s += sm[0][k][c]; q += sm[1][k][c]; }
      const double mean = s / count;
      double var = q / count - mean * mean;
      if (var < 0.0) var = 0.0;
      const float invstd = (float)(1.0 / sqrt(var + (double)eps));
      const float ga = gamma[c], be = beta[c];
      const float scale = ga * invstd;
      float* rec = bnp + g * rec_stride;
      rec[c] = (float)mean; rec[64 + c] = invstd; rec[128 + c] = scale; rec[192 + c] = be - (float)mean * scale;
      const double unbiased = (count > 1.0) ? var * count / (count - 1.0) : var;
      if (batch_stat) { batch_stat[g * 128 + c] = (float)mean; batch_stat[g * 128 + 64 + c] = (float)unbiased; }
      for (int r = 0; r < repeat; ++r) {
        rm = (1.f - momentum) * rm + momentum * (float)mean;
        rv = (1.f - momentum) * rv + momentum * (float)unbiased;
      }
    }
  }
  if (part == 0 && upd) { running_mean[c] = rm; running_var[c] = rv; }
}

__global__ void bn_eval_params_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                      float* bnp) {
  const int c = threadIdx.x;
  const float invstd = 1.f / sqrtf(rv[c] + eps);
  const float scale = gamma[c] * invstd;
  bnp[c] = rm[c]; bnp[64 + c] = invstd; bnp[128 + c] = scale; bnp[192 + c] = beta[c] - rm[c] * scale;
}

__global__ void bn_replay_kernel(const float* batch_stat, float momentum, float* rm, float* rv) {
  const int c = threadIdx.x;
  rm[c] = (1.f - momentum) * rm[c] + momentum * batch_stat[c];
  rv[c] = (1.f - momentum) * rv[c] + momentum * batch_stat[64 + c];
}

// ---- fused BN-apply + ReLU + MaxPool(3, stride 2, pad p) forward ----
// Element (pooled pixel, 4 channels) per thread.  argmax = window index ky*3+kx of the first maximum.
// A thread = a 2 x 2 block of pooled outputs x 4 channels (round 5; one output per thread until then): the four 3x3 stride-2 windows of
// a block share a 5 x 5 patch of y, so a thread issues 25 loads (6.25 per output) instead of 36 and evaluates relu(bn(.)) once per
// loaded element — the kernel moves HBM's algorithmic bytes once either way; same-box A/B: 182 -> 177 us per launch (3 launches per
// step: -16 us).  Same per-element arithmetic and the same scan order (ky, kx ascending, first maximum wins):
// pooled values and argmax bytes are bit-identical to the one-output form's.
// 1-D grid of gx * N * HB blocks (HB = block rows): gx blocks cover (bx, c4) of one block row -> no per-thread integer division.
// Block b runs on XCD b % 8; xcd_remap hands every XCD one contiguous run of (row, x-block) pairs: vertically adjacent block rows
// share a row of y, which in plain order would cross HBM twice, once into each XCD's L2 (rocprofv3 counted 1.5x the algorithmic reads).
// npg = images per BatchNorm group: image n uses record bnp[(n / npg) * 256 ..]
__global__ __launch_bounds__(256) void bn_relu_pool_fwd_kernel(const float* __restrict__ y, const float* __restrict__ bnp,
                                                               float* __restrict__ pooled, uint8_t* __restrict__ argmax,
                                                               int N, int H, int W, int HP, int WP, int pad, int out_nchw,
                                                               int npg, int gx) {
  const int HB = (HP + 1) >> 1, WB = (WP + 1) >> 1;
  const int t = xcd_remap(blockIdx.x, gridDim.x);
  const int brow = t / gx, xb = t - brow * gx;  // (uniform)
  const int c4 = threadIdx.x & 15;
  const int bx = (xb * blockDim.x + threadIdx.x) >> 4;
  if (bx >= WB) return;
  const int n = brow / HB, by = brow - n * HB;
  bnp += (n / npg) * 256;
  const f32x4 sc = *(const f32x4*)(bnp + 128 + c4 * 4);
  const f32x4 sh = *(const f32x4*)(bnp + 192 + c4 * 4);
  const int py0 = 2 * by, px0 = 2 * bx;
  const int iy0 = py0 * 2 - pad, ix0 = px0 * 2 - pad;
  // the 25 patch loads go out together, branch-free (a position outside the image reads a clamped address and is masked below)
  f32x4 win[25];
  unsigned inside = 0;
#pragma unroll
  for (int r = 0; r < 5; ++r)
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      const int iy = iy0 + r, ix = ix0 + c;
      const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
      win[r * 5 + c] = *(const f32x4*)(y + (ok ? ((size_t)(n * H + iy) * W + ix) * 64 : (size_t)0) + c4 * 4);
      inside |= (ok ? 1u : 0u) << (r * 5 + c);
    }
#pragma unroll
  for (int k = 0; k < 25; ++k) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float z = win[k][j] * sc[j] + sh[j];
      win[k][j] = z > 0.f ? z : 0.f;
    }
  }
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      const int py = py0 + dy, px = px0 + dx;
      if (py >= HP || px >= WP) continue;
      f32x4 best = {-1.f, -1.f, -1.f, -1.f};  // relu output is >= 0, so -1 marks "nothing seen yet"
      int bi[4] = {0, 0, 0, 0};
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int e = (2 * dy + ky) * 5 + 2 * dx + kx;
          const bool ok = (inside >> e) & 1u;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (ok && win[e][j] > best[j]) { best[j] = win[e][j]; bi[j] = ky * 3 + kx; }
        }
      const long long pix = ((long long)n * HP + py) * WP + px;
      if (out_nchw) {
#pragma unroll
        for (int j = 0; j < 4; ++j) pooled[((size_t)(n * 64 + c4 * 4 + j) * HP + py) * WP + px] = best[j];
      } else {
        *(f32x4*)(pooled + (size_t)pix * 64 + c4 * 4) = best;
      }
      if (argmax) {
        const uint32_t packed = (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24);
        *(uint32_t*)(argmax + (size_t)pix * 64 + c4 * 4) = packed;
      }
    }
}

// Per-block combine of the 16 pixel-rows' fp64 partial sums; partial[block][128] (fp64): [0,64) s1, [64,128) s2.
// The two BatchNorm-backward sums feed dy = scale*(dz - mean(dz) - xhat*mean(dz*xhat)): an error in either mean is a
// per-channel CONSTANT added to every dy element, which the following weight-gradient sums coherently over all
// (non-negative, post-ReLU) inputs.  fp32 accumulation here costs 1e-3..1e-2 of relative accuracy in dW; fp64 is free
// in an HBM-bound kernel.
__device__ __forceinline__ void block_reduce_store(const double (&s1)[4], const double (&s2)[4], double* __restrict__ partial) {
  __shared__ double sm[16][128];
  const int c4 = threadIdx.x & 15, prow = threadIdx.x >> 4;
#pragma unroll
  for (int j = 0; j < 4; ++j) { sm[prow][c4 * 4 + j] = s1[j]; sm[prow][64 + c4 * 4 + j] = s2[j]; }
  __syncthreads();
  if (threadIdx.x < 128) {
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += sm[r][threadIdx.x];
    partial[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 128 + threadIdx.x] = s;  // (blockIdx.y = BatchNorm group)
  }
}

// ---- backward, stage 1: per-channel sums of dz and dz*xhat over the pooled outputs (dz lives at the argmax) ----
// partial[block][128]: [0,64) sum dz, [64,128) sum dz*xhat
__global__ __launch_bounds__(256) void bn_relu_pool_bwd_reduce(const float* __restrict__ y, const float* __restrict__ bnp,
                                                              const uint8_t* __restrict__ argmax,
                                                              const float* __restrict__ dpooled,
                                                              const float* __restrict__ pooled, double* __restrict__ partial,
                                                              int N, int H, int W, int HP, int WP, int pad, int dp_nchw) {
  // N = images per BatchNorm group; blockIdx.y = group (its images start at blockIdx.y * N)
  bnp += blockIdx.y * 256;
  const int n0 = blockIdx.y * N;
  // dz is non-zero only at a window's argmax and only if the pooled value is positive.  The pooled value IS
  // z = scale*v + shift at that position, so xhat = ((z - shift)/scale - mean)*invstd needs no access to y — unless
  // `pooled` is not supplied or scale is (almost) 0, where v is gathered from y (one scattered 4-byte read per element).
  const int c4 = threadIdx.x & 15;
  const f32x4 mean = *(const f32x4*)(bnp + c4 * 4);
  const f32x4 invstd = *(const f32x4*)(bnp + 64 + c4 * 4);
  const f32x4 sc = *(const f32x4*)(bnp + 128 + c4 * 4);
  const f32x4 sh = *(const f32x4*)(bnp + 192 + c4 * 4);
  double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
  // block b walks pooled rows b, b + gridDim.x, ... (row = n*HP + py); 16 pixel-lanes x 16 channel-quads per block
  for (int lrow = blockIdx.x; lrow < N * HP; lrow += gridDim.x) {
    const int row = n0 * HP + lrow;
    const int n = row / HP, py = row - n * HP;
    for (int px = threadIdx.x >> 4; px < WP; px += 16) {
      const long long pix = (long long)row * WP + px;
      const uint32_t packed = *(const uint32_t*)(argmax + (size_t)pix * 64 + c4 * 4);
      f32x4 dp;
      if (dp_nchw) {
#pragma unroll
        for (int j = 0; j < 4; ++j) dp[j] = dpooled[((size_t)(n * 64 + c4 * 4 + j) * HP + py) * WP + px];
      } else {
        dp = *(const f32x4*)(dpooled + (size_t)pix * 64 + c4 * 4);
      }
      f32x4 pz = {0.f, 0.f, 0.f, 0.f};
      if (pooled) {
        if (dp_nchw) {
#pragma unroll
          for (int j = 0; j < 4; ++j) pz[j] = pooled[((size_t)(n * 64 + c4 * 4 + j) * HP + py) * WP + px];
        } else {
          pz = *(const f32x4*)(pooled + (size_t)pix * 64 + c4 * 4);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v, z;
        // (|scale| tiny against |shift|: z - shift cancels, (z - shift) / scale would carry ~ eps * |shift| / |scale| — the
        // same threshold as the epilogue form of these sums, conv64_dgrad_poolsum_kernel)
        if (pooled && fabsf(sc[j]) > 1e-3f * fabsf(sh[j])) {
          z = pz[j];
          v = (z - sh[j]) / sc[j];
        } else {
          const int a = (packed >> (8 * j)) & 0xff;
          const int iy = py * 2 - pad + a / 3, ix = px * 2 - pad + a % 3;
          v = y[((size_t)(n * H + iy) * W + ix) * 64 + c4 * 4 + j];
          z = v * sc[j] + sh[j];
        }
        if (z > 0.f) {
          s1[j] += (double)dp[j];
          s2[j] += (double)(dp[j] * ((v - mean[j]) * invstd[j]));
        }
      }
    }
  }
  block_reduce_store(s1, s2, partial);
}

// Per group g: sums[g*128 + 0..64) = sum dz, sums[g*128 + 64..128) = sum dz*xhat; dbeta / dgamma = their totals over the groups
// (the layer's parameters are shared by the batched calls).  partial: [groups][nblocks][128] fp64.
// 1024 threads: 8 row slices x 128 columns, combined through LDS in a fixed order.
__global__ __launch_bounds__(1024) void bn_bwd_finalize(const double* __restrict__ partial, int nblocks, float* __restrict__ sums,
                                                       float* __restrict__ dgamma, float* __restrict__ dbeta, int groups) {
  const int c = threadIdx.x & 127, part = threadIdx.x >> 7;
  __shared__ double sm[8][128];
  double total = 0.0;
  for (int g = 0; g < groups; ++g) {
    const double* pg = partial + (size_t)g * nblocks * 128;
    double s = 0.0;
    for (int i = part; i < nblocks; i += 8) s += pg[(size_t)i * 128 + c];
    __syncthreads();
    sm[part][c] = s;
    __syncthreads();
    if (part == 0) {
      s = ((sm[0][c] + sm[1][c]) + (sm[2][c] + sm[3][c])) + ((sm[4][c] + sm[5][c]) + (sm[6][c] + sm[7][c]));
      sums[g * 128 + c] = (float)s;
      total = (g == 0) ? s : total + s;
    }
  }
  if (part == 0) {
    // one group: (float)s exactly as before; several: the fp64 total, rounded once
    if (c < 64) { if (dbeta) dbeta[c] = (float)total; } else { if (dgamma) dgamma[c - 64] = (float)total; }
  }
}

// ---- backward, stage 2: dy = scale * (dz - m1 - xhat*m2) (training) or scale * dz (eval) for every y element ----
// One thread owns a 2x2 block of y positions x 4 channels.  With kernel 3 / stride 2 the block whose top-left corner
// has (iy + pad) even is covered by exactly the four pooling windows (py-1..py, px-1..px), py = (iy+pad)/2: the
// (even,even) corner belongs to all four, the two mixed corners to two each, the (odd,odd) corner to one.  So one
// thread loads 4 argmax words + 4 dp vectors (unconditionally, 16-byte) and resolves 9 membership tests for 4 outputs.
__global__ __launch_bounds__(256) void bn_relu_pool_bwd_apply(const float* __restrict__ y, const float* __restrict__ bnp,
                                                             const uint8_t* __restrict__ argmax,
                                                             const float* __restrict__ dpooled, const float* __restrict__ sums,
                                                             float* __restrict__ dy, int N, int H, int W, int HP, int WP,
                                                             int pad, int dp_nchw, int training, float inv_count, int npg, int gx) {
  // block grid over (by, bx): iy = 2*by - pad .. +1, ix = 2*bx - pad .. +1 ; by in [0, HB), bx in [0, WB)
  const int HB = (H + pad + 1) / 2, WB = (W + pad + 1) / 2;
  // 1-D grid of gx * N * HB blocks in XCD-contiguous order (see bn_relu_pool_fwd_kernel): gx blocks cover (bx, c4) of one block-row;
  // vertically adjacent block-rows share a pooled row (argmax, d pooled)
  {
    const int t = xcd_remap(blockIdx.x, gridDim.x);
    const int brow = t / gx, xb = t - brow * gx;  // (uniform)
    const int c4 = threadIdx.x & 15;
    const int bx = (xb * blockDim.x + threadIdx.x) >> 4;
    if (bx >= WB) return;
    const int n = brow / HB, by = brow - n * HB;
    bnp += (n / npg) * 256;   // this image's BatchNorm group (npg images per group)
    sums += (n / npg) * 128;
    const f32x4 mean = *(const f32x4*)(bnp + c4 * 4);
    const f32x4 invstd = *(const f32x4*)(bnp + 64 + c4 * 4);
    const f32x4 sc = *(const f32x4*)(bnp + 128 + c4 * 4);
    const f32x4 sh = *(const f32x4*)(bnp + 192 + c4 * 4);
    // Every load of the thread goes out up front, branch-free (clamped address + mask), and is waited for once before the first
    // store: with loads inside the bounds branches the compiler serialises them, and a load consumed after a conditional store
    // waits for that store.
    f32x4 m1 = {0.f, 0.f, 0.f, 0.f}, m2 = m1;
    if (training) { m1 = *(const f32x4*)(sums + c4 * 4); m2 = *(const f32x4*)(sums + 64 + c4 * 4); }
    // the four windows (wy, wx) in {by-1, by} x {bx-1, bx}
    uint32_t am[2][2];
    f32x4 dp[2][2];
    unsigned wok = 0;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int py = by - 1 + a, px = bx - 1 + b;
        const bool ok = py >= 0 && py < HP && px >= 0 && px < WP;
        const size_t pp = ok ? ((size_t)(n * HP + py) * WP + px) : (size_t)0;
        am[a][b] = *(const uint32_t*)(argmax + pp * 64 + c4 * 4);
        if (dp_nchw) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            dp[a][b][j] = dpooled[ok ? ((size_t)(n * 64 + c4 * 4 + j) * HP + py) * WP + px : (size_t)0];
        } else {
          dp[a][b] = *(const f32x4*)(dpooled + pp * 64 + c4 * 4);
        }
        wok |= (ok ? 1u : 0u) << (2 * a + b);
      }
    f32x4 yv[2][2];
    unsigned yok = 0;
#pragma unroll
    for (int ry = 0; ry < 2; ++ry)
#pragma unroll
      for (int rx = 0; rx < 2; ++rx) {
        const int iy = 2 * by - pad + ry, ix = 2 * bx - pad + rx;
        const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
        yv[ry][rx] = *(const f32x4*)(y + (ok ? ((size_t)(n * H + iy) * W + ix) * 64 : (size_t)0) + c4 * 4);
        yok |= (ok ? 1u : 0u) << (2 * ry + rx);
      }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        asm volatile("" : "+v"(am[a][b]), "+v"(dp[a][b]), "+v"(yv[a][b]));
        if (!((wok >> (2 * a + b)) & 1u)) am[a][b] = 0xffffffffu;  // matches no window index
      }
#pragma unroll
    for (int ry = 0; ry < 2; ++ry) {
      const int iy = 2 * by - pad + ry;
#pragma unroll
      for (int rx = 0; rx < 2; ++rx) {
        const int ix = 2 * bx - pad + rx;
        if (!((yok >> (2 * ry + rx)) & 1u)) continue;
        const size_t pix = ((size_t)(n * H + iy) * W + ix);
        const f32x4 v = yv[ry][rx];
        f32x4 dz = {0.f, 0.f, 0.f, 0.f};
        // row ry = 0: windows by-1 (ky = 2) and by (ky = 0); ry = 1: window by only (ky = 1)
#pragma unroll
        for (int a = (ry ? 1 : 0); a < 2; ++a) {
          const int ky = ry ? 1 : (a ? 0 : 2);
#pragma unroll
          for (int b = (rx ? 1 : 0); b < 2; ++b) {
            const int kx = rx ? 1 : (b ? 0 : 2);
            const uint32_t me = (uint32_t)(ky * 3 + kx);
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (((am[a][b] >> (8 * j)) & 0xffu) == me) dz[j] += dp[a][b][j];
          }
        }
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float z = v[j] * sc[j] + sh[j];
          const float d = z > 0.f ? dz[j] : 0.f;
          if (training) {
            const float xh = (v[j] - mean[j]) * invstd[j];
            o[j] = sc[j] * (d - m1[j] * inv_count - xh * m2[j] * inv_count);
          } else {
            o[j] = sc[j] * d;
          }
        }
        *(f32x4*)(dy + pix * 64 + c4 * 4) = o;
      }
    }
  }
}

// ---- BN-apply + ReLU (decoder) ----
__global__ __launch_bounds__(256) void bn_relu_fwd_kernel(const float* __restrict__ y, const float* __restrict__ bnp,
                                                         float* __restrict__ a, long long pixels) {
  const long long total = pixels * 16;
  for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(id & 15);
    const f32x4 sc = *(const f32x4*)(bnp + 128 + c4 * 4);
    const f32x4 sh = *(const f32x4*)(bnp + 192 + c4 * 4);
    const f32x4 v = *(const f32x4*)(y + id * 4);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float z = v[j] * sc[j] + sh[j]; o[j] = z > 0.f ? z : 0.f; }
    *(f32x4*)(a + id * 4) = o;
  }
}

__global__ __launch_bounds__(256) void bn_relu_bwd_reduce(const float* __restrict__ y, const float* __restrict__ bnp,
                                                         const float* __restrict__ da, double* __restrict__ partial,
                                                         long long pixels) {
  // pixels = positions of ONE BatchNorm group; blockIdx.y = group
  bnp += blockIdx.y * 256;
  y += (size_t)blockIdx.y * pixels * 64;
  da += (size_t)blockIdx.y * pixels * 64;
  const int c4 = threadIdx.x & 15;
  const f32x4 mean = *(const f32x4*)(bnp + c4 * 4);
  const f32x4 invstd = *(const f32x4*)(bnp + 64 + c4 * 4);
  const f32x4 sc = *(const f32x4*)(bnp + 128 + c4 * 4);
  const f32x4 sh = *(const f32x4*)(bnp + 192 + c4 * 4);
  double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
  for (long long pix = (long long)blockIdx.x * 16 + (threadIdx.x >> 4); pix < pixels; pix += (long long)gridDim.x * 16) {
    const f32x4 v = *(const f32x4*)(y + pix * 64 + c4 * 4);
    const f32x4 d = *(const f32x4*)(da + pix * 64 + c4 * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float z = v[j] * sc[j] + sh[j];
      if (z > 0.f) { s1[j] += (double)d[j]; s2[j] += (double)(d[j] * ((v[j] - mean[j]) * invstd[j])); }
    }
  }
  block_reduce_store(s1, s2, partial);
}

__global__ __launch_bounds__(256) void bn_relu_bwd_apply(const float* __restrict__ y, const float* __restrict__ bnp,
                                                        const float* __restrict__ da, const float* __restrict__ sums,
                                                        float* __restrict__ dy, long long pixels, int training,
                                                        float inv_count) {
  // pixels = positions of ONE BatchNorm group; blockIdx.y = group
  bnp += blockIdx.y * 256;
  sums += blockIdx.y * 128;
  y += (size_t)blockIdx.y * pixels * 64;
  da += (size_t)blockIdx.y * pixels * 64;
  dy += (size_t)blockIdx.y * pixels * 64;
  const long long total = pixels * 16;
  for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(id & 15);
    const f32x4 mean = *(const f32x4*)(bnp + c4 * 4);
    const f32x4 invstd = *(const f32x4*)(bnp + 64 + c4 * 4);
    const f32x4 sc = *(const f32x4*)(bnp + 128 + c4 * 4);
    const f32x4 sh = *(const f32x4*)(bnp + 192 + c4 * 4);
    const f32x4 v = *(const f32x4*)(y + id * 4);
    const f32x4 d = *(const f32x4*)(da + id * 4);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float z = v[j] * sc[j] + sh[j];
      const float g = z > 0.f ? d[j] : 0.f;
      if (training) {
        const float xh = (v[j] - mean[j]) * invstd[j];
        o[j] = sc[j] * (g - sums[c4 * 4 + j] * inv_count - xh * sums[64 + c4 * 4 + j] * inv_count);
      } else {
        o[j] = sc[j] * g;
      }
    }
    *(f32x4*)(dy + id * 4) = o;
  }
}

__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int C, int HW) {
  const long long total = (long long)N * C * HW;
  for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(id % C);
    const long long t = id / C;
    const int p = (int)(t % HW);
    const long long n = t / HW;
    dst[id] = src[((size_t)n * C + c) * HW + p];
  }
}
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int C, int HW) {
  const long long total = (long long)N * C * HW;
  for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long long)gridDim.x * blockDim.x) {
    const int p = (int)(id % HW);
    const long long t = id / HW;
    const int c = (int)(t % C);
    const long long n = t / C;
    dst[id] = src[((size_t)n * HW + p) * C + c];
  }
}

static int grid_for(long long work_items, int per_block) {
  long long b = (work_items + per_block - 1) / per_block;
  const long long cap = 8LL * srlz_device_cus();
  if (b > cap) b = cap;
  return b < 1 ? 1 : (int)b;
}

constexpr int RED_BLOCKS = 1024;
constexpr int MAX_GROUPS = 8;  // (RED_BLOCKS + STAGE_ROWS) rows of scratch hold MAX_GROUPS x STAGE_ROWS staged rows
static int norm_groups(int groups) { return groups > 1 ? groups : 1; }

static int check_pool(const srlz_pool_desc* d) {
  SRLZ_REQUIRE(d != nullptr, SRLZ_ERR_NULL, "pool: null descriptor");
  SRLZ_REQUIRE(d->n > 0 && (d->pool_pad == 0 || d->pool_pad == 1), SRLZ_ERR_BAD_DESC, "pool: bad descriptor");
  SRLZ_REQUIRE(d->groups >= 0 && d->groups <= 8 && (d->groups <= 1 || d->n % d->groups == 0), SRLZ_ERR_BAD_DESC,
               "pool: n = %d is not a multiple of groups = %d", d->n, d->groups);
  SRLZ_REQUIRE(d->hp == (d->h + 2 * d->pool_pad - 3) / 2 + 1 && d->wp == (d->w + 2 * d->pool_pad - 3) / 2 + 1,
               SRLZ_ERR_BAD_DESC, "pool: pooled size %dx%d inconsistent with %dx%d pad %d", d->hp, d->wp, d->h, d->w, d->pool_pad);
  return 0;
}

}  // namespace

extern "C" int srlz_bn_finalize(const float* stats_partial, int n_partials, int groups, long long count, const float* gamma,
                                const float* beta, float eps, float momentum, int repeat, float* running_mean,
                                float* running_var, long long* num_batches_tracked, float* bnp, float* batch_stat, void* ws,
                                size_t ws_bytes, srlz_stream_t stream) {
  SRLZ_REQUIRE(stats_partial && gamma && beta && bnp && ws, SRLZ_ERR_NULL, "bn_finalize: null pointer");
  const int G = norm_groups(groups);
  SRLZ_REQUIRE(n_partials > 0 && count > 0 && G <= MAX_GROUPS && n_partials % G == 0, SRLZ_ERR_BAD_DESC,
               "bn_finalize: %d partial records do not split into %d groups", n_partials, G);
  SRLZ_REQUIRE(ws_bytes >= srlz_bn_bwd_workspace(0), SRLZ_ERR_WORKSPACE, "bn_finalize: workspace too small");
  double* staged = (double*)ws;
  const int per = n_partials / G;
  const int g = stage_blocks(per);  // per group exactly what a single-group call uses: results are bit-identical to G calls
  hipLaunchKernelGGL(reduce_rows_kernel<float>, dim3(g, G), dim3(256), 0, as_stream(stream), stats_partial, per, staged);
  SRLZ_LAUNCHED();
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(1024), 0, as_stream(stream), (const double*)staged, g, (double)count,
                     gamma, beta, eps, momentum, repeat, running_mean, running_var, bnp, batch_stat, G, num_batches_tracked);
  SRLZ_LAUNCHED();
  return 0;
}

extern "C" size_t srlz_bn_finalize_chunks_workspace(int chunks, int groups) {
  const size_t pairs = (size_t)(chunks > 1 ? chunks : 1) * norm_groups(groups);
  const size_t need = pairs * STAGE_ROWS * 128 * sizeof(double);
  const size_t base = srlz_bn_bwd_workspace(0);
  return need > base ? need : base;
}

extern "C" int srlz_bn_finalize_chunks(const float* stats_partial, int tiles, int chunks, int groups, long long count,
                                       const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                                       float* running_var, long long* num_batches_tracked, float* bnp, void* ws, size_t ws_bytes,
                                       srlz_stream_t stream) {
  SRLZ_REQUIRE(stats_partial && gamma && beta && bnp && ws, SRLZ_ERR_NULL, "bn_finalize_chunks: null pointer");
  const int G = norm_groups(groups);
  SRLZ_REQUIRE(tiles > 0 && count > 0 && chunks >= 1 && chunks <= MAX_GROUPS && G <= MAX_GROUPS && tiles % G == 0, SRLZ_ERR_BAD_DESC,
               "bn_finalize_chunks: %d tiles, %d channel blocks, %d groups", tiles, chunks, G);
  SRLZ_REQUIRE(ws_bytes >= srlz_bn_finalize_chunks_workspace(chunks, G), SRLZ_ERR_WORKSPACE, "bn_finalize_chunks: workspace too small");
  double* staged = (double*)ws;
  // stats_partial is [chunk][group][tiles / G][128] (what srlz_convn_fwd writes: chunk-major, a group's tiles contiguous); every
  // (chunk, group) pair is staged with exactly the geometry a single-group call uses, so G batched calls = G separate calls bit for bit
  const int per = tiles / G;
  const int g = stage_blocks(per);
  hipLaunchKernelGGL(reduce_rows_kernel<float>, dim3(g, chunks * G), dim3(256), 0, as_stream(stream), stats_partial, per, staged);
  SRLZ_LAUNCHED();
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(chunks), dim3(1024), 0, as_stream(stream), (const double*)staged, g, (double)count,
                     gamma, beta, eps, momentum, 1, running_mean, running_var, bnp, (float*)nullptr, G, num_batches_tracked);
  SRLZ_LAUNCHED();
  return 0;
}

// eval-mode records for a C-channel BatchNorm: one 256-float record per block of 64 channels
__global__ void bn_eval_params_chunks_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                             float* bnp) {
  const int c = threadIdx.x, ch = blockIdx.x * 64 + c;
  const float invstd = 1.f / sqrtf(rv[ch] + eps);
  const float scale = gamma[ch] * invstd;
  float* rec = bnp + blockIdx.x * 256;
  rec[c] = rm[ch]; rec[64 + c] = invstd; rec[128 + c] = scale; rec[192 + c] = beta[ch] - rm[ch] * scale;
}

extern "C" int srlz_bn_eval_params_chunks(const float* gamma, const float* beta, const float* running_mean,
                                          const float* running_var, float eps, int chunks, float* bnp, srlz_stream_t stream) {
  SRLZ_REQUIRE(gamma && beta && running_mean && running_var && bnp && chunks >= 1, SRLZ_ERR_NULL, "bn_eval_params_chunks: null pointer");
  hipLaunchKernelGGL(bn_eval_params_chunks_kernel, dim3(chunks), dim3(64), 0, as_stream(stream), gamma, beta, running_mean,
                     running_var, eps, bnp);
  SRLZ_LAUNCHED();
  return 0;
}

// out = relu(bn_a(a) + (b_bnp ? bn_b(b) : b)) over `pixels` x C, C = 64 * chunks — the tail of a ResNet BasicBlock
// (torchvision resnet.py: out = self.bn2(out); out += identity; out = self.relu(out)); records per block of 64 channels
__global__ __launch_bounds__(256) void bn_add_relu_kernel(const float* __restrict__ a, const float* __restrict__ a_bnp,
                                                         const float* __restrict__ b, const float* __restrict__ b_bnp,
                                                         float* __restrict__ out, long long pixels, int chunks,
                                                         long long per_group) {
  // per_group = float4 elements of one BatchNorm group (records are [group][chunk][256]; one group: per_group = the whole tensor)
  const int q4 = chunks * 16;  // float4 per pixel
  const long long total = pixels * q4;
  for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long long)gridDim.x * blockDim.x) {
    const int cq = (int)(id % q4);
    const int chunk = (int)(id / per_group) * chunks + (cq >> 4), c4 = cq & 15;
    const f32x4 sa = *(const f32x4*)(a_bnp + chunk * 256 + 128 + c4 * 4), ha = *(const f32x4*)(a_bnp + chunk * 256 + 192 + c4 * 4);
    const f32x4 va = *(const f32x4*)(a + id * 4);
    f32x4 vb = *(const f32x4*)(b + id * 4);
    if (b_bnp) {
      const f32x4 sb = *(const f32x4*)(b_bnp + chunk * 256 + 128 + c4 * 4), hb = *(const f32x4*)(b_bnp + chunk * 256 + 192 + c4 * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) vb[j] = vb[j] * sb[j] + hb[j];
    }
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float z = (va[j] * sa[j] + ha[j]) + vb[j]; o[j] = z > 0.f ? z : 0.f; }
    *(f32x4*)(out + id * 4) = o;
  }
}

extern "C" int srlz_bn_add_relu(const float* a, const float* a_bnp, const float* b, const float* b_bnp, float* out,
                                long long pixels, int chunks, int groups, srlz_stream_t stream) {
  SRLZ_REQUIRE(a && a_bnp && b && out, SRLZ_ERR_NULL, "bn_add_relu: null pointer");
  const int G = norm_groups(groups);
  SRLZ_REQUIRE(pixels > 0 && chunks >= 1 && pixels % G == 0, SRLZ_ERR_BAD_DESC, "bn_add_relu: %lld pixels, %d groups", pixels, G);
  hipLaunchKernelGGL(bn_add_relu_kernel, dim3(grid_for(pixels * chunks * 16, 256)), dim3(256), 0, as_stream(stream), a, a_bnp, b,
                     b_bnp, out, pixels, chunks, pixels / G * chunks * 16);
  SRLZ_LAUNCHED();
  return 0;
}

// out[n][c] = mean over hw of x[n][hw][c]  (nn.AdaptiveAvgPool2d((1,1)) / AvgPool2d(7) of torchvision's resnet18)
__global__ void avgpool_nhwc_kernel(const float* __restrict__ x, float* __restrict__ out, int hw, int c) {
  const int n = blockIdx.x;
  for (int ch = threadIdx.x; ch < c; ch += blockDim.x) {
    float s = 0.f;
    for (int p = 0; p < hw; ++p) s += x[((size_t)n * hw + p) * c + ch];
    out[(size_t)n * c + ch] = s / (float)hw;
  }
}

extern "C" int srlz_avgpool_nhwc(const float* x, float* out, int n, int hw, int c, srlz_stream_t stream) {
  SRLZ_REQUIRE(x && out && n > 0 && hw > 0 && c > 0, SRLZ_ERR_NULL, "avgpool: bad arguments");
  hipLaunchKernelGGL(avgpool_nhwc_kernel, dim3(n), dim3(256), 0, as_stream(stream), x, out, hw, c);
  SRLZ_LAUNCHED();
  return 0;
}

extern "C" int srlz_bn_eval_params(const float* gamma, const float* beta, const float* running_mean,
                                   const float* running_var, float eps, float* bnp, srlz_stream_t stream) {
  SRLZ_REQUIRE(gamma && beta && running_mean && running_var && bnp, SRLZ_ERR_NULL, "bn_eval_params: null pointer");
  hipLaunchKernelGGL(bn_eval_params_kernel, dim3(1), dim3(64), 0, as_stream(stream), gamma, beta, running_mean, running_var,
                     eps, bnp);
  SRLZ_LAUNCHED();
  return 0;
}

extern "C" int srlz_bn_replay(const float* batch_stat, float momentum, float* running_mean, float* running_var,
                              srlz_stream_t stream) {
  SRLZ_REQUIRE(batch_stat && running_mean && running_var, SRLZ_ERR_NULL, "bn_replay: null pointer");
  hipLaunchKernelGGL(bn_replay_kernel, dim3(1), dim3(64), 0, as_stream(stream), batch_stat, momentum, running_mean, running_var);
  SRLZ_LAUNCHED();
  return 0;
}

// several layers' replays in ONE launch (the VAE getStates quirk replays the three encoder BatchNorms of a frame: one launch instead of
// three + three counter increments); items travel by value in the kernel arguments
struct ReplayTable { srlz_bn_replay_item it[SRLZ_BN_REPLAY_MAX]; };
__global__ void bn_replay_many_kernel(const ReplayTable t, float momentum) {
  const srlz_bn_replay_item it = t.it[blockIdx.x];
  const int c = threadIdx.x;
  it.running_mean[c] = (1.f - momentum) * it.running_mean[c] + momentum * it.batch_stat[c];
  it.running_var[c] = (1.f - momentum) * it.running_var[c] + momentum * it.batch_stat[64 + c];
  if (c == 0 && it.num_batches_tracked) it.num_batches_tracked[0] += 1;
}

extern "C" int srlz_bn_replay_many(const srlz_bn_replay_item* items, int n, float momentum, srlz_stream_t stream) {
  SRLZ_REQUIRE(items && n >= 1 && n <= SRLZ_BN_REPLAY_MAX, SRLZ_ERR_BAD_DESC, "bn_replay_many: %d items (1 .. %d)", n, SRLZ_BN_REPLAY_MAX);
  ReplayTable t;
  for (int i = 0; i < n; ++i) {
    SRLZ_REQUIRE(items[i].batch_stat && items[i].running_mean && items[i].running_var, SRLZ_ERR_NULL, "bn_replay_many: null pointer");
    t.it[i] = items[i];
  }
  for (int i = n; i < SRLZ_BN_REPLAY_MAX; ++i) t.it[i] = items[0];
  hipLaunchKernelGGL(bn_replay_many_kernel, dim3(n), dim3(64), 0, as_stream(stream), t, momentum);
  SRLZ_LAUNCHED();
  return 0;
}

extern "C" int srlz_bn_relu_pool_fwd(const float* y, const float* bnp, float* pooled, uint8_t* argmax,
                                     const srlz_pool_desc* d, srlz_stream_t stream) {
  if (int rc = check_pool(d)) return rc;
  SRLZ_REQUIRE(y && bnp && pooled, SRLZ_ERR_NULL, "bn_relu_pool_fwd: null pointer");
  // (a 1-D grid: the 65535 blocks of a grid.y capped a call at 1149 images until round 5; what bounds it now are the 31-bit row index
  //  n * H + iy and the grid itself)
  const int HB = (d->hp + 1) / 2, WB = (d->wp + 1) / 2;
  const int gx = (WB * 16 + 255) / 256;
  SRLZ_REQUIRE((long long)gx * d->n * HB <= 0x7fffffffLL && (long long)d->n * (d->h + 4) <= 0x7fffffffLL, SRLZ_ERR_BAD_DESC,
               "pool: %d images of %d x %d are too many for one launch", d->n, d->h, d->w);
  hipLaunchKernelGGL(bn_relu_pool_fwd_kernel, dim3(gx * d->n * HB), dim3(256), 0, as_stream(stream), y, bnp, pooled, argmax, d->n, d->h,
                     d->w, d->hp, d->wp, d->pool_pad, d->out_nchw, d->n / norm_groups(d->groups), gx);
  SRLZ_LAUNCHED();
  return 0;
}

extern "C" size_t srlz_bn_bwd_workspace(long long elems) {
  (void)elems;
  return (size_t)(RED_BLOCKS + STAGE_ROWS) * 128 * sizeof(double) + MAX_GROUPS * 128 * sizeof(float);
}

// stage 1 of the pooled-block backward: sums[0..64) = sum dz, sums[64..128) = sum dz*xhat (also dbeta / dgamma)
static int pool_bwd_sums(const float* y, const float* bnp, const uint8_t* argmax, const float* dpooled, const float* pooled,
                         float* sums, float* dgamma, float* dbeta, void* ws, const srlz_pool_desc* d, hipStream_t st) {
  double* partial = (double*)ws;
  double* staged = partial + RED_BLOCKS * 128;
  const int G = norm_groups(d->groups), npg = d->n / G;
  int nb = npg * d->hp;
  if (nb > RED_BLOCKS / G) nb = RED_BLOCKS / G;
  hipLaunchKernelGGL(bn_relu_pool_bwd_reduce, dim3(nb, G), dim3(256), 0, st, y, bnp, argmax, dpooled, pooled, partial, npg, d->h, d->w,
                     d->hp, d->wp, d->pool_pad, d->out_nchw);
  SRLZ_LAUNCHED();
  int sg = stage_blocks(nb);
  if (sg > STAGE_ROWS / G) sg = STAGE_ROWS / G;
  hipLaunchKernelGGL(reduce_rows_kernel<double>, dim3(sg, G), dim3(256), 0, st, (const double*)partial, nb, staged);
  SRLZ_LAUNCHED();
  hipLaunchKernelGGL(bn_bwd_finalize, dim3(1), dim3(1024), 0, st, (const double*)staged, sg, sums, dgamma, dbeta, G);
  SRLZ_LAUNCHED();
  return 0;
}

extern "C" int srlz_bn_relu_pool_bwd_sums(const float* y, const float* bnp, const uint8_t* argmax, const float* dpooled,
                                          const float* pooled, float* sums, float* dgamma, float* dbeta, void* ws,
                                          size_t ws_bytes, const srlz_pool_desc* d, srlz_stream_t stream) {
  if (int rc = check_pool(d)) return rc;
  SRLZ_REQUIRE(y && bnp && argmax && dpooled && sums && ws, SRLZ_ERR_NULL, "bn_relu_pool_bwd_sums: null pointer");
  SRLZ_REQUIRE(ws_bytes >= srlz_bn_bwd_workspace(0), SRLZ_ERR_WORKSPACE, "bn_relu_pool_bwd_sums: workspace too small");
  return pool_bwd_sums(y, bnp, argmax, dpooled, pooled, sums, dgamma, dbeta, ws, d, as_stream(stream));
}

extern "C" int srlz_bn_relu_pool_bwd(const float* y, const float* bnp, const uint8_t* argmax, const float* dpooled,
                                     const float* pooled, float* dy, float* dgamma, float* dbeta, int training, void* ws,
                                     size_t ws_bytes, const srlz_pool_desc* d, srlz_stream_t stream) {
  if (int rc = check_pool(d)) return rc;
  SRLZ_REQUIRE(y && bnp && argmax && dpooled && dy && ws, SRLZ_ERR_NULL, "bn_relu_pool_bwd: null pointer");
  SRLZ_REQUIRE(ws_bytes >= srlz_bn_bwd_workspace(0), SRLZ_ERR_WORKSPACE, "bn_relu_pool_bwd: workspace too small");
  hipStream_t st = as_stream(stream);
  float* sums = (float*)((double*)ws + (RED_BLOCKS + STAGE_ROWS) * 128);  // [groups][128]
  if (int rc = pool_bwd_sums(y, bnp, argmax, dpooled, pooled, sums, dgamma, dbeta, ws, d, st)) return rc;
  const int HB = (d->h + d->pool_pad + 1) / 2, WB = (d->w + d->pool_pad + 1) / 2;
  SRLZ_REQUIRE((long long)((WB * 16 + 255) / 256) * d->n * HB <= 0x7fffffffLL && (long long)d->n * (d->h + 2) <= 0x7fffffffLL,
               SRLZ_ERR_BAD_DESC, "pool: %d images of %d x %d are too many for one launch", d->n, d->h, d->w);
  const int npg = d->n / norm_groups(d->groups);
  const float inv_count = 1.0f / (float)((double)npg * d->h * d->w);
  const int gx = (WB * 16 + 255) / 256;
  hipLaunchKernelGGL(bn_relu_pool_bwd_apply, dim3(gx * d->n * HB), dim3(256), 0, st, y, bnp, argmax, dpooled, sums, dy,
                     d->n, d->h, d->w, d->hp, d->wp, d->pool_pad, d->out_nchw, training, inv_count, npg, gx);
  SRLZ_LAUNCHED();
  return 0;
}

extern "C" int srlz_bn_relu_pool_bwd_apply(const float* y, const float* bnp, const uint8_t* argmax, const float* dpooled,
                                           const float* sums, float* dy, int training, const srlz_pool_desc* d,
                                           srlz_stream_t stream) {
  if (int rc = check_pool(d)) return rc;
  SRLZ_REQUIRE(y && bnp && argmax && dpooled && sums && dy, SRLZ_ERR_NULL, "bn_relu_pool_bwd_apply: null pointer");
  const int HB = (d->h + d->pool_pad + 1) / 2, WB = (d->w + d->pool_pad + 1) / 2;
  SRLZ_REQUIRE((long long)((WB * 16 + 255) / 256) * d->n * HB <= 0x7fffffffLL && (long long)d->n * (d->h + 2) <= 0x7fffffffLL,
               SRLZ_ERR_BAD_DESC, "pool: %d images of %d x %d are too many for one launch", d->n, d->h, d->w);
  const int npg = d->n / norm_groups(d->groups);
  const float inv_count = 1.0f / (float)((double)npg * d->h * d->w);
  const int gx = (WB * 16 + 255) / 256;
  hipLaunchKernelGGL(bn_relu_pool_bwd_apply, dim3(gx * d->n * HB), dim3(256), 0, as_stream(stream), y, bnp, argmax,
                     dpooled, sums, dy, d->n, d->h, d->w, d->hp, d->wp, d->pool_pad, d->out_nchw, training, inv_count, npg, gx);
  SRLZ_LAUNCHED();
  return 0;
}

extern "C" int srlz_bn_relu_fwd(const float* y, const float* bnp, float* a, long long pixels, srlz_stream_t stream) {
  SRLZ_REQUIRE(y && bnp && a, SRLZ_ERR_NULL, "bn_relu_fwd: null pointer");
  hipLaunchKernelGGL(bn_relu_fwd_kernel, dim3(grid_for(pixels * 16, 256)), dim3(256), 0, as_stream(stream), y, bnp, a, pixels);
  SRLZ_LAUNCHED();
  return 0;
}

// pixels = positions of ALL groups together (a multiple of G)
static int bn_relu_bwd_sums_launch(const float* y, const float* bnp, const float* da, float* sums, float* dgamma, float* dbeta,
                                   void* ws, long long pixels, int G, hipStream_t st) {
  SRLZ_REQUIRE(G >= 1 && G <= MAX_GROUPS && pixels % G == 0, SRLZ_ERR_BAD_DESC, "bn_relu_bwd: %lld positions do not split into %d groups",
               pixels, G);
  double* partial = (double*)ws;
  double* staged = partial + RED_BLOCKS * 128;
  const long long ppg = pixels / G;
  int nb = (int)((ppg + 15) / 16);
  if (nb > RED_BLOCKS / G) nb = RED_BLOCKS / G;
  hipLaunchKernelGGL(bn_relu_bwd_reduce, dim3(nb, G), dim3(256), 0, st, y, bnp, da, partial, ppg);
  SRLZ_LAUNCHED();
  int sg = stage_blocks(nb);
  if (sg > STAGE_ROWS / G) sg = STAGE_ROWS / G;
  hipLaunchKernelGGL(reduce_rows_kernel<double>, dim3(sg, G), dim3(256), 0, st, (const double*)partial, nb, staged);
  SRLZ_LAUNCHED();
  hipLaunchKernelGGL(bn_bwd_finalize, dim3(1), dim3(1024), 0, st, (const double*)staged, sg, sums, dgamma, dbeta, G);
  SRLZ_LAUNCHED();
  return 0;
}

// Second stage for per-tile partials emitted by a data-gradient kernel's epilogue (srlz_convT_out_bwd_data with x_raw):
// partial[n_partials][128] (fp32: sum dz, sum dz*xhat per tile) -> sums / dgamma / dbeta, fp64 across tiles.
extern "C" int srlz_bn_bwd_finalize_partials(const float* partial, int n_partials, int groups, float* sums, float* dgamma,
                                             float* dbeta, void* ws, size_t ws_bytes, srlz_stream_t stream) {
  SRLZ_REQUIRE(partial && sums && ws, SRLZ_ERR_NULL, "bn_bwd_finalize_partials: null pointer");
  const int G = norm_groups(groups);
  SRLZ_REQUIRE(n_partials > 0 && G <= MAX_GROUPS && n_partials % G == 0, SRLZ_ERR_BAD_DESC,
               "bn_bwd_finalize_partials: %d records do not split into %d groups", n_partials, G);
  SRLZ_REQUIRE(ws_bytes >= srlz_bn_bwd_workspace(0), SRLZ_ERR_WORKSPACE, "bn_bwd_finalize_partials: workspace too small");
  hipStream_t st = as_stream(stream);
  double* staged = (double*)ws;
  const int per = n_partials / G;
  const int g = stage_blocks(per);
  hipLaunchKernelGGL(reduce_rows_kernel<float>, dim3(g, G), dim3(256), 0, st, partial, per, staged);
  SRLZ_LAUNCHED();
  hipLaunchKernelGGL(bn_bwd_finalize, dim3(1), dim3(1024), 0, st, (const double*)staged, g, sums, dgamma, dbeta, G);
  SRLZ_LAUNCHED();
  return 0;
}

extern "C" int srlz_bn_relu_bwd_sums(const float* y, const float* bnp, const float* da, float* sums, float* dgamma,
                                     float* dbeta, void* ws, size_t ws_bytes, long long pixels, int groups,
                                     srlz_stream_t stream) {
  SRLZ_REQUIRE(y && bnp && da && sums && ws, SRLZ_ERR_NULL, "bn_relu_bwd_sums: null pointer");
  SRLZ_REQUIRE(ws_bytes >= srlz_bn_bwd_workspace(0), SRLZ_ERR_WORKSPACE, "bn_relu_bwd_sums: workspace too small");
  return bn_relu_bwd_sums_launch(y, bnp, da, sums, dgamma, dbeta, ws, pixels, norm_groups(groups), as_stream(stream));
}

extern "C" int srlz_bn_relu_bwd(const float* y, const float* bnp, const float* da, float* dy, float* dgamma, float* dbeta,
                                int training, void* ws, size_t ws_bytes, long long pixels, int groups, srlz_stream_t stream) {
  SRLZ_REQUIRE(y && bnp && da && dy && ws, SRLZ_ERR_NULL, "bn_relu_bwd: null pointer");
  SRLZ_REQUIRE(ws_bytes >= srlz_bn_bwd_workspace(0), SRLZ_ERR_WORKSPACE, "bn_relu_bwd: workspace too small");
  hipStream_t st = as_stream(stream);
  const int G = norm_groups(groups);
  float* sums = (float*)((double*)ws + (RED_BLOCKS + STAGE_ROWS) * 128);  // [G][128]
  if (int rc = bn_relu_bwd_sums_launch(y, bnp, da, sums, dgamma, dbeta, ws, pixels, G, st)) return rc;
  const long long ppg = pixels / G;
  const float inv_count = 1.0f / (float)(double)ppg;
  hipLaunchKernelGGL(bn_relu_bwd_apply, dim3(grid_for(ppg * 16, 256), G), dim3(256), 0, st, y, bnp, da, sums, dy, ppg,
                     training, inv_count);
  SRLZ_LAUNCHED();
  return 0;
}

extern "C" int srlz_nchw_to_nhwc(const float* src, float* dst, int n, int c, int h, int w, srlz_stream_t stream) {
  SRLZ_REQUIRE(src && dst, SRLZ_ERR_NULL, "nchw_to_nhwc: null pointer");
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for((long long)n * c * h * w, 256)), dim3(256), 0, as_stream(stream), src,
                     dst, n, c, h * w);
  SRLZ_LAUNCHED();
  return 0;
}

extern "C" int srlz_nhwc_to_nchw(const float* src, float* dst, int n, int c, int h, int w, srlz_stream_t stream) {
  SRLZ_REQUIRE(src && dst, SRLZ_ERR_NULL, "nhwc_to_nchw: null pointer");
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for((long long)n * c * h * w, 256)), dim3(256), 0, as_stream(stream), src,
                     dst, n, c, h * w);
  SRLZ_LAUNCHED();
  return 0;
}
