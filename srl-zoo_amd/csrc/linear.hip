// linear.hip — nn.Linear forward / backward as one strided fp32-MFMA GEMM.
// Replaces nn.Linear of /root/reference/models/autoencoders.py:94-100, models/vae.py:52-57,
// models/forward_inverse.py:16,48-55 (and their autograd backward).
//   C[i][j] = sum_r A(i,r) * B(r,j)  (+ bias[j]) (+ ReLU) (+ res[i*ldr + j]),  A(i,r) = A[i*sai + r*sar],  B(r,j) = B[r*sbr + j*sbj]
//   (res: the residual of the forward model, next_state = state + Linear([state ; onehot(action)]), forward_inverse.py:27-37 — a
//    separately rounded fp32 add behind bias and ReLU, as `state + self.forward_net(concat)` rounds it)
//   forward   : i=m, j=n, r=k : A = x  (sai=K, sar=1), B = w (sbr=1, sbj=K)
//   data grad : i=m, j=k, r=n : A = dy (sai=N, sar=1), B = w (sbr=K, sbj=1)
//   weight grad: i=n, j=k, r=m : A = dy (sai=1, sar=N), B = x (sbr=K, sbj=1)
// 64x64 tile per workgroup, 4 waves = 4 quadrants of v_mfma_f32_32x32x2_f32.  Two kernels: gemm_vec_kernel (16-byte loads, r-steps
// of 64, ds_read_b128 operands) for operands made of whole aligned float4s — the model's layers at their training shapes — and
// gemm_strided_kernel (dword loads, r-steps of 32, [r][i] LDS layout) for everything else (e.g. K = 206 of the forward model).
#include "common.h"

namespace {

constexpr int BR = 32;
constexpr int LD = 68;  // 64 + 4 padding floats

// blockIdx.z = split of the reduction range [z*rchunk, min(R, (z+1)*rchunk)); with gridDim.z > 1 the kernel writes
// raw partial tiles to C + z*I*J (bias / ReLU are applied by splitk_combine_kernel).
__global__ __launch_bounds__(256) void gemm_strided_kernel(const float* __restrict__ A, long long sai, long long sar,
                                                          const float* __restrict__ B, long long sbr, long long sbj,
                                                          const float* __restrict__ bias, float* __restrict__ C, int I,
                                                          int J, int R, int relu, int rchunk, const float* __restrict__ res,
                                                          long long ldr) {
  // Double-buffered LDS tiles; the next r-step's global loads are issued before the current step's MFMAs and land in the
  // other buffer after them: one barrier per step and the load latency sits behind the matrix work.
  __shared__ float As[2][BR][LD];
  __shared__ float Bs[2][BR][LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int wi = wave & 1, wj = wave >> 1;
  const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  const int rbeg = blockIdx.z * rchunk;
  const int rend = (rbeg + rchunk < R) ? rbeg + rchunk : R;
  C += (size_t)blockIdx.z * I * J;
  constexpr int U = BR * 64 / 256;  // elements of a 64 x BR tile per thread
  float va[U], vb[U];
  unsigned ina = 0, inb = 0;  // bit u: element u is inside the matrix
  // Requests are branch-free (an element outside the matrix reads element 0 and is zeroed when it is stashed): with a load inside
  // a branch the compiler loses track of which loads have landed and falls back to "s_waitcnt vmcnt(0)" — here that made the
  // prefetch synchronous and put a full wait between every two stores of the epilogue (tools/isa_audit.py).
  auto fetch = [&](int r0) {
    ina = inb = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = tid + 256 * u;
      int i, r;
      if (sar == 1) { i = e / BR; r = e % BR; } else { r = e >> 6; i = e & 63; }
      const bool oka = i0 + i < I && r0 + r < rend;
      va[u] = A[oka ? (long long)(i0 + i) * sai + (long long)(r0 + r) * sar : 0LL];
      ina |= (oka ? 1u : 0u) << u;
      int j, rb;
      if (sbr == 1) { j = e / BR; rb = e % BR; } else { rb = e >> 6; j = e & 63; }
      const bool okb = j0 + j < J && r0 + rb < rend;
      vb[u] = B[okb ? (long long)(r0 + rb) * sbr + (long long)(j0 + j) * sbj : 0LL];
      inb |= (okb ? 1u : 0u) << u;
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = tid + 256 * u;
      int i, r;
      if (sar == 1) { i = e / BR; r = e % BR; } else { r = e >> 6; i = e & 63; }
      As[buf][r][i] = ((ina >> u) & 1u) ? va[u] : 0.f;
      int j, rb;
      if (sbr == 1) { j = e / BR; rb = e % BR; } else { rb = e >> 6; j = e & 63; }
      Bs[buf][rb][j] = ((inb >> u) & 1u) ? vb[u] : 0.f;
    }
  };
  fetch(rbeg);
  stash(0);
  __syncthreads();
  int buf = 0;
  for (int r0 = rbeg; r0 < rend; r0 += BR) {
    const bool more = r0 + BR < rend;
    if (more) fetch(r0 + BR);
#pragma unroll
    for (int s = 0; s < BR / 2; ++s) {
      const float a = As[buf][2 * s + h][wi * 32 + l31];
      const float b = Bs[buf][2 * s + h][wj * 32 + l31];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    if (more) stash(buf ^ 1);  // nobody reads buf^1 during this step (the barrier below closed the previous one)
    __syncthreads();
    buf ^= 1;
  }
  const int j = j0 + wj * 32 + l31;
  float bj = bias ? bias[j < J ? j : 0] : 0.f;
  asm volatile("" : "+v"(bj));  // waited for once, here, not behind every store of the loop below
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = i0 + wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
    if (i < I && j < J) {
      float v = acc[r] + bj;
      if (relu) v = v > 0.f ? v : 0.f;
      if (res) v = __fadd_rn(v, res[(long long)i * ldr + j]);
      C[(long long)i * J + j] = v;
    }
  }
}

// The same GEMM for operands whose rows and leading dimensions are multiples of four floats (every nn.Linear of the model at its
// training shapes; anything else takes gemm_strided_kernel): 16-byte global loads along whichever index is contiguous in memory,
// r-steps of 64 (four 16-byte loads per thread and operand in flight while the previous step's 32 MFMAs per wave run — these GEMMs are
// a handful of steps long, so what they wait for is memory latency, not the matrix pipe), and both LDS tiles kept [row][r] with r
// contiguous so that one ds_read_b128 feeds four MFMAs (lanes 0-31 take r = 8c .. 8c+3, lanes 32-63 r = 8c+4 .. 8c+7 of a chunk:
// the k-order of an MFMA is free as long as both operands agree — conv64.hip).  A row is 256 bytes; its 16-byte slot index is XORed with
// key(row) = (row & 3) << 2 | (row >> 2) & 3, a permutation of row & 15: conflict-free for the readers (consecutive rows) and for the
// transposing writers of an operand that is contiguous along its row index (rows 4 apart per lane).
// AI / BI: operand A / B is contiguous along i / j (else along r).
constexpr int TR = 64;
__device__ __forceinline__ int tile_key(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

template <bool AI, bool BI>
__global__ __launch_bounds__(256, 2) void gemm_vec_kernel(const float* __restrict__ A, long long lda, const float* __restrict__ B,
                                                         long long ldb, const float* __restrict__ bias, float* __restrict__ C,
                                                         int I, int J, int R, int relu, int rchunk, const float* __restrict__ res,
                                                         long long ldr) {
  // lda / ldb: floats between two consecutive values of the operand's NON-contiguous index
  __shared__ __attribute__((aligned(16))) float As[2][64 * TR];
  __shared__ __attribute__((aligned(16))) float Bs[2][64 * TR];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int wi = wave & 1, wj = wave >> 1;
  const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int rbeg = blockIdx.z * rchunk;
  const int rend = (rbeg + rchunk < R) ? rbeg + rchunk : R;
  C += (size_t)blockIdx.z * I * J;

  f32x4 va[4], vb[4];
  unsigned ina = 0, inb = 0;
  // element u of a thread: e = tid + 256 u.  Contiguous along r: row = e >> 4, r = 4 (e & 15) .. + 3.  Contiguous along the row index:
  // r = 4 (e >> 6) + (e & 3), rows 4 ((e >> 2) & 15) .. + 3 — four lanes take the same four rows at four consecutive r (their dword
  // writes into the transposed tile then fall into different banks), sixteen lanes cover 64 contiguous bytes of four r.
  auto fetch_one = [&](const float* __restrict__ P, long long ld, bool along_row, int row0, int nrow, int r0, f32x4 (&v)[4],
                       unsigned& in) {
    in = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = tid + 256 * u;
      long long off;
      bool ok;
      if (!along_row) {
        const int row = e >> 4, r = 4 * (e & 15);
        ok = row0 + row < nrow && r0 + r < rend;
        off = (long long)(row0 + row) * ld + (r0 + r);
      } else {
        const int r = 4 * (e >> 6) + (e & 3), row = 4 * ((e >> 2) & 15);
        ok = row0 + row < nrow && r0 + r < rend;
        off = (long long)(r0 + r) * ld + (row0 + row);
      }
      v[u] = *(const f32x4*)(P + (ok ? off : 0LL));  // branch-free: see gemm_strided_kernel
      in |= (ok ? 1u : 0u) << u;
    }
  };
  auto stash_one = [&](float* __restrict__ T, bool along_row, const f32x4 (&v)[4], unsigned in) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = tid + 256 * u;
      const f32x4 x = ((in >> u) & 1u) ? v[u] : f32x4{0.f, 0.f, 0.f, 0.f};
      if (!along_row) {
        const int row = e >> 4, slot = e & 15;
        *(f32x4*)(T + row * TR + ((slot ^ tile_key(row)) << 2)) = x;
      } else {
        const int r = 4 * (e >> 6) + (e & 3), row = 4 * ((e >> 2) & 15);
#pragma unroll
        for (int c = 0; c < 4; ++c) T[(row + c) * TR + ((((r >> 2) ^ tile_key(row + c)) << 2) | (r & 3))] = x[c];
      }
    }
  };
  fetch_one(A, lda, AI, i0, I, rbeg, va, ina);
  fetch_one(B, ldb, BI, j0, J, rbeg, vb, inb);
  stash_one(As[0], AI, va, ina);
  stash_one(Bs[0], BI, vb, inb);
  __syncthreads();
  int buf = 0;
  const int arow = wi * 32 + l31, brow = wj * 32 + l31;
  const int akey = tile_key(arow), bkey = tile_key(brow);
  for (int r0 = rbeg; r0 < rend; r0 += TR) {
    // (past the last step the requests read element 0 and are dropped: no run-time branch around loads — see gemm_strided_kernel)
    const int rn = (r0 + TR < rend) ? r0 + TR : rend;
    fetch_one(A, lda, AI, i0, I, rn, va, ina);
    fetch_one(B, ldb, BI, j0, J, rn, vb, inb);
    __builtin_amdgcn_sched_barrier(0);
    const float* Ab = As[buf] + arow * TR;
    const float* Bb = Bs[buf] + brow * TR;
#pragma unroll
    for (int c = 0; c < TR / 8; ++c) {
      const f32x4 a = *(const f32x4*)(Ab + (((2 * c + h) ^ akey) << 2));
      const f32x4 b = *(const f32x4*)(Bb + (((2 * c + h) ^ bkey) << 2));
#pragma unroll
      for (int m = 0; m < 4; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[m], acc, 0, 0, 0);
    }
    stash_one(As[buf ^ 1], AI, va, ina);  // nobody reads buf^1 during this step (the barrier below closed the previous one)
    stash_one(Bs[buf ^ 1], BI, vb, inb);
    __syncthreads();
    buf ^= 1;
  }
  const int j = j0 + wj * 32 + l31;
  float bj = bias ? bias[j < J ? j : 0] : 0.f;
  asm volatile("" : "+v"(bj));
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = i0 + wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
    if (i < I && j < J) {
      float v = acc[r] + bj;
      if (relu) v = v > 0.f ? v : 0.f;
      if (res) v = __fadd_rn(v, res[(long long)i * ldr + j]);
      C[(long long)i * J + j] = v;
    }
  }
}

// C[e] = sum_z partial[z][e] (+ bias[e % J]) (+ ReLU), fixed order
__global__ void splitk_combine_kernel(const float* __restrict__ partial, int nsplit, const float* __restrict__ bias,
                                      float* __restrict__ C, int I, int J, int relu, const float* __restrict__ res, long long ldr) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= I * J) return;
  float s = 0.f;
  for (int z = 0; z < nsplit; ++z) s += partial[(size_t)z * I * J + e];
  if (bias) s += bias[e % J];
  if (relu) s = s > 0.f ? s : 0.f;
  if (res) s = __fadd_rn(s, res[(long long)(e / J) * ldr + e % J]);
  C[e] = s;
}

// db[n] = sum_m dy[m][n]: block = 64 columns x 4 row slices, 4 independent accumulators per thread, fixed order
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ dy, float* __restrict__ db, int M, int N) {
  const int n = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
  const int per = (M + 3) / 4;
  const int m0 = part * per, m1 = (m0 + per < M) ? m0 + per : M;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (n < N) {
    int m = m0;
    for (; m + 3 < m1; m += 4) {
      s0 += dy[(long long)m * N + n];
      s1 += dy[(long long)(m + 1) * N + n];
      s2 += dy[(long long)(m + 2) * N + n];
      s3 += dy[(long long)(m + 3) * N + n];
    }
    for (; m < m1; ++m) s0 += dy[(long long)m * N + n];
  }
  __shared__ float sm[4][64];
  sm[part][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (part == 0 && n < N) db[n] = (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
}

__global__ void relu_bwd_kernel(const float* __restrict__ y, float* __restrict__ dy, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    if (!(y[i] > 0.f)) dy[i] = 0.f;
}

// Few output tiles + a long reduction (e.g. Linear(2304, 200) at batch 256: 16 tiles, R = 2304) would leave most CUs
// idle: split the reduction over blockIdx.z into `ws` and combine in a fixed order (deterministic, no atomics).
static int choose_splits(int I, int J, int R, size_t ws_bytes, bool vec) {
  const int tiles = ((I + 63) / 64) * ((J + 63) / 64);
  if (tiles >= 96 || R < 512) return 1;
  // (the vector kernel: two workgroups per CU and at least two of its 64-wide steps per split)
  int s = (vec ? 512 : 256) / tiles;
  if (s > R / 128) s = R / 128;
  if (s > 16) s = 16;
  while (s > 1 && (size_t)s * I * J * sizeof(float) > ws_bytes) --s;
  return s < 1 ? 1 : s;
}

// gemm_vec_kernel takes the call when every 16-byte access it makes is aligned and whole
static bool vec_ok(const float* P, long long s_row, long long s_r, int nrow, int R, bool* along_row, long long* ld) {
  if (((uintptr_t)P & 15) != 0 || R % 4 != 0) return false;
  if (s_r == 1 && s_row % 4 == 0) { *along_row = false; *ld = s_row; return true; }
  if (s_row == 1 && s_r % 4 == 0 && nrow % 4 == 0) { *along_row = true; *ld = s_r; return true; }
  return false;
}

static int launch(const float* A, long long sai, long long sar, const float* B, long long sbr, long long sbj,
                  const float* bias, float* C, int I, int J, int R, int relu, void* ws, size_t ws_bytes, hipStream_t st,
                  const float* res = nullptr, long long ldr = 0) {
  SRLZ_REQUIRE(I > 0 && J > 0 && R > 0, SRLZ_ERR_BAD_DESC, "linear: empty GEMM %dx%dx%d", I, J, R);
  bool ai = false, bi = false;
  long long lda = 0, ldb = 0;
  const bool vec = vec_ok(A, sai, sar, I, R, &ai, &lda) && vec_ok(B, sbj, sbr, J, R, &bi, &ldb);
  const int step = vec ? TR : BR;
  int splits = ws ? choose_splits(I, J, R, ws_bytes, vec) : 1;
  int rchunk = R, nz = 1;
  if (splits > 1) {
    rchunk = (R + splits - 1) / splits;
    rchunk = (rchunk + step - 1) / step * step;
    nz = (R + rchunk - 1) / rchunk;
  }
  const dim3 grid((J + 63) / 64, (I + 63) / 64, nz);
  const float* kbias = nz > 1 ? nullptr : bias;
  float* kC = nz > 1 ? (float*)ws : C;
  const int krelu = nz > 1 ? 0 : relu;
  const float* kres = nz > 1 ? nullptr : res;
  if (vec) {
#define SRLZ_VEC_LAUNCH(AIV, BIV) \
    hipLaunchKernelGGL((gemm_vec_kernel<AIV, BIV>), grid, dim3(256), 0, st, A, lda, B, ldb, kbias, kC, I, J, R, krelu, rchunk, kres, ldr)
    if (ai && bi) SRLZ_VEC_LAUNCH(true, true);
    else if (ai) SRLZ_VEC_LAUNCH(true, false);
    else if (bi) SRLZ_VEC_LAUNCH(false, true);
    else SRLZ_VEC_LAUNCH(false, false);
#undef SRLZ_VEC_LAUNCH
  } else {
    hipLaunchKernelGGL(gemm_strided_kernel, grid, dim3(256), 0, st, A, sai, sar, B, sbr, sbj, kbias, kC, I, J, R, krelu, rchunk, kres, ldr);
  }
  SRLZ_LAUNCHED();
  if (nz > 1) {
    hipLaunchKernelGGL(splitk_combine_kernel, dim3((I * J + 255) / 256), dim3(256), 0, st, (const float*)ws, nz, bias, C, I, J,
                       relu, res, ldr);
    SRLZ_LAUNCHED();
  }
  return 0;
}

}  // namespace

extern "C" size_t srlz_linear_workspace(int M, int N, int K) {
  // enough for 16 split-K partial copies of the largest of the three results (y[M,N], dx[M,K], dw[N,K])
  size_t a = (size_t)M * N, b = (size_t)M * K, c = (size_t)N * K;
  size_t m = a > b ? a : b;
  if (c > m) m = c;
  return 16 * m * sizeof(float);
}

extern "C" int srlz_linear_fwd(const float* x, const float* w, const float* b, float* y, int M, int N, int K, int relu,
                               void* ws, size_t ws_bytes, srlz_stream_t stream) {
  SRLZ_REQUIRE(x && w && y, SRLZ_ERR_NULL, "linear_fwd: null pointer");
  return launch(x, K, 1, w, 1, K, b, y, M, N, K, relu, ws, ws_bytes, as_stream(stream));
}

extern "C" int srlz_linear_fwd_res(const float* x, const float* w, const float* b, const float* res, float* y, int M, int N, int K,
                                   int relu, void* ws, size_t ws_bytes, srlz_stream_t stream) {
  SRLZ_REQUIRE(x && w && y && res, SRLZ_ERR_NULL, "linear_fwd_res: null pointer");
  return launch(x, K, 1, w, 1, K, b, y, M, N, K, relu, ws, ws_bytes, as_stream(stream), res, N);
}

extern "C" int srlz_linear_bwd_data_res(const float* dy, const float* w, const float* res, float* dx, int M, int N, int K, int Kout,
                                        void* ws, size_t ws_bytes, srlz_stream_t stream) {
  SRLZ_REQUIRE(dy && w && dx && res, SRLZ_ERR_NULL, "linear_bwd_data_res: null pointer");
  SRLZ_REQUIRE(Kout >= 1 && Kout <= K, SRLZ_ERR_BAD_DESC, "linear_bwd_data_res: %d of %d input columns", Kout, K);
  // the first Kout columns of dy . W, plus the residual branch's gradient (res: [M, Kout])
  return launch(dy, N, 1, w, K, 1, nullptr, dx, M, Kout, N, 0, ws, ws_bytes, as_stream(stream), res, Kout);
}

extern "C" int srlz_linear_bwd_data(const float* dy, const float* w, float* dx, int M, int N, int K, void* ws,
                                    size_t ws_bytes, srlz_stream_t stream) {
  SRLZ_REQUIRE(dy && w && dx, SRLZ_ERR_NULL, "linear_bwd_data: null pointer");
  return launch(dy, N, 1, w, K, 1, nullptr, dx, M, K, N, 0, ws, ws_bytes, as_stream(stream));
}

extern "C" int srlz_linear_bwd_weight(const float* dy, const float* x, float* dw, float* db, int M, int N, int K,
                                      void* ws, size_t ws_bytes, srlz_stream_t stream) {
  SRLZ_REQUIRE(dy && x && dw, SRLZ_ERR_NULL, "linear_bwd_weight: null pointer");
  if (int rc = launch(dy, 1, N, x, K, 1, nullptr, dw, N, K, M, 0, ws, ws_bytes, as_stream(stream))) return rc;
  if (db) {
    hipLaunchKernelGGL(colsum_kernel, dim3((N + 63) / 64), dim3(256), 0, as_stream(stream), dy, db, M, N);
    SRLZ_LAUNCHED();
  }
  return 0;
}

extern "C" int srlz_relu_bwd_inplace(const float* y, float* dy, long long n, srlz_stream_t stream) {
  SRLZ_REQUIRE(y && dy, SRLZ_ERR_NULL, "relu_bwd: null pointer");
  long long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(relu_bwd_kernel, dim3((int)blocks), dim3(256), 0, as_stream(stream), y, dy, n);
  SRLZ_LAUNCHED();
  return 0;
}
