// losses.hip — loss reductions, their gradients, the small head glue and Adam.
// Replaces the torch arithmetic behind /root/reference/losses/losses.py:102-129 (forward / inverse model losses),
// :172-214 (reconstruction / generation), :239-256 (KL), models/models.py:147-165 (reparameterisation),
// models/models.py:229-237 (one-hot) and th.optim.Adam (models/learner.py:199,495).
// Reductions: fp32 per thread over a strided slice, fp64 from the wave level up, fixed order -> deterministic.
#include "common.h"

namespace {

constexpr int RED_BLOCKS = 1024;

template <int OP>  // 0: (a-b)^2 ; 1: -0.5*(1 + b - a^2 - exp(b))  (a = mu, b = logvar)
__global__ __launch_bounds__(256) void reduce_partial(const float* __restrict__ a, const float* __restrict__ b, long long n,
                                                     double* __restrict__ partial) {
  // blockIdx.y = group (independent reductions over consecutive slices of n elements, e.g. the two frames of a step)
  a += (size_t)blockIdx.y * n;
  b += (size_t)blockIdx.y * n;
  partial += (size_t)blockIdx.y * gridDim.x;
  double acc = 0.0;
  const long long n4 = n >> 2;
  const long long stride = (long long)gridDim.x * blockDim.x;
  float local = 0.f;
  int cnt = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const f32x4 x = *(const f32x4*)(a + i * 4);
    const f32x4 y = *(const f32x4*)(b + i * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (OP == 0) { const float d = x[j] - y[j]; local += d * d; }
      else local += -0.5f * (1.f + y[j] - x[j] * x[j] - expf(y[j]));
    }
    if (++cnt == 8) { acc += (double)local; local = 0.f; cnt = 0; }
  }
  // tail
  for (long long i = n4 * 4 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    if (OP == 0) { const float d = a[i] - b[i]; local += d * d; }
    else local += -0.5f * (1.f + b[i] - a[i] * a[i] - expf(b[i]));
  }
  acc += (double)local;
  acc = wave_sum_d(acc);
  __shared__ double sm[4];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

__global__ void reduce_final(const double* __restrict__ partial, int nb, float* __restrict__ out, float div) {
  // one wave per group (blockIdx.x); div: the fp32 division of reconstructionLoss (losses.py:181: sum / numel) behind the sum's own
  // rounding to fp32 — 1 elsewhere (x / 1 is x)
  partial += (size_t)blockIdx.x * nb;
  double s = 0.0;
  for (int i = threadIdx.x; i < nb; i += 64) s += partial[i];
  s = wave_sum_d(s);
  if (threadIdx.x == 0) out[blockIdx.x] = __fdiv_rn((float)s, div);
}

__global__ __launch_bounds__(256) void sqdiff_grad_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         const float* __restrict__ coef_dev, float coef,
                                                         float* __restrict__ da, long long n, int coef_stride, float div) {
  // blockIdx.y = group: its slice of n elements and its upstream coefficient coef_dev[group * coef_stride] (stride 0: one
  // shared scalar); g = (upstream / div) * coef in exactly that order — what autograd computes for sum(..)/numel
  a += (size_t)blockIdx.y * n; b += (size_t)blockIdx.y * n; da += (size_t)blockIdx.y * n;
  const float g = ((coef_dev ? coef_dev[blockIdx.y * coef_stride] : 1.f) / div) * coef;
  const long long n4 = n >> 2;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const f32x4 x = *(const f32x4*)(a + i * 4);
    const f32x4 y = *(const f32x4*)(b + i * 4);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = g * (x[j] - y[j]);
    *(f32x4*)(da + i * 4) = o;
  }
  for (long long i = n4 * 4 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) da[i] = g * (a[i] - b[i]);
}

__global__ void kl_grad_kernel(const float* __restrict__ mu, const float* __restrict__ lv, const float* __restrict__ coef_dev,
                               float coef, float* __restrict__ dmu, float* __restrict__ dlv, long long n) {
  const float g = (coef_dev ? coef_dev[0] : 1.f) * coef;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    dmu[i] += g * mu[i];
    dlv[i] += g * 0.5f * (expf(lv[i]) - 1.f);
  }
}

__global__ void reparam_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ lv, const float* __restrict__ eps,
                                   float* __restrict__ z, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    z[i] = eps[i] * expf(0.5f * lv[i]) + mu[i];
}

__global__ void reparam_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ lv, const float* __restrict__ eps,
                                   float* __restrict__ dmu, float* __restrict__ dlv, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    dmu[i] = dz[i];
    dlv[i] = dz[i] * eps[i] * 0.5f * expf(0.5f * lv[i]);
  }
}

// one block; thread per sample
__global__ void cross_entropy_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target, int B, int A,
                                     float* __restrict__ out, float* __restrict__ dlogits) {
  double acc = 0.0;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    const float* l = logits + (size_t)b * A;
    float mx = l[0];
    for (int j = 1; j < A; ++j) mx = l[j] > mx ? l[j] : mx;
    float se = 0.f;
    for (int j = 0; j < A; ++j) se += expf(l[j] - mx);
    const float lse = mx + logf(se);
    const int t = (int)target[b];
    acc += (double)(lse - l[t]);
    if (dlogits) {
      const float invB = 1.f / (float)B;
      for (int j = 0; j < A; ++j) dlogits[(size_t)b * A + j] = (expf(l[j] - lse) - (j == t ? 1.f : 0.f)) * invB;
    }
  }
  acc = wave_sum_d(acc);
  __shared__ double sm[4];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (float)((sm[0] + sm[1] + sm[2] + sm[3]) / (double)B);
}

__global__ void concat_onehot_kernel(const float* __restrict__ s, const int64_t* __restrict__ a, float* __restrict__ cat,
                                     int B, int S, int A) {
  const int total = B * (S + A);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int b = i / (S + A), j = i - b * (S + A);
    cat[i] = (j < S) ? s[(size_t)b * S + j] : ((int)a[b] == j - S ? 1.f : 0.f);
  }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                  float* __restrict__ v, long long n, float step_size, float b1, float b2,
                                                  float eps, float omb1, float omb2, float bc2_sqrt, float grad_scale) {
  // torch.optim.Adam (no amsgrad, no weight decay): m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ;
  // p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
  // Every scalar is evaluated in DOUBLE on the host and rounded once, as torch does with its Python floats
  // (step_size = lr / (1 - b1^t), omb = 1 - beta): 1.f - 0.999f would be 0.00099998713, 1.3e-5 away from torch's 0.001f.
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float gi = g[i] * grad_scale;
    const float mi = b1 * m[i] + omb1 * gi;
    const float vi = b2 * v[i] + omb2 * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] -= step_size * (mi / denom);
  }
}

// y[r][c] = x[r][c] for lo <= c < hi, 0 elsewhere (SRLModulesSplit.detachSplit, models/modules.py:191-236: the state is
// rebuilt from zero blocks and one kept slice, i.e. a column mask; its backward is the same mask on the gradient)
__global__ void mask_columns_kernel(const float* __restrict__ x, float* __restrict__ y, int rows, int cols, int lo, int hi) {
  const int total = rows * cols;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int c = i % cols;
    y[i] = (c >= lo && c < hi) ? x[i] : 0.f;
  }
}

// Regularisers over a LIST of parameter tensors (losses/losses.py:132-155): one workgroup per tensor, fp64 accumulation in
// a fixed order.  mode 0: sum |p| ; mode 1: sqrt(sum p^2).  ptrs/lens are device arrays of nseg entries.
__global__ __launch_bounds__(256) void param_norms_kernel(const float* const* __restrict__ ptrs, const long long* __restrict__ lens,
                                                         int mode, float* __restrict__ norms) {
  const float* p = ptrs[blockIdx.x];
  const long long n = lens[blockIdx.x];
  double acc = 0.0;
  for (long long base = 0; base < n; base += 256 * 8) {
    float local = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const long long i = base + j * 256 + threadIdx.x;
      if (i < n) { const float v = p[i]; local += mode == 0 ? fabsf(v) : v * v; }
    }
    acc += (double)local;
  }
  acc = wave_sum_d(acc);
  __shared__ double sm[4];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double t = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    norms[blockIdx.x] = (float)(mode == 0 ? t : sqrt(t));
  }
}

// out = scale * sum_i norms[i]   (one wave; fp64)
__global__ void param_norms_total(const float* __restrict__ norms, int nseg, float scale, float* __restrict__ out) {
  double s = 0.0;
  for (int i = threadIdx.x; i < nseg; i += 64) s += (double)norms[i];
  s = wave_sum_d(s);
  if (threadIdx.x == 0) out[0] = (float)(s * (double)scale);
}

// g_i = coef * sign(p_i)  (mode 0)   or   coef * p_i / ||p_i||  (mode 1), coef = coef_dev[0] * scale
__global__ __launch_bounds__(256) void param_norms_grad_kernel(const float* const* __restrict__ ptrs,
                                                              float* const* __restrict__ gptrs,
                                                              const long long* __restrict__ lens, int mode,
                                                              const float* __restrict__ norms,
                                                              const float* __restrict__ coef_dev, float scale) {
  const int seg = blockIdx.y;
  const float* p = ptrs[seg];
  float* g = gptrs[seg];
  const long long n = lens[seg];
  const float coef = (coef_dev ? coef_dev[0] : 1.f) * scale;
  const float inv = (mode == 1 && norms[seg] > 0.f) ? 1.f / norms[seg] : 0.f;  // torch: d||p||/dp = 0 at p = 0
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float v = p[i];
    g[i] = mode == 0 ? coef * (v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f)) : coef * (v * inv);
  }
}

// g[i] = ((g[i] + s[0][i]) + s[1][i]) + ... ; s[k][i] = 0      (k < nstage, stage k at s + k*n)
// The per-parameter gradient contributions of one backward pass are written by the weight-gradient kernels straight into
// staging copies of the flat bucket (first contribution of a parameter -> stage 0, second -> stage 1, ...); this one
// launch folds them into the bucket and clears the stages, replacing autograd's per-parameter `grad += new` kernels.
__global__ __launch_bounds__(256) void fold_grads_kernel(float* __restrict__ g, float* __restrict__ s, long long n, int nstage) {
  const long long n4 = n >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    f32x4 v = *(f32x4*)(g + 4 * i);
    for (int k = 0; k < nstage; ++k) {
      f32x4* p = (f32x4*)(s + (size_t)k * n + 4 * i);
      const f32x4 t = *p;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += t[e];
      *p = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    *(f32x4*)(g + 4 * i) = v;
  }
}

// Device-side step counter for hipGraph replays (a captured kernel argument cannot change from step to step):
// step_dev[0] += 1, then the two step scalars of torch.optim.Adam for that step -> bc[0] = lr / (1 - b1^t), bc[1] = sqrt(1 - b2^t)
__global__ void adam_tick_kernel(int* __restrict__ step_dev, float* __restrict__ bc, double lr, double b1, double b2) {
  const int t = step_dev[0] + 1;
  step_dev[0] = t;
  bc[0] = (float)(lr / (1.0 - pow(b1, (double)t)));  // step_size
  bc[1] = (float)sqrt(1.0 - pow(b2, (double)t));
}

__global__ __launch_bounds__(256) void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                      float* __restrict__ v, long long n, float b1, float b2,
                                                      float eps, float omb1, float omb2, const float* __restrict__ bc,
                                                      float grad_scale) {
  const float step_size = bc[0];
  const float bc2_sqrt = bc[1];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float gi = g[i] * grad_scale;
    const float mi = b1 * m[i] + omb1 * gi;
    const float vi = b2 * v[i] + omb2 * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] -= step_size * (mi / denom);
  }
}

// nn.PReLU() (one slope, init 0.25) — EmbeddingNet.fc[0], /root/reference/models/triplet.py:24
__global__ void prelu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ slope, float* __restrict__ y, long long n) {
  const float a = slope[0];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    y[i] = x[i] > 0.f ? x[i] : a * x[i];
}

// dx = dy * (x > 0 ? 1 : a);  dslope = sum dy * x * [x <= 0]   (one block: the tensor is [B, 128]; fp64 sum, fixed order)
__global__ __launch_bounds__(256) void prelu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ slope,
                                                       const float* __restrict__ dy, float* __restrict__ dx,
                                                       float* __restrict__ dslope, long long n) {
  const float a = slope[0];
  double acc = 0.0;
  for (long long i = threadIdx.x; i < n; i += blockDim.x) {
    const float xi = x[i], g = dy[i];
    dx[i] = xi > 0.f ? g : a * g;
    if (!(xi > 0.f)) acc += (double)(g * xi);
  }
  acc = wave_sum_d(acc);
  __shared__ double sm[4];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) dslope[0] = (float)(sm[0] + sm[1] + sm[2] + sm[3]);
}

// tripletLoss, /root/reference/losses/losses.py:360-376: mean_b relu(|s-p|^2 - |s-n|^2 + alpha).  One block; hinge[b] keeps
// which rows are active for the backward.
__global__ __launch_bounds__(256) void triplet_fwd_kernel(const float* __restrict__ s, const float* __restrict__ p,
                                                         const float* __restrict__ ng, int B, int S, float alpha,
                                                         float* __restrict__ out, float* __restrict__ hinge) {
  double acc = 0.0;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    float dp = 0.f, dn = 0.f;
    for (int j = 0; j < S; ++j) {
      const float sv = s[(size_t)b * S + j];
      const float a = sv - p[(size_t)b * S + j], c = sv - ng[(size_t)b * S + j];
      dp += a * a; dn += c * c;
    }
    const float l = dp - dn + alpha;
    hinge[b] = l > 0.f ? 1.f : 0.f;
    acc += (double)(l > 0.f ? l : 0.f);
  }
  acc = wave_sum_d(acc);
  __shared__ double sm[4];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (float)((sm[0] + sm[1] + sm[2] + sm[3]) / (double)B);
}

// d/ds = g/B * hinge * 2 (n - p) ; d/dp = -g/B * hinge * 2 (s - p) ; d/dn = g/B * hinge * 2 (s - n)
__global__ void triplet_bwd_kernel(const float* __restrict__ s, const float* __restrict__ p, const float* __restrict__ ng,
                                   const float* __restrict__ hinge, const float* __restrict__ g, int B, int S,
                                   float* __restrict__ ds, float* __restrict__ dp, float* __restrict__ dn) {
  const float c = g[0] / (float)B;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * S; i += gridDim.x * blockDim.x) {
    const float hb = hinge[i / S] * c;
    const float sv = s[i], pv = p[i], nv = ng[i];
    ds[i] = hb * 2.f * (nv - pv);
    dp[i] = -hb * 2.f * (sv - pv);
    dn[i] = hb * 2.f * (sv - nv);
  }
}

static int blocks_for(long long n, int cap) {
  long long b = (n + 255) / 256;
  if (b > cap) b = cap;
  return b < 1 ? 1 : (int)b;
}

constexpr int MAX_GROUPS = 8;

// `groups` independent reductions over consecutive slices of n elements each -> out[groups]; per group exactly the launch
// geometry and summation order of a single-group call (bit-identical results).
template <int OP>
static int reduce_launch(const float* a, const float* b, long long n, int groups, float* out, void* ws, size_t ws_bytes,
                         hipStream_t st, float div = 1.f) {
  SRLZ_REQUIRE(a && b && out && ws, SRLZ_ERR_NULL, "reduce: null pointer");
  SRLZ_REQUIRE(groups >= 1 && groups <= MAX_GROUPS, SRLZ_ERR_BAD_DESC, "reduce: groups = %d", groups);
  SRLZ_REQUIRE(ws_bytes >= (size_t)groups * RED_BLOCKS * sizeof(double), SRLZ_ERR_WORKSPACE, "reduce: workspace too small");
  SRLZ_REQUIRE(((uintptr_t)a & 15) == 0 && ((uintptr_t)b & 15) == 0 && (groups == 1 || (n & 3) == 0), SRLZ_ERR_BAD_DESC,
               "reduce: inputs (and every group's slice) must be 16-byte aligned");
  const int nb = blocks_for((n + 3) / 4, RED_BLOCKS);
  hipLaunchKernelGGL(reduce_partial<OP>, dim3(nb, groups), dim3(256), 0, st, a, b, n, (double*)ws);
  SRLZ_LAUNCHED();
  hipLaunchKernelGGL(reduce_final, dim3(groups), dim3(64), 0, st, (const double*)ws, nb, out, div);
  SRLZ_LAUNCHED();
  return 0;
}

// The loss of a batched pair in one go: sums[g] = group g's sum (as reduce_final gives it) and
// comb[0] = sums[0]/n + sums[1]/n (mode 1: reconstructionLoss x 2, losses.py:172-196) or sums[0] + sums[1] (mode 0:
// F.mse_loss(sum) x 2, losses.py:199-214) with the roundings of the reference's separate fp32 operations.
__global__ void pair_loss_final(const double* __restrict__ partial, int nb, float* __restrict__ sums, float* __restrict__ comb,
                                float n, int mode) {
  const int g = threadIdx.x >> 6, lane = threadIdx.x & 63;  // one wave per group
  double s = 0.0;
  for (int i = lane; i < nb; i += 64) s += partial[(size_t)g * nb + i];
  s = wave_sum_d(s);
  __shared__ float sm[2];
  if (lane == 0) { sm[g] = (float)s; sums[g] = (float)s; }
  __syncthreads();
  if (threadIdx.x == 0) comb[0] = mode ? (sm[0] / n + sm[1] / n) : (sm[0] + sm[1]);
}

// out = [a ; b] (n floats each): joins the two halves of a batched pair when they do not already sit next to each other
__global__ __launch_bounds__(256) void join2_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                   float* __restrict__ out, long long n) {
  const float* __restrict__ src = blockIdx.y ? b : a;
  float* __restrict__ dst = out + (size_t)blockIdx.y * n;
  const long long n4 = n >> 2, stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
    *(f32x4*)(dst + i * 4) = *(const f32x4*)(src + i * 4);
  for (long long i = n4 * 4 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

// cat([a, b], dim 1) of two row-major matrices with the same number of rows (inverse / reward heads: [state ; next_state],
// forward_inverse.py:62,78) and its backward, the split of the gradient's columns
__global__ void cat_cols_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int rows, int ca, int cb) {
  const long long n = (long long)rows * (ca + cb);
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(e / (ca + cb)), c = (int)(e - (long long)r * (ca + cb));
    out[e] = c < ca ? a[(long long)r * ca + c] : b[(long long)r * cb + (c - ca)];
  }
}
__global__ void split_cols_kernel(const float* __restrict__ in, float* __restrict__ a, float* __restrict__ b, int rows, int ca, int cb) {
  const long long n = (long long)rows * (ca + cb);
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(e / (ca + cb)), c = (int)(e - (long long)r * (ca + cb));
    if (c < ca) { if (a) a[(long long)r * ca + c] = in[e]; }
    else if (b) b[(long long)r * cb + (c - ca)] = in[e];
  }
}
// out = ((g0 + g1) + g2) + g3 over the non-null terms, left to right in fp32: what autograd's accumulation of the gradients of a
// tensor with several consumers computes (one add per extra consumer), as ONE launch in a fixed order
struct TermList { const float* p[4]; int n; };
__global__ void sum_terms_kernel(const TermList L, float* __restrict__ out, long long n) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
    float t = L.p[0][e];
    for (int i = 1; i < L.n; ++i) t = __fadd_rn(t, L.p[i][e]);
    out[e] = t;
  }
}
// total = (((0 + w0*l0) + w1*l1) + ...) in fp32 with separately rounded products — exactly Python's sum([w_i * l_i]) over 0-dim
// fp32 tensors (reference losses/losses.py:55-56) — plus the step's scalars into the gradient bucket's tail: [total, l_0, l_1, ...]
struct ScalarList {
  const float* p[SRLZ_MAX_LOSS_TERMS];
  float w[SRLZ_MAX_LOSS_TERMS];
  int n;
};
__global__ void weighted_total_kernel(const ScalarList L, float* __restrict__ total, float* __restrict__ tail) {
  if (threadIdx.x != 0) return;
  float t = 0.f;
  for (int i = 0; i < L.n; ++i) {
    const float l = *L.p[i];
    t = __fadd_rn(t, __fmul_rn(L.w[i], l));
    if (tail) tail[1 + i] = l;
  }
  *total = t;
  if (tail) tail[0] = t;
}
// g[i] = dout * w_i (autograd of the same expression)
__global__ void weighted_total_bwd_kernel(const ScalarList L, const float* __restrict__ dout, float* __restrict__ g) {
  const int i = threadIdx.x;
  if (i < L.n) g[i] = __fmul_rn(*dout, L.w[i]);
}

}  // namespace

extern "C" int srlz_weighted_total(const float* const* scalars, const float* weights, int n, float* total, float* tail,
                                   srlz_stream_t stream) {
  SRLZ_REQUIRE(scalars && weights && total, SRLZ_ERR_NULL, "weighted_total: null pointer");
  SRLZ_REQUIRE(n >= 1 && n <= SRLZ_MAX_LOSS_TERMS, SRLZ_ERR_BAD_DESC, "weighted_total: %d terms (1..%d)", n, SRLZ_MAX_LOSS_TERMS);
  ScalarList L;
  for (int i = 0; i < n; ++i) {
    SRLZ_REQUIRE(scalars[i], SRLZ_ERR_NULL, "weighted_total: null term %d", i);
    L.p[i] = scalars[i]; L.w[i] = weights[i];
  }
  L.n = n;
  hipLaunchKernelGGL(weighted_total_kernel, dim3(1), dim3(64), 0, as_stream(stream), L, total, tail);
  SRLZ_LAUNCHED();
  return 0;
}

extern "C" int srlz_weighted_total_bwd(const float* dout, const float* weights, int n, float* g, srlz_stream_t stream) {
  SRLZ_REQUIRE(dout && weights && g, SRLZ_ERR_NULL, "weighted_total_bwd: null pointer");
  SRLZ_REQUIRE(n >= 1 && n <= SRLZ_MAX_LOSS_TERMS, SRLZ_ERR_BAD_DESC, "weighted_total_bwd: %d terms (1..%d)", n, SRLZ_MAX_LOSS_TERMS);
  ScalarList L;
  for (int i = 0; i < n; ++i) { L.p[i] = nullptr; L.w[i] = weights[i]; }
  L.n = n;
  hipLaunchKernelGGL(weighted_total_bwd_kernel, dim3(1), dim3(64), 0, as_stream(stream), L, dout, g);
  SRLZ_LAUNCHED();
  return 0;
}

extern "C" int srlz_cat_cols(const float* a, const float* b, float* out, int rows, int ca, int cb, srlz_stream_t stream) {
  SRLZ_REQUIRE(a && b && out, SRLZ_ERR_NULL, "cat_cols: null pointer");
  SRLZ_REQUIRE(rows > 0 && ca > 0 && cb > 0, SRLZ_ERR_BAD_DESC, "cat_cols: %d rows of %d + %d columns", rows, ca, cb);
  hipLaunchKernelGGL(cat_cols_kernel, dim3(blocks_for((long long)rows * (ca + cb), 1024)), dim3(256), 0, as_stream(stream), a, b, out,
                     rows, ca, cb);
  SRLZ_LAUNCHED();
  return 0;
}

extern "C" int srlz_split_cols(const float* in, float* a, float* b, int rows, int ca, int cb, srlz_stream_t stream) {
  SRLZ_REQUIRE(in && (a || b), SRLZ_ERR_NULL, "split_cols: null pointer");
  SRLZ_REQUIRE(rows > 0 && ca > 0 && cb > 0, SRLZ_ERR_BAD_DESC, "split_cols: %d rows of %d + %d columns", rows, ca, cb);
  hipLaunchKernelGGL(split_cols_kernel, dim3(blocks_for((long long)rows * (ca + cb), 1024)), dim3(256), 0, as_stream(stream), in, a, b,
                     rows, ca, cb);
  SRLZ_LAUNCHED();
  return 0;
}

extern "C" int srlz_sum_terms(const float* const* terms, int nterms, float* out, long long n, srlz_stream_t stream) {
  SRLZ_REQUIRE(terms && out, SRLZ_ERR_NULL, "sum_terms: null pointer");
  SRLZ_REQUIRE(nterms >= 1 && nterms <= 4 && n > 0, SRLZ_ERR_BAD_DESC, "sum_terms: %d terms (1..4) of %lld elements", nterms, n);
  TermList L;
  for (int i = 0; i < 4; ++i) L.p[i] = nullptr;
  for (int i = 0; i < nterms; ++i) {
    SRLZ_REQUIRE(terms[i], SRLZ_ERR_NULL, "sum_terms: null term %d", i);
    L.p[i] = terms[i];
  }
  L.n = nterms;
  hipLaunchKernelGGL(sum_terms_kernel, dim3(blocks_for(n, 1024)), dim3(256), 0, as_stream(stream), L, out, n);
  SRLZ_LAUNCHED();
  return 0;
}

extern "C" size_t srlz_reduce_workspace(long long n) {
  (void)n;
  return (size_t)MAX_GROUPS * RED_BLOCKS * sizeof(double);
}

extern "C" int srlz_sqdiff_sum(const float* a, const float* b, long long n, float* out, void* ws, size_t ws_bytes,
                               srlz_stream_t stream) {
  return reduce_launch<0>(a, b, n, 1, out, ws, ws_bytes, as_stream(stream));
}

extern "C" int srlz_sqdiff_mean(const float* a, const float* b, long long n, float div, float* out, void* ws, size_t ws_bytes,
                                srlz_stream_t stream) {
  SRLZ_REQUIRE(div != 0.f, SRLZ_ERR_BAD_DESC, "sqdiff_mean: division by zero");
  return reduce_launch<0>(a, b, n, 1, out, ws, ws_bytes, as_stream(stream), div);
}

extern "C" int srlz_sqdiff_sum_groups(const float* a, const float* b, long long n_per_group, int groups, float* out, void* ws,
                                      size_t ws_bytes, srlz_stream_t stream) {
  return reduce_launch<0>(a, b, n_per_group, groups, out, ws, ws_bytes, as_stream(stream));
}

extern "C" int srlz_sqdiff_grad(const float* a, const float* b, const float* coef_dev, float coef, float* da, long long n,
                                srlz_stream_t stream) {
  SRLZ_REQUIRE(a && b && da, SRLZ_ERR_NULL, "sqdiff_grad: null pointer");
  hipLaunchKernelGGL(sqdiff_grad_kernel, dim3(blocks_for((n + 3) / 4, 8192)), dim3(256), 0, as_stream(stream), a, b, coef_dev,
                     coef, da, n, 0, 1.0f);
  SRLZ_LAUNCHED();
  return 0;
}

extern "C" int srlz_sqdiff_grad_groups(const float* a, const float* b, const float* coef_dev, int coef_stride, float div,
                                       float coef, float* da, long long n_per_group, int groups, srlz_stream_t stream) {
  SRLZ_REQUIRE(a && b && da && coef_dev, SRLZ_ERR_NULL, "sqdiff_grad_groups: null pointer");
  SRLZ_REQUIRE(groups >= 1 && groups <= MAX_GROUPS && (groups == 1 || (n_per_group & 3) == 0) && div != 0.f &&
               (coef_stride == 0 || coef_stride == 1), SRLZ_ERR_BAD_DESC,
               "sqdiff_grad_groups: groups = %d, %lld elements each", groups, n_per_group);
  hipLaunchKernelGGL(sqdiff_grad_kernel, dim3(blocks_for((n_per_group + 3) / 4, 8192 / groups), groups), dim3(256), 0,
                     as_stream(stream), a, b, coef_dev, coef, da, n_per_group, coef_stride, div);
  SRLZ_LAUNCHED();
  return 0;
}

extern "C" int srlz_sqdiff_pair_loss(const float* a, const float* b, long long n_per_group, int mean, float* sums, float* comb,
                                     void* ws, size_t ws_bytes, srlz_stream_t stream) {
  SRLZ_REQUIRE(a && b && sums && comb && ws, SRLZ_ERR_NULL, "sqdiff_pair_loss: null pointer");
  SRLZ_REQUIRE(ws_bytes >= 2 * RED_BLOCKS * sizeof(double), SRLZ_ERR_WORKSPACE, "sqdiff_pair_loss: workspace too small");
  SRLZ_REQUIRE(((uintptr_t)a & 15) == 0 && ((uintptr_t)b & 15) == 0 && (n_per_group & 3) == 0 && n_per_group > 0, SRLZ_ERR_BAD_DESC,
               "sqdiff_pair_loss: halves must be 16-byte aligned multiples of 4 floats");
  hipStream_t st = as_stream(stream);
  const int nb = blocks_for((n_per_group + 3) / 4, RED_BLOCKS);  // per group the geometry of a single-group srlz_sqdiff_sum
  hipLaunchKernelGGL(reduce_partial<0>, dim3(nb, 2), dim3(256), 0, st, a, b, n_per_group, (double*)ws);
  SRLZ_LAUNCHED();
  hipLaunchKernelGGL(pair_loss_final, dim3(1), dim3(128), 0, st, (const double*)ws, nb, sums, comb, (float)n_per_group, mean);
  SRLZ_LAUNCHED();
  return 0;
}

// Second stage of a pair loss whose per-workgroup fp64 partials were written by another kernel (convT_out_kernel<true>: the
// reconstruction / generation loss taken in the last ConvTranspose's epilogue): partial[2][nb] -> sums[2], comb — the same
// fixed-order final sum and the same fp32 roundings as srlz_sqdiff_pair_loss.
extern "C" int srlz_pair_loss_finalize(const double* partial, int nb, long long n_per_group, int mean, float* sums, float* comb,
                                       srlz_stream_t stream) {
  SRLZ_REQUIRE(partial && sums && comb, SRLZ_ERR_NULL, "pair_loss_finalize: null pointer");
  SRLZ_REQUIRE(nb > 0 && n_per_group > 0, SRLZ_ERR_BAD_DESC, "pair_loss_finalize: nb = %d, n = %lld", nb, n_per_group);
  hipLaunchKernelGGL(pair_loss_final, dim3(1), dim3(128), 0, as_stream(stream), partial, nb, sums, comb, (float)n_per_group, mean);
  SRLZ_LAUNCHED();
  return 0;
}

// out = ((gain_dev[0] / div) * coef) * x — the gradient of a fused pair loss materialised from the stored error (only where the
// consumer cannot apply the factor itself: the un-fused ConvTranspose-5 backward of C = 6 / 9 images); in place when out == x.
namespace {
__global__ __launch_bounds__(256) void scale_by_scalar_kernel(const float* __restrict__ x, const float* __restrict__ gain_dev, float div,
                                                             float coef, float* __restrict__ out, long long n) {
  const float g = (gain_dev[0] / div) * coef;
  const long long n4 = n >> 2;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 v = *(const f32x4*)(x + i * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] *= g;
    *(f32x4*)(out + i * 4) = v;
  }
  for (long long i = n4 * 4 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = g * x[i];
}
}  // namespace

extern "C" int srlz_scale_by_scalar(const float* x, const float* gain_dev, float div, float coef, float* out, long long n,
                                    srlz_stream_t stream) {
  SRLZ_REQUIRE(x && gain_dev && out, SRLZ_ERR_NULL, "scale_by_scalar: null pointer");
  SRLZ_REQUIRE(div != 0.f && ((uintptr_t)x & 15) == 0 && ((uintptr_t)out & 15) == 0, SRLZ_ERR_BAD_DESC,
               "scale_by_scalar: zero divisor or unaligned buffers");
  hipLaunchKernelGGL(scale_by_scalar_kernel, dim3(blocks_for((n + 3) / 4, 8192)), dim3(256), 0, as_stream(stream), x, gain_dev, div,
                     coef, out, n);
  SRLZ_LAUNCHED();
  return 0;
}

extern "C" int srlz_join2(const float* a, const float* b, float* out, long long n_each, srlz_stream_t stream) {
  SRLZ_REQUIRE(a && b && out, SRLZ_ERR_NULL, "join2: null pointer");
  SRLZ_REQUIRE(n_each > 0 && (((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) == 0 && (n_each & 3) == 0, SRLZ_ERR_BAD_DESC,
               "join2: halves must be 16-byte aligned multiples of 4 floats (%lld)", n_each);
  hipLaunchKernelGGL(join2_kernel, dim3(blocks_for((n_each + 3) / 4, 4096), 2), dim3(256), 0, as_stream(stream), a, b, out, n_each);
  SRLZ_LAUNCHED();
  return 0;
}

extern "C" int srlz_kl_sum(const float* mu, const float* logvar, long long n, float* out, void* ws, size_t ws_bytes,
                           srlz_stream_t stream) {
  return reduce_launch<1>(mu, logvar, n, 1, out, ws, ws_bytes, as_stream(stream));
}

extern "C" int srlz_kl_grad(const float* mu, const float* logvar, const float* coef_dev, float coef, float* dmu,
                            float* dlogvar, long long n, srlz_stream_t stream) {
  SRLZ_REQUIRE(mu && logvar && dmu && dlogvar, SRLZ_ERR_NULL, "kl_grad: null pointer");
  hipLaunchKernelGGL(kl_grad_kernel, dim3(blocks_for(n, 1024)), dim3(256), 0, as_stream(stream), mu, logvar, coef_dev, coef,
                     dmu, dlogvar, n);
  SRLZ_LAUNCHED();
  return 0;
}

extern "C" int srlz_reparam_fwd(const float* mu, const float* logvar, const float* eps, float* z, long long n,
                                srlz_stream_t stream) {
  SRLZ_REQUIRE(mu && logvar && eps && z, SRLZ_ERR_NULL, "reparam_fwd: null pointer");
  hipLaunchKernelGGL(reparam_fwd_kernel, dim3(blocks_for(n, 1024)), dim3(256), 0, as_stream(stream), mu, logvar, eps, z, n);
  SRLZ_LAUNCHED();
  return 0;
}

extern "C" int srlz_reparam_bwd(const float* dz, const float* logvar, const float* eps, float* dmu, float* dlogvar,
                                long long n, srlz_stream_t stream) {
  SRLZ_REQUIRE(dz && logvar && eps && dmu && dlogvar, SRLZ_ERR_NULL, "reparam_bwd: null pointer");
  hipLaunchKernelGGL(reparam_bwd_kernel, dim3(blocks_for(n, 1024)), dim3(256), 0, as_stream(stream), dz, logvar, eps, dmu,
                     dlogvar, n);
  SRLZ_LAUNCHED();
  return 0;
}

extern "C" int srlz_prelu_fwd(const float* x, const float* slope, float* y, long long n, srlz_stream_t stream) {
  SRLZ_REQUIRE(x && slope && y, SRLZ_ERR_NULL, "prelu_fwd: null pointer");
  hipLaunchKernelGGL(prelu_fwd_kernel, dim3(blocks_for(n, 1024)), dim3(256), 0, as_stream(stream), x, slope, y, n);
  SRLZ_LAUNCHED();
  return 0;
}

extern "C" int srlz_prelu_bwd(const float* x, const float* slope, const float* dy, float* dx, float* dslope, long long n,
                              srlz_stream_t stream) {
  SRLZ_REQUIRE(x && slope && dy && dx && dslope, SRLZ_ERR_NULL, "prelu_bwd: null pointer");
  hipLaunchKernelGGL(prelu_bwd_kernel, dim3(1), dim3(256), 0, as_stream(stream), x, slope, dy, dx, dslope, n);
  SRLZ_LAUNCHED();
  return 0;
}

extern "C" int srlz_triplet_fwd(const float* s, const float* p, const float* n, int B, int S, float alpha, float* out,
                                float* hinge, srlz_stream_t stream) {
  SRLZ_REQUIRE(s && p && n && out && hinge, SRLZ_ERR_NULL, "triplet_fwd: null pointer");
  SRLZ_REQUIRE(B > 0 && S > 0, SRLZ_ERR_BAD_DESC, "triplet_fwd: empty batch");
  hipLaunchKernelGGL(triplet_fwd_kernel, dim3(1), dim3(256), 0, as_stream(stream), s, p, n, B, S, alpha, out, hinge);
  SRLZ_LAUNCHED();
  return 0;
}

extern "C" int srlz_triplet_bwd(const float* s, const float* p, const float* n, const float* hinge, const float* g, int B, int S,
                                float* ds, float* dp, float* dn, srlz_stream_t stream) {
  SRLZ_REQUIRE(s && p && n && hinge && g && ds && dp && dn, SRLZ_ERR_NULL, "triplet_bwd: null pointer");
  hipLaunchKernelGGL(triplet_bwd_kernel, dim3(blocks_for((long long)B * S, 1024)), dim3(256), 0, as_stream(stream), s, p, n, hinge,
                     g, B, S, ds, dp, dn);
  SRLZ_LAUNCHED();
  return 0;
}

extern "C" int srlz_cross_entropy(const float* logits, const int64_t* target, int B, int A, float* out, float* dlogits,
                                  srlz_stream_t stream) {
  SRLZ_REQUIRE(logits && target && out, SRLZ_ERR_NULL, "cross_entropy: null pointer");
  SRLZ_REQUIRE(B > 0 && A > 0, SRLZ_ERR_BAD_DESC, "cross_entropy: empty batch");
  hipLaunchKernelGGL(cross_entropy_kernel, dim3(1), dim3(256), 0, as_stream(stream), logits, target, B, A, out, dlogits);
  SRLZ_LAUNCHED();
  return 0;
}

extern "C" int srlz_concat_onehot(const float* s, const int64_t* a, float* cat, int B, int S, int A, srlz_stream_t stream) {
  SRLZ_REQUIRE(s && a && cat, SRLZ_ERR_NULL, "concat_onehot: null pointer");
  hipLaunchKernelGGL(concat_onehot_kernel, dim3(blocks_for((long long)B * (S + A), 1024)), dim3(256), 0, as_stream(stream), s, a,
                     cat, B, S, A);
  SRLZ_LAUNCHED();
  return 0;
}

extern "C" int srlz_mask_columns(const float* x, float* y, int rows, int cols, int lo, int hi, srlz_stream_t stream) {
  SRLZ_REQUIRE(x && y, SRLZ_ERR_NULL, "mask_columns: null pointer");
  SRLZ_REQUIRE(rows > 0 && cols > 0 && lo >= 0 && lo <= hi && hi <= cols, SRLZ_ERR_BAD_DESC,
               "mask_columns: bad range [%d,%d) of %d columns", lo, hi, cols);
  hipLaunchKernelGGL(mask_columns_kernel, dim3(blocks_for((long long)rows * cols, 1024)), dim3(256), 0, as_stream(stream), x, y,
                     rows, cols, lo, hi);
  SRLZ_LAUNCHED();
  return 0;
}

extern "C" int srlz_param_norms(const float* const* ptrs, const long long* lens, int nseg, int mode, float scale, float* norms,
                                float* out, srlz_stream_t stream) {
  SRLZ_REQUIRE(ptrs && lens && norms && out, SRLZ_ERR_NULL, "param_norms: null pointer");
  SRLZ_REQUIRE(nseg > 0 && (mode == 0 || mode == 1), SRLZ_ERR_BAD_DESC, "param_norms: nseg=%d mode=%d", nseg, mode);
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(param_norms_kernel, dim3(nseg), dim3(256), 0, st, ptrs, lens, mode, norms);
  SRLZ_LAUNCHED();
  hipLaunchKernelGGL(param_norms_total, dim3(1), dim3(64), 0, st, (const float*)norms, nseg, scale, out);
  SRLZ_LAUNCHED();
  return 0;
}

extern "C" int srlz_param_norms_grad(const float* const* ptrs, float* const* gptrs, const long long* lens, int nseg, int mode,
                                     const float* norms, const float* coef_dev, float scale, srlz_stream_t stream) {
  SRLZ_REQUIRE(ptrs && gptrs && lens && norms, SRLZ_ERR_NULL, "param_norms_grad: null pointer");
  SRLZ_REQUIRE(nseg > 0 && nseg <= 65535 && (mode == 0 || mode == 1), SRLZ_ERR_BAD_DESC, "param_norms_grad: nseg=%d mode=%d",
               nseg, mode);
  hipLaunchKernelGGL(param_norms_grad_kernel, dim3(64, nseg), dim3(256), 0, as_stream(stream), ptrs, gptrs, lens, mode, norms,
                     coef_dev, scale);
  SRLZ_LAUNCHED();
  return 0;
}

extern "C" int srlz_fold_grads(float* grad, float* stages, long long n, int nstage, srlz_stream_t stream) {
  SRLZ_REQUIRE(grad && stages, SRLZ_ERR_NULL, "fold_grads: null pointer");
  SRLZ_REQUIRE(n > 0 && (n & 3) == 0 && nstage >= 1 && nstage <= 8, SRLZ_ERR_BAD_DESC,
               "fold_grads: n=%lld (must be a multiple of 4) nstage=%d", n, nstage);
  hipLaunchKernelGGL(fold_grads_kernel, dim3(blocks_for(n / 4, 2048)), dim3(256), 0, as_stream(stream), grad, stages, n, nstage);
  SRLZ_LAUNCHED();
  return 0;
}

extern "C" int srlz_adam_step(float* p, const float* g, float* m, float* v, long long n, double lr, double beta1, double beta2,
                              double eps, int step, float grad_scale, srlz_stream_t stream) {
  SRLZ_REQUIRE(p && g && m && v, SRLZ_ERR_NULL, "adam_step: null pointer");
  SRLZ_REQUIRE(step >= 1, SRLZ_ERR_BAD_DESC, "adam_step: step is 1-based");
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  hipLaunchKernelGGL(adam_kernel, dim3(blocks_for(n, 2048)), dim3(256), 0, as_stream(stream), p, g, m, v, n, (float)(lr / bc1),
                     (float)beta1, (float)beta2, (float)eps, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)sqrt(bc2),
                     grad_scale);
  SRLZ_LAUNCHED();
  return 0;
}

extern "C" int srlz_adam_step_dev(float* p, const float* g, float* m, float* v, long long n, double lr, double beta1,
                                  double beta2, double eps, int* step_dev, float* bc_dev, float grad_scale,
                                  srlz_stream_t stream) {
  SRLZ_REQUIRE(p && g && m && v && step_dev && bc_dev, SRLZ_ERR_NULL, "adam_step_dev: null pointer");
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, st, step_dev, bc_dev, lr, beta1, beta2);
  SRLZ_LAUNCHED();
  hipLaunchKernelGGL(adam_dev_kernel, dim3(blocks_for(n, 2048)), dim3(256), 0, st, p, g, m, v, n, (float)beta1, (float)beta2,
                     (float)eps, (float)(1.0 - beta1), (float)(1.0 - beta2), (const float*)bc_dev, grad_scale);
  SRLZ_LAUNCHED();
  return 0;
}
