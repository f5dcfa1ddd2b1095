/*
 * oracle/srlz_oracle.c — TEST INFRASTRUCTURE, not product code.
 *
 * Plain-C restatement (reference tensor layouts: NCHW activations, [Cout,Cin,kh,kw] / [Cin,Cout,kh,kw] / [out,in]
 * weights) of every operator on srl-zoo's image-representation training hot path.  The reference reaches these
 * operators through a third-party dependency, PyTorch (pinned pytorch=0.4.1, /root/reference/environment.yml:83);
 * what is restated here is that library's published operator semantics at the reference's call sites:
 *   conv2d / conv_transpose2d .... nn.Conv2d, nn.ConvTranspose2d        models/models.py:47-83,217-226
 *   batch_norm (train / eval) .... nn.BatchNorm2d (eps 1e-5, mom 0.1)   models/models.py:50,55,60,67,71,75,79
 *   relu, max_pool2d(3,2,pad) .... nn.ReLU, nn.MaxPool2d                models/models.py:51-52,56-57,61-62
 *   linear ....................... nn.Linear                            models/autoencoders.py:94-100, vae.py:52-57
 *   sum((a-b)^2), KL, CE ......... losses/losses.py:172-181,199-214,239-256,117-129
 *   Adam ......................... th.optim.Adam                        models/learner.py:199
 * Accumulation is in double (results are rounded to float once), so this oracle sits closer to the exact value than
 * any fp32 implementation; tests/test_c_oracle.py pins it against torch and against the reference's golden vectors
 * (through oracle/c_oracle.py, which composes these ops into the auto-encoder train step).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * Build: make -C oracle   (gcc -O3 -fopenmp -shared -fPIC)
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define IDX4(n, c, h, w, C, H, W) ((((size_t)(n) * (C) + (c)) * (H) + (h)) * (W) + (w))

/* y[n,k,oh,ow] = b[k] + sum_{c,r,s} x[n,c,oh*st-pad+r,ow*st-pad+s] * w[k,c,r,s] */
void orc_conv2d_fwd(const float* x, const float* w, const float* b, float* y, int N, int C, int H, int W, int K, int R,
                    int S, int st, int pad) {
  const int OH = (H + 2 * pad - R) / st + 1, OW = (W + 2 * pad - S) / st + 1;
#pragma omp parallel for collapse(2) schedule(static)
  for (int n = 0; n < N; ++n)
    for (int k = 0; k < K; ++k)
      for (int oh = 0; oh < OH; ++oh)
        for (int ow = 0; ow < OW; ++ow) {
          double acc = b ? b[k] : 0.0;
          for (int c = 0; c < C; ++c)
            for (int r = 0; r < R; ++r) {
              const int ih = oh * st - pad + r;
              if (ih < 0 || ih >= H) continue;
              for (int s = 0; s < S; ++s) {
                const int iw = ow * st - pad + s;
                if (iw < 0 || iw >= W) continue;
                acc += (double)x[IDX4(n, c, ih, iw, C, H, W)] * (double)w[(((size_t)k * C + c) * R + r) * S + s];
              }
            }
          y[IDX4(n, k, oh, ow, K, OH, OW)] = (float)acc;
        }
}

/* dx, dw, db of the above (any of the three outputs may be NULL) */
void orc_conv2d_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int N, int C, int H,
                    int W, int K, int R, int S, int st, int pad) {
  const int OH = (H + 2 * pad - R) / st + 1, OW = (W + 2 * pad - S) / st + 1;
  if (dx) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
      for (int c = 0; c < C; ++c)
        for (int ih = 0; ih < H; ++ih)
          for (int iw = 0; iw < W; ++iw) {
            double acc = 0.0;
            for (int k = 0; k < K; ++k)
              for (int r = 0; r < R; ++r) {
                const int t = ih + pad - r;
                if (t < 0 || t % st) continue;
                const int oh = t / st;
                if (oh >= OH) continue;
                for (int s = 0; s < S; ++s) {
                  const int u = iw + pad - s;
                  if (u < 0 || u % st) continue;
                  const int ow = u / st;
                  if (ow >= OW) continue;
                  acc += (double)dy[IDX4(n, k, oh, ow, K, OH, OW)] * (double)w[(((size_t)k * C + c) * R + r) * S + s];
                }
              }
            dx[IDX4(n, c, ih, iw, C, H, W)] = (float)acc;
          }
  }
  if (dw) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < K; ++k)
      for (int c = 0; c < C; ++c)
        for (int r = 0; r < R; ++r)
          for (int s = 0; s < S; ++s) {
            double acc = 0.0;
            for (int n = 0; n < N; ++n)
              for (int oh = 0; oh < OH; ++oh) {
                const int ih = oh * st - pad + r;
                if (ih < 0 || ih >= H) continue;
                for (int ow = 0; ow < OW; ++ow) {
                  const int iw = ow * st - pad + s;
                  if (iw < 0 || iw >= W) continue;
                  acc += (double)dy[IDX4(n, k, oh, ow, K, OH, OW)] * (double)x[IDX4(n, c, ih, iw, C, H, W)];
                }
              }
            dw[(((size_t)k * C + c) * R + r) * S + s] = (float)acc;
          }
  }
  if (db) {
    for (int k = 0; k < K; ++k) {
      double acc = 0.0;
      for (int n = 0; n < N; ++n)
        for (int i = 0; i < OH * OW; ++i) acc += (double)dy[((size_t)n * K + k) * OH * OW + i];
      db[k] = (float)acc;
    }
  }
}

/* y[n,k,oh,ow] = b[k] + sum_{c,r,s: oh = ih*st - pad + r} x[n,c,ih,iw] * w[c,k,r,s]   (w: [Cin,Cout,R,S]) */
void orc_convT2d_fwd(const float* x, const float* w, const float* b, float* y, int N, int C, int H, int W, int K, int R,
                     int S, int st, int pad) {
  const int OH = (H - 1) * st - 2 * pad + R, OW = (W - 1) * st - 2 * pad + S;
#pragma omp parallel for collapse(2) schedule(static)
  for (int n = 0; n < N; ++n)
    for (int k = 0; k < K; ++k)
      for (int oh = 0; oh < OH; ++oh)
        for (int ow = 0; ow < OW; ++ow) {
          double acc = b ? b[k] : 0.0;
          for (int r = 0; r < R; ++r) {
            const int t = oh + pad - r;
            if (t < 0 || t % st) continue;
            const int ih = t / st;
            if (ih >= H) continue;
            for (int s = 0; s < S; ++s) {
              const int u = ow + pad - s;
              if (u < 0 || u % st) continue;
              const int iw = u / st;
              if (iw >= W) continue;
              for (int c = 0; c < C; ++c)
                acc += (double)x[IDX4(n, c, ih, iw, C, H, W)] * (double)w[(((size_t)c * K + k) * R + r) * S + s];
            }
          }
          y[IDX4(n, k, oh, ow, K, OH, OW)] = (float)acc;
        }
}

void orc_convT2d_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int N, int C, int H,
                     int W, int K, int R, int S, int st, int pad) {
  const int OH = (H - 1) * st - 2 * pad + R, OW = (W - 1) * st - 2 * pad + S;
  if (dx) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
      for (int c = 0; c < C; ++c)
        for (int ih = 0; ih < H; ++ih)
          for (int iw = 0; iw < W; ++iw) {
            double acc = 0.0;
            for (int k = 0; k < K; ++k)
              for (int r = 0; r < R; ++r) {
                const int oh = ih * st - pad + r;
                if (oh < 0 || oh >= OH) continue;
                for (int s = 0; s < S; ++s) {
                  const int ow = iw * st - pad + s;
                  if (ow < 0 || ow >= OW) continue;
                  acc += (double)dy[IDX4(n, k, oh, ow, K, OH, OW)] * (double)w[(((size_t)c * K + k) * R + r) * S + s];
                }
              }
            dx[IDX4(n, c, ih, iw, C, H, W)] = (float)acc;
          }
  }
  if (dw) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int c = 0; c < C; ++c)
      for (int k = 0; k < K; ++k)
        for (int r = 0; r < R; ++r)
          for (int s = 0; s < S; ++s) {
            double acc = 0.0;
            for (int n = 0; n < N; ++n)
              for (int ih = 0; ih < H; ++ih) {
                const int oh = ih * st - pad + r;
                if (oh < 0 || oh >= OH) continue;
                for (int iw = 0; iw < W; ++iw) {
                  const int ow = iw * st - pad + s;
                  if (ow < 0 || ow >= OW) continue;
                  acc += (double)x[IDX4(n, c, ih, iw, C, H, W)] * (double)dy[IDX4(n, k, oh, ow, K, OH, OW)];
                }
              }
            dw[(((size_t)c * K + k) * R + r) * S + s] = (float)acc;
          }
  }
  if (db) {
    for (int k = 0; k < K; ++k) {
      double acc = 0.0;
      for (int n = 0; n < N; ++n)
        for (int i = 0; i < OH * OW; ++i) acc += (double)dy[((size_t)n * K + k) * OH * OW + i];
      db[k] = (float)acc;
    }
  }
}

/* training-mode batch norm over (N, HW) per channel; updates running stats (unbiased variance), saves mean/invstd */
void orc_bn_train_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                      float* y, float* save_mean, float* save_invstd, int N, int C, int HW, float momentum, float eps) {
#pragma omp parallel for schedule(static)
  for (int c = 0; c < C; ++c) {
    const double cnt = (double)N * HW;
    double s = 0.0;
    for (int n = 0; n < N; ++n)
      for (int i = 0; i < HW; ++i) s += x[((size_t)n * C + c) * HW + i];
    const double mean = s / cnt;
    double q = 0.0;
    for (int n = 0; n < N; ++n)
      for (int i = 0; i < HW; ++i) { const double d = x[((size_t)n * C + c) * HW + i] - mean; q += d * d; }
    const double var = q / cnt;
    const double invstd = 1.0 / sqrt(var + eps);
    save_mean[c] = (float)mean; save_invstd[c] = (float)invstd;
    if (running_mean) {
      running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
      running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * (cnt > 1 ? q / (cnt - 1.0) : var));
    }
    for (int n = 0; n < N; ++n)
      for (int i = 0; i < HW; ++i) {
        const size_t o = ((size_t)n * C + c) * HW + i;
        y[o] = (float)((x[o] - mean) * invstd * gamma[c] + beta[c]);
      }
  }
}

void orc_bn_eval_fwd(const float* x, const float* gamma, const float* beta, const float* running_mean,
                     const float* running_var, float* y, int N, int C, int HW, float eps) {
#pragma omp parallel for schedule(static)
  for (int c = 0; c < C; ++c) {
    const double invstd = 1.0 / sqrt((double)running_var[c] + eps);
    for (int n = 0; n < N; ++n)
      for (int i = 0; i < HW; ++i) {
        const size_t o = ((size_t)n * C + c) * HW + i;
        y[o] = (float)((x[o] - running_mean[c]) * invstd * gamma[c] + beta[c]);
      }
  }
}

/* dx = gamma*invstd*(dy - mean(dy) - xhat*mean(dy*xhat)), dgamma = sum dy*xhat, dbeta = sum dy */
void orc_bn_train_bwd(const float* x, const float* dy, const float* gamma, const float* mean, const float* invstd,
                      float* dx, float* dgamma, float* dbeta, int N, int C, int HW) {
#pragma omp parallel for schedule(static)
  for (int c = 0; c < C; ++c) {
    const double cnt = (double)N * HW;
    double s1 = 0.0, s2 = 0.0;
    for (int n = 0; n < N; ++n)
      for (int i = 0; i < HW; ++i) {
        const size_t o = ((size_t)n * C + c) * HW + i;
        s1 += dy[o];
        s2 += (double)dy[o] * ((x[o] - (double)mean[c]) * invstd[c]);
      }
    dgamma[c] = (float)s2; dbeta[c] = (float)s1;
    for (int n = 0; n < N; ++n)
      for (int i = 0; i < HW; ++i) {
        const size_t o = ((size_t)n * C + c) * HW + i;
        const double xh = (x[o] - (double)mean[c]) * invstd[c];
        dx[o] = (float)((double)gamma[c] * invstd[c] * (dy[o] - s1 / cnt - xh * s2 / cnt));
      }
  }
}

void orc_relu_fwd(const float* x, float* y, size_t n) {
  for (size_t i = 0; i < n; ++i) y[i] = x[i] > 0.f ? x[i] : 0.f;
}
void orc_relu_bwd(const float* x, const float* dy, float* dx, size_t n) {
  for (size_t i = 0; i < n; ++i) dx[i] = x[i] > 0.f ? dy[i] : 0.f;
}

/* max_pool2d(kernel 3, stride 2, padding pad); idx = flat h*W+w of the first maximum (torch's convention) */
void orc_maxpool_fwd(const float* x, float* y, int32_t* idx, int N, int C, int H, int W, int pad) {
  const int OH = (H + 2 * pad - 3) / 2 + 1, OW = (W + 2 * pad - 3) / 2 + 1;
#pragma omp parallel for collapse(2) schedule(static)
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < C; ++c)
      for (int oh = 0; oh < OH; ++oh)
        for (int ow = 0; ow < OW; ++ow) {
          float best = -INFINITY;
          int bi = -1;
          for (int r = 0; r < 3; ++r) {
            const int ih = oh * 2 - pad + r;
            if (ih < 0 || ih >= H) continue;
            for (int s = 0; s < 3; ++s) {
              const int iw = ow * 2 - pad + s;
              if (iw < 0 || iw >= W) continue;
              const float v = x[IDX4(n, c, ih, iw, C, H, W)];
              if (v > best || bi < 0) { best = v; bi = ih * W + iw; }
            }
          }
          y[IDX4(n, c, oh, ow, C, OH, OW)] = best;
          idx[IDX4(n, c, oh, ow, C, OH, OW)] = bi;
        }
}
void orc_maxpool_bwd(const float* dy, const int32_t* idx, float* dx, int N, int C, int H, int W, int OH, int OW) {
  memset(dx, 0, (size_t)N * C * H * W * sizeof(float));
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < C; ++c)
      for (int i = 0; i < OH * OW; ++i) {
        const size_t o = ((size_t)n * C + c) * OH * OW + i;
        dx[((size_t)n * C + c) * H * W + idx[o]] += dy[o];
      }
}

/* y[M,N] = x[M,K] w[N,K]^T + b */
void orc_linear_fwd(const float* x, const float* w, const float* b, float* y, int M, int N, int K) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int m = 0; m < M; ++m)
    for (int n = 0; n < N; ++n) {
      double acc = b ? b[n] : 0.0;
      for (int k = 0; k < K; ++k) acc += (double)x[(size_t)m * K + k] * (double)w[(size_t)n * K + k];
      y[(size_t)m * N + n] = (float)acc;
    }
}
void orc_linear_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int M, int N, int K) {
  if (dx) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int m = 0; m < M; ++m)
      for (int k = 0; k < K; ++k) {
        double acc = 0.0;
        for (int n = 0; n < N; ++n) acc += (double)dy[(size_t)m * N + n] * (double)w[(size_t)n * K + k];
        dx[(size_t)m * K + k] = (float)acc;
      }
  }
  if (dw) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
      for (int k = 0; k < K; ++k) {
        double acc = 0.0;
        for (int m = 0; m < M; ++m) acc += (double)dy[(size_t)m * N + n] * (double)x[(size_t)m * K + k];
        dw[(size_t)n * K + k] = (float)acc;
      }
  }
  if (db)
    for (int n = 0; n < N; ++n) {
      double acc = 0.0;
      for (int m = 0; m < M; ++m) acc += dy[(size_t)m * N + n];
      db[n] = (float)acc;
    }
}

double orc_sqdiff_sum(const float* a, const float* b, size_t n) {
  double s = 0.0;
  for (size_t i = 0; i < n; ++i) { const double d = (double)a[i] - (double)b[i]; s += d * d; }
  return s;
}
double orc_kl_sum(const float* mu, const float* logvar, size_t n) {
  double s = 0.0;
  for (size_t i = 0; i < n; ++i) s += 1.0 + logvar[i] - (double)mu[i] * mu[i] - exp((double)logvar[i]);
  return -0.5 * s;
}
/* mean cross entropy; dlogits (may be NULL) = (softmax - onehot)/B */
double orc_cross_entropy(const float* logits, const int64_t* target, int B, int A, float* dlogits) {
  double total = 0.0;
  for (int b = 0; b < B; ++b) {
    const float* l = logits + (size_t)b * A;
    double mx = l[0];
    for (int j = 1; j < A; ++j) if (l[j] > mx) mx = l[j];
    double se = 0.0;
    for (int j = 0; j < A; ++j) se += exp(l[j] - mx);
    const double lse = mx + log(se);
    total += lse - l[target[b]];
    if (dlogits)
      for (int j = 0; j < A; ++j) dlogits[(size_t)b * A + j] = (float)((exp(l[j] - lse) - (j == target[b] ? 1.0 : 0.0)) / B);
  }
  return total / B;
}
/* torch.optim.Adam step (no weight decay / amsgrad); step is 1-based */
void orc_adam_step(float* p, const float* g, float* m, float* v, size_t n, double lr, double b1, double b2, double eps, int step) {
  const double bc1 = 1.0 - pow(b1, step), bc2 = 1.0 - pow(b2, step);
  for (size_t i = 0; i < n; ++i) {
    const double mi = b1 * m[i] + (1.0 - b1) * g[i];
    const double vi = b2 * v[i] + (1.0 - b2) * (double)g[i] * g[i];
    m[i] = (float)mi; v[i] = (float)vi;
    p[i] = (float)(p[i] - (lr / bc1) * mi / (sqrt(vi) / sqrt(bc2) + eps));
  }
}
