/*
 * srlz.h — C ABI of libsrlz_hip.so: the MI355X (gfx950) kernels behind srl-zoo's image-representation
 * training hot path.
 *
 * The reference (araffin/srl-zoo, /root/reference) has NO native code: every op below is reached there through
 * PyTorch (torch.nn / autograd / torch.optim).  Each entry point therefore cites the reference call site whose
 * arithmetic it replaces (file:line into /root/reference).  The Python binding a maintainer adds is a ctypes stub
 * (see INTEGRATION.md); this build's own is srl-zoo_amd/srlz/_cabi.py.
 *
 * Conventions
 *   - plain C types only; every pointer is a DEVICE pointer unless the name ends in _host.
 *   - all floating point data is fp32; activations are NHWC ([N][H][W][C], C fastest) unless stated NCHW.
 *     The reference's tensors are NCHW ([B,C,D1,D2]; D1/D2 are the image's W/H because the loader transposes,
 *     preprocessing/data_loader.py:255); the NCHW<->NHWC change happens INSIDE conv1 (reads NCHW) and the last
 *     ConvTranspose (writes NCHW), so callers only ever hand over / receive reference-layout images.
 *   - parameters cross the ABI in the reference's state_dict layouts (Conv2d [Cout,Cin,kh,kw],
 *     ConvTranspose2d [Cin,Cout,kh,kw], Linear [out,in]); *_pack_* entry points build the kernels' private copies.
 *   - the caller owns all memory (weights, activations, workspaces); nothing is allocated, freed or retained.
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, no call synchronises.
 *   - return 0 on success, a negative srlz_status otherwise; srlz_last_error() gives the text (thread-local).
 */
#ifndef SRLZ_H
#define SRLZ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* srlz_stream_t; /* hipStream_t */

enum srlz_status {
  SRLZ_OK = 0,
  SRLZ_ERR_BAD_DESC = -1,   /* unsupported shape / inconsistent descriptor */
  SRLZ_ERR_WORKSPACE = -2,  /* workspace too small */
  SRLZ_ERR_HIP = -3,        /* a HIP runtime call failed (text in srlz_last_error) */
  SRLZ_ERR_NULL = -4        /* required pointer is NULL */
};

/* ABI version: bumped whenever an existing entry point changes its signature or the meaning / size of a buffer it fills, so that a
 * binding built against an older header fails at load time instead of passing a pointer where a float is expected (a binding
 * compares srlz_version() with the SRLZ_ABI_VERSION it was written for; srl-zoo_amd/srlz/_cabi.py does).
 * 101 (round 4): srlz_convT_out_bwd_fused carries three gain arguments (added in round 3 without a bump); its partial records and
 *                workspace follow the strip geometry of csrc/convt_out.hip; srlz_convT_out_fwd_loss_workgroups counts those strips'
 *                workgroups. */
#define SRLZ_ABI_VERSION 104
int srlz_version(void);
const char* srlz_last_error(void);
/* Number of CUs of the current device (used by callers to size persistent grids / workspaces). */
int srlz_device_cus(void);

/* ------------------------------------------------------------------------------------------------------------
 * 64 -> 64 channel convolutions (3x3, stride 1 or 2): fp32 MFMA implicit GEMM.
 * Replaces nn.Conv2d / nn.ConvTranspose2d forward + autograd backward for
 *   conv3x3 s1 p1      models/models.py:54,217-226      conv3x3 s2 p1   models/models.py:59
 *   ConvTranspose2d(64,64,3,stride=2) x4               models/models.py:66,70,74,78
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct {
  int n;          /* images */
  int hi, wi;     /* forward-input spatial size  (NHWC, 64 channels) */
  int ho, wo;     /* forward-output spatial size (NHWC, 64 channels) */
  int ksize;      /* 3 */
  int stride;     /* 1 or 2 */
  int pad;        /* conv: zero padding; convT: `padding` argument */
  int transposed; /* 0 = nn.Conv2d, 1 = nn.ConvTranspose2d */
  int groups;     /* BatchNorm groups batched along n (0 / 1 = one): images [g*n/groups, (g+1)*n/groups) are group g, i.e. the
                     launch is the union of `groups` independent model calls — `obs || next_obs` of one training step,
                     models/learner.py:392-393.  Every BatchNorm record argument (x_bnp, bnp, sums) then holds `groups`
                     consecutive records, statistics partials come group after group (tiles never straddle two groups),
                     weight gradients are summed over all groups. */
} srlz_conv64_desc;

/* floats needed for one packed weight copy (9 taps x 64 x 64) */
size_t srlz_conv64_packed_floats(void);
/* w_ref (reference layout) -> wpack_fwd (forward / weight-grad operand) and wpack_bwd (data-grad operand). */
int srlz_conv64_pack_weights(const float* w_ref, float* wpack_fwd, float* wpack_bwd,
                             const srlz_conv64_desc* d, srlz_stream_t stream);
/* number of per-tile BatchNorm partial records forward() writes (each record = 128 floats: sum[64], sumsq[64]) */
int srlz_conv64_fwd_tiles(const srlz_conv64_desc* d);
/* y = conv(x) (+bias).  bias may be NULL.  stats_partial may be NULL; otherwise receives
 * srlz_conv64_fwd_tiles(d) x 128 floats of per-tile sum / sum-of-squares per channel over y (BatchNorm input).
 * x_bnp (may be NULL): x is the RAW output of the previous convolution and the layer input is relu(batchnorm(x)) with
 * this 4x64 record (srlz_bn_finalize / srlz_bn_eval_params) — nn.BatchNorm2d + nn.ReLU of models/models.py:67-80 fused
 * into the operand load, so the activated tensor is never written to memory. */
int srlz_conv64_fwd(const float* x, const float* wpack_fwd, const float* bias, float* y, float* stats_partial,
                    const float* x_bnp, const srlz_conv64_desc* d, srlz_stream_t stream);
/* Fused BatchNorm + ReLU BACKWARD operand.  Wherever a `dy` argument is followed by a `const srlz_bn_bwd_operand* dy_bn`,
 * a non-NULL record means: the convolution whose gradient is being taken was followed by BatchNorm2d + ReLU
 * (models/models.py:67-80), `dy` actually holds dA = d(loss)/d(relu(bn(y))), and the kernel rebuilds
 *   d(loss)/dy = scale * ( dA*[bn(y)>0] - sums[c]/count - xhat*sums[64+c]/count )      (training; eval: scale*dA*[..])
 * while loading its operand, so d(loss)/dy is never written to memory.  y = the raw convolution output (same shape as
 * dA), bnp = its 4x64 BatchNorm record, sums = output of srlz_bn_relu_bwd_sums, count = N*H*W of y.
 * dy_out (srlz_conv64_bwd_data only, may be NULL): the rebuilt d(loss)/dy is also stored there, every element exactly
 * once, as a by-product of the operand staging — the weight-gradient kernel then reads it as a plain dy. */
typedef struct {
  const float* y;
  const float* bnp;
  const float* sums;
  long long count;   /* N*H*W of y PER GROUP */
  int training;
  float* dy_out;
} srlz_bn_bwd_operand; /* with srlz_conv64_desc.groups > 1: bnp / sums hold one record per group */
/* dx = d(loss)/d(x) from dy (dy_bn may be NULL: dy is then the plain gradient). */
int srlz_conv64_bwd_data(const float* dy, const float* wpack_bwd, float* dx, const srlz_bn_bwd_operand* dy_bn,
                         const srlz_conv64_desc* d, srlz_stream_t stream);
/* The same data gradient for a layer whose INPUT was a pooled map — conv3x3 after BatchNorm2d + ReLU + MaxPool2d(3, 2)
 * (models/models.py:50-54, 55-59) — with the BatchNorm-backward sums of that pooled block taken in the epilogue: dx is d(loss)/d(pooled),
 * and the tile that writes it also reads `pooled` (same shape) and leaves, per tile,  bn_bwd_partial[tile][0..64) = sum dz,
 * [64..128) = sum dz * xhat  with dz = dx where pooled > 0 (the gradient of max-pool + ReLU lives at the argmax, where the pooled value
 * is relu(bn(y)): xhat follows from it; pool_y / pool_argmax are read only for channels whose BatchNorm scale is exactly 0).
 * srlz_bn_bwd_finalize_partials turns the srlz_conv64_bwd_data_tiles(d) records into sums / dgamma / dbeta: the separate pass of
 * srlz_bn_relu_pool_bwd_sums over (d pooled, pooled, argmax) disappears.  pd describes the pooling that produced `pooled`. */
struct srlz_pool_desc_s;
int srlz_conv64_bwd_data_tiles(const srlz_conv64_desc* d);
int srlz_conv64_bwd_data_pool_sums(const float* dy, const float* wpack_bwd, float* dx, const float* pooled, const float* pool_bnp,
                                   const float* pool_y, const uint8_t* pool_argmax, const struct srlz_pool_desc_s* pd,
                                   float* bn_bwd_partial, const srlz_conv64_desc* d, srlz_stream_t stream);
/* The WHOLE backward of a decoder block's ConvTranspose2d(64, 64, 3, stride 2) + BatchNorm2d + ReLU (models/models.py:70-80, as
 * autograd runs it for loss.backward(), models/learner.py:489) in ONE launch: srlz_conv64_bwd_data(dy, dy_bn) and
 * srlz_conv64_bwd_weight(x, dy_out, x_bnp) on a single staging of the rebuilt d(loss)/dy, which therefore never touches memory
 * (dy_bn->dy_out must be NULL).  x = the RAW input tensor of the layer, x_bnp its BatchNorm record(s) (the layer's input is
 * relu(batchnorm(x)); required), dy / dy_bn as in srlz_conv64_bwd_data (dy_bn required).  dx is bit-identical to
 * srlz_conv64_bwd_data's; dw_ref / dbias are deterministic (per-workgroup partials in `ws`, fixed-order fp64 second stage).
 * Only where srlz_conv64_bwd_fused_supported(d) != 0 (stride-2 transposed layers with at least 8 tiles; groups <= 2). */
int srlz_conv64_bwd_fused_supported(const srlz_conv64_desc* d);
/* != 0: srlz_conv64_fwd (backward_data = 0) / srlz_conv64_bwd_data (1) of this layer with a plain operand and no bias run as
 * conv64_gather_pipe_kernel — the software-pipelined persistent kernel of the stride-2 gather programs (conv3's forward,
 * /root/reference/models/models.py:59, and the first ConvTranspose's data gradient) — rather than conv64_fwd_kernel: what a profile
 * will list the launch under. */
int srlz_conv64_gather_pipe_supported(const srlz_conv64_desc* d, int backward_data);
size_t srlz_conv64_bwd_fused_workspace(const srlz_conv64_desc* d);
/* x_bn_bwd_partial (may be NULL; round 6): dx of this call is dA of the BatchNorm + ReLU that produced the layer's input from x —
 * the previous decoder block's (models/models.py:67-77) — and the tile that writes it holds relu(bn(x)) of the same positions, so the
 * launch also leaves that layer's two BatchNorm-backward sums,  sum dA*[bn(x)>0]  and  sum dA*[bn(x)>0]*xhat,  as
 * srlz_conv64_bwd_fused_bn_rows(d) partial records of 128 floats ([groups][rows / groups][128]) for srlz_bn_bwd_finalize_partials:
 * srlz_bn_relu_bwd_sums' separate pass over (x, dx) disappears.  (Channels whose BatchNorm scale is (almost) 0 are summed from x
 * itself by a companion launch inside this call — normally a ~4 us no-op.) */
int srlz_conv64_bwd_fused_bn_rows(const srlz_conv64_desc* d);
int srlz_conv64_bwd_fused(const float* x, const float* x_bnp, const float* dy, const srlz_bn_bwd_operand* dy_bn,
                          const float* wpack_bwd, float* dx, float* dw_ref, float* dbias /* may be NULL */,
                          float* x_bn_bwd_partial /* may be NULL */, void* ws, size_t ws_bytes, const srlz_conv64_desc* d,
                          srlz_stream_t stream);
/* ---- conv3x3 stride 1 pad 1 (conv2 of the encoder, models/models.py:54) as Winograd F(2x2, 3x3) on the fp32 matrix cores (round 6;
 * csrc/wino.hip): 16 multiplications per (ci, co) and 2x2 output patch instead of 36 — what cuDNN, the reference's backend, runs for
 * this layer.  Products and accumulation stay fp32; outputs differ from the direct fp32 chain by a few 1e-7 of the output scale and do
 * not depend on how images are batched or grouped.  srlz_conv64_wino_supported: non-transposed, stride 1, pad 1, even sizes.
 * upack_* = srlz_conv64_wino_packed_floats() floats each (G g G^T in the kernel's layout; fwd for srlz_conv64_wino_fwd, bwd for
 * srlz_conv64_wino_bwd_data*; either may be NULL).  stats_partial: srlz_conv64_wino_tiles(d) records of 128 floats, group after group. */
int srlz_conv64_wino_supported(const srlz_conv64_desc* d);
size_t srlz_conv64_wino_packed_floats(void);
int srlz_conv64_wino_pack_weights(const float* w_ref, float* upack_fwd, float* upack_bwd, srlz_stream_t stream);
int srlz_conv64_wino_tiles(const srlz_conv64_desc* d);
int srlz_conv64_wino_fwd(const float* x, const float* upack_fwd, const float* bias /* may be NULL */, float* y,
                         float* stats_partial /* may be NULL */, const float* x_bnp /* may be NULL: as srlz_conv64_fwd's — x is the raw
                         output of the previous convolution, the layer's input relu(batchnorm(x)) with these records */,
                         const srlz_conv64_desc* d, srlz_stream_t stream);
/* dx = d(loss)/dx from dy, same kernel with upack_bwd.  The _pool_sums form is srlz_conv64_bwd_data_pool_sums' (above): dx is the gradient
 * of the pooled map `pd` describes and the launch also leaves the pooled block's two BatchNorm-backward sums as
 * srlz_conv64_wino_bwd_data_rows(d) records of 128 floats (group after group; a group's last 64 records come from a companion launch
 * for channels whose BatchNorm scale is (almost) 0 — normally zeros) for srlz_bn_bwd_finalize_partials. */
int srlz_conv64_wino_bwd_data(const float* dy, const float* upack_bwd, float* dx, const srlz_conv64_desc* d, srlz_stream_t stream);
int srlz_conv64_wino_bwd_data_rows(const srlz_conv64_desc* d);
int srlz_conv64_wino_bwd_data_pool_sums(const float* dy, const float* upack_bwd, float* dx, const float* pooled, const float* pool_bnp,
                                        const float* pool_y, const uint8_t* pool_argmax, const struct srlz_pool_desc_s* pd,
                                        float* bn_bwd_partial, const srlz_conv64_desc* d, srlz_stream_t stream);
/* dw_ref (reference layout [co][ci][3][3]) = d(loss)/d(w) of the same layer by the transposed Winograd algorithm: per transform-domain
 * component one [co] x [ci] contraction over all 2x2 patches of (A dY A^T) and (B^T x B), G^T . G in the (fixed-order, fp64) second
 * stage.  Layers without a bias (conv3x3 of the reference has none); ws: srlz_conv64_wino_bwd_weight_workspace(d) bytes. */
size_t srlz_conv64_wino_bwd_weight_workspace(const srlz_conv64_desc* d);
int srlz_conv64_wino_bwd_weight(const float* x, const float* dy, float* dw_ref, void* ws, size_t ws_bytes, const srlz_conv64_desc* d,
                                srlz_stream_t stream);
/* workspace (bytes) for bwd_weight */
size_t srlz_conv64_bwd_weight_workspace(const srlz_conv64_desc* d);
/* dw_ref (reference layout) = d(loss)/d(w); dbias[64] = sum of dy over n,h,w (may be NULL).
 * Deterministic: split-K partials in `ws`, fixed-order second-stage reduction (learner.py:62 asks cuDNN for the same). */
int srlz_conv64_bwd_weight(const float* x, const float* dy, float* dw_ref, float* dbias, const float* x_bnp /* as above */,
                           const srlz_bn_bwd_operand* dy_bn /* may be NULL */,
                           void* ws, size_t ws_bytes, const srlz_conv64_desc* d, srlz_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * convN — FORWARD-ONLY 3x3 (pad 1, stride 1/2) and 1x1 (stride 2) convolutions without bias for Cin, Cout multiples of 64:
 * the frozen ResNet-18 trunk behind EmbeddingNet (models/triplet.py:6-39; torchvision 0.2.1 resnet18: BasicBlock conv3x3
 * layers and the 1x1 downsample convolutions).  Same fp32-MFMA implicit GEMM as the 64->64 family; no gradient exists
 * because the reference freezes the trunk (triplet.py:17-19).
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct {
  int n, hi, wi, ho, wo; /* NHWC activations: x [n,hi,wi,cin] -> y [n,ho,wo,cout] */
  int cin, cout;         /* multiples of 64 */
  int ksize, stride, pad; /* (3, 1|2, 1) or (1, 2, 0) */
  int groups;            /* BatchNorm groups batched along n (0 / 1 = one), as in srlz_conv64_desc: the launch is the union of
                            `groups` independent trunk calls — the anchor / positive / negative views of obs and next_obs of one
                            time-contrastive step, models/learner.py:383-391.  stats_partial is then
                            [cout/64][groups][tiles / groups][128], x_bnp [groups][cin/64][256] */
} srlz_convn_desc;
size_t srlz_convn_packed_floats(const srlz_convn_desc* d);
/* w_ref [cout,cin,k,k] (torch layout) -> the kernel's packed copy (srlz_convn_packed_floats floats) */
int srlz_convn_pack_weights(const float* w_ref, float* wpack, const srlz_convn_desc* d, srlz_stream_t stream);
/* per-tile BatchNorm partial records per block of 64 output channels (stats_partial is [cout/64][tiles][128]) */
int srlz_convn_fwd_tiles(const srlz_convn_desc* d);
/* y = conv(x); x_bnp (may be NULL): the input is relu(batchnorm(x)) with one 256-float record per block of 64 input channels */
int srlz_convn_fwd(const float* x, const float* wpack, float* y, float* stats_partial, const float* x_bnp,
                   const srlz_convn_desc* d, srlz_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * "Skinny" convolutions: one side has C in {3,6,9} image channels stored NCHW, the other 64 channels NHWC.
 *   kind 0: nn.Conv2d(C,64,k=7,s=2,p=3,bias=False)       models/models.py:49   (encoder conv1)
 *   kind 1: nn.ConvTranspose2d(64,C,k=4,s=2)             models/models.py:82   (decoder's last layer)
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct {
  int n;       /* images */
  int c;       /* image channels: 3, 6 or 9 */
  int himg, wimg; /* image-side spatial size (224 x 224) */
  int hf, wf;  /* 64-channel feature-map spatial size (112x112 for kind 0, 111x111 for kind 1) */
  int kind;    /* 0 = conv1 7x7 s2 p3 ; 1 = convT 4x4 s2 p0 */
  int groups;  /* BatchNorm groups along n (0 / 1 = one), as in srlz_conv64_desc */
} srlz_skinny_desc;

int srlz_skinny_tiles(const srlz_skinny_desc* d); /* BN partial records written by conv1 forward */
/* kind 0 forward: x_nchw [N,C,H,W] -> y_nhwc [N,hf,wf,64]; w_ref [64,C,7,7]. stats_partial as in srlz_conv64_fwd. */
int srlz_conv1_fwd(const float* x_nchw, const float* w_ref, float* y_nhwc, float* stats_partial,
                   const srlz_skinny_desc* d, srlz_stream_t stream);
/* The same layer on frames that are still the loader's BYTES: x_u8 [N,C,himg,wimg] uint8, planar — the reference's tensor layout
 * (preprocessing/data_loader.py:255: transpose(0,3,2,1) of the [B,H,W,C] image batch) taken BEFORE preprocessInput — and
 * norm_lut = the 3 x 256 table of srlz_normalize_lut.  ((v/255) - mean[c]) / std[c] (preprocessing/utils.py:20-32) is applied
 * while the window lands in LDS: identical bits to srlz_conv1_fwd on the normalised float tensor, a quarter of the input bytes,
 * and no normalisation pass in the step.  Same for the fused weight gradient and the fused reconstruction loss below. */
int srlz_conv1_fwd_u8(const uint8_t* x_u8, const float* norm_lut, const float* w_ref, float* y_nhwc, float* stats_partial,
                      const srlz_skinny_desc* d, srlz_stream_t stream);
size_t srlz_skinny_bwd_weight_workspace(const srlz_skinny_desc* d);
/* kind 0 data gradient: dx_nchw [N,C,himg,wimg] = d(loss)/d(image) from dy_nhwc [N,hf,wf,64].  The training path never
 * needs it (conv1 is the first layer); it exists for the perceptual loss (losses/losses.py:217-236), whose frozen
 * denoiser encodes a DECODED image that carries a gradient (models/learner.py:404-412). */
int srlz_conv1_bwd_data(const float* dy_nhwc, const float* w_ref, float* dx_nchw, const srlz_skinny_desc* d,
                        srlz_stream_t stream);
/* kind 0 weight gradient: dw_ref [64,C,7,7] from x_nchw and dy_nhwc. */
int srlz_conv1_bwd_weight(const float* x_nchw, const float* dy_nhwc, float* dw_ref, void* ws, size_t ws_bytes,
                          const srlz_skinny_desc* d, srlz_stream_t stream);
/* kind 0 weight gradient with the BatchNorm + ReLU + MaxPool backward of the FOLLOWING block fused into the operand load:
 * dy = d(loss)/d(conv1 output) is never materialised (it has no other consumer: conv1 needs no data gradient).
 * y_nhwc / bnp / argmax as in srlz_bn_relu_pool_fwd, dpooled = gradient of the pooled map (NHWC),
 * sums = output of srlz_bn_relu_pool_bwd_sums; pd describes that pooling layer. */
int srlz_conv1_bwd_weight_fused(const float* x_nchw, const float* y_nhwc, const float* bnp, const uint8_t* argmax,
                                const float* dpooled, const float* sums, int training, float* dw_ref, void* ws,
                                size_t ws_bytes, const srlz_skinny_desc* d, const struct srlz_pool_desc_s* pd,
                                srlz_stream_t stream);
int srlz_conv1_bwd_weight_fused_u8(const uint8_t* x_u8, const float* norm_lut, const float* y_nhwc, const float* bnp,
                                   const uint8_t* argmax, const float* dpooled, const float* sums, int training, float* dw_ref,
                                   void* ws, size_t ws_bytes, const srlz_skinny_desc* d, const struct srlz_pool_desc_s* pd,
                                   srlz_stream_t stream);
/* kind 1 forward: x_nhwc [N,hf,wf,64] -> y_nchw [N,C,H,W] = convT(x) + bias; w_ref [64,C,4,4]. */
int srlz_convT_out_fwd(const float* x_nhwc, const float* w_ref, const float* bias, float* y_nchw,
                       const float* x_bnp /* may be NULL, see srlz_conv64_fwd */, const srlz_skinny_desc* d,
                       srlz_stream_t stream);
/* kind 1 data gradient: dx_nhwc [N,hf,wf,64] from dy_nchw.
 * x_raw / x_bnp / bn_bwd_partial (all three or none): the layer input was relu(batchnorm(x_raw)) (models/models.py:79-81),
 * so dx is dA of that BatchNorm+ReLU; the epilogue then also reads x_raw and writes, per 16x16 tile, the partial
 * BatchNorm-backward sums  bn_bwd_partial[tile][0..64) = sum dA*[bn(x)>0],  [64..128) = sum dA*[bn(x)>0]*xhat
 * (srlz_skinny_tiles(d) rows) for srlz_bn_bwd_finalize_partials — no separate pass over (dA, x_raw). */
int srlz_convT_out_bwd_data(const float* dy_nchw, const float* w_ref, float* dx_nhwc, const float* x_raw,
                            const float* x_bnp, float* bn_bwd_partial, const srlz_skinny_desc* d, srlz_stream_t stream);
/* kind 1 weight gradient: dw_ref [64,C,4,4], dbias [C]. */
int srlz_convT_out_bwd_weight(const float* x_nhwc, const float* dy_nchw, float* dw_ref, float* dbias,
                              const float* x_bnp, void* ws, size_t ws_bytes, const srlz_skinny_desc* d,
                              srlz_stream_t stream);

/* kind 1 data gradient + BatchNorm-backward partials + weight gradient + bias gradient in ONE pass over dy and x_raw (C == 3 or 6,
 * the layer input was relu(batchnorm(x_raw)): the autograd backward of nn.ConvTranspose2d(64, C, 4, stride=2), models/models.py:82,
 * as srlz_convT_out_bwd_data(x_raw, x_bnp, bn_bwd_partial) followed by srlz_convT_out_bwd_weight(x_bnp) would compute it, with
 * x_raw (1.6 GB at 512 images) and dy crossing HBM once instead of twice.  Round 4: output-stationary, one wave per strip of 16
 * feature columns x R rows (csrc/convt_out.hip); bn_bwd_partial has srlz_convT_out_bwd_fused_tiles(d) rows of 128 floats (one per
 * strip, image-major: the records of BatchNorm group g are rows [g, g+1) * tiles / groups); ws >=
 * srlz_convT_out_bwd_fused_workspace(d) bytes.  srlz_convT_out_bwd_fused_supported: 1 for C in {3, 6}, else 0 (use the two calls). */
int srlz_convT_out_bwd_fused_supported(const srlz_skinny_desc* d);
int srlz_convT_out_bwd_fused_tiles(const srlz_skinny_desc* d);
size_t srlz_convT_out_bwd_fused_workspace(const srlz_skinny_desc* d);
/* dy_gain_dev != NULL: `dy_nchw` holds the reconstruction ERROR dec - obs left behind by srlz_convT_out_fwd_loss, and the loss
 * gradient d(loss)/d(dec) = ((dy_gain_dev[0] / dy_gain_div) * dy_gain_coef) * (dec - obs) is formed while it is staged
 * (dy_gain_dev = the upstream gradient of the loss scalar on the device; div = numel per frame for the mean form, 1 for the sum
 * form; coef = 2) — the rounding order autograd uses for sum((dec-obs)^2)/numel.  NULL: dy_nchw is the gradient itself. */
int srlz_convT_out_bwd_fused(const float* dy_nchw, const float* w_ref, float* dx_nhwc, const float* x_raw, const float* x_bnp,
                             float* bn_bwd_partial, float* dw_ref, float* dbias /* may be NULL */, void* ws, size_t ws_bytes,
                             const float* dy_gain_dev, float dy_gain_div, float dy_gain_coef,
                             const srlz_skinny_desc* d, srlz_stream_t stream);

/* The last ConvTranspose forward WITH the reconstruction / generation loss of the step taken in its epilogue (SURVEY.md 8a' K11 /
 * K12; replaces reconstructionLoss / autoEncoderLoss, /root/reference/losses/losses.py:172-196, and generationLoss, :199-214,
 * on the training path — there `obs` is read twice and `decoded` three times, here `decoded` never leaves the chip):
 * the batch is the two frames of a step (images [0, n/2) = obs, [n/2, n) = next_obs; d->groups = BatchNorm groups as usual);
 * err_nchw = dec - target (what the backward needs), dec_nchw = the reconstruction itself (optional, NULL on the training path),
 * loss_partial[2][srlz_convT_out_fwd_loss_workgroups(d)] = per-frame, per-workgroup sums of squared errors (fp64), to be
 * combined by srlz_pair_loss_finalize.  The gradient is err times a scalar: see srlz_convT_out_bwd_fused / srlz_scale_by_scalar. */
int srlz_convT_out_fwd_loss_workgroups(const srlz_skinny_desc* d);
int srlz_convT_out_fwd_loss(const float* x_nhwc, const float* w_ref, const float* bias, const float* target_nchw, float* err_nchw,
                            float* dec_nchw, const float* x_bnp, double* loss_partial, const srlz_skinny_desc* d,
                            srlz_stream_t stream);
/* target_u8: the observations as uint8 [N,C,himg,wimg] + the normalisation table (see srlz_conv1_fwd_u8) */
int srlz_convT_out_fwd_loss_u8(const float* x_nhwc, const float* w_ref, const float* bias, const uint8_t* target_u8,
                               const float* norm_lut, float* err_nchw, float* dec_nchw, const float* x_bnp, double* loss_partial,
                               const srlz_skinny_desc* d, srlz_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * BatchNorm2d(64) (+ ReLU (+ MaxPool 3x3 s2)) — nn.BatchNorm2d / nn.ReLU / nn.MaxPool2d,
 * models/models.py:50-52,55-57,60-62 (encoder) and 67-68,71-72,75-76,79-80 (decoder).
 * bnp is a 4x64 float record {mean, invstd, scale = gamma*invstd, shift = beta - mean*scale}.
 * ------------------------------------------------------------------------------------------------------------ */
/* Training-mode statistics from the convolution's per-tile partials: fills bnp and applies `repeat` momentum
 * updates of running_mean / running_var (momentum, unbiased variance — torch defaults eps 1e-5, momentum 0.1).
 * batch_stat[2][64] (may be NULL) receives {mean, unbiased var} so the update can be replayed (VAE getStates quirk,
 * models/learner.py:402).  ws: srlz_bn_bwd_workspace(0) bytes of scratch (two-stage fp64 reduction of the partials).
 * groups > 1: `groups` independent BatchNorm calls batched along n (the conv descriptors' `groups`): stats_partial holds
 * n_partials / groups records per group, group after group; count is PER GROUP; bnp receives `groups` records of 256
 * floats, batch_stat `groups` x 128; the running statistics take the groups' momentum updates in order (obs, then
 * next_obs, models/learner.py:392-393).  Per group the arithmetic is exactly that of a single-group call. */
int srlz_bn_finalize(const float* stats_partial, int n_partials, int groups, long long count, const float* gamma,
                     const float* beta, float eps, float momentum, int repeat, float* running_mean,
                     float* running_var, long long* num_batches_tracked /* int64 counter += groups; may be NULL */,
                     float* bnp, float* batch_stat, void* ws, size_t ws_bytes, srlz_stream_t stream);
/* A C-channel BatchNorm (C = 64 * chunks) of the frozen ResNet-18 trunk as `chunks` independent 64-channel layers:
 * stats_partial is srlz_convn_fwd's [chunks][tiles][128]; gamma / beta / running_* hold C floats; bnp receives `chunks`
 * records of 256 floats (training-mode forward of nn.BatchNorm2d: batch statistics + one momentum update; the trunk's
 * parameters are frozen but the reference leaves it in train() mode, models/learner.py:365 — so its statistics do move).
 * groups > 1 (ABI 103): `groups` independent BatchNorm calls batched along n — stats_partial is
 * [chunks][groups][tiles / groups][128] (`tiles` counts all groups), count is PER GROUP, bnp receives [groups][chunks][256]
 * (a group's records contiguous), the running statistics take the groups' momentum updates in order and the counter += groups:
 * bit for bit what `groups` separate calls leave.  ws >= srlz_bn_finalize_chunks_workspace(chunks, groups) bytes. */
size_t srlz_bn_finalize_chunks_workspace(int chunks, int groups);
int srlz_bn_finalize_chunks(const float* stats_partial, int tiles, int chunks, int groups, long long count, const float* gamma,
                            const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                            long long* num_batches_tracked /* += groups; may be NULL */, float* bnp, void* ws, size_t ws_bytes,
                            srlz_stream_t stream);
int srlz_bn_eval_params_chunks(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                               float eps, int chunks, float* bnp, srlz_stream_t stream);
/* out = relu(bn(a) + (b_bnp ? bn(b) : b)) over pixels x 64*chunks — BasicBlock's `out = relu(bn2(out) + identity)`;
 * groups > 1: records [groups][chunks][256], group g = pixels [g, g+1) * pixels / groups */
int srlz_bn_add_relu(const float* a, const float* a_bnp, const float* b, const float* b_bnp, float* out, long long pixels,
                     int chunks, int groups, srlz_stream_t stream);
/* out[n,c] = mean over hw of x[n,hw,c] — resnet18.avgpool */
int srlz_avgpool_nhwc(const float* x, float* out, int n, int hw, int c, srlz_stream_t stream);
/* Eval-mode bnp from running statistics. */
int srlz_bn_eval_params(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                        float eps, float* bnp, srlz_stream_t stream);
/* running = (1-m)*running + m*batch_stat, once (replay of a previous call's statistics). */
int srlz_bn_replay(const float* batch_stat, float momentum, float* running_mean, float* running_var,
                   srlz_stream_t stream);
/* The same for up to SRLZ_BN_REPLAY_MAX different layers in ONE launch, each with its nn.BatchNorm2d.num_batches_tracked (+= 1; may be
 * NULL): the learner's getStates(obs) after forward(obs) in train mode (models/learner.py:402 with models.py:131-139) re-runs the three
 * encoder BatchNorms of a frame on the same batch.  `items` is a HOST array. */
#define SRLZ_BN_REPLAY_MAX 8
typedef struct {
  const float* batch_stat;  /* [128]: batch mean, unbiased batch variance (srlz_bn_finalize's batch_stat record of the group) */
  float* running_mean;
  float* running_var;
  long long* num_batches_tracked;
} srlz_bn_replay_item;
int srlz_bn_replay_many(const srlz_bn_replay_item* items, int n, float momentum, srlz_stream_t stream);

typedef struct srlz_pool_desc_s {
  int n, h, w;      /* input  [N,h,w,64] */
  int hp, wp;       /* pooled [N,hp,wp,64] */
  int pool_pad;     /* 0 or 1 (kernel 3, stride 2) */
  int out_nchw;     /* 1: write the pooled map as [N,64,hp,wp] (feeds Linear(2304,S), autoencoders.py:107-108) */
  int groups;       /* BatchNorm groups along n (0 / 1 = one), as in srlz_conv64_desc: bnp / sums hold one record per group */
} srlz_pool_desc;

/* pooled = maxpool3x3s2(relu(y*scale+shift)); argmax (uint8 per output, window index 0..8) may be NULL (eval). */
int srlz_bn_relu_pool_fwd(const float* y, const float* bnp, float* pooled, uint8_t* argmax,
                          const srlz_pool_desc* d, srlz_stream_t stream);
size_t srlz_bn_bwd_workspace(long long elems);
/* First half of the backward below: sums[128] = {sum dz[64], sum dz*xhat[64]} (= dbeta, dgamma). */
int srlz_bn_relu_pool_bwd_sums(const float* y, const float* bnp, const uint8_t* argmax, const float* dpooled,
                               const float* pooled, float* sums, float* dgamma, float* dbeta, void* ws, size_t ws_bytes,
                               const srlz_pool_desc* d, srlz_stream_t stream);
/* dy (same shape as y) and dgamma[64], dbeta[64] from dpooled.  training != 0: batch-statistics backward;
 * training == 0: running-statistics backward (validation minibatches, models/learner.py:362-364,489). */
int srlz_bn_relu_pool_bwd(const float* y, const float* bnp, const uint8_t* argmax, const float* dpooled,
                          const float* pooled /* forward output, may be NULL (slower) */, float* dy, float* dgamma,
                          float* dbeta, int training, void* ws, size_t ws_bytes, const srlz_pool_desc* d,
                          srlz_stream_t stream);
/* Stage 2 alone: dy from sums that are already known (srlz_bn_relu_pool_bwd_sums, or srlz_bn_bwd_finalize_partials over the records of
 * srlz_conv64_bwd_data_pool_sums). */
int srlz_bn_relu_pool_bwd_apply(const float* y, const float* bnp, const uint8_t* argmax, const float* dpooled, const float* sums,
                                float* dy, int training, const srlz_pool_desc* d, srlz_stream_t stream);
/* a = relu(y*scale+shift) over `pixels` x 64 */
int srlz_bn_relu_fwd(const float* y, const float* bnp, float* a, long long pixels, srlz_stream_t stream);
/* Second stage for the per-tile partials of a data-gradient epilogue (srlz_convT_out_bwd_data, srlz_conv64_bwd_data):
 * partial[n_partials][128] -> sums[128], dgamma[64], dbeta[64]; fp64 across tiles, fixed order. */
int srlz_bn_bwd_finalize_partials(const float* partial, int n_partials, int groups, float* sums, float* dgamma, float* dbeta,
                                  void* ws, size_t ws_bytes, srlz_stream_t stream);
/* First half of srlz_bn_relu_bwd: sums[128] = {sum dz[64], sum dz*xhat[64]} (= dbeta, dgamma), for srlz_bn_bwd_operand. */
/* (groups > 1: bnp / sums hold one record per group, `pixels` counts all groups; dgamma / dbeta are the groups' totals) */
int srlz_bn_relu_bwd_sums(const float* y, const float* bnp, const float* da, float* sums, float* dgamma, float* dbeta,
                          void* ws, size_t ws_bytes, long long pixels, int groups, srlz_stream_t stream);
int srlz_bn_relu_bwd(const float* y, const float* bnp, const float* da, float* dy, float* dgamma, float* dbeta,
                     int training, void* ws, size_t ws_bytes, long long pixels, int groups, srlz_stream_t stream);
/* [N,C,H,W] <-> [N,H,W,C] for the two 6x6x64 seams around the FC layers (autoencoders.py:107-108,116-117). */
int srlz_nchw_to_nhwc(const float* src, float* dst, int n, int c, int h, int w, srlz_stream_t stream);
int srlz_nhwc_to_nchw(const float* src, float* dst, int n, int c, int h, int w, srlz_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Linear layers — nn.Linear: autoencoders.py:94-100, vae.py:52-57, forward_inverse.py:16,48-55.
 * y[M,N] = x[M,K] . w[N,K]^T + b[N]   (w in torch [out,in] layout), optional ReLU on y.
 * ------------------------------------------------------------------------------------------------------------ */
/* ws (may be NULL): srlz_linear_workspace bytes; lets skinny GEMMs split their reduction over more workgroups
 * (deterministic two-stage sum). */
size_t srlz_linear_workspace(int M, int N, int K);
int srlz_linear_fwd(const float* x, const float* w, const float* b, float* y, int M, int N, int K, int relu,
                    void* ws, size_t ws_bytes, srlz_stream_t stream);
/* dx[M,K] = dy[M,N] . w[N,K] */
/* The forward model's residual (forward_inverse.py:27-37: next_state = state + Linear([state ; onehot(action)])) in the GEMM's
 * epilogue: y = (x . w^T + b) + res with res [M, N] — two separately rounded fp32 adds, as the reference rounds them — and its data
 * gradient restricted to the first Kout input columns (the state part of the concatenation) with the residual branch's gradient
 * added: dx[M, Kout] = (dy . w)[:, :Kout] + res, res [M, Kout] (res = dy for the residual above: N == Kout). */
int srlz_linear_fwd_res(const float* x, const float* w, const float* b, const float* res, float* y, int M, int N, int K, int relu,
                        void* ws, size_t ws_bytes, srlz_stream_t stream);
int srlz_linear_bwd_data_res(const float* dy, const float* w, const float* res, float* dx, int M, int N, int K, int Kout, void* ws,
                             size_t ws_bytes, srlz_stream_t stream);
int srlz_linear_bwd_data(const float* dy, const float* w, float* dx, int M, int N, int K, void* ws, size_t ws_bytes,
                         srlz_stream_t stream);
/* dw[N,K] = dy^T . x ; db[N] = column sums of dy (may be NULL) */
int srlz_linear_bwd_weight(const float* dy, const float* x, float* dw, float* db, int M, int N, int K,
                           void* ws, size_t ws_bytes, srlz_stream_t stream);
/* dy *= (y > 0)  — backward of the fused ReLU */
int srlz_relu_bwd_inplace(const float* y, float* dy, long long n, srlz_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Losses — losses/losses.py.  `out` scalars are device floats; partial sums are reduced in fp64 in a fixed order.
 * ------------------------------------------------------------------------------------------------------------ */
size_t srlz_reduce_workspace(long long n);
/* out[0] = sum((a-b)^2)            reconstructionLoss 172-181 (caller divides by numel) / F.mse_loss(sum) 210-211 */
int srlz_sqdiff_sum(const float* a, const float* b, long long n, float* out, void* ws, size_t ws_bytes,
                    srlz_stream_t stream);
/* The same for `groups` consecutive slices of n_per_group elements (the two frames of a step batched along n): out[g] is
 * group g's sum, each exactly as a single call on that slice would compute it. */
int srlz_sqdiff_sum_groups(const float* a, const float* b, long long n_per_group, int groups, float* out, void* ws,
                           size_t ws_bytes, srlz_stream_t stream);
/* da[i] = coef_dev[0] * coef * (a[i]-b[i])   (gradient of the above w.r.t. a; coef_dev may be NULL = 1) */
int srlz_sqdiff_grad(const float* a, const float* b, const float* coef_dev, float coef, float* da, long long n,
                     srlz_stream_t stream);
/* grouped form: slice g is scaled by (coef_dev[g * coef_stride] / div) * coef   (coef_stride 0: one shared upstream scalar) */
int srlz_sqdiff_grad_groups(const float* a, const float* b, const float* coef_dev, int coef_stride, float div, float coef,
                            float* da, long long n_per_group, int groups, srlz_stream_t stream);
/* The loss of a batched pair a = [a0 ; a1], b = [b0 ; b1] in one go: sums[g] = sum((a_g - b_g)^2) and
 * comb[0] = sums[0]/n + sums[1]/n (mean != 0: autoEncoderLoss, losses.py:184-196) or sums[0] + sums[1] (generationLoss,
 * losses.py:199-214), rounded like the reference's separate fp32 operations. */
/* Second stage of a pair loss whose fp64 partials [2][nb] another kernel wrote (srlz_convT_out_fwd_loss): sums[g] and
 * comb = sums[0]/n + sums[1]/n (mean) or sums[0] + sums[1], rounded exactly like srlz_sqdiff_pair_loss.
 * (/root/reference/losses/losses.py:181,196 / :210-214) */
int srlz_pair_loss_finalize(const double* partial, int nb, long long n_per_group, int mean, float* sums, float* comb,
                            srlz_stream_t stream);
/* out = ((gain_dev[0] / div) * coef) * x (in place allowed): the gradient of a fused pair loss materialised from the stored
 * error for consumers that cannot apply the factor themselves (autograd of F.mse_loss, /root/reference/losses/losses.py:210). */
int srlz_scale_by_scalar(const float* x, const float* gain_dev, float div, float coef, float* out, long long n,
                         srlz_stream_t stream);
int srlz_sqdiff_pair_loss(const float* a, const float* b, long long n_per_group, int mean, float* sums, float* comb,
                          void* ws, size_t ws_bytes, srlz_stream_t stream);
/* out = [a ; b] (n_each floats each) — joins the halves of a batched pair (th.cat of learner.py's obs / next_obs) */
int srlz_join2(const float* a, const float* b, float* out, long long n_each, srlz_stream_t stream);

/* LossManager.computeTotalLoss (losses/losses.py:55-56): total = sum_i w_i * l_i over n <= SRLZ_MAX_LOSS_TERMS device scalars, in
 * Python's left-to-right fp32 order with separately rounded products.  `scalars` and `weights` are HOST arrays (of device pointers /
 * of floats).  tail != NULL: also tail[0] = total, tail[1 + i] = l_i (the step's scalars in the gradient bucket's tail,
 * models/learner.py:485,501-522).  _bwd: g[i] = dout * w_i. */
#define SRLZ_MAX_LOSS_TERMS 15
int srlz_weighted_total(const float* const* scalars, const float* weights, int n, float* total, float* tail,
                        srlz_stream_t stream);
int srlz_weighted_total_bwd(const float* dout, const float* weights, int n, float* g, srlz_stream_t stream);
/* out[0] = -0.5*sum(1 + logvar - mu^2 - exp(logvar))      kullbackLeiblerLoss 239-256 */
/* out[0] = fp32(sum((a-b)^2)) / div      reconstructionLoss 172-181 (div = numel): the fp32 division behind the sum's rounding */
int srlz_sqdiff_mean(const float* a, const float* b, long long n, float div, float* out, void* ws, size_t ws_bytes,
                     srlz_stream_t stream);
int srlz_kl_sum(const float* mu, const float* logvar, long long n, float* out, void* ws, size_t ws_bytes,
                srlz_stream_t stream);
/* dmu += g*mu ; dlogvar += g*0.5*(exp(logvar)-1)   with g = coef_dev[0]*coef */
int srlz_kl_grad(const float* mu, const float* logvar, const float* coef_dev, float coef, float* dmu,
                 float* dlogvar, long long n, srlz_stream_t stream);
/* z = eps*exp(0.5*logvar) + mu        BaseModelVAE.reparameterize models/models.py:147-165 (eps drawn by the host) */
int srlz_reparam_fwd(const float* mu, const float* logvar, const float* eps, float* z, long long n,
                     srlz_stream_t stream);
/* dmu = dz ; dlogvar = dz*eps*0.5*exp(0.5*logvar) */
int srlz_reparam_bwd(const float* dz, const float* logvar, const float* eps, float* dmu, float* dlogvar,
                     long long n, srlz_stream_t stream);
/* out[0] = mean_b( logsumexp(logits[b,:]) - logits[b,target[b]] )   nn.CrossEntropyLoss, inverseModelLoss 117-129
 * dlogits (may be NULL) = (softmax - onehot)/B */
int srlz_cross_entropy(const float* logits, const int64_t* target, int B, int A, float* out, float* dlogits,
                       srlz_stream_t stream);
/* nn.PReLU() with one slope — EmbeddingNet.fc[0], models/triplet.py:24 */
int srlz_prelu_fwd(const float* x, const float* slope, float* y, long long n, srlz_stream_t stream);
int srlz_prelu_bwd(const float* x, const float* slope, const float* dy, float* dx, float* dslope, long long n,
                   srlz_stream_t stream);
/* tripletLoss losses/losses.py:360-376: out[0] = mean_b relu(|s-p|^2 - |s-n|^2 + alpha); hinge[B] = active rows (for bwd) */
int srlz_triplet_fwd(const float* s, const float* p, const float* n, int B, int S, float alpha, float* out, float* hinge,
                     srlz_stream_t stream);
int srlz_triplet_bwd(const float* s, const float* p, const float* n, const float* hinge, const float* g, int B, int S,
                     float* ds, float* dp, float* dn, srlz_stream_t stream);
/* cat[b,:] = [s[b,:S], onehot(a[b])]          forwardModel forward_inverse.py:21-31 + encodeOneHot models.py:229-237 */
/* th.cat((state, next_state), dim=1) of the inverse / reward heads (forward_inverse.py:62,78-95) and the split of its gradient
 * (a or b may be NULL: that half is not wanted); out / in: [rows, ca + cb], a: [rows, ca], b: [rows, cb]. */
int srlz_cat_cols(const float* a, const float* b, float* out, int rows, int ca, int cb, srlz_stream_t stream);
int srlz_split_cols(const float* in, float* a, float* b, int rows, int ca, int cb, srlz_stream_t stream);
/* out = ((t0 + t1) + t2) + t3 over nterms (1..4) tensors of n floats, left to right in fp32: the gradient of a tensor with several
 * consumers (states feed the decoder, the forward / inverse / reward heads and the forward loss: models/learner.py:392-449) summed
 * by ONE launch in a fixed order instead of one accumulation kernel per extra consumer.  terms: HOST array of device pointers. */
int srlz_sum_terms(const float* const* terms, int nterms, float* out, long long n, srlz_stream_t stream);
int srlz_concat_onehot(const float* s, const int64_t* a, float* cat, int B, int S, int A, srlz_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Input pipeline tail: decoded uint8 frames [N,H,W,C] (RGB per 3-group) -> normalised fp32 [N,C,W,H]
 * = preprocessInput(x, "image_net") (preprocessing/utils.py:20-32) + the loader's transpose(0,3,2,1)
 * (preprocessing/data_loader.py:255), bit-identical to the host arithmetic.  Lets the loader ship uint8 over PCIe.
 * ------------------------------------------------------------------------------------------------------------ */
int srlz_normalize_u8(const uint8_t* img_nhwc, float* out_ncwh, int n, int h, int w, int c, srlz_stream_t stream);
/* norm_lut[c][v] (3 x 256 floats) = ((v / 255) - mean[c]) / std[c] for v = 0..255 and the image_net mean / std of channel c —
 * preprocessInput (preprocessing/utils.py:20-32) tabulated with its three fp32 roundings; the operand of the *_u8 entry points. */
int srlz_normalize_lut(float* norm_lut, srlz_stream_t stream);
/* Frames already in the reference's tensor layout but still uint8 ([N,C,*] planar, `plane` bytes per channel plane) -> the float
 * tensor, through the table (for consumers that have no *_u8 form; bit-identical to srlz_normalize_u8 of the NHWC frames). */
int srlz_normalize_u8_planar(const uint8_t* x_u8, const float* norm_lut, float* out, int n, int c, long long plane,
                             srlz_stream_t stream);
/* The dataset resident in HBM (round 4): the loader process of /root/reference/preprocessing/data_loader.py:195-256
 * decodes every frame of every epoch again; here a decoded frame is kept ([frames][C][W][H] uint8, 150 KB each: 100 k frames = 15 GB
 * of the 288) and later epochs move frames BY INDEX.  Whole frames of frame_bytes (a multiple of 16) bytes:
 *   dst[dst_index ? dst_index[i] + dst_shift : i] = src[src_index ? src_index[i] + src_shift : i],  i < n  (index arrays: int64, device)
 * gather (next_obs = store[minibatch + 1]: src_shift = 1), scatter (store <- a freshly decoded minibatch), plain copy. */
int srlz_copy_frames_u8(const uint8_t* src, const long long* src_index, long long src_shift, uint8_t* dst,
                        const long long* dst_index, long long dst_shift, int n, long long frame_bytes, srlz_stream_t stream);
/* The same between frames of different pitch, for a byte range inside the frame:
 *   dst[df][dst_offset .. dst_offset + copy_bytes) = src[sf][src_offset .. src_offset + copy_bytes),  frames src_frame_bytes / dst_frame_bytes apart.
 * The time-contrastive triplets of /root/reference/preprocessing/data_loader.py:219-243 are [view 1 ; view 2 ; view 1 of ANOTHER time step
 * of the same record] = 9 channels; the store keeps the two views of every time step (6 channels), so a triplet minibatch is two
 * gathers by index — the frame's own two views, and the negative's first — and no pixel of it is decoded again (round 5). */
int srlz_copy_frames_u8_strided(const uint8_t* src, const long long* src_index, long long src_shift, long long src_frame_bytes,
                                long long src_offset_bytes, uint8_t* dst, const long long* dst_index, long long dst_shift,
                                long long dst_frame_bytes, long long dst_offset_bytes, int n, long long copy_bytes, srlz_stream_t stream);
/* The DAE loader's occluded copies (preprocessing/data_loader.py:100-111) made on the device from resident frames: out [n,C,W,H] fp32 =
 * the normalised frame store[index[i] + shift] with the rectangle rects[i][view] = (h1, h2, w1, w2) set to 0, one rectangle per
 * camera view (group of 3 channels); the rectangles are drawn by the loader process with the reference's np.random calls. */
int srlz_occlude_frames_u8(const uint8_t* store, const long long* index, long long shift, const int* rects, const float* norm_lut,
                           float* out, int n, int c, int w, int h, srlz_stream_t stream);

/* SRLModulesSplit.detachSplit (models/modules.py:191-236): the state rebuilt from zero blocks and ONE kept slice is a
 * column mask: y[r][c] = x[r][c] for lo <= c < hi, else 0.  Its backward is the same call on the gradient. */
int srlz_mask_columns(const float* x, float* y, int rows, int cols, int lo, int hi, srlz_stream_t stream);

/* l1Loss / l2Loss (losses/losses.py:132-155) over the list of regularised parameters (LossManager.reg_params,
 * losses.py:30-31).  ptrs[nseg] / lens[nseg]: DEVICE arrays with the device address and element count of each tensor.
 * mode 0: norms[i] = sum |p_i|,  out = scale * sum_i norms[i]           (l1: scale = 1)
 * mode 1: norms[i] = ||p_i||_2,  out = scale * sum_i norms[i]           (l2: scale = 1/nseg)
 * fp64 accumulation, one workgroup per tensor, fixed order. */
int srlz_param_norms(const float* const* ptrs, const long long* lens, int nseg, int mode, float scale, float* norms,
                     float* out, srlz_stream_t stream);
/* gradient of the above into gptrs[i] (same shapes): coef * sign(p) (mode 0) or coef * p / norms[i] (mode 1),
 * coef = coef_dev[0] * scale (coef_dev may be NULL = 1). */
int srlz_param_norms_grad(const float* const* ptrs, float* const* gptrs, const long long* lens, int nseg, int mode,
                          const float* norms, const float* coef_dev, float scale, srlz_stream_t stream);

/* Gradient delivery.  The weight-gradient kernels write each parameter's k-th contribution of a backward pass into
 * stage k (stages = nstage copies of the flat bucket, each n floats); this call folds them into the bucket,
 * grad[i] = ((grad[i] + stage0[i]) + stage1[i]) ..., and clears the stages.  One launch replaces autograd's per-parameter,
 * per-contribution `grad += new` kernels (AccumulateGrad behind loss.backward(), models/learner.py:487-489). */
int srlz_fold_grads(float* grad, float* stages, long long n, int nstage, srlz_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Adam over one flat parameter buffer — th.optim.Adam(params, lr) models/learner.py:199,495 (torch defaults).
 * step is 1-based; grad_scale multiplies g first (1/world_size after the RCCL sum).  The hyper-parameters are doubles, as
 * torch holds them: lr / (1 - beta1^step), sqrt(1 - beta2^step) and (1 - beta) are evaluated in double and rounded once.
 * ------------------------------------------------------------------------------------------------------------ */
int srlz_adam_step(float* p, const float* g, float* m, float* v, long long n, double lr, double beta1, double beta2,
                   double eps, int step, float grad_scale, srlz_stream_t stream);
/* The same update with the step count kept ON THE DEVICE (step_dev[0] is incremented first; bc_dev = 2 floats of scratch):
 * the form a captured hipGraph can replay, since a kernel argument frozen at capture time cannot advance. */
int srlz_adam_step_dev(float* p, const float* g, float* m, float* v, long long n, double lr, double beta1, double beta2,
                       double eps, int* step_dev, float* bc_dev, float grad_scale, srlz_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Data-parallel exchange over RCCL (xGMI): one process per GPU, ONE in-place sum all-reduce of the flat fp32 bucket per
 * training step (gradients + scalar tail).  New — the reference is single-device (models/learner.py:65,187); the call
 * sits between loss.backward() (learner.py:489) and optimizer.step() (learner.py:495).
 * Rank 0 calls srlz_comm_unique_id() and shares the srlz_comm_unique_id_bytes() bytes with every rank (any host
 * rendezvous); every rank then calls srlz_comm_init() once with its HIP device current.  RCCL is dlopen'ed on first use
 * (the copy already loaded into the process is reused).  id buffers are HOST memory.
 * ------------------------------------------------------------------------------------------------------------ */
size_t srlz_comm_unique_id_bytes(void);
int srlz_comm_unique_id(void* id_out_host);
int srlz_comm_init(const void* id_host, int rank, int world);
int srlz_comm_world(void); /* 0 before srlz_comm_init */
int srlz_comm_allreduce_f32(float* buf, long long n, srlz_stream_t stream);
int srlz_comm_destroy(void);

/* ------------------------------------------------------------------------------------------------------------
 * Debug / calibration hooks (not on the product path).
 * ------------------------------------------------------------------------------------------------------------ */
/* Host-only: dump the virtual-grid program of a 64->64 convolution (tests interpret it on the CPU).
 * out: N,PH,PW,ss,Hs,Ws,ds,Hd,Wd,min_off,span,s2, then 9 x {src class, dst class, flat offset, weight tap}. */
int srlz_conv64_debug_program(const srlz_conv64_desc* d, int backward_data, int* out, int cap);
/* Back-to-back fp32 MFMA on random operands (no memory traffic): the matrix rate the chip sustains under load. */
int srlz_debug_mfma_peak(float* out, int blocks, int iters, srlz_stream_t stream);
/* The same MFMA stream with valu_per_mfma (0, 1, 2, 4, 8, 16) independent instructions of `kind` (0 v_fma_f32, 1 v_add_u32,
 * 2 v_pk_fma_f32, 3 v_mov_b32) behind every MFMA — in the same wave, or (split) in the SIMD's second wave: what an
 * instruction next to fp32 MFMAs costs on this chip (profiles/NOTES.md 5.3).  out: blocks * 512 floats. */
int srlz_debug_mfma_valu(float* out, int blocks, int iters, int valu_per_mfma, int kind, int split, srlz_stream_t stream);
/* Debug: out[4*b..] = {XCC id, HW_ID register, start clock, end clock} of workgroup b of a launch whose workgroups each
 * hold lds_bytes of LDS and spin for `spin` ticks (how the dispatcher places / replaces co-resident workgroups). */
int srlz_debug_placement(unsigned* out, int blocks, int lds_bytes, int spin, srlz_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SRLZ_H */
