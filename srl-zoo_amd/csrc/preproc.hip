// preproc.hip — input pipeline tail on the device: uint8 frames -> normalised fp32 tensors in the reference's layout.
// Replaces, for frames that are already decoded and resized, the host arithmetic of
// /root/reference/preprocessing/utils.py:20-32 (x/255, ImageNet mean/std per RGB channel) and the layout change of
// /root/reference/preprocessing/data_loader.py:255 (transpose(0,3,2,1): (H,W,C) image -> [C,W,H] tensor).
// Moving uint8 over PCIe instead of fp32 cuts the per-step host->device traffic 4x.
// Arithmetic order and rounding are the host's: ((x / 255) - mean) / std in fp32 with IEEE division, no FMA
// contraction -> bit-identical to the numpy result.
#include "common.h"

namespace {

// block = 256 threads handles a 32 (h) x 32 (w) pixel tile of one image, all C channels, through LDS.
__global__ __launch_bounds__(256) void normalize_u8_kernel(const uint8_t* __restrict__ img, float* __restrict__ out, int N,
                                                          int H, int W, int C) {
  __shared__ float tile[9][32][33];  // [c][h][w], padded
  const int tiles_w = (W + 31) / 32, tiles_h = (H + 31) / 32;
  const int n = blockIdx.x / (tiles_h * tiles_w);
  const int trem = blockIdx.x - n * (tiles_h * tiles_w);
  const int h0 = (trem / tiles_w) * 32, w0 = (trem % tiles_w) * 32;
  const float mean[3] = {0.485f, 0.456f, 0.406f};
  const float stdv[3] = {0.229f, 0.224f, 0.225f};
  // load: a tile row is 32*C contiguous bytes
  const int row_bytes = 32 * C;
  for (int idx = threadIdx.x; idx < 32 * row_bytes; idx += 256) {
    const int hl = idx / row_bytes, r = idx - hl * row_bytes;
    const int wl = r / C, c = r - wl * C;
    const int h = h0 + hl, w = w0 + wl;
    float v = 0.f;
    if (h < H && w < W) {
      const float x = (float)img[((size_t)(n * H + h) * W + w) * C + c];
      const float q = __fdiv_rn(x, 255.0f);
      v = __fdiv_rn(__fsub_rn(q, mean[c % 3]), stdv[c % 3]);
    }
    tile[c][hl][wl] = v;
  }
  __syncthreads();
  // store: out[n][c][w][h], h contiguous
  for (int idx = threadIdx.x; idx < C * 32 * 32; idx += 256) {
    const int hl = idx & 31, wl = (idx >> 5) & 31, c = idx >> 10;
    const int h = h0 + hl, w = w0 + wl;
    if (h < H && w < W) out[(((size_t)n * C + c) * W + w) * H + h] = tile[c][hl][wl];
  }
}

// lut[c][v] = ((v / 255) - mean[c]) / std[c], v = 0..255, c = R, G, B: the same three roundings as normalize_u8_kernel
__global__ void normalize_lut_kernel(float* __restrict__ lut) {
  const float mean[3] = {0.485f, 0.456f, 0.406f};
  const float stdv[3] = {0.229f, 0.224f, 0.225f};
  const int v = threadIdx.x, c = blockIdx.x;
  lut[c * 256 + v] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)v, 255.0f), mean[c]), stdv[c]);
}

// planar frames [N*C planes][plane] uint8 -> float through the table (16 bytes in, 4 x 16 bytes out per thread and step)
__global__ __launch_bounds__(256) void normalize_u8_planar_kernel(const uint8_t* __restrict__ x, const float* __restrict__ lut,
                                                                 float* __restrict__ out, int C, long long plane) {
  __shared__ float tab[256];
  const int c = (blockIdx.y % C) % 3;
  tab[threadIdx.x] = lut[c * 256 + threadIdx.x];
  __syncthreads();
  const uint8_t* src = x + (size_t)blockIdx.y * plane;
  float* dst = out + (size_t)blockIdx.y * plane;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < plane; i += (long long)gridDim.x * 256) dst[i] = tab[src[i]];
}

// ---- the dataset resident in HBM (round 4): whole uint8 frames moved by index --------------------------------
// dst frame (dst_index ? dst_index[i] : i) <- src frame (src_index ? src_index[i] + src_shift : i); one frame per blockIdx.y,
// 16 bytes per thread and step.  Gather (minibatch <- store), scatter (store <- freshly decoded minibatch) and plain copy.
__global__ __launch_bounds__(256) void copy_frames_kernel(const uint8_t* __restrict__ src, const long long* __restrict__ src_index,
                                                         long long src_shift, uint8_t* __restrict__ dst,
                                                         const long long* __restrict__ dst_index, long long dst_shift,
                                                         long long words) {
  const long long i = blockIdx.y;
  const long long sf = src_index ? src_index[i] + src_shift : i, df = dst_index ? dst_index[i] + dst_shift : i;
  const uint4* __restrict__ s = (const uint4*)src + sf * words;
  uint4* __restrict__ d = (uint4*)dst + df * words;
  for (long long w = (long long)blockIdx.x * 256 + threadIdx.x; w < words; w += (long long)gridDim.x * 256) d[w] = s[w];
}

// The same with different frame pitches on the two sides and a byte offset inside the frame: `words` 16-byte words of
// src frame sf (pitch src_pitch words, starting src_off words in) -> dst frame df (pitch dst_pitch, starting dst_off).  What assembles the
// 9-channel triplet observation [view 1 ; view 2 ; view 1 of the negative] from a store of 6-channel frames, and what keeps the
// first two views of a freshly decoded 9-channel minibatch.
__global__ __launch_bounds__(256) void copy_frames_strided_kernel(const uint8_t* __restrict__ src, const long long* __restrict__ src_index,
                                                                 long long src_shift, long long src_pitch, long long src_off,
                                                                 uint8_t* __restrict__ dst, const long long* __restrict__ dst_index,
                                                                 long long dst_shift, long long dst_pitch, long long dst_off,
                                                                 long long words) {
  const long long i = blockIdx.y;
  const long long sf = src_index ? src_index[i] + src_shift : i, df = dst_index ? dst_index[i] + dst_shift : i;
  const uint4* __restrict__ s = (const uint4*)src + sf * src_pitch + src_off;
  uint4* __restrict__ d = (uint4*)dst + df * dst_pitch + dst_off;
  for (long long w = (long long)blockIdx.x * 256 + threadIdx.x; w < words; w += (long long)gridDim.x * 256) d[w] = s[w];
}

// The DAE's occluded copy of a frame taken from the store (reference preprocessing/data_loader.py:100-111: the NORMALISED image
// with a random rectangle set to 0): out[i][c][w][h] = (h1 <= h < h2 && w1 <= w < w2) ? 0 : lut[c % 3][store[index[i]+shift][c][w][h]],
// one rectangle (h1, h2, w1, w2) per frame and camera view (group of 3 channels), drawn by the loader process.
__global__ __launch_bounds__(256) void occlude_frames_kernel(const uint8_t* __restrict__ store, const long long* __restrict__ index,
                                                            long long shift, const int* __restrict__ rects,
                                                            const float* __restrict__ lut, float* __restrict__ out, int C, int Wd,
                                                            int Hd) {
  __shared__ float tab[256];
  const long long i = blockIdx.y / C;
  const int c = (int)(blockIdx.y % C);
  tab[threadIdx.x] = lut[(c % 3) * 256 + threadIdx.x];
  __syncthreads();
  const int* r = rects + (i * (C / 3) + c / 3) * 4;
  const int h1 = r[0], h2 = r[1], w1 = r[2], w2 = r[3];
  const long long plane = (long long)Wd * Hd;
  const uint8_t* __restrict__ src = store + ((index[i] + shift) * C + c) * plane;
  float* __restrict__ dst = out + (size_t)blockIdx.y * plane;
  for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < plane; k += (long long)gridDim.x * 256) {
    const int w = (int)(k / Hd), h = (int)(k - (long long)w * Hd);
    dst[k] = (h >= h1 && h < h2 && w >= w1 && w < w2) ? 0.f : tab[src[k]];
  }
}

}  // namespace

extern "C" int srlz_copy_frames_u8(const uint8_t* src, const long long* src_index, long long src_shift, uint8_t* dst,
                                   const long long* dst_index, long long dst_shift, int n, long long frame_bytes,
                                   srlz_stream_t stream) {
  SRLZ_REQUIRE(src && dst, SRLZ_ERR_NULL, "copy_frames_u8: null pointer");
  SRLZ_REQUIRE(n > 0 && n <= 65535 && frame_bytes > 0 && frame_bytes % 16 == 0, SRLZ_ERR_BAD_DESC,
               "copy_frames_u8: 1..65535 frames of a multiple of 16 bytes (got %d x %lld)", n, frame_bytes);
  SRLZ_REQUIRE((((uintptr_t)src) & 15) == 0 && (((uintptr_t)dst) & 15) == 0, SRLZ_ERR_BAD_DESC, "copy_frames_u8: unaligned buffer");
  const long long words = frame_bytes / 16;
  int gx = (int)((words + 255) / 256);
  if (gx > 16) gx = 16;
  hipLaunchKernelGGL(copy_frames_kernel, dim3(gx, n), dim3(256), 0, as_stream(stream), src, src_index, src_shift, dst, dst_index,
                     dst_shift, words);
  SRLZ_LAUNCHED();
  return 0;
}

extern "C" int srlz_copy_frames_u8_strided(const uint8_t* src, const long long* src_index, long long src_shift,
                                           long long src_frame_bytes, long long src_offset_bytes, uint8_t* dst,
                                           const long long* dst_index, long long dst_shift, long long dst_frame_bytes,
                                           long long dst_offset_bytes, int n, long long copy_bytes, srlz_stream_t stream) {
  SRLZ_REQUIRE(src && dst, SRLZ_ERR_NULL, "copy_frames_u8_strided: null pointer");
  SRLZ_REQUIRE(n > 0 && n <= 65535 && copy_bytes > 0 && copy_bytes % 16 == 0 && src_frame_bytes % 16 == 0 && dst_frame_bytes % 16 == 0 &&
                   src_offset_bytes % 16 == 0 && dst_offset_bytes % 16 == 0,
               SRLZ_ERR_BAD_DESC, "copy_frames_u8_strided: 1..65535 frames; sizes, pitches and offsets in multiples of 16 bytes (got %d x %lld)",
               n, copy_bytes);
  SRLZ_REQUIRE(src_offset_bytes >= 0 && dst_offset_bytes >= 0 && src_offset_bytes + copy_bytes <= src_frame_bytes &&
                   dst_offset_bytes + copy_bytes <= dst_frame_bytes,
               SRLZ_ERR_BAD_DESC, "copy_frames_u8_strided: [offset, offset + %lld) must lie inside a frame (%lld+ of %lld -> %lld+ of %lld)",
               copy_bytes, src_offset_bytes, src_frame_bytes, dst_offset_bytes, dst_frame_bytes);
  SRLZ_REQUIRE((((uintptr_t)src) & 15) == 0 && (((uintptr_t)dst) & 15) == 0, SRLZ_ERR_BAD_DESC, "copy_frames_u8_strided: unaligned buffer");
  const long long words = copy_bytes / 16;
  int gx = (int)((words + 255) / 256);
  if (gx > 16) gx = 16;
  hipLaunchKernelGGL(copy_frames_strided_kernel, dim3(gx, n), dim3(256), 0, as_stream(stream), src, src_index, src_shift, src_frame_bytes / 16,
                     src_offset_bytes / 16, dst, dst_index, dst_shift, dst_frame_bytes / 16, dst_offset_bytes / 16, words);
  SRLZ_LAUNCHED();
  return 0;
}

extern "C" int srlz_occlude_frames_u8(const uint8_t* store, const long long* index, long long shift, const int* rects,
                                      const float* norm_lut, float* out, int n, int c, int w, int h, srlz_stream_t stream) {
  SRLZ_REQUIRE(store && index && rects && norm_lut && out, SRLZ_ERR_NULL, "occlude_frames_u8: null pointer");
  SRLZ_REQUIRE(n > 0 && c > 0 && c <= 9 && c % 3 == 0 && (long long)n * c <= 65535 && w > 0 && h > 0, SRLZ_ERR_BAD_DESC,
               "occlude_frames_u8: channels must be 3, 6 or 9 (got %d) and n * c <= 65535", c);
  int gx = (w * h + 255) / 256;
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(occlude_frames_kernel, dim3(gx, n * c), dim3(256), 0, as_stream(stream), store, index, shift, rects, norm_lut, out,
                     c, w, h);
  SRLZ_LAUNCHED();
  return 0;
}

extern "C" int srlz_normalize_u8(const uint8_t* img_nhwc, float* out_ncwh, int n, int h, int w, int c,
                                 srlz_stream_t stream) {
  SRLZ_REQUIRE(img_nhwc && out_ncwh, SRLZ_ERR_NULL, "normalize_u8: null pointer");
  SRLZ_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && c <= 9 && c % 3 == 0, SRLZ_ERR_BAD_DESC,
               "normalize_u8: channels must be 3, 6 or 9 (got %d)", c);
  const int tiles = n * ((h + 31) / 32) * ((w + 31) / 32);
  hipLaunchKernelGGL(normalize_u8_kernel, dim3(tiles), dim3(256), 0, as_stream(stream), img_nhwc, out_ncwh, n, h, w, c);
  SRLZ_LAUNCHED();
  return 0;
}

extern "C" int srlz_normalize_lut(float* lut, srlz_stream_t stream) {
  SRLZ_REQUIRE(lut, SRLZ_ERR_NULL, "normalize_lut: null pointer");
  hipLaunchKernelGGL(normalize_lut_kernel, dim3(3), dim3(256), 0, as_stream(stream), lut);
  SRLZ_LAUNCHED();
  return 0;
}

extern "C" int srlz_normalize_u8_planar(const uint8_t* x_u8, const float* norm_lut, float* out, int n, int c, long long plane,
                                        srlz_stream_t stream) {
  SRLZ_REQUIRE(x_u8 && norm_lut && out, SRLZ_ERR_NULL, "normalize_u8_planar: null pointer");
  SRLZ_REQUIRE(n > 0 && plane > 0 && c > 0 && c <= 9 && c % 3 == 0 && (long long)n * c <= 65535, SRLZ_ERR_BAD_DESC,
               "normalize_u8_planar: channels must be 3, 6 or 9 (got %d) and n * c <= 65535", c);
  int gx = (int)((plane + 255) / 256);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(normalize_u8_planar_kernel, dim3(gx, n * c), dim3(256), 0, as_stream(stream), x_u8, norm_lut, out, c, plane);
  SRLZ_LAUNCHED();
  return 0;
}
