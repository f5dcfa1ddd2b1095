// skinny.hip — the two image-side layers: one operand has C in {3,6,9} channels stored NCHW (the reference's image
// layout), the other 64 channels NHWC.  Both are stride-2 and share one formulation:
//   kind 0: nn.Conv2d(C,64,k=7,s=2,p=3,bias=False)    /root/reference/models/models.py:49   (K=7, PAD=3)
//   kind 1: nn.ConvTranspose2d(64,C,k=4,s=2)          /root/reference/models/models.py:82   (K=4, PAD=0)
// Kernels
//   skinny_conv_kernel<K,PAD>   feat[pix][64] = im2col(img)[pix][3*K*K] . W        kind0 forward, kind1 data-grad
//   skinny_wgrad_kernel<K,PAD>  dW[64][3*K*K] = sum_pix feat[pix][64] (x) im2col(img)[pix][:]  kind0/kind1 weight-grad
//   convT_out_kernel            img = col2im(feat[pix][64] . W[64][C*16]) + bias   kind1 forward
// The image tile lives in LDS with even/odd columns de-interleaved (stride-2 taps become unit-stride, conflict-free
// ds_read_b32) and is re-used by all taps; im2col is never materialised.  fp32 MFMA 32x32x2 (16x16x4 for the
// 48-column convT GEMM).  Channels are processed in groups of 3 (C=6/9 multi-view loops over groups).
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int XP = 24;  // LDS row pitch (floats) of one half-row; 2*XP % 32 == 16 -> two tile rows never collide

template <int K, int TR = 16>
struct Geo {
  static constexpr int ROWS = 2 * (TR - 1) + K;    // image rows/cols feeding a TR x 16 output tile
  static constexpr int COLS = 30 + K;
  static constexpr int KT = 3 * K * K;             // taps per channel group
  static constexpr int KS = (KT + 1) / 2;          // MFMA k-steps (2 taps per step)
  // pitch of one (channel, column-parity) plane: ROWS*XP rounded so that PP % 32 == 4 — the weight-gradient kernel
  // reads 32 different taps per instruction, and a pitch congruent to XP (24) folds them onto 4 banks
  static constexpr int PP = ((ROWS * XP + 31) / 32) * 32 + 4;
  static constexpr int TILE_FLOATS = 3 * 2 * PP;
};

// Optional fusion for conv1's weight gradient: its feature-side operand dy1 = d(loss)/d(conv1 output) is the output of the
// BatchNorm + ReLU + MaxPool backward and has NO other consumer (conv1 needs no data gradient), so instead of writing
// dy1 (0.82 GB per 256 images) and reading it back, the staging code of the wgrad kernel computes it on the fly from
// y1, the pooling argmax, the pooled-output gradient and the two BatchNorm-backward sums.
struct PoolFuse {
  const float* y;          // raw conv output [N,HF,WF,64]; NULL = fusion off
  const uint8_t* argmax;   // [N,HP,WP,64] window index of the maximum
  const float* dpooled;    // [N,HP,WP,64]
  const float* bnp;        // mean / invstd / scale / shift
  const float* sums;       // [128] sum dz, sum dz*xhat
  int HP, WP, pad, training;
  float inv_count;
};

// A persistent workgroup walks the virtual sequence v = blockIdx.x, blockIdx.x + gridDim.x, ...; the tile it works on is
// xcd_remap(v): block b runs on XCD b % 8, so every XCD gets ONE contiguous run of tiles and its workgroups process neighbouring
// tiles at the same time — the image windows of neighbouring tiles overlap (37 rows / columns for 32 at conv1) and share cache
// lines, and walked in plain order (tile = v) neighbours sit on different XCDs, each fetching those lines into its own L2
// (rocprofv3: conv1 forward read 884 MB for a 308 MB input).  v >= ntiles -> ntiles ("no tile").
__device__ __forceinline__ int vtile(int v, int ntiles) { return v < ntiles ? xcd_remap(v, ntiles) : ntiles; }

template <int K, int TR = 16>
__host__ __device__ constexpr int koff(int k) {
  // LDS offset of tap k = (c,ky,kx) relative to the pixel base (2*ty*XP + tx)
  const int kk = (k < Geo<K, TR>::KT) ? k : 0;
  const int c = kk / (K * K), ky = (kk / K) % K, kx = kk % K;
  return (c * 2 + (kx & 1)) * Geo<K, TR>::PP + ky * XP + (kx >> 1);
}

// Staging of the image window of an output tile, split in two (request into registers / land in LDS) so that callers can
// software-pipeline it; a synchronous staging is a request followed by a land.
// Element j of a thread is window element idx = tid + 256*j = (channel c, window row, window column xl); its decomposition is
// tile-independent and is kept PACKED in one register per element (pk, set once by image_index).  Left to itself the compiler
// hoists the unpacked (c, row, xl, LDS offset) of every element out of the caller's persistent tile loop — ~3 registers per
// element for the whole kernel, which is what pushed these kernels into scratch; the opaque copy below keeps the (cheap) unpacking
// inside the loop.
template <int K, int TR = 16>
struct ImgRegs {
  static constexpr int TOT = 3 * Geo<K, TR>::ROWS * Geo<K, TR>::COLS;
  static constexpr int PER = (TOT + 255) / 256;
  int pk[PER];       // (LDS offset of the element << 14) | (c << 12) | (row << 6) | xl, or -1 past the end of the window
  float v[PER];      // raw loads (from a clamped address when the element is outside the image)
  unsigned inside;   // bit j: element j is inside the image; applied when the value LANDS — a select (or a branch merge) at the
                     // request would make the compiler wait for the load right there, and nothing would travel under the MFMAs
};

// XPV / PPV: row pitch and (channel, column-parity) plane pitch of the LDS tile the elements will land in (image_land): the element's
// offset in it is tile-independent too and rides in the upper bits — one shift at the landing instead of eight instructions
template <int K, int TR = 16, int XPV = XP, int PPV = Geo<K, TR>::PP>
__device__ __forceinline__ void image_index(ImgRegs<K, TR>& r) {
  constexpr int ROWS = Geo<K, TR>::ROWS, COLS = Geo<K, TR>::COLS, TOT = ImgRegs<K, TR>::TOT;
  static_assert(6 * PPV < (1 << 17), "the LDS offset has 17 bits in pk");
#pragma unroll
  for (int j = 0; j < ImgRegs<K, TR>::PER; ++j) {
    const int idx = threadIdx.x + 256 * j;
    const int c = idx / (ROWS * COLS);
    const int rem = idx - c * (ROWS * COLS);
    const int row = rem / COLS, xl = rem - row * COLS;
    const int dst = (c * 2 + (xl & 1)) * PPV + row * XPV + (xl >> 1);
    r.pk[j] = idx < TOT ? ((dst << 14) | (c << 12) | (row << 6) | xl) : -1;
  }
}

// (channel, window row, window column) of element j; false past the end of the window.
// HOIST: straight from the thread index — tile-independent, so the compiler keeps all of it in registers across the caller's tile
// loop; otherwise unpacked from pk behind an opaque copy, i.e. recomputed at every use.
template <int K, bool HOIST, int TR = 16>
__device__ __forceinline__ bool image_decode(const ImgRegs<K, TR>& r, int j, int& c, int& row, int& xl) {
  if (HOIST) {
    constexpr int ROWS = Geo<K, TR>::ROWS, COLS = Geo<K, TR>::COLS;
    const int idx = threadIdx.x + 256 * j;
    c = idx / (ROWS * COLS);
    const int rem = idx - c * (ROWS * COLS);
    row = rem / COLS; xl = rem - row * COLS;
    return idx < ImgRegs<K, TR>::TOT;
  }
  int pk = r.pk[j];
  asm volatile("" : "+v"(pk));
  c = (pk >> 12) & 3; row = (pk >> 6) & 63; xl = pk & 63;
  return pk >= 0;
}

// HOIST = true: the unpacked decomposition may live in registers for the whole kernel (no opaque copy): ~35 more registers, ~270
// fewer VALU instructions per tile — worth it where the registers exist (conv1 forward: 1.10 vs 1.16 ms).
// PT = uint8_t: the frames are still the loader's bytes ([N,C,W,H] planar, /root/reference/preprocessing/data_loader.py:255 applied
// to the uint8 image); the raw byte travels in the register and is normalised when it lands (image_land<U8>).
template <int K, int PAD, bool HOIST = false, int TR = 16, typename PT = float>
__device__ __forceinline__ void image_request(ImgRegs<K, TR>& r, const PT* __restrict__ img, int n, int C, int cg, int H,
                                              int W, int oy0, int ox0, bool valid) {
  const int iy0 = 2 * oy0 - PAD, ix0 = 2 * ox0 - PAD;
  const int plane0 = n * C + cg * 3;  // (uniform)
  r.inside = 0;
#pragma unroll
  for (int j = 0; j < ImgRegs<K, TR>::PER; ++j) {
    int c, row, xl;
    const bool live = image_decode<K, HOIST, TR>(r, j, c, row, xl);
    const int iy = iy0 + row, ix = ix0 + xl;
    // every instruction here is paid for in MFMA time: 32-bit element offsets from the uniform base (the host checks
    // that the tensor has fewer than 2^31 elements), one unsigned compare per axis, a select instead of a masked 64-bit address chain
    const bool ok = valid && live && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    const unsigned off = ((unsigned)(plane0 + c) * (unsigned)H + (unsigned)iy) * (unsigned)W + (unsigned)ix;
    const PT raw = img[ok ? off : 0u];
    if constexpr (sizeof(PT) == 1) r.v[j] = __uint_as_float((unsigned)raw);  // (the byte load zero-extends: no instruction, no wait)
    else r.v[j] = raw;
    r.inside |= (ok ? 1u : 0u) << j;
  }
}

// `gain`: factor applied to every element on its way into LDS (1 = none; the fused ConvTranspose-5 backward turns the stored
// reconstruction error dec - obs into d(loss)/d(dec) = gain * (dec - obs) here, see convT_out_kernel<true>)
// U8: the registers hold raw bytes; the value that lands is lut[c][byte] = ((byte / 255) - mean[c]) / std[c], the three fp32
// roundings of /root/reference/preprocessing/utils.py:20-32 tabulated once by srlz_normalize_lut (768 floats, L1-resident) — a
// table because two IEEE divisions per element cost more VALU time than the rest of the landing, and because it makes the
// bit-for-bit equality with the separate normalisation kernel a matter of construction.
// XPV / PPV: row pitch and (channel, column-parity) plane pitch of T (the conv1 weight gradient has its own, see TAP7).
template <int K, bool HOIST = false, int TR = 16, bool U8 = false, int XPV = XP, int PPV = Geo<K, TR>::PP>
__device__ __forceinline__ void image_land(float* __restrict__ T, const ImgRegs<K, TR>& r, const float gain = 1.f,
                                           const float* __restrict__ lut = nullptr) {
  float val[ImgRegs<K, TR>::PER];
  if constexpr (U8) {
#pragma unroll
    for (int j = 0; j < ImgRegs<K, TR>::PER; ++j) {  // all look-ups requested before the first one is used
      int c, row, xl;
      const bool live = image_decode<K, HOIST, TR>(r, j, c, row, xl);
      val[j] = lut[(live ? c : 0) * 256 + (int)__float_as_uint(r.v[j])];
    }
  } else {
#pragma unroll
    for (int j = 0; j < ImgRegs<K, TR>::PER; ++j) val[j] = r.v[j] * gain;
  }
#pragma unroll
  for (int j = 0; j < ImgRegs<K, TR>::PER; ++j) {
    int c, row, xl;
    const bool live = image_decode<K, HOIST, TR>(r, j, c, row, xl);
    // (branch-free — elements past the end of the window go to a spare float of the first plane's padding: a wait inside a branch
    // leaves the compiler unsure, at the join, whether the load has landed, and it then waits for EVERYTHING at the next re-use
    // of the register, including the loads meant to stay in flight)
    int dst;
    if constexpr (HOIST) dst = (c * 2 + (xl & 1)) * PPV + row * XPV + (xl >> 1);
    else { dst = r.pk[j]; asm volatile("" : "+v"(dst)); dst >>= 14; }  // (image_index was given this tile's pitches)
    T[live ? dst : PPV - 1] = ((r.inside >> j) & 1u) ? val[j] : 0.f;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// feat[n,oy,ox,:] = sum_{c,ky,kx} img[n,c,2oy-PAD+ky,2ox-PAD+kx] * w_ref[:,c,ky,kx]      (w_ref: [64][C][K][K])
// 256 threads, tile = 16x16 output pixels; wave w owns tile rows 4w..4w+3 (two 32-pixel M-tiles) x 64 channels.
// Persistent over tiles; for C == 3 the 64 x KT weight matrix is staged in LDS once per workgroup.
// ------------------------------------------------------------------------------------------------------------------
template <int K, int PAD, bool BNBWD = false, bool MULTI = false, typename PT = float>
__global__ __launch_bounds__(256, 2) void skinny_conv_kernel(const PT* __restrict__ img,
                                                            const float* __restrict__ w_ref,
                                                            float* __restrict__ feat,
                                                            float* __restrict__ stats_partial, int N, int C, int H,
                                                            int W, int HF, int WF, int tiles_y, int tiles_x,
                                                            const float* __restrict__ y_raw,
                                                            const float* __restrict__ y_bnp, int npg,
                                                            const float* __restrict__ lut = nullptr) {
  constexpr bool U8 = sizeof(PT) == 1;  // uint8 frames + the normalisation table (image_land)
  // npg = images per BatchNorm group (N when there is one group): image n uses the record y_bnp[(n / npg) * 256 ..]
  // BNBWD (data-gradient use, kind 1; y_raw != NULL): `feat` is dA = d(loss)/d(relu(bn(y_raw))) and the per-tile partials become
  // the two BatchNorm-backward sums  sum dz  and  sum dz*xhat  (dz = dA*[bn(y)>0]) instead of  sum v  and  sum v^2 —
  // the separate pass that re-reads dA and y (srlz_bn_relu_bwd_sums) disappears.
  constexpr int KT = Geo<K>::KT, KS = Geo<K>::KS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* T = (float*)smem;                      // image window
  float* Wl = T + Geo<K>::TILE_FLOATS;          // [KS][2][64]
  float* red = Wl + KS * 128;                   // [4][128]
  // PIPE (every use but the one that also prefetches y_raw, which has no registers to spare): the image window of the NEXT
  // (tile, channel group) is requested into registers right after the barrier that publishes the current one, travels under
  // the MFMA loop and lands in T after the barrier that ends it, so no HBM round trip sits between two barriers; the
  // epilogue then transposes through its own 16 KB (E) instead of through T.
  constexpr bool PIPE = !BNBWD;
  constexpr bool HOIST = (K == 7) && !BNBWD;  // the LDS offsets of the landing are kept in registers (see image_decode); the request
                                              // side (3 values per element) unpacks from pk — hoisting both spills
  float* E = PIPE ? red + 512 : T;              // [4 waves][16 pixels][64 channels]
  if constexpr (U8) {  // the normalisation table moves into LDS (PIPE kernels only: E is their own 16 KB)
    float* L = E + 4096;
    for (int i = threadIdx.x; i < 768; i += 256) L[i] = lut[i];
    lut = L;
    __syncthreads();  // (the first window lands before the tile loop's first barrier)
  }

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int ncg = C / 3;
  const int ntiles = N * tiles_y * tiles_x;

  auto stage_weights = [&](int cg) {
    for (int idx = tid; idx < KS * 128; idx += 256) {
      const int co = idx & 63, sh = idx >> 6;
      const int k = (sh & 1) * KS + (sh >> 1);
      float v = 0.f;
      if (k < KT) v = w_ref[((size_t)co * C + cg * 3) * (K * K) + k];  // [co][cg*3 + c][ky][kx], k = (c*K+ky)*K+kx
      Wl[idx] = v;
    }
  };
  // MULTI (C = 6 / 9: more than one group of 3 image channels) re-stages the weights per group inside the tile loop; its own
  // instantiation, because the staging loop's loads in the middle of the pipeline cost the single-group kernel a full
  // "s_waitcnt vmcnt(0)" in front of every MFMA loop (the compiler merges the two paths conservatively)
  if (!MULTI) stage_weights(0);

  const int tx = l31 & 15, tyl = l31 >> 4;
  const int pb0 = 2 * (wave * 4 + tyl) * XP + tx;        // M-tile 0: tile rows 4w, 4w+1
  const int pb1 = pb0 + 4 * XP;                          // M-tile 1: tile rows 4w+2, 4w+3
  const float* wl0 = Wl + h * 64 + l31;

  ImgRegs<K> nx;
  image_index<K>(nx);
  if (PIPE && (int)blockIdx.x < ntiles) {  // the first window is staged the plain way
    const int t0 = vtile(blockIdx.x, ntiles);
    const int n = t0 / (tiles_y * tiles_x);
    const int trem = t0 - n * (tiles_y * tiles_x);
    image_request<K, PAD, false>(nx, img, n, C, 0, H, W, (trem / tiles_x) * 16, (trem % tiles_x) * 16, true);
    image_land<K, HOIST, 16, U8>(T, nx, 1.f, lut);
    if (MULTI) stage_weights(0);
  }
  for (int v = blockIdx.x; v < ntiles; v += gridDim.x) {
    const int tile = xcd_remap(v, ntiles);
    const int n = tile / (tiles_y * tiles_x);
    const int trem = tile - n * (tiles_y * tiles_x);
    const int oy0 = (trem / tiles_x) * 16, ox0 = (trem % tiles_x) * 16;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    // Epilogue layout: lane (g = lane >> 4, slot = lane & 15) owns channels [4*slot, 4*slot+4) of pixel column 4*it + g of tile
    // row  wave*4 + rowq  (rowq = 0..3, it = 0..3): every global access of the epilogue is a 16-byte one (a dword-per-lane
    // access pattern sustains only ~5 B/cycle/CU here — it was the bound of the ConvTranspose-5 data gradient).
    const int eg = lane >> 4, eslot = lane & 15;
    // BNBWD: this lane's 16 x float4 of y_raw are requested before the MFMA loop and consumed in the epilogue
    f32x4 yv[BNBWD ? 16 : 1];

    for (int cg = 0; cg < ncg; ++cg) {
      __syncthreads();
      int cg2 = cg + 1;
      if (PIPE) {
        int tile2 = tile;
        if (cg2 == ncg) { cg2 = 0; tile2 = vtile(v + gridDim.x, ntiles); }
        const int n2 = tile2 / (tiles_y * tiles_x);
        const int trem2 = tile2 - n2 * (tiles_y * tiles_x);
        image_request<K, PAD, false>(nx, img, n2, C, cg2, H, W, (trem2 / tiles_x) * 16, (trem2 % tiles_x) * 16, tile2 < ntiles);
      } else {
        image_request<K, PAD, false>(nx, img, n, C, cg, H, W, oy0, ox0, true);
        image_land<K, HOIST, 16, U8>(T, nx, 1.f, lut);
        if (MULTI) stage_weights(cg);
      }
      if (BNBWD && (!MULTI || cg == 0)) {  // (after the image loads: they are waited for first, these stay in flight behind the MFMAs)
#pragma unroll
        for (int rowq = 0; rowq < 4; ++rowq)
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const int oy = oy0 + wave * 4 + rowq, ox = ox0 + it * 4 + eg;
            // (branch-free: a pixel outside the feature map reads a clamped address; the epilogue never uses it)
            const bool ok = oy < HF && ox < WF;
            yv[BNBWD ? rowq * 4 + it : 0] = *(const f32x4*)(y_raw + (ok ? ((size_t)(n * HF + oy) * WF + ox) * 64 : (size_t)0) + eslot * 4);
          }
        __builtin_amdgcn_sched_barrier(0);  // (the scheduler would sink the requests below the MFMA loop, next to their first use)
      }
      if (!PIPE) __syncthreads();
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int ko = h ? koff<K>(KS + s) : koff<K>(s);
        const float a0 = T[pb0 + ko], a1 = T[pb1 + ko];
        const float b0 = wl0[s * 128], b1 = wl0[s * 128 + 32];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      }
      if (PIPE) {
        __syncthreads();  // every wave is done reading T (and Wl): the next window lands
        image_land<K, HOIST, 16, U8>(T, nx, 1.f, lut);
        if (MULTI) stage_weights(cg2);
      }
    }
    if (!PIPE) __syncthreads();  // every wave is done reading the image window: T becomes the transposition buffer (4 KB per wave)
    float* Ew = E + wave * 1024;  // [16 pixels of one tile row][64 channels]
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    f32x4 bmean = s1, binv = s1, bsc = s1, bsh = s1;
    if (BNBWD) {
      const float* __restrict__ yb = y_bnp + (n / npg) * 256;
      bmean = *(const f32x4*)(yb + eslot * 4); binv = *(const f32x4*)(yb + 64 + eslot * 4);
      bsc = *(const f32x4*)(yb + 128 + eslot * 4); bsh = *(const f32x4*)(yb + 192 + eslot * 4);
      // every load this epilogue consumes is waited for HERE, once: the stores below sit in branches, and at each join the compiler
      // no longer knows which loads have landed — it would put "s_waitcnt vmcnt(0)" behind every store (one HBM round trip each)
      asm volatile("" : "+v"(bmean), "+v"(binv), "+v"(bsc), "+v"(bsh));
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(yv[BNBWD ? i : 0]));
    }
#pragma unroll
    for (int rowq = 0; rowq < 4; ++rowq) {
      const int mt = rowq >> 1, half = rowq & 1;
      // accumulator register r = 8*half + rr of M-tile mt holds pixel (rr & 3) + 8*(rr >> 2) + 4*h of tile row wave*4 + rowq
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        const int il = (rr & 3) + 8 * (rr >> 2) + 4 * h;
        Ew[il * 64 + l31] = acc[mt][0][8 * half + rr];
        Ew[il * 64 + 32 + l31] = acc[mt][1][8 * half + rr];
      }
      // (wave-private region: the compiler's lgkmcnt wait orders these writes before the reads below)
      int oy = oy0 + wave * 4 + rowq;
      // (opaque: the 16 store addresses are the ones y_raw was prefetched from before the MFMA loop; kept alive across the loop
      // for re-use they cost 32 registers the BNBWD instantiation does not have)
      if (BNBWD) asm volatile("" : "+v"(oy));
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int pl = it * 4 + eg, ox = ox0 + pl;
        const f32x4 v = *(const f32x4*)(Ew + pl * 64 + eslot * 4);
        if (oy < HF && ox < WF) {
          *(f32x4*)(feat + ((size_t)(n * HF + oy) * WF + ox) * 64 + eslot * 4) = v;
          if (BNBWD) {
            const f32x4 yy = yv[BNBWD ? rowq * 4 + it : 0];
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (yy[e] * bsc[e] + bsh[e] > 0.f) { s1[e] += v[e]; s2[e] += v[e] * ((yy[e] - bmean[e]) * binv[e]); }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) { s1[e] += v[e]; s2[e] += v[e] * v[e]; }
          }
        }
      }
    }
    if (stats_partial) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s1[e] += __shfl_xor(s1[e], 16, 64); s1[e] += __shfl_xor(s1[e], 32, 64);
        s2[e] += __shfl_xor(s2[e], 16, 64); s2[e] += __shfl_xor(s2[e], 32, 64);
      }
      if (eg == 0) {
        *(f32x4*)(red + wave * 128 + eslot * 4) = s1;
        *(f32x4*)(red + wave * 128 + 64 + eslot * 4) = s2;
      }
      __syncthreads();
      if (tid < 128) stats_partial[(size_t)tile * 128 + tid] = red[tid] + red[128 + tid] + red[256 + tid] + red[384 + tid];
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// conv1 DATA gradient (only needed when the image itself carries a gradient: the frozen denoiser of the perceptual loss,
// /root/reference/losses/losses.py:217-236, reads a decoded image):
//   dx[n,c,iy,ix] = sum_{f,ky,kx} dy[n,oy,ox,f] * w_ref[f,c,ky,kx],   iy = 2*oy - 3 + ky,  ix = 2*ox - 3 + kx
// Same two-phase scheme as convT_out_kernel: per 16x16 IMAGE tile the 11x11 feature pixels that reach it are multiplied
// with W[64][3*49] (v_mfma_f32_16x16x4_f32, N padded to 160) into LDS, then each image pixel gathers its <= 4x4
// products.  Wave w owns the N-tiles {w, w+4, w+8} for all eight M-tiles, so its 48 B-fragment registers are loaded once.
// ------------------------------------------------------------------------------------------------------------------
constexpr int DG_NF = 11;                 // feature rows / columns reaching a 16x16 image tile (1 below, 2 above)
constexpr int DG_NPROD = 147;             // 3 channels x 49 taps
constexpr int DG_TP = 149;                // LDS pitch (floats) of one feature pixel's products

__global__ __launch_bounds__(256, 2) void conv1_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w_ref,
                                                            float* __restrict__ dx, int N, int C, int H, int W, int HF,
                                                            int WF, int tiles_y, int tiles_x) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* Tt = (float*)smem;  // [128][DG_TP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int cg = blockIdx.y;
  const int ntiles = N * tiles_y * tiles_x;

  // B fragments of this wave's N-tiles: column n = (wave + 4j)*16 + li = (co, tap); breg[j][c][r] = W[f = 16c+4kq+r][n]
  f32x4 breg[3][4];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int ncol = (wave + 4 * j) * 16 + li;
    const bool valid = (wave + 4 * j) < 10 && ncol < DG_NPROD;
    const int co = valid ? ncol / 49 : 0, tap = valid ? ncol % 49 : 0;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        breg[j][c][r] = valid ? w_ref[((size_t)(16 * c + 4 * kq + r) * C + cg * 3 + co) * 49 + tap] : 0.f;
  }

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int n = tile / (tiles_y * tiles_x);
    const int trem = tile - n * (tiles_y * tiles_x);
    const int ty = trem / tiles_x, tx = trem % tiles_x;
    const int fy0 = 8 * ty - 1, fx0 = 8 * tx - 1;
    __syncthreads();  // previous tile's gather is done with Tt
    auto load_a = [&](int mtile, f32x4 (&a)[4]) {
      const int p = mtile * 16 + li;
      const int ia = p / DG_NF, ib = p - ia * DG_NF;
      const int fy = fy0 + ia, fx = fx0 + ib;
      const bool ok = mtile < 8 && p < DG_NF * DG_NF && fy >= 0 && fy < HF && fx >= 0 && fx < WF;
      const float* src = dy + ((size_t)(n * HF + (ok ? fy : 0)) * WF + (ok ? fx : 0)) * 64 + 4 * kq;
#pragma unroll
      for (int c = 0; c < 4; ++c) a[c] = ok ? *(const f32x4*)(src + 16 * c) : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    f32x4 a[4], an[4];
    load_a(0, a);
#pragma unroll 1
    for (int mtile = 0; mtile < 8; ++mtile) {
      load_a(mtile + 1, an);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        if (wave + 4 * j < 10) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][r], breg[j][c][r], acc, 0, 0, 0);
          // D layout: column = lane & 15, row (pixel) = (lane >> 4) * 4 + reg
          const int ncol = (wave + 4 * j) * 16 + li;
          if (ncol < DG_NPROD) {
#pragma unroll
            for (int r = 0; r < 4; ++r) Tt[(mtile * 16 + kq * 4 + r) * DG_TP + ncol] = acc[r];
          }
        }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) a[c] = an[c];
    }
    __syncthreads();
    // gather: thread = image pixel (oyl, oxl) of the tile; ky = py + 2*dyk with py = (oyl + 1) & 1 (iy + 3 - ky must be even)
    const int oxl = tid & 15, oyl = tid >> 4;
    const int iy = 16 * ty + oyl, ix = 16 * tx + oxl;
    const int py = (oyl + 1) & 1, px = (oxl + 1) & 1;
    const int ay = (oyl + 3 - py) / 2 + 1, ax = (oxl + 3 - px) / 2 + 1;  // local feature index for dyk = 0 / dxk = 0
#pragma unroll
    for (int co = 0; co < 3; ++co) {
      float v = 0.f;
#pragma unroll
      for (int dyk = 0; dyk < 4; ++dyk) {
        const int ky = py + 2 * dyk;
        if (ky > 6) continue;
#pragma unroll
        for (int dxk = 0; dxk < 4; ++dxk) {
          const int kx = px + 2 * dxk;
          if (kx > 6) continue;
          v += Tt[((ay - dyk) * DG_NF + (ax - dxk)) * DG_TP + co * 49 + ky * 7 + kx];
        }
      }
      if (iy < H && ix < W) dx[((size_t)(n * C + cg * 3 + co) * H + iy) * W + ix] = v;
    }
  }
}

// Column order of conv1's weight-gradient GEMM (K = 7: 147 taps in 5 N-tiles of 32 lanes).  The 32 lanes of a tile read 32
// DIFFERENT taps of the image window with one ds_read_b32, and with the taps in natural order some two of them always share one of
// the 32 banks whatever the pitches (a tile spans 4.6 window rows of 7 taps): every operand read took two LDS cycles per lane group
// — 67 M conflict cycles per launch, 8 % of the kernel's time (profiles/r03d_pmc_mfma.json).  dW's columns can be computed in any order,
// so the taps are DEALT to the tiles by bank instead: with row pitch 23 and plane pitch 861 no bank holds more than five taps, and
// tile j takes the j-th tap of every bank (generated by tools/tap_banks.py, which also checks it).  255 = unused lane (it reads its
// tile's first tap — a broadcast — and its column is not stored).
constexpr int WG7_XP = 23, WG7_PP = 861;
__constant__ unsigned char TAP7[160] = {
      0,   1,   2,   4,   5,   7,   8,  11,  12,  13,  14,  15,  18,  19,  20,  21,
     25,  27,  28,  32,  33,  35,  42,  58,  59,  89,  96, 108, 124, 138, 255, 255,
     10,  22,  26,  29,  30,  31,  34,  36,  37,  38,  41,  43,  44,  45,  48,  50,
     51,  53,  57,  60,  64,  67,  72,  74,  75,  88, 101, 107, 114, 137, 255, 255,
      9,  23,  39,  49,  54,  55,  56,  61,  62,  63,  68,  69,  70,  71,  76,  77,
     78,  79,  84,  85,  86,  91,  92,  93, 115, 123, 130, 131, 140, 255, 255, 255,
      6,  17,  47,  52,  66,  73,  80,  82,  83,  87,  90,  94,  97,  99, 102, 103,
    104, 106, 109, 110, 111, 113, 116, 117, 118, 121, 125, 141, 144, 255, 255, 255,
      3,  16,  24,  40,  46,  65,  81,  95,  98, 100, 105, 112, 119, 120, 122, 126,
    127, 128, 129, 132, 133, 134, 135, 136, 139, 142, 143, 145, 146, 255, 255, 255};

// ------------------------------------------------------------------------------------------------------------------
// dW[ch][k] = sum_{n,pix} feat[n,pix,ch] * im2col(img)[n,pix,k]     ch in [0,64), k = (c,ky,kx) of channel group cg.
// GEMM view: M = 64 feature channels (2 M-tiles), N = KT taps (NT tiles of 32), K = pixels.
// Per 16x16-pixel tile the image window is staged in LDS (T) and the 64-channel feature rows are staged in two halves
// of 128 pixels (F, 32 KB) with 16-byte coalesced loads — a dword-per-lane fragment load straight from HBM is
// address-path bound (measured 2 TB/s).  Wave w owns M-tile (w&1), ALL N-tiles and 64 of a half's 128 pixels.
// Persistent over tiles; each (workgroup, w>>1) writes its own partial [64][NT*32] -> skinny_wgrad_reduce.
// ------------------------------------------------------------------------------------------------------------------
template <int K, int PAD, bool FUSED = false, typename PT = float>
__global__ __launch_bounds__(256, 2) void skinny_wgrad_kernel(const PT* __restrict__ img,
                                                             const float* __restrict__ feat,
                                                             float* __restrict__ partial, int N, int C, int H, int W,
                                                             int HF, int WF, int tiles_y, int tiles_x,
                                                             const float* __restrict__ feat_bnp, const PoolFuse pf, int npg,
                                                             const float* __restrict__ lut = nullptr) {
  constexpr bool U8 = sizeof(PT) == 1;  // uint8 frames + the normalisation table (image_land)
  // FUSED (K = 7): the feature operand is rebuilt from (y, argmax, dpooled) by the BatchNorm + ReLU + MaxPool backward (PoolFuse)
  // npg = images per BatchNorm group: image n uses the records feat_bnp / pf.bnp [(n / npg) * 256 ..], pf.sums [(n / npg) * 128 ..]
  constexpr int KT = Geo<K>::KT;
  constexpr int NT = (KT + 31) / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* T = (float*)smem;                  // image window
  float* F = T + Geo<K>::TILE_FLOATS;       // [RPS*16 pixels][64 channels] feature rows of the current stage
  if constexpr (U8) {  // the normalisation table moves into LDS (published by the first stage's barrier)
    float* L = F + 128 * 64;
    for (int i = threadIdx.x; i < 768; i += 256) L[i] = lut[i];
    lut = L;
  }
  // A tile is consumed in NSTAGE stages of RPS tile rows (8 = half a tile; 4 for FUSED, whose in-flight state per row is larger)
  constexpr int RPS = FUSED ? 4 : 8, NSTAGE = 16 / RPS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int mt = wave & 1, sub = wave >> 1;
  const int cg = blockIdx.y;
  const int ntiles = N * tiles_y * tiles_x;
  const int tpi = tiles_y * tiles_x;

  constexpr bool DEALT = (K == 7);  // columns dealt to the N-tiles by LDS bank (TAP7), window pitches of their own
  constexpr int WXP = DEALT ? WG7_XP : XP, WPP = DEALT ? WG7_PP : Geo<K>::PP;
  static_assert(6 * WPP <= Geo<K>::TILE_FLOATS && Geo<K>::ROWS * WXP <= WPP - 1, "window does not fit its planes");
  int kb[NT];  // per-lane LDS offset of this lane's tap in N-tile j (+h: the odd pixel of a k-step is one column on)
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int k = j * 32 + l31;
    int kk = (k < KT) ? k : 0;
    if constexpr (DEALT) { kk = TAP7[k]; if (kk == 255) kk = TAP7[j * 32]; }
    const int c = kk / (K * K), ky = (kk / K) % K, kx = kk % K;
    kb[j] = (c * 2 + (kx & 1)) * WPP + ky * WXP + (kx >> 1) + h;
  }
  f32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const float* tb[NT];  // this lane's image-operand column of N-tile j (tap offset and the odd-pixel shift h folded in)
#pragma unroll
  for (int j = 0; j < NT; ++j) tb[j] = T + kb[j];
  const float* fcol = F + h * 64 + mt * 32 + l31;  // this lane's feature-operand column (h: the odd pixel of a k-step is one row on)

  const int slot = tid & 15, prow = tid >> 4;  // feature staging: 16 lanes per pixel row (256 B), 16 pixels per pass
  // Per-group records, kept in registers and re-read when the tile sequence crosses into the next BatchNorm group:
  //  * feat_bnp != NULL (K = 4): `feat` is a raw convolution output and the operand is relu(batchnorm(feat)): sc4 / sh4;
  //  * FUSED: scale / shift of the forward BatchNorm (ReLU mask) and the two coefficients of its backward,
  //    dy = scale*dz - (c0 + c1*y),  c1 = scale*invstd*S2/count,  c0 = scale*S1/count - c1*mean  (S1 = S2 = 0 in eval mode).
  f32x4 sc4 = {1.f, 1.f, 1.f, 1.f}, sh4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 fc0 = {0.f, 0.f, 0.f, 0.f}, fc1 = {0.f, 0.f, 0.f, 0.f};
  int cur_grp = -1;
  // The staging of both operands is software-pipelined: what a thread contributes to the NEXT stage (RPS feature rows; FUSED:
  // RPS rows of y plus the <= 2*NWR pooling windows that can have picked them) is REQUESTED into registers before the current
  // stage's MFMA loop and RESOLVED into LDS after it, and the image window of the next tile travels under the last stage of this one.
  f32x4 pv[RPS];
  unsigned pmask = 0;
  f32x4 dpv[FUSED ? 2 : 1][2];
  uint32_t am[FUSED ? 2 : 1][2];
  unsigned wmask = 0;  // bit 2wy+wx: window (wy, wx) exists
  // (wave-uniform) every pooling window the requested stage looks at lies inside the pooled map — true for all but the last row and
  // column of tiles, and then f_resolve skips their masking.  (The same for the pixel mask measured as nothing: hipcc if-converts it.)
  bool all_win = false;
  // FUSED thread mapping: a thread owns a 2x2 block of pixels of the 4x16 stage — rows 2*brow + i, columns 2*bcol + j — and the
  // 2x2 pooling windows that can have picked them (3x3, stride 2, pad 1: window (R + wy, Cx + wx) sees pixel (2R + i, 2Cx + j) at
  // ky = i - 2wy + 1, kx = j - 2wx + 1), i.e. 9 membership tests with compile-time codes for 4 pixels on EVERY lane.  (One column
  // of 4 rows per thread needs 6 tests on even and 12 on odd columns, and a wave then pays 12 everywhere.)
  const int brow = (tid >> 4) >> 3, bcol = (tid >> 4) & 7;
  auto f_request = [&](int tile_, int half_) {
    const int n_ = tile_ / tpi;
    const int trem_ = tile_ - n_ * tpi;
    const int oy0_ = (trem_ / tiles_x) * 16, ox0_ = (trem_ % tiles_x) * 16;
    const bool live = tile_ < ntiles;
    const float* __restrict__ fsrc = FUSED ? pf.y : feat;
    pmask = 0;
    if constexpr (FUSED) {
      all_win = live && ((oy0_ + RPS * half_) >> 1) + RPS / 2 < pf.HP && (ox0_ >> 1) + 8 < pf.WP;
      const int oyb = oy0_ + RPS * half_ + 2 * brow, oxb = ox0_ + 2 * bcol;  // both even
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int oy = oyb + i, ox = oxb + j;
          const bool ok = live && oy < HF && ox < WF;  // (raw load from a clamped address; the mask is applied in f_resolve)
          pv[2 * i + j] = *(const f32x4*)(fsrc + (ok ? ((size_t)(n_ * HF + oy) * WF + ox) * 64 : (size_t)0) + slot * 4);
          pmask |= (ok ? 1u : 0u) << (2 * i + j);
        }
      wmask = 0;
#pragma unroll
      for (int wy = 0; wy < 2; ++wy)
#pragma unroll
        for (int wx = 0; wx < 2; ++wx) {
          const int py = (oyb >> 1) + wy, px = (oxb >> 1) + wx;
          const bool ok = live && py < pf.HP && px < pf.WP;
          const size_t pp = (ok ? ((size_t)(n_ * pf.HP + py) * pf.WP + px) * 64 : (size_t)0) + slot * 4;
          am[wy][wx] = *(const uint32_t*)(pf.argmax + pp);
          dpv[wy][wx] = *(const f32x4*)(pf.dpooled + pp);
          wmask |= (ok ? 1u : 0u) << (2 * wy + wx);
        }
    } else {
#pragma unroll
      for (int j = 0; j < RPS; ++j) {  // pixel p = 16*j + prow of the stage: tile row RPS*stage + j, column prow
        const int oy = oy0_ + RPS * half_ + j, ox = ox0_ + prow;
        const bool ok = live && oy < HF && ox < WF;  // (raw load from a clamped address; the mask is applied in f_resolve — see ImgRegs)
        pv[j] = *(const f32x4*)(fsrc + (ok ? ((size_t)(n_ * HF + oy) * WF + ox) * 64 : (size_t)0) + slot * 4);
        pmask |= (ok ? 1u : 0u) << j;
      }
    }
  };
  auto f_resolve = [&]() {  // the requested stage -> F
    if constexpr (FUSED) {
      uint32_t a[2][2];
#pragma unroll
      for (int wy = 0; wy < 2; ++wy)
#pragma unroll
        for (int wx = 0; wx < 2; ++wx) a[wy][wx] = am[wy][wx];
      if (!all_win) {
#pragma unroll
        for (int wy = 0; wy < 2; ++wy)
#pragma unroll
          for (int wx = 0; wx < 2; ++wx) a[wy][wx] = ((wmask >> (2 * wy + wx)) & 1u) ? am[wy][wx] : 0xffffffffu;  // 0xff matches nothing
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f32x4 dz = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int wy = 0; wy <= i; ++wy)
#pragma unroll
            for (int wx = 0; wx <= j; ++wx) {
              const uint32_t code = (uint32_t)((i - 2 * wy + 1) * 3 + (j - 2 * wx + 1));
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (((a[wy][wx] >> (8 * e)) & 0xffu) == code) dz[e] += dpv[wy][wx][e];
            }
          const f32x4 yy = pv[2 * i + j];
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float z = yy[e] * sc4[e] + sh4[e];
            const float d = z > 0.f ? dz[e] : 0.f;
            o[e] = sc4[e] * d - (fc0[e] + fc1[e] * yy[e]);
          }
          if (!((pmask >> (2 * i + j)) & 1u)) o = f32x4{0.f, 0.f, 0.f, 0.f};
          *(f32x4*)(F + (16 * (2 * brow + i) + 2 * bcol + j) * 64 + slot * 4) = o;
        }
    } else {
      if (K == 4 && feat_bnp) {  // a fused BatchNorm+ReLU is applied now, at consumption
#pragma unroll
        for (int j = 0; j < RPS; ++j) {
          const bool ok = (pmask >> j) & 1u;
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float z = pv[j][e] * sc4[e] + sh4[e]; pv[j][e] = (ok && z > 0.f) ? z : 0.f; }
        }
      }
      else {
#pragma unroll
        for (int j = 0; j < RPS; ++j)
          if (!((pmask >> j) & 1u)) pv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int j = 0; j < RPS; ++j) *(f32x4*)(F + (16 * j + prow) * 64 + slot * 4) = pv[j];
    }
  };
  ImgRegs<K> ir;  // the image window of the next tile, requested during the last stage of this one
  image_index<K, 16, WXP, WPP>(ir);
  auto i_request = [&](int tile_) {
    const int n_ = tile_ / tpi;
    const int trem_ = tile_ - n_ * tpi;
    image_request<K, PAD, K == 4>(ir, img, n_, C, cg, H, W, (trem_ / tiles_x) * 16, (trem_ % tiles_x) * 16, tile_ < ntiles);
  };
  if ((int)blockIdx.x < ntiles) { f_request(vtile(blockIdx.x, ntiles), 0); i_request(vtile(blockIdx.x, ntiles)); }
  for (int v = blockIdx.x; v < ntiles; v += gridDim.x) {
    const int tile = xcd_remap(v, ntiles);
    const int n = tile / tpi;
    const int grp = n / npg;
    if (grp != cur_grp) {
      cur_grp = grp;
      if constexpr (FUSED) {
        const float* __restrict__ pbnp = pf.bnp + grp * 256;
        const float* __restrict__ psums = pf.sums + grp * 128;
        const f32x4 mean = *(const f32x4*)(pbnp + slot * 4), invstd = *(const f32x4*)(pbnp + 64 + slot * 4);
        sc4 = *(const f32x4*)(pbnp + 128 + slot * 4); sh4 = *(const f32x4*)(pbnp + 192 + slot * 4);
        f32x4 m1 = *(const f32x4*)(psums + slot * 4), m2 = *(const f32x4*)(psums + 64 + slot * 4);
        if (!pf.training) m1 = m2 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          fc1[e] = sc4[e] * invstd[e] * m2[e] * pf.inv_count;
          fc0[e] = sc4[e] * m1[e] * pf.inv_count - fc1[e] * mean[e];
        }
      } else if (K == 4 && feat_bnp) {
        sc4 = *(const f32x4*)(feat_bnp + grp * 256 + 128 + slot * 4);
        sh4 = *(const f32x4*)(feat_bnp + grp * 256 + 192 + slot * 4);
      }
    }
#pragma unroll 1
    for (int half = 0; half < NSTAGE; ++half) {
      __syncthreads();
      if (half == 0) image_land<K, K == 4, 16, U8, WXP, WPP>(T, ir, 1.f, lut);
      f_resolve();
      __syncthreads();
      if (half + 1 < NSTAGE) f_request(tile, half + 1);
      else { f_request(vtile(v + gridDim.x, ntiles), 0); i_request(vtile(v + gridDim.x, ntiles)); }
      // RPS*4 k-steps per wave, in blocks of 4: step i of block blk covers pixel (row (RPS/2)*sub + (blk>>1) of the stage, column
      // 8*(blk&1) + 2i + h).  Everything that varies inside a block is an immediate offset of a ds_read2 (columns 0,2 / 4,6) and
      // everything that varies between blocks is wave-uniform, so a block costs one address per operand column (1 + NT VALU adds)
      // for its 4*NT MFMAs — the instruction stream, not the matrix pipe, is what bounds these loops.
      const int subu = __builtin_amdgcn_readfirstlane(sub);
#pragma unroll 2
      for (int blk = 0; blk < RPS; ++blk) {
        const int lrow = (RPS / 2) * subu + (blk >> 1), tx0 = 8 * (blk & 1);
        const float* fa = fcol + (lrow * 16 + tx0) * 64;
        const int boff = 2 * (RPS * half + lrow) * WXP + tx0;
        float a[4], b[NT][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = fa[2 * i * 64];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i) b[j][i] = tb[j][boff + 2 * i];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j][i], acc[j], 0, 0, 0);
      }
    }
  }
  float* out = partial + (((size_t)cg * gridDim.x + blockIdx.x) * 2 + sub) * (64 * NT * 32);
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    int col = j * 32 + l31;  // the column of dW this lane accumulated (natural tap index; skinny_wgrad_reduce reads columns < KT)
    if constexpr (DEALT) col = TAP7[col];
    if (col == 255) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      out[row * (NT * 32) + col] = acc[j][r];
    }
  }
}

// dw_ref[ch][cg*3+c][ky][kx] = sum over workgroups of partial[cg][wg][ch][k]; 64 outputs x 16 slices of the
// workgroup range per 1024-thread block, fp64, fixed order.  (With 256 outputs x 4 slices the launch had 12 blocks for the
// ConvTranspose layer — twelve CUs streaming 33 MB: 37 us; the slices are what spreads it over the chip.)
constexpr int WRED_OUTS = 64, WRED_PARTS = 16;
__global__ __launch_bounds__(1024) void skinny_wgrad_reduce(const float* __restrict__ partial, int nwg, int C, int KK, int KT,
                                                           int NTW, float* __restrict__ dw_ref) {
  const int o = threadIdx.x & (WRED_OUTS - 1), part = threadIdx.x / WRED_OUTS;
  const int id = blockIdx.x * WRED_OUTS + o;
  const int ncg = C / 3;
  const bool live = id < ncg * 64 * KT;
  int cg = 0, ch = 0, k = 0;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (live) {
    cg = id / (64 * KT);
    const int rem = id - cg * 64 * KT;
    ch = rem / KT; k = rem - ch * KT;
    const float* base = partial + (size_t)cg * nwg * (64 * NTW) + ch * NTW + k;
    const int per = (nwg + WRED_PARTS - 1) / WRED_PARTS;
    const int w0 = part * per, w1 = (w0 + per < nwg) ? w0 + per : nwg;
    int w = w0;
    for (; w + 3 < w1; w += 4) {
      s0 += (double)base[(size_t)w * (64 * NTW)];
      s1 += (double)base[(size_t)(w + 1) * (64 * NTW)];
      s2 += (double)base[(size_t)(w + 2) * (64 * NTW)];
      s3 += (double)base[(size_t)(w + 3) * (64 * NTW)];
    }
    for (; w < w1; ++w) s0 += (double)base[(size_t)w * (64 * NTW)];
  }
  __shared__ double sm[WRED_PARTS][WRED_OUTS];
  sm[part][o] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (part == 0 && live) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < WRED_PARTS; ++q) t += sm[q][o];
    dw_ref[((size_t)ch * C + cg * 3) * KK + k] = (float)t;
  }
}

// per-channel sum of an NCHW tensor (bias gradient of the last ConvTranspose): one block per (n, c) plane with float4
// loads -> partial[c][n] (fp64), then one thread per channel sums over n in a fixed order.
__global__ __launch_bounds__(256) void nchw_chan_sum_partial(const float* __restrict__ x, int C, int HW, double* __restrict__ partial,
                                                            int N) {
  const int n = blockIdx.x / C, c = blockIdx.x % C;
  const float* plane = x + (size_t)blockIdx.x * HW;
  float s = 0.f;
  const int hw4 = HW >> 2;
  for (int i = threadIdx.x; i < hw4; i += 256) {
    const f32x4 v = *(const f32x4*)(plane + i * 4);
    s += (v[0] + v[1]) + (v[2] + v[3]);
  }
  for (int i = hw4 * 4 + threadIdx.x; i < HW; i += 256) s += plane[i];
  double d = wave_sum_d((double)s);
  __shared__ double sm[4];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) partial[(size_t)c * N + n] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}
__global__ void nchw_chan_sum_final(const double* __restrict__ partial, int N, float* __restrict__ out) {
  const int c = blockIdx.x;
  double s = 0.0;
  for (int i = threadIdx.x; i < N; i += 64) s += partial[(size_t)c * N + i];
  s = wave_sum_d(s);
  if (threadIdx.x == 0) out[c] = (float)s;
}

static int check_skinny(const srlz_skinny_desc* d) {
  SRLZ_REQUIRE(d != nullptr, SRLZ_ERR_NULL, "skinny: null descriptor");
  SRLZ_REQUIRE((long long)d->n * d->c * d->himg * d->wimg < (1LL << 31), SRLZ_ERR_BAD_DESC,
               "skinny: image tensor of %lld elements (the window staging keeps 32-bit element offsets)",
               (long long)d->n * d->c * d->himg * d->wimg);
  SRLZ_REQUIRE(d->n > 0 && d->c > 0 && d->c % 3 == 0 && d->c <= 9, SRLZ_ERR_BAD_DESC, "skinny: C must be 3, 6 or 9 (got %d)", d->c);
  SRLZ_REQUIRE(d->groups >= 0 && (d->groups <= 1 || d->n % d->groups == 0), SRLZ_ERR_BAD_DESC,
               "skinny: n = %d is not a multiple of groups = %d", d->n, d->groups);
  if (d->kind == 0) {
    SRLZ_REQUIRE(d->hf == (d->himg + 6 - 7) / 2 + 1 && d->wf == (d->wimg + 6 - 7) / 2 + 1, SRLZ_ERR_BAD_DESC,
                 "conv1: feature map %dx%d inconsistent with image %dx%d", d->hf, d->wf, d->himg, d->wimg);
  } else if (d->kind == 1) {
    SRLZ_REQUIRE(d->himg == (d->hf - 1) * 2 + 4 && d->wimg == (d->wf - 1) * 2 + 4, SRLZ_ERR_BAD_DESC,
                 "convT_out: image %dx%d inconsistent with feature map %dx%d", d->himg, d->wimg, d->hf, d->wf);
  } else {
    SRLZ_REQUIRE(false, SRLZ_ERR_BAD_DESC, "skinny: unknown kind %d", d->kind);
  }
  return 0;
}

template <int K>
static size_t conv_lds(bool pipe) { return (size_t)(Geo<K>::TILE_FLOATS + Geo<K>::KS * 128 + 512 + (pipe ? 4096 : 0)) * 4; }

static int images_per_group(const srlz_skinny_desc* d) { return d->n / (d->groups > 1 ? d->groups : 1); }

static int persistent_grid(int ntiles) {
  const int g = 2 * srlz_device_cus();  // two persistent workgroups per CU
  return g > ntiles ? ntiles : g;
}

template <int K, int PAD, typename PT = float>
static int launch_conv(const PT* img, const float* w, float* feat, float* stats, const srlz_skinny_desc* d, hipStream_t st,
                       const float* y_raw = nullptr, const float* y_bnp = nullptr, const float* lut = nullptr) {
  const int ty = (d->hf + 15) / 16, tx = (d->wf + 15) / 16;
  const int ntiles = d->n * ty * tx;
  const bool multi = d->c > 3;
  const dim3 grid(persistent_grid(ntiles)), block(256);
#define SRLZ_CONV_LAUNCH(BN, MU, PIPED)                                                                                   \
  do {                                                                                                                    \
    const size_t lds = conv_lds<K>(PIPED) + (sizeof(PT) == 1 ? 768 * 4 : 0);                                              \
    SRLZ_MAX_LDS((skinny_conv_kernel<K, PAD, BN, MU, PT>), lds);                                                          \
    hipLaunchKernelGGL((skinny_conv_kernel<K, PAD, BN, MU, PT>), grid, block, lds, st, img, w, feat, stats, d->n, d->c,   \
                       d->himg, d->wimg, d->hf, d->wf, ty, tx, y_raw, y_bnp, images_per_group(d), lut);                   \
  } while (0)
  bool launched = false;
  if constexpr (K == 4 && sizeof(PT) == 4) if (y_raw) {
    if (multi) SRLZ_CONV_LAUNCH(true, true, false); else SRLZ_CONV_LAUNCH(true, false, false);
    launched = true;
  }
  if (!launched) {
    if (multi) SRLZ_CONV_LAUNCH(false, true, true); else SRLZ_CONV_LAUNCH(false, false, true);
  }
#undef SRLZ_CONV_LAUNCH
  SRLZ_LAUNCHED();
  return 0;
}

template <int K>
static size_t wgrad_ws(const srlz_skinny_desc* d) {
  constexpr int NT = (Geo<K>::KT + 31) / 32;
  const int ty = (d->hf + 15) / 16, tx = (d->wf + 15) / 16;
  const int g = persistent_grid(d->n * ty * tx);
  return (size_t)(d->c / 3) * g * 2 * 64 * NT * 32 * sizeof(float) + (size_t)d->n * d->c * sizeof(double);
}

template <int K, int PAD, typename PT = float>
static int launch_wgrad(const PT* img, const float* feat, float* dw, void* ws, size_t ws_bytes, const srlz_skinny_desc* d,
                        hipStream_t st, const float* feat_bnp = nullptr, const PoolFuse* pfuse = nullptr, const float* lut = nullptr) {
  constexpr int NT = (Geo<K>::KT + 31) / 32;
  const int ty = (d->hf + 15) / 16, tx = (d->wf + 15) / 16;
  const int ntiles = d->n * ty * tx;
  const int g = persistent_grid(ntiles);
  SRLZ_REQUIRE(ws_bytes >= wgrad_ws<K>(d), SRLZ_ERR_WORKSPACE, "skinny wgrad: workspace too small (%zu)", ws_bytes);
  SRLZ_REQUIRE(K == 4 || feat_bnp == nullptr, SRLZ_ERR_BAD_DESC, "skinny wgrad: a fused forward operand exists for the 4x4 layer only");
  const size_t lds = (size_t)(Geo<K>::TILE_FLOATS + 128 * 64) * 4 + (sizeof(PT) == 1 ? 768 * 4 : 0);
  float* partial = (float*)ws;
  PoolFuse pf = {};
  if (pfuse) pf = *pfuse;
  bool launched = false;
  if constexpr (K == 7) if (pf.y) {
    hipLaunchKernelGGL((skinny_wgrad_kernel<K, PAD, true, PT>), dim3(g, d->c / 3), dim3(256), lds, st, img, feat, partial, d->n, d->c,
                       d->himg, d->wimg, d->hf, d->wf, ty, tx, feat_bnp, pf, images_per_group(d), lut);
    launched = true;
  }
  if (!launched)
    hipLaunchKernelGGL((skinny_wgrad_kernel<K, PAD, false, PT>), dim3(g, d->c / 3), dim3(256), lds, st, img, feat, partial, d->n, d->c,
                       d->himg, d->wimg, d->hf, d->wf, ty, tx, feat_bnp, pf, images_per_group(d), lut);
  SRLZ_LAUNCHED();
  const int total = (d->c / 3) * 64 * Geo<K>::KT;
  hipLaunchKernelGGL(skinny_wgrad_reduce, dim3((total + WRED_OUTS - 1) / WRED_OUTS), dim3(1024), 0, st, partial, 2 * g, d->c, K * K, Geo<K>::KT,
                     NT * 32, dw);
  SRLZ_LAUNCHED();
  return 0;
}

}  // namespace

extern "C" int srlz_skinny_tiles(const srlz_skinny_desc* d) {
  if (check_skinny(d)) return -1;
  return d->n * ((d->hf + 15) / 16) * ((d->wf + 15) / 16);
}

extern "C" int srlz_conv1_fwd(const float* x_nchw, const float* w_ref, float* y_nhwc, float* stats_partial,
                              const srlz_skinny_desc* d, srlz_stream_t stream) {
  if (int rc = check_skinny(d)) return rc;
  SRLZ_REQUIRE(d->kind == 0, SRLZ_ERR_BAD_DESC, "conv1_fwd: descriptor kind must be 0");
  SRLZ_REQUIRE(x_nchw && w_ref && y_nhwc, SRLZ_ERR_NULL, "conv1_fwd: null pointer");
  return launch_conv<7, 3>(x_nchw, w_ref, y_nhwc, stats_partial, d, as_stream(stream));
}

extern "C" int srlz_conv1_fwd_u8(const uint8_t* x_u8, const float* norm_lut, const float* w_ref, float* y_nhwc,
                                 float* stats_partial, const srlz_skinny_desc* d, srlz_stream_t stream) {
  if (int rc = check_skinny(d)) return rc;
  SRLZ_REQUIRE(d->kind == 0, SRLZ_ERR_BAD_DESC, "conv1_fwd_u8: descriptor kind must be 0");
  SRLZ_REQUIRE(x_u8 && norm_lut && w_ref && y_nhwc, SRLZ_ERR_NULL, "conv1_fwd_u8: null pointer");
  return launch_conv<7, 3, uint8_t>(x_u8, w_ref, y_nhwc, stats_partial, d, as_stream(stream), nullptr, nullptr, norm_lut);
}

extern "C" size_t srlz_skinny_bwd_weight_workspace(const srlz_skinny_desc* d) {
  if (check_skinny(d)) return 0;
  return d->kind == 0 ? wgrad_ws<7>(d) : wgrad_ws<4>(d);
}

extern "C" int srlz_conv1_bwd_data(const float* dy_nhwc, const float* w_ref, float* dx_nchw, const srlz_skinny_desc* d,
                                   srlz_stream_t stream) {
  if (int rc = check_skinny(d)) return rc;
  SRLZ_REQUIRE(d->kind == 0, SRLZ_ERR_BAD_DESC, "conv1_bwd_data: descriptor kind must be 0");
  SRLZ_REQUIRE(dy_nhwc && w_ref && dx_nchw, SRLZ_ERR_NULL, "conv1_bwd_data: null pointer");
  const int ty = (d->himg + 15) / 16, tx = (d->wimg + 15) / 16;
  const int ntiles = d->n * ty * tx;
  const size_t lds = (size_t)128 * DG_TP * sizeof(float);
  SRLZ_MAX_LDS(conv1_dgrad_kernel, lds);
  hipLaunchKernelGGL(conv1_dgrad_kernel, dim3(persistent_grid(ntiles), d->c / 3), dim3(256), lds, as_stream(stream), dy_nhwc,
                     w_ref, dx_nchw, d->n, d->c, d->himg, d->wimg, d->hf, d->wf, ty, tx);
  SRLZ_LAUNCHED();
  return 0;
}

extern "C" int srlz_conv1_bwd_weight(const float* x_nchw, const float* dy_nhwc, float* dw_ref, void* ws, size_t ws_bytes,
                                     const srlz_skinny_desc* d, srlz_stream_t stream) {
  if (int rc = check_skinny(d)) return rc;
  SRLZ_REQUIRE(d->kind == 0, SRLZ_ERR_BAD_DESC, "conv1_bwd_weight: descriptor kind must be 0");
  SRLZ_REQUIRE(x_nchw && dy_nhwc && dw_ref && ws, SRLZ_ERR_NULL, "conv1_bwd_weight: null pointer");
  return launch_wgrad<7, 3>(x_nchw, dy_nhwc, dw_ref, ws, ws_bytes, d, as_stream(stream));
}

template <typename PT>
static int conv1_bwd_weight_fused(const PT* x_nchw, const float* lut, const float* y_nhwc, const float* bnp, const uint8_t* argmax,
                                  const float* dpooled, const float* sums, int training, float* dw_ref, void* ws,
                                  size_t ws_bytes, const srlz_skinny_desc* d, const srlz_pool_desc* pd,
                                  srlz_stream_t stream) {
  if (int rc = check_skinny(d)) return rc;
  SRLZ_REQUIRE(d->kind == 0 && pd, SRLZ_ERR_BAD_DESC, "conv1_bwd_weight_fused: descriptor kind must be 0");
  SRLZ_REQUIRE(x_nchw && y_nhwc && bnp && argmax && dpooled && sums && dw_ref && ws, SRLZ_ERR_NULL,
               "conv1_bwd_weight_fused: null pointer");
  SRLZ_REQUIRE(pd->n == d->n && pd->h == d->hf && pd->w == d->wf && !pd->out_nchw && pd->pool_pad == 1 &&
               (pd->groups > 1 ? pd->groups : 1) == (d->groups > 1 ? d->groups : 1), SRLZ_ERR_BAD_DESC,
               "conv1_bwd_weight_fused: pooling descriptor does not match conv1's output");
  PoolFuse pf;
  pf.y = y_nhwc; pf.argmax = argmax; pf.dpooled = dpooled; pf.bnp = bnp; pf.sums = sums;
  pf.HP = pd->hp; pf.WP = pd->wp; pf.pad = pd->pool_pad; pf.training = training;
  pf.inv_count = 1.0f / (float)((double)images_per_group(d) * d->hf * d->wf);  // BatchNorm count of ONE group
  return launch_wgrad<7, 3, PT>(x_nchw, y_nhwc, dw_ref, ws, ws_bytes, d, as_stream(stream), nullptr, &pf, lut);
}

extern "C" int srlz_conv1_bwd_weight_fused(const float* x_nchw, const float* y_nhwc, const float* bnp, const uint8_t* argmax,
                                           const float* dpooled, const float* sums, int training, float* dw_ref, void* ws,
                                           size_t ws_bytes, const srlz_skinny_desc* d, const srlz_pool_desc* pd,
                                           srlz_stream_t stream) {
  return conv1_bwd_weight_fused<float>(x_nchw, nullptr, y_nhwc, bnp, argmax, dpooled, sums, training, dw_ref, ws, ws_bytes, d, pd,
                                       stream);
}

extern "C" int srlz_conv1_bwd_weight_fused_u8(const uint8_t* x_u8, const float* norm_lut, const float* y_nhwc, const float* bnp,
                                              const uint8_t* argmax, const float* dpooled, const float* sums, int training,
                                              float* dw_ref, void* ws, size_t ws_bytes, const srlz_skinny_desc* d,
                                              const srlz_pool_desc* pd, srlz_stream_t stream) {
  SRLZ_REQUIRE(norm_lut, SRLZ_ERR_NULL, "conv1_bwd_weight_fused_u8: null normalisation table");
  return conv1_bwd_weight_fused<uint8_t>(x_u8, norm_lut, y_nhwc, bnp, argmax, dpooled, sums, training, dw_ref, ws, ws_bytes, d, pd,
                                         stream);
}

extern "C" int srlz_convT_out_bwd_data(const float* dy_nchw, const float* w_ref, float* dx_nhwc, const float* x_raw,
                                       const float* x_bnp, float* bn_bwd_partial, const srlz_skinny_desc* d,
                                       srlz_stream_t stream) {
  if (int rc = check_skinny(d)) return rc;
  SRLZ_REQUIRE(d->kind == 1, SRLZ_ERR_BAD_DESC, "convT_out_bwd_data: descriptor kind must be 1");
  SRLZ_REQUIRE(dy_nchw && w_ref && dx_nhwc, SRLZ_ERR_NULL, "convT_out_bwd_data: null pointer");
  SRLZ_REQUIRE((x_raw != nullptr) == (x_bnp != nullptr) && (x_raw != nullptr) == (bn_bwd_partial != nullptr), SRLZ_ERR_NULL,
               "convT_out_bwd_data: x_raw, x_bnp and bn_bwd_partial go together");
  // dx[n,iy,ix,ci] = sum_{co,ky,kx} dy[n,co,2iy+ky,2ix+kx] * w_ref[ci,co,ky,kx]  == a 4x4 s2 p0 "conv" of dy
  return launch_conv<4, 0>(dy_nchw, w_ref, dx_nhwc, bn_bwd_partial, d, as_stream(stream), x_raw, x_bnp);
}

extern "C" int srlz_convT_out_bwd_weight(const float* x_nhwc, const float* dy_nchw, float* dw_ref, float* dbias,
                                         const float* x_bnp, void* ws, size_t ws_bytes, const srlz_skinny_desc* d,
                                         srlz_stream_t stream) {
  if (int rc = check_skinny(d)) return rc;
  SRLZ_REQUIRE(d->kind == 1, SRLZ_ERR_BAD_DESC, "convT_out_bwd_weight: descriptor kind must be 1");
  SRLZ_REQUIRE(x_nhwc && dy_nchw && dw_ref && ws, SRLZ_ERR_NULL, "convT_out_bwd_weight: null pointer");
  // dw_ref[ci,co,ky,kx] = sum_{n,iy,ix} x[n,iy,ix,ci] * dy[n,co,2iy+ky,2ix+kx]
  if (int rc = launch_wgrad<4, 0>(dy_nchw, x_nhwc, dw_ref, ws, ws_bytes, d, as_stream(stream), x_bnp)) return rc;
  if (dbias) {
    double* part = (double*)((char*)ws + wgrad_ws<4>(d) - (size_t)d->n * d->c * sizeof(double));
    SRLZ_REQUIRE((((uintptr_t)part) & 7) == 0 && ((d->himg * d->wimg) & 3) == 0, SRLZ_ERR_BAD_DESC,
                 "convT_out_bwd_weight: unaligned workspace / image plane");
    hipLaunchKernelGGL(nchw_chan_sum_partial, dim3(d->n * d->c), dim3(256), 0, as_stream(stream), dy_nchw, d->c,
                       d->himg * d->wimg, part, d->n);
    SRLZ_LAUNCHED();
    hipLaunchKernelGGL(nchw_chan_sum_final, dim3(d->c), dim3(64), 0, as_stream(stream), part, d->n, dbias);
    SRLZ_LAUNCHED();
  }
  return 0;
}
