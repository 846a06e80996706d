// Last decoder stage into the first layer of decoder.final_dense, one launch:
//     base = bilinear_x2(t4)[:, :3] + x                      (models/cdan.py:153-154)
//     g0   = conv3x3(relu(bn0(base)))                        (DenseBlock layer 0, models/cdan.py:35-36,41-46,155)
// `base` is still written (layers 1-3 and the transition read it), as ONE 16-byte channel group per pixel; g0 is the
// block's first 16-channel growth map.
//
// Why fused.  Both steps live on 3 real channels at full resolution.  As two launches (upsample2x_add_nchw3 + conv_kernel
// with the base padded to one 16-byte K group) they cost 24 + 45 us at B = 32, 256x256, bf16 -- nine K = 32 MFMAs and nine
// LDS fragment reads per 16 pixels on 3 channels, a second staging pass over a tensor the first launch had in registers.
// Here the tile's 18x18 base patch is computed straight from t4 (4 taps, channels 0..3 in one load each) and the fp32 NCHW
// input planes, activated and laid out [pixel][4] in LDS; the 9 taps x (3 + 1 zero) channels are im2col'ed into two MFMA steps
// per 16 pixels (taps 0..7 | tap 8; fp32: 2 x 16 of k = tap*3 + c), exactly as conv_first_kernel does for encoder.conv1.  The halo ring of
// the patch is recomputed by the neighbouring tiles (324 / 256 pixels): cheaper than a round trip through HBM.
// Arithmetic is kept identical to the unfused pair: the base is rounded to the storage type before the pre-activation,
// the pre-activation is one fused multiply-add rounded once, zero padding applies to the ACTIVATED tensor.
#include <algorithm>

#include "common.hpp"

#pragma clang fp contract(off)   // (the interpolation must round like upsample2x_add_nchw3's, resample.hip)

namespace mdie {

#ifdef EXP_UDSTAMPS   // diagnostic build only (tools/stamp_updense0.py): shader-clock stamps of wave 0 of every workgroup
static unsigned long long* g_ud_dbg = nullptr;
#define USTAMP(k) do { if (udbg && tid == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); udbg[(size_t)blockIdx.x * 8 + (k)] = t_; } } while (0)
#else
#define USTAMP(k) do {} while (0)
#endif

constexpr int UD_THREADS = 256;
constexpr int UD_TILE = 16, UD_PW = UD_TILE + 2;

struct UpDense0Args {
  int B, H, W;                    // OUTPUT extent (= network input extent); t4 is H/2 x W/2
  const char* lo; int lo_stride;  // decoder.conv4 output, NHWC, >= 4 stored channels per pixel
  const float* x;                 // network input, fp32 NCHW [B,3,H,W]
  char* base; int base_ch;        // out: NHWC, base_ch stored channels per pixel (one 16-byte group, or 16)
  const char* weight;             // mdie_pack_conv_first_weight layout: [2 steps][16][64 B]
  const float *pre_scale, *pre_shift;   // folded BatchNorm of dense layer 0, >= 3 entries
  const float* bias;              // [16]
  char* g0; int g0_stride;        // out: NHWC, 16 channels
  // TR: the block's transition folded into its producers (mdie_tr_fuse; the scheme is described in conv_thin.hip)
  unsigned long long* dbg;        // (diagnostic builds)
  const char* tr_w; int tr_c0;    // the transition's packed 1x1 weights; its stored input channel of g0's channel 0 (base sits at 0..2)
  const float *tr_scale, *tr_shift;   // the transition's folded BatchNorm, by its stored input channel
  float* tr_out;                  // [pixel][4] fp32: the partial sums (base term + g0 term)
  const long long* delta;         // several weight sets in one launch (mdie_up_dense0_desc.blob_delta): per-image byte offset of every parameter pointer
};

__device__ __forceinline__ void ud_src(int dst, int in_size, int& i0, int& i1, float& l0, float& l1) {   // = resample.hip src_index
  float src = ((float)dst + 0.5f) * 0.5f - 0.5f;
  src = src < 0.f ? 0.f : src;
  i0 = (int)src;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = src - (float)i0;
  l0 = 1.0f - l1;
}

template <typename T> __device__ __forceinline__ f32x4 ud_mma(const uint4& w, const uint4& x, f32x4 acc);
template <> __device__ __forceinline__ f32x4 ud_mma<bf16>(const uint4& w, const uint4& x, f32x4 acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4 ud_mma<f16>(const uint4& w, const uint4& x, f32x4 acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4 ud_mma<float>(const uint4& w, const uint4& x, f32x4 acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.x), __uint_as_float(x.x), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.y), __uint_as_float(x.y), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.z), __uint_as_float(x.z), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.w), __uint_as_float(x.w), acc, 0, 0, 0);
  return acc;
}

// BASE_ST: stored channels per pixel of `base` (= BASE_CH, or 4 of the 8-channel group: the half-group form of mdie_seg)
template <typename T, int BASE_CH, bool TR = false, int BASE_ST = BASE_CH>
__global__ __launch_bounds__(UD_THREADS, 4) void up_dense0_kernel(const UpDense0Args a0) {
  static_assert(!TR || sizeof(T) == 2, "the transition fusion is built for the 16-bit storage types");
  constexpr int E = sizeof(T);
  constexpr int VEC = Traits<T>::VEC;
  constexpr int STEPS = 2;   // (as conv_first_kernel: fp32 2 x 16 of k = tap*3 + c; 16-bit k' = tap*4 + c, taps 0..7 | tap 8)
  constexpr int NPS = 4;
  constexpr int PW = UD_PW;
  __shared__ __attribute__((aligned(16))) T patch[PW * PW * 4];
  __shared__ __attribute__((aligned(16))) T trpatch[TR ? UD_TILE * UD_TILE * 4 : 4];   // TR: relu(bn_tr(base)) of the tile's own pixels, [pixel][4]
  // 16-bit types: the tile's 10 x 10 low-resolution pixels (channels 0..3, 8 bytes each) are loaded ONCE into LDS and the 4 bilinear taps of a
  // patch pixel are LDS reads: 100 global loads per tile instead of 4 x 324.  The kernel ran at 3.0 TB/s of the bytes it must move with 9 scattered
  // loads per thread; what such loads cost is their ISSUE on the vector memory path, not the wait (round 6: the one-launch block's base phase,
  // profiles/r06f_final_block_v2_stamps.txt).  Same values, same arithmetic: outputs bit-identical.
  constexpr int LW = UD_TILE / 2 + 2;
  __shared__ __attribute__((aligned(8))) uint2 lostage[E == 2 ? LW * LW : 1];

  const int tid = threadIdx.x, lane = tid & 63, lq = lane >> 4, lp = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid = blockIdx.x;
  const int tiles_x = (a0.W + UD_TILE - 1) / UD_TILE, tiles_y = (a0.H + UD_TILE - 1) / UD_TILE;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y; bid /= tiles_y;
  const int img = bid;
  const int y0 = ty * UD_TILE, x0 = tx * UD_TILE;
  UpDense0Args a = a0;
  if (a0.delta) {   // (launch-uniform) this tile's image selects its weight set
    const long long dl = a0.delta[img];
    a.weight = a0.weight + dl;
    a.pre_scale = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a0.pre_scale) + dl);
    a.pre_shift = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a0.pre_shift) + dl);
    a.bias = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a0.bias) + dl);
    if constexpr (TR) {
      a.tr_w = a0.tr_w + dl;
      a.tr_scale = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a0.tr_scale) + dl);
      a.tr_shift = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a0.tr_shift) + dl);
    }
  }

#ifdef EXP_UDSTAMPS
  unsigned long long* const udbg = a0.dbg;
#endif
  USTAMP(0);
  // weight fragments and constants first: they land while the patch is computed
  uint4 wf[STEPS];
#pragma unroll
  for (int s = 0; s < STEPS; ++s) wf[s] = *reinterpret_cast<const uint4*>(a.weight + ((size_t)s * 16 + lp) * 64 + lq * 16);
  const float4 bias = *reinterpret_cast<const float4*>(a.bias + lq * 4);
  const float ps0 = a.pre_scale[0], ps1 = a.pre_scale[1], ps2 = a.pre_scale[2];
  const float pb0 = a.pre_shift[0], pb1 = a.pre_shift[1], pb2 = a.pre_shift[2];
  // transition term (TR): the A fragment of row subtile ps has output o in row 4 ps + o (zeros elsewhere); K group lq: elements
  // 0..3 = g0's channels 4 lq .. 4 lq + 3, elements 4..7 = the base channels 0..3 in K group 0 (zeros elsewhere) -- and the constants
  uint4 tra = make_uint4(0u, 0u, 0u, 0u);   // the lane's row o = lp & 3 of the transition's weights; subtile ps uses it on the lanes with lp >> 2 == ps
  f32x2 trs[2], trb[2];
  float tbs[3] = {0.f, 0.f, 0.f}, tbb[3] = {0.f, 0.f, 0.f};
  if constexpr (TR) {
#pragma unroll
    for (int c3 = 0; c3 < 3; ++c3) { tbs[c3] = a.tr_scale[c3]; tbb[c3] = a.tr_shift[c3]; }
  }

  // ---- the tile's base patch: every load of both iterations is issued before the first use ----
  const int Hl = a.H >> 1, Wl = a.W >> 1;
  const size_t plane = (size_t)a.H * a.W;
  constexpr int PIT = (PW * PW + UD_THREADS - 1) / UD_THREADS;
  float t[PIT][4][4], xin[PIT][3], hy[PIT][2], wx[PIT][2];
  bool inside[PIT];
  const int ly0 = (y0 >> 1) - 1, lx0 = (x0 >> 1) - 1;      // low-resolution pixel of lostage[0][0] (may be -1: clamped when loaded, never addressed then)
  int ltap[PIT][4];                                        // E == 2: byte offsets of the 4 taps in lostage
  // INTERIOR tiles (the 18x18 patch and its 10x10 low-resolution pixels inside the picture, no tap clamped: 77 % of the tiles of a 256x256 picture):
  // the half-pixel source of an x2 upsample is the pixel's parity -- tap weights 0.25 / 0.75, first tap at (p >> 1) of the staged pixels -- so the
  // float arithmetic of ud_src, the clamps and the in-picture tests go (the kernel is bound by vector issue: profiles/r06j_up_dense0_stamps.txt,
  // 4.3 k of a workgroup's 12.7 k cycles in this index phase).  ud_src's values exactly: bit-identical.
  const bool interior = E == 2 && y0 >= 2 && y0 + UD_TILE + 2 <= a.H && x0 >= 2 && x0 + UD_TILE + 2 <= a.W;     // (workgroup-uniform)
  if constexpr (E == 2) {
    if (tid < LW * LW) {
      const int r = tid / LW, c = tid - r * LW;
      const int ry = interior ? ly0 + r : min(max(ly0 + r, 0), Hl - 1), rx = interior ? lx0 + c : min(max(lx0 + c, 0), Wl - 1);
      const char* lb = a.lo + (size_t)img * Hl * Wl * a.lo_stride * E;                   // wave-uniform
      lostage[tid] = *reinterpret_cast<const uint2*>(lb + __umul24(__umul24(ry, Wl) + rx, (unsigned)a.lo_stride * E));
    }
  }
  if (interior) {
    if constexpr (E == 2) {
      const float* const xtile = a.x + (size_t)img * 3 * plane + (size_t)(y0 - 1) * a.W + (x0 - 1);      // wave-uniform: patch pixel (0, 0)
#pragma unroll
      for (int it = 0; it < PIT; ++it) {
        const int p = tid + it * UD_THREADS;
        const int py = p / PW, px = p - py * PW;
        inside[it] = p < PW * PW;
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int c = 0; c < 4; ++c) t[it][k][c] = 0.f;
        xin[it][0] = xin[it][1] = xin[it][2] = 0.f;
        hy[it][1] = (py & 1) ? 0.75f : 0.25f; hy[it][0] = 1.0f - hy[it][1];      // patch row py = picture row y0 - 1 + py: odd py is an even row
        wx[it][1] = (px & 1) ? 0.75f : 0.25f; wx[it][0] = 1.0f - wx[it][1];
        ltap[it][0] = ((py >> 1) * LW + (px >> 1)) * 8;
        ltap[it][1] = ltap[it][0] + 8; ltap[it][2] = ltap[it][0] + LW * 8; ltap[it][3] = ltap[it][0] + LW * 8 + 8;
        if (inside[it]) {
          const unsigned xo = (__umul24(py, a.W) + px) * 4u;
          xin[it][0] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(xtile) + xo);
          xin[it][1] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(xtile + plane) + xo);
          xin[it][2] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(xtile + 2 * plane) + xo);
        }
      }
    }
  } else
#pragma unroll
  for (int it = 0; it < PIT; ++it) {
    const int p = tid + it * UD_THREADS;
    const int py = p / PW, px = p - py * PW;
    const int gy = y0 + py - 1, gx = x0 + px - 1;
    inside[it] = p < PW * PW && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int c = 0; c < 4; ++c) t[it][k][c] = 0.f;
    xin[it][0] = xin[it][1] = xin[it][2] = 0.f;
    hy[it][0] = hy[it][1] = wx[it][0] = wx[it][1] = 0.f;
    if (inside[it]) {
      int ya, yb, xa, xb;
      ud_src(gy, Hl, ya, yb, hy[it][0], hy[it][1]);
      ud_src(gx, Wl, xa, xb, wx[it][0], wx[it][1]);
      // addresses = wave-uniform image base + 32-bit lane offset from 24-bit multiplies (stamps of the first-layer kernel: a
      // 32-bit integer multiply or a 64-bit mad is 16 cycles of the SIMD, v_mul_u32_u24 is 4; the host checks the ranges)
      if constexpr (E == 2) {
        const int ra = (ya - ly0) * LW, rb = (yb - ly0) * LW;
        ltap[it][0] = (ra + xa - lx0) * 8; ltap[it][1] = (ra + xb - lx0) * 8; ltap[it][2] = (rb + xa - lx0) * 8; ltap[it][3] = (rb + xb - lx0) * 8;
      } else {
        const char* lb = a.lo + (size_t)img * Hl * Wl * a.lo_stride * E;
        const unsigned ls = (unsigned)a.lo_stride * E, ra = __umul24(ya, Wl), rb = __umul24(yb, Wl);
        const char* q[4] = {lb + __umul24(ra + xa, ls), lb + __umul24(ra + xb, ls), lb + __umul24(rb + xa, ls), lb + __umul24(rb + xb, ls)};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float4 u = *reinterpret_cast<const float4*>(q[k]);
          t[it][k][0] = u.x; t[it][k][1] = u.y; t[it][k][2] = u.z;
        }
      }
      const float* xbase = a.x + (size_t)img * 3 * plane;                    // wave-uniform
      const unsigned xo = (__umul24(gy, a.W) + gx) * 4u;
      xin[it][0] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(xbase) + xo);
      xin[it][1] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(xbase + plane) + xo);
      xin[it][2] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(xbase + 2 * plane) + xo);
    }
  }
  if constexpr (E == 2) {
    USTAMP(1);
    __syncthreads();                                       // the low-resolution pixels are staged
    USTAMP(2);
#pragma unroll
    for (int it = 0; it < PIT; ++it)
      if (inside[it]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(lostage) + ltap[it][k]);
          t[it][k][0] = Half<T>::lo(u.x); t[it][k][1] = Half<T>::hi(u.x); t[it][k][2] = Half<T>::lo(u.y);
        }
      }
  }
#pragma unroll
  for (int it = 0; it < PIT; ++it) {
    const int p = tid + it * UD_THREADS;
    if (p < PW * PW) {
      const int py = p / PW, px = p - py * PW;
      float f[3], act[3];
      const float ps[3] = {ps0, ps1, ps2}, pb[3] = {pb0, pb1, pb2};
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        f[c] = hy[it][0] * (wx[it][0] * t[it][0][c] + wx[it][1] * t[it][1][c]) + hy[it][1] * (wx[it][0] * t[it][2][c] + wx[it][1] * t[it][3][c]) + xin[it][c];
        if constexpr (E == 2) f[c] = (float)(T)f[c];                 // the stored base: what the unfused layer 0 would read back
        act[c] = inside[it] ? fmaxf(fmaf(f[c], ps[c], pb[c]), 0.f) : 0.f;   // zero padding of the ACTIVATED tensor
      }
      T* d = patch + p * 4;
      if constexpr (E == 2) *reinterpret_cast<uint2*>(d) = make_uint2(Half<T>::pack(act[0], act[1]), Half<T>::pack(act[2], 0.f));
      else *reinterpret_cast<float4*>(d) = make_float4(act[0], act[1], act[2], 0.f);
      if (inside[it] && py >= 1 && py <= UD_TILE && px >= 1 && px <= UD_TILE) {   // this tile's own pixels: the base tensor
        float o[BASE_CH];
#pragma unroll
        for (int c = 0; c < BASE_CH; ++c) o[c] = c < 3 ? f[c] : 0.f;
        char* const dstb = a.base + (size_t)img * plane * (BASE_ST * E) + (__umul24(y0 + py - 1, a.W) + (x0 + px - 1)) * (unsigned)(BASE_ST * E);
        if constexpr (BASE_ST == BASE_CH) {
          uint4* dst = reinterpret_cast<uint4*>(dstb);
#pragma unroll
          for (int v = 0; v < BASE_CH / VEC; ++v) dst[v] = Vec16<T>::pack(o + v * VEC);
        } else {   // half group (16-bit types): the 3 real channels and one zero, 8 bytes per pixel
          static_assert(E == 2 && BASE_ST == 4 && BASE_CH == 8, "half-group base: 16-bit types, 4 of 8 channels");
          *reinterpret_cast<uint2*>(dstb) = make_uint2(Half<T>::pack(o[0], o[1]), Half<T>::pack(o[2], 0.f));
          // the consumers' 16-byte column load of the LAST pixel of the buffer reads 8 bytes behind it into zero-weighted channels:
          // 0 * NaN is NaN, so those bytes are this kernel's to define (mdie_seg: the buffer is writable 8 bytes past its last pixel)
          if (img == a.B - 1 && y0 + py == a.H && x0 + px == a.W) *reinterpret_cast<uint2*>(dstb + 8) = make_uint2(0u, 0u);
        }
        if constexpr (TR) {   // the transition's pre-activation of the stored base (f[] is already rounded to T)
          const float t0 = fmaxf(fmaf(f[0], tbs[0], tbb[0]), 0.f), t1 = fmaxf(fmaf(f[1], tbs[1], tbb[1]), 0.f), t2 = fmaxf(fmaf(f[2], tbs[2], tbb[2]), 0.f);
          *reinterpret_cast<uint2*>(trpatch + ((py - 1) * UD_TILE + (px - 1)) * 4) = make_uint2(Half<T>::pack(t0, t1), Half<T>::pack(t2, 0.f));
        }
      }
    }
  }
  if constexpr (TR) {   // (loaded here, not at the top: not live across the staging arithmetic, whose registers set the occupancy)
    const int c = a.tr_c0 + 4 * lq;
    const char* const wrow = a.tr_w + ((size_t)(c >> 5) * 4 + ((c & 31) >> 3)) * (16 * 16) + (c & 7) * 2;
    {
      const uint2 wg = *reinterpret_cast<const uint2*>(wrow + (unsigned)(lp & 3) * 16);
      const uint2 wb = *reinterpret_cast<const uint2*>(a.tr_w + (unsigned)(lp & 3) * 16);        // stored channels 0..3: chunk 0, K group 0
      tra = make_uint4(wg.x, wg.y, lq == 0 ? wb.x : 0u, lq == 0 ? wb.y : 0u);
    }
    const float4 s4 = *reinterpret_cast<const float4*>(a.tr_scale + a.tr_c0 + 4 * lq), b4 = *reinterpret_cast<const float4*>(a.tr_shift + a.tr_c0 + 4 * lq);
    trs[0] = f32x2{s4.x, s4.y}; trs[1] = f32x2{s4.z, s4.w}; trb[0] = f32x2{b4.x, b4.y}; trb[1] = f32x2{b4.z, b4.w};
  }
  USTAMP(3);
  __syncthreads();
  USTAMP(4);

  // ---- im2col gather (k = tap*3 + c) and one MFMA step per 16 pixels ----
  constexpr int KPL = E == 2 ? 3 : 4;
  int goff[STEPS][KPL];
  if constexpr (E == 2) {
    const int ta = 2 * lq, tb = 2 * lq + 1;
    goff[0][0] = ((ta / 3) * PW + ta % 3) * 4; goff[0][1] = ((tb / 3) * PW + tb % 3) * 4; goff[0][2] = (2 * PW + 2) * 4;
    goff[1][0] = goff[1][1] = goff[1][2] = 0;
  } else {
#pragma unroll
    for (int s = 0; s < STEPS; ++s)
#pragma unroll
      for (int i = 0; i < KPL; ++i) {
        const int k = 16 * s + 4 * lq + i;
        const int tap = k / 3, c = k - tap * 3;
        goff[s][i] = tap < 9 ? ((tap / 3) * PW + (tap % 3)) * 4 + c : 3;   // k >= 27: the zero channel of the pixel
      }
  }
  f32x4 tacc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ps = 0; ps < NPS; ++ps) {
    const int blk = (wave * NPS + ps) * 4 + (lp >> 2);
    const int y = 2 * (blk / (UD_TILE / 2)) + ((lp >> 1) & 1), x = 2 * (blk % (UD_TILE / 2)) + (lp & 1);
    const T* bp = patch + (y * PW + x) * 4;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (E == 2) {   // a patch pixel = 8 aligned bytes = half an operand
      const uint2 pa = *reinterpret_cast<const uint2*>(bp + goff[0][0]), pb = *reinterpret_cast<const uint2*>(bp + goff[0][1]);
      const uint2 pc = *reinterpret_cast<const uint2*>(bp + goff[0][2]);
      acc = ud_mma<T>(wf[0], make_uint4(pa.x, pa.y, pb.x, pb.y), acc);
      acc = ud_mma<T>(wf[1], make_uint4(pc.x, pc.y, pc.x, pc.y), acc);
    } else {
#pragma unroll
      for (int s = 0; s < STEPS; ++s) {
        uint32_t h[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) h[i] = *reinterpret_cast<const uint32_t*>(bp + goff[s][i]);
        acc = ud_mma<T>(wf[s], make_uint4(h[0], h[1], h[2], h[3]), acc);
      }
    }
    const int gy = y0 + y, gx = x0 + x;
    if (gy < a.H && gx < a.W) {     // lane: output channels 4 lq .. 4 lq + 3 of this pixel
      const float v0 = acc[0] + bias.x, v1 = acc[1] + bias.y, v2 = acc[2] + bias.z, v3 = acc[3] + bias.w;
      char* dst = a.g0 + (size_t)img * plane * ((size_t)a.g0_stride * E) + __umul24(__umul24(gy, a.W) + gx, (unsigned)a.g0_stride * E) + lq * 4 * E;
      if constexpr (E == 2) *reinterpret_cast<uint2*>(dst) = make_uint2(Half<T>::pack(v0, v1), Half<T>::pack(v2, v3));
      else *reinterpret_cast<float4*>(dst) = make_float4(v0, v1, v2, v3);
    }
    if constexpr (TR) {   // the transition's term of this pixel's g0 channels (from the STORED values) and of its base channels
      const uint32_t u0 = Half<T>::pack(acc[0] + bias.x, acc[1] + bias.y), u1 = Half<T>::pack(acc[2] + bias.z, acc[3] + bias.w);
      const f32x2 r0 = __builtin_elementwise_fma(f32x2{Half<T>::lo(u0), Half<T>::hi(u0)}, trs[0], trb[0]);
      const f32x2 r1 = __builtin_elementwise_fma(f32x2{Half<T>::lo(u1), Half<T>::hi(u1)}, trs[1], trb[1]);
      const uint32_t t0 = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(half_bits<T>(r0), i16x2{0, 0}));
      const uint32_t t1 = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(half_bits<T>(r1), i16x2{0, 0}));
      const uint2 tb = *reinterpret_cast<const uint2*>(trpatch + (y * UD_TILE + x) * 4);
      const bool mine = (lp >> 2) == ps;       // (one register set for the four A operands: 92 -> 80 VGPRs, 5 -> 6 waves per SIMD)
      tacc = ud_mma<T>(make_uint4(mine ? tra.x : 0u, mine ? tra.y : 0u, mine ? tra.z : 0u, mine ? tra.w : 0u), make_uint4(t0, t1, tb.x, tb.y), tacc);
    }
  }
  if constexpr (TR) {   // lane (lq, lp): the 3 partial outputs of pixel lp of row subtile lq
    const int blk = (wave * NPS + lq) * 4 + (lp >> 2);
    const int y = 2 * (blk / (UD_TILE / 2)) + ((lp >> 1) & 1), x = 2 * (blk % (UD_TILE / 2)) + (lp & 1);
    const int gy = y0 + y, gx = x0 + x;
    if (gy < a.H && gx < a.W)
      *reinterpret_cast<float4*>(reinterpret_cast<char*>(a.tr_out + (size_t)img * plane * 4) + (__umul24(gy, a.W) + gx) * 16u) = make_float4(tacc[0], tacc[1], tacc[2], 0.f);
  }
  USTAMP(5);
#ifdef EXP_UDSTAMPS
  if (udbg && tid == 0) { unsigned long long r_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r_) :: "memory"); udbg[(size_t)blockIdx.x * 8 + 6] = r_; }
#endif
}

}  // namespace mdie

using namespace mdie;

extern "C" int mdie_up_add_dense0_fwd(const mdie_up_dense0_desc* d, void* stream) {
  MDIE_REQUIRE(d != nullptr, "mdie_up_add_dense0_fwd: null descriptor");
  MDIE_REQUIRE(dtype_valid(d->dtype), "mdie_up_add_dense0_fwd: bad dtype %d", d->dtype);
  MDIE_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->H % 2 == 0 && d->W % 2 == 0, "mdie_up_add_dense0_fwd: extent %dx%dx%d (H, W even)", d->B, d->H, d->W);
  MDIE_REQUIRE(d->lo && d->x && d->base && d->weight && d->pre_scale && d->pre_shift && d->bias && d->g0, "mdie_up_add_dense0_fwd: null pointer");
  const int vec = dtype_vec(d->dtype);
  MDIE_REQUIRE(d->lo_stride >= 4 && d->lo_stride % 4 == 0 && ((uintptr_t)d->lo & 15) == 0,
               "mdie_up_add_dense0_fwd: lo must be 16-byte aligned with a pixel stride that is a multiple of 4 channels (%d)", d->lo_stride);
  MDIE_REQUIRE(d->base_channels == 16 || d->base_channels == vec, "mdie_up_add_dense0_fwd: base_channels %d (16 or %d)", d->base_channels, vec);
  const int base_stride = d->base_stride ? d->base_stride : d->base_channels;
  MDIE_REQUIRE(base_stride == d->base_channels || (d->tr && d->dtype != MDIE_F32 && d->base_channels == 8 && base_stride == 4),
               "mdie_up_add_dense0_fwd: base_stride %d (base_channels %d; 4 of 8 only for the 16-bit types with tr)", base_stride, d->base_channels);
  MDIE_REQUIRE(((uintptr_t)d->base & 15) == 0, "mdie_up_add_dense0_fwd: base must be 16-byte aligned");
  MDIE_REQUIRE(d->g0_stride >= 16 && d->g0_stride % 4 == 0 && (((uintptr_t)d->g0 | (uintptr_t)d->base | (uintptr_t)d->weight) & 15) == 0,
               "mdie_up_add_dense0_fwd: g0_stride %d / alignment", d->g0_stride);
  MDIE_REQUIRE((size_t)d->H * d->W < ((size_t)1 << 24) && (size_t)d->H * d->W * (size_t)std::max(d->g0_stride, 16) * dtype_size(d->dtype) < ((size_t)1 << 32) &&
               (size_t)(d->H / 2) * (d->W / 2) * d->lo_stride * dtype_size(d->dtype) < ((size_t)1 << 32) && d->lo_stride * 4 < (1 << 24) && d->g0_stride * 4 < (1 << 24),
               "mdie_up_add_dense0_fwd: image too large for the kernel's 24-bit pixel / 32-bit byte offsets (%dx%d)", d->H, d->W);
  UpDense0Args a{};
  a.B = d->B; a.H = d->H; a.W = d->W;
  a.lo = reinterpret_cast<const char*>(d->lo); a.lo_stride = d->lo_stride;
  a.x = d->x;
  a.base = reinterpret_cast<char*>(d->base); a.base_ch = d->base_channels;
  a.weight = reinterpret_cast<const char*>(d->weight);
  a.pre_scale = d->pre_scale; a.pre_shift = d->pre_shift; a.bias = d->bias;
  a.g0 = reinterpret_cast<char*>(d->g0); a.g0_stride = d->g0_stride;
  a.delta = d->blob_delta;
#ifdef EXP_UDSTAMPS
  a.dbg = g_ud_dbg;
#endif
  if (d->tr) {
    MDIE_REQUIRE(d->dtype != MDIE_F32 && d->base_channels == vec, "mdie_up_add_dense0_fwd: tr needs a 16-bit type with the base stored as one 16-byte group");
    MDIE_REQUIRE(d->tr->weight && d->tr->pre_scale && d->tr->pre_shift && d->tr->partial_out && d->tr->c0 >= vec && d->tr->c0 % 8 == 0 && !d->tr->out_nchw3,
                 "mdie_up_add_dense0_fwd: tr needs weight, pre_scale / pre_shift, partial_out and c0 (a multiple of 8 behind the base group); it is never the last producer");
    MDIE_REQUIRE((((uintptr_t)d->tr->partial_out | (uintptr_t)d->tr->weight) & 15) == 0 && (size_t)d->H * d->W * 16 < ((size_t)1 << 32), "mdie_up_add_dense0_fwd: tr alignment / extent");
    a.tr_w = reinterpret_cast<const char*>(d->tr->weight); a.tr_c0 = d->tr->c0;
    a.tr_scale = d->tr->pre_scale; a.tr_shift = d->tr->pre_shift; a.tr_out = d->tr->partial_out;
  }
  const int grid = cdiv(d->W, UD_TILE) * cdiv(d->H, UD_TILE) * d->B;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  TimedLaunch tl(MDIE_K_CONV3);
  if (d->tr) {
    if (base_stride == 4) {
      if (d->dtype == MDIE_BF16) hipLaunchKernelGGL((up_dense0_kernel<bf16, 8, true, 4>), dim3(grid), dim3(UD_THREADS), 0, s, a);
      else hipLaunchKernelGGL((up_dense0_kernel<f16, 8, true, 4>), dim3(grid), dim3(UD_THREADS), 0, s, a);
    } else {
      if (d->dtype == MDIE_BF16) hipLaunchKernelGGL((up_dense0_kernel<bf16, 8, true>), dim3(grid), dim3(UD_THREADS), 0, s, a);
      else hipLaunchKernelGGL((up_dense0_kernel<f16, 8, true>), dim3(grid), dim3(UD_THREADS), 0, s, a);
    }
  } else
  MDIE_SWITCH_T(d->dtype,
    if (d->base_channels == 16) hipLaunchKernelGGL((up_dense0_kernel<T, 16>), dim3(grid), dim3(UD_THREADS), 0, s, a);
    else hipLaunchKernelGGL((up_dense0_kernel<T, Traits<T>::VEC>), dim3(grid), dim3(UD_THREADS), 0, s, a));
  MDIE_LAUNCH_CHECK("mdie_up_add_dense0_fwd");
  return MDIE_OK;
}

#ifdef EXP_UDSTAMPS
extern "C" void mdie_exp_set_ud_dbg(void* p) { mdie::g_ud_dbg = (unsigned long long*)p; }
#endif
