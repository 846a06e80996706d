// Pieces shared by the convolution kernels (conv.hip, conv_wide.hip): launch arguments, the MFMA step per element type, the
// pixel <-> lane map of a 16x16 tile and the fused epilogue (affine, activation, residual, 2x2 max-pool, NHWC store, CBAM
// pooling partials).  See conv.hip for the formulation.
#pragma once
#include <type_traits>
#include "common.hpp"

namespace mdie {

constexpr int CONV_THREADS = 256;
constexpr int SMALL_GRID_WGS = 512;   // fewer 16x16-tile workgroups than this -> 8x8 tiles
constexpr int PWP = 24;  // LDS patch row pitch in pixels: >= TILE+2 and == 8 (mod 16)

struct SegDev {
  const char* ptr;
  int ch_begin, ch_end;  // stored channel range [begin, end)
  int stride;            // elements per pixel
};

struct EpiArgs {
  int H, W;
  const float* post_scale;
  const float* post_shift;
  int act, pool;
  const char* residual;
  int res_stride;
  char* out;
  int out_stride;
  float* nchw3;  // optional fp32 NCHW [B,3,Ho,Wo] destination for output channels 0..2
  long out_gs;   // PLANAR kernels only: elements between consecutive 16-channel groups of the output (mdie_conv_desc.out_group_stride)
  // BNRED kernels only (mdie_conv_desc.bnred): the tensor x whose gradient this convolution's output is, per stored output
  // channel the constants of the BatchNorm + ReLU that was applied to it, and where the per-tile sums go
  SegDev bx[MDIE_MAX_SEG];
  int bx_nseg;
  const float* b_scale;
  const float* b_shift;
  float* b_partial;   // [B * tiles per image][2][cout]: sum of dz, sum of dz * x   (dz = output * [x * scale + shift > 0])
};

struct ConvArgs {
  int B, H, W;
  int tiles_x, tiles_y, n_tiles;
  int cin, nchunk, cout;
  int nseg;
  SegDev seg[MDIE_MAX_SEG];
  const float* pre_scale;
  const float* pre_shift;
  const char* weight;
  float* pool_partial;   // STATS kernels: [B][tiles per image][2][cout] channel sums / maxima of the output
  EpiArgs e;
  // several weight sets in one launch (mdie_conv_desc.blob_delta): image b adds delta[b] bytes to weight, pre_scale / pre_shift
  // and e.post_scale / e.post_shift (param_shift below); nullptr = one weight set
  const long long* delta;
};

// parameter pointer of image `img`'s weight set
template <typename P> __device__ __forceinline__ const P* param_shift(const P* p, long long dl) {
  return reinterpret_cast<const P*>(reinterpret_cast<const char*>(p) + dl);
}

template <typename T> __device__ __forceinline__ f32x4 mma16(const uint4& w, const uint4& x, f32x4 acc);
template <> __device__ __forceinline__ f32x4 mma16<bf16>(const uint4& w, const uint4& x, f32x4 acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4 mma16<f16>(const uint4& w, const uint4& x, f32x4 acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4 mma16<float>(const uint4& w, const uint4& x, f32x4 acc) {
  // lane group g = lane>>4 holds channels 4g..4g+3; MFMA j pairs channel 4g+j of both operands
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.x), __uint_as_float(x.x), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.y), __uint_as_float(x.y), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.z), __uint_as_float(x.z), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.w), __uint_as_float(x.w), acc, 0, 0, 0);
  return acc;
}

// pixel of the tile held by (pixel-subtile ps, lane column p): 4 consecutive 2x2 blocks per subtile
template <int TILE>
__device__ __forceinline__ void tile_pixel(int ps, int p, int& y, int& x) {
  constexpr int BPR = TILE / 2;
  const int blk = ps * 4 + (p >> 2);
  y = 2 * (blk / BPR) + ((p >> 1) & 1);
  x = 2 * (blk % BPR) + (p & 1);
}
// A wave's NPS subtiles start on a multiple of NPS, so subtile ps of the wave sits at a compile-time offset from
// subtile 0 (TILE = 16: 4 subtiles = a 4x16 strip of rows, dy = 2*(ps/2), dx = 8*(ps%2)): every per-subtile address
// is one lane-dependent base plus a wave-uniform constant, instead of a fresh index computation per subtile.
template <int TILE, int NPS> struct TileStep {
  static_assert(NPS == 1 || (TILE == 16 && NPS == 4), "subtile offsets are derived for 8x8 (1 subtile) and 16x16 (4 subtiles) tiles");
  __device__ __forceinline__ static constexpr int dy(int ps) { return 2 * (ps >> 1); }
  __device__ __forceinline__ static constexpr int dx(int ps) { return 8 * (ps & 1); }
};

// ---- epilogue: affine, activation, residual, 2x2 max-pool, NHWC store -----------------------------------------
// max over the 4 lanes of a quad (the 2x2 pooling window) with DPP quad_perm swaps: no LDS traffic
__device__ __forceinline__ float quad_max(float v) {
  const int a = __builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
  v = fmaxf(v, __int_as_float(a));
  const int b = __builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
  return fmaxf(v, __int_as_float(b));
}

// the same for values known to be >= 0 (a ReLU output): non-negative floats order like their bit patterns, and an integer
// max needs no NaN canonicalisation, so each lane swap folds into the max (v_max_i32_dpp): 2 instructions instead of 6
__device__ __forceinline__ float quad_max_nonneg(float v) {
  int a = __float_as_int(v);
  a = max(a, __builtin_amdgcn_mov_dpp(a, 0xB1, 0xF, 0xF, true));
  a = max(a, __builtin_amdgcn_mov_dpp(a, 0x4E, 0xF, 0xF, true));
  return __int_as_float(a);
}

template <int ACT> __device__ __forceinline__ float act_fn(float v) {
  if constexpr (ACT == MDIE_ACT_RELU) return fmaxf(v, 0.0f);
  else if constexpr (ACT == MDIE_ACT_SIGMOID) return sigmoidf(v);
  else return v;
}

// PLANAR: output channel group n / 16 lives at out + (n / 16) * e.out_gs (its pixels out_stride apart) instead of at out + n -- the
// input gradient of a DenseBlock layer written one plane per 16 channels, so that the passes which later gather ONE feature
// segment out of five such tensors read dense streams (mdie_bn_bwd_apply_multi).  Compile-time: only conv_planar.hip's
// instantiations carry it.
template <typename T, int NCS, int NPS, int TILE, int ACT, bool POOL, bool STATS = false, bool PLANAR = false>
__device__ __forceinline__ void conv_epilogue_t(const EpiArgs& e, const float4 (&esc)[NCS], const float4 (&esh)[NCS],
                                                f32x4 (&acc)[NCS][NPS], int img, int y0, int x0, int n0, int ps_base, int lq, int lp,
                                                float (*st_sum)[4] = nullptr, float (*st_max)[4] = nullptr) {
  using TS = TileStep<TILE, NPS>;
  constexpr int SH = POOL ? 1 : 0;
  const int Ho = e.H >> SH, Wo = e.W >> SH;
  int ty0, tx0;
  tile_pixel<TILE>(ps_base, lp, ty0, tx0);
  const int gy0 = y0 + ty0, gx0 = x0 + tx0;
  const int oy0 = gy0 >> SH, ox0 = gx0 >> SH;
  const size_t opix0 = ((size_t)img * Ho + oy0) * Wo + ox0;
  const ptrdiff_t ogs = PLANAR ? (ptrdiff_t)e.out_gs : 16;      // elements from one 16-channel group to the next
  T* const orow0 = reinterpret_cast<T*>(e.out) + opix0 * e.out_stride + (PLANAR ? (ptrdiff_t)(n0 >> 4) * ogs : (ptrdiff_t)n0) + lq * 4;
  const T* const rrow0 = e.residual ? reinterpret_cast<const T*>(e.residual) + opix0 * e.res_stride + n0 + lq * 4 : nullptr;
  if constexpr (sizeof(T) == 2 && (ACT == MDIE_ACT_RELU || (ACT == MDIE_ACT_NONE && !POOL)) && !STATS) {
    // 16-bit output + ReLU (or no activation, unpooled: the DenseLayers), no residual: the whole epilogue in packed form -- v_pk_fma_f32 for the affine, one v_cvt_pk_{bf16,f16}_f32 per pair,
    // ReLU and the 2x2 max on the ROUNDED halves as packed 16-bit integer maxima (rounding is monotonic and keeps the sign, so
    // this equals rounding relu(max(...)) of the fp32 values): 14 instead of 18 vector instructions per 4 channels when
    // pooling, 6 instead of 10 without
    if (!rrow0 && !e.nchw3) {   // launch-uniform
#pragma unroll
      for (int ps = 0; ps < NPS; ++ps) {
        const int gy = gy0 + TS::dy(ps), gx = gx0 + TS::dx(ps);
        const bool inside = gy < e.H && gx < e.W;
        const int dpix = (TS::dy(ps) >> SH) * Wo + (TS::dx(ps) >> SH);   // wave-uniform
        T* orow = orow0 + (ptrdiff_t)dpix * e.out_stride;
        uint2 sel = make_uint2(0u, 0u);
#pragma unroll
        for (int cs = 0; cs < NCS; ++cs) {
          const f32x2 lo = __builtin_elementwise_fma(f32x2{acc[cs][ps][0], acc[cs][ps][1]}, f32x2{esc[cs].x, esc[cs].y}, f32x2{esh[cs].x, esh[cs].y});
          const f32x2 hi = __builtin_elementwise_fma(f32x2{acc[cs][ps][2], acc[cs][ps][3]}, f32x2{esc[cs].z, esc[cs].w}, f32x2{esh[cs].z, esh[cs].w});
          i16x2 p0 = half_bits<T>(lo);
          i16x2 p1 = half_bits<T>(hi);
          if constexpr (ACT == MDIE_ACT_RELU) { p0 = __builtin_elementwise_max(p0, i16x2{0, 0}); p1 = __builtin_elementwise_max(p1, i16x2{0, 0}); }
          if constexpr (POOL) {   // non-negative bf16 order like their bit patterns
            p0 = __builtin_elementwise_max(p0, __builtin_bit_cast(i16x2, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, p0), 0xB1, 0xF, 0xF, true)));
            p1 = __builtin_elementwise_max(p1, __builtin_bit_cast(i16x2, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, p1), 0xB1, 0xF, 0xF, true)));
            p0 = __builtin_elementwise_max(p0, __builtin_bit_cast(i16x2, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, p0), 0x4E, 0xF, 0xF, true)));
            p1 = __builtin_elementwise_max(p1, __builtin_bit_cast(i16x2, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, p1), 0x4E, 0xF, 0xF, true)));
          }
          const uint2 packed = make_uint2(__builtin_bit_cast(uint32_t, p0), __builtin_bit_cast(uint32_t, p1));
          if constexpr (POOL && NCS == 4 && NPS == 4) {   // lane j of the quad keeps channel group j (one full-lane store below)
            if ((lp & 3) == cs) sel = packed;
          } else {
            const bool writer = POOL ? (inside && (lp & 3) == 0) : inside;
            if (writer) *reinterpret_cast<uint2*>(orow + cs * ogs) = packed;
          }
        }
        if constexpr (POOL && NCS == 4 && NPS == 4) {
          if (inside) *reinterpret_cast<uint2*>(orow + (lp & 3) * 16) = sel;
        }
      }
      return;
    }
  }
#pragma unroll
  for (int ps = 0; ps < NPS; ++ps) {
    const int gy = gy0 + TS::dy(ps), gx = gx0 + TS::dx(ps);
    const bool inside = gy < e.H && gx < e.W;
    const bool writer = POOL ? (inside && (lp & 3) == 0) : inside;
    const int oy = oy0 + (TS::dy(ps) >> SH), ox = ox0 + (TS::dx(ps) >> SH);
    const int dpix = (TS::dy(ps) >> SH) * Wo + (TS::dx(ps) >> SH);   // wave-uniform
    T* orow = orow0 + (ptrdiff_t)dpix * e.out_stride;
    const T* rrow = rrow0 ? rrow0 + (ptrdiff_t)dpix * e.res_stride : nullptr;
    if constexpr (POOL && NCS == 4 && NPS == 4 && !STATS) {   // (conv_kernel's 16x16 tiles; conv_first_kernel, one subtile per call at 128 VGPRs, spills with it)
      // Pooled 64-wide tile: after the lane max all 4 lanes of a quad hold the pooled pixel, so instead of lane 0 storing
      // its 4 channels once per 16-channel subtile (4 store instructions with a quarter of the lanes active), lane j of the
      // quad stores subtile j: ONE store instruction with every lane active writes the pixel's 64 channels.
      if (!e.nchw3) {   // launch-uniform
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        const int j = lp & 3;
#pragma unroll
        for (int cs = 0; cs < NCS; ++cs) {
          const float4 sc = esc[cs], sh = esh[cs];
          float v[4];
          v[0] = act_fn<ACT>(fmaf(acc[cs][ps][0], sc.x, sh.x));
          v[1] = act_fn<ACT>(fmaf(acc[cs][ps][1], sc.y, sh.y));
          v[2] = act_fn<ACT>(fmaf(acc[cs][ps][2], sc.z, sh.z));
          v[3] = act_fn<ACT>(fmaf(acc[cs][ps][3], sc.w, sh.w));
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            v[i] = ACT == MDIE_ACT_RELU ? quad_max_nonneg(v[i]) : quad_max(v[i]);
            o[i] = j == cs ? v[i] : o[i];
          }
        }
        if (inside) {
          if (rrow) {
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] += ld(rrow + j * 16 + i);
          }
          if constexpr (sizeof(T) == 4) *reinterpret_cast<float4*>(orow + j * 16) = make_float4(o[0], o[1], o[2], o[3]);
          else *reinterpret_cast<uint2*>(orow + j * 16) = make_uint2(Half<T>::pack(o[0], o[1]), Half<T>::pack(o[2], o[3]));
        }
        continue;
      }
    }
#pragma unroll
    for (int cs = 0; cs < NCS; ++cs) {
      const float4 sc = esc[cs], sh = esh[cs];
      float v[4];
      v[0] = act_fn<ACT>(fmaf(acc[cs][ps][0], sc.x, sh.x));
      v[1] = act_fn<ACT>(fmaf(acc[cs][ps][1], sc.y, sh.y));
      v[2] = act_fn<ACT>(fmaf(acc[cs][ps][2], sc.z, sh.z));
      v[3] = act_fn<ACT>(fmaf(acc[cs][ps][3], sc.w, sh.w));
      if constexpr (POOL) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = ACT == MDIE_ACT_RELU ? quad_max_nonneg(v[i]) : quad_max(v[i]);
      }
      if (e.nchw3) {  // final tensor of the network: channels 0..2 straight to fp32 NCHW planes
        if (writer && cs == 0 && lq == 0 && n0 == 0) {
          const size_t plane = (size_t)Ho * Wo;
          float* o = e.nchw3 + (size_t)img * 3 * plane + (size_t)oy * Wo + ox;
          o[0] = v[0]; o[plane] = v[1]; o[2 * plane] = v[2];
        }
      } else
#ifdef EXP_NO_STORE
      if (writer && v[0] == 1234.5f) {
#else
      if (writer) {
#endif
        if (rrow) {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] += ld(rrow + cs * 16 + i);
        }
        if constexpr (STATS) {   // channel sums / maxima of the STORED values: what a separate pooling pass over `out` would read
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float qv = v[i];
            if constexpr (sizeof(T) == 2) qv = (float)(T)qv;
            st_sum[cs][i] += qv;
            st_max[cs][i] = fmaxf(st_max[cs][i], qv);
          }
        }
        if constexpr (sizeof(T) == 4) {
          *reinterpret_cast<float4*>(orow + cs * ogs) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          *reinterpret_cast<uint2*>(orow + cs * ogs) = make_uint2(Half<T>::pack(v[0], v[1]), Half<T>::pack(v[2], v[3]));
        }
      }
    }
  }
}

// PLANAR + BNRED epilogue (conv_planar.hip): store the plain output one plane per 16 channels AND accumulate, per output channel,
//   s1 += dz,  s2 += dz * x      with dz = stored output * [x * b_scale + b_shift > 0]
// -- the two sums of the BatchNorm-ReLU backward of the tensor x this output is the gradient of (sum dz * xhat follows in the
// finishing kernel as invstd * (s2 - mean * s1): no per-channel mean / invstd in registers here).  x is read at the lane's
// pixel and 4 channels: 8 (16-bit) / 16 (fp32) bytes per subtile and channel group, every byte of a line used across the wave.
// Channel-group outer, subtile inner: one group's constants and sums are live at a time (BN = 64 would hold 64 registers otherwise).
template <typename T, int NCS, int NPS, int TILE>
__device__ __forceinline__ void conv_epilogue_bnred(const EpiArgs& e, const float4 (&esc)[NCS], const float4 (&esh)[NCS], f32x4 (&acc)[NCS][NPS], int img, int y0,
                                                    int x0, int n0, int ps_base, int lq, int lp, float* red, int wave, int BN) {
  using TS = TileStep<TILE, NPS>;
  int ty0, tx0;
  tile_pixel<TILE>(ps_base, lp, ty0, tx0);
  const int gy0 = y0 + ty0, gx0 = x0 + tx0;
  const size_t pix0 = ((size_t)img * e.H + gy0) * e.W + gx0;
  const ptrdiff_t ogs = (ptrdiff_t)e.out_gs;
  T* const orow0 = reinterpret_cast<T*>(e.out) + pix0 * e.out_stride + (ptrdiff_t)(n0 >> 4) * ogs + lq * 4;
  // EVERY x load of the tile (all channel groups, all subtiles) and the groups' constants are requested up front: the stores below sit
  // under per-lane branches, across which the compiler does not move the next group's loads -- group by group this was NCS exposed
  // memory round trips behind the MFMA phase (4 for the 64-output input-gradient layers of the DenseBlocks: 1.2 ms of an 8.3 ms step)
  typedef typename std::conditional<sizeof(T) == 4, float4, uint2>::type RawX;
  RawX xr[NCS][NPS];
  float4 bsc_[NCS], bsh_[NCS];
  bool inside[NPS];
#pragma unroll
  for (int ps = 0; ps < NPS; ++ps) inside[ps] = gy0 + TS::dy(ps) < e.H && gx0 + TS::dx(ps) < e.W;
#pragma unroll
  for (int cs = 0; cs < NCS; ++cs) {
    const int c16 = n0 + cs * 16;                                   // wave-uniform: first stored channel of this group
    const T* xb = nullptr; int xs = 0;
#pragma unroll
    for (int k = 0; k < MDIE_MAX_SEG; ++k)
      if (k < e.bx_nseg && c16 >= e.bx[k].ch_begin && c16 < e.bx[k].ch_end) {
        xb = reinterpret_cast<const T*>(e.bx[k].ptr) + (c16 - e.bx[k].ch_begin) + lq * 4; xs = e.bx[k].stride;
      }
    bsc_[cs] = *reinterpret_cast<const float4*>(e.b_scale + c16 + lq * 4); bsh_[cs] = *reinterpret_cast<const float4*>(e.b_shift + c16 + lq * 4);
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps) {
      const size_t pix = inside[ps] ? pix0 + (size_t)TS::dy(ps) * e.W + TS::dx(ps) : pix0 - (size_t)ty0 * e.W - tx0;   // (outside: the tile's first pixel, not used)
      xr[cs][ps] = *reinterpret_cast<const RawX*>(xb + pix * xs);
    }
  }
#pragma unroll
  for (int cs = 0; cs < NCS; ++cs) {
    const float4 bsc = bsc_[cs], bsh = bsh_[cs];
    const float bs[4] = {bsc.x, bsc.y, bsc.z, bsc.w}, bh[4] = {bsh.x, bsh.y, bsh.z, bsh.w};
    const float sc[4] = {esc[cs].x, esc[cs].y, esc[cs].z, esc[cs].w}, sh[4] = {esh[cs].x, esh[cs].y, esh[cs].z, esh[cs].w};
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    float xv[NPS][4];
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps) {
      if constexpr (sizeof(T) == 4) {
        xv[ps][0] = xr[cs][ps].x; xv[ps][1] = xr[cs][ps].y; xv[ps][2] = xr[cs][ps].z; xv[ps][3] = xr[cs][ps].w;
      } else {
        xv[ps][0] = Half<T>::lo(xr[cs][ps].x); xv[ps][1] = Half<T>::hi(xr[cs][ps].x); xv[ps][2] = Half<T>::lo(xr[cs][ps].y); xv[ps][3] = Half<T>::hi(xr[cs][ps].y);
      }
    }
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps) {
      const int dpix = TS::dy(ps) * e.W + TS::dx(ps);               // wave-uniform
      T* orow = orow0 + (ptrdiff_t)dpix * e.out_stride + cs * ogs;
      float v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = fmaf(acc[cs][ps][i], sc[i], sh[i]);
      if constexpr (sizeof(T) == 4) {
        if (inside[ps]) *reinterpret_cast<float4*>(orow) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
        const uint2 pk = make_uint2(Half<T>::pack(v[0], v[1]), Half<T>::pack(v[2], v[3]));
        if (inside[ps]) *reinterpret_cast<uint2*>(orow) = pk;
        v[0] = Half<T>::lo(pk.x); v[1] = Half<T>::hi(pk.x); v[2] = Half<T>::lo(pk.y); v[3] = Half<T>::hi(pk.y);   // the STORED values: what a pass over `out` would read
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float dz = (inside[ps] && fmaf(xv[ps][i], bs[i], bh[i]) > 0.f) ? v[i] : 0.f;
        s1[i] += dz;
        s2[i] = fmaf(dz, xv[ps][i], s2[i]);
      }
    }
    // over the 16 pixel lanes of the row group (same lq = same 4 channels); lane lp == 0 of each group writes the wave's sums
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int d = 8; d > 0; d >>= 1) { s1[i] += __shfl_xor(s1[i], d); s2[i] += __shfl_xor(s2[i], d); }
    if (lp == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { red[(wave * 2 + 0) * BN + cs * 16 + lq * 4 + i] = s1[i]; red[(wave * 2 + 1) * BN + cs * 16 + lq * 4 + i] = s2[i]; }
    }
  }
}

// runtime -> compile-time dispatch (act and pool are launch-uniform, so this is one scalar branch)
template <typename T, int NCS, int NPS, int TILE>
__device__ __forceinline__ void conv_epilogue(const EpiArgs& e, const float4 (&esc)[NCS], const float4 (&esh)[NCS],
                                              f32x4 (&acc)[NCS][NPS], int img, int y0, int x0, int n0, int ps_base, int lq, int lp) {
  if (e.pool) {
    // only ReLU is ever pooled on this path (encoder blocks, models/cdan.py:74-75)
    if (e.act == MDIE_ACT_RELU) conv_epilogue_t<T, NCS, NPS, TILE, MDIE_ACT_RELU, true>(e, esc, esh, acc, img, y0, x0, n0, ps_base, lq, lp);
    else if (e.act == MDIE_ACT_SIGMOID) conv_epilogue_t<T, NCS, NPS, TILE, MDIE_ACT_SIGMOID, true>(e, esc, esh, acc, img, y0, x0, n0, ps_base, lq, lp);
    else conv_epilogue_t<T, NCS, NPS, TILE, MDIE_ACT_NONE, true>(e, esc, esh, acc, img, y0, x0, n0, ps_base, lq, lp);
  } else {
    if (e.act == MDIE_ACT_RELU) conv_epilogue_t<T, NCS, NPS, TILE, MDIE_ACT_RELU, false>(e, esc, esh, acc, img, y0, x0, n0, ps_base, lq, lp);
    else if (e.act == MDIE_ACT_SIGMOID) conv_epilogue_t<T, NCS, NPS, TILE, MDIE_ACT_SIGMOID, false>(e, esc, esh, acc, img, y0, x0, n0, ps_base, lq, lp);
    else conv_epilogue_t<T, NCS, NPS, TILE, MDIE_ACT_NONE, false>(e, esc, esh, acc, img, y0, x0, n0, ps_base, lq, lp);
  }
}


// conv_wide.hip: LDS-DMA staged 3x3 convolution for wide layers (>= 64 channels in and out); MDIE_OK, or MDIE_EINVAL when
// the shape is not one it handles (the caller then uses conv_kernel)
bool conv_wide_applicable(int dtype, const ConvArgs& a, int ksize, bool has_nchw3);
int launch_conv_wide(int dtype, ConvArgs& a, hipStream_t stream, bool yield_cu = false);   // yield_cu: twice the workgroups, each with half the items (mdie_conv_desc.share_cu = 2)
// conv_thin.hip: persistent, register-prefetched 3x3 convolution for pre-activated 16-output layers with <= 64 stored input
// channels on full 16x16 tiles (decoder.final_dense)
bool conv_thin_applicable(int dtype, const ConvArgs& a, int ksize, bool has_nchw3, bool any_batch = false);
int launch_conv_thin(int dtype, const ConvArgs& a, hipStream_t stream, const mdie_tr_fuse* tr = nullptr);
// conv_ksplit.hip: the deep DenseBlock layers (16 outputs, >= 128 pre-activated input channels, maps up to 64x64): K split over
// the four waves of a workgroup, barrier-free chunk loop.  Chosen by (layer, map) only, never by the batch.
bool conv_ksplit_applicable(int dtype, const ConvArgs& a, int ksize, bool has_nchw3);
int launch_conv_ksplit(int dtype, ConvArgs& a, hipStream_t stream);
// conv_planar.hip: conv_kernel with the output written one plane per 16 channels (no activation, pooling, residual): the input
// gradients of the DenseBlock layers in training
int launch_conv_planar(int dtype, ConvArgs& a, int ksize, hipStream_t stream);
int conv_planar_tile(int B, int H, int W, int cout);

}  // namespace mdie
