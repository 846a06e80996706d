// Fused decoder tail (gfx950):
//   y = sigmoid( DenseBlock(3,3,16,4)( bilinear_x2(lo) + x ) )          models/cdan.py:153-157
// i.e. F.interpolate + torch.add (:153-154), decoder.final_dense (:119,156 with the layer recipes
// at :41-53) and nn.Sigmoid (:157) in ONE kernel.  Unfused, this block moves 640 B per output pixel
// through HBM (it is 30 % of the network's algorithmic traffic, SURVEY.md 8d: 31.7 MB/img); fused,
// the four 16-channel growth maps never leave LDS and HBM sees the 3-channel input (with halo) and
// the 3-channel output only.
//
// A 256-thread workgroup owns an 8x16 output tile.  Layer l (1..4) produces its growth map on the
// tile extended by 4-l pixels (halo recompute), so region k is (8+2(4-k)) x (16+2(4-k)):
//   base 16x24 -> g1 14x22 -> g2 12x20 -> g3 10x18 -> g4 8x16 -> 1x1 transition 8x16.
// Per layer: (a) "act pass": relu(bn_l(.)) of every input channel into an LDS operand image, zero
// outside the picture (nn.Conv2d pads the activated tensor); (b) MFMA pass (rows = 16 couts,
// cols = 16 pixels, the same 16x16 MFMAs as conv.hip), raw growth output (+bias) back to LDS.
// K is packed in 1 KiB "steps" (16 couts x 64 B): the 3 base channels are im2col'ed (27 -> 32) into
// one bf16 step / two f32 steps and gathered from a 4-channel image; growth channels go 16 at a time,
// two (tap, group) half-steps per bf16 MFMA.
#include <math.h>
#include <string.h>

#include <string>

#include "common.hpp"

namespace mdie {

constexpr int TL_TH = 8, TL_TW = 16, TL_HALO = 4;
constexpr int TL_THREADS = 256;
constexpr int TL_FH = TL_TH + 2 * TL_HALO, TL_FW = TL_TW + 2 * TL_HALO;  // common frame 16 x 24
constexpr int TL_WROW = 80;                                               // weight row pitch in LDS

__host__ __device__ constexpr int tl_rh(int k) { return TL_TH + 2 * (TL_HALO - k); }
__host__ __device__ constexpr int tl_rw(int k) { return TL_TW + 2 * (TL_HALO - k); }
__host__ __device__ constexpr int tl_np(int k) { return tl_rh(k) * tl_rw(k); }

// per-layer constants (L = 1..4 dense layers, 5 = transition)
__host__ __device__ constexpr int tl_ng(int L) { return L - 1; }
__host__ __device__ constexpr int tl_ks(int L) { return L == 5 ? 1 : 3; }
__host__ __device__ constexpr int tl_kin(int L) { return L == 5 ? 4 : L - 1; }
__host__ __device__ constexpr int tl_kout(int L) { return L == 5 ? 4 : L; }
__host__ __device__ constexpr int tl_nh(int L) { return tl_ng(L) * tl_ks(L) * tl_ks(L); }  // growth half-steps
__host__ __device__ constexpr int tl_base_steps(int esz) { return esz == 2 ? 1 : 2; }
__host__ __device__ constexpr int tl_growth_steps(int L, int esz) { return esz == 2 ? (tl_nh(L) + 1) / 2 : tl_nh(L); }
__host__ __device__ constexpr int tl_steps(int L, int esz) { return tl_base_steps(esz) + tl_growth_steps(L, esz); }
__host__ __device__ constexpr int tl_cin(int L) { return 3 + 16 * tl_ng(L); }
__host__ __device__ constexpr int tl_act_pitch(int L, int esz) { return tl_ng(L) * 16 * esz + 16; }

struct TailParams {           // device pointers into the packed blob
  const float* pre_scale[5];  // [3 + 16*(L-1)]
  const float* pre_shift[5];
  const float* bias[5];       // [16]
  const char* w[5];           // tl_steps(L) x 1 KiB
};

struct TailArgs {
  int B, H, W;          // full resolution
  const char* lo;       // NHWC [B, H/2, W/2, lo_stride], channels 0..2 used; may be null
  int lo_stride;
  const float* x;       // NCHW fp32 [B,3,H,W]
  float* y;             // NCHW fp32 [B,3,H,W]
  TailParams p;
};

template <typename T> struct TailLds {
  static constexpr int E = sizeof(T);
  static constexpr int RAWBASE = 0;                                   // [FH*FW] float4
  static constexpr int ACTBASE = RAWBASE + TL_FH * TL_FW * 16;        // [FH*FW][4] T
  static constexpr int RAW1 = ACTBASE + TL_FH * TL_FW * 4 * E;        // region 1..4, [np][16] T
  static constexpr int RAW2 = RAW1 + tl_np(1) * 16 * E;
  static constexpr int RAW3 = RAW2 + tl_np(2) * 16 * E;
  static constexpr int RAW4 = RAW3 + tl_np(3) * 16 * E;
  static constexpr int ACT = RAW4 + tl_np(4) * 16 * E;                // operand image of the current layer
  static constexpr int cmax(int a, int b) { return a > b ? a : b; }
  static constexpr int ACT_BYTES = cmax(cmax(tl_np(1) * tl_act_pitch(2, E), tl_np(2) * tl_act_pitch(3, E)),
                                        cmax(tl_np(3) * tl_act_pitch(4, E), tl_np(4) * tl_act_pitch(5, E)));
  static constexpr int WGT = ACT + ACT_BYTES;
  static constexpr int WGT_BYTES = E == 2 ? 0 : tl_steps(4, E) * 16 * TL_WROW;   // bf16 keeps its weights in registers
  static constexpr int TOTAL = WGT + WGT_BYTES;
  __device__ static constexpr int raw(int j) { return j == 1 ? RAW1 : j == 2 ? RAW2 : j == 3 ? RAW3 : RAW4; }
};

template <typename T> __device__ __forceinline__ f32x4 tl_mma(const uint4& w, const uint4& x, f32x4 acc);
template <> __device__ __forceinline__ f32x4 tl_mma<bf16>(const uint4& w, const uint4& x, f32x4 acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4 tl_mma<f16>(const uint4& w, const uint4& x, f32x4 acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4 tl_mma<float>(const uint4& w, const uint4& x, f32x4 acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.x), __uint_as_float(x.x), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.y), __uint_as_float(x.y), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.z), __uint_as_float(x.z), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.w), __uint_as_float(x.w), acc, 0, 0, 0);
  return acc;
}

// ATen's half-pixel source index for scale 2 (see resample.hip)
__device__ __forceinline__ void tl_src(int dst, int in_size, int& i0, int& i1, float& l0, float& l1) {
  float src = ((float)dst + 0.5f) * 0.5f - 0.5f;
  src = src < 0.f ? 0.f : src;
  i0 = (int)src;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = src - (float)i0;
  l0 = 1.0f - l1;
}

template <typename T, int L>
__device__ __forceinline__ void tail_layer(const TailArgs& a, char* smem, int img, int ty0, int tx0) {
  using LD = TailLds<T>;
  constexpr int E = sizeof(T);
  constexpr int VEC = 16 / E;
  constexpr int NG = tl_ng(L), KS = tl_ks(L), KIN = tl_kin(L), KOUT = tl_kout(L);
  constexpr int NH = tl_nh(L);
  constexpr int BS = tl_base_steps(E), GS = tl_growth_steps(L, E), STEPS = BS + GS;
  constexpr int AP = tl_act_pitch(L, E);
  constexpr int RWI = tl_rw(KIN), NPI = tl_np(KIN);
  constexpr int RWO = tl_rw(KOUT), NPO = tl_np(KOUT);
  constexpr int PADK = KS / 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lq = lane >> 4, lp = lane & 15;

  // ---- (a) operand images --------------------------------------------------------------------------------
  // bf16: this lane's weight fragments of EVERY step of the layer go straight from global memory (L2-resident,
  // <= 15 KiB per layer) into registers, issued first so they land during the activation passes; each MFMA of the
  // pass below then costs one LDS read (its pixel operand) instead of two.  f32 (29 steps) keeps the LDS image.
  constexpr bool WREG = (E == 2);
  uint4 wreg[WREG ? STEPS : 1];
  if constexpr (WREG) {
#pragma unroll
    for (int st_ = 0; st_ < STEPS; ++st_) wreg[st_] = *reinterpret_cast<const uint4*>(a.p.w[L - 1] + ((size_t)st_ * 16 + lp) * 64 + lq * 16);
  } else {
    for (int u = tid; u < STEPS * 64; u += TL_THREADS) {
      const uint4 v = *reinterpret_cast<const uint4*>(a.p.w[L - 1] + (size_t)u * 16);
      *reinterpret_cast<uint4*>(smem + LD::WGT + (u >> 2) * TL_WROW + (u & 3) * 16) = v;
    }
  }
  // activated base (3 channels + zero) over the input region, common-frame coordinates
  {
    const float s0 = a.p.pre_scale[L - 1][0], s1 = a.p.pre_scale[L - 1][1], s2 = a.p.pre_scale[L - 1][2];
    const float b0 = a.p.pre_shift[L - 1][0], b1 = a.p.pre_shift[L - 1][1], b2 = a.p.pre_shift[L - 1][2];
    for (int p = tid; p < NPI; p += TL_THREADS) {
      const int py = p / RWI, px = p - py * RWI;
      const int fy = py + KIN, fx = px + KIN;
      const int gy = ty0 - TL_HALO + fy, gx = tx0 - TL_HALO + fx;
      const bool inside = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      const float4 r = *reinterpret_cast<const float4*>(smem + LD::RAWBASE + (fy * TL_FW + fx) * 16);
      const float v0 = inside ? fmaxf(fmaf(r.x, s0, b0), 0.f) : 0.f;
      const float v1 = inside ? fmaxf(fmaf(r.y, s1, b1), 0.f) : 0.f;
      const float v2 = inside ? fmaxf(fmaf(r.z, s2, b2), 0.f) : 0.f;
      T* dst = reinterpret_cast<T*>(smem + LD::ACTBASE) + (fy * TL_FW + fx) * 4;
      st(dst + 0, v0); st(dst + 1, v1); st(dst + 2, v2); st(dst + 3, 0.f);
    }
  }
  // activated growth channels: thread = (channel vector jv, pixel row r); BN constants stay in registers
  if constexpr (NG > 0) {
    constexpr int VPG = 16 / VEC;           // 16-byte vectors per 16-channel group
    constexpr int NJV = NG * VPG;
    constexpr int ROWS = TL_THREADS / NJV;
    const int jv = tid % NJV, r0 = tid / NJV;
    if (r0 < ROWS) {
      const int j = jv / VPG, v = jv - j * VPG;      // growth group j (0-based), vector v
      float sc[VEC], sh[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        sc[i] = a.p.pre_scale[L - 1][3 + j * 16 + v * VEC + i];
        sh[i] = a.p.pre_shift[L - 1][3 + j * 16 + v * VEC + i];
      }
      const int kj = j + 1;                          // region of growth map j
      const int rwj = TL_TW + 2 * (TL_HALO - kj);
      const int rawoff = (j == 0 ? LD::RAW1 : j == 1 ? LD::RAW2 : j == 2 ? LD::RAW3 : LD::RAW4);
      for (int p = r0; p < NPI; p += ROWS) {
        const int py = p / RWI, px = p - py * RWI;
        const int gy = ty0 - TL_HALO + py + KIN, gx = tx0 - TL_HALO + px + KIN;
        const bool inside = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        const int qy = py + KIN - kj, qx = px + KIN - kj;  // same pixel in region kj's own coordinates
        uint4 o = make_uint4(0, 0, 0, 0);
        if (inside) {
          const uint4 u = *reinterpret_cast<const uint4*>(smem + rawoff + (qy * rwj + qx) * 16 * E + v * 16);
          float f[VEC];
          Vec16<T>::unpack(u, f);
#pragma unroll
          for (int i = 0; i < VEC; ++i) f[i] = fmaxf(fmaf(f[i], sc[i], sh[i]), 0.f);
          o = Vec16<T>::pack(f);
        }
        *reinterpret_cast<uint4*>(smem + LD::ACT + p * AP + j * 16 * E + v * 16) = o;
      }
    }
  }
  __syncthreads();

  // ---- (b) MFMA pass ---------------------------------------------------------------------------------------
  // base gather offsets of this lane: element k = 8*lq + i of the im2col row (k = tap*3 + c)
  int boff[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int k = 8 * lq + i;
    const int tap = k / 3, c = k - tap * 3;
    const int dy = tap / KS, dx = tap - dy * KS;
    boff[i] = (tap < KS * KS) ? ((dy * TL_FW + dx) * 4 + c) * E : 0;
  }
  const float4 bias = *reinterpret_cast<const float4*>(a.p.bias[L - 1] + lq * 4);
  const char* wl = smem + LD::WGT + lp * TL_WROW + lq * 16;

  constexpr int NS = (NPO + 15) / 16;
  // NSW pixel subtiles in flight per wave: with ONE accumulator every MFMA waits for the previous one (a chain of up to
  // 15 dependent 16x16x32 MFMAs per subtile); independent accumulators let the matrix pipe and the LDS reads overlap
  constexpr int NSW = 4, WAVES = TL_THREADS / 64;
  for (int s0 = wave; s0 < NS; s0 += NSW * WAVES) {
    bool valid[NSW];
    int pcv[NSW], oyv[NSW], oxv[NSW];
    f32x4 acc[NSW];
#pragma unroll
    for (int u = 0; u < NSW; ++u) {
      const int sub = s0 + u * WAVES;
      const int pi = sub * 16 + lp;
      valid[u] = sub < NS && pi < NPO;
      pcv[u] = valid[u] ? pi : NPO - 1;
      oyv[u] = pcv[u] / RWO; oxv[u] = pcv[u] - oyv[u] * RWO;
      acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // base: top-left tap of this output pixel in frame coordinates
#pragma unroll
    for (int u = 0; u < NSW; ++u) {
      const char* bb = smem + LD::ACTBASE + (((oyv[u] + KOUT - PADK) * TL_FW) + (oxv[u] + KOUT - PADK)) * 4 * E;
      if constexpr (E == 2) {
        unsigned short h[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) h[i] = *reinterpret_cast<const unsigned short*>(bb + boff[i]);
        const uint4 xf = make_uint4(h[0] | ((uint32_t)h[1] << 16), h[2] | ((uint32_t)h[3] << 16), h[4] | ((uint32_t)h[5] << 16),
                                    h[6] | ((uint32_t)h[7] << 16));
        acc[u] = tl_mma<T>(wreg[0], xf, acc[u]);
      } else {
        // f32: two steps of 16 k; lane group lq covers k = 16*m + 4*lq .. +3
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          uint32_t h[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int k = 16 * m + 4 * lq + i;
            const int tap = k / 3, c = k - tap * 3;
            const int dy = tap / KS, dx = tap - dy * KS;
            const int off = (tap < KS * KS) ? ((dy * TL_FW + dx) * 4 + c) * E : 0;
            h[i] = *reinterpret_cast<const uint32_t*>(bb + off);
          }
          acc[u] = tl_mma<T>(*reinterpret_cast<const uint4*>(wl + m * 16 * TL_WROW), make_uint4(h[0], h[1], h[2], h[3]), acc[u]);
        }
      }
    }
    if constexpr (NG > 0) {
      const char* ab[NSW];
#pragma unroll
      for (int u = 0; u < NSW; ++u) ab[u] = smem + LD::ACT + ((oyv[u] + KOUT - KIN - PADK) * RWI + (oxv[u] + KOUT - KIN - PADK)) * AP;
#pragma unroll
      for (int g = 0; g < GS; ++g) {
        int off;
        if constexpr (E == 2) {
          const int hA = 2 * g, hB = (2 * g + 1 < NH) ? 2 * g + 1 : 2 * g;
          const int tA = hA / NG, jA = hA - tA * NG, tB = hB / NG, jB = hB - tB * NG;
          const int offA = (((tA / KS) * RWI + (tA % KS)) * AP + jA * 16 * E);
          const int offB = (((tB / KS) * RWI + (tB % KS)) * AP + jB * 16 * E);
          off = ((lq >> 1) ? offB : offA) + (lq & 1) * 16;
        } else {
          const int t = g / NG, j = g - t * NG;
          off = (((t / KS) * RWI + (t % KS)) * AP + j * 16 * E) + lq * 16;
        }
        uint4 xf[NSW];
#pragma unroll
        for (int u = 0; u < NSW; ++u) xf[u] = *reinterpret_cast<const uint4*>(ab[u] + off);
        uint4 wg;
        if constexpr (WREG) wg = wreg[BS + g];
        else wg = *reinterpret_cast<const uint4*>(wl + (BS + g) * 16 * TL_WROW);
#pragma unroll
        for (int u = 0; u < NSW; ++u) acc[u] = tl_mma<T>(wg, xf[u], acc[u]);
      }
    }
    // epilogue: lane holds couts 4*lq .. 4*lq+3 of pixel pc
#pragma unroll
    for (int u = 0; u < NSW; ++u) {
      const float v0 = acc[u][0] + bias.x, v1 = acc[u][1] + bias.y, v2 = acc[u][2] + bias.z, v3 = acc[u][3] + bias.w;
      if constexpr (L < 5) {
        if (valid[u]) {
          T* dst = reinterpret_cast<T*>(smem + LD::raw(L)) + pcv[u] * 16 + lq * 4;
          if constexpr (E == 4) *reinterpret_cast<float4*>(dst) = make_float4(v0, v1, v2, v3);
          else *reinterpret_cast<uint2*>(dst) = make_uint2(Half<T>::pack(v0, v1), Half<T>::pack(v2, v3));
        }
      } else {
        const int gy = ty0 + oyv[u], gx = tx0 + oxv[u];
        if (valid[u] && lq == 0 && gy < a.H && gx < a.W) {
          const size_t plane = (size_t)a.H * a.W;
          float* o = a.y + (size_t)img * 3 * plane + (size_t)gy * a.W + gx;
          o[0] = sigmoidf(v0);
          o[plane] = sigmoidf(v1);
          o[2 * plane] = sigmoidf(v2);
        }
      }
    }
  }
  __syncthreads();
}

template <typename T>
__global__ __launch_bounds__(TL_THREADS) void tail_kernel(const TailArgs a) {
  using LD = TailLds<T>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tiles_x = cdiv(a.W, TL_TW), tiles_y = cdiv(a.H, TL_TH);
  int bid = blockIdx.x;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int img = bid / tiles_y;
  const int ty0 = ty * TL_TH, tx0 = tx * TL_TW;
  const int tid = threadIdx.x;

  // ---- phase 0: base = bilinear_x2(lo) + x over the 16x24 frame (fp32 in LDS) ---------------------------------
  const size_t plane = (size_t)a.H * a.W;
  const int Hl = a.H >> 1, Wl = a.W >> 1;
  for (int p = tid; p < TL_FH * TL_FW; p += TL_THREADS) {
    const int fy = p / TL_FW, fx = p - fy * TL_FW;
    const int gy = ty0 - TL_HALO + fy, gx = tx0 - TL_HALO + fx;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
      const float* xp = a.x + (size_t)img * 3 * plane + (size_t)gy * a.W + gx;
      r.x = xp[0]; r.y = xp[plane]; r.z = xp[2 * plane];
      if (a.lo) {
        int y0, y1, x0, x1;
        float hy0, hy1, wx0, wx1;
        tl_src(gy, Hl, y0, y1, hy0, hy1);
        tl_src(gx, Wl, x0, x1, wx0, wx1);
        const T* lb = reinterpret_cast<const T*>(a.lo) + (size_t)img * Hl * Wl * a.lo_stride;
        auto at = [&](int yy, int xx, int c) { return ld(lb + ((size_t)yy * Wl + xx) * a.lo_stride + c); };
        float u[3];
#pragma unroll
        for (int c = 0; c < 3; ++c)
          u[c] = hy0 * (wx0 * at(y0, x0, c) + wx1 * at(y0, x1, c)) + hy1 * (wx0 * at(y1, x0, c) + wx1 * at(y1, x1, c));
        r.x += u[0]; r.y += u[1]; r.z += u[2];
      }
    }
    *reinterpret_cast<float4*>(smem + LD::RAWBASE + p * 16) = r;
  }
  __syncthreads();
  tail_layer<T, 1>(a, smem, img, ty0, tx0);
  tail_layer<T, 2>(a, smem, img, ty0, tx0);
  tail_layer<T, 3>(a, smem, img, ty0, tx0);
  tail_layer<T, 4>(a, smem, img, ty0, tx0);
  tail_layer<T, 5>(a, smem, img, ty0, tx0);
}

// ---- host: layout + packing -------------------------------------------------------------------------------------------
static size_t tl_align(size_t v) { return (v + 255) & ~(size_t)255; }

struct TailBlobLayout {
  size_t pre_scale[5], pre_shift[5], bias[5], w[5], total;
};
static TailBlobLayout tail_layout(int dtype) {
  const int esz = (int)dtype_size(dtype);
  TailBlobLayout L{};
  size_t off = 0;
  for (int l = 1; l <= 5; ++l) {
    L.pre_scale[l - 1] = off; off += tl_align(tl_cin(l) * sizeof(float));
    L.pre_shift[l - 1] = off; off += tl_align(tl_cin(l) * sizeof(float));
    L.bias[l - 1] = off; off += tl_align(16 * sizeof(float));
    L.w[l - 1] = off; off += tl_align((size_t)tl_steps(l, esz) * 1024);
  }
  L.total = off;
  return L;
}

// w: [cout][cin][ks][ks] fp32 (nn.Conv2d), cin = 3 + 16*ng
static void tail_pack_layer(int dtype, int l, const float* w, int cout, char* dst) {
  const int esz = (int)dtype_size(dtype);
  const int ng = tl_ng(l), ks = tl_ks(l), cin = tl_cin(l), nh = tl_nh(l);
  const int bs = tl_base_steps(esz);
  memset(dst, 0, (size_t)tl_steps(l, esz) * 1024);
  auto put = [&](int step, int o, int k, float v) {  // element k of row o in a 1 KiB step
    if (esz == 4) reinterpret_cast<float*>(dst + (size_t)step * 1024)[o * 16 + k] = v;
    else reinterpret_cast<uint16_t*>(dst + (size_t)step * 1024)[o * 32 + k] = f32_to_half_bits(dtype, v);
  };
  const int kper = esz == 4 ? 16 : 32;
  for (int o = 0; o < cout; ++o) {
    for (int k = 0; k < 32; ++k) {  // base im2col row: k = tap*3 + c
      const int tap = k / 3, c = k % 3;
      if (tap >= ks * ks) continue;
      put(k / kper, o, k % kper, w[((size_t)o * cin + c) * ks * ks + tap]);
    }
    for (int h = 0; h < nh; ++h) {  // growth half-steps, tap-major
      const int tap = h / ng, j = h % ng;
      for (int i = 0; i < 16; ++i) {
        const float v = w[((size_t)o * cin + 3 + j * 16 + i) * ks * ks + tap];
        if (esz == 4) put(bs + h, o, i, v);
        else put(bs + h / 2, o, (h & 1) * 16 + i, v);
      }
    }
  }
}

}  // namespace mdie

using namespace mdie;

extern "C" size_t mdie_tail_param_bytes(int dtype) {
  if (!dtype_valid(dtype)) return 0;
  return tail_layout(dtype).total;
}

extern "C" int mdie_tail_pack_params(int dtype, const mdie_tensor* tensors, int n, const char* prefix, void* dst, size_t dst_bytes) {
  MDIE_REQUIRE(dtype_valid(dtype), "mdie_tail_pack_params: bad dtype %d", dtype);
  MDIE_REQUIRE(tensors && n > 0 && dst && prefix, "mdie_tail_pack_params: null argument");
  const TailBlobLayout L = tail_layout(dtype);
  if (dst_bytes < L.total) { set_error("mdie_tail_pack_params: %zu < %zu bytes", dst_bytes, L.total); return MDIE_ENOSPC; }
  char* out = reinterpret_cast<char*>(dst);
  memset(out, 0, L.total);
  auto find = [&](const std::string& key, int64_t numel) -> const float* {
    for (int i = 0; i < n; ++i)
      if (tensors[i].name && key == tensors[i].name) {
        if (tensors[i].numel != numel || !tensors[i].data) {
          set_error("mdie_tail_pack_params: '%s' has %lld elements, expected %lld", key.c_str(), (long long)tensors[i].numel, (long long)numel);
          return nullptr;
        }
        return tensors[i].data;
      }
    set_error("mdie_tail_pack_params: checkpoint entry '%s' missing", key.c_str());
    return nullptr;
  };
  const std::string pre = strlen(prefix) ? std::string(prefix) + "." : std::string();
  for (int l = 1; l <= 5; ++l) {
    const std::string p = pre + (l == 5 ? std::string("transition_layer") : "layers." + std::to_string(l - 1));
    const int cin = tl_cin(l), cout = l == 5 ? 3 : 16, ks = tl_ks(l);
    const float *g = find(p + ".0.weight", cin), *b = find(p + ".0.bias", cin), *m = find(p + ".0.running_mean", cin),
                *v = find(p + ".0.running_var", cin), *w = find(p + ".2.weight", (int64_t)cout * cin * ks * ks), *cb = find(p + ".2.bias", cout);
    if (!g || !b || !m || !v || !w || !cb) return MDIE_ENOENT;
    float* ps = reinterpret_cast<float*>(out + L.pre_scale[l - 1]);
    float* pt = reinterpret_cast<float*>(out + L.pre_shift[l - 1]);
    for (int c = 0; c < cin; ++c) {
      const double inv = 1.0 / sqrt((double)v[c] + 1e-5);
      ps[c] = (float)(g[c] * inv);
      pt[c] = (float)(b[c] - m[c] * g[c] * inv);
    }
    memcpy(out + L.bias[l - 1], cb, cout * sizeof(float));
    tail_pack_layer(dtype, l, w, cout, out + L.w[l - 1]);
  }
  return MDIE_OK;
}

extern "C" int mdie_tail_fwd(const mdie_tail_desc* d, void* stream) {
  MDIE_REQUIRE(d != nullptr, "mdie_tail_fwd: null descriptor");
  MDIE_REQUIRE(dtype_valid(d->dtype), "mdie_tail_fwd: bad dtype %d", d->dtype);
  MDIE_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0, "mdie_tail_fwd: empty extent");
  MDIE_REQUIRE(d->x && d->y && d->params, "mdie_tail_fwd: null pointer");
  MDIE_REQUIRE(!d->lo || (d->H % 2 == 0 && d->W % 2 == 0 && d->lo_stride >= 3), "mdie_tail_fwd: lo needs even H, W and >= 3 channels");
  MDIE_REQUIRE(((uintptr_t)d->params & 255) == 0, "mdie_tail_fwd: params must be 256-byte aligned");
  const TailBlobLayout L = tail_layout(d->dtype);
  TailArgs a{};
  a.B = d->B; a.H = d->H; a.W = d->W;
  a.lo = reinterpret_cast<const char*>(d->lo); a.lo_stride = d->lo_stride;
  a.x = d->x; a.y = d->y;
  const char* base = reinterpret_cast<const char*>(d->params);
  for (int l = 0; l < 5; ++l) {
    a.p.pre_scale[l] = reinterpret_cast<const float*>(base + L.pre_scale[l]);
    a.p.pre_shift[l] = reinterpret_cast<const float*>(base + L.pre_shift[l]);
    a.p.bias[l] = reinterpret_cast<const float*>(base + L.bias[l]);
    a.p.w[l] = base + L.w[l];
  }
  const int grid = cdiv(d->W, TL_TW) * cdiv(d->H, TL_TH) * d->B;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  TimedLaunch tl(MDIE_K_TAIL);
  MDIE_SWITCH_T(d->dtype,
    static LdsOptIn opt;
    if (!opt.ensure(reinterpret_cast<const void*>(&tail_kernel<T>), TailLds<T>::TOTAL)) return MDIE_ELAUNCH;
    hipLaunchKernelGGL((tail_kernel<T>), dim3(grid), dim3(TL_THREADS), TailLds<T>::TOTAL, s, a));
  MDIE_LAUNCH_CHECK("mdie_tail_fwd");
  return MDIE_OK;
}
