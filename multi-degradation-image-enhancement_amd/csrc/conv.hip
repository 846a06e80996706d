// Fused convolution for the CDAN path (gfx950):
//   out = pool2x2?( act( conv_k(pre(in)) * post_scale + post_shift ) + residual? )
// Replaces ConvBlock (models/cdan.py:8-19), the dense-layer / transition recipes
// (models/cdan.py:41-53), the decoder ConvTranspose2d+BN+ReLU stages (models/cdan.py:127-152)
// and the nn.MaxPool2d(2,2) that follows the encoder blocks (models/cdan.py:75,82,89).
//
// Formulation: implicit GEMM, D[cout][pixel] = sum_{tap, cin} W[tap][cout][cin] * X[pixel+tap][cin],
// on the 16x16 MFMA (bf16: v_mfma_f32_16x16x32_bf16, f32: 4 x v_mfma_f32_16x16x4_f32 per 16 bytes
// of channels).  A workgroup (4 waves) owns a 16x16-pixel tile x BN output channels and walks the
// input channels in 64-byte chunks (32 bf16 / 16 f32 channels):
//   - the (16+2)x(16+2) input patch of the chunk is staged in LDS once (pre-activation BN+ReLU and
//     zero padding applied on the way in) and re-read for all 9 taps from LDS, so HBM/L2 sees each
//     input element ~1.27x (halo) instead of 9x;
//   - the chunk's weights [tap][BN][64 B] are staged next to it.
// MFMA rows are output channels and columns are pixels, so a lane ends up with 4 consecutive
// channels of one pixel (8/16-byte NHWC stores), and the 4 pixels of a 2x2 pooling window sit in
// 4 adjacent lanes (max-pool = two lane swaps in the epilogue).
// The input is a list of channel segments (mdie_seg): a DenseBlock's torch.cat is never built.
#include "common.hpp"

namespace mdie {

constexpr int TILE = 16;        // output tile edge (pixels)
constexpr int ROWB = 80;        // LDS row pitch in bytes: 64 B of channels + 16 B pad
constexpr int CONV_THREADS = 256;

struct SegDev {
  const char* ptr;
  int ch_begin, ch_end;  // stored channel range [begin, end)
  int stride;            // elements per pixel
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
  const float* post_scale;
  const float* post_shift;
  int act, pool;
  const char* residual;
  int res_stride;
  char* out;
  int out_stride;
};

template <typename T> __device__ __forceinline__ f32x4 mma16(const uint4& w, const uint4& x, f32x4 acc);
template <> __device__ __forceinline__ f32x4 mma16<bf16>(const uint4& w, const uint4& x, f32x4 acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), acc, 0, 0, 0);
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
__device__ __forceinline__ void tile_pixel(int ps, int p, int& y, int& x) {
  const int blk = ps * 4 + (p >> 2);
  y = 2 * (blk >> 3) + ((p >> 1) & 1);
  x = 2 * (blk & 7) + (p & 1);
}

template <typename T, int KS, int BN>
__global__ __launch_bounds__(CONV_THREADS) void conv_kernel(const ConvArgs a) {
  constexpr int VEC = Traits<T>::VEC;
  constexpr int KC = Traits<T>::KC;
  constexpr int PAD = KS / 2;
  constexpr int PW = TILE + 2 * PAD;
  constexpr int NPIX = PW * PW;
  constexpr int NTAP = KS * KS;
  constexpr int NCS = BN / 16;                       // cout subtiles per wave
  constexpr int NPS = 4;                             // pixel subtiles per wave
  constexpr int PATCH_UNITS = NPIX * 4;              // 16-byte units
  constexpr int W_UNITS = NTAP * BN * 4;
  constexpr int PATCH_IT = (PATCH_UNITS + CONV_THREADS - 1) / CONV_THREADS;
  constexpr int W_IT = (W_UNITS + CONV_THREADS - 1) / CONV_THREADS;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_patch = smem;
  char* lds_w = smem + NPIX * ROWB;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int lq = lane >> 4;   // 16-byte column / k group
  const int lp = lane & 15;

  int bid = blockIdx.x;
  const int nt = bid % a.n_tiles; bid /= a.n_tiles;
  const int tx = bid % a.tiles_x; bid /= a.tiles_x;
  const int ty = bid % a.tiles_y; bid /= a.tiles_y;
  const int img = bid;
  const int y0 = ty * TILE, x0 = tx * TILE, n0 = nt * BN;

  f32x4 acc[NCS][NPS];
#pragma unroll
  for (int i = 0; i < NCS; ++i)
#pragma unroll
    for (int j = 0; j < NPS; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // per-lane LDS read offsets
  int xoff[NPS];
#pragma unroll
  for (int ps = 0; ps < NPS; ++ps) {
    int y, x;
    tile_pixel(wave * NPS + ps, lp, y, x);
    xoff[ps] = (y * PW + x) * ROWB + lq * 16;
  }
  const int woff = lp * ROWB + lq * 16;

  const int q = tid & 3;  // this thread's 16-byte column while staging (CONV_THREADS % 4 == 0)
  const bool has_pre = a.pre_scale != nullptr;

  for (int chunk = 0; chunk < a.nchunk; ++chunk) {
    // ---- stage: global -> registers ---------------------------------------------------------
    const int c0 = chunk * KC + q * VEC;  // first stored channel of this thread's column
    const char* sbase = nullptr;
    int sstride = 0;
#pragma unroll
    for (int s = 0; s < MDIE_MAX_SEG; ++s) {
      if (s < a.nseg && c0 >= a.seg[s].ch_begin && c0 < a.seg[s].ch_end) {
        sbase = a.seg[s].ptr + (size_t)(c0 - a.seg[s].ch_begin) * sizeof(T);
        sstride = a.seg[s].stride;
      }
    }
    float ps_[VEC], pb_[VEC];
    if (has_pre && sbase) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) { ps_[i] = a.pre_scale[c0 + i]; pb_[i] = a.pre_shift[c0 + i]; }
    }

    uint4 pv[PATCH_IT];
    bool pin[PATCH_IT];
#pragma unroll
    for (int it = 0; it < PATCH_IT; ++it) {
      const int u = tid + it * CONV_THREADS;
      const int pix = u >> 2;
      const int py = pix / PW, px = pix - py * PW;
      const int gy = y0 + py - PAD, gx = x0 + px - PAD;
      pin[it] = (u < PATCH_UNITS) && sbase && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      pv[it] = make_uint4(0, 0, 0, 0);
      if (pin[it]) {
        const size_t pixel = ((size_t)img * a.H + gy) * a.W + gx;
        pv[it] = *reinterpret_cast<const uint4*>(sbase + pixel * sstride * sizeof(T));
      }
    }
    uint4 wv[W_IT];
    const char* wsrc = a.weight + ((size_t)chunk * NTAP * a.cout) * 64;
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
      const int u = tid + it * CONV_THREADS;
      const int row = u >> 2;               // tap * BN + n
      const int tap = row / BN, n = row - tap * BN;
      wv[it] = make_uint4(0, 0, 0, 0);
      if (u < W_UNITS) wv[it] = *reinterpret_cast<const uint4*>(wsrc + ((size_t)tap * a.cout + n0 + n) * 64 + q * 16);
    }

    if (chunk > 0) __syncthreads();  // previous chunk's LDS reads are done

    // ---- registers -> LDS (pre-activation on the way) --------------------------------------------
#pragma unroll
    for (int it = 0; it < PATCH_IT; ++it) {
      const int u = tid + it * CONV_THREADS;
      if (u < PATCH_UNITS) {
        uint4 v = pv[it];
        if (has_pre && pin[it]) {
          float f[VEC];
          Vec16<T>::unpack(v, f);
#pragma unroll
          for (int i = 0; i < VEC; ++i) f[i] = fmaxf(fmaf(f[i], ps_[i], pb_[i]), 0.0f);
          v = Vec16<T>::pack(f);
        }
        *reinterpret_cast<uint4*>(lds_patch + (u >> 2) * ROWB + q * 16) = v;
      }
    }
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
      const int u = tid + it * CONV_THREADS;
      if (u < W_UNITS) *reinterpret_cast<uint4*>(lds_w + (u >> 2) * ROWB + q * 16) = wv[it];
    }
    __syncthreads();

    // ---- MFMA over the taps -------------------------------------------------------------------
#pragma unroll
    for (int tap = 0; tap < NTAP; ++tap) {
      const int kh = tap / KS, kw = tap - kh * KS;
      uint4 wf[NCS], xf[NPS];
#pragma unroll
      for (int cs = 0; cs < NCS; ++cs)
        wf[cs] = *reinterpret_cast<const uint4*>(lds_w + (tap * BN + cs * 16) * ROWB + woff);
#pragma unroll
      for (int ps = 0; ps < NPS; ++ps)
        xf[ps] = *reinterpret_cast<const uint4*>(lds_patch + (kh * PW + kw) * ROWB + xoff[ps]);
#pragma unroll
      for (int cs = 0; cs < NCS; ++cs)
#pragma unroll
        for (int ps = 0; ps < NPS; ++ps) acc[cs][ps] = mma16<T>(wf[cs], xf[ps], acc[cs][ps]);
    }
  }

  // ---- epilogue: affine, activation, residual, pool, NHWC store ------------------------------------
  const int Ho = a.pool ? a.H >> 1 : a.H, Wo = a.pool ? a.W >> 1 : a.W;
#pragma unroll
  for (int cs = 0; cs < NCS; ++cs) {
    const int c = n0 + cs * 16 + lq * 4;
    const float4 sc = *reinterpret_cast<const float4*>(a.post_scale + c);
    const float4 sh = *reinterpret_cast<const float4*>(a.post_shift + c);
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps) {
      int y, x;
      tile_pixel(wave * NPS + ps, lp, y, x);
      const int gy = y0 + y, gx = x0 + x;
      const bool inside = gy < a.H && gx < a.W;
      float v[4];
      v[0] = apply_act(fmaf(acc[cs][ps][0], sc.x, sh.x), a.act);
      v[1] = apply_act(fmaf(acc[cs][ps][1], sc.y, sh.y), a.act);
      v[2] = apply_act(fmaf(acc[cs][ps][2], sc.z, sh.z), a.act);
      v[3] = apply_act(fmaf(acc[cs][ps][3], sc.w, sh.w), a.act);
      int oy = gy, ox = gx;
      bool writer = inside;
      if (a.pool) {
        // 2x2 window = lanes lp^1 (x neighbour) and lp^2 (y neighbour)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          v[i] = fmaxf(v[i], __shfl_xor(v[i], 1));
          v[i] = fmaxf(v[i], __shfl_xor(v[i], 2));
        }
        oy = gy >> 1; ox = gx >> 1;
        writer = inside && (lp & 3) == 0;
      }
      if (writer) {
        const size_t opix = ((size_t)img * Ho + oy) * Wo + ox;
        if (a.residual) {
          const T* r = reinterpret_cast<const T*>(a.residual) + opix * a.res_stride + c;
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] += ld(r + i);
        }
        T* o = reinterpret_cast<T*>(a.out) + opix * a.out_stride + c;
        if constexpr (sizeof(T) == 4) {
          *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          *reinterpret_cast<uint2*>(o) = make_uint2(bf_pack(v[0], v[1]), bf_pack(v[2], v[3]));
        }
      }
    }
  }
}

template <typename T, int KS, int BN>
static int launch_conv(const ConvArgs& a, hipStream_t stream) {
  constexpr int PW = TILE + 2 * (KS / 2);
  const size_t lds = (size_t)(PW * PW + KS * KS * BN) * ROWB;
  const int grid = a.n_tiles * a.tiles_x * a.tiles_y * a.B;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_kernel<T, KS, BN>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  TimedLaunch tl(KS == 3 ? MDIE_K_CONV3 : MDIE_K_CONV1);
  hipLaunchKernelGGL((conv_kernel<T, KS, BN>), dim3(grid), dim3(CONV_THREADS), lds, stream, a);
  MDIE_LAUNCH_CHECK("mdie_conv_fwd");
  return MDIE_OK;
}

template <typename T>
static int dispatch_conv(const mdie_conv_desc* d, hipStream_t stream) {
  constexpr int KC = Traits<T>::KC;
  ConvArgs a{};
  a.B = d->B; a.H = d->H; a.W = d->W;
  a.tiles_x = cdiv(d->W, TILE); a.tiles_y = cdiv(d->H, TILE);
  a.cin = d->cin; a.nchunk = cdiv(d->cin, KC); a.cout = d->cout;
  a.nseg = d->nseg;
  int c = 0;
  for (int s = 0; s < d->nseg; ++s) {
    MDIE_REQUIRE(d->in[s].ptr && d->in[s].channels > 0 && d->in[s].channels % 16 == 0,
                 "mdie_conv_fwd: segment %d must have a multiple of 16 channels (got %d)", s, d->in[s].channels);
    MDIE_REQUIRE(d->in[s].stride % 16 == 0 && d->in[s].stride >= d->in[s].channels,
                 "mdie_conv_fwd: segment %d stride %d invalid", s, d->in[s].stride);
    MDIE_REQUIRE(((uintptr_t)d->in[s].ptr & 15) == 0, "mdie_conv_fwd: segment %d not 16-byte aligned", s);
    a.seg[s].ptr = reinterpret_cast<const char*>(d->in[s].ptr);
    a.seg[s].ch_begin = c; c += d->in[s].channels; a.seg[s].ch_end = c;
    a.seg[s].stride = d->in[s].stride;
  }
  MDIE_REQUIRE(c == d->cin, "mdie_conv_fwd: segments hold %d channels, cin = %d", c, d->cin);
  a.pre_scale = d->pre_scale; a.pre_shift = d->pre_shift;
  a.weight = reinterpret_cast<const char*>(d->weight);
  a.post_scale = d->post_scale; a.post_shift = d->post_shift;
  a.act = d->act; a.pool = d->pool;
  a.residual = reinterpret_cast<const char*>(d->residual); a.res_stride = d->res_stride;
  a.out = reinterpret_cast<char*>(d->out); a.out_stride = d->out_stride;
  const int bn = (d->cout % 64 == 0) ? 64 : 16;
  a.n_tiles = d->cout / bn;
  if (d->ksize == 3) {
    return bn == 64 ? launch_conv<T, 3, 64>(a, stream) : launch_conv<T, 3, 16>(a, stream);
  }
  return bn == 64 ? launch_conv<T, 1, 64>(a, stream) : launch_conv<T, 1, 16>(a, stream);
}

}  // namespace mdie

extern "C" int mdie_conv_fwd(const mdie_conv_desc* d, void* stream) {
  using namespace mdie;
  MDIE_REQUIRE(d != nullptr, "mdie_conv_fwd: null descriptor");
  MDIE_REQUIRE(d->dtype == MDIE_F32 || d->dtype == MDIE_BF16, "mdie_conv_fwd: bad dtype %d", d->dtype);
  MDIE_REQUIRE(d->ksize == 3 || d->ksize == 1, "mdie_conv_fwd: ksize %d (3 or 1)", d->ksize);
  MDIE_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0, "mdie_conv_fwd: empty extent %dx%dx%d", d->B, d->H, d->W);
  MDIE_REQUIRE(d->nseg >= 1 && d->nseg <= MDIE_MAX_SEG, "mdie_conv_fwd: nseg %d", d->nseg);
  MDIE_REQUIRE(d->cout > 0 && d->cout % 16 == 0, "mdie_conv_fwd: cout %d must be a multiple of 16", d->cout);
  MDIE_REQUIRE(d->weight && d->post_scale && d->post_shift && d->out, "mdie_conv_fwd: null pointer");
  MDIE_REQUIRE((d->pre_scale == nullptr) == (d->pre_shift == nullptr), "mdie_conv_fwd: pre_scale/pre_shift mismatch");
  MDIE_REQUIRE(!d->pool || (d->H % 2 == 0 && d->W % 2 == 0), "mdie_conv_fwd: pool needs even H, W");
  MDIE_REQUIRE(d->out_stride % 4 == 0 && d->out_stride >= 4, "mdie_conv_fwd: out_stride %d", d->out_stride);
  MDIE_REQUIRE(((uintptr_t)d->out & 15) == 0 && ((uintptr_t)d->weight & 15) == 0, "mdie_conv_fwd: out/weight alignment");
  MDIE_REQUIRE(!d->residual || (d->res_stride % 4 == 0), "mdie_conv_fwd: res_stride %d", d->res_stride);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  return d->dtype == MDIE_F32 ? dispatch_conv<float>(d, s) : dispatch_conv<mdie::bf16>(d, s);
}
