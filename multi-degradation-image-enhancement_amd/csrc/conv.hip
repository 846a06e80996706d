// Fused convolution for the CDAN path (gfx950):
//   out = pool2x2?( act( conv_k(pre(in)) * post_scale + post_shift ) + residual? )
// Replaces ConvBlock (models/cdan.py:8-19), the dense-layer / transition recipes
// (models/cdan.py:41-53), the decoder ConvTranspose2d+BN+ReLU stages (models/cdan.py:127-152)
// and the nn.MaxPool2d(2,2) that follows the encoder blocks (models/cdan.py:75,82,89).
//
// Formulation: implicit GEMM, D[cout][pixel] = sum_{tap, cin} W[tap][cout][cin] * X[pixel+tap][cin],
// on the 16x16 MFMA (bf16: v_mfma_f32_16x16x32_bf16, f32: 4 x v_mfma_f32_16x16x4_f32 per 16 bytes
// of channels).  A workgroup (4 waves) owns a TILE x TILE pixel tile x BN output channels and walks
// the input channels in 64-byte chunks (32 bf16 / 16 f32 channels):
//   - the (TILE+2)^2 input patch of the chunk is staged in LDS once (pre-activation BN+ReLU and zero
//     padding applied on the way in) and re-read for all 9 taps, so HBM/L2 sees each input element
//     ~1.27x (halo) instead of 9x; the chunk's weights [tap][BN] are staged next to it;
//   - the next chunk's global loads are issued before the MFMAs of the current one (registers),
//     so HBM/L2 latency hides under the matrix work.
// LDS images are PLANAR by 16-byte K group (the `lane>>4` of an MFMA operand): plane q holds
// 16 bytes per pixel / per weight row.  With a 24-pixel row pitch (== 8 mod 16) the 8+8 pixels a
// ds_read_b128 lane group touches in its two planes land on 16 distinct 16-byte bank slots for every
// tap: operand reads are conflict-free (the 80-byte padded rows this replaces were 2-3 way).
// MFMA rows are output channels and columns are pixels, so a lane ends up with 4 consecutive
// channels of one pixel (8/16-byte NHWC stores), and the 4 pixels of a 2x2 pooling window sit in
// 4 adjacent lanes (max-pool = two lane swaps in the epilogue).
// The input is a list of channel segments (mdie_seg): a DenseBlock's torch.cat is never built.
#include "common.hpp"

namespace mdie {

constexpr int CONV_THREADS = 256;
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
  EpiArgs e;
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
template <int TILE>
__device__ __forceinline__ void tile_pixel(int ps, int p, int& y, int& x) {
  constexpr int BPR = TILE / 2;
  const int blk = ps * 4 + (p >> 2);
  y = 2 * (blk / BPR) + ((p >> 1) & 1);
  x = 2 * (blk % BPR) + (p & 1);
}

// ---- epilogue: affine, activation, residual, 2x2 max-pool, NHWC store -----------------------------------------
template <typename T, int NCS, int NPS, int TILE>
__device__ __forceinline__ void conv_epilogue(const EpiArgs& e, f32x4 (&acc)[NCS][NPS], int img, int y0, int x0, int n0,
                                              int wave, int lq, int lp) {
  const int Ho = e.pool ? e.H >> 1 : e.H, Wo = e.pool ? e.W >> 1 : e.W;
#pragma unroll
  for (int cs = 0; cs < NCS; ++cs) {
    const int c = n0 + cs * 16 + lq * 4;
    const float4 sc = *reinterpret_cast<const float4*>(e.post_scale + c);
    const float4 sh = *reinterpret_cast<const float4*>(e.post_shift + c);
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps) {
      int y, x;
      tile_pixel<TILE>(wave * NPS + ps, lp, y, x);
      const int gy = y0 + y, gx = x0 + x;
      const bool inside = gy < e.H && gx < e.W;
      float v[4];
      v[0] = apply_act(fmaf(acc[cs][ps][0], sc.x, sh.x), e.act);
      v[1] = apply_act(fmaf(acc[cs][ps][1], sc.y, sh.y), e.act);
      v[2] = apply_act(fmaf(acc[cs][ps][2], sc.z, sh.z), e.act);
      v[3] = apply_act(fmaf(acc[cs][ps][3], sc.w, sh.w), e.act);
      int oy = gy, ox = gx;
      bool writer = inside;
      if (e.pool) {
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
        if (e.residual) {
          const T* r = reinterpret_cast<const T*>(e.residual) + opix * e.res_stride + c;
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] += ld(r + i);
        }
        T* o = reinterpret_cast<T*>(e.out) + opix * e.out_stride + c;
        if constexpr (sizeof(T) == 4) {
          *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          *reinterpret_cast<uint2*>(o) = make_uint2(bf_pack(v[0], v[1]), bf_pack(v[2], v[3]));
        }
      }
    }
  }
}

template <int KS, int BN, int TILE> struct ConvGeom {
  static constexpr int PAD = KS / 2;
  static constexpr int PW = TILE + 2 * PAD;                 // logical patch edge
  static constexpr int NTAP = KS * KS;
  static constexpr int PLANE = ((PW * PWP * 16 + 127) / 256) * 256 + 128;  // == 128 (mod 256): 2-way staging writes at worst
  static constexpr int WPLANE = NTAP * BN * 16;             // multiple of 256 for BN in {16, 64}
  static constexpr int LDS_BYTES = 4 * PLANE + 4 * WPLANE;
};

template <typename T, int KS, int BN, int TILE>
__global__ __launch_bounds__(CONV_THREADS) void conv_kernel(const ConvArgs a) {
  using G = ConvGeom<KS, BN, TILE>;
  constexpr int VEC = Traits<T>::VEC;
  constexpr int KC = Traits<T>::KC;
  constexpr int PAD = G::PAD, PW = G::PW, NTAP = G::NTAP;
  constexpr int NCS = BN / 16;                       // cout subtiles per wave
  constexpr int NPS = TILE * TILE / 64;              // pixel subtiles per wave
  constexpr int PATCH_UNITS = PW * PW * 4;           // 16-byte units
  constexpr int W_UNITS = 4 * NTAP * BN;
  constexpr int PATCH_IT = (PATCH_UNITS + CONV_THREADS - 1) / CONV_THREADS;
  constexpr int W_IT = (W_UNITS + CONV_THREADS - 1) / CONV_THREADS;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_patch = smem;
  char* lds_w = smem + 4 * G::PLANE;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int lq = lane >> 4;   // 16-byte K group = LDS plane
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
    tile_pixel<TILE>(wave * NPS + ps, lp, y, x);
    xoff[ps] = lq * G::PLANE + (y * PWP + x) * 16;
  }
  const int woff = lq * G::WPLANE + lp * 16;

  const int q = tid & 3;  // this thread's 16-byte column while staging the patch (CONV_THREADS % 4 == 0)
  const bool has_pre = a.pre_scale != nullptr;

  // staging registers (chunk in flight)
  uint4 pv[PATCH_IT];
  bool pin[PATCH_IT];
  uint4 wv[W_IT];
  float ps_[VEC], pb_[VEC];

  auto load_chunk = [&](int chunk) {
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
    if (has_pre && sbase) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) { ps_[i] = a.pre_scale[c0 + i]; pb_[i] = a.pre_shift[c0 + i]; }
    }
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
    // weights of this chunk: global [chunk][q][tap][cout] x 16 B  ->  LDS [q][tap][BN] x 16 B (linear copy per (q, tap))
    const char* wsrc = a.weight + (size_t)chunk * 4 * NTAP * a.cout * 16;
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
      const int u = tid + it * CONV_THREADS;
      const int qt = u / BN, n = u - qt * BN;  // qt = q * NTAP + tap
      wv[it] = make_uint4(0, 0, 0, 0);
      if (u < W_UNITS) wv[it] = *reinterpret_cast<const uint4*>(wsrc + ((size_t)qt * a.cout + n0 + n) * 16);
    }
  };

  auto store_chunk = [&]() {
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
        const int pix = u >> 2;
        const int py = pix / PW, px = pix - py * PW;
        *reinterpret_cast<uint4*>(lds_patch + q * G::PLANE + (py * PWP + px) * 16) = v;
      }
    }
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
      const int u = tid + it * CONV_THREADS;
      if (u < W_UNITS) *reinterpret_cast<uint4*>(lds_w + u * 16) = wv[it];
    }
  };

  load_chunk(0);
  for (int chunk = 0; chunk < a.nchunk; ++chunk) {
    if (chunk > 0) __syncthreads();  // previous chunk's LDS reads are done
    store_chunk();
    __syncthreads();
    if (chunk + 1 < a.nchunk) load_chunk(chunk + 1);  // in flight during the MFMAs below

#pragma unroll
    for (int tap = 0; tap < NTAP; ++tap) {
      const int kh = tap / KS, kw = tap - kh * KS;
      uint4 wf[NCS], xf[NPS];
#pragma unroll
      for (int cs = 0; cs < NCS; ++cs)
        wf[cs] = *reinterpret_cast<const uint4*>(lds_w + (tap * BN + cs * 16) * 16 + woff);
#pragma unroll
      for (int ps = 0; ps < NPS; ++ps)
        xf[ps] = *reinterpret_cast<const uint4*>(lds_patch + (kh * PWP + kw) * 16 + xoff[ps]);
#pragma unroll
      for (int cs = 0; cs < NCS; ++cs)
#pragma unroll
        for (int ps = 0; ps < NPS; ++ps) acc[cs][ps] = mma16<T>(wf[cs], xf[ps], acc[cs][ps]);
    }
  }

  conv_epilogue<T, NCS, NPS, TILE>(a.e, acc, img, y0, x0, n0, wave, lq, lp);
}

// ---------------------------------------------------------------------------------------------------------------
// First layer: conv3x3(3 -> cout) straight from the fp32 NCHW network input (encoder.conv1,
// models/cdan.py:58,74).  K = 27 is im2col'ed to one 32-deep bf16 MFMA step (two 16-deep f32 steps):
// 9x less matrix work than padding 3 channels to a 32-channel chunk, and the NCHW->NHWC layout pass
// disappears.  The input patch lives in LDS as [pixel][4] (channel 3 = 0); each lane gathers its 8
// (tap, channel) operands; weight fragments come straight from global (4 KiB in total) into registers.
// ---------------------------------------------------------------------------------------------------------------
struct FirstArgs {
  int B, H, W;
  int tiles_x, tiles_y, n_tiles;
  int cout;
  const float* x;      // NCHW fp32 [B,3,H,W]
  const char* weight;  // [step][cout][64 B], k = tap*3 + c
  EpiArgs e;
};

template <typename T, int BN>
__global__ __launch_bounds__(CONV_THREADS) void conv_first_kernel(const FirstArgs a) {
  constexpr int TILE = 16, PW = TILE + 2;
  constexpr int E = sizeof(T);
  constexpr int NCS = BN / 16, NPS = 4;
  constexpr int STEPS = E == 2 ? 1 : 2;
  __shared__ __attribute__((aligned(16))) T patch[PW * PW * 4];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lq = lane >> 4, lp = lane & 15;
  int bid = blockIdx.x;
  const int nt = bid % a.n_tiles; bid /= a.n_tiles;
  const int tx = bid % a.tiles_x; bid /= a.tiles_x;
  const int ty = bid % a.tiles_y; bid /= a.tiles_y;
  const int img = bid;
  const int y0 = ty * TILE, x0 = tx * TILE, n0 = nt * BN;

  // weight fragments -> registers (issued first; they land while the patch is staged)
  uint4 wf[STEPS][NCS];
#pragma unroll
  for (int s = 0; s < STEPS; ++s)
#pragma unroll
    for (int cs = 0; cs < NCS; ++cs)
      wf[s][cs] = *reinterpret_cast<const uint4*>(a.weight + ((size_t)s * a.cout + n0 + cs * 16 + lp) * 64 + lq * 16);

  const size_t plane = (size_t)a.H * a.W;
  for (int p = tid; p < PW * PW; p += CONV_THREADS) {
    const int py = p / PW, px = p - py * PW;
    const int gy = y0 + py - 1, gx = x0 + px - 1;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f;
    if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
      const float* xp = a.x + (size_t)img * 3 * plane + (size_t)gy * a.W + gx;
      v0 = xp[0]; v1 = xp[plane]; v2 = xp[2 * plane];
    }
    T* d = patch + p * 4;
    st(d + 0, v0); st(d + 1, v1); st(d + 2, v2); st(d + 3, 0.f);
  }
  __syncthreads();

  f32x4 acc[NCS][NPS];
#pragma unroll
  for (int i = 0; i < NCS; ++i)
#pragma unroll
    for (int j = 0; j < NPS; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int ps = 0; ps < NPS; ++ps) {
    int y, x;
    tile_pixel<TILE>(wave * NPS + ps, lp, y, x);
    const T* base = patch + (y * PW + x) * 4;
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      uint4 xf;
      if constexpr (E == 2) {
        uint32_t h[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int k = 8 * lq + i;                 // im2col column: k = tap*3 + c
          const int tap = k / 3, c = k - tap * 3;
          const int off = tap < 9 ? ((tap / 3) * PW + (tap % 3)) * 4 + c : 0;
          h[i] = *reinterpret_cast<const unsigned short*>(base + off);
        }
        xf = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
      } else {
        uint32_t h[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int k = 16 * s + 4 * lq + i;
          const int tap = k / 3, c = k - tap * 3;
          const int off = tap < 9 ? ((tap / 3) * PW + (tap % 3)) * 4 + c : 0;
          h[i] = *reinterpret_cast<const uint32_t*>(base + off);
        }
        xf = make_uint4(h[0], h[1], h[2], h[3]);
      }
#pragma unroll
      for (int cs = 0; cs < NCS; ++cs) acc[cs][ps] = mma16<T>(wf[s][cs], xf, acc[cs][ps]);
    }
  }
  conv_epilogue<T, NCS, NPS, TILE>(a.e, acc, img, y0, x0, n0, wave, lq, lp);
}

// ---- host ---------------------------------------------------------------------------------------------------------
template <typename T, int KS, int BN, int TILE>
static int launch_conv(ConvArgs& a, hipStream_t stream) {
  using G = ConvGeom<KS, BN, TILE>;
  a.tiles_x = cdiv(a.W, TILE); a.tiles_y = cdiv(a.H, TILE);
  const int grid = a.n_tiles * a.tiles_x * a.tiles_y * a.B;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_kernel<T, KS, BN, TILE>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    attr_set = true;
  }
  TimedLaunch tl(KS == 3 ? MDIE_K_CONV3 : MDIE_K_CONV1);
  hipLaunchKernelGGL((conv_kernel<T, KS, BN, TILE>), dim3(grid), dim3(CONV_THREADS), G::LDS_BYTES, stream, a);
  MDIE_LAUNCH_CHECK("mdie_conv_fwd");
  return MDIE_OK;
}

static void fill_epi(EpiArgs& e, int H, int W, const float* sc, const float* sh, int act, int pool, const void* res, int res_stride,
                     void* out, int out_stride) {
  e.H = H; e.W = W; e.post_scale = sc; e.post_shift = sh; e.act = act; e.pool = pool;
  e.residual = reinterpret_cast<const char*>(res); e.res_stride = res_stride;
  e.out = reinterpret_cast<char*>(out); e.out_stride = out_stride;
}

template <typename T>
static int dispatch_conv(const mdie_conv_desc* d, hipStream_t stream) {
  constexpr int KC = Traits<T>::KC;
  ConvArgs a{};
  a.B = d->B; a.H = d->H; a.W = d->W;
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
  fill_epi(a.e, d->H, d->W, d->post_scale, d->post_shift, d->act, d->pool, d->residual, d->res_stride, d->out, d->out_stride);
  const int bn = (d->cout % 64 == 0) ? 64 : 16;
  a.n_tiles = d->cout / bn;
  // Small feature maps (32x32, 64x64 at the network's deep end) do not fill 256 CUs with 16x16 tiles:
  // switch to 8x8 tiles (4x the workgroups) when the 16x16 grid would leave CUs idle.
  const long wgs16 = (long)cdiv(d->H, 16) * cdiv(d->W, 16) * d->B * a.n_tiles;
  const bool small = wgs16 < (bn == 16 ? 1024 : 512);
  if (d->ksize == 3) {
    if (bn == 64) return small ? launch_conv<T, 3, 64, 8>(a, stream) : launch_conv<T, 3, 64, 16>(a, stream);
    return small ? launch_conv<T, 3, 16, 8>(a, stream) : launch_conv<T, 3, 16, 16>(a, stream);
  }
  if (bn == 64) return small ? launch_conv<T, 1, 64, 8>(a, stream) : launch_conv<T, 1, 64, 16>(a, stream);
  return small ? launch_conv<T, 1, 16, 8>(a, stream) : launch_conv<T, 1, 16, 16>(a, stream);
}

template <typename T>
static int dispatch_first(const mdie_conv_first_desc* d, hipStream_t stream) {
  FirstArgs a{};
  a.B = d->B; a.H = d->H; a.W = d->W; a.cout = d->cout;
  a.tiles_x = cdiv(d->W, 16); a.tiles_y = cdiv(d->H, 16);
  a.x = d->x; a.weight = reinterpret_cast<const char*>(d->weight);
  fill_epi(a.e, d->H, d->W, d->post_scale, d->post_shift, d->act, d->pool, nullptr, 0, d->out, d->out_stride);
  const int bn = (d->cout % 64 == 0) ? 64 : 16;
  a.n_tiles = d->cout / bn;
  const int grid = a.n_tiles * a.tiles_x * a.tiles_y * a.B;
  TimedLaunch tl(MDIE_K_CONV3);
  if (bn == 64) hipLaunchKernelGGL((conv_first_kernel<T, 64>), dim3(grid), dim3(CONV_THREADS), 0, stream, a);
  else hipLaunchKernelGGL((conv_first_kernel<T, 16>), dim3(grid), dim3(CONV_THREADS), 0, stream, a);
  MDIE_LAUNCH_CHECK("mdie_conv_first_fwd");
  return MDIE_OK;
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

extern "C" int mdie_conv_first_fwd(const mdie_conv_first_desc* d, void* stream) {
  using namespace mdie;
  MDIE_REQUIRE(d != nullptr, "mdie_conv_first_fwd: null descriptor");
  MDIE_REQUIRE(d->dtype == MDIE_F32 || d->dtype == MDIE_BF16, "mdie_conv_first_fwd: bad dtype %d", d->dtype);
  MDIE_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0, "mdie_conv_first_fwd: empty extent");
  MDIE_REQUIRE(d->cout > 0 && d->cout % 16 == 0, "mdie_conv_first_fwd: cout %d must be a multiple of 16", d->cout);
  MDIE_REQUIRE(d->x && d->weight && d->post_scale && d->post_shift && d->out, "mdie_conv_first_fwd: null pointer");
  MDIE_REQUIRE(!d->pool || (d->H % 2 == 0 && d->W % 2 == 0), "mdie_conv_first_fwd: pool needs even H, W");
  MDIE_REQUIRE(d->out_stride % 4 == 0 && d->out_stride >= d->cout, "mdie_conv_first_fwd: out_stride %d", d->out_stride);
  MDIE_REQUIRE(((uintptr_t)d->out & 15) == 0 && ((uintptr_t)d->weight & 15) == 0, "mdie_conv_first_fwd: out/weight alignment");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  return d->dtype == MDIE_F32 ? dispatch_first<float>(d, s) : dispatch_first<mdie::bf16>(d, s);
}
