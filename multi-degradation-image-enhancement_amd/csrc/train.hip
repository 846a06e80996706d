// Training-side kernels: the convolution backward pieces that carry ~2/3 of a
// training step's FLOPs.
//   dgrad  needs no kernel of its own: the input gradient of a 3x3 / pad-1 convolution is the same
//          convolution of dY with the flipped, in/out-transposed weights, i.e. mdie_conv_fwd on weights
//          packed with the opposite `transposed` flag (mdie_pack_conv_weight_dev below repacks on the GPU
//          every step, since the weights change every step).
//   wgrad  dW[tap][c][o] = sum over pixels X[p + tap][c] * dY[p][o]: a GEMM whose K dimension is the pixel index.
//          A workgroup owns a (channel tile x output tile) of all 9 taps and walks a run of 16x16-pixel tiles: the X
//          patch (with halo) and the dY tile are staged once per tile in their natural NHWC layout (planes of 16
//          channels, one row per pixel) and serve every tap at an immediate byte offset.  bf16 reads its operands with
//          the transposing ds_read_b64_tr_b16 into v_mfma_f32_16x16x32_bf16; fp32 reads one float per lane into the
//          exact-f32 v_mfma_f32_16x16x4_f32.  Accumulators stay in registers over the whole run; each pixel split
//          writes a scratch slab and a second kernel folds the splits in a fixed order (no float atomics).
#include <stdlib.h>

#include "common.hpp"

namespace mdie {

constexpr int TR_THREADS = 256;

struct SegT {
  const char* ptr;
  int ch_begin, ch_end, stride;
};

// ---- weight repack on the device --------------------------------------------------------------------------------
// same layout as pack_conv_weight() in engine.hip: [chunk][q][tap][cout_st][16 bytes]
template <typename T>
__device__ __forceinline__ void pack_job_run(const mdie_pack_job& j) {
  constexpr int VEC = Traits<T>::VEC, KC = Traits<T>::KC;
  const float* const w = j.w;
  T* const dst = reinterpret_cast<T*>(j.dst);
  const int ks = j.ksize, ntap = ks * ks;
  const size_t total = (size_t)cdiv(j.cin_stored, KC) * 4 * ntap * j.cout_stored * VEC;
  for (size_t u = (size_t)blockIdx.x * TR_THREADS + threadIdx.x; u < total; u += (size_t)gridDim.x * TR_THREADS) {
    size_t r = u;
    const int i = (int)(r % VEC); r /= VEC;
    const int os = (int)(r % j.cout_stored); r /= j.cout_stored;
    const int tap = (int)(r % ntap); r /= ntap;
    const int q = (int)(r % 4);
    const int chunk = (int)(r / 4);
    const int cs = chunk * KC + q * VEC + i;             // stored input channel
    int c = -1, o = -1;                                  // real input / output channel, -1 = padding
    if (cs < j.split) c = cs;
    else if (cs >= j.split + j.gap) c = cs - j.gap;
    if (os < j.out_split) o = os;
    else if (os >= j.out_split + j.out_gap) o = os - j.out_gap;
    float v = 0.f;
    if (o >= 0 && o < j.cout && c >= 0 && c < j.cin) {
      const int kh = tap / ks, kw = tap - kh * ks;
      v = j.transposed ? w[(((size_t)c * j.cout + o) * ks + (ks - 1 - kh)) * ks + (ks - 1 - kw)]
                       : w[(((size_t)o * j.cin + c) * ks + kh) * ks + kw];
    }
    st(dst + u, v);
  }
}

template <typename T>
__global__ __launch_bounds__(TR_THREADS) void pack_weight_kernel(const mdie_pack_job j) { pack_job_run<T>(j); }

// every weight repack of a training step in ONE launch (blockIdx.y = job): a step repacks each of the 35 convolutions twice
// (forward form, flipped / transposed input-gradient form) from the fp32 parameters the optimizer has just updated -- 55 launches
// of 4-5 us each on a GPU-bound step
template <typename T>
__global__ __launch_bounds__(TR_THREADS) void pack_weight_batch_kernel(const mdie_pack_job* jobs) { pack_job_run<T>(jobs[blockIdx.y]); }

// ---- wgrad ------------------------------------------------------------------------------------------------------------
// fold the splits and scatter into PyTorch's layout (dropping padded channels):
//   transposed = 0: dw[o][c][kh][kw]             (nn.Conv2d)
//   transposed = 1: dw[c][o][ks-1-kh][ks-1-kw]   (nn.ConvTranspose2d run as a flipped convolution)
__global__ __launch_bounds__(TR_THREADS) void wgrad_reduce_kernel(int splits, int ks, int transposed, int cout, int cin, int cout_st, int cin_st,
                                                                  int split_c, int gap, const float* scratch, float* dw) {
  // block = 64 consecutive slab elements (output channel fastest: coalesced) x 4 interleaved groups of splits;
  // every thread keeps 4 independent partial sums in flight, the groups are folded through LDS in a fixed order;
  // only the single write per element is scattered into PyTorch's layout
  __shared__ float red[4][64];
  const int ntap = ks * ks;
  const size_t slab = (size_t)ntap * cin_st * cout_st;
  const int e = threadIdx.x & 63, kg = threadIdx.x >> 6;
  const size_t u = (size_t)blockIdx.x * 64 + e;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (u < slab) {
    int k = kg;
    for (; k + 12 < splits; k += 16) {
      s0 += scratch[(size_t)k * slab + u]; s1 += scratch[(size_t)(k + 4) * slab + u];
      s2 += scratch[(size_t)(k + 8) * slab + u]; s3 += scratch[(size_t)(k + 12) * slab + u];
    }
    for (; k < splits; k += 4) s0 += scratch[(size_t)k * slab + u];
  }
  red[kg][e] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (kg != 0 || u >= slab) return;
  const float s = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
  size_t r = u;
  const int o = (int)(r % cout_st); r /= cout_st;
  const int cs = (int)(r % cin_st);
  const int stap = (int)(r / cin_st);                  // tap of the convolution that was run
  int c = -1;                                          // real input channel of stored channel cs
  if (cs < split_c) c = cs;
  else if (cs >= split_c + gap) c = cs - gap;
  if (o >= cout || c < 0 || c >= cin) return;
  const int skh = stap / ks, skw = stap - skh * ks;
  if (transposed) dw[(((size_t)c * cout + o) * ks + (ks - 1 - skh)) * ks + (ks - 1 - skw)] = s;
  else dw[(((size_t)o * cin + c) * ks + skh) * ks + skw] = s;
}

static int tr_grid(size_t total) {
  size_t g = (total + TR_THREADS - 1) / TR_THREADS;
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

// ---- wgrad, bf16: pixel-tiled, all taps from one LDS image, v_mfma_f32_16x16x32_bf16 ---------------------------------
// The K dimension of the weight-gradient GEMM is the pixel index, but NHWC keeps the CHANNEL contiguous, so a
// 16x16x32 operand (8 consecutive k per lane) cannot be loaded directly.  gfx950's transposing LDS read
// ds_read_b64_tr_b16 delivers exactly that: the tile is staged in its natural layout -- planes of 16 channels,
// one 32-byte row per pixel -- and each 16-lane group reads 4 pixel rows x 16 channels column-major.
//   * one workgroup = (c-tile x o-tile) x a run of 16x16-pixel tiles; the X patch (18x18 with halo) and the dY
//     tile are staged once and serve all 9 taps: a tap is a constant byte offset into the patch image, so every
//     fragment read is `base + immediate`;
//   * k <-> pixel map of one 32-pixel step (rows 2k, 2k+1 of the tile): lane group g, read h, element q holds
//     pixel (2k + h, 4g + q); a 32-lane half then reads 8 consecutive 32-byte rows = all 64 banks once;
//   * accumulators (taps x c-subtiles x o-subtiles of 16x16) stay in registers over the whole run and are
//     written once to the split's scratch slab; wgrad_reduce_kernel folds the splits in a fixed order.
typedef short v4s __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));

struct WgradTileArgs {
  int B, H, W;
  int nseg;
  SegT seg[MDIE_MAX_SEG];
  int cin_st, cout_st;
  const char* dy; int dy_stride;
  float* scratch;
  int tiles_x, tiles_y, total_tiles, tiles_per_split;
  int o_tiles;
  const float *pre_scale, *pre_shift;
};

constexpr int WT = 16;  // tile edge
#ifndef WG_KUNROLL
#define WG_KUNROLL 2
#endif

template <int NTAP> struct WgGeom {
  static constexpr int PAD = NTAP == 9 ? 1 : 0;
  static constexpr int PWD = WT + 2 * PAD;
  static constexpr int PROWS = PWD * PWD;
  static constexpr int XPLANE = PROWS * 32 + 32;   // 32-byte rows; +32 staggers the planes' banks for the staging writes
  static constexpr int YPLANE = WT * WT * 32;
};

__device__ __forceinline__ v4s lds_tr16(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(p));
}

template <typename T> __device__ __forceinline__ f32x4 mma_tr(const v8s& a, const v8s& b, f32x4 c);
template <> __device__ __forceinline__ f32x4 mma_tr<bf16>(const v8s& a, const v8s& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4 mma_tr<f16>(const v8s& a, const v8s& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// waves are arranged WC x WO; each owns CSW x OSW 16x16 subtiles of every tap; T = bf16 or f16 (same layouts, other MFMA)
// PF (the thin configurations, one subtile pair per wave: 36 accumulator registers): the NEXT tile's X patch and dY tile are requested
// into registers right after the barrier and land under this tile's MFMAs; without it a tile is "load, wait, stage, barrier, MFMA" in
// sequence -- two exposed memory round trips per tile for the 64-channel patch (11 staging iterations in batches of 6), hidden only by
// the CU's second workgroup.  The sums are formed in the same order either way.
template <typename T, int NTAP, int CSW, int OSW, int WC, int WO, bool PRE, bool PF = false>
__global__ __launch_bounds__(TR_THREADS, 2) void wgrad_tile_kernel(const WgradTileArgs a) {
  using G = WgGeom<NTAP>;
  constexpr int KS = NTAP == 9 ? 3 : 1;
  constexpr int NCS_T = CSW * WC, NOS_T = OSW * WO;
  constexpr int UPX = NCS_T * 2, UPY = NOS_T * 2;             // 16-byte units per pixel
  constexpr int X_UNITS = G::PROWS * UPX, Y_UNITS = WT * WT * UPY;
  constexpr int X_IT = (X_UNITS + TR_THREADS - 1) / TR_THREADS, Y_IT = (Y_UNITS + TR_THREADS - 1) / TR_THREADS;
  constexpr int X_BATCH = X_IT > 6 ? 6 : X_IT;
  constexpr int X_PPI = TR_THREADS / UPX, Y_PPI = TR_THREADS / UPY;   // pixels advanced per staging iteration
  static_assert(WC * WO == TR_THREADS / 64, "wave grid");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_x = smem;
  char* lds_y = smem + NCS_T * G::XPLANE;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave % WC, wo = wave / WC;
  const int ot = blockIdx.x % a.o_tiles, ct = blockIdx.x / a.o_tiles;
  const int c0 = ct * NCS_T * 16, o0 = ot * NOS_T * 16;
  const int split = blockIdx.y;

  // ---- staging geometry (tile independent) ----
  const int xchunk = tid % UPX;                      // this thread's 16-byte channel column of the c-tile
  const char* xbase = nullptr; int xstride = 0;
  {
    const int c = c0 + xchunk * 8;
#pragma unroll
    for (int s = 0; s < MDIE_MAX_SEG; ++s)
      if (s < a.nseg && c >= a.seg[s].ch_begin && c < a.seg[s].ch_end) {
        xbase = a.seg[s].ptr + (size_t)(c - a.seg[s].ch_begin) * 2;
        xstride = a.seg[s].stride * 2;
      }
  }
  const int xdst0 = (xchunk >> 1) * G::XPLANE + (xchunk & 1) * 16;
  const bool has_pre = PRE && xbase != nullptr;
  const int ychunk = tid % UPY;
  const bool ylive = o0 + ychunk * 8 < a.cout_st;
  const char* ybase = a.dy + (size_t)(o0 + ychunk * 8) * 2;
  const int ydst0 = (ychunk >> 1) * G::YPLANE + (ychunk & 1) * 16;

  // ---- fragment read addresses ----
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const char* xrd = lds_x + (wc * CSW) * G::XPLANE + (4 * g + q) * 32 + p * 8;
  const char* yrd = lds_y + (wo * OSW) * G::YPLANE + (4 * g + q) * 32 + p * 8;
  const bool wave_live = c0 + wc * CSW * 16 < a.cin_st && o0 + wo * OSW * 16 < a.cout_st;

  f32x4 acc[NTAP][CSW][OSW];
#pragma unroll
  for (int t = 0; t < NTAP; ++t)
#pragma unroll
    for (int i = 0; i < CSW; ++i)
#pragma unroll
      for (int j = 0; j < OSW; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int t_begin = split * a.tiles_per_split, t_end = min(t_begin + a.tiles_per_split, a.total_tiles);
  const int tpi = a.tiles_x * a.tiles_y;

  // ---- PF: the tile in flight (registers) ----
  uint4 xv[PF ? X_IT : 1], yv[PF ? Y_IT : 1];
  unsigned xokm = 0;
  auto issue_tile = [&](int tile) {
    const int img = tile / tpi;
    const int trem = tile - img * tpi;
    const int ty = trem / a.tiles_x, tx = trem - ty * a.tiles_x;
    const int y0 = ty * WT, x0 = tx * WT;
    const size_t ibase = (size_t)img * a.H * a.W;
    int pix = tid / UPX;
    int py = pix / G::PWD, px = pix - py * G::PWD;
    xokm = 0;
#pragma unroll
    for (int it = 0; it < (PF ? X_IT : 0); ++it) {
      const int gy = y0 + py - G::PAD, gx = x0 + px - G::PAD;
      const bool ok = py < G::PWD && xbase != nullptr && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      xv[it] = make_uint4(0, 0, 0, 0);
      if (ok) xv[it] = *reinterpret_cast<const uint4*>(xbase + (ibase + (size_t)gy * a.W + gx) * xstride);
      xokm |= ok ? 1u << it : 0u;
      px += X_PPI % G::PWD; py += X_PPI / G::PWD;
      if (px >= G::PWD) { px -= G::PWD; py += 1; }
    }
    pix = tid / UPY;
#pragma unroll
    for (int it = 0; it < (PF ? Y_IT : 0); ++it) {
      const int gy = y0 + (pix >> 4), gx = x0 + (pix & 15);
      yv[it] = make_uint4(0, 0, 0, 0);
      if (pix < WT * WT && ylive && gy < a.H && gx < a.W) yv[it] = *reinterpret_cast<const uint4*>(ybase + (ibase + (size_t)gy * a.W + gx) * a.dy_stride * 2);
      pix += Y_PPI;
    }
  };
  auto commit_tile = [&]() {   // pre-activation and the LDS images of the tile in flight
    float psc[8], psh[8];
    if (has_pre) {
      const float4 s0 = *reinterpret_cast<const float4*>(a.pre_scale + c0 + xchunk * 8), s1 = *reinterpret_cast<const float4*>(a.pre_scale + c0 + xchunk * 8 + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(a.pre_shift + c0 + xchunk * 8), b1 = *reinterpret_cast<const float4*>(a.pre_shift + c0 + xchunk * 8 + 4);
      psc[0] = s0.x; psc[1] = s0.y; psc[2] = s0.z; psc[3] = s0.w; psc[4] = s1.x; psc[5] = s1.y; psc[6] = s1.z; psc[7] = s1.w;
      psh[0] = b0.x; psh[1] = b0.y; psh[2] = b0.z; psh[3] = b0.w; psh[4] = b1.x; psh[5] = b1.y; psh[6] = b1.z; psh[7] = b1.w;
    }
    int pix = tid / UPX;
    int py = pix / G::PWD, px = pix - py * G::PWD;
#pragma unroll
    for (int it = 0; it < (PF ? X_IT : 0); ++it) {
      uint4 v = xv[it];
      if (has_pre && ((xokm >> it) & 1u)) {   // pre-activation BN + ReLU of the dense layers; the zero padding stays zero
        float f[8];
        Vec16<T>::unpack(v, f);
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = fmaxf(fmaf(f[i], psc[i], psh[i]), 0.f);
        v = Vec16<T>::pack(f);
      }
      if (py < G::PWD) *reinterpret_cast<uint4*>(lds_x + xdst0 + (py * G::PWD + px) * 32) = v;
      px += X_PPI % G::PWD; py += X_PPI / G::PWD;
      if (px >= G::PWD) { px -= G::PWD; py += 1; }
    }
    pix = tid / UPY;
#pragma unroll
    for (int it = 0; it < (PF ? Y_IT : 0); ++it) {
      if (pix < WT * WT) *reinterpret_cast<uint4*>(lds_y + ydst0 + pix * 32) = yv[it];
      pix += Y_PPI;
    }
  };
  if constexpr (PF) { if (t_begin < t_end) issue_tile(t_begin); }

  for (int tile = t_begin; tile < t_end; ++tile) {
    const int img = tile / tpi;
    const int trem = tile - img * tpi;
    const int ty = trem / a.tiles_x, tx = trem - ty * a.tiles_x;
    const int y0 = ty * WT, x0 = tx * WT;
    const size_t ibase = (size_t)img * a.H * a.W;

    if (tile > t_begin) __syncthreads();   // the previous tile's fragment reads are done
    if constexpr (PF) {
      commit_tile();
      __syncthreads();
      if (tile + 1 < t_end) issue_tile(tile + 1);   // (workgroup-uniform)
    } else {
    // ---- stage X patch: unit u = tid + it*256 -> patch pixel u / UPX (in batches, to bound the registers in flight) ----
    {
      int pix = tid / UPX;
      int py = pix / G::PWD, px = pix - py * G::PWD;
#pragma unroll
      for (int b0 = 0; b0 < X_IT; b0 += X_BATCH) {
        uint4 v[X_BATCH];
        int dsts[X_BATCH];
        bool oks[X_BATCH];
#pragma unroll
        for (int j = 0; j < X_BATCH; ++j) {
          const int gy = y0 + py - G::PAD, gx = x0 + px - G::PAD;
          const bool in_patch = b0 + j < X_IT && py < G::PWD;
          const bool ok = in_patch && xbase != nullptr && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
          v[j] = make_uint4(0, 0, 0, 0);
          if (ok) v[j] = *reinterpret_cast<const uint4*>(xbase + (ibase + (size_t)gy * a.W + gx) * xstride);
          oks[j] = ok;
          dsts[j] = in_patch ? xdst0 + (py * G::PWD + px) * 32 : -1;
          px += X_PPI % G::PWD; py += X_PPI / G::PWD;
          if (px >= G::PWD) { px -= G::PWD; py += 1; }
        }
        float psc[8], psh[8];      // re-read per batch (L1-resident) rather than held across the MFMA phase
        if (has_pre) {
          const float4 s0 = *reinterpret_cast<const float4*>(a.pre_scale + c0 + xchunk * 8), s1 = *reinterpret_cast<const float4*>(a.pre_scale + c0 + xchunk * 8 + 4);
          const float4 b0 = *reinterpret_cast<const float4*>(a.pre_shift + c0 + xchunk * 8), b1 = *reinterpret_cast<const float4*>(a.pre_shift + c0 + xchunk * 8 + 4);
          psc[0] = s0.x; psc[1] = s0.y; psc[2] = s0.z; psc[3] = s0.w; psc[4] = s1.x; psc[5] = s1.y; psc[6] = s1.z; psc[7] = s1.w;
          psh[0] = b0.x; psh[1] = b0.y; psh[2] = b0.z; psh[3] = b0.w; psh[4] = b1.x; psh[5] = b1.y; psh[6] = b1.z; psh[7] = b1.w;
        }
#pragma unroll
        for (int j = 0; j < X_BATCH; ++j) {
          if (has_pre && oks[j]) {   // pre-activation BN + ReLU of the dense layers; the zero padding stays zero
            float f[8];
            Vec16<T>::unpack(v[j], f);
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] = fmaxf(fmaf(f[i], psc[i], psh[i]), 0.f);
            v[j] = Vec16<T>::pack(f);
          }
          if (dsts[j] >= 0) *reinterpret_cast<uint4*>(lds_x + dsts[j]) = v[j];
        }
      }
    }
    // ---- stage dY tile ----
    {
      uint4 v[Y_IT];
      int pix = tid / UPY;
#pragma unroll
      for (int it = 0; it < Y_IT; ++it) {
        const int py = pix >> 4, px = pix & 15;
        const int gy = y0 + py, gx = x0 + px;
        v[it] = make_uint4(0, 0, 0, 0);
        if (pix < WT * WT && ylive && gy < a.H && gx < a.W)
          v[it] = *reinterpret_cast<const uint4*>(ybase + (ibase + (size_t)gy * a.W + gx) * a.dy_stride * 2);
        pix += Y_PPI;
      }
      pix = tid / UPY;
#pragma unroll
      for (int it = 0; it < Y_IT; ++it) {
        if (pix < WT * WT) *reinterpret_cast<uint4*>(lds_y + ydst0 + pix * 32) = v[it];
        pix += Y_PPI;
      }
    }
    __syncthreads();
    }   // (!PF)

    if (wave_live) {   // wave-uniform: EXEC stays all ones around the transposing reads
#pragma unroll WG_KUNROLL
      for (int k = 0; k < WT / 2; ++k) {
        v8s bf[OSW];
#pragma unroll
        for (int os = 0; os < OSW; ++os) {
          const v4s lo = lds_tr16(yrd + os * G::YPLANE + (2 * k) * WT * 32);
          const v4s hi = lds_tr16(yrd + os * G::YPLANE + (2 * k + 1) * WT * 32);
          bf[os] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int t = 0; t < NTAP; ++t) {
          const int kh = t / KS, kw = t - kh * KS;
#pragma unroll
          for (int cs = 0; cs < CSW; ++cs) {
            const v4s lo = lds_tr16(xrd + cs * G::XPLANE + ((2 * k + kh) * G::PWD + kw) * 32);
            const v4s hi = lds_tr16(xrd + cs * G::XPLANE + ((2 * k + 1 + kh) * G::PWD + kw) * 32);
            const v8s af = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int os = 0; os < OSW; ++os)
              acc[t][cs][os] = mma_tr<T>(af, bf[os], acc[t][cs][os]);
          }
        }
      }
    }
  }

  if (wave_live) {
    float* out = a.scratch + (size_t)split * NTAP * a.cin_st * a.cout_st;
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
      for (int cs = 0; cs < CSW; ++cs)
#pragma unroll
        for (int os = 0; os < OSW; ++os)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int c = c0 + (wc * CSW + cs) * 16 + g * 4 + r, o = o0 + (wo * OSW + os) * 16 + (lane & 15);
            if (c < a.cin_st && o < a.cout_st) out[((size_t)t * a.cin_st + c) * a.cout_st + o] = acc[t][cs][os][r];
          }
  }
}

// ---- wgrad, fp32: the same tiling on the exact-f32 MFMA v_mfma_f32_16x16x4_f32 ------------------------------------------------
// Operand of lane l: row (l & 15) = channel, k = (l >> 4) = one of 4 consecutive pixels -- one float, read with a plain
// ds_read_b32 from the same planar image (planes of 16 channels, 64-byte rows): 16 lanes x 4 rows = 256 contiguous bytes
// per read, a tap is an immediate offset.  The MFMA is 16x slower than the bf16 one, so this kernel is matrix-pipe bound
// (2304 MFMAs of 32 cycles per wave per 16x16-pixel tile) and one workgroup per CU (148 KiB of LDS) is enough.
template <int NTAP> struct WgGeomF {
  static constexpr int PAD = NTAP == 9 ? 1 : 0;
  static constexpr int PWD = WT + 2 * PAD;
  static constexpr int PROWS = PWD * PWD;
  static constexpr int XPLANE = PROWS * 64 + 64;
  static constexpr int YPLANE = WT * WT * 64;
};

template <int NTAP, int CSW, int OSW, int WC, int WO, bool PRE>
__global__ __launch_bounds__(TR_THREADS, 1) void wgrad_tile_f32_kernel(const WgradTileArgs a) {
  using G = WgGeomF<NTAP>;
  constexpr int KS = NTAP == 9 ? 3 : 1;
  constexpr int NCS_T = CSW * WC, NOS_T = OSW * WO;
  constexpr int UPX = NCS_T * 4, UPY = NOS_T * 4;             // 16-byte units (4 floats) per pixel
  constexpr int X_UNITS = G::PROWS * UPX, Y_UNITS = WT * WT * UPY;
  constexpr int X_IT = (X_UNITS + TR_THREADS - 1) / TR_THREADS, Y_IT = (Y_UNITS + TR_THREADS - 1) / TR_THREADS;
  constexpr int XB = X_IT > 8 ? 8 : X_IT, YB = Y_IT > 8 ? 8 : Y_IT;
  constexpr int X_PPI = TR_THREADS / UPX, Y_PPI = TR_THREADS / UPY;
  static_assert(WC * WO == TR_THREADS / 64, "wave grid");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_x = smem;
  char* lds_y = smem + NCS_T * G::XPLANE;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave % WC, wo = wave / WC;
  const int ot = blockIdx.x % a.o_tiles, ct = blockIdx.x / a.o_tiles;
  const int c0 = ct * NCS_T * 16, o0 = ot * NOS_T * 16;
  const int split = blockIdx.y;

  const int xchunk = tid % UPX;                      // this thread's 4-channel column of the c-tile
  const char* xbase = nullptr; int xstride = 0;
  {
    const int c = c0 + xchunk * 4;
#pragma unroll
    for (int s = 0; s < MDIE_MAX_SEG; ++s)
      if (s < a.nseg && c >= a.seg[s].ch_begin && c < a.seg[s].ch_end) {
        xbase = a.seg[s].ptr + (size_t)(c - a.seg[s].ch_begin) * 4;
        xstride = a.seg[s].stride * 4;
      }
  }
  const int xdst0 = (xchunk >> 2) * G::XPLANE + (xchunk & 3) * 16;
  const bool has_pre = PRE && xbase != nullptr;
  const int ychunk = tid % UPY;
  const bool ylive = o0 + ychunk * 4 < a.cout_st;
  const char* ybase = a.dy + (size_t)(o0 + ychunk * 4) * 4;
  const int ydst0 = (ychunk >> 2) * G::YPLANE + (ychunk & 3) * 16;

  const int kq = lane >> 4, m = lane & 15;
  const char* xrd = lds_x + (wc * CSW) * G::XPLANE + kq * 64 + m * 4;
  const char* yrd = lds_y + (wo * OSW) * G::YPLANE + kq * 64 + m * 4;
  const bool wave_live = c0 + wc * CSW * 16 < a.cin_st && o0 + wo * OSW * 16 < a.cout_st;

  f32x4 acc[NTAP][CSW][OSW];
#pragma unroll
  for (int t = 0; t < NTAP; ++t)
#pragma unroll
    for (int i = 0; i < CSW; ++i)
#pragma unroll
      for (int j = 0; j < OSW; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int t_begin = split * a.tiles_per_split, t_end = min(t_begin + a.tiles_per_split, a.total_tiles);
  const int tpi = a.tiles_x * a.tiles_y;
  for (int tile = t_begin; tile < t_end; ++tile) {
    const int img = tile / tpi;
    const int trem = tile - img * tpi;
    const int ty = trem / a.tiles_x, tx = trem - ty * a.tiles_x;
    const int y0 = ty * WT, x0 = tx * WT;
    const size_t ibase = (size_t)img * a.H * a.W;
    if (tile > t_begin) __syncthreads();
    {
      int pix = tid / UPX;
      int py = pix / G::PWD, px = pix - py * G::PWD;
      float psc[4] = {1.f, 1.f, 1.f, 1.f}, psh[4] = {0.f, 0.f, 0.f, 0.f};
      if (has_pre) {
        const float4 s4 = *reinterpret_cast<const float4*>(a.pre_scale + c0 + xchunk * 4), b4 = *reinterpret_cast<const float4*>(a.pre_shift + c0 + xchunk * 4);
        psc[0] = s4.x; psc[1] = s4.y; psc[2] = s4.z; psc[3] = s4.w; psh[0] = b4.x; psh[1] = b4.y; psh[2] = b4.z; psh[3] = b4.w;
      }
#pragma unroll 1
      for (int b0 = 0; b0 < X_IT; b0 += XB) {
        float4 v[XB];
        int dsts[XB];
#pragma unroll
        for (int j = 0; j < XB; ++j) {
          const int gy = y0 + py - G::PAD, gx = x0 + px - G::PAD;
          const bool in_patch = b0 + j < X_IT && py < G::PWD;
          const bool ok = in_patch && xbase != nullptr && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
          v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (ok) {
            v[j] = *reinterpret_cast<const float4*>(xbase + (ibase + (size_t)gy * a.W + gx) * xstride);
            if (has_pre) {   // pre-activation BN + ReLU of the dense layers; the zero padding stays zero
              v[j].x = fmaxf(fmaf(v[j].x, psc[0], psh[0]), 0.f); v[j].y = fmaxf(fmaf(v[j].y, psc[1], psh[1]), 0.f);
              v[j].z = fmaxf(fmaf(v[j].z, psc[2], psh[2]), 0.f); v[j].w = fmaxf(fmaf(v[j].w, psc[3], psh[3]), 0.f);
            }
          }
          dsts[j] = in_patch ? xdst0 + (py * G::PWD + px) * 64 : -1;
          px += X_PPI % G::PWD; py += X_PPI / G::PWD;
          if (px >= G::PWD) { px -= G::PWD; py += 1; }
        }
#pragma unroll
        for (int j = 0; j < XB; ++j)
          if (dsts[j] >= 0) *reinterpret_cast<float4*>(lds_x + dsts[j]) = v[j];
      }
    }
    {
      int pix = tid / UPY;
#pragma unroll 1
      for (int b0 = 0; b0 < Y_IT; b0 += YB) {
        float4 v[YB];
        int pp[YB];
#pragma unroll
        for (int j = 0; j < YB; ++j) {
          const int py = pix >> 4, px = pix & 15;
          const int gy = y0 + py, gx = x0 + px;
          v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
          pp[j] = (b0 + j < Y_IT && pix < WT * WT) ? pix : -1;
          if (pp[j] >= 0 && ylive && gy < a.H && gx < a.W)
            v[j] = *reinterpret_cast<const float4*>(ybase + (ibase + (size_t)gy * a.W + gx) * a.dy_stride * 4);
          pix += Y_PPI;
        }
#pragma unroll
        for (int j = 0; j < YB; ++j)
          if (pp[j] >= 0) *reinterpret_cast<float4*>(lds_y + ydst0 + pp[j] * 64) = v[j];
      }
    }
    __syncthreads();

    if (wave_live) {
#pragma unroll 1
      for (int ky = 0; ky < WT; ++ky) {
        const char* xr = xrd + ky * G::PWD * 64;
        const char* yr = yrd + ky * WT * 64;
#pragma unroll
        for (int kx = 0; kx < WT / 4; ++kx) {
          float bfv[OSW];
#pragma unroll
          for (int os = 0; os < OSW; ++os) bfv[os] = *reinterpret_cast<const float*>(yr + os * G::YPLANE + kx * 4 * 64);
#pragma unroll
          for (int t = 0; t < NTAP; ++t) {
            const int kh = t / KS, kw = t - kh * KS;
#pragma unroll
            for (int cs = 0; cs < CSW; ++cs) {
              const float af = *reinterpret_cast<const float*>(xr + cs * G::XPLANE + (kh * G::PWD + kx * 4 + kw) * 64);
#pragma unroll
              for (int os = 0; os < OSW; ++os) acc[t][cs][os] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bfv[os], acc[t][cs][os], 0, 0, 0);
            }
          }
        }
      }
    }
  }

  if (wave_live) {
    float* out = a.scratch + (size_t)split * NTAP * a.cin_st * a.cout_st;
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
      for (int cs = 0; cs < CSW; ++cs)
#pragma unroll
        for (int os = 0; os < OSW; ++os)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int c = c0 + (wc * CSW + cs) * 16 + kq * 4 + r, o = o0 + (wo * OSW + os) * 16 + m;
            if (c < a.cin_st && o < a.cout_st) out[((size_t)t * a.cin_st + c) * a.cout_st + o] = acc[t][cs][os][r];
          }
  }
}

struct WgTilePlan { int cfg, c_tile, o_tile, c_tiles, o_tiles, splits, tiles_per_split, total_tiles, tiles_x, tiles_y; };

// cfg 0: 64c x 64o (2x2 waves of 32x32); 1: 64c x 16o (cout-thin); 2: 16c x 64o (cin-thin)
// (pre9: a 3x3 layer with the pre-activation prologue never takes cfg 0 -- that instantiation needs 17 registers more than
//  the 256 of two waves per SIMD and would spill; CDAN has no such layer, its pre-activated 3x3 layers have 16 outputs, but
//  the C ABI admits one.  cfg 1 has at least as many (c, o) tiles, so never more splits: the workspace query stays an upper bound.)
static WgTilePlan wgrad_tile_plan(int B, int H, int W, int cin_st, int cout_st, bool pre9 = false) {
  WgTilePlan p{};
  if (cout_st % 64 != 0 || pre9) { p.cfg = 1; p.c_tile = 64; p.o_tile = 16; }
  else if (cin_st <= 16) { p.cfg = 2; p.c_tile = 16; p.o_tile = 64; }
  else { p.cfg = 0; p.c_tile = 64; p.o_tile = 64; }
  p.c_tiles = cdiv(cin_st, p.c_tile); p.o_tiles = cdiv(cout_st, p.o_tile);
  p.tiles_x = cdiv(W, WT); p.tiles_y = cdiv(H, WT);
  p.total_tiles = B * p.tiles_x * p.tiles_y;
  const int base = p.c_tiles * p.o_tiles;
  int splits = cdiv(base >= 16 ? 256 : 512, base);     // 1-2 resident workgroups per CU; fewer slabs for the big layers
  if (splits > p.total_tiles) splits = p.total_tiles;
  if (splits < 1) splits = 1;
  p.tiles_per_split = cdiv(p.total_tiles, splits);
  p.splits = cdiv(p.total_tiles, p.tiles_per_split);
  return p;
}

template <typename T, int NTAP, int CSW, int OSW, int WC, int WO, bool PRE>
static void launch_wgrad_tile_t(const WgradTileArgs& a, const WgTilePlan& p, hipStream_t s) {
  using G = WgGeom<NTAP>;
  const size_t lds = (size_t)CSW * WC * G::XPLANE + (size_t)OSW * WO * G::YPLANE;
#ifdef EXP_NO_WGRAD_PF   // (A/B builds only)
  constexpr bool PF = false;
#else
  constexpr bool PF = CSW * OSW == 1;
#endif
  auto kern = wgrad_tile_kernel<T, NTAP, CSW, OSW, WC, WO, PRE, PF>;
  static LdsOptIn opt;   // > 64 KiB of dynamic LDS needs the opt-in once per kernel and device
  if (!opt.ensure(reinterpret_cast<const void*>(kern), (int)lds)) return;   // (the caller's MDIE_LAUNCH_CHECK reports it)
  hipLaunchKernelGGL(kern, dim3(p.c_tiles * p.o_tiles, p.splits), dim3(TR_THREADS), lds, s, a);
}

template <int NTAP, int CSW, int OSW, int WC, int WO>
static void launch_wgrad_tile_f32(const WgradTileArgs& a, const WgTilePlan& p, hipStream_t s) {
  using G = WgGeomF<NTAP>;
  const size_t lds = (size_t)CSW * WC * G::XPLANE + (size_t)OSW * WO * G::YPLANE;
  const dim3 grid(p.c_tiles * p.o_tiles, p.splits);
  if (a.pre_scale) {
    auto kern = wgrad_tile_f32_kernel<NTAP, CSW, OSW, WC, WO, true>;
    static LdsOptIn opt;
    if (!opt.ensure(reinterpret_cast<const void*>(kern), (int)lds)) return;
    hipLaunchKernelGGL(kern, grid, dim3(TR_THREADS), lds, s, a);
  } else {
    auto kern = wgrad_tile_f32_kernel<NTAP, CSW, OSW, WC, WO, false>;
    static LdsOptIn opt;
    if (!opt.ensure(reinterpret_cast<const void*>(kern), (int)lds)) return;
    hipLaunchKernelGGL(kern, grid, dim3(TR_THREADS), lds, s, a);
  }
}

// the pre-activation prologue is a separate instantiation: its 16 extra live registers would push the
// 9-tap 64x64 configuration (never used with a prologue by this network: dense 3x3 layers have 16 outputs) into spills
template <typename T, int NTAP, int CSW, int OSW, int WC, int WO>
static void launch_wgrad_tile_h(const WgradTileArgs& a, const WgTilePlan& p, hipStream_t s) {
  if constexpr (NTAP == 9 && CSW * OSW == 4) launch_wgrad_tile_t<T, NTAP, CSW, OSW, WC, WO, false>(a, p, s);   // (wgrad_tile_plan: never with a prologue)
  else if (a.pre_scale) launch_wgrad_tile_t<T, NTAP, CSW, OSW, WC, WO, true>(a, p, s);
  else launch_wgrad_tile_t<T, NTAP, CSW, OSW, WC, WO, false>(a, p, s);
}
template <int NTAP, int CSW, int OSW, int WC, int WO>
static void launch_wgrad_tile(const WgradTileArgs& a, const WgTilePlan& p, hipStream_t s) { launch_wgrad_tile_h<bf16, NTAP, CSW, OSW, WC, WO>(a, p, s); }
template <int NTAP, int CSW, int OSW, int WC, int WO>
static void launch_wgrad_tile_f16(const WgradTileArgs& a, const WgTilePlan& p, hipStream_t s) { launch_wgrad_tile_h<f16, NTAP, CSW, OSW, WC, WO>(a, p, s); }

}  // namespace mdie

using namespace mdie;

static int pack_job_check(const char* what, const mdie_pack_job& j) {
  MDIE_REQUIRE(j.ksize == 1 || j.ksize == 3, "%s: ksize %d", what, j.ksize);
  MDIE_REQUIRE(j.w && j.dst && j.cout > 0 && j.cin > 0, "%s: null/empty", what);
  MDIE_REQUIRE(j.cout_stored % 16 == 0 && j.out_gap >= 0 && j.out_split >= 0 && j.cout_stored >= j.cout + (j.out_split < j.cout ? j.out_gap : 0),
               "%s: cout_stored %d too small for cout %d out_split %d out_gap %d", what, j.cout_stored, j.cout, j.out_split, j.out_gap);
  MDIE_REQUIRE(j.cin_stored % 16 == 0 && j.cin_stored >= j.cin + (j.split < j.cin ? j.gap : 0) && j.gap >= 0 && j.split >= 0,
               "%s: cin_stored %d too small for cin %d split %d gap %d", what, j.cin_stored, j.cin, j.split, j.gap);
  return MDIE_OK;
}

extern "C" int mdie_pack_conv_weight_job(int dtype, const mdie_pack_job* job, void* stream) {
  MDIE_REQUIRE(dtype_valid(dtype), "mdie_pack_conv_weight_job: bad dtype %d", dtype);
  MDIE_REQUIRE(job != nullptr, "mdie_pack_conv_weight_job: null job");
  if (int e = pack_job_check("mdie_pack_conv_weight_job", *job)) return e;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int kc = dtype_kc(dtype);
  const size_t total = (size_t)cdiv(job->cin_stored, kc) * 4 * job->ksize * job->ksize * job->cout_stored * (kc / 4);
  MDIE_SWITCH_T(dtype, hipLaunchKernelGGL((pack_weight_kernel<T>), dim3(tr_grid(total)), dim3(TR_THREADS), 0, s, *job));
  MDIE_LAUNCH_CHECK("mdie_pack_conv_weight_job");
  return MDIE_OK;
}

extern "C" int mdie_pack_conv_weight_dev(int dtype, int ksize, int transposed, const float* w, int cout, int cin, int cout_stored,
                                         int cin_stored, int split, int gap, void* dst, void* stream) {
  mdie_pack_job j{};
  j.w = w; j.dst = dst; j.ksize = ksize; j.transposed = transposed; j.cout = cout; j.cin = cin; j.cout_stored = cout_stored; j.cin_stored = cin_stored;
  j.split = split; j.gap = gap; j.out_split = cout; j.out_gap = 0;
  return mdie_pack_conv_weight_job(dtype, &j, stream);
}

extern "C" int mdie_pack_conv_weights_batch(int dtype, const mdie_pack_job* jobs_dev, int n_jobs, void* stream) {
  MDIE_REQUIRE(dtype_valid(dtype), "mdie_pack_conv_weights_batch: bad dtype %d", dtype);
  MDIE_REQUIRE(jobs_dev != nullptr && n_jobs > 0 && n_jobs <= 65535, "mdie_pack_conv_weights_batch: %d jobs", n_jobs);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  MDIE_SWITCH_T(dtype, hipLaunchKernelGGL((pack_weight_batch_kernel<T>), dim3(48, n_jobs), dim3(TR_THREADS), 0, s, jobs_dev));
  MDIE_LAUNCH_CHECK("mdie_pack_conv_weights_batch");
  return MDIE_OK;
}

extern "C" size_t mdie_conv_wgrad_workspace_bytes(int B, int H, int W, int ksize, int cin_stored, int cout_stored) {
  if (B <= 0 || H <= 0 || W <= 0 || cin_stored <= 0 || cout_stored <= 0) return 0;
  return (size_t)wgrad_tile_plan(B, H, W, cin_stored, cout_stored).splits * ksize * ksize * cin_stored * cout_stored * sizeof(float);
}

extern "C" int mdie_conv_wgrad(const mdie_wgrad_desc* d, void* stream) {
  MDIE_REQUIRE(d != nullptr, "mdie_conv_wgrad: null descriptor");
  MDIE_REQUIRE(dtype_valid(d->dtype), "mdie_conv_wgrad: bad dtype %d", d->dtype);
  MDIE_REQUIRE(d->ksize == 3 || d->ksize == 1, "mdie_conv_wgrad: ksize %d", d->ksize);
  MDIE_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0, "mdie_conv_wgrad: empty extent");
  MDIE_REQUIRE(d->nseg >= 1 && d->nseg <= MDIE_MAX_SEG, "mdie_conv_wgrad: nseg %d", d->nseg);
  MDIE_REQUIRE(d->dy && d->dw && d->workspace && d->cout > 0 && d->cin > 0, "mdie_conv_wgrad: null/empty");
  MDIE_REQUIRE(d->cout_stored % 16 == 0 && d->cout_stored >= d->cout && d->dy_stride >= d->cout_stored, "mdie_conv_wgrad: cout_stored %d", d->cout_stored);
  MDIE_REQUIRE((d->pre_scale == nullptr) == (d->pre_shift == nullptr), "mdie_conv_wgrad: pre_scale / pre_shift must both be given or both be null");
  WgradTileArgs t{};
  t.B = d->B; t.H = d->H; t.W = d->W; t.nseg = d->nseg;
  int c = 0;
  for (int s = 0; s < d->nseg; ++s) {
    MDIE_REQUIRE(d->in[s].ptr && d->in[s].channels > 0 && d->in[s].channels % 16 == 0, "mdie_conv_wgrad: segment %d channels %d", s, d->in[s].channels);
    t.seg[s].ptr = reinterpret_cast<const char*>(d->in[s].ptr);
    t.seg[s].ch_begin = c; c += d->in[s].channels; t.seg[s].ch_end = c;
    t.seg[s].stride = d->in[s].stride;
  }
  MDIE_REQUIRE(c >= d->cin + (d->split < d->cin ? d->gap : 0), "mdie_conv_wgrad: segments hold %d channels < cin %d + gap", c, d->cin);
  const int taps = d->ksize * d->ksize;
  const WgTilePlan p = wgrad_tile_plan(d->B, d->H, d->W, c, d->cout_stored, d->pre_scale != nullptr && taps == 9 && d->dtype != MDIE_F32);
  const size_t need = (size_t)p.splits * taps * c * d->cout_stored * sizeof(float);
  if (d->workspace_bytes < need) { set_error("mdie_conv_wgrad: workspace %zu < %zu", d->workspace_bytes, need); return MDIE_ENOSPC; }
  t.cin_st = c; t.cout_st = d->cout_stored;
  t.dy = reinterpret_cast<const char*>(d->dy); t.dy_stride = d->dy_stride;
  t.scratch = reinterpret_cast<float*>(d->workspace);
  t.pre_scale = d->pre_scale; t.pre_shift = d->pre_shift;
  t.tiles_x = p.tiles_x; t.tiles_y = p.tiles_y; t.total_tiles = p.total_tiles; t.tiles_per_split = p.tiles_per_split; t.o_tiles = p.o_tiles;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define MDIE_WG(LAUNCH)                                                \
  do {                                                                 \
    if (taps == 9) {                                                   \
      if (p.cfg == 0) LAUNCH<9, 2, 2, 2, 2>(t, p, s);                  \
      else if (p.cfg == 1) LAUNCH<9, 1, 1, 4, 1>(t, p, s);             \
      else LAUNCH<9, 1, 1, 1, 4>(t, p, s);                             \
    } else {                                                           \
      if (p.cfg == 0) LAUNCH<1, 2, 2, 2, 2>(t, p, s);                  \
      else if (p.cfg == 1) LAUNCH<1, 1, 1, 4, 1>(t, p, s);             \
      else LAUNCH<1, 1, 1, 1, 4>(t, p, s);                             \
    }                                                                  \
  } while (0)
  if (d->dtype == MDIE_F32) MDIE_WG(launch_wgrad_tile_f32);
  else if (d->dtype == MDIE_F16) MDIE_WG(launch_wgrad_tile_f16);
  else MDIE_WG(launch_wgrad_tile);
#undef MDIE_WG
  MDIE_LAUNCH_CHECK("mdie_conv_wgrad");
  const size_t total = (size_t)taps * c * d->cout_stored;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(TR_THREADS), 0, s, p.splits, d->ksize, d->transposed, d->cout, d->cin,
                     d->cout_stored, c, d->split, d->gap, t.scratch, d->dw);
  MDIE_LAUNCH_CHECK("mdie_conv_wgrad");
  return MDIE_OK;
}
