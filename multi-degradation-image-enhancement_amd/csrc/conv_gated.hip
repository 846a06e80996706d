// CBAM's last pass fused into the convolution that consumes it: decoder stage 4 of the CDAN path,
//   out = relu(bn4(convT3x3( x * gate_c * sigmoid(bn(conv7x7(map))) * d1 )))           (models/cbam.py:72-82,91-95; models/cdan.py:148-152)
// where x is cbam3's input (64 channels at H/2 x W/2), d1 the encoder DenseBlock it is multiplied with and the 3x3 has 16 stored
// outputs.  As separate launches (cbam_spatial_kernel, then conv_kernel<3,16,16>) the gated tensor is written once and read back
// 1.27 times (18x18 patches): 134 + 170 MB per 32-image step for a tensor nothing else needs.  Here the gated value is formed
// while the convolution stages its patch:
//   * prologue: the tile's (16+2+6)^2 compressed-map patch -> the spatial gate s of the 18x18 patch pixels (the 7x7 convolution of
//     cbam_spatial_kernel, same accumulation order), the channel gate and ALL of the layer's weights (2 K chunks, 18 KB) -> LDS;
//   * staging of a K chunk: x and d1 units in registers -> ((x * gate) * s) * d1 in fp32, ONE rounding to the storage type -- the very
//     value cbam_spatial_kernel would have stored -- into conv_kernel's planar LDS image; zero padding stays zero;
//   * MFMA phase and epilogue: conv_kernel's (chunks, then taps, in sequence; conv_epilogue_t).
// The result is bit-identical to the two-launch path (tests/test_gpu_parity.py::test_cbam_spatial_fused_into_its_convolution): the
// choice between the two is free of the batch-independence question.
#include <algorithm>

#include "conv_kernel.hpp"

namespace mdie {


constexpr int GT_MAX_C = 64;     // cbam3 (the only CBAM whose consumer is a 16-output convolution); 48.6 KB of LDS: 3 workgroups per CU
constexpr int GT_CP = 24;      // compressed-map patch edge: 16 + 2 (conv halo) + 6 (7x7 halo)

#ifdef EXP_GSTAMPS   // diagnostic build only (tools/stamp_gated.py): shader-clock stamps of wave 0 into a buffer passed as e.residual
#define GSTAMP(i) do { if (dbg && tid == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); dbg[(size_t)patch * 16 + (i)] = t_; } } while (0)
#else
#define GSTAMP(i) do {} while (0)
#endif

template <typename T>
__global__ __launch_bounds__(CONV_THREADS, 3) void conv_gated_kernel(const GatedArgs a) {
  using G = ConvGeom<3, 16, 16>;
  constexpr int VEC = Traits<T>::VEC, KC = Traits<T>::KC;
  constexpr int PW = G::PW, NPS = 4;
  constexpr int PATCH_UNITS = PW * PW * 4, PATCH_IT = (PATCH_UNITS + CONV_THREADS - 1) / CONV_THREADS;   // 1296 units, 6 per thread (the last: 16 threads)
  constexpr int WCHUNK = 4 * 9 * 16 * 16;                  // bytes of one K chunk's weights
  static_assert(sizeof(T) == 2, "16-bit storage types only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_patch = smem;                                   // [4 planes]; the prologue's compressed-map patch lives here first
  char* lds_w = smem + 4 * G::PLANE;                        // [nchunk][q][tap][16][16 B]
  float* lds_gate = reinterpret_cast<float*>(lds_w + (GT_MAX_C / KC) * WCHUNK);   // [C]
  float* lds_s = lds_gate + GT_MAX_C;                       // [PW * PW] spatial gate of the patch pixels (0 outside the picture)
  float* lds_w7 = lds_s + PW * PW + 4;                      // [98]
  float* cpatch = reinterpret_cast<float*>(lds_patch);      // [2][GT_CP][GT_CP]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lq = lane >> 4, lp = lane & 15;
  const int patch = blockIdx.x * gridDim.y + blockIdx.y;    // an XCD slot (fastest index) owns a contiguous run of tiles
  const int tpi = a.tiles_x * a.tiles_y;
  if (patch >= tpi * a.B) return;
  const int img = patch / tpi, trem = patch - img * tpi;
  const int ty = trem / a.tiles_x, tx = trem - ty * a.tiles_x;
  const int y0 = ty * 16, x0 = tx * 16;
  const size_t img_pix = (size_t)img * a.H * a.W;
#ifdef EXP_GSTAMPS
  unsigned long long* dbg = reinterpret_cast<unsigned long long*>(const_cast<char*>(a.e.residual));
  EpiArgs e2 = a.e; e2.residual = nullptr;
#else
  const EpiArgs& e2 = a.e;
#endif
  GSTAMP(0);

  // ---- staging geometry: unit u = tid + 256 it -> patch pixel u >> 2, K group q = tid & 3.  Loads are unconditional (clamped
  // addresses; DESIGN.md section 4, gfx950 finding 2), the zeroing is a select; threads without a unit in the last iteration
  // write into the unused tail of their plane ----
  const int q = tid & 3;
  unsigned upix[PATCH_IT];      // bits 0..21: clamped pixel index inside the picture; bits 22..31: patch pixel (row * PW + col), or 0x3ff
  unsigned inside = 0;
#pragma unroll
  for (int it = 0; it < PATCH_IT; ++it) {
    const int pix = (tid >> 2) + 64 * it;
    const int py = pix / PW, px = pix - py * PW;
    const int gy = y0 + py - 1, gx = x0 + px - 1;
    const bool in_patch = pix < PW * PW;
    const bool ok = in_patch && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    const int cy = min(max(gy, 0), a.H - 1), cx = min(max(gx, 0), a.W - 1);
    upix[it] = (unsigned)(cy * a.W + cx) | ((unsigned)(in_patch ? pix : 0x3ff) << 22);
    inside |= (ok ? 1u : 0u) << it;
  }
  const unsigned xs = (unsigned)a.x_stride * sizeof(T), ms = (unsigned)a.mul_stride * sizeof(T);
  const char* const xb = a.x + img_pix * xs + q * 16;
  const char* const mb = a.mul + img_pix * ms + q * 16;
  uint4 xv[PATCH_IT], mv[PATCH_IT];
  auto load_chunk = [&](int chunk) {
#pragma unroll
    for (int it = 0; it < PATCH_IT; ++it) {
      const unsigned p = upix[it] & 0x3fffffu;
      xv[it] = *reinterpret_cast<const uint4*>(xb + __umul24(p, xs) + chunk * 64);
      mv[it] = *reinterpret_cast<const uint4*>(mb + __umul24(p, ms) + chunk * 64);
    }
  };
  // ---- prologue: compressed-map patch, 7x7 weights, channel gate, the layer's weights -> LDS.  Their loads go out FIRST and
  // unconditionally (clamped addresses, zeroing by select): vector loads return in order, so behind the chunk's 12 HBM loads the
  // LDS writes of these few L2-resident dwords would wait for all of them (tools/stamp_gated.py: 16 k cycles to the first barrier) ----
  {
    constexpr int CP_IT = (GT_CP * GT_CP + CONV_THREADS - 1) / CONV_THREADS;      // 3 (the last: 64 threads)
    constexpr int W_IT = ((GT_MAX_C / KC) * (WCHUNK / 16) + CONV_THREADS - 1) / CONV_THREADS;   // 5 (the last: half the threads)
    float2 cv[CP_IT];
    bool cin_[CP_IT];
#pragma unroll
    for (int k = 0; k < CP_IT; ++k) {
      const int i = min(tid + k * CONV_THREADS, GT_CP * GT_CP - 1);
      const int py = i / GT_CP, px = i - py * GT_CP;
      const int gy = y0 + py - 4, gx = x0 + px - 4;
      cin_[k] = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      const int cy = min(max(gy, 0), a.H - 1), cx = min(max(gx, 0), a.W - 1);
      cv[k] = *reinterpret_cast<const float2*>(a.map + (img_pix + (size_t)(cy * a.W + cx)) * 2);
    }
    const float w7v = a.w7[tid < 98 ? tid : 0];
    const float gv = a.gate[(size_t)img * a.C + (tid < a.C ? tid : 0)];
    uint4 wl[W_IT];
    const int wunits = a.nchunk * (WCHUNK / 16);
#pragma unroll
    for (int k = 0; k < W_IT; ++k) wl[k] = *reinterpret_cast<const uint4*>(a.weight + (size_t)min(tid + k * CONV_THREADS, wunits - 1) * 16);
    load_chunk(0);                                          // (in flight under everything up to the first staging)
    GSTAMP(1);
#pragma unroll
    for (int k = 0; k < CP_IT; ++k) {
      const int i = tid + k * CONV_THREADS;
      if (i < GT_CP * GT_CP) { cpatch[i] = cin_[k] ? cv[k].x : 0.f; cpatch[GT_CP * GT_CP + i] = cin_[k] ? cv[k].y : 0.f; }
    }
    if (tid < 98) lds_w7[tid] = w7v;
    if (tid < a.C) lds_gate[tid] = gv;
#pragma unroll
    for (int k = 0; k < W_IT; ++k)
      if (tid + k * CONV_THREADS < wunits) *reinterpret_cast<uint4*>(lds_w + (tid + k * CONV_THREADS) * 16) = wl[k];
  }
  const float bn0 = a.bn[0], bn1 = a.bn[1];
  __syncthreads();
  GSTAMP(2);
  // spatial gate of the 18x18 patch pixels (cbam_spatial_kernel's accumulation order: channel, row, column).  One channel's 49 taps
  // unrolled: their LDS reads are batched (rolled up, every row of 7 was an exposed LDS round trip: 8.6 k cycles for two pixels)
  for (int i = tid; i < PW * PW; i += CONV_THREADS) {
    const int py = i / PW, px = i - py * PW;
    const int gy = y0 + py - 1, gx = x0 + px - 1;
    float acc = 0.f;
#pragma unroll 1
    for (int ch = 0; ch < 2; ++ch) {
      const float* cp = cpatch + ch * GT_CP * GT_CP + py * GT_CP + px;
      const float* wp = lds_w7 + ch * 49;
#pragma unroll
      for (int kh = 0; kh < 7; ++kh)
#pragma unroll
        for (int kw = 0; kw < 7; ++kw) acc = fmaf(wp[kh * 7 + kw], cp[kh * GT_CP + kw], acc);
    }
    const bool in_img = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    lds_s[i] = in_img ? sigmoidf(fmaf(acc, bn0, bn1)) : 0.f;
  }
  __syncthreads();                                          // (the compressed-map patch is dead: the planes may be written)
  GSTAMP(3);

  f32x4 acc[1][NPS];
#pragma unroll
  for (int j = 0; j < NPS; ++j) acc[0][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  int xoff0;
  {
    int y, x;
    tile_pixel<16>(wave * NPS, lp, y, x);
    xoff0 = lq * G::PLANE + (y * PWP + x) * 16;
  }
  using TS = TileStep<16, NPS>;
  const int woff = lq * (9 * 16 * 16) + lp * 16;

  for (int chunk = 0; chunk < a.nchunk; ++chunk) {
    if (chunk > 0) __syncthreads();                         // the previous chunk's operand reads are done
    {
      float g[VEC];
      const float4 g0 = *reinterpret_cast<const float4*>(lds_gate + chunk * KC + q * VEC), g1 = *reinterpret_cast<const float4*>(lds_gate + chunk * KC + q * VEC + 4);
      g[0] = g0.x; g[1] = g0.y; g[2] = g0.z; g[3] = g0.w; g[4] = g1.x; g[5] = g1.y; g[6] = g1.z; g[7] = g1.w;
#pragma unroll
      for (int it = 0; it < PATCH_IT; ++it) {
        const unsigned pp = upix[it] >> 22;
        const bool has = pp != 0x3ffu;
        const float sv = lds_s[has ? pp : 0];
        float f[VEC], m[VEC];
        Vec16<T>::unpack(xv[it], f);
        Vec16<T>::unpack(mv[it], m);
        // ((x * gate) * s) * d1: cbam_spatial_kernel's order and single rounding, as packed multiplies on DISTINCT register pairs --
        // s is copied into a pair of its own behind an opaque asm, so the compiler cannot fold the splat into op_sel (the form
        // tools/isa_guard.py bans from MFMA kernels, DESIGN.md section 4 finding 6)
        f32x2 svv = {sv, sv};
        asm volatile("" : "+v"(svv));
#pragma unroll
        for (int i = 0; i < VEC; i += 2) {
          f32x2 t = f32x2{f[i], f[i + 1]} * f32x2{g[i], g[i + 1]};
          t = t * svv;
          t = t * f32x2{m[i], m[i + 1]};
          f[i] = t[0]; f[i + 1] = t[1];
        }
        uint4 v = Vec16<T>::pack(f);
        const bool keep = (inside >> it) & 1u;
        v.x = keep ? v.x : 0u; v.y = keep ? v.y : 0u; v.z = keep ? v.z : 0u; v.w = keep ? v.w : 0u;
        const int py = (int)pp / PW, px = (int)pp - py * PW;
        const int dst = has ? q * G::PLANE + (py * PWP + px) * 16 : q * G::PLANE + PW * PWP * 16 + (tid >> 2 & 7) * 16;
        *reinterpret_cast<uint4*>(lds_patch + dst) = v;
      }
    }
    GSTAMP(4 + 4 * chunk);
    __syncthreads();
    GSTAMP(5 + 4 * chunk);
    if (chunk + 1 < a.nchunk) load_chunk(chunk + 1);        // (nchunk is 2 for this network: the second chunk's loads fly under the first chunk's MFMAs)
    const char* wl = lds_w + chunk * WCHUNK + woff;
    uint4 wf[2], xf[2][NPS];
    auto read_tap = [&](int tap, int b) {
      const int kh = tap / 3, kw = tap - kh * 3;
      wf[b] = *reinterpret_cast<const uint4*>(wl + tap * 16 * 16);
#pragma unroll
      for (int ps = 0; ps < NPS; ++ps)
        xf[b][ps] = *reinterpret_cast<const uint4*>(lds_patch + ((kh + TS::dy(ps)) * PWP + kw + TS::dx(ps)) * 16 + xoff0);
    };
    read_tap(0, 0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (tap + 1 < 9) read_tap(tap + 1, (tap + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ps = 0; ps < NPS; ++ps) acc[0][ps] = mma16<T>(wf[tap & 1], xf[tap & 1][ps], acc[0][ps]);
      __builtin_amdgcn_sched_barrier(0);
    }
    GSTAMP(6 + 4 * chunk);
  }
  float4 esc[1], esh[1];
  esc[0] = *reinterpret_cast<const float4*>(a.e.post_scale + lq * 4);
  esh[0] = *reinterpret_cast<const float4*>(a.e.post_shift + lq * 4);
  conv_epilogue<T, 1, NPS, 16>(e2, esc, esh, acc, img, y0, x0, 0, wave * NPS, lq, lp);
  GSTAMP(12);
}

constexpr int GT_LDS = 4 * ConvGeom<3, 16, 16>::PLANE + (GT_MAX_C / 32) * (4 * 9 * 16 * 16) + (GT_MAX_C + 18 * 18 + 4 + 100) * (int)sizeof(float);

// the convolution side of mdie_cbam_conv_fwd (cbam.hip runs the pool / gate / channel-pool passes first)
int launch_conv_gated(int dtype, const GatedArgs& g, hipStream_t stream) {
  GatedArgs a = g;
  a.tiles_x = cdiv(a.W, 16); a.tiles_y = cdiv(a.H, 16);
  const int tiles = a.tiles_x * a.tiles_y * a.B;
  const dim3 grid(8, cdiv(tiles, 8));
  TimedLaunch tl(MDIE_K_CONV3);
  if (dtype == MDIE_BF16) hipLaunchKernelGGL((conv_gated_kernel<bf16>), grid, dim3(CONV_THREADS), GT_LDS, stream, a);
  else hipLaunchKernelGGL((conv_gated_kernel<f16>), grid, dim3(CONV_THREADS), GT_LDS, stream, a);
  MDIE_LAUNCH_CHECK("mdie_cbam_conv_fwd");
  return MDIE_OK;
}

bool conv_gated_applicable(int dtype, int H, int W, int C, int cout_stored, int x_stride, int mul_stride, bool has_mul) {
  if (dtype == MDIE_F32 || !has_mul || C % 32 != 0 || C > GT_MAX_C || cout_stored != 16) return false;
  const size_t npx = (size_t)H * W, smax = (size_t)std::max(x_stride, mul_stride) * 2;
  return npx < ((size_t)1 << 22) && smax < ((size_t)1 << 24) && npx * smax < ((size_t)1 << 32);   // 22-bit pixel index, 24-bit multiplies
}

}  // namespace mdie
