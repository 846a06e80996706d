// Training-side kernels: the convolution backward pieces that carry ~2/3 of a
// training step's FLOPs.
//   dgrad  needs no kernel of its own: the input gradient of a 3x3 / pad-1 convolution is the same
//          convolution of dY with the flipped, in/out-transposed weights, i.e. mdie_conv_fwd on weights
//          packed with the opposite `transposed` flag (mdie_pack_conv_weight_dev below repacks on the GPU
//          every step, since the weights change every step).
//   wgrad  dW[tap][c][o] = sum over pixels X[p + tap][c] * dY[p][o]: a GEMM whose K dimension is the
//          pixel index.  In NHWC both operands have the CHANNEL contiguous, which is exactly the
//          operand shape of the exact-f32 MFMA v_mfma_f32_16x16x4_f32 (lane l holds row l&15 = channel,
//          k = l>>4 = one of 4 consecutive pixels): no transposes, fp32 accumulation, bf16 inputs are
//          widened on load.  Pixels are split over workgroups; partial sums go to a scratch slab per
//          split and a second kernel folds them in a fixed order (deterministic, no float atomics).
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
__global__ __launch_bounds__(TR_THREADS) void pack_weight_kernel(int ks, int transposed, const float* w, int cout, int cin, int cout_st,
                                                                 int cin_st, int split, int gap, T* dst) {
  constexpr int VEC = Traits<T>::VEC, KC = Traits<T>::KC;
  const int ntap = ks * ks;
  const int nchunk = cdiv(cin_st, KC);
  const size_t total = (size_t)nchunk * 4 * ntap * cout_st * VEC;
  for (size_t u = (size_t)blockIdx.x * TR_THREADS + threadIdx.x; u < total; u += (size_t)gridDim.x * TR_THREADS) {
    size_t r = u;
    const int i = (int)(r % VEC); r /= VEC;
    const int o = (int)(r % cout_st); r /= cout_st;
    const int tap = (int)(r % ntap); r /= ntap;
    const int q = (int)(r % 4);
    const int chunk = (int)(r / 4);
    const int cs = chunk * KC + q * VEC + i;             // stored input channel
    int c = -1;                                          // real input channel, -1 = padding
    if (cs < split) c = cs;
    else if (cs >= split + gap) c = cs - gap;
    float v = 0.f;
    if (o < cout && c >= 0 && c < cin) {
      const int kh = tap / ks, kw = tap - kh * ks;
      v = transposed ? w[(((size_t)c * cout + o) * ks + (ks - 1 - kh)) * ks + (ks - 1 - kw)]
                     : w[(((size_t)o * cin + c) * ks + kh) * ks + kw];
    }
    st(dst + u, v);
  }
}

// ---- wgrad ------------------------------------------------------------------------------------------------------------
struct WgradArgs {
  int B, H, W, ks;
  int nseg;
  SegT seg[MDIE_MAX_SEG];
  int cin_st, cout_st;
  const char* dy; int dy_stride;
  float* scratch;          // [splits][taps][cin_st][cout_st]
  int splits, groups_per_split;   // pixel groups of 4
};

__device__ __forceinline__ float ldf(const float* p) { return *p; }
__device__ __forceinline__ float ldf(const bf16* p) { return (float)*p; }

// grid (o_tiles * c_tiles, taps, splits); workgroup = 64 stored input channels x (16*NOS) output channels of one tap
template <typename T, int NOS>
__global__ __launch_bounds__(TR_THREADS) void wgrad_kernel(const WgradArgs a) {
  __shared__ float red[TR_THREADS / 64 - 1][4][NOS][64][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lq = lane >> 4, lp = lane & 15;
  const int o_tiles = a.cout_st / (16 * NOS);
  const int ot = blockIdx.x % o_tiles, ct = blockIdx.x / o_tiles;
  const int tap = blockIdx.y, split = blockIdx.z;
  const int pad = a.ks / 2;
  const int dyk = tap / a.ks - pad, dxk = tap % a.ks - pad;
  const int c0 = ct * 64, o0 = ot * 16 * NOS;

  // channel group cs of this lane: segment base pointer (or null beyond cin_st)
  const T* xb[4];
  int xs[4];
#pragma unroll
  for (int cs = 0; cs < 4; ++cs) {
    const int c = c0 + cs * 16 + lp;
    xb[cs] = nullptr; xs[cs] = 0;
#pragma unroll
    for (int s = 0; s < MDIE_MAX_SEG; ++s)
      if (s < a.nseg && c >= a.seg[s].ch_begin && c < a.seg[s].ch_end) {
        xb[cs] = reinterpret_cast<const T*>(a.seg[s].ptr) + (c - a.seg[s].ch_begin);
        xs[cs] = a.seg[s].stride;
      }
  }
  const T* dyb = reinterpret_cast<const T*>(a.dy) + o0 + lp;

  f32x4 acc[4][NOS];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NOS; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int npix = a.B * a.H * a.W;
  const int g_begin = split * a.groups_per_split, g_end = min(g_begin + a.groups_per_split, (npix + 3) / 4);
  for (int g = g_begin + wave; g < g_end; g += TR_THREADS / 64) {
    const int p = g * 4 + lq;                  // this lane's pixel (k index of the MFMA)
    float av[4], bv[NOS];
#pragma unroll
    for (int cs = 0; cs < 4; ++cs) av[cs] = 0.f;
#pragma unroll
    for (int os = 0; os < NOS; ++os) bv[os] = 0.f;
    if (p < npix) {
      const int img = p / (a.H * a.W);
      const int rem = p - img * a.H * a.W;
      const int y = rem / a.W, x = rem - y * a.W;
      const int yy = y + dyk, xx = x + dxk;
#pragma unroll
      for (int os = 0; os < NOS; ++os) bv[os] = ldf(dyb + (size_t)p * a.dy_stride + os * 16);
      if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) {
        const size_t q = ((size_t)img * a.H + yy) * a.W + xx;
#pragma unroll
        for (int cs = 0; cs < 4; ++cs)
          if (xb[cs]) av[cs] = ldf(xb[cs] + q * xs[cs]);
      }
    }
#pragma unroll
    for (int cs = 0; cs < 4; ++cs)
#pragma unroll
      for (int os = 0; os < NOS; ++os)
        acc[cs][os] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cs], bv[os], acc[cs][os], 0, 0, 0);
  }
  // fold the 4 waves (fixed order), then write this split's slab
  if (wave > 0) {
#pragma unroll
    for (int cs = 0; cs < 4; ++cs)
#pragma unroll
      for (int os = 0; os < NOS; ++os)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave - 1][cs][os][lane][r] = acc[cs][os][r];
  }
  __syncthreads();
  if (wave == 0) {
    float* out = a.scratch + (((size_t)split * gridDim.y + tap) * a.cin_st) * a.cout_st;
#pragma unroll
    for (int cs = 0; cs < 4; ++cs)
#pragma unroll
      for (int os = 0; os < NOS; ++os)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[cs][os][r];
          for (int w = 0; w < TR_THREADS / 64 - 1; ++w) v += red[w][cs][os][lane][r];
          // D[row = 4*lq + r = channel within the 16-group][col = lp = output channel]
          const int c = c0 + cs * 16 + lq * 4 + r, o = o0 + os * 16 + lp;
          if (c < a.cin_st) out[(size_t)c * a.cout_st + o] = v;
        }
  }
}

// fold the splits and scatter into PyTorch's layout (dropping padded channels):
//   transposed = 0: dw[o][c][kh][kw]             (nn.Conv2d)
//   transposed = 1: dw[c][o][ks-1-kh][ks-1-kw]   (nn.ConvTranspose2d run as a flipped convolution)
__global__ __launch_bounds__(TR_THREADS) void wgrad_reduce_kernel(int splits, int ks, int transposed, int cout, int cin, int cout_st, int cin_st,
                                                                  int split_c, int gap, const float* scratch, float* dw) {
  const int ntap = ks * ks;
  const size_t total = (size_t)cout * cin * ntap;
  const size_t slab = (size_t)ntap * cin_st * cout_st;
  for (size_t u = (size_t)blockIdx.x * TR_THREADS + threadIdx.x; u < total; u += (size_t)gridDim.x * TR_THREADS) {
    size_t r = u;
    const int tap = (int)(r % ntap); r /= ntap;
    int o, c;
    if (transposed) { o = (int)(r % cout); c = (int)(r / cout); }
    else { c = (int)(r % cin); o = (int)(r / cin); }
    const int kh = tap / ks, kw = tap - kh * ks;
    const int stap = transposed ? (ks - 1 - kh) * ks + (ks - 1 - kw) : tap;   // tap of the convolution that was run
    const int cs = c + (c >= split_c ? gap : 0);
    const float* p = scratch + ((size_t)stap * cin_st + cs) * cout_st + o;
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += p[(size_t)k * slab];
    dw[u] = s;
  }
}

static int tr_grid(size_t total) {
  size_t g = (total + TR_THREADS - 1) / TR_THREADS;
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace mdie

using namespace mdie;

extern "C" int mdie_pack_conv_weight_dev(int dtype, int ksize, int transposed, const float* w, int cout, int cin, int cout_stored,
                                         int cin_stored, int split, int gap, void* dst, void* stream) {
  MDIE_REQUIRE(dtype == MDIE_F32 || dtype == MDIE_BF16, "mdie_pack_conv_weight_dev: bad dtype %d", dtype);
  MDIE_REQUIRE(ksize == 1 || ksize == 3, "mdie_pack_conv_weight_dev: ksize %d", ksize);
  MDIE_REQUIRE(w && dst && cout > 0 && cin > 0, "mdie_pack_conv_weight_dev: null/empty");
  MDIE_REQUIRE(cout_stored >= cout && cout_stored % 16 == 0, "mdie_pack_conv_weight_dev: cout_stored %d", cout_stored);
  MDIE_REQUIRE(cin_stored % 16 == 0 && cin_stored >= cin + (split < cin ? gap : 0) && gap >= 0 && split >= 0,
               "mdie_pack_conv_weight_dev: cin_stored %d too small for cin %d split %d gap %d", cin_stored, cin, split, gap);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int kc = dtype == MDIE_F32 ? 16 : 32;
  const size_t total = (size_t)cdiv(cin_stored, kc) * 4 * ksize * ksize * cout_stored * (kc / 4);
  if (dtype == MDIE_F32)
    hipLaunchKernelGGL((pack_weight_kernel<float>), dim3(tr_grid(total)), dim3(TR_THREADS), 0, s, ksize, transposed, w, cout, cin, cout_stored, cin_stored,
                       split, gap, reinterpret_cast<float*>(dst));
  else
    hipLaunchKernelGGL((pack_weight_kernel<mdie::bf16>), dim3(tr_grid(total)), dim3(TR_THREADS), 0, s, ksize, transposed, w, cout, cin, cout_stored,
                       cin_stored, split, gap, reinterpret_cast<mdie::bf16*>(dst));
  MDIE_LAUNCH_CHECK("mdie_pack_conv_weight_dev");
  return MDIE_OK;
}

static int wgrad_splits(int B, int H, int W, int base_wgs) {
  const long groups = ((long)B * H * W + 3) / 4;
  long s = 2048 / (base_wgs > 0 ? base_wgs : 1);      // aim for ~2k workgroups
  if (s < 1) s = 1;
  if (s > 256) s = 256;
  if (s > groups / 16) s = groups / 16 > 0 ? groups / 16 : 1;   // at least 64 pixels per wave-quartet
  return (int)s;
}

extern "C" size_t mdie_conv_wgrad_workspace_bytes(int B, int H, int W, int ksize, int cin_stored, int cout_stored) {
  if (B <= 0 || H <= 0 || W <= 0 || cin_stored <= 0 || cout_stored <= 0) return 0;
  const int nos = cout_stored % 64 == 0 ? 4 : 1;
  const int base = (cout_stored / (16 * nos)) * cdiv(cin_stored, 64) * ksize * ksize;
  return (size_t)wgrad_splits(B, H, W, base) * ksize * ksize * cin_stored * cout_stored * sizeof(float);
}

extern "C" int mdie_conv_wgrad(const mdie_wgrad_desc* d, void* stream) {
  MDIE_REQUIRE(d != nullptr, "mdie_conv_wgrad: null descriptor");
  MDIE_REQUIRE(d->dtype == MDIE_F32 || d->dtype == MDIE_BF16, "mdie_conv_wgrad: bad dtype %d", d->dtype);
  MDIE_REQUIRE(d->ksize == 3 || d->ksize == 1, "mdie_conv_wgrad: ksize %d", d->ksize);
  MDIE_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0, "mdie_conv_wgrad: empty extent");
  MDIE_REQUIRE(d->nseg >= 1 && d->nseg <= MDIE_MAX_SEG, "mdie_conv_wgrad: nseg %d", d->nseg);
  MDIE_REQUIRE(d->dy && d->dw && d->workspace && d->cout > 0 && d->cin > 0, "mdie_conv_wgrad: null/empty");
  MDIE_REQUIRE(d->cout_stored % 16 == 0 && d->cout_stored >= d->cout && d->dy_stride >= d->cout_stored, "mdie_conv_wgrad: cout_stored %d", d->cout_stored);
  WgradArgs a{};
  a.B = d->B; a.H = d->H; a.W = d->W; a.ks = d->ksize;
  a.nseg = d->nseg;
  int c = 0;
  for (int s = 0; s < d->nseg; ++s) {
    MDIE_REQUIRE(d->in[s].ptr && d->in[s].channels > 0 && d->in[s].channels % 16 == 0, "mdie_conv_wgrad: segment %d channels %d", s, d->in[s].channels);
    a.seg[s].ptr = reinterpret_cast<const char*>(d->in[s].ptr);
    a.seg[s].ch_begin = c; c += d->in[s].channels; a.seg[s].ch_end = c;
    a.seg[s].stride = d->in[s].stride;
  }
  a.cin_st = c; a.cout_st = d->cout_stored;
  MDIE_REQUIRE(c >= d->cin + (d->split < d->cin ? d->gap : 0), "mdie_conv_wgrad: segments hold %d channels < cin %d + gap", c, d->cin);
  a.dy = reinterpret_cast<const char*>(d->dy); a.dy_stride = d->dy_stride;
  const int nos = d->cout_stored % 64 == 0 ? 4 : 1;
  const int o_tiles = d->cout_stored / (16 * nos), c_tiles = cdiv(c, 64), taps = d->ksize * d->ksize;
  a.splits = wgrad_splits(d->B, d->H, d->W, o_tiles * c_tiles * taps);
  const long groups = ((long)d->B * d->H * d->W + 3) / 4;
  a.groups_per_split = (int)((groups + a.splits - 1) / a.splits);
  const size_t need = (size_t)a.splits * taps * c * d->cout_stored * sizeof(float);
  if (d->workspace_bytes < need) { set_error("mdie_conv_wgrad: workspace %zu < %zu", d->workspace_bytes, need); return MDIE_ENOSPC; }
  a.scratch = reinterpret_cast<float*>(d->workspace);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid(o_tiles * c_tiles, taps, a.splits);
  if (d->dtype == MDIE_F32) {
    if (nos == 4) hipLaunchKernelGGL((wgrad_kernel<float, 4>), grid, dim3(TR_THREADS), 0, s, a);
    else hipLaunchKernelGGL((wgrad_kernel<float, 1>), grid, dim3(TR_THREADS), 0, s, a);
  } else {
    if (nos == 4) hipLaunchKernelGGL((wgrad_kernel<mdie::bf16, 4>), grid, dim3(TR_THREADS), 0, s, a);
    else hipLaunchKernelGGL((wgrad_kernel<mdie::bf16, 1>), grid, dim3(TR_THREADS), 0, s, a);
  }
  MDIE_LAUNCH_CHECK("mdie_conv_wgrad");
  const size_t total = (size_t)d->cout * d->cin * taps;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(tr_grid(total)), dim3(TR_THREADS), 0, s, a.splits, d->ksize, d->transposed, d->cout, d->cin, d->cout_stored, c,
                     d->split, d->gap, a.scratch, d->dw);
  MDIE_LAUNCH_CHECK("mdie_conv_wgrad");
  return MDIE_OK;
}
