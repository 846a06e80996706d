// CBAM (models/cbam.py:84-95) on NHWC tensors, four HBM-bound passes:
//   pool      per-(image, channel) sum / max over H*W          ChannelGate pools, cbam.py:41,44
//   gate      sigmoid(MLP(avg) + MLP(max))                     cbam.py:30-35,42-59  (folded into chanpool's prologue)
//   chanpool  per-pixel max / mean over channels of x*gate     ChannelPool, cbam.py:68-70
//   spatial   sigmoid(BN(conv7x7(map))) ; out = x*gate*s [*mul]  SpatialGate, cbam.py:72-82
// The channel-scaled tensor x*gate is never written: passes 3 and 4 recompute it from x.
// NHWC makes the channel reductions contiguous (one 16-byte vector per lane, wave shuffles to
// finish), and the global pools are deterministic two-level reductions (no float atomics).
#include "common.hpp"

namespace mdie {

constexpr int CB_THREADS = 256;
constexpr int POOL_MIN_SLAB = 128;  // pixels per pool block (at least)
// partials per image the gate has to fold: a function of the resolution only (never of B: an image's fp32 summation
// order, hence its output bits, must not depend on its batch)
static int pool_max_slabs(int H, int W) { return (long)H * W >= 16384 ? 64 : 16; }

struct CbamArgs {
  int B, H, W, C;
  const char* x; int x_stride;
  const float *w1, *b1, *w2, *b2, *w7;
  const float* bn;
  const char* mul; int mul_stride;
  char* out; int out_stride;
  float* partial;  // [B][nslab][2][C]
  float* gate;     // [B][C]
  float* map;      // [B][H][W][2]  (max, mean)
  int nslab, slab;  // pool blocks per image, pixels per block
  int spatial;     // 0: channel gate only
};

// ---- pass 1 ----------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(CB_THREADS) void cbam_pool_kernel(const CbamArgs a) {
  constexpr int VEC = Traits<T>::VEC;
  extern __shared__ __attribute__((aligned(16))) char dyn[];
  const int CV = a.C / VEC;            // 16-byte vectors per pixel (power of two, <= 128)
  const int rows = CB_THREADS / CV;    // pixels in flight per block
  float* rsum = reinterpret_cast<float*>(dyn);  // [rows][C]
  float* rmax = rsum + rows * a.C;              // [rows][C]
  const int tid = threadIdx.x;
  const int slab = blockIdx.x, img = blockIdx.y;
  const int npix = a.H * a.W;
  const int p_begin = slab * a.slab;
  const int p_end = min(npix, p_begin + a.slab);
  const int v = tid % CV, r = tid / CV;
  float s[VEC], m[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { s[i] = 0.f; m[i] = -INFINITY; }
  for (int p = p_begin + r; p < p_end; p += rows) {
    const uint4 u = *reinterpret_cast<const uint4*>(a.x + ((size_t)img * npix + p) * a.x_stride * sizeof(T) + (size_t)v * 16);
    float f[VEC];
    Vec16<T>::unpack(u, f);
#pragma unroll
    for (int i = 0; i < VEC; ++i) { s[i] += f[i]; m[i] = fmaxf(m[i], f[i]); }
  }
#pragma unroll
  for (int i = 0; i < VEC; ++i) { rsum[r * a.C + v * VEC + i] = s[i]; rmax[r * a.C + v * VEC + i] = m[i]; }
  __syncthreads();
  float* dst = a.partial + ((size_t)img * a.nslab + slab) * 2 * a.C;
  for (int c = tid; c < a.C; c += CB_THREADS) {
    float ss = 0.f, mm = -INFINITY;
    for (int k = 0; k < rows; ++k) { ss += rsum[k * a.C + c]; mm = fmaxf(mm, rmax[k * a.C + c]); }
    dst[c] = ss;
    dst[a.C + c] = mm;
  }
}

// ---- pass 2 ----------------------------------------------------------------------------------------
// One block per image.  The hidden layer is split over (unit j, part q): 256/Hd parts per unit, each
// part a contiguous run of C/parts channels for BOTH pooled vectors, so all weight loads of the block
// are issued at once (a per-output loop serialises 2*Hd cold-miss round trips: 42 us at C=512).
// gate[c] of image `img` into LDS (`gate`, [C]); scratch: avg[C], mx[C], part[2*CB_THREADS], hid[Hd]
__device__ __forceinline__ void cbam_gate_block(const CbamArgs& a, int img, float* avg, float* mx, float* part, float* hid, float* gate) {
  const int Hd = a.C / 16;
  const int tid = threadIdx.x;
  const float inv = 1.0f / (float)(a.H * a.W);
  for (int c = tid; c < a.C; c += CB_THREADS) {
    float s = 0.f, m = -INFINITY;
    for (int k = 0; k < a.nslab; ++k) {
      const float* p = a.partial + ((size_t)img * a.nslab + k) * 2 * a.C;
      s += p[c];
      m = fmaxf(m, p[a.C + c]);
    }
    avg[c] = s * inv;
    mx[c] = m;
  }
  __syncthreads();
  const int parts = CB_THREADS / Hd;           // 8 (C=512) .. 256 (C=16)
  const int run = a.C / parts > 0 ? a.C / parts : 1;  // channels per part
  {
    const int j = tid / parts, q = tid - j * parts;
    float sa = 0.f, sm = 0.f;
    if (j < Hd) {
      const int c0 = q * run;
      if (c0 < a.C) {
        const float* w = a.w1 + (size_t)j * a.C + c0;
        for (int i = 0; i < run; ++i) { const float wv = w[i]; sa = fmaf(wv, avg[c0 + i], sa); sm = fmaf(wv, mx[c0 + i], sm); }
      }
    }
    part[tid] = sa;
    part[CB_THREADS + tid] = sm;
  }
  __syncthreads();
  if (tid < Hd) {
    float sa = 0.f, sm = 0.f;
    for (int q = 0; q < parts; ++q) { sa += part[tid * parts + q]; sm += part[CB_THREADS + tid * parts + q]; }
    const float b = a.b1[tid];
    hid[tid] = fmaxf(sa + b, 0.f) + fmaxf(sm + b, 0.f);
  }
  __syncthreads();
  for (int c = tid; c < a.C; c += CB_THREADS) {
    float s = 2.0f * a.b2[c];  // the MLP (bias included) is applied to both pooled vectors
    const float* w = a.w2 + (size_t)c * Hd;
    for (int j = 0; j < Hd; ++j) s = fmaf(w[j], hid[j], s);
    gate[c] = sigmoidf(s);
  }
  __syncthreads();
}

// standalone gate launch: only the channel-gate-only path (CBAM(no_spatial=True)) still uses it
__global__ __launch_bounds__(CB_THREADS) void cbam_gate_kernel(const CbamArgs a) {
  extern __shared__ __attribute__((aligned(16))) char dyn[];
  float* avg = reinterpret_cast<float*>(dyn);
  float* mx = avg + a.C;
  float* part = mx + a.C;
  float* hid = part + 2 * CB_THREADS;
  float* gate = hid + a.C / 16;
  cbam_gate_block(a, blockIdx.x, avg, mx, part, hid, gate);
  for (int c = threadIdx.x; c < a.C; c += CB_THREADS) a.gate[(size_t)blockIdx.x * a.C + c] = gate[c];
}

// ---- pass 3 ----------------------------------------------------------------------------------------
// LPP lanes share one pixel (LPP = min(C/VEC, 64), a power of two); each lane owns NV vectors.
template <typename T, int NV>
__global__ __launch_bounds__(CB_THREADS) void cbam_chanpool_kernel(const CbamArgs a, const int LPP) {
  constexpr int VEC = Traits<T>::VEC;
  const int tid = threadIdx.x;
  const int img = blockIdx.y;
  const int npix = a.H * a.W;
  const int sub = tid % LPP;                 // lane within the pixel group
  const int groups = CB_THREADS / LPP;       // pixels in flight per block
  // pass 2 folded in: every block derives the image's channel gate from the pooled partials (a few k MACs,
  // weights L2-resident) instead of waiting for a separate 32-block launch; block 0 publishes it for pass 4
  extern __shared__ __attribute__((aligned(16))) char dyn[];
  float* avg = reinterpret_cast<float*>(dyn);
  float* mx = avg + a.C;
  float* part = mx + a.C;
  float* hid = part + 2 * CB_THREADS;
  float* gsh = hid + a.C / 16;
  cbam_gate_block(a, img, avg, mx, part, hid, gsh);
  if (blockIdx.x == 0)
    for (int c = tid; c < a.C; c += CB_THREADS) a.gate[(size_t)img * a.C + c] = gsh[c];
  float g[NV][VEC];
#pragma unroll
  for (int k = 0; k < NV; ++k)
#pragma unroll
    for (int i = 0; i < VEC; ++i) g[k][i] = gsh[(k * LPP + sub) * VEC + i];
  const float invC = 1.0f / (float)a.C;
  for (int p = blockIdx.x * groups + tid / LPP; p < npix; p += gridDim.x * groups) {
    const char* px = a.x + ((size_t)img * npix + p) * a.x_stride * sizeof(T);
    float m = -INFINITY, s = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const uint4 u = *reinterpret_cast<const uint4*>(px + (size_t)(k * LPP + sub) * 16);
      float f[VEC];
      Vec16<T>::unpack(u, f);
#pragma unroll
      for (int i = 0; i < VEC; ++i) { const float y = f[i] * g[k][i]; m = fmaxf(m, y); s += y; }
    }
    for (int d = LPP >> 1; d > 0; d >>= 1) { m = fmaxf(m, __shfl_xor(m, d)); s += __shfl_xor(s, d); }
    if (sub == 0) *reinterpret_cast<float2*>(a.map + ((size_t)img * npix + p) * 2) = make_float2(m, s * invC);
  }
}

// ---- pass 4 ----------------------------------------------------------------------------------------
template <typename T, int TS>
__global__ __launch_bounds__(CB_THREADS) void cbam_spatial_kernel(const CbamArgs a) {
  constexpr int VEC = Traits<T>::VEC;
  constexpr int PW = TS + 6;
  extern __shared__ __attribute__((aligned(16))) char dyn[];
  float* gate = reinterpret_cast<float*>(dyn);     // [C]
  float* patch = gate + a.C;                       // [2][PW][PW]
  float* sg = patch + 2 * PW * PW;                 // [TS*TS]
  float* w7 = sg + TS * TS;                        // [98]
  const int tid = threadIdx.x;
  const int tiles_x = cdiv(a.W, TS);
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x, img = blockIdx.y;
  const int y0 = ty * TS, x0 = tx * TS;
  for (int c = tid; c < a.C; c += CB_THREADS) gate[c] = a.gate[(size_t)img * a.C + c];
  if (a.spatial) {
    for (int i = tid; i < PW * PW; i += CB_THREADS) {
      const int py = i / PW, px = i - py * PW;
      const int gy = y0 + py - 3, gx = x0 + px - 3;
      float2 v = make_float2(0.f, 0.f);
      if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
        v = *reinterpret_cast<const float2*>(a.map + (((size_t)img * a.H + gy) * a.W + gx) * 2);
      patch[i] = v.x;
      patch[PW * PW + i] = v.y;
    }
    if (tid < 98) w7[tid] = a.w7[tid];
  }
  __syncthreads();
  if (tid < TS * TS) {
    float s = 1.0f;
    if (a.spatial) {
      const int py = tid / TS, px = tid % TS;
      float acc = 0.f;
#pragma unroll
      for (int ch = 0; ch < 2; ++ch)
#pragma unroll
        for (int kh = 0; kh < 7; ++kh)
#pragma unroll
          for (int kw = 0; kw < 7; ++kw)
            acc = fmaf(w7[(ch * 7 + kh) * 7 + kw], patch[ch * PW * PW + (py + kh) * PW + px + kw], acc);
      s = sigmoidf(fmaf(acc, a.bn[0], a.bn[1]));
    }
    sg[tid] = s;
  }
  __syncthreads();
  const int CV = a.C / VEC;
  const int total = TS * TS * CV;
  for (int u = tid; u < total; u += CB_THREADS) {
    const int pix = u / CV, v = u - pix * CV;
    const int gy = y0 + pix / TS, gx = x0 + pix % TS;
    if (gy >= a.H || gx >= a.W) continue;
    const size_t gp = ((size_t)img * a.H + gy) * a.W + gx;
    const uint4 xv = *reinterpret_cast<const uint4*>(a.x + gp * a.x_stride * sizeof(T) + (size_t)v * 16);
    float f[VEC];
    Vec16<T>::unpack(xv, f);
    const float s = sg[pix];
#pragma unroll
    for (int i = 0; i < VEC; ++i) f[i] = f[i] * gate[v * VEC + i] * s;
    if (a.mul) {
      const uint4 mv = *reinterpret_cast<const uint4*>(a.mul + gp * a.mul_stride * sizeof(T) + (size_t)v * 16);
      float m[VEC];
      Vec16<T>::unpack(mv, m);
#pragma unroll
      for (int i = 0; i < VEC; ++i) f[i] *= m[i];
    }
    *reinterpret_cast<uint4*>(a.out + gp * a.out_stride * sizeof(T) + (size_t)v * 16) = Vec16<T>::pack(f);
  }
}

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

static int nslab_for(int H, int W) { const int n = cdiv(H * W, POOL_MIN_SLAB), m = pool_max_slabs(H, W); return n < m ? n : m; }

template <typename T>
static int run_cbam(const mdie_cbam_desc* d, bool spatial, hipStream_t stream) {
  constexpr int VEC = Traits<T>::VEC;
  CbamArgs a{};
  a.B = d->B; a.H = d->H; a.W = d->W; a.C = d->C;
  a.x = reinterpret_cast<const char*>(d->x); a.x_stride = d->x_stride;
  a.w1 = d->w1; a.b1 = d->b1; a.w2 = d->w2; a.b2 = d->b2; a.w7 = d->w7;
  a.bn = d->bn;
  a.mul = reinterpret_cast<const char*>(d->mul); a.mul_stride = d->mul_stride;
  a.out = reinterpret_cast<char*>(d->out); a.out_stride = d->out_stride;
  a.nslab = nslab_for(d->H, d->W);
  a.slab = cdiv(d->H * d->W, a.nslab);
  a.spatial = spatial ? 1 : 0;
  char* ws = reinterpret_cast<char*>(d->workspace);
  a.partial = reinterpret_cast<float*>(ws);
  ws += align256((size_t)d->B * a.nslab * 2 * d->C * sizeof(float));
  a.gate = reinterpret_cast<float*>(ws);
  ws += align256((size_t)d->B * d->C * sizeof(float));
  a.map = reinterpret_cast<float*>(ws);

  const int CV = d->C / VEC;
  const size_t gate_lds = (size_t)(3 * d->C + 2 * CB_THREADS + d->C / 16) * sizeof(float);
  if (d->pool_partial) {
    // the producer of x already reduced it (mdie_upsample2x_add with pool_partial): skip pass 1
    a.partial = const_cast<float*>(d->pool_partial);
    a.nslab = d->pool_slabs;
  } else {
    const int rows = CB_THREADS / CV;
    const size_t lds = (size_t)2 * rows * d->C * sizeof(float);
    TimedLaunch tl(MDIE_K_CBAM_POOL);
    hipLaunchKernelGGL((cbam_pool_kernel<T>), dim3(a.nslab, d->B), dim3(CB_THREADS), lds, stream, a);
    MDIE_LAUNCH_CHECK("cbam_pool");
  }
  if (!spatial) {
    TimedLaunch tl(MDIE_K_CBAM_GATE);
    hipLaunchKernelGGL(cbam_gate_kernel, dim3(d->B), dim3(CB_THREADS), gate_lds, stream, a);
    MDIE_LAUNCH_CHECK("cbam_gate");
  }
  if (spatial) {
    const int LPP = CV < 64 ? CV : 64;
    const int NV = CV / LPP;
    const int groups = CB_THREADS / LPP;
    int gx = cdiv(d->H * d->W, groups * 4);
    // every block re-derives the gate: keep the block count per image moderate for the wide tensors, whose MLP
    // weights are 32-128 KB (L2 reads per block)
    int cap = d->C >= 256 ? 16 : 64;
    if (cap < 1024 / d->B) cap = 1024 / d->B;   // small batches: more blocks per image
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    TimedLaunch tl(MDIE_K_CBAM_CHANPOOL);
    if (NV == 1) hipLaunchKernelGGL((cbam_chanpool_kernel<T, 1>), dim3(gx, d->B), dim3(CB_THREADS), gate_lds, stream, a, LPP);
    else if (NV == 2) hipLaunchKernelGGL((cbam_chanpool_kernel<T, 2>), dim3(gx, d->B), dim3(CB_THREADS), gate_lds, stream, a, LPP);
    else { set_error("mdie_cbam_fwd: C = %d too wide", d->C); return MDIE_EINVAL; }
    MDIE_LAUNCH_CHECK("cbam_chanpool");
  }
  {
    const size_t lds = (size_t)(d->C + 2 * 22 * 22 + 256 + 100) * sizeof(float);
    // 1 KiB-per-pixel tensors (C >= 256) sit at 32x32 in this network: 16x16 tiles would give 128 blocks
    const int ts = d->C >= 256 ? 8 : 16;
    const int tiles = cdiv(d->W, ts) * cdiv(d->H, ts);
    TimedLaunch tl(MDIE_K_CBAM_SPATIAL);
    if (ts == 8) hipLaunchKernelGGL((cbam_spatial_kernel<T, 8>), dim3(tiles, d->B), dim3(CB_THREADS), lds, stream, a);
    else hipLaunchKernelGGL((cbam_spatial_kernel<T, 16>), dim3(tiles, d->B), dim3(CB_THREADS), lds, stream, a);
    MDIE_LAUNCH_CHECK("cbam_spatial");
  }
  return MDIE_OK;
}

static int check_cbam(const mdie_cbam_desc* d) {
  MDIE_REQUIRE(d != nullptr, "mdie_cbam_fwd: null descriptor");
  MDIE_REQUIRE(d->dtype == MDIE_F32 || d->dtype == MDIE_BF16, "mdie_cbam_fwd: bad dtype %d", d->dtype);
  MDIE_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0, "mdie_cbam_fwd: empty extent");
  MDIE_REQUIRE(d->C >= 16 && d->C <= 512 && (d->C & (d->C - 1)) == 0, "mdie_cbam_fwd: C = %d must be a power of two in [16, 512]", d->C);
  MDIE_REQUIRE(d->x && d->out && d->w1 && d->b1 && d->w2 && d->b2 && d->w7 && d->bn && d->workspace, "mdie_cbam_fwd: null pointer");
  MDIE_REQUIRE(d->x_stride % 16 == 0 && d->out_stride % 16 == 0 && (!d->mul || d->mul_stride % 16 == 0), "mdie_cbam_fwd: strides must be multiples of 16");
  MDIE_REQUIRE(((uintptr_t)d->x & 15) == 0 && ((uintptr_t)d->out & 15) == 0 && ((uintptr_t)d->mul & 15) == 0, "mdie_cbam_fwd: alignment");
  MDIE_REQUIRE(!d->pool_partial || (d->pool_slabs >= 1 && d->pool_slabs <= MDIE_POOL_SLABS_MAX), "mdie_cbam_fwd: pool_slabs %d", d->pool_slabs);
  if (d->workspace_bytes < mdie_cbam_workspace_bytes(d->B, d->H, d->W, d->C)) {
    set_error("mdie_cbam_fwd: workspace %zu < %zu", d->workspace_bytes, mdie_cbam_workspace_bytes(d->B, d->H, d->W, d->C));
    return MDIE_ENOSPC;
  }
  return MDIE_OK;
}

}  // namespace mdie

extern "C" size_t mdie_cbam_workspace_bytes(int B, int H, int W, int C) {
  using namespace mdie;
  return align256((size_t)B * nslab_for(H, W) * 2 * C * sizeof(float)) + align256((size_t)B * C * sizeof(float)) +
         align256((size_t)B * H * W * 2 * sizeof(float));
}

extern "C" int mdie_cbam_fwd(const mdie_cbam_desc* d, void* stream) {
  using namespace mdie;
  if (int e = check_cbam(d)) return e;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  return d->dtype == MDIE_F32 ? run_cbam<float>(d, true, s) : run_cbam<mdie::bf16>(d, true, s);
}

extern "C" int mdie_cbam_channel_only_fwd(const mdie_cbam_desc* d, void* stream) {
  using namespace mdie;
  if (int e = check_cbam(d)) return e;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  return d->dtype == MDIE_F32 ? run_cbam<float>(d, false, s) : run_cbam<mdie::bf16>(d, false, s);
}
