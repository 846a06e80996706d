// Training-mode glue around the convolutions: batch-statistic
// BatchNorm, ReLU, 2x2 max-pool, dropout, bilinear 2x upsampling + skip add, the final sigmoid -- forward and
// backward -- as a handful of fused bandwidth kernels over NHWC tensors.
//
// BatchNorm in training mode (nn.BatchNorm2d, models/cdan.py:12,43,50,105-116) splits into
//   stats     per-channel sum / sum of squares over B*H*W         (two-level, ordered: bit-reproducible)
//   fold      mean, biased var -> scale = gamma / sqrt(var + eps), shift = beta - mean * scale, invstd; running
//             statistics updated with momentum and the unbiased variance
//   apply     never on its own: the normalisation is either the PROLOGUE of the next convolution (dense layers:
//             relu(x * scale + shift) while staging, mdie_conv_fwd / mdie_conv_wgrad pre_scale) or fused with
//             ReLU + max-pool + dropout (encoder ConvBlocks) or ReLU + upsample + skip add (decoder stages).
// and its backward into
//   producer  the masked upstream gradient dz = da * [z > 0] (+ pool routing / dropout / transposed upsampling)
//             together with per-block partial sums of dz and dz * xhat
//   final     dgamma = sum dz * xhat, dbeta = sum dz, per-channel constants k2 = dbeta / N, k3 = dgamma / N
//   apply     dx = scale * (dz - k2 - xhat * k3), written or accumulated into the input's gradient
// A dense block's channel statistics are computed ONCE per segment (every later layer normalises the same
// tensor with the same batch statistics and only its own gamma / beta), instead of once per consuming layer.
#include "common.hpp"

namespace mdie {

constexpr int BN_THREADS = 256;
constexpr int BN_MAX_BLOCKS = 512;

struct SegP { const char* ptr; int ch_begin, ch_end, stride; };   // stride in elements
struct SegW { char* ptr; int ch_begin, ch_end, stride; };

// (channel vector, pixel row) decomposition of a block: thread t -> cv = t % CV, row = t / CV, rows = 256 / CV
struct BlkMap {
  int cv, row, rows;
  bool active;
};
__device__ __forceinline__ BlkMap blk_map(int CV) {
  BlkMap m;
  m.rows = BN_THREADS / CV;
  m.cv = threadIdx.x % CV;
  m.row = threadIdx.x / CV;
  m.active = m.row < m.rows;
  return m;
}

// fold the per-thread sums s[K][VEC] over the rows of the block and write partial[blk][k][c]
template <int VEC, int K>
__device__ __forceinline__ void block_fold(const float (&s)[K][VEC], const BlkMap& m, int CV, int C, float* lds, float* partial) {
  // lds[row][k][cv*VEC + i]
  if (m.active) {
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int i = 0; i < VEC; ++i) lds[(m.row * K + k) * C + m.cv * VEC + i] = s[k][i];
  }
  __syncthreads();
  for (int o = threadIdx.x; o < K * C; o += BN_THREADS) {
    float t = 0.f;
    for (int r = 0; r < m.rows; ++r) t += lds[r * K * C + o];
    partial[(size_t)blockIdx.x * K * C + o] = t;
  }
}

// ---- statistics ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(BN_THREADS) void bn_stats_kernel(long N, const char* x, int C, int stride, long chunk, float* partial) {
  constexpr int VEC = Traits<T>::VEC;
  extern __shared__ __attribute__((aligned(16))) char dyn[];
  const int CV = C / VEC;
  const BlkMap m = blk_map(CV);
  float s[2][VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) s[0][i] = s[1][i] = 0.f;
  if (m.active) {
    const long b = (long)blockIdx.x * chunk, e = min(N, b + chunk);
#pragma unroll 4
    for (long p = b + m.row; p < e; p += m.rows) {   // (unrolled: a read-only loop, 4 loads in flight per thread)
      float f[VEC];
      Vec16<T>::unpack(*reinterpret_cast<const uint4*>(x + ((size_t)p * stride) * sizeof(T) + (size_t)m.cv * 16), f);
#pragma unroll
      for (int i = 0; i < VEC; ++i) { s[0][i] += f[i]; s[1][i] = fmaf(f[i], f[i], s[1][i]); }
    }
  }
  block_fold<VEC, 2>(s, m, CV, C, reinterpret_cast<float*>(dyn), partial);
}

// one 64-lane block per channel: ordered double-precision fold of the block partials
__global__ __launch_bounds__(64) void bn_stats_final_kernel(int nblk, int C, double n, const float* partial, float* mean, float* var) {
  const int c = blockIdx.x;
  double s1 = 0.0, s2 = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 64) { s1 += partial[((size_t)b * 2 + 0) * C + c]; s2 += partial[((size_t)b * 2 + 1) * C + c]; }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) { s1 += __shfl_xor(s1, d); s2 += __shfl_xor(s2, d); }
  if (threadIdx.x == 0) {
    const double mu = s1 / n;
    mean[c] = (float)mu;
    var[c] = (float)fmax(s2 / n - mu * mu, 0.0);
  }
}

// stored channel cs -> real parameter channel (or -1 for padding): real c >= split sits at c + gap
__device__ __forceinline__ int real_channel(int cs, int split, int gap, int c_real) {
  int c = -1;
  if (cs < split) c = cs;
  else if (cs >= split + gap) c = cs - gap;
  return (c >= 0 && c < c_real) ? c : -1;
}

__global__ __launch_bounds__(BN_THREADS) void bn_fold_kernel(int C_st, int C_real, int split, int gap, const float* mean, const float* var, const float* gamma,
                                                             const float* beta, float eps, float momentum, double n, float* rmean, float* rvar, float* scale,
                                                             float* shift, float* invstd) {
  const int cs = blockIdx.x * BN_THREADS + threadIdx.x;
  if (cs >= C_st) return;
  const int c = real_channel(cs, split, gap, C_real);
  if (c < 0) { scale[cs] = 0.f; shift[cs] = 0.f; invstd[cs] = 0.f; return; }
  const float is = 1.0f / sqrtf(var[cs] + eps);
  const float sc = gamma[c] * is;
  scale[cs] = sc; shift[cs] = beta[c] - mean[cs] * sc; invstd[cs] = is;
  if (rmean) {
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean[cs];
    const float unbiased = n > 1.0 ? (float)(var[cs] * n / (n - 1.0)) : var[cs];
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * unbiased;
  }
}

// statistics final + fold in ONE launch (a training step has 28 BatchNorms: 28 launches of ~5 us less).  One 64-lane block per
// stored channel of the FOLD range; the blocks of the `C` channels whose partial sums have just been written (stored channels
// new_off .. new_off + C of the fold range) first fold them in double precision -- the same ordered sum as
// bn_stats_final_kernel -- and store mean / var; every block then folds its channel exactly as bn_fold_kernel does.
struct BnFinalFoldArgs {
  int nblk, C, new_off; double n;
  const float* partial; float* mean; float* var;
  int C_fold, C_real, split, gap;
  const float* fold_mean; const float* fold_var; const float* gamma; const float* beta; float eps, momentum;
  float* rmean; float* rvar; float* scale; float* shift; float* invstd;
};
__global__ __launch_bounds__(64) void bn_final_fold_kernel(const BnFinalFoldArgs a) {
  const int cs = blockIdx.x;
  const int cn = cs - a.new_off;
  float mu_f, var_f;
  if (cn >= 0 && cn < a.C) {
    double s1 = 0.0, s2 = 0.0;
    for (int b = threadIdx.x; b < a.nblk; b += 64) { s1 += a.partial[((size_t)b * 2 + 0) * a.C + cn]; s2 += a.partial[((size_t)b * 2 + 1) * a.C + cn]; }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { s1 += __shfl_xor(s1, d); s2 += __shfl_xor(s2, d); }
    const double mu = s1 / a.n;
    mu_f = (float)mu;
    var_f = (float)fmax(s2 / a.n - mu * mu, 0.0);
    if (threadIdx.x == 0) { a.mean[cn] = mu_f; a.var[cn] = var_f; }
  } else {
    if (cs >= a.C_fold) return;
    mu_f = a.fold_mean[cs]; var_f = a.fold_var[cs];
  }
  if (threadIdx.x != 0 || cs >= a.C_fold || a.gamma == nullptr) return;
  const int c = real_channel(cs, a.split, a.gap, a.C_real);
  if (c < 0) { a.scale[cs] = 0.f; a.shift[cs] = 0.f; a.invstd[cs] = 0.f; return; }
  const float is = 1.0f / sqrtf(var_f + a.eps);
  const float sc = a.gamma[c] * is;
  a.scale[cs] = sc; a.shift[cs] = a.beta[c] - mu_f * sc; a.invstd[cs] = is;
  if (a.rmean) {
    a.rmean[c] = (1.f - a.momentum) * a.rmean[c] + a.momentum * mu_f;
    const float unbiased = a.n > 1.0 ? (float)(var_f * a.n / (a.n - 1.0)) : var_f;
    a.rvar[c] = (1.f - a.momentum) * a.rvar[c] + a.momentum * unbiased;
  }
}

// ---- dropout: counter-based, recomputed in backward from (seed, element index) ----------------------------------------
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
// seed of a launch: the caller's host value, mixed with a counter in DEVICE memory when one is given -- a captured training
// step (hipGraph replay) freezes every host argument, the counter is bumped by a captured kernel, so each replay drops
// other elements while forward and backward of one step still see the same mask
__device__ __forceinline__ uint32_t launch_seed(uint32_t seed, const uint32_t* seed_dev) {
  return seed_dev ? seed ^ mix32(*seed_dev + 0x9e3779b9U) : seed;
}
// multiplier applied to element idx: 0 (dropped) or 1 / (1 - p)
__device__ __forceinline__ float drop_factor(uint32_t seed, size_t idx, float p, float keep_scale) {
  const uint32_t h = mix32((uint32_t)idx ^ mix32(seed ^ (uint32_t)(idx >> 32) * 0x9e3779b9U));
  return ((h >> 8) * (1.0f / 16777216.0f)) >= p ? keep_scale : 0.f;
}

// ---- forward: BN + ReLU (+ 2x2 max-pool) (+ dropout) ---------------------------------------------------------------------
// out_o = pool?(relu(y * scale + shift)), out_t = dropout(out_o); either output may be null
template <typename T, bool POOL>
__global__ __launch_bounds__(BN_THREADS) void bn_act_pool_fwd_kernel(int B, int H, int W, int C, const char* y, int y_stride, const float* scale,
                                                                     const float* shift, char* out_o, int o_stride, char* out_t, int t_stride, float p,
                                                                     uint32_t seed_host, const uint32_t* seed_dev) {
  constexpr int VEC = Traits<T>::VEC;
  const uint32_t seed = launch_seed(seed_host, seed_dev);
  const int CV = C / VEC;
  const int Ho = POOL ? H / 2 : H, Wo = POOL ? W / 2 : W;
  const size_t total = (size_t)B * Ho * Wo * CV;
  const float ks = 1.0f / (1.0f - p);
  for (size_t u = (size_t)blockIdx.x * BN_THREADS + threadIdx.x; u < total; u += (size_t)gridDim.x * BN_THREADS) {
    const int v = (int)(u % CV);
    const size_t op = u / CV;                       // output pixel index
    float sc[VEC], sh[VEC], r[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) { sc[i] = scale[v * VEC + i]; sh[i] = shift[v * VEC + i]; }
    if (POOL) {
      const int ox = (int)(op % Wo);
      const size_t q = op / Wo;
      const int oy = (int)(q % Ho), img = (int)(q / Ho);
      const size_t ip = ((size_t)img * H + 2 * oy) * W + 2 * ox;
#pragma unroll
      for (int i = 0; i < VEC; ++i) r[i] = 0.f;      // relu folded into the running maximum
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float f[VEC];
        Vec16<T>::unpack(*reinterpret_cast<const uint4*>(y + (ip + (k >> 1) * W + (k & 1)) * y_stride * sizeof(T) + (size_t)v * 16), f);
#pragma unroll
        for (int i = 0; i < VEC; ++i) r[i] = fmaxf(r[i], fmaf(f[i], sc[i], sh[i]));
      }
    } else {
      float f[VEC];
      Vec16<T>::unpack(*reinterpret_cast<const uint4*>(y + op * y_stride * sizeof(T) + (size_t)v * 16), f);
#pragma unroll
      for (int i = 0; i < VEC; ++i) r[i] = fmaxf(fmaf(f[i], sc[i], sh[i]), 0.f);
    }
    if (out_o) *reinterpret_cast<uint4*>(out_o + op * o_stride * sizeof(T) + (size_t)v * 16) = Vec16<T>::pack(r);
    if (out_t) {
      if (p > 0.f) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) r[i] *= drop_factor(seed, op * C + v * VEC + i, p, ks);
      }
      *reinterpret_cast<uint4*>(out_t + op * t_stride * sizeof(T) + (size_t)v * 16) = Vec16<T>::pack(r);
    }
  }
}

// ---- backward producer of the same site ------------------------------------------------------------------------------------
// g = d_o + dropout_mask * d_t (at the output resolution); dz = g routed to the arg-max of the 2x2 window (first maximum in
// scan order, like ATen) and masked by z > 0; writes dz at the input resolution and the block's partial sums.
struct BnBwdPoolArgs {
  int B, H, W, C;
  const char* y; int y_stride;
  const float *scale, *shift, *mean, *invstd;
  const char* d_o; int do_stride;
  const char* d_t; int dt_stride;
  float p; uint32_t seed; const uint32_t* seed_dev;
  char* dz; int dz_stride;
  float* partial;
  long chunk;    // output pixels per block
  const float* coef;   // PASS 2: [2][C] k2 = dbeta / N, k3 = dgamma / N
};

// PASS 0: dz and the partial sums (the caller applies mdie_bn_bwd_apply to dz afterwards: dz is written, read and written again).
// PASS 1 / PASS 2 -- the two-pass form (mdie_bn_pool_bwd_desc.two_pass): 1 = the partial sums only, nothing stored; 2 = dz is
// formed again from the same inputs and dL/dy = scale * (dz - k2 - xhat * k3) is stored directly: y and the (pooled: a quarter
// of the size) upstream gradients are read twice, the full-resolution gradient is written ONCE and never read
// (5.5 -> 4 full-tensor passes; and dz is never rounded to the storage type on its way into the formula).
template <typename T, bool POOL, int PASS = 0>
__global__ __launch_bounds__(BN_THREADS) void bn_act_pool_bwd_kernel(const BnBwdPoolArgs a) {
  constexpr int VEC = Traits<T>::VEC;
  extern __shared__ __attribute__((aligned(16))) char dyn[];
  const int CV = a.C / VEC;
  const BlkMap m = blk_map(CV);
  const int Ho = POOL ? a.H / 2 : a.H, Wo = POOL ? a.W / 2 : a.W;
  const long NO = (long)a.B * Ho * Wo;
  const float ks = 1.0f / (1.0f - a.p);
  const uint32_t seed = launch_seed(a.seed, a.seed_dev);
  float s[2][VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) s[0][i] = s[1][i] = 0.f;
  if (m.active) {
    const int v = m.cv;
    float sc[VEC], sh[VEC], mu[VEC], is[VEC], k2[VEC], k3[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      sc[i] = a.scale[v * VEC + i]; sh[i] = a.shift[v * VEC + i]; mu[i] = a.mean[v * VEC + i]; is[i] = a.invstd[v * VEC + i];
      k2[i] = PASS == 2 ? a.coef[v * VEC + i] : 0.f; k3[i] = PASS == 2 ? a.coef[a.C + v * VEC + i] : 0.f;
    }
    const long b = (long)blockIdx.x * a.chunk, e = min(NO, b + a.chunk);
    for (long op = b + m.row; op < e; op += m.rows) {
      float g[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) g[i] = 0.f;
      if (a.d_o) {
        float f[VEC];
        Vec16<T>::unpack(*reinterpret_cast<const uint4*>(a.d_o + (size_t)op * a.do_stride * sizeof(T) + (size_t)v * 16), f);
#pragma unroll
        for (int i = 0; i < VEC; ++i) g[i] = f[i];
      }
      if (a.d_t) {
        float f[VEC];
        Vec16<T>::unpack(*reinterpret_cast<const uint4*>(a.d_t + (size_t)op * a.dt_stride * sizeof(T) + (size_t)v * 16), f);
#pragma unroll
        for (int i = 0; i < VEC; ++i) g[i] += a.p > 0.f ? f[i] * drop_factor(seed, (size_t)op * a.C + v * VEC + i, a.p, ks) : f[i];
      }
      if (POOL) {
        const int ox = (int)(op % Wo);
        const long q = op / Wo;
        const int oy = (int)(q % Ho), img = (int)(q / Ho);
        const size_t ip = ((size_t)img * a.H + 2 * oy) * a.W + 2 * ox;
        float yv[4][VEC];
#pragma unroll
        for (int k = 0; k < 4; ++k)
          Vec16<T>::unpack(*reinterpret_cast<const uint4*>(a.y + (ip + (k >> 1) * a.W + (k & 1)) * a.y_stride * sizeof(T) + (size_t)v * 16), yv[k]);
        float dz[4][VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          float best = fmaf(yv[0][i], sc[i], sh[i]);
          int arg = 0;
#pragma unroll
          for (int k = 1; k < 4; ++k) {
            const float z = fmaf(yv[k][i], sc[i], sh[i]);
            if (z > best) { best = z; arg = k; }
          }
          const float d = best > 0.f ? g[i] : 0.f;
          if constexpr (PASS == 2) {
#pragma unroll
            for (int k = 0; k < 4; ++k) dz[k][i] = sc[i] * ((k == arg ? d : 0.f) - k2[i] - (yv[k][i] - mu[i]) * is[i] * k3[i]);
          } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) dz[k][i] = k == arg ? d : 0.f;
            float ya = yv[0][i];
#pragma unroll
            for (int k = 1; k < 4; ++k) ya = k == arg ? yv[k][i] : ya;
            s[0][i] += d;
            s[1][i] = fmaf(d, (ya - mu[i]) * is[i], s[1][i]);
          }
        }
        if constexpr (PASS != 1) {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            *reinterpret_cast<uint4*>(a.dz + (ip + (k >> 1) * a.W + (k & 1)) * a.dz_stride * sizeof(T) + (size_t)v * 16) = Vec16<T>::pack(dz[k]);
        }
      } else {
        float yv[VEC], dz[VEC];
        Vec16<T>::unpack(*reinterpret_cast<const uint4*>(a.y + (size_t)op * a.y_stride * sizeof(T) + (size_t)v * 16), yv);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          const float d = fmaf(yv[i], sc[i], sh[i]) > 0.f ? g[i] : 0.f;
          if constexpr (PASS == 2) {
            dz[i] = sc[i] * (d - k2[i] - (yv[i] - mu[i]) * is[i] * k3[i]);
          } else {
            dz[i] = d;
            s[0][i] += d;
            s[1][i] = fmaf(d, (yv[i] - mu[i]) * is[i], s[1][i]);
          }
        }
        if constexpr (PASS != 1) *reinterpret_cast<uint4*>(a.dz + (size_t)op * a.dz_stride * sizeof(T) + (size_t)v * 16) = Vec16<T>::pack(dz);
      }
    }
  }
  if constexpr (PASS != 2) block_fold<VEC, 2>(s, m, CV, a.C, reinterpret_cast<float*>(dyn), a.partial);
}

// ---- forward: BN + ReLU (+ bilinear 2x) + skip ---------------------------------------------------------------------------
__device__ __forceinline__ void up_src(int dst, int in_size, int& i0, int& i1, float& l0, float& l1) {
  // ATen's half-pixel source index (align_corners=False): src = max(0, (dst + 0.5) / 2 - 0.5)
  float src = ((float)dst + 0.5f) * 0.5f - 0.5f;
  src = src < 0.f ? 0.f : src;
  i0 = (int)src;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = src - (float)i0;
  l0 = 1.0f - l1;
}

// out = up2?(relu(y * scale + shift)) + skip    (y at [B,H,W]; out and skip at [B,2H,2W] when UP)
template <typename T, bool UP>
__global__ __launch_bounds__(BN_THREADS) void bn_act_up_add_fwd_kernel(int B, int H, int W, int C, const char* y, int y_stride, const float* scale,
                                                                       const float* shift, const char* skip, int skip_stride, char* out, int out_stride) {
  constexpr int VEC = Traits<T>::VEC;
  const int CV = C / VEC;
  const int Ho = UP ? 2 * H : H, Wo = UP ? 2 * W : W;
  const size_t total = (size_t)B * Ho * Wo * CV;
  for (size_t u = (size_t)blockIdx.x * BN_THREADS + threadIdx.x; u < total; u += (size_t)gridDim.x * BN_THREADS) {
    const int v = (int)(u % CV);
    const size_t op = u / CV;
    float sc[VEC], sh[VEC], r[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) { sc[i] = scale[v * VEC + i]; sh[i] = shift[v * VEC + i]; }
    auto act = [&](size_t pix, float* f) {
      Vec16<T>::unpack(*reinterpret_cast<const uint4*>(y + pix * y_stride * sizeof(T) + (size_t)v * 16), f);
#pragma unroll
      for (int i = 0; i < VEC; ++i) f[i] = fmaxf(fmaf(f[i], sc[i], sh[i]), 0.f);
    };
    if (UP) {
      const int ox = (int)(op % Wo);
      const size_t q = op / Wo;
      const int oy = (int)(q % Ho), img = (int)(q / Ho);
      int y0, y1, x0, x1;
      float hy0, hy1, wx0, wx1;
      up_src(oy, H, y0, y1, hy0, hy1);
      up_src(ox, W, x0, x1, wx0, wx1);
      const size_t ib = (size_t)img * H * W;
      float a00[VEC], a01[VEC], a10[VEC], a11[VEC];
      act(ib + (size_t)y0 * W + x0, a00); act(ib + (size_t)y0 * W + x1, a01);
      act(ib + (size_t)y1 * W + x0, a10); act(ib + (size_t)y1 * W + x1, a11);
#pragma unroll
      for (int i = 0; i < VEC; ++i) r[i] = hy0 * (wx0 * a00[i] + wx1 * a01[i]) + hy1 * (wx0 * a10[i] + wx1 * a11[i]);
    } else {
      act(op, r);
    }
    if (skip) {
      float sk[VEC];
      Vec16<T>::unpack(*reinterpret_cast<const uint4*>(skip + op * skip_stride * sizeof(T) + (size_t)v * 16), sk);
#pragma unroll
      for (int i = 0; i < VEC; ++i) r[i] += sk[i];
    }
    *reinterpret_cast<uint4*>(out + op * out_stride * sizeof(T) + (size_t)v * 16) = Vec16<T>::pack(r);
  }
}

// backward producer: dz[lo] = (transposed bilinear 2x of dout)[lo] * [z > 0], + partial sums
struct BnBwdUpArgs {
  int B, H, W, C;                    // low (y) resolution
  const char* y; int y_stride;
  const float *scale, *shift, *mean, *invstd;
  const char* dout; int dout_stride; // [B, 2H, 2W, C]
  char* dz; int dz_stride;
  float* partial;
  long chunk;
};

template <typename T>
__global__ __launch_bounds__(BN_THREADS) void bn_act_up_bwd_kernel(const BnBwdUpArgs a) {
  constexpr int VEC = Traits<T>::VEC;
  extern __shared__ __attribute__((aligned(16))) char dyn[];
  const int CV = a.C / VEC;
  const BlkMap m = blk_map(CV);
  const long N = (long)a.B * a.H * a.W;
  const int Ho = 2 * a.H, Wo = 2 * a.W;
  float s[2][VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) s[0][i] = s[1][i] = 0.f;
  if (m.active) {
    const int v = m.cv;
    float sc[VEC], sh[VEC], mu[VEC], is[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) { sc[i] = a.scale[v * VEC + i]; sh[i] = a.shift[v * VEC + i]; mu[i] = a.mean[v * VEC + i]; is[i] = a.invstd[v * VEC + i]; }
    const long b = (long)blockIdx.x * a.chunk, e = min(N, b + a.chunk);
    for (long p = b + m.row; p < e; p += m.rows) {
      const int x = (int)(p % a.W);
      const long q = p / a.W;
      const int yy = (int)(q % a.H), img = (int)(q / a.H);
      float g[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) g[i] = 0.f;
      // every output row/column whose two sources can include (yy, x): 2*yy-1 .. 2*yy+2
      for (int dy = -1; dy <= 2; ++dy) {
        const int oy = 2 * yy + dy;
        if (oy < 0 || oy >= Ho) continue;
        int y0, y1; float h0, h1;
        up_src(oy, a.H, y0, y1, h0, h1);
        const float wy = (y0 == yy ? h0 : 0.f) + (y1 == yy ? h1 : 0.f);
        if (wy == 0.f) continue;
        for (int dx = -1; dx <= 2; ++dx) {
          const int ox = 2 * x + dx;
          if (ox < 0 || ox >= Wo) continue;
          int x0, x1; float w0, w1;
          up_src(ox, a.W, x0, x1, w0, w1);
          const float wx = (x0 == x ? w0 : 0.f) + (x1 == x ? w1 : 0.f);
          if (wx == 0.f) continue;
          float f[VEC];
          Vec16<T>::unpack(*reinterpret_cast<const uint4*>(a.dout + (((size_t)img * Ho + oy) * Wo + ox) * a.dout_stride * sizeof(T) + (size_t)v * 16), f);
          const float wgt = wy * wx;
#pragma unroll
          for (int i = 0; i < VEC; ++i) g[i] = fmaf(wgt, f[i], g[i]);
        }
      }
      float yv[VEC], dz[VEC];
      Vec16<T>::unpack(*reinterpret_cast<const uint4*>(a.y + (size_t)p * a.y_stride * sizeof(T) + (size_t)v * 16), yv);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const float d = fmaf(yv[i], sc[i], sh[i]) > 0.f ? g[i] : 0.f;
        dz[i] = d;
        s[0][i] += d;
        s[1][i] = fmaf(d, (yv[i] - mu[i]) * is[i], s[1][i]);
      }
      *reinterpret_cast<uint4*>(a.dz + (size_t)p * a.dz_stride * sizeof(T) + (size_t)v * 16) = Vec16<T>::pack(dz);
    }
  }
  block_fold<VEC, 2>(s, m, CV, a.C, reinterpret_cast<float*>(dyn), a.partial);
}

// ---- backward of a pre-activation BN (dense layers): reduce over (da, x), then apply -----------------------------------------
struct BnBwdArgs {
  long N;
  int nseg;
  SegP x[MDIE_MAX_SEG];
  SegW g[MDIE_MAX_SEG];          // gradient destination, same channel partition as x (apply only)
  unsigned accumulate;            // bit s: g[s] += instead of =
  SegW acc32[MDIE_MAX_SEG];      // optional fp32 running sums (ptr = nullptr: none); see mdie_bn_bwd_desc
  int final_from[MDIE_MAX_SEG];
  const char* da; int da_stride;
  long da_plane;                  // 0, or elements between the 16-channel planes da is stored in
  int C;
  const float *mean, *invstd, *scale, *shift;
  int relu;                       // mask da by x * scale + shift > 0
  const float* coef;              // [2][coef_stride]: k2, k3 (apply only)
  int coef_stride;
  float* partial;                 // (reduce only)
  long chunk;
};

// first byte of channel vector v (VEC channels from c0 = v * VEC) of pixel 0 in a da tensor: rows [N][da_stride], or one plane
// per 16 channels (da_plane elements apart, pixels da_stride apart inside a plane)
template <typename T>
__device__ __forceinline__ const char* da_vec_base(const char* da, long da_plane, int c0) {
  return da_plane ? da + ((size_t)(c0 >> 4) * da_plane + (c0 & 15)) * sizeof(T) : da + (size_t)c0 * sizeof(T);
}

template <typename T>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_reduce_kernel(const BnBwdArgs a) {
  constexpr int VEC = Traits<T>::VEC;
  extern __shared__ __attribute__((aligned(16))) char dyn[];
  const int CV = a.C / VEC;
  const BlkMap m = blk_map(CV);
  float s[2][VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) s[0][i] = s[1][i] = 0.f;
  if (m.active) {
    const int v = m.cv, c0 = v * VEC;
    const char* xb = nullptr; int xs = 0;
#pragma unroll
    for (int k = 0; k < MDIE_MAX_SEG; ++k)
      if (k < a.nseg && c0 >= a.x[k].ch_begin && c0 < a.x[k].ch_end) { xb = a.x[k].ptr + (size_t)(c0 - a.x[k].ch_begin) * sizeof(T); xs = a.x[k].stride; }
    float sc[VEC], sh[VEC], mu[VEC], is[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) { sc[i] = a.scale[c0 + i]; sh[i] = a.shift[c0 + i]; mu[i] = a.mean[c0 + i]; is[i] = a.invstd[c0 + i]; }
    const char* const dab = da_vec_base<T>(a.da, a.da_plane, c0);
    const long b = (long)blockIdx.x * a.chunk, e = min(a.N, b + a.chunk);
#pragma unroll 4
    for (long p = b + m.row; p < e; p += m.rows) {   // (unrolled: a read-only loop, 8 loads in flight per thread)
      float xv[VEC], d[VEC];
      Vec16<T>::unpack(*reinterpret_cast<const uint4*>(xb + (size_t)p * xs * sizeof(T)), xv);
      Vec16<T>::unpack(*reinterpret_cast<const uint4*>(dab + (size_t)p * a.da_stride * sizeof(T)), d);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const float dd = (!a.relu || fmaf(xv[i], sc[i], sh[i]) > 0.f) ? d[i] : 0.f;
        s[0][i] += dd;
        s[1][i] = fmaf(dd, (xv[i] - mu[i]) * is[i], s[1][i]);
      }
    }
  }
  block_fold<VEC, 2>(s, m, CV, a.C, reinterpret_cast<float*>(dyn), a.partial);
}

// dgamma / dbeta in the parameter's (real-channel) layout, k2 / k3 per stored channel
__global__ __launch_bounds__(64) void bn_bwd_final_kernel(int nblk, int C_st, int C_real, int split, int gap, double n, const float* partial, float* dgamma,
                                                          float* dbeta, float* coef) {
  const int cs = blockIdx.x;
  double s1 = 0.0, s2 = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 64) { s1 += partial[((size_t)b * 2 + 0) * C_st + cs]; s2 += partial[((size_t)b * 2 + 1) * C_st + cs]; }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) { s1 += __shfl_xor(s1, d); s2 += __shfl_xor(s2, d); }
  if (threadIdx.x == 0) {
    coef[cs] = (float)(s1 / n);
    coef[C_st + cs] = (float)(s2 / n);
    const int c = real_channel(cs, split, gap, C_real);
    if (c >= 0) { dbeta[c] = (float)s1; dgamma[c] = (float)s2; }
  }
}

// the same fold for partial sums a CONVOLUTION left (mdie_conv_desc.bnred): slab sums of dz and of dz * x (raw x: the convolution's
// epilogue holds no mean / invstd), so  sum dz * xhat = invstd * (sum dz * x - mean * sum dz).  One 256-thread block per stored
// channel (up to B * tiles slabs: 8 k at 8 x 512 x 512), lanes by shuffles, the four waves in order.
__global__ __launch_bounds__(256) void bn_bwd_final_raw_kernel(int nblk, int C_st, int C_real, int split, int gap, double n, const float* partial, const float* mean,
                                                              const float* invstd, float* dgamma, float* dbeta, float* coef) {
  __shared__ double red[2][4];
  const int cs = blockIdx.x;
  double s1 = 0.0, s2 = 0.0;
#pragma unroll 4
  for (int b = threadIdx.x; b < nblk; b += 256) { s1 += partial[((size_t)b * 2 + 0) * C_st + cs]; s2 += partial[((size_t)b * 2 + 1) * C_st + cs]; }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) { s1 += __shfl_xor(s1, d); s2 += __shfl_xor(s2, d); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s1; red[1][threadIdx.x >> 6] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    s1 = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    s2 = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    s2 = (double)invstd[cs] * (s2 - (double)mean[cs] * s1);
    coef[cs] = (float)(s1 / n);
    coef[C_st + cs] = (float)(s2 / n);
    const int c = real_channel(cs, split, gap, C_real);
    if (c >= 0) { dbeta[c] = (float)s1; dgamma[c] = (float)s2; }
  }
}

// thread = (channel vector, pixel row) with the channel fixed for the whole kernel: the six per-channel constants
// live in registers and the pixel loop is pure streaming (da, x in; g in/out)
template <typename T>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_apply_kernel(const BnBwdArgs a) {
  constexpr int VEC = Traits<T>::VEC;
  const int CV = a.C / VEC;
  const BlkMap m = blk_map(CV);
  if (!m.active) return;
  const int v = m.cv, c0 = v * VEC;
  const char* xb = nullptr; int xs = 0;
  char* gb = nullptr; int gs = 0; bool acc = false;
  float* sb = nullptr; int ss = 0; bool fin = true;     // fp32 running sums of this thread's channels; fin: write the final T value
#pragma unroll
  for (int k = 0; k < MDIE_MAX_SEG; ++k)
    if (k < a.nseg && c0 >= a.x[k].ch_begin && c0 < a.x[k].ch_end) {
      xb = a.x[k].ptr + (size_t)(c0 - a.x[k].ch_begin) * sizeof(T); xs = a.x[k].stride;
      gb = a.g[k].ptr + (size_t)(c0 - a.g[k].ch_begin) * sizeof(T); gs = a.g[k].stride;
      acc = (a.accumulate >> k) & 1u;
      if (a.acc32[k].ptr) {
        sb = reinterpret_cast<float*>(a.acc32[k].ptr) + (c0 - a.x[k].ch_begin); ss = a.acc32[k].stride;
        fin = (c0 - a.x[k].ch_begin) >= a.final_from[k];
      }
    }
  float sc[VEC], sh[VEC], mu[VEC], is[VEC], k2[VEC], k3[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    sc[i] = a.scale[c0 + i]; sh[i] = a.shift[c0 + i]; mu[i] = a.mean[c0 + i]; is[i] = a.invstd[c0 + i];
    k2[i] = a.coef[c0 + i]; k3[i] = a.coef[a.coef_stride + c0 + i];
  }
  const long b = (long)blockIdx.x * a.chunk, e = min(a.N, b + a.chunk);
  // Streaming loop, software-pipelined: the 2-3 loads of the NEXT pixel are requested before the current one is combined and
  // stored (past the end they re-read the thread's first pixel; the result is unused), so two pixels are in flight per thread.
  struct It { uint4 x, d, g, g2; };
  const long p_first = b + m.row;
  const bool acc_t = acc && !sb, acc_f = acc && sb;    // earlier contributions are in g (element type) / in the fp32 sums
  auto fetch = [&](long p, It& t) {
    const long pp = p < e ? p : p_first;
    t.x = *reinterpret_cast<const uint4*>(xb + (size_t)pp * xs * sizeof(T));
    t.d = *reinterpret_cast<const uint4*>(da_vec_base<T>(a.da, a.da_plane, c0) + (size_t)pp * a.da_stride * sizeof(T));
    if (acc_t) t.g = *reinterpret_cast<const uint4*>(gb + (size_t)pp * gs * sizeof(T));
    if (acc_f) {
      t.g = *reinterpret_cast<const uint4*>(sb + (size_t)pp * ss);
      if constexpr (VEC == 8) t.g2 = *reinterpret_cast<const uint4*>(sb + (size_t)pp * ss + 4);
    }
  };
  It cur, nxt;
  if (p_first < e) fetch(p_first, cur);
  for (long p = p_first; p < e; p += m.rows) {
    fetch(p + m.rows, nxt);
    float xv[VEC], d[VEC], r[VEC];
    Vec16<T>::unpack(cur.x, xv);
    Vec16<T>::unpack(cur.d, d);
    if (acc_t) Vec16<T>::unpack(cur.g, r);
    else if (acc_f) {
      Vec16<float>::unpack(cur.g, r);
      if constexpr (VEC == 8) Vec16<float>::unpack(cur.g2, r + 4);
    } else {
#pragma unroll
      for (int i = 0; i < VEC; ++i) r[i] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const float dd = (!a.relu || fmaf(xv[i], sc[i], sh[i]) > 0.f) ? d[i] : 0.f;
      r[i] += sc[i] * (dd - k2[i] - (xv[i] - mu[i]) * is[i] * k3[i]);
    }
    if (fin) *reinterpret_cast<uint4*>(gb + (size_t)p * gs * sizeof(T)) = Vec16<T>::pack(r);
    else {
      *reinterpret_cast<uint4*>(sb + (size_t)p * ss) = Vec16<float>::pack(r);
      if constexpr (VEC == 8) *reinterpret_cast<uint4*>(sb + (size_t)p * ss + 4) = Vec16<float>::pack(r + 4);
    }
    cur = nxt;
  }
}

// The same pass for the common case -- element-type running gradient, every segment accumulating (ACC) or none -- with two
// register sets used alternately instead of `cur = nxt`: handed over by assignment, the freshly loaded registers are copied
// at the bottom of the loop behind an s_waitcnt vmcnt that waits out the loads just issued (one pixel in flight, not two);
// and with the ACC load unconditional the waits stay counted.  Past the block's last pixel a thread re-reads its first
// pixel and stores nothing.
template <typename T, bool ACC>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_apply2_kernel(const BnBwdArgs a) {
  constexpr int VEC = Traits<T>::VEC;
  const int CV = a.C / VEC;
  const BlkMap m = blk_map(CV);
  if (!m.active) return;
  const int v = m.cv, c0 = v * VEC;
  const char* xb = nullptr; int xs = 0;
  char* gb = nullptr; int gs = 0;
#pragma unroll
  for (int k = 0; k < MDIE_MAX_SEG; ++k)
    if (k < a.nseg && c0 >= a.x[k].ch_begin && c0 < a.x[k].ch_end) {
      xb = a.x[k].ptr + (size_t)(c0 - a.x[k].ch_begin) * sizeof(T); xs = a.x[k].stride;
      gb = a.g[k].ptr + (size_t)(c0 - a.g[k].ch_begin) * sizeof(T); gs = a.g[k].stride;
    }
  float sc[VEC], sh[VEC], mu[VEC], is[VEC], k2[VEC], k3[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    sc[i] = a.scale[c0 + i]; sh[i] = a.shift[c0 + i]; mu[i] = a.mean[c0 + i]; is[i] = a.invstd[c0 + i];
    k2[i] = a.coef[c0 + i]; k3[i] = a.coef[a.coef_stride + c0 + i];
  }
  const long b = (long)blockIdx.x * a.chunk, e = min(a.N, b + a.chunk);
  struct It { uint4 x, d, g; };
  const long p_first = b + m.row;
  auto fetch = [&](long p, It& t) {
    const long pp = p < e ? p : p_first;
    t.x = *reinterpret_cast<const uint4*>(xb + (size_t)pp * xs * sizeof(T));
    t.d = *reinterpret_cast<const uint4*>(da_vec_base<T>(a.da, a.da_plane, c0) + (size_t)pp * a.da_stride * sizeof(T));
    if constexpr (ACC) t.g = *reinterpret_cast<const uint4*>(gb + (size_t)pp * gs * sizeof(T));
  };
  auto combine = [&](long p, const It& t) {
    float xv[VEC], d[VEC], r[VEC];
    Vec16<T>::unpack(t.x, xv);
    Vec16<T>::unpack(t.d, d);
    if constexpr (ACC) Vec16<T>::unpack(t.g, r);
    else {
#pragma unroll
      for (int i = 0; i < VEC; ++i) r[i] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const float dd = (!a.relu || fmaf(xv[i], sc[i], sh[i]) > 0.f) ? d[i] : 0.f;
      r[i] += sc[i] * (dd - k2[i] - (xv[i] - mu[i]) * is[i] * k3[i]);
    }
    *reinterpret_cast<uint4*>(gb + (size_t)p * gs * sizeof(T)) = Vec16<T>::pack(r);
  };
  if (p_first >= e) return;
  It A, B;
  fetch(p_first, A);
  long p = p_first;
  for (; p + m.rows < e; p += 2 * m.rows) {       // whole pairs: no branch around a load or a store (counted waits)
    fetch(p + m.rows, B);
    __builtin_amdgcn_sched_barrier(0);
    combine(p, A);
    __builtin_amdgcn_sched_barrier(0);
    fetch(p + 2 * m.rows, A);                      // (past the end: the first pixel again, unused)
    __builtin_amdgcn_sched_barrier(0);
    combine(p + m.rows, B);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (p < e) combine(p, A);                        // odd tail
}

// ---- one pass for a tensor that SEVERAL BatchNorm layers consume ---------------------------------------------------------------------
// The input x of a DenseBlock is normalised by each of its four layers and by the transition (models/cdan.py:35,38: every
// BatchNorm sees cat(features)), so its gradient is the sum of five BatchNorm-ReLU backward terms.  Adding them layer by layer
// (mdie_bn_bwd_apply on all segments, accumulate) reads x and the running sum and writes the sum five times: 20 passes over
// the C channels where 7 do -- x once, each layer's da once, the sum once:
//     g = sum_j scale_j * (da_j * [x * scale_j + shift_j > 0]) - sum_j scale_j * k2_j - xhat * sum_j scale_j * k3_j
// with ONE rounding to the element type instead of five.  The x part of a layer's da is the contiguous prefix [0, C) of each of
// its rows (>= 32 bytes, 128-512 for the encoder blocks), so the five streams are read in whole sectors -- unlike the growth
// segments, 32-byte slices in the middle of those rows, for which the same idea measured SLOWER (profiles/LEDGER.md (rounds 1-4) section 5b) and which
// keep the per-layer pass.
struct BnMultiArgs {
  long N; int C;
  const char* x; int x_stride;
  char* g; int g_stride;
  const float *mean, *invstd;
  const char* da[5]; int da_stride[5]; long da_plane[5];
  const float *scale[5], *shift[5], *coef[5]; int coef_stride[5];
  long chunk;
};

template <typename T, int NL>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_apply_multi_kernel(const BnMultiArgs a) {
  constexpr int VEC = Traits<T>::VEC;
  const int CV = a.C / VEC;
  const BlkMap m = blk_map(CV);
  if (!m.active) return;
  const int v = m.cv, c0 = v * VEC;
  float sc[NL][VEC], sh[NL][VEC], mu[VEC], is[VEC], K2[VEC], K3[VEC];
  // (the constants as 16-byte loads: c0 is a multiple of 4 and every array starts on a 16-byte boundary -- with 5 layers a
  //  thread would otherwise open with 176 scalar-sized loads for a few pixels of work)
  auto ld4 = [&](const float* p, float* dst) {
#pragma unroll
    for (int i = 0; i < VEC; i += 4) { const float4 v4 = *reinterpret_cast<const float4*>(p + c0 + i); dst[i] = v4.x; dst[i + 1] = v4.y; dst[i + 2] = v4.z; dst[i + 3] = v4.w; }
  };
  ld4(a.mean, mu); ld4(a.invstd, is);
#pragma unroll
  for (int i = 0; i < VEC; ++i) { K2[i] = 0.f; K3[i] = 0.f; }
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    float k2[VEC], k3[VEC];
    ld4(a.scale[j], sc[j]); ld4(a.shift[j], sh[j]); ld4(a.coef[j], k2); ld4(a.coef[j] + a.coef_stride[j], k3);
#pragma unroll
    for (int i = 0; i < VEC; ++i) { K2[i] += sc[j][i] * k2[i]; K3[i] += sc[j][i] * k3[i]; }
  }
  const char* const xb = a.x + (size_t)c0 * sizeof(T);
  char* const gb = a.g + (size_t)c0 * sizeof(T);
  const char* dab[NL];
#pragma unroll
  for (int j = 0; j < NL; ++j) dab[j] = da_vec_base<T>(a.da[j], a.da_plane[j], c0);
  const long b = (long)blockIdx.x * a.chunk, e = min(a.N, b + a.chunk);
  struct It { uint4 x, d[NL]; };
  const long p_first = b + m.row;
  auto fetch = [&](long p, It& t) {
    const long pp = p < e ? p : p_first;
    t.x = *reinterpret_cast<const uint4*>(xb + (size_t)pp * a.x_stride * sizeof(T));
#pragma unroll
    for (int j = 0; j < NL; ++j) t.d[j] = *reinterpret_cast<const uint4*>(dab[j] + (size_t)pp * a.da_stride[j] * sizeof(T));
  };
  auto combine = [&](long p, const It& t) {
    float xv[VEC], r[VEC];
    Vec16<T>::unpack(t.x, xv);
#pragma unroll
    for (int i = 0; i < VEC; ++i) r[i] = -K2[i] - (xv[i] - mu[i]) * is[i] * K3[i];
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      float d[VEC];
      Vec16<T>::unpack(t.d[j], d);
#pragma unroll
      for (int i = 0; i < VEC; ++i) r[i] += fmaf(xv[i], sc[j][i], sh[j][i]) > 0.f ? sc[j][i] * d[i] : 0.f;
    }
    *reinterpret_cast<uint4*>(gb + (size_t)p * a.g_stride * sizeof(T)) = Vec16<T>::pack(r);
  };
  if (p_first >= e) return;
  It A, B;
  fetch(p_first, A);
  long p = p_first;
  for (; p + m.rows < e; p += 2 * m.rows) {       // whole pairs: no branch around a load or a store (counted waits), as bn_bwd_apply2_kernel
    fetch(p + m.rows, B);
    __builtin_amdgcn_sched_barrier(0);
    combine(p, A);
    __builtin_amdgcn_sched_barrier(0);
    fetch(p + 2 * m.rows, A);
    __builtin_amdgcn_sched_barrier(0);
    combine(p + m.rows, B);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (p < e) combine(p, A);
}

template <typename T>
__global__ __launch_bounds__(BN_THREADS) void sigmoid_bwd_nchw3_kernel(int B, int HW, const float* g, const float* y, T* dz, int dz_stride) {
  const size_t total = (size_t)B * HW;
  for (size_t u = (size_t)blockIdx.x * BN_THREADS + threadIdx.x; u < total; u += (size_t)gridDim.x * BN_THREADS) {
    const size_t img = u / HW, p = u - img * HW;
    T* o = dz + u * dz_stride;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const size_t i = (img * 3 + c) * HW + p;
      const float yy = y[i];
      st(o + c, g[i] * yy * (1.f - yy));
    }
#pragma unroll
    for (int c = 3; c < 16; ++c) st(o + c, 0.f);
  }
}

static int bn_grid(size_t total) {
  size_t g = (total + BN_THREADS - 1) / BN_THREADS;
  return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

// pixel-chunked reduction plan: block count and pixels per block
struct RedPlan { int blocks; long chunk; int rows; };
static RedPlan red_plan(long N, int CV) {
  RedPlan p;
  p.rows = BN_THREADS / CV;
  long want = (N + (long)p.rows * 8 - 1) / ((long)p.rows * 8);    // >= 8 pixels per thread
  if (want < 1) want = 1;
  if (want > BN_MAX_BLOCKS) want = BN_MAX_BLOCKS;
  p.chunk = (N + want - 1) / want;
  p.blocks = (int)((N + p.chunk - 1) / p.chunk);
  return p;
}

static int bn_check(const char* what, int dtype, long N, int C) {
  MDIE_REQUIRE(dtype_valid(dtype), "%s: bad dtype %d", what, dtype);
  MDIE_REQUIRE(N > 0, "%s: empty tensor", what);
  const int vec = dtype_vec(dtype);
  MDIE_REQUIRE(C > 0 && C % 16 == 0 && C / vec <= BN_THREADS, "%s: C = %d must be a multiple of 16 and <= %d", what, C, BN_THREADS * vec);
  return MDIE_OK;
}

}  // namespace mdie

using namespace mdie;

extern "C" size_t mdie_bn_workspace_bytes(int C) {
  if (C <= 0) return 0;
  return (size_t)BN_MAX_BLOCKS * 2 * C * sizeof(float);
}

extern "C" int mdie_bn_stats(int dtype, long N, const void* x, int C, int stride, float* mean, float* var, void* workspace, size_t workspace_bytes,
                             void* stream) {
  if (int e = bn_check("mdie_bn_stats", dtype, N, C)) return e;
  MDIE_REQUIRE(x && mean && var && workspace && stride >= C, "mdie_bn_stats: null pointer / stride %d < C %d", stride, C);
  if (workspace_bytes < mdie_bn_workspace_bytes(C)) { set_error("mdie_bn_stats: workspace %zu < %zu", workspace_bytes, mdie_bn_workspace_bytes(C)); return MDIE_ENOSPC; }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int vec = dtype_vec(dtype);
  const RedPlan p = red_plan(N, C / vec);
  const size_t lds = (size_t)p.rows * 2 * C * sizeof(float);
  float* partial = reinterpret_cast<float*>(workspace);
  MDIE_SWITCH_T(dtype, hipLaunchKernelGGL((bn_stats_kernel<T>), dim3(p.blocks), dim3(BN_THREADS), lds, s, N, (const char*)x, C, stride, p.chunk, partial));
  hipLaunchKernelGGL(bn_stats_final_kernel, dim3(C), dim3(64), 0, s, p.blocks, C, (double)N, partial, mean, var);
  MDIE_LAUNCH_CHECK("mdie_bn_stats");
  return MDIE_OK;
}

extern "C" int mdie_bn_stats_fold(const mdie_bn_stats_fold_desc* d, void* stream) {
  MDIE_REQUIRE(d != nullptr, "mdie_bn_stats_fold: null descriptor");
  MDIE_REQUIRE(d->mean && d->var && d->workspace && d->C > 0 && d->C % 16 == 0 && d->N > 0, "mdie_bn_stats_fold: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  BnFinalFoldArgs a{};
  a.C = d->C; a.n = (double)d->N; a.mean = d->mean; a.var = d->var;
  a.partial = reinterpret_cast<const float*>(d->workspace);
  if (d->x) {
    if (int e = bn_check("mdie_bn_stats_fold", d->dtype, d->N, d->C)) return e;
    MDIE_REQUIRE(d->stride >= d->C, "mdie_bn_stats_fold: stride %d < C %d", d->stride, d->C);
    if (d->workspace_bytes < mdie_bn_workspace_bytes(d->C)) { set_error("mdie_bn_stats_fold: workspace too small"); return MDIE_ENOSPC; }
    const int vec = dtype_vec(d->dtype);
    const RedPlan p = red_plan(d->N, d->C / vec);
    const size_t lds = (size_t)p.rows * 2 * d->C * sizeof(float);
    MDIE_SWITCH_T(d->dtype, hipLaunchKernelGGL((bn_stats_kernel<T>), dim3(p.blocks), dim3(BN_THREADS), lds, s, d->N, (const char*)d->x, d->C, d->stride, p.chunk,
                                               reinterpret_cast<float*>(d->workspace)));
    a.nblk = p.blocks;
  } else {   // the partial sums are already there (a convolution wrote them: mdie_conv_desc.bn_partial)
    MDIE_REQUIRE(d->n_partial > 0 && d->workspace_bytes >= (size_t)d->n_partial * 2 * d->C * sizeof(float), "mdie_bn_stats_fold: %d partial sums, %zu bytes",
                 d->n_partial, d->workspace_bytes);
    a.nblk = d->n_partial;
  }
  int grid = d->C;
  if (d->gamma) {
    MDIE_REQUIRE(d->C_fold > 0 && d->C_real > 0 && d->beta && d->scale && d->shift && d->invstd && d->fold_mean && d->fold_var, "mdie_bn_stats_fold: fold arguments");
    MDIE_REQUIRE(d->gap >= 0 && d->split >= 0 && d->C_fold >= d->C_real + (d->split < d->C_real ? d->gap : 0), "mdie_bn_stats_fold: %d stored channels cannot hold %d real + gap %d",
                 d->C_fold, d->C_real, d->gap);
    MDIE_REQUIRE((d->running_mean == nullptr) == (d->running_var == nullptr), "mdie_bn_stats_fold: running_mean / running_var must both be given or both be null");
    const ptrdiff_t off = d->mean - d->fold_mean;
    MDIE_REQUIRE(off >= 0 && off + d->C <= d->C_fold && d->var - d->fold_var == off, "mdie_bn_stats_fold: mean / var must lie inside the fold range");
    a.new_off = (int)off; a.C_fold = d->C_fold; a.C_real = d->C_real; a.split = d->split; a.gap = d->gap;
    a.fold_mean = d->fold_mean; a.fold_var = d->fold_var; a.gamma = d->gamma; a.beta = d->beta; a.eps = d->eps; a.momentum = d->momentum;
    a.rmean = d->running_mean; a.rvar = d->running_var; a.scale = d->scale; a.shift = d->shift; a.invstd = d->invstd;
    grid = d->C_fold;
  } else {
    a.new_off = 0; a.C_fold = d->C;
  }
  hipLaunchKernelGGL(bn_final_fold_kernel, dim3(grid), dim3(64), 0, s, a);
  MDIE_LAUNCH_CHECK("mdie_bn_stats_fold");
  return MDIE_OK;
}

extern "C" int mdie_bn_fold(int C_stored, int C_real, int split, int gap, const float* mean, const float* var, const float* gamma, const float* beta,
                            float eps, float momentum, long count, float* running_mean, float* running_var, float* scale, float* shift, float* invstd,
                            void* stream) {
  MDIE_REQUIRE(C_stored > 0 && C_real > 0 && mean && var && gamma && beta && scale && shift && invstd, "mdie_bn_fold: bad argument");
  MDIE_REQUIRE(gap >= 0 && split >= 0 && C_stored >= C_real + (split < C_real ? gap : 0), "mdie_bn_fold: %d stored channels cannot hold %d real + gap %d",
               C_stored, C_real, gap);
  MDIE_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "mdie_bn_fold: running_mean / running_var must both be given or both be null");
  hipLaunchKernelGGL(bn_fold_kernel, dim3(cdiv(C_stored, BN_THREADS)), dim3(BN_THREADS), 0, reinterpret_cast<hipStream_t>(stream), C_stored, C_real, split, gap,
                     mean, var, gamma, beta, eps, momentum, (double)count, running_mean, running_var, scale, shift, invstd);
  MDIE_LAUNCH_CHECK("mdie_bn_fold");
  return MDIE_OK;
}

extern "C" int mdie_bn_act_pool_fwd(int dtype, int B, int H, int W, int C, const void* y, int y_stride, const float* scale, const float* shift, int pool,
                                    void* out, int out_stride, void* out_drop, int drop_stride, float p, unsigned seed, const unsigned* seed_dev,
                                    void* stream) {
  if (int e = bn_check("mdie_bn_act_pool_fwd", dtype, (long)B * H * W, C)) return e;
  MDIE_REQUIRE(y && scale && shift && (out || out_drop), "mdie_bn_act_pool_fwd: null pointer");
  MDIE_REQUIRE(!pool || (H % 2 == 0 && W % 2 == 0), "mdie_bn_act_pool_fwd: pooling needs even H, W (got %dx%d)", H, W);
  MDIE_REQUIRE(p >= 0.f && p < 1.f, "mdie_bn_act_pool_fwd: dropout p = %f", (double)p);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int vec = dtype_vec(dtype);
  const size_t total = (size_t)B * (pool ? H / 2 : H) * (pool ? W / 2 : W) * (C / vec);
  const dim3 grid(bn_grid(total)), blk(BN_THREADS);
#define MDIE_BN_FWD(T, P) hipLaunchKernelGGL((bn_act_pool_fwd_kernel<T, P>), grid, blk, 0, s, B, H, W, C, (const char*)y, y_stride, scale, shift, (char*)out, \
                                             out_stride, (char*)out_drop, drop_stride, p, (uint32_t)seed, (const uint32_t*)seed_dev)
  MDIE_SWITCH_T(dtype, if (pool) MDIE_BN_FWD(T, true); else MDIE_BN_FWD(T, false));
#undef MDIE_BN_FWD
  MDIE_LAUNCH_CHECK("mdie_bn_act_pool_fwd");
  return MDIE_OK;
}

// launches the partial-sum fold shared by every backward producer
static int bn_bwd_finish(const char* what, int nblk, int C, int C_real, int split, int gap, double n, float* partial, float* dgamma, float* dbeta, float* coef,
                         hipStream_t s) {
  hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(C), dim3(64), 0, s, nblk, C, C_real, split, gap, n, partial, dgamma, dbeta, coef);
  MDIE_LAUNCH_CHECK(what);
  return MDIE_OK;
}

extern "C" int mdie_bn_act_pool_bwd(const mdie_bn_pool_bwd_desc* d, void* stream) {
  MDIE_REQUIRE(d != nullptr, "mdie_bn_act_pool_bwd: null descriptor");
  if (int e = bn_check("mdie_bn_act_pool_bwd", d->dtype, (long)d->B * d->H * d->W, d->C)) return e;
  MDIE_REQUIRE(d->y && d->scale && d->shift && d->mean && d->invstd && d->dz && d->dgamma && d->dbeta && d->coef && d->workspace && (d->d_out || d->d_drop),
               "mdie_bn_act_pool_bwd: null pointer");
  MDIE_REQUIRE(!d->pool || (d->H % 2 == 0 && d->W % 2 == 0), "mdie_bn_act_pool_bwd: pooling needs even H, W");
  if (d->workspace_bytes < mdie_bn_workspace_bytes(d->C)) { set_error("mdie_bn_act_pool_bwd: workspace too small"); return MDIE_ENOSPC; }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int vec = dtype_vec(d->dtype);
  const long NO = (long)d->B * (d->pool ? d->H / 2 : d->H) * (d->pool ? d->W / 2 : d->W);
  const RedPlan p = red_plan(NO, d->C / vec);
  BnBwdPoolArgs a{};
  a.B = d->B; a.H = d->H; a.W = d->W; a.C = d->C;
  a.y = (const char*)d->y; a.y_stride = d->y_stride;
  a.scale = d->scale; a.shift = d->shift; a.mean = d->mean; a.invstd = d->invstd;
  a.d_o = (const char*)d->d_out; a.do_stride = d->d_out_stride;
  a.d_t = (const char*)d->d_drop; a.dt_stride = d->d_drop_stride;
  a.p = d->p; a.seed = d->seed; a.seed_dev = (const uint32_t*)d->seed_dev;
  a.dz = (char*)d->dz; a.dz_stride = d->dz_stride;
  a.partial = reinterpret_cast<float*>(d->workspace);
  a.chunk = p.chunk;
  const size_t lds = (size_t)p.rows * 2 * d->C * sizeof(float);
#define MDIE_BN_BWD(T, P, PASS) hipLaunchKernelGGL((bn_act_pool_bwd_kernel<T, P, PASS>), dim3(p.blocks), dim3(BN_THREADS), (PASS) == 2 ? 0 : lds, s, a)
  if (!d->two_pass) {
    MDIE_SWITCH_T(d->dtype, if (d->pool) MDIE_BN_BWD(T, true, 0); else MDIE_BN_BWD(T, false, 0));
    return bn_bwd_finish("mdie_bn_act_pool_bwd", p.blocks, d->C, d->c_real, d->C, 0, (double)d->B * d->H * d->W, a.partial, d->dgamma, d->dbeta, d->coef, s);
  }
  MDIE_SWITCH_T(d->dtype, if (d->pool) MDIE_BN_BWD(T, true, 1); else MDIE_BN_BWD(T, false, 1));
  if (int e = bn_bwd_finish("mdie_bn_act_pool_bwd", p.blocks, d->C, d->c_real, d->C, 0, (double)d->B * d->H * d->W, a.partial, d->dgamma, d->dbeta, d->coef, s)) return e;
  a.coef = d->coef;
  MDIE_SWITCH_T(d->dtype, if (d->pool) MDIE_BN_BWD(T, true, 2); else MDIE_BN_BWD(T, false, 2));
#undef MDIE_BN_BWD
  MDIE_LAUNCH_CHECK("mdie_bn_act_pool_bwd");
  return MDIE_OK;
}

extern "C" int mdie_bn_act_up_add_fwd(int dtype, int B, int H, int W, int C, const void* y, int y_stride, const float* scale, const float* shift, int up,
                                      const void* skip, int skip_stride, void* out, int out_stride, void* stream) {
  if (int e = bn_check("mdie_bn_act_up_add_fwd", dtype, (long)B * H * W, C)) return e;
  MDIE_REQUIRE(y && scale && shift && out, "mdie_bn_act_up_add_fwd: null pointer");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int vec = dtype_vec(dtype);
  const size_t total = (size_t)B * H * W * (up ? 4 : 1) * (C / vec);
  const dim3 grid(bn_grid(total)), blk(BN_THREADS);
#define MDIE_BN_UP(T, U) hipLaunchKernelGGL((bn_act_up_add_fwd_kernel<T, U>), grid, blk, 0, s, B, H, W, C, (const char*)y, y_stride, scale, shift, \
                                            (const char*)skip, skip_stride, (char*)out, out_stride)
  MDIE_SWITCH_T(dtype, if (up) MDIE_BN_UP(T, true); else MDIE_BN_UP(T, false));
#undef MDIE_BN_UP
  MDIE_LAUNCH_CHECK("mdie_bn_act_up_add_fwd");
  return MDIE_OK;
}

extern "C" int mdie_bn_act_up_bwd(const mdie_bn_up_bwd_desc* d, void* stream) {
  MDIE_REQUIRE(d != nullptr, "mdie_bn_act_up_bwd: null descriptor");
  if (int e = bn_check("mdie_bn_act_up_bwd", d->dtype, (long)d->B * d->H * d->W, d->C)) return e;
  MDIE_REQUIRE(d->y && d->scale && d->shift && d->mean && d->invstd && d->dz && d->dgamma && d->dbeta && d->coef && d->workspace && d->dout,
               "mdie_bn_act_up_bwd: null pointer");
  if (d->workspace_bytes < mdie_bn_workspace_bytes(d->C)) { set_error("mdie_bn_act_up_bwd: workspace too small"); return MDIE_ENOSPC; }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int vec = dtype_vec(d->dtype);
  const long N = (long)d->B * d->H * d->W;
  const RedPlan p = red_plan(N, d->C / vec);
  BnBwdUpArgs a{};
  a.B = d->B; a.H = d->H; a.W = d->W; a.C = d->C;
  a.y = (const char*)d->y; a.y_stride = d->y_stride;
  a.scale = d->scale; a.shift = d->shift; a.mean = d->mean; a.invstd = d->invstd;
  a.dout = (const char*)d->dout; a.dout_stride = d->dout_stride;
  a.dz = (char*)d->dz; a.dz_stride = d->dz_stride;
  a.partial = reinterpret_cast<float*>(d->workspace);
  a.chunk = p.chunk;
  const size_t lds = (size_t)p.rows * 2 * d->C * sizeof(float);
  MDIE_SWITCH_T(d->dtype, hipLaunchKernelGGL((bn_act_up_bwd_kernel<T>), dim3(p.blocks), dim3(BN_THREADS), lds, s, a));
  return bn_bwd_finish("mdie_bn_act_up_bwd", p.blocks, d->C, d->c_real, d->C, 0, (double)N, a.partial, d->dgamma, d->dbeta, d->coef, s);
}

static int fill_bwd_args(const char* what, const mdie_bn_bwd_desc* d, BnBwdArgs& a, bool need_g) {
  MDIE_REQUIRE(d != nullptr, "%s: null descriptor", what);
  MDIE_REQUIRE(d->nseg >= 1 && d->nseg <= MDIE_MAX_SEG, "%s: nseg %d", what, d->nseg);
  int c = 0;
  for (int k = 0; k < d->nseg; ++k) {
    MDIE_REQUIRE(d->x[k].ptr && d->x[k].channels > 0 && d->x[k].channels % 16 == 0 && d->x[k].stride >= d->x[k].channels, "%s: x segment %d", what, k);
    a.x[k].ptr = (const char*)d->x[k].ptr; a.x[k].ch_begin = c; a.x[k].ch_end = c + d->x[k].channels; a.x[k].stride = d->x[k].stride;
    if (need_g) {
      MDIE_REQUIRE(d->g[k].ptr && d->g[k].channels == d->x[k].channels && d->g[k].stride >= d->g[k].channels, "%s: gradient segment %d", what, k);
      a.g[k].ptr = (char*)d->g[k].ptr; a.g[k].ch_begin = c; a.g[k].ch_end = c + d->g[k].channels; a.g[k].stride = d->g[k].stride;
      a.acc32[k].ptr = nullptr; a.final_from[k] = 0;
      if (d->acc32[k].ptr) {
        MDIE_REQUIRE(d->dtype != MDIE_F32 && d->acc32[k].channels == d->x[k].channels && d->acc32[k].stride >= d->acc32[k].channels &&
                     d->acc32[k].stride % 4 == 0 && ((uintptr_t)d->acc32[k].ptr & 15) == 0 && d->final_from[k] >= 0 && d->final_from[k] % 8 == 0,
                     "%s: fp32 accumulator of segment %d (16-bit element types only; same channels; final_from a multiple of 8)", what, k);
        a.acc32[k].ptr = (char*)d->acc32[k].ptr; a.acc32[k].ch_begin = c; a.acc32[k].ch_end = c + d->acc32[k].channels; a.acc32[k].stride = d->acc32[k].stride;
        a.final_from[k] = d->final_from[k];
      }
    }
    c += d->x[k].channels;
  }
  if (int e = bn_check(what, d->dtype, d->N, c)) return e;
  MDIE_REQUIRE(d->da && (d->da_plane ? (d->da_stride >= 16 && d->da_plane % 16 == 0) : d->da_stride >= c) && d->mean && d->invstd && d->scale && d->shift && d->coef, "%s: null pointer / da_stride", what);
  a.N = d->N; a.nseg = d->nseg; a.C = c;
  a.da = (const char*)d->da; a.da_stride = d->da_stride; a.da_plane = d->da_plane;
  a.mean = d->mean; a.invstd = d->invstd; a.scale = d->scale; a.shift = d->shift;
  a.relu = d->relu; a.coef = d->coef; a.accumulate = d->accumulate;
  MDIE_REQUIRE(d->coef_stride == 0 || d->coef_stride >= c, "%s: coef_stride %d < %d channels", what, d->coef_stride, c);
  a.coef_stride = d->coef_stride ? d->coef_stride : c;
  return MDIE_OK;
}

extern "C" int mdie_bn_bwd_reduce(const mdie_bn_bwd_desc* d, void* stream) {
  BnBwdArgs a{};
  if (int e = fill_bwd_args("mdie_bn_bwd_reduce", d, a, false)) return e;
  MDIE_REQUIRE(d->dgamma && d->dbeta && d->workspace, "mdie_bn_bwd_reduce: null pointer");
  if (d->workspace_bytes < mdie_bn_workspace_bytes(a.C)) { set_error("mdie_bn_bwd_reduce: workspace too small"); return MDIE_ENOSPC; }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int vec = dtype_vec(d->dtype);
  const RedPlan p = red_plan(a.N, a.C / vec);
  a.partial = reinterpret_cast<float*>(d->workspace);
  a.chunk = p.chunk;
  const size_t lds = (size_t)p.rows * 2 * a.C * sizeof(float);
  MDIE_SWITCH_T(d->dtype, hipLaunchKernelGGL((bn_bwd_reduce_kernel<T>), dim3(p.blocks), dim3(BN_THREADS), lds, s, a));
  return bn_bwd_finish("mdie_bn_bwd_reduce", p.blocks, a.C, d->c_real, d->split, d->gap, (double)a.N, a.partial, d->dgamma, d->dbeta, d->coef, s);
}

extern "C" int mdie_bn_bwd_finish(const mdie_bn_bwd_finish_desc* d, void* stream) {
  MDIE_REQUIRE(d != nullptr, "mdie_bn_bwd_finish: null descriptor");
  MDIE_REQUIRE(d->C > 0 && d->C % 16 == 0 && d->N > 0 && d->n_partial > 0, "mdie_bn_bwd_finish: C %d, N %ld, %d slabs", d->C, d->N, d->n_partial);
  MDIE_REQUIRE(d->partial && d->mean && d->invstd && d->dgamma && d->dbeta && d->coef, "mdie_bn_bwd_finish: null pointer");
  MDIE_REQUIRE(d->c_real > 0 && d->gap >= 0 && d->split >= 0 && d->C >= d->c_real + (d->split < d->c_real ? d->gap : 0), "mdie_bn_bwd_finish: %d stored channels cannot hold %d real + gap %d",
               d->C, d->c_real, d->gap);
  hipLaunchKernelGGL(bn_bwd_final_raw_kernel, dim3(d->C), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), d->n_partial, d->C, d->c_real, d->split, d->gap, (double)d->N,
                     d->partial, d->mean, d->invstd, d->dgamma, d->dbeta, d->coef);
  MDIE_LAUNCH_CHECK("mdie_bn_bwd_finish");
  return MDIE_OK;
}

extern "C" int mdie_bn_bwd_apply(const mdie_bn_bwd_desc* d, void* stream) {
  BnBwdArgs a{};
  if (int e = fill_bwd_args("mdie_bn_bwd_apply", d, a, true)) return e;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int vec = dtype_vec(d->dtype);
  const int rows = BN_THREADS / (a.C / vec);
  long blocks = (a.N + (long)rows * 4 - 1) / ((long)rows * 4);          // >= 4 pixels per thread
  if (blocks > 4096) blocks = 4096;
  a.chunk = (a.N + blocks - 1) / blocks;
  blocks = (a.N + a.chunk - 1) / a.chunk;
  bool any32 = false;
  for (int k = 0; k < d->nseg; ++k) any32 = any32 || d->acc32[k].ptr != nullptr;
  const unsigned all = (1u << d->nseg) - 1u;
  if (!any32 && ((d->accumulate & all) == all || (d->accumulate & all) == 0)) {
    if ((d->accumulate & all) == all) MDIE_SWITCH_T(d->dtype, hipLaunchKernelGGL((bn_bwd_apply2_kernel<T, true>), dim3((int)blocks), dim3(BN_THREADS), 0, s, a));
    else MDIE_SWITCH_T(d->dtype, hipLaunchKernelGGL((bn_bwd_apply2_kernel<T, false>), dim3((int)blocks), dim3(BN_THREADS), 0, s, a));
  } else {
    MDIE_SWITCH_T(d->dtype, hipLaunchKernelGGL((bn_bwd_apply_kernel<T>), dim3((int)blocks), dim3(BN_THREADS), 0, s, a));
  }
  MDIE_LAUNCH_CHECK("mdie_bn_bwd_apply");
  return MDIE_OK;
}

extern "C" int mdie_bn_bwd_apply_multi(const mdie_bn_bwd_multi_desc* d, void* stream) {
  MDIE_REQUIRE(d != nullptr, "mdie_bn_bwd_apply_multi: null descriptor");
  if (int e = bn_check("mdie_bn_bwd_apply_multi", d->dtype, d->N, d->C)) return e;
  MDIE_REQUIRE(d->nlayer >= 1 && d->nlayer <= 5, "mdie_bn_bwd_apply_multi: nlayer %d (1..5)", d->nlayer);
  MDIE_REQUIRE(d->x && d->g && d->mean && d->invstd && d->x_stride >= d->C && d->g_stride >= d->C && (((uintptr_t)d->mean | (uintptr_t)d->invstd) & 15) == 0,
               "mdie_bn_bwd_apply_multi: null pointer / stride / alignment");
  BnMultiArgs a{};
  a.N = d->N; a.C = d->C;
  a.x = (const char*)d->x; a.x_stride = d->x_stride; a.g = (char*)d->g; a.g_stride = d->g_stride;
  a.mean = d->mean; a.invstd = d->invstd;
  for (int j = 0; j < d->nlayer; ++j) {
    MDIE_REQUIRE(((((uintptr_t)d->scale[j] | (uintptr_t)d->shift[j] | (uintptr_t)d->coef[j]) & 15) == 0) && d->coef_stride[j] % 4 == 0, "mdie_bn_bwd_apply_multi: layer %d constants must be 16-byte aligned", j);
    MDIE_REQUIRE(d->da[j] && (d->da_plane[j] ? (d->da_stride[j] >= 16 && d->da_plane[j] % 16 == 0) : d->da_stride[j] >= d->C) && d->scale[j] && d->shift[j] && d->coef[j] && d->coef_stride[j] >= d->C && ((uintptr_t)d->da[j] & 15) == 0,
                 "mdie_bn_bwd_apply_multi: layer %d", j);
    a.da[j] = (const char*)d->da[j]; a.da_stride[j] = d->da_stride[j]; a.da_plane[j] = d->da_plane[j];
    a.scale[j] = d->scale[j]; a.shift[j] = d->shift[j]; a.coef[j] = d->coef[j]; a.coef_stride[j] = d->coef_stride[j];
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int vec = dtype_vec(d->dtype);
  const int rows = BN_THREADS / (a.C / vec);
  long blocks = (a.N + (long)rows * 16 - 1) / ((long)rows * 16);         // >= 16 pixels per thread: its constants are 4 * nlayer + 2 vectors
  if (blocks > 2048) blocks = 2048;
  a.chunk = (a.N + blocks - 1) / blocks;
  blocks = (a.N + a.chunk - 1) / a.chunk;
#define MDIE_MULTI(NL) MDIE_SWITCH_T(d->dtype, hipLaunchKernelGGL((bn_bwd_apply_multi_kernel<T, NL>), dim3((int)blocks), dim3(BN_THREADS), 0, s, a))
  switch (d->nlayer) { case 1: MDIE_MULTI(1); break; case 2: MDIE_MULTI(2); break; case 3: MDIE_MULTI(3); break; case 4: MDIE_MULTI(4); break; default: MDIE_MULTI(5); break; }
#undef MDIE_MULTI
  MDIE_LAUNCH_CHECK("mdie_bn_bwd_apply_multi");
  return MDIE_OK;
}

extern "C" int mdie_sigmoid_bwd_nchw3(int dtype, int B, int H, int W, const float* grad_nchw, const float* y_nchw, void* dz_nhwc16, int dz_stride, void* stream) {
  MDIE_REQUIRE(dtype_valid(dtype), "mdie_sigmoid_bwd_nchw3: bad dtype %d", dtype);
  MDIE_REQUIRE(B > 0 && H > 0 && W > 0 && grad_nchw && y_nchw && dz_nhwc16 && dz_stride >= 16, "mdie_sigmoid_bwd_nchw3: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const size_t total = (size_t)B * H * W;
  MDIE_SWITCH_T(dtype, hipLaunchKernelGGL((sigmoid_bwd_nchw3_kernel<T>), dim3(bn_grid(total)), dim3(BN_THREADS), 0, s, B, H * W, grad_nchw, y_nchw, (T*)dz_nhwc16, dz_stride));
  MDIE_LAUNCH_CHECK("mdie_sigmoid_bwd_nchw3");
  return MDIE_OK;
}
