// What sits on either side of the network in the reference's inference loop:
//   feed      uint8 HWC image batch -> fp32 NCHW in [0,1]        (albumentations Normalize(0,1,255)+ToTensorV2,
//                                                                 utils/transforms_factory.py:78-81, config/low_light.json:105-106)
//   postproc  enhance_contrast / enhance_color / sharpen / soft_denoise   (utils/post_processing.py:5-77,
//             dispatched by utils/postprocessing_factory.py:19-41) and the (img*255).clip(0,255).astype(uint8)
//             HWC conversion of models/model.py:80-84
//   metrics   PSNR and SSIM of a batch (utils/metrics_factory.py:74-94 -> torchmetrics defaults, restated:
//             torchmetrics is not installed here, "parity unpinned" at that boundary)
// All tensors here have 3 channels and are tiny next to the network's activations (25 MB fp32 at
// B=32, 256x256), so these are plain bandwidth kernels with deterministic two-level reductions.
#include <math.h>

#include "common.hpp"

namespace mdie {

constexpr int PP_THREADS = 256;
constexpr int PP_SLABS = 32;  // partial sums per plane

static int pp_grid(size_t total) {
  size_t g = (total + PP_THREADS - 1) / PP_THREADS;
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }

// block-wide sum of one float per thread (deterministic order)
__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float s = 0.f;
  for (int i = 0; i < PP_THREADS / 64; ++i) s += red[i];
  return s;
}

// ---- feed / output conversion --------------------------------------------------------------------------------
__global__ __launch_bounds__(PP_THREADS) void u8hwc_to_f32nchw_kernel(int B, int H, int W, const uint8_t* in, float* out) {
  const size_t HW = (size_t)H * W, total = (size_t)B * HW;
  for (size_t p = (size_t)blockIdx.x * PP_THREADS + threadIdx.x; p < total; p += (size_t)gridDim.x * PP_THREADS) {
    const size_t img = p / HW, hw = p - img * HW;
    const uint8_t* s = in + p * 3;
    float* o = out + img * 3 * HW + hw;
    o[0] = (float)s[0] / 255.0f; o[HW] = (float)s[1] / 255.0f; o[2 * HW] = (float)s[2] / 255.0f;
  }
}

__global__ __launch_bounds__(PP_THREADS) void f32nchw_to_u8hwc_kernel(int B, int H, int W, const float* in, uint8_t* out) {
  const size_t HW = (size_t)H * W, total = (size_t)B * HW;
  for (size_t p = (size_t)blockIdx.x * PP_THREADS + threadIdx.x; p < total; p += (size_t)gridDim.x * PP_THREADS) {
    const size_t img = p / HW, hw = p - img * HW;
    const float* s = in + img * 3 * HW + hw;
    uint8_t* o = out + p * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      // (img * 255).clip(0, 255).astype(np.uint8): clip, then truncate toward zero (models/model.py:83-84)
      const float v = fminf(fmaxf(s[c * HW] * 255.0f, 0.0f), 255.0f);
      o[c] = (uint8_t)(int)v;
    }
  }
}

// ---- post-processing ---------------------------------------------------------------------------------------------
// per-plane partial sums: grid (PP_SLABS, B*3)
__global__ __launch_bounds__(PP_THREADS) void plane_partial_sum_kernel(int HW, const float* in, float* partial) {
  __shared__ float red[PP_THREADS / 64];
  const int plane = blockIdx.y, slab = blockIdx.x;
  const int per = (HW + PP_SLABS - 1) / PP_SLABS;
  const int b = slab * per, e = min(HW, b + per);
  float s = 0.f;
  for (int i = b + threadIdx.x; i < e; i += PP_THREADS) s += in[(size_t)plane * HW + i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) partial[plane * PP_SLABS + slab] = s;
}

// enhance_contrast (post_processing.py:5-15): (x - mean_hw) * f + mean_hw, clamp
__global__ __launch_bounds__(PP_THREADS) void contrast_kernel(int HW, int planes, const float* in, const float* partial, float factor, float* out) {
  const size_t total = (size_t)planes * HW;
  for (size_t u = (size_t)blockIdx.x * PP_THREADS + threadIdx.x; u < total; u += (size_t)gridDim.x * PP_THREADS) {
    const int plane = (int)(u / HW);
    float m = 0.f;
    for (int i = 0; i < PP_SLABS; ++i) m += partial[plane * PP_SLABS + i];
    m /= (float)HW;
    out[u] = clamp01((in[u] - m) * factor + m);
  }
}

// enhance_color (post_processing.py:18-30): gray + f * (x - gray), clamp
__global__ __launch_bounds__(PP_THREADS) void color_kernel(int B, int HW, const float* in, float factor, float* out) {
  const size_t total = (size_t)B * HW;
  for (size_t p = (size_t)blockIdx.x * PP_THREADS + threadIdx.x; p < total; p += (size_t)gridDim.x * PP_THREADS) {
    const size_t img = p / HW, hw = p - img * HW;
    const float* s = in + img * 3 * HW + hw;
    float* o = out + img * 3 * HW + hw;
    const float r = s[0], g = s[HW], b = s[2 * (size_t)HW];
    const float gray = 0.2989f * r + 0.5870f * g + 0.1140f * b;
    o[0] = clamp01(gray + factor * (r - gray));
    o[HW] = clamp01(gray + factor * (g - gray));
    o[2 * (size_t)HW] = clamp01(gray + factor * (b - gray));
  }
}

// depthwise 3x3 with zero padding, then blend / clamp:  out = clamp((1 - mix) * x + mix * conv3x3(x, k))
// sharpen (post_processing.py:33-54): k = (K*strength + eye(3)) / sum, mix = 1
// soft_denoise (:57-77):              k = [[1,2,1],[2,4,2],[1,2,1]]/16, mix = sigma
struct K9 { float k[9]; };
__global__ __launch_bounds__(PP_THREADS) void stencil3_kernel(int planes, int H, int W, const float* in, K9 kk, float mix, float* out) {
  const size_t HW = (size_t)H * W, total = (size_t)planes * HW;
  for (size_t u = (size_t)blockIdx.x * PP_THREADS + threadIdx.x; u < total; u += (size_t)gridDim.x * PP_THREADS) {
    const size_t plane = u / HW, hw = u - plane * HW;
    const int y = (int)(hw / W), x = (int)(hw - (size_t)y * W);
    const float* s = in + plane * HW;
    float acc = 0.f;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int yy = y + dy, xx = x + dx;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) acc = fmaf(kk.k[(dy + 1) * 3 + dx + 1], s[(size_t)yy * W + xx], acc);
      }
    const float c = s[hw];
    out[u] = clamp01((1.0f - mix) * c + mix * acc);
  }
}

__global__ __launch_bounds__(PP_THREADS) void copy_f32_kernel(size_t n, const float* in, float* out) {
  for (size_t u = (size_t)blockIdx.x * PP_THREADS + threadIdx.x; u < n; u += (size_t)gridDim.x * PP_THREADS) out[u] = in[u];
}

// ---- metrics -------------------------------------------------------------------------------------------------------
struct MetricWs {
  float* part;   // [PP_SLABS * planes][6]: sum sq err, min p, max p, min t, max t, (unused)
  float* ranges; // [4]: mse, psnr data range, ssim data range, (unused)
  float* spart;  // [tiles * planes]: SSIM partial sums
};

__global__ __launch_bounds__(PP_THREADS) void metric_stats_kernel(int HW, const float* pred, const float* target, float* part) {
  __shared__ float red[PP_THREADS / 64];
  __shared__ float mm[4][PP_THREADS / 64];
  const int plane = blockIdx.y, slab = blockIdx.x;
  const int per = (HW + PP_SLABS - 1) / PP_SLABS;
  const int b = slab * per, e = min(HW, b + per);
  float se = 0.f, pmin = INFINITY, pmax = -INFINITY, tmin = INFINITY, tmax = -INFINITY;
  for (int i = b + threadIdx.x; i < e; i += PP_THREADS) {
    const float p = pred[(size_t)plane * HW + i], t = target[(size_t)plane * HW + i];
    const float d = p - t;
    se = fmaf(d, d, se);
    pmin = fminf(pmin, p); pmax = fmaxf(pmax, p); tmin = fminf(tmin, t); tmax = fmaxf(tmax, t);
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    pmin = fminf(pmin, __shfl_xor(pmin, d)); pmax = fmaxf(pmax, __shfl_xor(pmax, d));
    tmin = fminf(tmin, __shfl_xor(tmin, d)); tmax = fmaxf(tmax, __shfl_xor(tmax, d));
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { mm[0][wave] = pmin; mm[1][wave] = pmax; mm[2][wave] = tmin; mm[3][wave] = tmax; }
  se = block_sum(se, red);  // contains the barriers that also publish mm[]
  if (threadIdx.x == 0) {
    for (int i = 1; i < PP_THREADS / 64; ++i) {
      mm[0][0] = fminf(mm[0][0], mm[0][i]); mm[1][0] = fmaxf(mm[1][0], mm[1][i]);
      mm[2][0] = fminf(mm[2][0], mm[2][i]); mm[3][0] = fmaxf(mm[3][0], mm[3][i]);
    }
    float* o = part + ((size_t)plane * PP_SLABS + slab) * 6;
    o[0] = se; o[1] = mm[0][0]; o[2] = mm[1][0]; o[3] = mm[2][0]; o[4] = mm[3][0]; o[5] = 0.f;
  }
}

__global__ void metric_ranges_kernel(int nparts, double count, const float* part, float* ranges) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double se = 0.0;
  float pmin = INFINITY, pmax = -INFINITY, tmin = INFINITY, tmax = -INFINITY;
  for (int i = 0; i < nparts; ++i) {
    const float* o = part + (size_t)i * 6;
    se += (double)o[0];
    pmin = fminf(pmin, o[1]); pmax = fmaxf(pmax, o[2]); tmin = fminf(tmin, o[3]); tmax = fmaxf(tmax, o[4]);
  }
  ranges[0] = (float)(se / count);
  // torchmetrics PeakSignalNoiseRatio(data_range=None): range of the target, tracked from an initial 0
  ranges[1] = fmaxf(tmax, 0.0f) - fminf(tmin, 0.0f);
  // torchmetrics SSIM(data_range=None): max(range(preds), range(target))
  ranges[2] = fmaxf(pmax - pmin, tmax - tmin);
}

struct G11 { float g[11]; };

// SSIM map of one plane on a 16x16 tile (valid 11x11 Gaussian window), summed over the cropped interior
// [5, H-5) x [5, W-5): torchmetrics reflect-pads by 5, filters, and crops 5 again, so only windows that
// lie inside the picture survive.  grid (tiles, planes).
__global__ __launch_bounds__(PP_THREADS) void ssim_kernel(int H, int W, const float* pred, const float* target, const float* ranges,
                                                          G11 gw, float* spart) {
  constexpr int TS = 16, PW = TS + 10;
  __shared__ float sp[PW][PW + 1], st[PW][PW + 1];
  __shared__ float hrow[5][PW][TS + 1];
  __shared__ float red[PP_THREADS / 64];
  const int tiles_x = cdiv(W, TS);
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x, plane = blockIdx.y;
  const int y0 = ty * TS - 5, x0 = tx * TS - 5;
  const float* P = pred + (size_t)plane * H * W;
  const float* T = target + (size_t)plane * H * W;
  for (int i = threadIdx.x; i < PW * PW; i += PP_THREADS) {
    const int py = i / PW, px = i - py * PW;
    const int gy = y0 + py, gx = x0 + px;
    const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
    sp[py][px] = in ? P[(size_t)gy * W + gx] : 0.f;
    st[py][px] = in ? T[(size_t)gy * W + gx] : 0.f;
  }
  __syncthreads();
  // horizontal pass: 5 moments on PW rows x TS columns
  for (int i = threadIdx.x; i < PW * TS; i += PP_THREADS) {
    const int py = i / TS, ox = i - py * TS;
    float a = 0.f, b = 0.f, aa = 0.f, bb = 0.f, ab = 0.f;
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      const float p = sp[py][ox + k], t = st[py][ox + k], w = gw.g[k];
      a = fmaf(w, p, a); b = fmaf(w, t, b); aa = fmaf(w, p * p, aa); bb = fmaf(w, t * t, bb); ab = fmaf(w, p * t, ab);
    }
    hrow[0][py][ox] = a; hrow[1][py][ox] = b; hrow[2][py][ox] = aa; hrow[3][py][ox] = bb; hrow[4][py][ox] = ab;
  }
  __syncthreads();
  const int oy = threadIdx.x / TS, ox = threadIdx.x % TS;
  float mu_p = 0.f, mu_t = 0.f, e_pp = 0.f, e_tt = 0.f, e_pt = 0.f;
#pragma unroll
  for (int k = 0; k < 11; ++k) {
    const float w = gw.g[k];
    mu_p = fmaf(w, hrow[0][oy + k][ox], mu_p); mu_t = fmaf(w, hrow[1][oy + k][ox], mu_t);
    e_pp = fmaf(w, hrow[2][oy + k][ox], e_pp); e_tt = fmaf(w, hrow[3][oy + k][ox], e_tt);
    e_pt = fmaf(w, hrow[4][oy + k][ox], e_pt);
  }
  const float L = ranges[2];
  const float c1 = (0.01f * L) * (0.01f * L), c2 = (0.03f * L) * (0.03f * L);
  const float s_pp = e_pp - mu_p * mu_p, s_tt = e_tt - mu_t * mu_t, s_pt = e_pt - mu_p * mu_t;
  float v = ((2.f * mu_p * mu_t + c1) * (2.f * s_pt + c2)) / ((mu_p * mu_p + mu_t * mu_t + c1) * (s_pp + s_tt + c2));
  const int gy = ty * TS + oy, gx = tx * TS + ox;
  if (!(gy >= 5 && gy < H - 5 && gx >= 5 && gx < W - 5)) v = 0.f;
  v = block_sum(v, red);
  if (threadIdx.x == 0) spart[(size_t)plane * gridDim.x + blockIdx.x] = v;
}

__global__ void metric_final_kernel(int planes, int tiles, double interior, const float* ranges, const float* spart, float* out2) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  // PSNR = 10 log10(range^2 / MSE)
  out2[0] = 10.0f * log10f(ranges[1] * ranges[1] / ranges[0]);
  // SSIM: mean over (C, H-10, W-10) per image, then mean over images == mean over everything (equal sizes)
  double s = 0.0;
  for (int i = 0; i < planes * tiles; ++i) s += (double)spart[i];
  out2[1] = (float)(s / ((double)planes * interior));
}

static size_t a256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace mdie

using namespace mdie;

extern "C" int mdie_u8hwc_to_f32nchw(int B, int H, int W, const uint8_t* in, float* out, void* stream) {
  MDIE_REQUIRE(B > 0 && H > 0 && W > 0 && in && out, "mdie_u8hwc_to_f32nchw: bad argument");
  hipLaunchKernelGGL(u8hwc_to_f32nchw_kernel, dim3(pp_grid((size_t)B * H * W)), dim3(PP_THREADS), 0, reinterpret_cast<hipStream_t>(stream), B, H, W, in, out);
  MDIE_LAUNCH_CHECK("mdie_u8hwc_to_f32nchw");
  return MDIE_OK;
}

extern "C" int mdie_f32nchw_to_u8hwc(int B, int H, int W, const float* in, uint8_t* out, void* stream) {
  MDIE_REQUIRE(B > 0 && H > 0 && W > 0 && in && out, "mdie_f32nchw_to_u8hwc: bad argument");
  hipLaunchKernelGGL(f32nchw_to_u8hwc_kernel, dim3(pp_grid((size_t)B * H * W)), dim3(PP_THREADS), 0, reinterpret_cast<hipStream_t>(stream), B, H, W, in, out);
  MDIE_LAUNCH_CHECK("mdie_f32nchw_to_u8hwc");
  return MDIE_OK;
}

extern "C" size_t mdie_postprocess_workspace_bytes(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  return 2 * a256((size_t)B * 3 * H * W * sizeof(float)) + a256((size_t)B * 3 * PP_SLABS * sizeof(float));
}

extern "C" int mdie_postprocess(int B, int H, int W, const float* y, const mdie_pp_op* ops, int nops, float* out_f32, uint8_t* out_u8_hwc,
                                void* workspace, size_t workspace_bytes, void* stream) {
  MDIE_REQUIRE(B > 0 && H > 0 && W > 0 && y, "mdie_postprocess: bad argument");
  MDIE_REQUIRE(nops >= 0 && (nops == 0 || ops), "mdie_postprocess: ops missing");
  MDIE_REQUIRE(out_f32 || out_u8_hwc, "mdie_postprocess: no output requested");
  if (workspace_bytes < mdie_postprocess_workspace_bytes(B, H, W) || !workspace) {
    set_error("mdie_postprocess: workspace %zu < %zu", workspace_bytes, mdie_postprocess_workspace_bytes(B, H, W));
    return MDIE_ENOSPC;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const size_t n = (size_t)B * 3 * H * W;
  char* ws = reinterpret_cast<char*>(workspace);
  float* buf[2] = {reinterpret_cast<float*>(ws), reinterpret_cast<float*>(ws + a256(n * sizeof(float)))};
  float* partial = reinterpret_cast<float*>(ws + 2 * a256(n * sizeof(float)));
  const float* cur = y;
  int flip = 0;
  const int HW = H * W, planes = B * 3;
  for (int i = 0; i < nops; ++i) {
    float* dst = buf[flip];
    const float p = ops[i].param;
    switch (ops[i].kind) {
      case MDIE_PP_CONTRAST:
        hipLaunchKernelGGL(plane_partial_sum_kernel, dim3(PP_SLABS, planes), dim3(PP_THREADS), 0, s, HW, cur, partial);
        hipLaunchKernelGGL(contrast_kernel, dim3(pp_grid(n)), dim3(PP_THREADS), 0, s, HW, planes, cur, partial, p, dst);
        break;
      case MDIE_PP_COLOR:
        hipLaunchKernelGGL(color_kernel, dim3(pp_grid((size_t)B * HW)), dim3(PP_THREADS), 0, s, B, HW, cur, p, dst);
        break;
      case MDIE_PP_SHARPEN: {
        // kernel = K*strength + eye(3) (an identity MATRIX, as the reference writes it), normalised by its sum
        const float base[9] = {0, -1, 0, -1, 5, -1, 0, -1, 0};
        K9 kk; float sum = 0.f;
        for (int j = 0; j < 9; ++j) { kk.k[j] = base[j] * p + ((j % 4 == 0) ? 1.0f : 0.0f); sum += kk.k[j]; }
        for (int j = 0; j < 9; ++j) kk.k[j] /= sum;
        hipLaunchKernelGGL(stencil3_kernel, dim3(pp_grid(n)), dim3(PP_THREADS), 0, s, planes, H, W, cur, kk, 1.0f, dst);
        break;
      }
      case MDIE_PP_DENOISE: {
        const float base[9] = {1, 2, 1, 2, 4, 2, 1, 2, 1};
        K9 kk;
        for (int j = 0; j < 9; ++j) kk.k[j] = base[j] / 16.0f;
        hipLaunchKernelGGL(stencil3_kernel, dim3(pp_grid(n)), dim3(PP_THREADS), 0, s, planes, H, W, cur, kk, p, dst);
        break;
      }
      default:
        set_error("mdie_postprocess: unknown op kind %d", ops[i].kind);
        return MDIE_EINVAL;
    }
    MDIE_LAUNCH_CHECK("mdie_postprocess");
    cur = dst;
    flip ^= 1;
  }
  if (out_f32 && out_f32 != cur) {
    hipLaunchKernelGGL(copy_f32_kernel, dim3(pp_grid(n)), dim3(PP_THREADS), 0, s, n, cur, out_f32);
    MDIE_LAUNCH_CHECK("mdie_postprocess");
  }
  if (out_u8_hwc) return mdie_f32nchw_to_u8hwc(B, H, W, cur, out_u8_hwc, stream);
  return MDIE_OK;
}

extern "C" size_t mdie_metrics_workspace_bytes(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  const size_t planes = (size_t)B * 3, tiles = (size_t)cdiv(W, 16) * cdiv(H, 16);
  return a256(planes * PP_SLABS * 6 * sizeof(float)) + a256(4 * sizeof(float)) + a256(planes * tiles * sizeof(float));
}

extern "C" int mdie_psnr_ssim(int B, int H, int W, const float* pred, const float* target, float* out2, void* workspace,
                              size_t workspace_bytes, void* stream) {
  MDIE_REQUIRE(B > 0 && pred && target && out2, "mdie_psnr_ssim: bad argument");
  MDIE_REQUIRE(H > 10 && W > 10, "mdie_psnr_ssim: SSIM needs H, W > 10 (11x11 window), got %dx%d", H, W);
  if (workspace_bytes < mdie_metrics_workspace_bytes(B, H, W) || !workspace) {
    set_error("mdie_psnr_ssim: workspace %zu < %zu", workspace_bytes, mdie_metrics_workspace_bytes(B, H, W));
    return MDIE_ENOSPC;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int planes = B * 3, HW = H * W, tiles = cdiv(W, 16) * cdiv(H, 16);
  char* ws = reinterpret_cast<char*>(workspace);
  float* part = reinterpret_cast<float*>(ws);
  float* ranges = reinterpret_cast<float*>(ws + a256((size_t)planes * PP_SLABS * 6 * sizeof(float)));
  float* spart = reinterpret_cast<float*>(reinterpret_cast<char*>(ranges) + a256(4 * sizeof(float)));
  G11 gw;
  double gs = 0.0;
  for (int i = 0; i < 11; ++i) { const double d = (i - 5) / 1.5; gw.g[i] = (float)exp(-0.5 * d * d); gs += gw.g[i]; }
  for (int i = 0; i < 11; ++i) gw.g[i] = (float)(gw.g[i] / gs);
  hipLaunchKernelGGL(metric_stats_kernel, dim3(PP_SLABS, planes), dim3(PP_THREADS), 0, s, HW, pred, target, part);
  hipLaunchKernelGGL(metric_ranges_kernel, dim3(1), dim3(64), 0, s, planes * PP_SLABS, (double)planes * HW, part, ranges);
  hipLaunchKernelGGL(ssim_kernel, dim3(tiles, planes), dim3(PP_THREADS), 0, s, H, W, pred, target, ranges, gw, spart);
  hipLaunchKernelGGL(metric_final_kernel, dim3(1), dim3(64), 0, s, planes, tiles, (double)(H - 10) * (W - 10), ranges, spart, out2);
  MDIE_LAUNCH_CHECK("mdie_psnr_ssim");
  return MDIE_OK;
}
