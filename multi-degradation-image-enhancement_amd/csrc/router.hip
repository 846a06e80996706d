// The pieces of the degradation classifier (ResNet18 backbone + two linear heads,
// classification/train_multilabel_classifier.py:117-131) that are not convolutions.  The 3x3 / 1x1 convolutions of
// the BasicBlocks run on mdie_conv_fwd (BatchNorm folded into post_scale / post_shift, the identity branch as
// `residual`, activation applied afterwards by relu_inplace), the 7x7 stem on mdie_stem7_fwd (conv.hip); here:
//   relu_inplace     the ReLU after `conv2 + identity` (mdie_conv_fwd adds its residual after the activation)
//   maxpool3x3s2     nn.MaxPool2d(kernel_size=3, stride=2, padding=1) after the stem
//   subsample2       x[:, ::2, ::2, :]: a stride-2 convolution == the stride-1 convolution sampled at even pixels
//                    (3x3, pad 1) / the 1x1 convolution of the sampled input (downsample branch)
//   avgpool_heads    AdaptiveAvgPool2d(1) + flatten + head_cls / head_sev (nn.Linear) + sigmoid
#include "common.hpp"

namespace mdie {

constexpr int RT_THREADS = 256;

static int rt_grid(size_t total) {
  size_t g = (total + RT_THREADS - 1) / RT_THREADS;
  return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

template <typename T>
__global__ __launch_bounds__(RT_THREADS) void maxpool3x3s2_kernel(int B, int H, int W, int C, const char* in, int in_stride, char* out, int out_stride) {
  constexpr int VEC = Traits<T>::VEC;
  const int CV = C / VEC, Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const size_t total = (size_t)B * Ho * Wo * CV;
  for (size_t u = (size_t)blockIdx.x * RT_THREADS + threadIdx.x; u < total; u += (size_t)gridDim.x * RT_THREADS) {
    const int v = (int)(u % CV);
    size_t p = u / CV;
    const int ox = (int)(p % Wo); p /= Wo;
    const int oy = (int)(p % Ho);
    const int img = (int)(p / Ho);
    float m[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) m[i] = -INFINITY;      // padding never wins (nn.MaxPool2d pads with -inf)
    for (int dy = -1; dy <= 1; ++dy) {
      const int y = 2 * oy + dy;
      if (y < 0 || y >= H) continue;
      for (int dx = -1; dx <= 1; ++dx) {
        const int x = 2 * ox + dx;
        if (x < 0 || x >= W) continue;
        float f[VEC];
        Vec16<T>::unpack(*reinterpret_cast<const uint4*>(in + (((size_t)img * H + y) * W + x) * in_stride * sizeof(T) + (size_t)v * 16), f);
#pragma unroll
        for (int i = 0; i < VEC; ++i) m[i] = fmaxf(m[i], f[i]);
      }
    }
    *reinterpret_cast<uint4*>(out + (((size_t)img * Ho + oy) * Wo + ox) * out_stride * sizeof(T) + (size_t)v * 16) = Vec16<T>::pack(m);
  }
}

template <typename T>
__global__ __launch_bounds__(RT_THREADS) void relu_inplace_kernel(size_t npix, int C, char* x, int stride) {
  constexpr int VEC = Traits<T>::VEC;
  const int CV = C / VEC;
  const size_t total = npix * CV;
  for (size_t u = (size_t)blockIdx.x * RT_THREADS + threadIdx.x; u < total; u += (size_t)gridDim.x * RT_THREADS) {
    uint4* p = reinterpret_cast<uint4*>(x + (u / CV) * stride * sizeof(T) + (u % CV) * 16);
    float f[VEC];
    Vec16<T>::unpack(*p, f);
#pragma unroll
    for (int i = 0; i < VEC; ++i) f[i] = fmaxf(f[i], 0.f);
    *p = Vec16<T>::pack(f);
  }
}

__global__ __launch_bounds__(RT_THREADS) void subsample2_kernel(int B, int H, int W, int cv16, const char* in, size_t in_pix_bytes, char* out,
                                                                size_t out_pix_bytes) {
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const size_t total = (size_t)B * Ho * Wo * cv16;
  for (size_t u = (size_t)blockIdx.x * RT_THREADS + threadIdx.x; u < total; u += (size_t)gridDim.x * RT_THREADS) {
    const int v = (int)(u % cv16);
    size_t p = u / cv16;
    const int ox = (int)(p % Wo); p /= Wo;
    const int oy = (int)(p % Ho);
    const int img = (int)(p / Ho);
    *reinterpret_cast<uint4*>(out + (((size_t)img * Ho + oy) * Wo + ox) * out_pix_bytes + (size_t)v * 16) =
        *reinterpret_cast<const uint4*>(in + (((size_t)img * H + 2 * oy) * W + 2 * ox) * in_pix_bytes + (size_t)v * 16);
  }
}

// one block per image: feat[c] = mean over pixels; logits = W feat + b for both heads; sigmoid
template <typename T>
__global__ __launch_bounds__(RT_THREADS) void avgpool_heads_kernel(int HW, int C, const T* x, int stride, const float* w_cls, const float* b_cls,
                                                                   const float* w_sev, const float* b_sev, int ncls, float* feat_out, float* prob_cls,
                                                                   float* sev) {
  extern __shared__ __attribute__((aligned(16))) char dyn[];
  float* feat = reinterpret_cast<float*>(dyn);   // [C]
  const int img = blockIdx.x, tid = threadIdx.x;
  const T* xi = x + (size_t)img * HW * stride;
  const float inv = 1.0f / (float)HW;
  for (int c = tid; c < C; c += RT_THREADS) {
    float s = 0.f;
    for (int p = 0; p < HW; ++p) s += ld(xi + (size_t)p * stride + c);
    feat[c] = s * inv;
    if (feat_out) feat_out[(size_t)img * C + c] = s * inv;
  }
  __syncthreads();
  // 2 * ncls dot products of length C: one wave each, lanes stride the channels, fixed-order shuffle fold
  const int lane = tid & 63, wave = tid >> 6;
  for (int o = wave; o < 2 * ncls; o += RT_THREADS / 64) {
    const bool is_sev = o >= ncls;
    const int k = is_sev ? o - ncls : o;
    const float* w = (is_sev ? w_sev : w_cls) + (size_t)k * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s = fmaf(w[c], feat[c], s);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d);
    if (lane == 0) {
      const float z = s + (is_sev ? b_sev[k] : b_cls[k]);
      (is_sev ? sev : prob_cls)[(size_t)img * ncls + k] = sigmoidf(z);
    }
  }
}

}  // namespace mdie

using namespace mdie;

static int rt_check(const char* what, int dtype, int B, int H, int W, int C, const void* in, const void* out, int in_stride, int out_stride) {
  MDIE_REQUIRE(dtype_valid(dtype), "%s: bad dtype %d", what, dtype);
  MDIE_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 16 == 0, "%s: bad extent %dx%dx%dx%d", what, B, H, W, C);
  MDIE_REQUIRE(in && out && in_stride >= C && out_stride >= C && in_stride % 4 == 0 && out_stride % 4 == 0, "%s: null pointer / stride", what);
  MDIE_REQUIRE((((uintptr_t)in | (uintptr_t)out) & 15) == 0, "%s: alignment", what);
  return MDIE_OK;
}

extern "C" int mdie_maxpool3x3s2(int dtype, int B, int H, int W, int C, const void* in, int in_stride, void* out, int out_stride, void* stream) {
  if (int e = rt_check("mdie_maxpool3x3s2", dtype, B, H, W, C, in, out, in_stride, out_stride)) return e;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const size_t total = (size_t)B * ((H + 1) / 2) * ((W + 1) / 2) * (C / dtype_vec(dtype));
  MDIE_SWITCH_T(dtype, hipLaunchKernelGGL((maxpool3x3s2_kernel<T>), dim3(rt_grid(total)), dim3(RT_THREADS), 0, s, B, H, W, C, (const char*)in, in_stride, (char*)out, out_stride));
  MDIE_LAUNCH_CHECK("mdie_maxpool3x3s2");
  return MDIE_OK;
}

extern "C" int mdie_relu_inplace(int dtype, long npix, int C, void* x, int stride, void* stream) {
  MDIE_REQUIRE(dtype_valid(dtype), "mdie_relu_inplace: bad dtype %d", dtype);
  MDIE_REQUIRE(npix > 0 && C > 0 && C % 16 == 0 && x && stride >= C && ((uintptr_t)x & 15) == 0 && stride % 4 == 0, "mdie_relu_inplace: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const size_t total = (size_t)npix * (C / dtype_vec(dtype));
  MDIE_SWITCH_T(dtype, hipLaunchKernelGGL((relu_inplace_kernel<T>), dim3(rt_grid(total)), dim3(RT_THREADS), 0, s, (size_t)npix, C, (char*)x, stride));
  MDIE_LAUNCH_CHECK("mdie_relu_inplace");
  return MDIE_OK;
}

extern "C" int mdie_subsample2(int dtype, int B, int H, int W, int C, const void* in, int in_stride, void* out, int out_stride, void* stream) {
  if (int e = rt_check("mdie_subsample2", dtype, B, H, W, C, in, out, in_stride, out_stride)) return e;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const size_t es = dtype_size(dtype);
  const int cv16 = (int)(C * es / 16);
  const size_t total = (size_t)B * ((H + 1) / 2) * ((W + 1) / 2) * cv16;
  hipLaunchKernelGGL(subsample2_kernel, dim3(rt_grid(total)), dim3(RT_THREADS), 0, s, B, H, W, cv16, (const char*)in, (size_t)in_stride * es, (char*)out,
                     (size_t)out_stride * es);
  MDIE_LAUNCH_CHECK("mdie_subsample2");
  return MDIE_OK;
}

extern "C" int mdie_avgpool_heads(int dtype, int B, int H, int W, int C, const void* x, int stride, const float* w_cls, const float* b_cls,
                                  const float* w_sev, const float* b_sev, int ncls, float* feat, float* prob_cls, float* sev, void* stream) {
  MDIE_REQUIRE(dtype_valid(dtype), "mdie_avgpool_heads: bad dtype %d", dtype);
  MDIE_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C <= 8192 && stride >= C, "mdie_avgpool_heads: bad extent");
  MDIE_REQUIRE(x && w_cls && b_cls && w_sev && b_sev && prob_cls && sev && ncls > 0, "mdie_avgpool_heads: null pointer");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  MDIE_SWITCH_T(dtype, hipLaunchKernelGGL((avgpool_heads_kernel<T>), dim3(B), dim3(RT_THREADS), (size_t)C * 4, s, H * W, C, (const T*)x, stride, w_cls, b_cls, w_sev, b_sev, ncls,
                       feat, prob_cls, sev));
  MDIE_LAUNCH_CHECK("mdie_avgpool_heads");
  return MDIE_OK;
}
