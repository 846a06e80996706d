// Bandwidth-bound glue of the decoder and the NCHW<->NHWC boundary.
//   upsample2x_add : F.interpolate(scale_factor=2, 'bilinear', align_corners=False) + torch.add
//                    (models/cdan.py:137-138,145-146,153-154) in one pass
//   layout kernels : fp32 NCHW (the nn.Module boundary) <-> internal NHWC
#include "common.hpp"

// No implicit FMA contraction in this file: its loops are unrolled, and the unrolled body and the remainder loop must
// round identically -- which copy handles a pixel depends on the grid, i.e. on the batch size, and an image's result
// must not (tests/test_gpu_parity.py::test_full_batch_properties).  Fused multiply-adds are written as fmaf() where wanted.
#pragma clang fp contract(off)

namespace mdie {

constexpr int RS_THREADS = 256;

// half-pixel source index exactly as ATen computes it: src = max(0, (dst + 0.5) * 0.5 - 0.5)
__device__ __forceinline__ void src_index(int dst, int in_size, int& i0, int& i1, float& l0, float& l1) {
  float src = ((float)dst + 0.5f) * 0.5f - 0.5f;
  src = src < 0.f ? 0.f : src;
  i0 = (int)src;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = src - (float)i0;
  l0 = 1.0f - l1;
}

template <typename T>
__global__ __launch_bounds__(RS_THREADS) void upsample2x_add_kernel(int B, int H, int W, int C, const char* __restrict__ lo, int lo_stride,
                                                                    const char* __restrict__ skip, int skip_stride, char* __restrict__ out, int out_stride) {
  constexpr int VEC = Traits<T>::VEC;
  const int CV = C / VEC;
  const int Ho = 2 * H, Wo = 2 * W;
  const size_t total = (size_t)B * Ho * Wo * CV;
#pragma unroll 2
  for (size_t u = (size_t)blockIdx.x * RS_THREADS + threadIdx.x; u < total; u += (size_t)gridDim.x * RS_THREADS) {
    const int v = (int)(u % CV);
    size_t p = u / CV;
    const int ox = (int)(p % Wo); p /= Wo;
    const int oy = (int)(p % Ho);
    const int img = (int)(p / Ho);
    int y0, y1, x0, x1;
    float hy0, hy1, wx0, wx1;
    src_index(oy, H, y0, y1, hy0, hy1);
    src_index(ox, W, x0, x1, wx0, wx1);
    const char* base = lo + (size_t)img * H * W * lo_stride * sizeof(T) + (size_t)v * 16;
    auto at = [&](int y, int x) { return *reinterpret_cast<const uint4*>(base + ((size_t)y * W + x) * lo_stride * sizeof(T)); };
    float a00[VEC], a01[VEC], a10[VEC], a11[VEC], sk[VEC], r[VEC];
    Vec16<T>::unpack(at(y0, x0), a00);
    Vec16<T>::unpack(at(y0, x1), a01);
    Vec16<T>::unpack(at(y1, x0), a10);
    Vec16<T>::unpack(at(y1, x1), a11);
    const size_t op = ((size_t)img * Ho + oy) * Wo + ox;
    Vec16<T>::unpack(*reinterpret_cast<const uint4*>(skip + op * skip_stride * sizeof(T) + (size_t)v * 16), sk);
#pragma unroll
    for (int i = 0; i < VEC; ++i)
      r[i] = hy0 * (wx0 * a00[i] + wx1 * a01[i]) + hy1 * (wx0 * a10[i] + wx1 * a11[i]) + sk[i];
    *reinterpret_cast<uint4*>(out + op * out_stride * sizeof(T) + (size_t)v * 16) = Vec16<T>::pack(r);
  }
}

// upsample + skip that also reduces what it writes (the next kernel is a CBAM whose first pass is exactly this
// reduction): grid (slabs, B); thread = (channel vector v, pixel row r); per-slab sums / maxima -> partial[b][slab][2][C]
template <typename T>
__global__ __launch_bounds__(RS_THREADS) void upsample2x_add_pool_kernel(int H, int W, int C, const char* __restrict__ lo, int lo_stride,
                                                                         const char* __restrict__ skip, int skip_stride, char* __restrict__ out,
                                                                         int out_stride, float* __restrict__ partial) {
  constexpr int VEC = Traits<T>::VEC;
  extern __shared__ __attribute__((aligned(16))) char dyn[];
  const int CV = C / VEC;
  const int rows = RS_THREADS / CV;
  float* rsum = reinterpret_cast<float*>(dyn);
  float* rmax = rsum + rows * C;
  const int Ho = 2 * H, Wo = 2 * W, npix = Ho * Wo;
  const int img = blockIdx.y, slab = blockIdx.x, nslab = gridDim.x;
  const int per = (npix + nslab - 1) / nslab;
  const int p_begin = slab * per, p_end = min(npix, p_begin + per);
  const int v = threadIdx.x % CV, r = threadIdx.x / CV;
  float s[VEC], m[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { s[i] = 0.f; m[i] = -INFINITY; }
  const char* base = lo + (size_t)img * H * W * lo_stride * sizeof(T) + (size_t)v * 16;
  if (r < rows) {
    // Latency-bound loop (5 loads per pixel, then their use).  Two pixels are in flight per thread, in two register sets used
    // alternately (the loop is unrolled by two): a single set handed over with `cur = nxt` makes the compiler copy the
    // freshly loaded registers at the bottom of the loop -- behind an s_waitcnt vmcnt that waits out the loads it has
    // just issued, i.e. one pixel in flight after all (ISA of the round-1 form; the same in bn_bwd_apply_kernel).  Every load
    // and store is unconditional: past the slab's end a thread re-reads and re-writes the slab's first pixel (the same value)
    // and leaves it out of the sums.  (lo / skip / out never overlap.)
    struct Px { uint4 a00, a01, a10, a11, sk; float hy0, hy1, wx0, wx1; };
    auto fetch = [&](int p, Px& q) {
      const int pp = p < p_end ? p : p_begin;
      const int oy = pp / Wo, ox = pp - oy * Wo;
      int y0, y1, x0, x1;
      src_index(oy, H, y0, y1, q.hy0, q.hy1);
      src_index(ox, W, x0, x1, q.wx0, q.wx1);
      auto at = [&](int y, int x) { return *reinterpret_cast<const uint4*>(base + ((size_t)y * W + x) * lo_stride * sizeof(T)); };
      q.a00 = at(y0, x0); q.a01 = at(y0, x1); q.a10 = at(y1, x0); q.a11 = at(y1, x1);
      q.sk = *reinterpret_cast<const uint4*>(skip + ((size_t)img * npix + pp) * skip_stride * sizeof(T) + (size_t)v * 16);
    };
    auto blend = [&](int p, const Px& c) {
      const bool live = p < p_end;
      const int pp = live ? p : p_begin;
      float a00[VEC], a01[VEC], a10[VEC], a11[VEC], sk[VEC], rr[VEC];
      Vec16<T>::unpack(c.a00, a00);
      Vec16<T>::unpack(c.a01, a01);
      Vec16<T>::unpack(c.a10, a10);
      Vec16<T>::unpack(c.a11, a11);
      Vec16<T>::unpack(c.sk, sk);
#pragma unroll
      for (int i = 0; i < VEC; ++i)
        rr[i] = c.hy0 * (c.wx0 * a00[i] + c.wx1 * a01[i]) + c.hy1 * (c.wx0 * a10[i] + c.wx1 * a11[i]) + sk[i];
      const uint4 packed = Vec16<T>::pack(rr);
      *reinterpret_cast<uint4*>(out + ((size_t)img * npix + pp) * out_stride * sizeof(T) + (size_t)v * 16) = packed;
      // reduce the STORED values (bf16-rounded), exactly what a separate pool pass over `out` would read
      float q[VEC];
      Vec16<T>::unpack(packed, q);
#pragma unroll
      for (int i = 0; i < VEC; ++i) { s[i] += live ? q[i] : 0.f; m[i] = live ? fmaxf(m[i], q[i]) : m[i]; }
    };
    Px A, B;
    const int p0 = p_begin + r;
    if (p0 < p_end) {
      fetch(p0, A);
      for (int p = p0; p < p_end; p += 2 * rows) {
        fetch(p + rows, B);
        __builtin_amdgcn_sched_barrier(0);      // (pinned: left alone the scheduler sinks these loads below the blend they are meant to fly under)
        blend(p, A);
        __builtin_amdgcn_sched_barrier(0);
        fetch(p + 2 * rows, A);
        __builtin_amdgcn_sched_barrier(0);
        blend(p + rows, B);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) { rsum[r * C + v * VEC + i] = s[i]; rmax[r * C + v * VEC + i] = m[i]; }
  }
  __syncthreads();
  float* dst = partial + ((size_t)img * nslab + slab) * 2 * C;
  for (int c = threadIdx.x; c < C; c += RS_THREADS) {
    float ss = 0.f, mm = -INFINITY;
    for (int k = 0; k < rows; ++k) { ss += rsum[k * C + c]; mm = fmaxf(mm, rmax[k * C + c]); }
    dst[c] = ss;
    dst[C + c] = mm;
  }
}

// last decoder stage: out[B,2H,2W,CST] = bilinear_x2(lo)[:, :3] + x (fp32 NCHW), one output pixel per thread
template <typename T, int CST>
__global__ __launch_bounds__(RS_THREADS) void upsample2x_add_nchw3_kernel(int B, int H, int W, const T* __restrict__ lo, int lo_stride,
                                                                          const float* __restrict__ x, T* __restrict__ out) {
  const int Ho = 2 * H, Wo = 2 * W;
  const size_t plane = (size_t)Ho * Wo;
  const size_t total = (size_t)B * plane;
#pragma unroll 2
  for (size_t p = (size_t)blockIdx.x * RS_THREADS + threadIdx.x; p < total; p += (size_t)gridDim.x * RS_THREADS) {
    const size_t img = p / plane, hw = p - img * plane;
    const int oy = (int)(hw / Wo), ox = (int)(hw - (size_t)oy * Wo);
    int y0, y1, x0, x1;
    float hy0, hy1, wx0, wx1;
    src_index(oy, H, y0, y1, hy0, hy1);
    src_index(ox, W, x0, x1, wx0, wx1);
    const T* lb = lo + img * H * W * lo_stride;
    const T *p00 = lb + ((size_t)y0 * W + x0) * lo_stride, *p01 = lb + ((size_t)y0 * W + x1) * lo_stride;
    const T *p10 = lb + ((size_t)y1 * W + x0) * lo_stride, *p11 = lb + ((size_t)y1 * W + x1) * lo_stride;
    const float* xp = x + img * 3 * plane + hw;
    float f[CST];
#pragma unroll
    for (int i = 0; i < CST; ++i) f[i] = 0.f;
    // the 4 taps: channels 0..3 of each with ONE load (8 bytes bf16 / 16 bytes f32; pixels are at least 16 bytes apart and
    // 16-byte aligned) instead of three 2- or 4-byte loads per tap
    float t00[4], t01[4], t10[4], t11[4];
    auto tap = [](const T* q, float (&t)[4]) {
      if constexpr (sizeof(T) == 2) {
        const uint2 u = *reinterpret_cast<const uint2*>(q);
        t[0] = Half<T>::lo(u.x); t[1] = Half<T>::hi(u.x); t[2] = Half<T>::lo(u.y); t[3] = Half<T>::hi(u.y);
      } else {
        const float4 u = *reinterpret_cast<const float4*>(q);
        t[0] = u.x; t[1] = u.y; t[2] = u.z; t[3] = u.w;
      }
    };
    tap(p00, t00); tap(p01, t01); tap(p10, t10); tap(p11, t11);
#pragma unroll
    for (int c = 0; c < 3; ++c)
      f[c] = hy0 * (wx0 * t00[c] + wx1 * t01[c]) + hy1 * (wx0 * t10[c] + wx1 * t11[c]) + xp[c * plane];
    uint4* o = reinterpret_cast<uint4*>(out + p * CST);
    constexpr int VEC = Traits<T>::VEC;
#pragma unroll
    for (int v = 0; v < CST / VEC; ++v) o[v] = Vec16<T>::pack(f + v * VEC);
  }
}

// fp32 NCHW [B,Creal,H,W] -> NHWC [B,H,W,Cst] (channels >= Creal zero filled)
template <typename T>
__global__ __launch_bounds__(RS_THREADS) void nchw_to_nhwc_kernel(int B, int Creal, int Cst, int H, int W, const float* x, T* out) {
  const size_t total = (size_t)B * H * W * Cst;
  const size_t HW = (size_t)H * W;
  for (size_t u = (size_t)blockIdx.x * RS_THREADS + threadIdx.x; u < total; u += (size_t)gridDim.x * RS_THREADS) {
    const int c = (int)(u % Cst);
    const size_t p = u / Cst;
    const size_t img = p / HW, hw = p % HW;
    const float v = c < Creal ? x[(img * Creal + c) * HW + hw] : 0.f;
    st(out + u, v);
  }
}

// NHWC [B,H,W,Cst] -> fp32 NCHW [B,Creal,H,W]
template <typename T>
__global__ __launch_bounds__(RS_THREADS) void nhwc_to_nchw_kernel(int B, int Creal, int Cst, int H, int W, const T* in, float* y) {
  const size_t HW = (size_t)H * W;
  const size_t total = (size_t)B * Creal * HW;
  for (size_t u = (size_t)blockIdx.x * RS_THREADS + threadIdx.x; u < total; u += (size_t)gridDim.x * RS_THREADS) {
    const size_t hw = u % HW;
    const size_t t = u / HW;
    const int c = (int)(t % Creal);
    const size_t img = t / Creal;
    y[u] = ld(in + (img * HW + hw) * Cst + c);
  }
}

// fp32 NCHW [B,3,H,W] -> NHWC16: one pixel per thread (three coalesced plane reads, one 32/64-byte pixel write)
template <typename T>
__global__ __launch_bounds__(RS_THREADS) void nchw3_to_nhwc16_kernel(int B, int H, int W, const float* x, T* out) {
  const size_t HW = (size_t)H * W;
  const size_t total = (size_t)B * HW;
  for (size_t p = (size_t)blockIdx.x * RS_THREADS + threadIdx.x; p < total; p += (size_t)gridDim.x * RS_THREADS) {
    const size_t img = p / HW, hw = p - img * HW;
    const float* xp = x + img * 3 * HW + hw;
    float f[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) f[i] = 0.f;
    f[0] = xp[0]; f[1] = xp[HW]; f[2] = xp[2 * HW];
    uint4* o = reinterpret_cast<uint4*>(out + p * 16);
    constexpr int VEC = Traits<T>::VEC;
#pragma unroll
    for (int v = 0; v < 16 / VEC; ++v) o[v] = Vec16<T>::pack(f + v * VEC);
  }
}

static int grid_for(size_t total) {
  size_t g = (total + RS_THREADS - 1) / RS_THREADS;
  const size_t cap = 256 * 16;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

template <typename T>
static int to_nhwc(int B, int Creal, int Cst, int H, int W, const float* x, void* out, hipStream_t s) {
  TimedLaunch tl(MDIE_K_LAYOUT);
  hipLaunchKernelGGL((nchw_to_nhwc_kernel<T>), dim3(grid_for((size_t)B * H * W * Cst)), dim3(RS_THREADS), 0, s, B, Creal, Cst, H, W, x,
                     reinterpret_cast<T*>(out));
  MDIE_LAUNCH_CHECK("nchw_to_nhwc");
  return MDIE_OK;
}
template <typename T>
static int to_nchw(int B, int Creal, int Cst, int H, int W, const void* in, float* y, hipStream_t s) {
  TimedLaunch tl(MDIE_K_LAYOUT);
  hipLaunchKernelGGL((nhwc_to_nchw_kernel<T>), dim3(grid_for((size_t)B * H * W * Creal)), dim3(RS_THREADS), 0, s, B, Creal, Cst, H, W,
                     reinterpret_cast<const T*>(in), y);
  MDIE_LAUNCH_CHECK("nhwc_to_nchw");
  return MDIE_OK;
}

static int check_layout(const char* who, int dtype, int B, int C, int H, int W, const void* a, const void* b) {
  MDIE_REQUIRE(dtype_valid(dtype), "%s: bad dtype %d", who, dtype);
  MDIE_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "%s: empty extent", who);
  MDIE_REQUIRE(a && b, "%s: null pointer", who);
  return MDIE_OK;
}

}  // namespace mdie

using namespace mdie;

extern "C" int mdie_upsample2x_add(int dtype, int B, int H, int W, int C, const void* lo, int lo_stride, const void* skip,
                                   int skip_stride, void* out, int out_stride, void* stream) {
  if (int e = check_layout("mdie_upsample2x_add", dtype, B, C, H, W, lo, out)) return e;
  MDIE_REQUIRE(skip != nullptr, "mdie_upsample2x_add: null skip");
  MDIE_REQUIRE(C % 16 == 0 && lo_stride % 16 == 0 && skip_stride % 16 == 0 && out_stride % 16 == 0,
               "mdie_upsample2x_add: channels/strides must be multiples of 16");
  MDIE_REQUIRE((((uintptr_t)lo | (uintptr_t)skip | (uintptr_t)out) & 15) == 0, "mdie_upsample2x_add: alignment");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const size_t total = (size_t)B * 4 * H * W * (C / dtype_vec(dtype));
  TimedLaunch tl(MDIE_K_UPSAMPLE);
  MDIE_SWITCH_T(dtype, hipLaunchKernelGGL((upsample2x_add_kernel<T>), dim3(grid_for(total)), dim3(RS_THREADS), 0, s, B, H, W, C, (const char*)lo, lo_stride,
                       (const char*)skip, skip_stride, (char*)out, out_stride));
  MDIE_LAUNCH_CHECK("mdie_upsample2x_add");
  return MDIE_OK;
}

extern "C" int mdie_pool_slabs(int H_out, int W_out) { return (long)H_out * W_out >= 65536 ? 128 : 32; }

extern "C" int mdie_upsample2x_add_pool(int dtype, int B, int H, int W, int C, const void* lo, int lo_stride, const void* skip,
                                        int skip_stride, void* out, int out_stride, float* pool_partial, int pool_slabs, void* stream) {
  if (int e = check_layout("mdie_upsample2x_add_pool", dtype, B, C, H, W, lo, out)) return e;
  MDIE_REQUIRE(skip != nullptr && pool_partial != nullptr, "mdie_upsample2x_add_pool: null skip / pool_partial");
  MDIE_REQUIRE(pool_slabs >= 1 && pool_slabs <= MDIE_POOL_SLABS_MAX, "mdie_upsample2x_add_pool: pool_slabs %d", pool_slabs);
  MDIE_REQUIRE(C % 16 == 0 && C <= 512 && (C & (C - 1)) == 0 && lo_stride % 16 == 0 && skip_stride % 16 == 0 && out_stride % 16 == 0,
               "mdie_upsample2x_add_pool: C must be a power of two <= 512, strides multiples of 16");
  MDIE_REQUIRE((((uintptr_t)lo | (uintptr_t)skip | (uintptr_t)out) & 15) == 0, "mdie_upsample2x_add_pool: alignment");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int vec = dtype_vec(dtype);
  const int rows = RS_THREADS / (C / vec);
  const size_t lds = (size_t)2 * rows * C * sizeof(float);
  TimedLaunch tl(MDIE_K_UPSAMPLE);
  const dim3 grid(pool_slabs, B);
  MDIE_SWITCH_T(dtype, hipLaunchKernelGGL((upsample2x_add_pool_kernel<T>), grid, dim3(RS_THREADS), lds, s, H, W, C, (const char*)lo, lo_stride, (const char*)skip,
                       skip_stride, (char*)out, out_stride, pool_partial));
  MDIE_LAUNCH_CHECK("mdie_upsample2x_add_pool");
  return MDIE_OK;
}

extern "C" int mdie_upsample2x_add_nchw3(int dtype, int B, int H, int W, const void* lo, int lo_stride, const float* x_nchw, void* out,
                                         int out_channels, void* stream) {
  if (int e = check_layout("mdie_upsample2x_add_nchw3", dtype, B, 3, H, W, lo, out)) return e;
  MDIE_REQUIRE(x_nchw != nullptr && lo_stride >= 4 && lo_stride % 4 == 0 && ((uintptr_t)lo & 15) == 0,
               "mdie_upsample2x_add_nchw3: null x, or lo not 16-byte aligned with a pixel stride that is a multiple of 4 channels (%d)", lo_stride);
  MDIE_REQUIRE(((uintptr_t)out & 15) == 0, "mdie_upsample2x_add_nchw3: alignment");
  const int vec = dtype_vec(dtype);
  MDIE_REQUIRE(out_channels == 16 || out_channels == vec, "mdie_upsample2x_add_nchw3: out_channels %d (16 or %d)", out_channels, vec);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int grid = grid_for((size_t)B * 4 * H * W);
  TimedLaunch tl(MDIE_K_UPSAMPLE);
  MDIE_SWITCH_T(dtype,
    if (out_channels == 16) hipLaunchKernelGGL((upsample2x_add_nchw3_kernel<T, 16>), dim3(grid), dim3(RS_THREADS), 0, s, B, H, W, (const T*)lo, lo_stride, x_nchw, (T*)out);
    else hipLaunchKernelGGL((upsample2x_add_nchw3_kernel<T, Traits<T>::VEC>), dim3(grid), dim3(RS_THREADS), 0, s, B, H, W, (const T*)lo, lo_stride, x_nchw, (T*)out));
  MDIE_LAUNCH_CHECK("mdie_upsample2x_add_nchw3");
  return MDIE_OK;
}

extern "C" int mdie_nchw_to_nhwc(int dtype, int B, int C, int H, int W, const float* x, void* out, void* stream) {
  if (int e = check_layout("mdie_nchw_to_nhwc", dtype, B, C, H, W, x, out)) return e;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  MDIE_SWITCH_T(dtype, return to_nhwc<T>(B, C, C, H, W, x, out, s));
}
extern "C" int mdie_nhwc_to_nchw(int dtype, int B, int C, int H, int W, const void* in, float* y, void* stream) {
  if (int e = check_layout("mdie_nhwc_to_nchw", dtype, B, C, H, W, in, y)) return e;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  MDIE_SWITCH_T(dtype, return to_nchw<T>(B, C, C, H, W, in, y, s));
}
extern "C" int mdie_nchw3_to_nhwc16(int dtype, int B, int H, int W, const float* x, void* out, void* stream) {
  if (int e = check_layout("mdie_nchw3_to_nhwc16", dtype, B, 3, H, W, x, out)) return e;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  TimedLaunch tl(MDIE_K_LAYOUT);
  const int grid = grid_for((size_t)B * H * W);
  MDIE_SWITCH_T(dtype, hipLaunchKernelGGL((nchw3_to_nhwc16_kernel<T>), dim3(grid), dim3(RS_THREADS), 0, s, B, H, W, x, reinterpret_cast<T*>(out)));
  MDIE_LAUNCH_CHECK("mdie_nchw3_to_nhwc16");
  return MDIE_OK;
}
extern "C" int mdie_nhwc16_to_nchw3(int dtype, int B, int H, int W, const void* in, float* y, void* stream) {
  if (int e = check_layout("mdie_nhwc16_to_nchw3", dtype, B, 3, H, W, in, y)) return e;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  MDIE_SWITCH_T(dtype, return to_nchw<T>(B, 3, 16, H, W, in, y, s));
}
