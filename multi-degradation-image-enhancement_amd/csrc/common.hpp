// Shared device/host helpers for libmdie_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mdie.h"

namespace mdie {

void set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));

#define MDIE_REQUIRE(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      ::mdie::set_error(__VA_ARGS__);      \
      return MDIE_EINVAL;                  \
    }                                      \
  } while (0)

#define MDIE_LAUNCH_CHECK(what)                                                   \
  do {                                                                            \
    hipError_t e_ = hipGetLastError();                                            \
    if (e_ != hipSuccess) {                                                       \
      ::mdie::set_error("%s: launch failed: %s", what, hipGetErrorString(e_));    \
      return MDIE_ELAUNCH;                                                        \
    }                                                                             \
  } while (0)

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) short i16x2;

typedef __bf16 bf16;

template <typename T> struct Traits;
template <> struct Traits<float> {
  static constexpr int DT = MDIE_F32;
  static constexpr int VEC = 4;   // elements per 16 bytes
  static constexpr int KC = 16;   // channels per 64-byte K chunk
};
template <> struct Traits<bf16> {
  static constexpr int DT = MDIE_BF16;
  static constexpr int VEC = 8;
  static constexpr int KC = 32;
};

static inline size_t dtype_size(int dtype) { return dtype == MDIE_F32 ? 4 : 2; }

// ---- packed bf16 <-> f32 ------------------------------------------------------------------------
__device__ __forceinline__ float bf_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ uint32_t bf_pack(float lo, float hi) {
  bf16x2 v = {(bf16)lo, (bf16)hi};  // v_cvt_pk_bf16_f32, round-to-nearest-even, NaN preserving
  return __builtin_bit_cast(uint32_t, v);
}

// 16 bytes of T <-> VEC floats
template <typename T> struct Vec16;
template <> struct Vec16<float> {
  static constexpr int N = 4;
  __device__ __forceinline__ static void unpack(const uint4& u, float* f) {
    f[0] = __uint_as_float(u.x); f[1] = __uint_as_float(u.y);
    f[2] = __uint_as_float(u.z); f[3] = __uint_as_float(u.w);
  }
  __device__ __forceinline__ static uint4 pack(const float* f) {
    return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
  }
};
template <> struct Vec16<bf16> {
  static constexpr int N = 8;
  __device__ __forceinline__ static void unpack(const uint4& u, float* f) {
    f[0] = bf_lo(u.x); f[1] = bf_hi(u.x); f[2] = bf_lo(u.y); f[3] = bf_hi(u.y);
    f[4] = bf_lo(u.z); f[5] = bf_hi(u.z); f[6] = bf_lo(u.w); f[7] = bf_hi(u.w);
  }
  __device__ __forceinline__ static uint4 pack(const float* f) {
    return make_uint4(bf_pack(f[0], f[1]), bf_pack(f[2], f[3]), bf_pack(f[4], f[5]), bf_pack(f[6], f[7]));
  }
};

// relu(x * s + b) on the VEC channels of a 16-byte unit (the DenseLayer pre-activation, models/cdan.py:35-36, folded BatchNorm).
// Issue-bound kernels stage every input unit through this, so it is written for instruction count: packed fp32 FMAs
// (v_pk_fma_f32), one v_cvt_pk_bf16_f32 per pair, and the ReLU AFTER rounding as a packed int16 max (rounding keeps the
// sign, so max(bits, 0) on the bf16 halves equals rounding relu(x)): 20 VALU instructions per unit instead of 28.
template <typename T> struct PreAct;
template <> struct PreAct<bf16> {
  static constexpr int NP = 4;
  __device__ __forceinline__ static uint4 apply(const uint4& v, const f32x2 (&s)[4], const f32x2 (&b)[4]) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x2 x = {bf_lo(w[i]), bf_hi(w[i])};
      const f32x2 r = __builtin_elementwise_fma(x, s[i], b[i]);
      const i16x2 m = __builtin_elementwise_max(__builtin_bit_cast(i16x2, __builtin_convertvector(r, bf16x2)), i16x2{0, 0});
      o[i] = __builtin_bit_cast(uint32_t, m);
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
  }
};
template <> struct PreAct<float> {
  static constexpr int NP = 2;
  __device__ __forceinline__ static uint4 apply(const uint4& v, const f32x2 (&s)[2], const f32x2 (&b)[2]) {
    const f32x2 r0 = __builtin_elementwise_fma(f32x2{__uint_as_float(v.x), __uint_as_float(v.y)}, s[0], b[0]);
    const f32x2 r1 = __builtin_elementwise_fma(f32x2{__uint_as_float(v.z), __uint_as_float(v.w)}, s[1], b[1]);
    return make_uint4(__float_as_uint(fmaxf(r0.x, 0.f)), __float_as_uint(fmaxf(r0.y, 0.f)),
                      __float_as_uint(fmaxf(r1.x, 0.f)), __float_as_uint(fmaxf(r1.y, 0.f)));
  }
};

// scalar load/store of T as float
__device__ __forceinline__ float ld(const float* p) { return *p; }
__device__ __forceinline__ float ld(const bf16* p) { return (float)*p; }
__device__ __forceinline__ void st(float* p, float v) { *p = v; }
__device__ __forceinline__ void st(bf16* p, float v) { *p = (bf16)v; }

__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + __expf(-x)); }

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == MDIE_ACT_RELU) return fmaxf(v, 0.0f);
  if (act == MDIE_ACT_SIGMOID) return sigmoidf(v);
  return v;
}

__host__ __device__ static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- optional per-launch instrumentation (mdie_cdan_forward's launch_ms mode) ------------------------
struct LaunchTimer {
  hipStream_t stream = nullptr;
  hipEvent_t* ev = nullptr;  // 2 * cap events
  int* kind = nullptr;
  int cap = 0, n = 0;
  bool on() const { return ev != nullptr; }
  void begin(int k) {
    if (ev && n < cap) { kind[n] = k; (void)hipEventRecord(ev[2 * n], stream); }
  }
  void end() {
    if (ev && n < cap) { (void)hipEventRecord(ev[2 * n + 1], stream); ++n; }
  }
};
LaunchTimer*& current_timer();  // thread-local; nullptr when not instrumenting

struct TimedLaunch {
  LaunchTimer* t;
  explicit TimedLaunch(int kind) : t(current_timer()) { if (t) t->begin(kind); }
  ~TimedLaunch() { if (t) t->end(); }
};

}  // namespace mdie
