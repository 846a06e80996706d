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
typedef _Float16 f16;   // IEEE binary16: the reference's own mixed-precision dtype (torch.cuda.amp.autocast, models/model.py:15,159)
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

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
template <> struct Traits<f16> {
  static constexpr int DT = MDIE_F16;
  static constexpr int VEC = 8;
  static constexpr int KC = 32;
};

static inline bool dtype_valid(int dtype) { return dtype == MDIE_F32 || dtype == MDIE_BF16 || dtype == MDIE_F16; }
static inline size_t dtype_size(int dtype) { return dtype == MDIE_F32 ? 4 : 2; }
static inline int dtype_vec(int dtype) { return dtype == MDIE_F32 ? 4 : 8; }   // elements per 16 bytes
static inline int dtype_kc(int dtype) { return dtype == MDIE_F32 ? 16 : 32; }  // channels per 64-byte K chunk

// Runtime dtype -> compile-time element type: `stmt` is compiled three times with T = float / bf16 / f16.
#define MDIE_SWITCH_T(dtype, ...)                                           \
  do {                                                                      \
    if ((dtype) == MDIE_F32) { using T = float; __VA_ARGS__; }              \
    else if ((dtype) == MDIE_BF16) { using T = ::mdie::bf16; __VA_ARGS__; } \
    else { using T = ::mdie::f16; __VA_ARGS__; }                            \
  } while (0)

// host: f32 -> the bit pattern of the 16-bit storage type, round to nearest even (what the device conversions do)
static inline uint16_t f32_to_half_bits(int dtype, float f) {
  if (dtype == MDIE_F16) { const _Float16 h = (_Float16)f; uint16_t b; __builtin_memcpy(&b, &h, 2); return b; }
  uint32_t u; __builtin_memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

// ---- packed bf16 <-> f32 ------------------------------------------------------------------------
__device__ __forceinline__ float bf_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ uint32_t bf_pack(float lo, float hi) {
  bf16x2 v = {(bf16)lo, (bf16)hi};  // v_cvt_pk_bf16_f32, round-to-nearest-even, NaN preserving
  return __builtin_bit_cast(uint32_t, v);
}

// the two 16-bit element types share one interface: halves of a packed dword <-> f32 (fp32 is never routed through this)
template <typename T> struct Half;
template <> struct Half<bf16> {
  typedef bf16x2 v2;
  __device__ __forceinline__ static float lo(uint32_t u) { return bf_lo(u); }
  __device__ __forceinline__ static float hi(uint32_t u) { return bf_hi(u); }
  __device__ __forceinline__ static uint32_t pack(float lo, float hi) { return bf_pack(lo, hi); }
};
template <> struct Half<f16> {
  typedef f16x2 v2;
  __device__ __forceinline__ static float lo(uint32_t u) { return (float)__builtin_bit_cast(f16x2, u)[0]; }
  __device__ __forceinline__ static float hi(uint32_t u) { return (float)__builtin_bit_cast(f16x2, u)[1]; }
  __device__ __forceinline__ static uint32_t pack(float lo, float hi) {
    f16x2 v = {(f16)lo, (f16)hi};   // round-to-nearest-even (v_cvt_pk_f16_f32 / v_cvt_f16_f32), overflow -> inf like torch.half
    return __builtin_bit_cast(uint32_t, v);
  }
};
// f32 pair -> the 16-bit type's packed bits as two int16 (sign + monotonic magnitude: integer max == float max for >= 0)
template <typename T> __device__ __forceinline__ i16x2 half_bits(const f32x2& r) {
  return __builtin_bit_cast(i16x2, __builtin_convertvector(r, typename Half<T>::v2));
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

template <> struct Vec16<f16> {
  static constexpr int N = 8;
  __device__ __forceinline__ static void unpack(const uint4& u, float* f) {
    const f16x8 h = __builtin_bit_cast(f16x8, u);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (float)h[i];
  }
  __device__ __forceinline__ static uint4 pack(const float* f) {
    return make_uint4(Half<f16>::pack(f[0], f[1]), Half<f16>::pack(f[2], f[3]), Half<f16>::pack(f[4], f[5]), Half<f16>::pack(f[6], f[7]));
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
template <> struct PreAct<f16> {   // same recipe: fp32 FMA, one rounding, ReLU on the rounded halves
  static constexpr int NP = 4;
  __device__ __forceinline__ static uint4 apply(const uint4& v, const f32x2 (&s)[4], const f32x2 (&b)[4]) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x2 x = {Half<f16>::lo(w[i]), Half<f16>::hi(w[i])};
      const f32x2 r = __builtin_elementwise_fma(x, s[i], b[i]);
      o[i] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(half_bits<f16>(r), i16x2{0, 0}));
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
__device__ __forceinline__ float ld(const f16* p) { return (float)*p; }
__device__ __forceinline__ void st(f16* p, float v) { *p = (f16)v; }

__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + __expf(-x)); }

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == MDIE_ACT_RELU) return fmaxf(v, 0.0f);
  if (act == MDIE_ACT_SIGMOID) return sigmoidf(v);
  return v;
}

__host__ __device__ static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) -- needed above 64 KiB of dynamic LDS -- is a per-DEVICE setting of a
// kernel: one static instance per launch site remembers the devices it has been applied on (a process may drive several
// GPUs: RoutedEngine / CdanEngine on cuda:1 after cuda:0).  Returns false (error text set) when HIP refuses.
struct LdsOptIn {
  unsigned long long done = 0;   // bit per device ordinal; a lost race only repeats the idempotent call
  bool ensure(const void* kernel, int bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 63;
    if ((__atomic_load_n(&done, __ATOMIC_RELAXED) >> dev) & 1ull) return true;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) { set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize = %d) failed on device %d: %s", bytes, dev, hipGetErrorString(e)); return false; }
    if (dev != 63) __atomic_fetch_or(&done, 1ull << dev, __ATOMIC_RELAXED);
    return true;
  }
};

// ---- optional per-launch instrumentation (mdie_cdan_forward's launch_ms mode) ------------------------
struct LaunchTimer {
  hipStream_t stream = nullptr;
  hipEvent_t* ev = nullptr;  // 2 * cap events
  int* kind = nullptr;
  mdie_launch_info* info = nullptr;   // optional: labels + model shares, filled by the engine after each call (engine.hip: note)
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

// cbam.hip: mdie_cbam_fwd with a hook between its channel-pool pass and its last pass (used by engine.hip)
typedef int (*cbam_hook_fn)(void* ctx);
int cbam_fwd_hooked(const mdie_cbam_desc* d, hipStream_t stream, cbam_hook_fn before_last, void* ctx);

}  // namespace mdie
