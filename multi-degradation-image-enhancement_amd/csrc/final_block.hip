// decoder.final_dense as ONE launch (models/cdan.py:22-53,119,153-157):
//     base = bilinear_x2(t4)[:, :3] + x
//     g_l  = conv3x3(relu(bn_l(cat(base, g_0 .. g_{l-1}))))            l = 0..3   (DenseBlock layers, 16 channels each)
//     y    = sigmoid(conv1x1(relu(bn_t(cat(base, g_0 .. g_3)))))      (transition 67 -> 3)
// Nothing of the block ever reaches HBM: a workgroup computes one 16 x 8 output tile from the 24 x 16 base patch around it; the
// growth maps live in LDS, the halo rings of g0 / g1 / g2 (22x14, 20x12, 18x10) are recomputed by the neighbouring tiles.
//
// Why.  As a chain of four launches (updense0.hip, 3 x conv_thin.hip with the transition folded in) the block is 235 us of a
// 1 007 us step at B = 32, 256x256, bf16 and 1.04 GB of its 3.36 GB: every byte of it is a growth map or a partial sum written
// to HBM and read back (up to 3 x with an 18x18 halo) by the next launch.  Here the block reads t4 (8 bytes per low-resolution
// pixel) and x and writes y: ~45 MB per step.
//
// MEASURED (round 6, profiles/r06*_final_block_*, LEDGER round 6): bit-identical to the chain on every shape, and SLOWER -- 292-319 us against
// 235 us.  The traffic is gone; what binds instead is vector-instruction issue and per-wave latency chains: every layer pre-activates every input
// element (175 per output pixel before any halo), two waves per SIMD is all that 250 registers and 78 KB of LDS leave, and a wave runs at
// 8-10 cycles per instruction (SQ counters: 40 % issuing, 39 % in s_waitcnt / barriers, 21 % dependency stalls, the matrix pipe 22 % busy; LDS is
// not the limit).  The engine therefore runs the chain by default; MDIE_FWD_BLOCK_TAIL selects this kernel (A/B runs, tests, a starting point).
//
// LDS holds ACTIVATED operands, per consumer.  Layer l applies its OWN BatchNorm to every channel of its input
// (models/cdan.py:41-46), so relu(bn_l(g_j)) differs per (l, j).  The producer of g_j has the map in registers (4 channels of one
// pixel per lane = the MFMA accumulator layout): its epilogue rounds g_j to the storage type once (what the chain stores) and
// writes relu(bn_l(g_j)) for EVERY later layer l into that layer's own input image A_l -- 6 vector instructions + one 8-byte LDS
// write per consumer, no staging pass, no raw map.  Out-of-picture pixels are written as zeros (nn.Conv2d pads the ACTIVATED tensor).
//     A_1: base, g0         on 22 x 14     A_2: base, g0, g1    on 20 x 12     A_3: base, g0, g1, g2  on 18 x 10
// one 16-byte column (8 channels) per plane, planes 256-byte aligned (conflict-free ds_read_b128, see conv_thin.hip); A_3's two
// g2 planes reuse A_1's g0 planes (dead by then): 49.9 KB of planes, 78.5 KB with the rest -- two workgroups per CU, so one
// workgroup's epilogue / base phase (vector pipe, memory) runs under the other's matrix phase.
//
// The matrix work is conv_thin's: weights of layers 0..2 live in REGISTERS for the whole persistent run (layer 3's fragments are
// read from LDS, 9 per K chunk, right before the chunk's MFMAs: 188 registers of weights would not leave two waves per SIMD a
// working set), every per-channel constant and the transition's weight rows sit in a 4 KB LDS table; a wave
// owns a band of output rows of the 16-wide centre strip, so the operand fragment of input row r and column shift kw feeds the
// three output rows r, r-1, r-2; the halo columns left and right of the strip (2 x 3 / 2 / 1) are gathered into extra 16-pixel
// groups with per-lane addresses.  Per 16 x 8 tile: 40 + 135 + 216 + 144 MFMAs + the transition's 32.
//
// Arithmetic is the chain's, operation for operation -- the base rounded to the storage type, every growth map rounded once,
// every pre-activation one fused multiply-add rounded once, products accumulated in the same (chunk, tap) order on the same K
// slots, the transition's terms summed in the same order (base + g0, + g1, + g2, + g3) -- so y is BIT-IDENTICAL to the chain's
// (tests/test_gpu_parity.py::test_final_block_one_launch_equals_the_chain).
#include <algorithm>

#include "conv_common.hpp"

#pragma clang fp contract(off)   // (the interpolation must round like updense0.hip / resample.hip)

namespace mdie {

constexpr int FB_THREADS = 256;
constexpr int FB_TW = 16, FB_TH = 8;
constexpr int FB_RBW = FB_TW + 8, FB_RBH = FB_TH + 8;   // the base patch: 24 x 16

// input image of layer L = 1..3 (= the region layer L-1's output is needed on): halo K, one plane per 16-byte column
template <int L> struct FbIn {
  static constexpr int K = 4 - L, W = FB_TW + 2 * K, H = FB_TH + 2 * K, NCOL = 1 + 2 * L;
  static constexpr int PLANE = (W * H * 16 + 255) / 256 * 256;
};
constexpr int FB_A1 = 0;
constexpr int FB_A2 = FB_A1 + 3 * FbIn<1>::PLANE;
constexpr int FB_A3 = FB_A2 + 5 * FbIn<2>::PLANE;
constexpr int FB_A3_G2 = FB_A1 + FbIn<1>::PLANE;            // A_3's planes 5, 6 (g2) over A_1's planes 1, 2 (g0: last read by layer 1)
constexpr int FB_PATCH = FB_A3 + 5 * FbIn<3>::PLANE;        // relu(bn_0(base)), [24 x 16 pixels][4]: layer 0's im2col source
constexpr int FB_TRPATCH = FB_PATCH + FB_RBW * FB_RBH * 8;  // relu(bn_t(base)) of the tile's own pixels, [16 x 8][4]
constexpr int FB_TRP = FB_TRPATCH + FB_TW * FB_TH * 8;      // the transition's fp32 partial sums, [16 x 8] float4
constexpr int FB_W3 = FB_TRP + FB_TW * FB_TH * 16;          // layer 3's packed weights (2 chunks)
constexpr int FB_WCHUNK = 4 * 9 * 16 * 16;
// every per-channel constant of the block, copied once: the loop then needs no parameter pointer but lo, x and y (26 pointers in
// SGPRs spilled 36 of them and, with them, the weights)
constexpr int FB_CPS = FB_W3 + 2 * FB_WCHUNK;               // float [4 layers][72]: pre-activation scale by stored input channel
constexpr int FB_CPB = FB_CPS + 4 * 72 * 4;                 //                       ... shift
constexpr int FB_CTS = FB_CPB + 4 * 72 * 4;                 // float [72] + [72]: the transition's
constexpr int FB_CTB = FB_CTS + 72 * 4;
constexpr int FB_CES = FB_CTB + 72 * 4;                     // float [4][16] + [4][16]: the layers' epilogue scale / shift
constexpr int FB_CEB = FB_CES + 4 * 16 * 4;
constexpr int FB_CTE = FB_CEB + 4 * 16 * 4;                 // float [4] + [4]: the transition's epilogue
constexpr int FB_TRW = FB_CTE + 32;                         // rows 0..3 of the transition's weights: [9 K groups (8 stored channels)][4 rows][16 B]
constexpr int FB_DUMP = FB_TRW + 9 * 64;                    // 8 bytes per lane nobody reads: where a lane writes when its pixel lies outside a consumer's region
constexpr int FB_LDS = FB_DUMP + 64 * 8;                    // (ONE shared slot made every such write a 64-way bank conflict: round 6, 308 us)
static_assert(2 * FbIn<3>::PLANE <= 2 * FbIn<1>::PLANE, "g2's planes must fit A_1's g0 planes");
static_assert(FB_LDS <= 80 * 1024, "two workgroups per CU");

template <int L> __device__ __forceinline__ int fb_plane(int c) {   // byte offset of column c of A_L
  if constexpr (L == 1) return FB_A1 + c * FbIn<1>::PLANE;
  else if constexpr (L == 2) return FB_A2 + c * FbIn<2>::PLANE;
  else return c < 5 ? FB_A3 + c * FbIn<3>::PLANE : FB_A3_G2 + (c - 5) * FbIn<3>::PLANE;
}

struct FbArgs {
  int B, H, W, tiles_x, tiles_y;
  const char* lo; unsigned lo_stride;      // decoder.conv4's output, NHWC at H/2 x W/2; bytes per pixel
  const float* x;                          // fp32 NCHW
  const char* w0;                          // layer 0, mdie_pack_conv_first_weight layout
  const char* w[3];                        // layers 1..3, mdie_pack_conv_weight layout (16 stored outputs)
  const float* ps[4]; const float* pb[4];  // folded pre-activation BatchNorm of layers 0..3, by stored input channel (base 0..2, g_j at 8 + 16 j)
  const float* esc[4]; const float* esh[4];   // [16] each: the layers' epilogue (1, bias)
  const char* wt;                          // the transition's packed 1x1 weights
  const float* tps; const float* tpb;      // its folded BatchNorm by stored input channel
  const float* tes; const float* teb;      // its epilogue (1, bias), >= 3 entries
  float* y;                                // fp32 NCHW
  unsigned long long* dbg;                 // (diagnostic builds)
};

__device__ __forceinline__ void fb_src(int dst, int in_size, int& i0, int& i1, float& l0, float& l1) {   // = resample.hip src_index
  float src = ((float)dst + 0.5f) * 0.5f - 0.5f;
  src = src < 0.f ? 0.f : src;
  i0 = (int)src;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = src - (float)i0;
  l0 = 1.0f - l1;
}

// relu(x * s + b) of 4 channels, rounded to the storage type: PreAct<T>::apply, two dwords at a time
template <typename T> __device__ __forceinline__ uint2 fb_preact4(const f32x2& x0, const f32x2& x1, const float4& s, const float4& b) {
  const f32x2 r0 = __builtin_elementwise_fma(x0, f32x2{s.x, s.y}, f32x2{b.x, b.y});
  const f32x2 r1 = __builtin_elementwise_fma(x1, f32x2{s.z, s.w}, f32x2{b.z, b.w});
  const i16x2 m0 = __builtin_elementwise_max(half_bits<T>(r0), i16x2{0, 0}), m1 = __builtin_elementwise_max(half_bits<T>(r1), i16x2{0, 0});
  return make_uint2(__builtin_bit_cast(uint32_t, m0), __builtin_bit_cast(uint32_t, m1));
}

// the lane's 4 channels of g_J at pixel (ry, rx) of g_J's region -> column 1 + 2 J + lq / 2 of A_L, pre-activated with layer L's constants
template <typename T, int J, int L, bool MAIN>
__device__ __forceinline__ void fb_emit_one(char* smem, const f32x2& x0, const f32x2& x1, int ry, int rx, bool img_ok, bool lane_ok, int lq,
                                            const float4& s, const float4& b) {
  using R = FbIn<L>;
  constexpr int D = L - 1 - J;                  // A_L's region starts D pixels inside g_J's
  const int cy = ry - D, cx = rx - D;
  bool in = cy >= 0 && cy < R::H;               // (MAIN: wave-uniform; the centre strip's columns lie inside every region)
  if constexpr (!MAIN) in = in && lane_ok && cx >= 0 && cx < R::W;
  if (in) {
    uint2 v = fb_preact4<T>(x0, x1, s, b);
    if (!img_ok) v = make_uint2(0u, 0u);        // zero padding of the ACTIVATED tensor
    *reinterpret_cast<uint2*>(smem + fb_plane<L>(1 + 2 * J + (lq >> 1)) + (cy * R::W + cx) * 16 + (lq & 1) * 8) = v;
  }
}
template <typename T, int J, bool MAIN>
__device__ __forceinline__ void fb_emit(char* smem, uint32_t u0, uint32_t u1, int ry, int rx, bool img_ok, bool lane_ok, int lq,
                                        const float4 (&cs)[3], const float4 (&cb)[3]) {
  const f32x2 x0 = {Half<T>::lo(u0), Half<T>::hi(u0)}, x1 = {Half<T>::lo(u1), Half<T>::hi(u1)};
  if constexpr (J < 1) fb_emit_one<T, J, 1, MAIN>(smem, x0, x1, ry, rx, img_ok, lane_ok, lq, cs[0], cb[0]);
  if constexpr (J < 2) fb_emit_one<T, J, 2, MAIN>(smem, x0, x1, ry, rx, img_ok, lane_ok, lq, cs[1], cb[1]);
  if constexpr (J < 3) fb_emit_one<T, J, 3, MAIN>(smem, x0, x1, ry, rx, img_ok, lane_ok, lq, cs[2], cb[2]);
}

// a band of NROWS output rows of layer L's centre strip (conv_thin's matrix phase): xb[k] = the lane's byte offset of input row
// row0, column shift 0, K group lq of chunk k
template <typename T, int L, int NCHUNK, int NROWS>
__device__ __forceinline__ void fb_band(const char* smem, const int (&xb)[NCHUNK], const uint4 (&w)[NCHUNK][9], f32x4 (&acc)[NROWS]) {
  constexpr int PW = FbIn<L>::W;
#pragma unroll
  for (int ps = 0; ps < NROWS; ++ps) acc[ps] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < NCHUNK; ++k) {
    const char* const xk = smem + xb[k];
    uint4 xf[2][3];
    auto read_row = [&](int r, int b) {
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) xf[b][kw] = *reinterpret_cast<const uint4*>(xk + (r * PW + kw) * 16);
    };
    read_row(0, 0);
#pragma unroll
    for (int r = 0; r < NROWS + 2; ++r) {
      if (r + 1 < NROWS + 2) read_row(r + 1, (r + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int kh = 2; kh >= 0; --kh) {
          const int ps = r - kh;
          if (ps >= 0 && ps < NROWS) acc[ps] = mma16<T>(w[k][kh * 3 + kw], xf[r & 1][kw], acc[ps]);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// the same with the chunk's 9 weight fragments read from LDS (offset wl = the lane's fragment of chunk 0, tap 0) in front of its MFMAs
template <typename T, int L, int NCHUNK, int NROWS>
__device__ __forceinline__ void fb_band_wlds(const char* smem, const int (&xb)[NCHUNK], int wl, f32x4 (&acc)[NROWS]) {
  constexpr int PW = FbIn<L>::W;
#pragma unroll
  for (int ps = 0; ps < NROWS; ++ps) acc[ps] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < NCHUNK; ++k) {
    uint4 w[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) w[tap] = *reinterpret_cast<const uint4*>(smem + wl + k * FB_WCHUNK + tap * 256);
    const char* const xk = smem + xb[k];
    uint4 xf[2][3];
    auto read_row = [&](int r, int b) {
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) xf[b][kw] = *reinterpret_cast<const uint4*>(xk + (r * PW + kw) * 16);
    };
    read_row(0, 0);
#pragma unroll
    for (int r = 0; r < NROWS + 2; ++r) {
      if (r + 1 < NROWS + 2) read_row(r + 1, (r + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int kh = 2; kh >= 0; --kh) {
          const int ps = r - kh;
          if (ps >= 0 && ps < NROWS) acc[ps] = mma16<T>(w[kh * 3 + kw], xf[r & 1][kw], acc[ps]);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// one 16-pixel group of halo columns: xg[k] = the lane's byte offset of ITS pixel's tap (0, 0), K group lq of chunk k
template <typename T, int L, int NCHUNK>
__device__ __forceinline__ f32x4 fb_group(const char* smem, const int (&xg)[NCHUNK], const uint4 (&w)[NCHUNK][9]) {
  constexpr int PW = FbIn<L>::W;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < NCHUNK; ++k)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const uint4 xv = *reinterpret_cast<const uint4*>(smem + xg[k] + ((tap / 3) * PW + tap % 3) * 16);
      acc = mma16<T>(w[k][tap], xv, acc);
    }
  return acc;
}

// rows of the transition's weights for the 4 channels (stored input channel c0 + 4 lq ..) of a segment: A operand rows 0..2
// (lane lp < 4 holds row lp; row 3 of the packed weights is zero), zeros elsewhere
__device__ __forceinline__ uint2 fb_tr_rows(const char* smem, int c0, int lq, int lp) {
  const int c = c0 + 4 * lq;
  const uint2 w = *reinterpret_cast<const uint2*>(smem + FB_TRW + (c >> 3) * 64 + (lp & 3) * 16 + (c & 7) * 2);
  return lp < 4 ? w : make_uint2(0u, 0u);
}
__device__ __forceinline__ float4 fb_c4(const char* smem, int table, int index) { return *reinterpret_cast<const float4*>(smem + table + index * 4); }
__device__ __forceinline__ float4 fb_c3(const char* smem, int table, int index) {   // 3 channels of the base: the group's 4th (a pad channel) counts as 0
  float4 v = *reinterpret_cast<const float4*>(smem + table + index * 4);
  v.w = 0.f;
  return v;
}

#ifdef EXP_FBSTAMPS   // diagnostic build only (tools/stamp_final.py): per-phase shader-clock sums of every wave, summed over its tiles
static unsigned long long* g_fb_dbg = nullptr;
#define FSEG(k) do { if (dbg) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); if ((k) >= 0) facc[(k) < 0 ? 0 : (k)] += t_ - fprev; fprev = t_; } } while (0)
#else
#define FSEG(k) do {} while (0)
#endif

// Lane / wave indices laundered through an empty asm at the top of every phase: everything a phase derives from them (plane offsets,
// halo-group pixels, gather offsets) is tile-invariant, and hoisted out of the tile loop it stayed live across ALL phases -- next to
// 116 registers of weights that meant scratch spills.  Recomputing a few integer instructions per phase is cheaper.
#define FB_FRESH_IDS()                                                                       \
  int lqf_ = lq_, lpf_ = lp_, tidf_ = tid_, wavef_ = wave_;                                  \
  asm volatile("" : "+v"(lqf_), "+v"(lpf_), "+v"(tidf_));                                    \
  asm volatile("" : "+s"(wavef_));                                                           \
  const int lq = lqf_, lp = lpf_, tid = tidf_, wave = wavef_;                                \
  (void)lq; (void)lp; (void)tid; (void)wave

template <typename T>
__global__ __launch_bounds__(FB_THREADS, 2) void final_block_kernel(const FbArgs a, const int n_items) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid_ = threadIdx.x, lane_ = tid_ & 63, lq_ = lane_ >> 4, lp_ = lane_ & 15;
  const int wave_ = __builtin_amdgcn_readfirstlane(tid_ >> 6);

  // ---- this workgroup's run of tiles: contiguous, inside its XCD's share (conv_thin.hip) ----
  const int per_xcd = (n_items + 7) >> 3, nwg = gridDim.x >> 3;
  const int band0 = ((int)blockIdx.x & 7) * per_xcd, band1 = min(band0 + per_xcd, n_items);
  const int run = (per_xcd + nwg - 1) / nwg;
  int item = band0 + ((int)blockIdx.x >> 3) * run;
  const int run_end = min(item + run, band1);
  if (item >= run_end) return;
#ifdef EXP_FBSTAMPS
  unsigned long long* const dbg = a.dbg;
  unsigned long long facc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, fprev = 0, fstart = 0, rstart = 0;
  if (dbg) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(fstart), "=s"(rstart) :: "memory");
  const int item_first = item;
#endif

  uint4 wf0[2], w1r[1][9], w2r[2][9];
  {
    FB_FRESH_IDS();
    const int wlane = (lq * 9 * 16 + lp) * 16;
    // ---- once: the base planes (their upper 8 bytes per pixel -- channels 4..7 of the base group -- stay zero for the whole run),
    //      layer 3's weights -> LDS, layers 0..2 -> registers (A row lp = output channel, K group lq) ----
    for (int i = tid; i < FbIn<1>::PLANE / 16; i += FB_THREADS) *reinterpret_cast<uint4*>(smem + fb_plane<1>(0) + i * 16) = make_uint4(0u, 0u, 0u, 0u);
    for (int i = tid; i < FbIn<2>::PLANE / 16; i += FB_THREADS) *reinterpret_cast<uint4*>(smem + fb_plane<2>(0) + i * 16) = make_uint4(0u, 0u, 0u, 0u);
    for (int i = tid; i < FbIn<3>::PLANE / 16; i += FB_THREADS) *reinterpret_cast<uint4*>(smem + fb_plane<3>(0) + i * 16) = make_uint4(0u, 0u, 0u, 0u);
    for (int i = tid; i < 2 * FB_WCHUNK / 16; i += FB_THREADS) *reinterpret_cast<uint4*>(smem + FB_W3 + i * 16) = *reinterpret_cast<const uint4*>(a.w[2] + i * 16);
    if (tid < 72) {
      float* const cps = reinterpret_cast<float*>(smem + FB_CPS), * const cpb = reinterpret_cast<float*>(smem + FB_CPB);
      cps[tid] = tid < 8 ? a.ps[0][tid] : 0.f; cpb[tid] = tid < 8 ? a.pb[0][tid] : 0.f;
      cps[72 + tid] = tid < 24 ? a.ps[1][tid] : 0.f; cpb[72 + tid] = tid < 24 ? a.pb[1][tid] : 0.f;
      cps[144 + tid] = tid < 40 ? a.ps[2][tid] : 0.f; cpb[144 + tid] = tid < 40 ? a.pb[2][tid] : 0.f;
      cps[216 + tid] = tid < 56 ? a.ps[3][tid] : 0.f; cpb[216 + tid] = tid < 56 ? a.pb[3][tid] : 0.f;
      reinterpret_cast<float*>(smem + FB_CTS)[tid] = a.tps[tid]; reinterpret_cast<float*>(smem + FB_CTB)[tid] = a.tpb[tid];
    } else if (tid < 72 + 16) {
      const int c = tid - 72;
      float* const ces = reinterpret_cast<float*>(smem + FB_CES), * const ceb = reinterpret_cast<float*>(smem + FB_CEB);
      ces[c] = a.esc[0][c]; ces[16 + c] = a.esc[1][c]; ces[32 + c] = a.esc[2][c]; ces[48 + c] = a.esc[3][c];
      ceb[c] = a.esh[0][c]; ceb[16 + c] = a.esh[1][c]; ceb[32 + c] = a.esh[2][c]; ceb[48 + c] = a.esh[3][c];
    } else if (tid < 72 + 16 + 4) {
      const int c = tid - 88;
      reinterpret_cast<float*>(smem + FB_CTE)[c] = c < 3 ? a.tes[c] : 0.f; reinterpret_cast<float*>(smem + FB_CTE)[4 + c] = c < 3 ? a.teb[c] : 0.f;
    } else if (tid >= 96 && tid < 96 + 9 * 16) {   // rows 0..3 of every K group of the transition's weights (K group g = stored channels 8 g .. 8 g + 7 at byte 256 g)
      const int i = tid - 96, g = i >> 4, rw = (i >> 2) & 3, dwi = i & 3;
      *reinterpret_cast<uint32_t*>(smem + FB_TRW + g * 64 + rw * 16 + dwi * 4) = *reinterpret_cast<const uint32_t*>(a.wt + g * 256 + rw * 16 + dwi * 4);
    }
  #pragma unroll
    for (int s = 0; s < 2; ++s) wf0[s] = *reinterpret_cast<const uint4*>(a.w0 + ((size_t)s * 16 + lp) * 64 + lq * 16);
  #pragma unroll
    for (int tap = 0; tap < 9; ++tap) w1r[0][tap] = *reinterpret_cast<const uint4*>(a.w[0] + wlane + tap * 256);
  #pragma unroll
    for (int k = 0; k < 2; ++k)
  #pragma unroll
      for (int tap = 0; tap < 9; ++tap) w2r[k][tap] = *reinterpret_cast<const uint4*>(a.w[1] + (size_t)k * FB_WCHUNK + wlane + tap * 256);

  }
  const int Hl = a.H >> 1, Wl = a.W >> 1;
  const size_t hw = (size_t)a.H * a.W;
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};

  int tx, ty, img;
  { int r = item; tx = r % a.tiles_x; r /= a.tiles_x; ty = r % a.tiles_y; img = r / a.tiles_y; }
  __syncthreads();


  // ===================================================================================================================
  // INTERIOR tiles (the whole 24 x 16 base patch inside the picture: 82 % of the tiles of a 256 x 256 picture).  Same arithmetic as
  // the general path below, none of its border logic, and everything a thread can know before the tile exists is hoisted or made
  // compile-time: version 1 of this kernel (one path for all tiles) ran 313 us against 235 us for the chain -- issue-bound, 2 540
  // instructions per wave and tile of which 143 MFMAs (profiles/r06c_final_block_v1_stamps.txt).  Here:
  //   * no in-picture predicates, no selects of zeros, unconditional loads; the bilinear weights are the pixel's parity (0.25 / 0.75);
  //   * rows are dealt to waves so that wave w owns rows 2 w, 2 w + 1 of the tile in EVERY layer: the transition's running sums of
  //     those two rows stay in registers (lanes 0..15 / 16..31 through two A-operand row slots), no LDS read-modify-write;
  //     the halo rows above / below go to waves 0 / 3, the halo-column groups to waves 1 / 2;
  //   * a pixel outside a consumer's region is written to a dump slot (one select) instead of branched around.
  // ===================================================================================================================
  // (Requesting the NEXT tile's base inputs at the start of layer 3 was built and measured: 292 -> 308 us.  Issuing the 14 scattered 8-byte / 4-byte
  //  loads costs the two waves with two patch pixels ~1.2 k cycles wherever it is placed -- the vector memory path, not the wait, is what the
  //  base phase pays for: profiles/r06f_final_block_v2_stamps.txt.)
  auto issue_base_loads = [&](const int img, const int y0, const int x0, uint2 (&ptap)[2][4], float (&pxin)[2][3]) {
    const int tid = tid_;      // (NOT laundered: a thread's patch pixel and its tap offsets are tile-invariant and meant to be hoisted)
    const unsigned ls = a.lo_stride;
    const char* const lb = a.lo + ((size_t)(img * Hl + ((y0 - 4) >> 1)) * Wl + ((x0 - 4) >> 1)) * ls;   // scalar: tap (0, 0) of patch pixel (1, 1)
    const float* const xb = a.x + (size_t)img * 3 * hw + (size_t)(y0 - 4) * a.W + (x0 - 4);             // scalar: patch pixel (0, 0)
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      if (it == 0 || tid < FB_RBW * FB_RBH - FB_THREADS) {
        const int p = tid + it * FB_THREADS, py = p / FB_RBW, px = p - py * FB_RBW;
        const int lo_off = (((py - 1) >> 1) * Wl + ((px - 1) >> 1)) * (int)ls;     // (may be negative: the patch starts one low-resolution pixel before lb)
        const char* const q = lb + lo_off;
        ptap[it][0] = *reinterpret_cast<const uint2*>(q); ptap[it][1] = *reinterpret_cast<const uint2*>(q + ls);
        ptap[it][2] = *reinterpret_cast<const uint2*>(q + (size_t)Wl * ls); ptap[it][3] = *reinterpret_cast<const uint2*>(q + (size_t)Wl * ls + ls);
        const float* const xq = xb + py * a.W + px;
        pxin[it][0] = xq[0]; pxin[it][1] = xq[hw]; pxin[it][2] = xq[2 * hw];
      }
    }
  };
  auto tile_interior = [&](const int img, const int y0, const int x0) {
    float fbase[2][3];
    f32x4 tsum = zero4;          // lanes (lq < 2, lp): transition outputs 0..2 of pixel (row 2 wave + lq, column lp), summed over the segments so far
    // ---- P0 ----
    {
      const int tid = tid_;      // (NOT laundered: a thread's patch pixel and its LDS slots are tile-invariant and meant to be hoisted)
      const int dump = FB_DUMP + (tid & 63) * 8;
      uint2 ptap[2][4];
      float pxin[2][3];
      issue_base_loads(img, y0, x0, ptap, pxin);
      const float4 c0s = fb_c3(smem, FB_CPS, 0), c0b = fb_c3(smem, FB_CPB, 0), c1s = fb_c3(smem, FB_CPS, 72), c1b = fb_c3(smem, FB_CPB, 72);
      const float4 c2s = fb_c3(smem, FB_CPS, 144), c2b = fb_c3(smem, FB_CPB, 144), cts = fb_c3(smem, FB_CTS, 0), ctb = fb_c3(smem, FB_CTB, 0);
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        if (it == 0 || tid < FB_RBW * FB_RBH - FB_THREADS) {
          const int p = tid + it * FB_THREADS, py = p / FB_RBW, px = p - py * FB_RBW;
          // src = (dst + 0.5) / 2 - 0.5 away from the picture's edge: the upper / left tap weighs 0.25 on even, 0.75 on odd pixels (fb_src's values)
          const float hy1 = (py & 1) ? 0.25f : 0.75f, hy0 = 1.0f - hy1, wx1 = (px & 1) ? 0.25f : 0.75f, wx0 = 1.0f - wx1;
          float f[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float t0 = c == 0 ? Half<T>::lo(ptap[it][0].x) : c == 1 ? Half<T>::hi(ptap[it][0].x) : Half<T>::lo(ptap[it][0].y);
            const float t1 = c == 0 ? Half<T>::lo(ptap[it][1].x) : c == 1 ? Half<T>::hi(ptap[it][1].x) : Half<T>::lo(ptap[it][1].y);
            const float t2 = c == 0 ? Half<T>::lo(ptap[it][2].x) : c == 1 ? Half<T>::hi(ptap[it][2].x) : Half<T>::lo(ptap[it][2].y);
            const float t3 = c == 0 ? Half<T>::lo(ptap[it][3].x) : c == 1 ? Half<T>::hi(ptap[it][3].x) : Half<T>::lo(ptap[it][3].y);
            f[c] = hy0 * (wx0 * t0 + wx1 * t1) + hy1 * (wx0 * t2 + wx1 * t3) + pxin[it][c];
            f[c] = (float)(T)f[c];
            fbase[it][c] = f[c];
          }
          *reinterpret_cast<uint2*>(smem + FB_PATCH + p * 8) =
              make_uint2(Half<T>::pack(fmaxf(fmaf(f[0], c0s.x, c0b.x), 0.f), fmaxf(fmaf(f[1], c0s.y, c0b.y), 0.f)), Half<T>::pack(fmaxf(fmaf(f[2], c0s.z, c0b.z), 0.f), 0.f));
          const f32x2 x01 = {f[0], f[1]}, x2 = {f[2], 0.f};
          const bool in1 = (unsigned)(py - 1) < (unsigned)FbIn<1>::H && (unsigned)(px - 1) < (unsigned)FbIn<1>::W;
          const bool in2 = (unsigned)(py - 2) < (unsigned)FbIn<2>::H && (unsigned)(px - 2) < (unsigned)FbIn<2>::W;
          const bool in4 = (unsigned)(py - 4) < (unsigned)FB_TH && (unsigned)(px - 4) < (unsigned)FB_TW;
          *reinterpret_cast<uint2*>(smem + (in1 ? fb_plane<1>(0) + ((py - 1) * FbIn<1>::W + (px - 1)) * 16 : dump)) = fb_preact4<T>(x01, x2, c1s, c1b);
          *reinterpret_cast<uint2*>(smem + (in2 ? fb_plane<2>(0) + ((py - 2) * FbIn<2>::W + (px - 2)) * 16 : dump)) = fb_preact4<T>(x01, x2, c2s, c2b);
          *reinterpret_cast<uint2*>(smem + (in4 ? FB_TRPATCH + ((py - 4) * FB_TW + (px - 4)) * 8 : dump)) =
              make_uint2(Half<T>::pack(fmaxf(fmaf(f[0], cts.x, ctb.x), 0.f), fmaxf(fmaf(f[1], cts.y, ctb.y), 0.f)), Half<T>::pack(fmaxf(fmaf(f[2], cts.z, ctb.z), 0.f), 0.f));
        }
      }
    }
    FSEG(0);
    __syncthreads();   // B1
    FSEG(1);
    // ---- P1: layer 0 ----
    {
      {   // column 0 of A_3 (layer 3 of the previous tile read it until B1)
        const int tid = tid_;
        const int dump = FB_DUMP + (tid & 63) * 8;
        const float4 c3s = fb_c3(smem, FB_CPS, 216), c3b = fb_c3(smem, FB_CPB, 216);
#pragma unroll
        for (int it = 0; it < 2; ++it)
          if (it == 0 || tid < FB_RBW * FB_RBH - FB_THREADS) {
            const int p = tid + it * FB_THREADS, py = p / FB_RBW, px = p - py * FB_RBW;
            const bool in3 = (unsigned)(py - 3) < (unsigned)FbIn<3>::H && (unsigned)(px - 3) < (unsigned)FbIn<3>::W;
            *reinterpret_cast<uint2*>(smem + (in3 ? fb_plane<3>(0) + ((py - 3) * FbIn<3>::W + (px - 3)) * 16 : dump)) =
                fb_preact4<T>(f32x2{fbase[it][0], fbase[it][1]}, f32x2{fbase[it][2], 0.f}, c3s, c3b);
          }
      }
      FB_FRESH_IDS();
      const int goA = (((2 * lq) / 3) * FB_RBW + (2 * lq) % 3) * 8, goB = (((2 * lq + 1) / 3) * FB_RBW + (2 * lq + 1) % 3) * 8, goC = (2 * FB_RBW + 2) * 8;
      float4 cs[3], cb[3];
#pragma unroll
      for (int L = 1; L <= 3; ++L) { cs[L - 1] = fb_c4(smem, FB_CPS, 72 * L + 8 + 4 * lq); cb[L - 1] = fb_c4(smem, FB_CPB, 72 * L + 8 + 4 * lq); }
      const float4 ts = fb_c4(smem, FB_CTS, 8 + 4 * lq), tb = fb_c4(smem, FB_CTB, 8 + 4 * lq);
      const float4 bias = fb_c4(smem, FB_CEB, 4 * lq);
      uint4 trw;   // the transition's rows (lp & 3) for g0's channels 4 lq .. and, in K group 0, the base's
      {
        const int c = 8 + 4 * lq;
        const uint2 wg = *reinterpret_cast<const uint2*>(smem + FB_TRW + (c >> 3) * 64 + (lp & 3) * 16 + (c & 7) * 2);
        const uint2 wb = *reinterpret_cast<const uint2*>(smem + FB_TRW + (lp & 3) * 16);
        trw = make_uint4(wg.x, wg.y, lq == 0 ? wb.x : 0u, lq == 0 ? wb.y : 0u);
      }
      // the lane's column of g0 (2 planes per growth map: lq >> 1; 8-byte half: lq & 1) in the three images, at the centre strip's pixel lp of row 0
      const int half = (lq & 1) * 8;
      const int e1 = fb_plane<1>(1 + (lq >> 1)) + half + (3 + lp) * 16, e2 = fb_plane<2>(1 + (lq >> 1)) + half + (2 + lp) * 16, e3 = fb_plane<3>(1 + (lq >> 1)) + half + (1 + lp) * 16;
      f32x4 ta = zero4;
      // Five units per wave.  Waves 0 / 3: rows 0..4 (taken 4, 3, .. 0) / 9..13 of the centre strip -- the two tile rows first, then the halo
      // rows outwards, so unit u means the same thing in both; waves 1 / 2: rows 5, 6 / 7, 8 and three groups of halo columns (0..2, 19..21 of
      // all 14 rows: 84 pixels in 6 groups).  Per unit: the window corner in the patch and the slot of the lane's 8 bytes in A_1, A_2, A_3
      // (the dump slot where the pixel lies outside the image of that layer); units 0, 1 are rows of the tile itself.
      // ALL reads of the phase, then all MFMAs, then the epilogues: LDS operations keep their program order (one address space to the
      // compiler), so a unit-by-unit loop paid a full read -> MFMA -> convert -> write latency chain per unit (version 1: ~800 cycles each).
      int vb[5], a1[5], a2[5], a3[5], crow[2];
      const int dump = FB_DUMP + (tid & 63) * 8;
      const int plane_lane = FB_PATCH + (3 + lp) * 8;
      if (wave == 0 || wave == 3) {
#pragma unroll
        for (int u = 0; u < 5; ++u) {
          const int row = wave == 0 ? 4 - u : 9 + u;
          vb[u] = plane_lane + row * (FB_RBW * 8);
          a1[u] = e1 + row * (FbIn<1>::W * 16);
          a2[u] = u <= 3 ? e2 + (row - 1) * (FbIn<2>::W * 16) : dump;
          a3[u] = u <= 2 ? e3 + (row - 2) * (FbIn<3>::W * 16) : dump;
          if (u < 2) crow[u] = row - 3;
        }
      } else {
        const int pl = (1 + (lq >> 1));
#pragma unroll
        for (int u = 0; u < 5; ++u) {
          if (u < 2) {
            const int row = 3 + 2 * wave + u;
            vb[u] = plane_lane + row * (FB_RBW * 8);
            a1[u] = e1 + row * (FbIn<1>::W * 16); a2[u] = e2 + (row - 1) * (FbIn<2>::W * 16); a3[u] = e3 + (row - 2) * (FbIn<3>::W * 16);
            crow[u] = 2 * wave + u;
          } else {
            const int q = 16 * (3 * (wave - 1) + u - 2) + lp;
            const bool ok = q < 6 * FbIn<1>::H;
            const int qq = ok ? q : 0, ry = qq / 6, cc = qq - 6 * ry, rx = cc < 3 ? cc : cc + 16;
            const bool in2 = ok && (unsigned)(ry - 1) < (unsigned)FbIn<2>::H && (unsigned)(rx - 1) < (unsigned)FbIn<2>::W;
            const bool in3 = ok && (unsigned)(ry - 2) < (unsigned)FbIn<3>::H && (unsigned)(rx - 2) < (unsigned)FbIn<3>::W;
            vb[u] = FB_PATCH + (ry * FB_RBW + rx) * 8;
            a1[u] = ok ? fb_plane<1>(pl) + half + (ry * FbIn<1>::W + rx) * 16 : dump;
            a2[u] = in2 ? fb_plane<2>(pl) + half + ((ry - 1) * FbIn<2>::W + rx - 1) * 16 : dump;
            a3[u] = in3 ? fb_plane<3>(pl) + half + ((ry - 2) * FbIn<3>::W + rx - 2) * 16 : dump;
          }
        }
      }
      uint2 pa[5], pb[5], pc[5], bv[2];
#pragma unroll
      for (int u = 0; u < 5; ++u) {
        pa[u] = *reinterpret_cast<const uint2*>(smem + vb[u] + goA); pb[u] = *reinterpret_cast<const uint2*>(smem + vb[u] + goB); pc[u] = *reinterpret_cast<const uint2*>(smem + vb[u] + goC);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) bv[u] = *reinterpret_cast<const uint2*>(smem + FB_TRPATCH + (crow[u] * FB_TW + lp) * 8);
      f32x4 acc[5];
#pragma unroll
      for (int u = 0; u < 5; ++u) acc[u] = mma16<T>(wf0[0], make_uint4(pa[u].x, pa[u].y, pb[u].x, pb[u].y), zero4);
#pragma unroll
      for (int u = 0; u < 5; ++u) acc[u] = mma16<T>(wf0[1], make_uint4(pc[u].x, pc[u].y, pc[u].x, pc[u].y), acc[u]);
#pragma unroll
      for (int u = 0; u < 5; ++u) {
        const uint32_t u0 = Half<T>::pack(acc[u][0] + bias.x, acc[u][1] + bias.y), u1 = Half<T>::pack(acc[u][2] + bias.z, acc[u][3] + bias.w);
        const f32x2 x0_ = {Half<T>::lo(u0), Half<T>::hi(u0)}, x1_ = {Half<T>::lo(u1), Half<T>::hi(u1)};
        *reinterpret_cast<uint2*>(smem + a1[u]) = fb_preact4<T>(x0_, x1_, cs[0], cb[0]);
        *reinterpret_cast<uint2*>(smem + a2[u]) = fb_preact4<T>(x0_, x1_, cs[1], cb[1]);
        *reinterpret_cast<uint2*>(smem + a3[u]) = fb_preact4<T>(x0_, x1_, cs[2], cb[2]);
        if (u < 2) {   // a row of the tile: the transition's terms of base + g0 into row slot crow & 1
          const uint2 tv = fb_preact4<T>(x0_, x1_, ts, tb);
          const bool mine = (lp >> 2) == (crow[u] & 1);
          ta = mma16<T>(make_uint4(mine ? trw.x : 0u, mine ? trw.y : 0u, mine ? trw.z : 0u, mine ? trw.w : 0u), make_uint4(tv.x, tv.y, bv[u].x, bv[u].y), ta);
        }
      }
      tsum = ta;
    }
    FSEG(2);
    __syncthreads();   // B2
    FSEG(3);
    // ---- P2: layer 1 ----
    {
      FB_FRESH_IDS();
      float4 cs[3], cb[3];
      cs[0] = cb[0] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int L = 2; L <= 3; ++L) { cs[L - 1] = fb_c4(smem, FB_CPS, 72 * L + 24 + 4 * lq); cb[L - 1] = fb_c4(smem, FB_CPB, 72 * L + 24 + 4 * lq); }
      const float4 ts = fb_c4(smem, FB_CTS, 24 + 4 * lq), tb = fb_c4(smem, FB_CTB, 24 + 4 * lq);
      const float4 esc = fb_c4(smem, FB_CES, 16 + 4 * lq), esh = fb_c4(smem, FB_CEB, 16 + 4 * lq);
      const uint2 trw = *reinterpret_cast<const uint2*>(smem + FB_TRW + ((24 + 4 * lq) >> 3) * 64 + (lp & 3) * 16 + ((24 + 4 * lq) & 7) * 2);
      const int half = (lq & 1) * 8, pl = 3 + (lq >> 1);
      const int e2 = fb_plane<2>(pl) + half + (2 + lp) * 16, e3 = fb_plane<3>(pl) + half + (1 + lp) * 16;
      f32x4 ta = zero4;
      auto finish = [&](const f32x4& acc, f32x2& x0_, f32x2& x1_) {
        const f32x2 lo = __builtin_elementwise_fma(f32x2{acc[0], acc[1]}, f32x2{esc.x, esc.y}, f32x2{esh.x, esh.y});
        const f32x2 hi = __builtin_elementwise_fma(f32x2{acc[2], acc[3]}, f32x2{esc.z, esc.w}, f32x2{esh.z, esh.w});
        const uint32_t u0 = __builtin_bit_cast(uint32_t, half_bits<T>(lo)), u1 = __builtin_bit_cast(uint32_t, half_bits<T>(hi));
        x0_ = f32x2{Half<T>::lo(u0), Half<T>::hi(u0)}; x1_ = f32x2{Half<T>::lo(u1), Half<T>::hi(u1)};
      };
      auto tr_term = [&](const f32x2& x0_, const f32x2& x1_, int slot) {
        const uint2 tv = fb_preact4<T>(x0_, x1_, ts, tb);
        const bool mine = (lp >> 2) == slot;
        ta = mma16<T>(make_uint4(mine ? trw.x : 0u, mine ? trw.y : 0u, 0u, 0u), make_uint4(tv.x, tv.y, 0u, 0u), ta);
      };
      if (wave == 0 || wave == 3) {   // rows 0..3 / 8..11 of the centre strip
        const int row0 = wave == 0 ? 0 : 8;
        const int xb[1] = {fb_plane<1>(min(lq, 2)) + (row0 * FbIn<1>::W + 2 + lp) * 16};
        f32x4 acc[4];
        fb_band<T, 1, 1, 4>(smem, xb, w1r, acc);
        FSEG(4);
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
          const int row = row0 + ps;
          f32x2 x0_, x1_;
          finish(acc[ps], x0_, x1_);
          *reinterpret_cast<uint2*>(smem + e2 + row * (FbIn<2>::W * 16)) = fb_preact4<T>(x0_, x1_, cs[1], cb[1]);
          if (wave == 0 ? ps >= 1 : ps <= 2) *reinterpret_cast<uint2*>(smem + e3 + (row - 1) * (FbIn<3>::W * 16)) = fb_preact4<T>(x0_, x1_, cs[2], cb[2]);
          if (wave == 0 ? ps >= 2 : ps <= 1) tr_term(x0_, x1_, ps & 1);
        }
      } else {                        // rows 4, 5 / 6, 7 and two groups of halo columns (0, 1, 18, 19 of 12 rows: 48 pixels in 3 groups; wave 2's second group is empty)
        const int row0 = 2 + 2 * wave;
        const int xb[1] = {fb_plane<1>(min(lq, 2)) + (row0 * FbIn<1>::W + 2 + lp) * 16};
        int g2a[2], g3a[2], xg[2][1];
        const int dump = FB_DUMP + (tid & 63) * 8;
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
          const int q = 16 * (2 * (wave - 1) + gi) + lp;
          const bool ok = q < 4 * FbIn<2>::H;
          const int qq = ok ? q : 0, ry = qq >> 2, cc = qq & 3, rx = cc < 2 ? cc : cc + 16;
          const bool in3 = ok && (unsigned)(ry - 1) < (unsigned)FbIn<3>::H && (unsigned)(rx - 1) < (unsigned)FbIn<3>::W;
          xg[gi][0] = fb_plane<1>(min(lq, 2)) + (ry * FbIn<1>::W + rx) * 16;
          g2a[gi] = ok ? fb_plane<2>(pl) + half + (ry * FbIn<2>::W + rx) * 16 : dump;
          g3a[gi] = in3 ? fb_plane<3>(pl) + half + ((ry - 1) * FbIn<3>::W + rx - 1) * 16 : dump;
        }
        f32x4 acc[2], ga[2];
        fb_band<T, 1, 1, 2>(smem, xb, w1r, acc);
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) ga[gi] = fb_group<T, 1, 1>(smem, xg[gi], w1r);
        FSEG(4);
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
          const int row = row0 + ps;
          f32x2 x0_, x1_;
          finish(acc[ps], x0_, x1_);
          *reinterpret_cast<uint2*>(smem + e2 + row * (FbIn<2>::W * 16)) = fb_preact4<T>(x0_, x1_, cs[1], cb[1]);
          *reinterpret_cast<uint2*>(smem + e3 + (row - 1) * (FbIn<3>::W * 16)) = fb_preact4<T>(x0_, x1_, cs[2], cb[2]);
          tr_term(x0_, x1_, ps);
        }
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
          f32x2 x0_, x1_;
          finish(ga[gi], x0_, x1_);
          *reinterpret_cast<uint2*>(smem + g2a[gi]) = fb_preact4<T>(x0_, x1_, cs[1], cb[1]);
          *reinterpret_cast<uint2*>(smem + g3a[gi]) = fb_preact4<T>(x0_, x1_, cs[2], cb[2]);
        }
      }
      tsum = f32x4{tsum[0] + ta[0], tsum[1] + ta[1], tsum[2] + ta[2], 0.f};
    }
    FSEG(9);
    __syncthreads();   // B3
    FSEG(5);
    // ---- P3: layer 2 ----
    {
      FB_FRESH_IDS();
      const float4 c3s = fb_c4(smem, FB_CPS, 216 + 40 + 4 * lq), c3b = fb_c4(smem, FB_CPB, 216 + 40 + 4 * lq);
      const float4 ts = fb_c4(smem, FB_CTS, 40 + 4 * lq), tb = fb_c4(smem, FB_CTB, 40 + 4 * lq);
      const float4 esc = fb_c4(smem, FB_CES, 32 + 4 * lq), esh = fb_c4(smem, FB_CEB, 32 + 4 * lq);
      const uint2 trw = *reinterpret_cast<const uint2*>(smem + FB_TRW + ((40 + 4 * lq) >> 3) * 64 + (lp & 3) * 16 + ((40 + 4 * lq) & 7) * 2);
      const int half = (lq & 1) * 8, pl = 5 + (lq >> 1);
      const int e3 = fb_plane<3>(pl) + half + (1 + lp) * 16;
      f32x4 ta = zero4;
      auto finish = [&](const f32x4& acc, f32x2& x0_, f32x2& x1_) {
        const f32x2 lo = __builtin_elementwise_fma(f32x2{acc[0], acc[1]}, f32x2{esc.x, esc.y}, f32x2{esh.x, esh.y});
        const f32x2 hi = __builtin_elementwise_fma(f32x2{acc[2], acc[3]}, f32x2{esc.z, esc.w}, f32x2{esh.z, esh.w});
        const uint32_t u0 = __builtin_bit_cast(uint32_t, half_bits<T>(lo)), u1 = __builtin_bit_cast(uint32_t, half_bits<T>(hi));
        x0_ = f32x2{Half<T>::lo(u0), Half<T>::hi(u0)}; x1_ = f32x2{Half<T>::lo(u1), Half<T>::hi(u1)};
      };
      auto tr_term = [&](const f32x2& x0_, const f32x2& x1_, int slot) {
        const uint2 tv = fb_preact4<T>(x0_, x1_, ts, tb);
        const bool mine = (lp >> 2) == slot;
        ta = mma16<T>(make_uint4(mine ? trw.x : 0u, mine ? trw.y : 0u, 0u, 0u), make_uint4(tv.x, tv.y, 0u, 0u), ta);
      };
      if (wave == 0 || wave == 3) {   // rows 0..2 / 7..9
        const int row0 = wave == 0 ? 0 : 7;
        int xb[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) xb[k] = fb_plane<2>(min(4 * k + lq, 4)) + (row0 * FbIn<2>::W + 1 + lp) * 16;
        f32x4 acc[3];
        fb_band<T, 2, 2, 3>(smem, xb, w2r, acc);
        FSEG(6);
#pragma unroll
        for (int ps = 0; ps < 3; ++ps) {
          f32x2 x0_, x1_;
          finish(acc[ps], x0_, x1_);
          *reinterpret_cast<uint2*>(smem + e3 + (row0 + ps) * (FbIn<3>::W * 16)) = fb_preact4<T>(x0_, x1_, c3s, c3b);
          if (wave == 0 ? ps >= 1 : ps <= 1) tr_term(x0_, x1_, wave == 0 ? ps - 1 : ps);
        }
      } else {                        // rows 3, 4 / 5, 6 and one group of halo columns (0, 17 of 10 rows: 20 pixels in 2 groups)
        const int row0 = 1 + 2 * wave;
        int xb[2], xg[2];
        const int dump = FB_DUMP + (tid & 63) * 8;
        const int q = 16 * (wave - 1) + lp;
        const bool ok = q < 2 * FbIn<3>::H;
        const int qq = ok ? q : 0, ry = qq >> 1, rx = (qq & 1) ? 17 : 0;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          xb[k] = fb_plane<2>(min(4 * k + lq, 4)) + (row0 * FbIn<2>::W + 1 + lp) * 16;
          xg[k] = fb_plane<2>(min(4 * k + lq, 4)) + (ry * FbIn<2>::W + rx) * 16;
        }
        const int g3a = ok ? fb_plane<3>(pl) + half + (ry * FbIn<3>::W + rx) * 16 : dump;
        f32x4 acc[2];
        fb_band<T, 2, 2, 2>(smem, xb, w2r, acc);
        const f32x4 ga = fb_group<T, 2, 2>(smem, xg, w2r);
        FSEG(6);
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
          f32x2 x0_, x1_;
          finish(acc[ps], x0_, x1_);
          *reinterpret_cast<uint2*>(smem + e3 + (row0 + ps) * (FbIn<3>::W * 16)) = fb_preact4<T>(x0_, x1_, c3s, c3b);
          tr_term(x0_, x1_, ps);
        }
        f32x2 x0_, x1_;
        finish(ga, x0_, x1_);
        *reinterpret_cast<uint2*>(smem + g3a) = fb_preact4<T>(x0_, x1_, c3s, c3b);
      }
      tsum = f32x4{tsum[0] + ta[0], tsum[1] + ta[1], tsum[2] + ta[2], 0.f};
    }
    FSEG(10);
    __syncthreads();   // B4
    FSEG(7);
    // ---- P4: layer 3, transition, sigmoid ----
    {
      FB_FRESH_IDS();
      const int wlane = (lq * 9 * 16 + lp) * 16;
      const int row0 = 2 * wave;
      int xb[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) xb[k] = fb_plane<3>(min(4 * k + lq, 6)) + (row0 * FbIn<3>::W + lp) * 16;
      f32x4 acc[2];
      fb_band_wlds<T, 3, 2, 2>(smem, xb, FB_W3 + wlane, acc);
      FSEG(8);
      const float4 ts = fb_c4(smem, FB_CTS, 56 + 4 * lq), tb = fb_c4(smem, FB_CTB, 56 + 4 * lq);
      const float4 esc = fb_c4(smem, FB_CES, 48 + 4 * lq), esh = fb_c4(smem, FB_CEB, 48 + 4 * lq);
      const uint2 trw = *reinterpret_cast<const uint2*>(smem + FB_TRW + ((56 + 4 * lq) >> 3) * 64 + (lp & 3) * 16 + ((56 + 4 * lq) & 7) * 2);
      const float4 tes = fb_c4(smem, FB_CTE, 0), teb = fb_c4(smem, FB_CTE, 4);
      f32x4 ta = zero4;
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) {
        const f32x2 lo = __builtin_elementwise_fma(f32x2{acc[ps][0], acc[ps][1]}, f32x2{esc.x, esc.y}, f32x2{esh.x, esh.y});
        const f32x2 hi = __builtin_elementwise_fma(f32x2{acc[ps][2], acc[ps][3]}, f32x2{esc.z, esc.w}, f32x2{esh.z, esh.w});
        const uint32_t u0 = __builtin_bit_cast(uint32_t, half_bits<T>(lo)), u1 = __builtin_bit_cast(uint32_t, half_bits<T>(hi));
        const uint2 tv = fb_preact4<T>(f32x2{Half<T>::lo(u0), Half<T>::hi(u0)}, f32x2{Half<T>::lo(u1), Half<T>::hi(u1)}, ts, tb);
        const bool mine = (lp >> 2) == ps;
        ta = mma16<T>(make_uint4(mine ? trw.x : 0u, mine ? trw.y : 0u, 0u, 0u), make_uint4(tv.x, tv.y, 0u, 0u), ta);
      }
      if (lq < 2) {   // lane (lq, lp): pixel (row 2 wave + lq, column lp) of the tile
        float* const ob = a.y + (size_t)img * 3 * hw + ((size_t)(y0 + row0) * a.W + x0);   // wave-uniform
        const unsigned po = (unsigned)lq * (unsigned)a.W + (unsigned)lp;
        const float v0 = fmaf(tsum[0] + ta[0], tes.x, teb.x), v1 = fmaf(tsum[1] + ta[1], tes.y, teb.y), v2 = fmaf(tsum[2] + ta[2], tes.z, teb.z);
        ob[po] = sigmoidf(v0);
        ob[hw + po] = sigmoidf(v1);
        ob[2 * hw + po] = sigmoidf(v2);
      }
    }
    FSEG(11);
  };

  for (; item < run_end; ++item) {
    const int y0 = ty * FB_TH, x0 = tx * FB_TW;
    FSEG(-1);
    if (y0 >= 8 && y0 + 12 <= a.H && x0 >= 16 && x0 + 20 <= a.W) {   // the whole base patch (and its low-resolution taps) inside: no clamp, no padding
      tile_interior(img, y0, x0);
      if (++tx == a.tiles_x) { tx = 0; if (++ty == a.tiles_y) { ty = 0; ++img; } }
      continue;
    }

    // =========================== P0: the base patch, 24 x 16 ===========================
    float fbase[2][3];
    bool inimg[2];
    {
      FB_FRESH_IDS();
      float t[2][4][3], xin[2][3], hy[2][2], wx[2][2];
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int p = tid + it * FB_THREADS;
        const int py = p / FB_RBW, px = p - py * FB_RBW;
        const int gy = y0 - 4 + py, gx = x0 - 4 + px;
        inimg[it] = p < FB_RBW * FB_RBH && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int c = 0; c < 3; ++c) t[it][k][c] = 0.f;
        xin[it][0] = xin[it][1] = xin[it][2] = 0.f;
        hy[it][0] = hy[it][1] = wx[it][0] = wx[it][1] = 0.f;
        if (inimg[it]) {
          int ya, yb, xa, xb;
          fb_src(gy, Hl, ya, yb, hy[it][0], hy[it][1]);
          fb_src(gx, Wl, xa, xb, wx[it][0], wx[it][1]);
          const char* lb = a.lo + (size_t)img * Hl * Wl * a.lo_stride;             // wave-uniform
          const unsigned ls = a.lo_stride, ra = __umul24(ya, Wl), rb = __umul24(yb, Wl);
          const char* q[4] = {lb + __umul24(ra + xa, ls), lb + __umul24(ra + xb, ls), lb + __umul24(rb + xa, ls), lb + __umul24(rb + xb, ls)};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const uint2 u = *reinterpret_cast<const uint2*>(q[k]);
            t[it][k][0] = Half<T>::lo(u.x); t[it][k][1] = Half<T>::hi(u.x); t[it][k][2] = Half<T>::lo(u.y);
          }
          const float* xbase = a.x + (size_t)img * 3 * hw;                          // wave-uniform
          const unsigned xo = (__umul24(gy, a.W) + gx) * 4u;
          xin[it][0] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(xbase) + xo);
          xin[it][1] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(xbase + hw) + xo);
          xin[it][2] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(xbase + 2 * hw) + xo);
        }
      }
      // (the base channels' constants in four of its five consumers: broadcast LDS reads)
      const float4 c0s = fb_c3(smem, FB_CPS, 0), c0b = fb_c3(smem, FB_CPB, 0), c1s = fb_c3(smem, FB_CPS, 72), c1b = fb_c3(smem, FB_CPB, 72);
      const float4 c2s = fb_c3(smem, FB_CPS, 144), c2b = fb_c3(smem, FB_CPB, 144), cts = fb_c3(smem, FB_CTS, 0), ctb = fb_c3(smem, FB_CTB, 0);
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int p = tid + it * FB_THREADS;
        float f[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          f[c] = hy[it][0] * (wx[it][0] * t[it][0][c] + wx[it][1] * t[it][1][c]) + hy[it][1] * (wx[it][0] * t[it][2][c] + wx[it][1] * t[it][3][c]) + xin[it][c];
          f[c] = (float)(T)f[c];                   // the stored base of the chain
          fbase[it][c] = f[c];
        }
        if (p < FB_RBW * FB_RBH) {                 // (it == 1: waves 0, 1 only)
          const int py = p / FB_RBW, px = p - py * FB_RBW;
          const bool in = inimg[it];
          // layer 0's source: relu before the rounding, as updense0.hip forms it
          uint2 pv = make_uint2(0u, 0u);
          if (in) pv = make_uint2(Half<T>::pack(fmaxf(fmaf(f[0], c0s.x, c0b.x), 0.f), fmaxf(fmaf(f[1], c0s.y, c0b.y), 0.f)), Half<T>::pack(fmaxf(fmaf(f[2], c0s.z, c0b.z), 0.f), 0.f));
          *reinterpret_cast<uint2*>(smem + FB_PATCH + p * 8) = pv;
          const f32x2 x01 = {f[0], f[1]}, x2 = {f[2], 0.f};
          {   // column 0 of A_1
            const int cy = py - 1, cx = px - 1;
            if (cy >= 0 && cy < FbIn<1>::H && cx >= 0 && cx < FbIn<1>::W) {
              uint2 v = fb_preact4<T>(x01, x2, c1s, c1b);
              if (!in) v = make_uint2(0u, 0u);
              *reinterpret_cast<uint2*>(smem + fb_plane<1>(0) + (cy * FbIn<1>::W + cx) * 16) = v;
            }
          }
          {   // column 0 of A_2
            const int cy = py - 2, cx = px - 2;
            if (cy >= 0 && cy < FbIn<2>::H && cx >= 0 && cx < FbIn<2>::W) {
              uint2 v = fb_preact4<T>(x01, x2, c2s, c2b);
              if (!in) v = make_uint2(0u, 0u);
              *reinterpret_cast<uint2*>(smem + fb_plane<2>(0) + (cy * FbIn<2>::W + cx) * 16) = v;
            }
          }
          {   // the transition's pre-activation of the tile's own base pixels (always inside the picture)
            const int cy = py - 4, cx = px - 4;
            if (cy >= 0 && cy < FB_TH && cx >= 0 && cx < FB_TW)
              *reinterpret_cast<uint2*>(smem + FB_TRPATCH + (cy * FB_TW + cx) * 8) =
                  make_uint2(Half<T>::pack(fmaxf(fmaf(f[0], cts.x, ctb.x), 0.f), fmaxf(fmaf(f[1], cts.y, ctb.y), 0.f)), Half<T>::pack(fmaxf(fmaf(f[2], cts.z, ctb.z), 0.f), 0.f));
          }
        }
      }
    }
    FSEG(0);
    __syncthreads();   // B1: patch, A_1 / A_2 column 0 and trpatch are in LDS; every wave has left the previous tile's layer 3

    FSEG(1);
    // =========================== P1: layer 0 on 22 x 14 -> A_1, A_2, A_3, transition terms of base + g0 ===========================
    {
      FB_FRESH_IDS();
      // layer 0's im2col gather (updense0.hip): K group lq = taps 2 lq, 2 lq + 1 of the pixel's 3x3 window, second step = tap 8
      const int goA = (((2 * lq) / 3) * FB_RBW + (2 * lq) % 3) * 8, goB = (((2 * lq + 1) / 3) * FB_RBW + (2 * lq + 1) % 3) * 8, goC = (2 * FB_RBW + 2) * 8;
      // column 0 of A_3 (layer 3 of the PREVIOUS tile read it until B1)
      const float4 c3s = fb_c3(smem, FB_CPS, 216), c3b = fb_c3(smem, FB_CPB, 216);
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int p = tid + it * FB_THREADS;
        if (p < FB_RBW * FB_RBH) {
          const int py = p / FB_RBW, px = p - py * FB_RBW;
          const int cy = py - 3, cx = px - 3;
          if (cy >= 0 && cy < FbIn<3>::H && cx >= 0 && cx < FbIn<3>::W) {
            uint2 v = fb_preact4<T>(f32x2{fbase[it][0], fbase[it][1]}, f32x2{fbase[it][2], 0.f}, c3s, c3b);
            if (!inimg[it]) v = make_uint2(0u, 0u);
            *reinterpret_cast<uint2*>(smem + fb_plane<3>(0) + (cy * FbIn<3>::W + cx) * 16) = v;
          }
        }
      }
      float4 cs[3], cb[3];
#pragma unroll
      for (int L = 1; L <= 3; ++L) { cs[L - 1] = fb_c4(smem, FB_CPS, 72 * L + 8 + 4 * lq); cb[L - 1] = fb_c4(smem, FB_CPB, 72 * L + 8 + 4 * lq); }
      const float4 ts = fb_c4(smem, FB_CTS, 8 + 4 * lq), tb = fb_c4(smem, FB_CTB, 8 + 4 * lq);
      const float4 bias = fb_c4(smem, FB_CEB, 4 * lq);
      uint4 tra;
      {
        const uint2 wg = fb_tr_rows(smem, 8, lq, lp);
        const uint2 wb = *reinterpret_cast<const uint2*>(smem + FB_TRW + (lp & 3) * 16);   // the base's stored channels 0..3: K group 0
        tra = make_uint4(wg.x, wg.y, (lq == 0 && lp < 4) ? wb.x : 0u, (lq == 0 && lp < 4) ? wb.y : 0u);
      }
      auto l0 = [&](int ry, int rx, uint32_t& u0, uint32_t& u1) {
        const char* const bp = smem + FB_PATCH + (ry * FB_RBW + rx) * 8;
        const uint2 pa = *reinterpret_cast<const uint2*>(bp + goA), pb = *reinterpret_cast<const uint2*>(bp + goB), pc = *reinterpret_cast<const uint2*>(bp + goC);
        f32x4 acc = mma16<T>(wf0[0], make_uint4(pa.x, pa.y, pb.x, pb.y), zero4);
        acc = mma16<T>(wf0[1], make_uint4(pc.x, pc.y, pc.x, pc.y), acc);
        u0 = Half<T>::pack(acc[0] + bias.x, acc[1] + bias.y); u1 = Half<T>::pack(acc[2] + bias.z, acc[3] + bias.w);
      };
      // centre strip: rows 0..13, 4 / 4 / 3 / 3 per wave
      const int row0 = wave < 2 ? 4 * wave : 8 + 3 * (wave - 2), nrows = wave < 2 ? 4 : 3;
      for (int r = 0; r < nrows; ++r) {
        const int ry = row0 + r, gy = y0 - 3 + ry;
        const bool img_ok = gy >= 0 && gy < a.H;                                   // wave-uniform
        uint32_t u0, u1;
        l0(ry, 3 + lp, u0, u1);
        fb_emit<T, 0, true>(smem, u0, u1, ry, 3 + lp, img_ok, true, lq, cs, cb);
        const int c = ry - 3;                                                      // row of the tile itself
        if (c >= 0 && c < FB_TH) {
          const uint2 tv = fb_preact4<T>(f32x2{Half<T>::lo(u0), Half<T>::hi(u0)}, f32x2{Half<T>::lo(u1), Half<T>::hi(u1)}, ts, tb);
          const uint2 bv = *reinterpret_cast<const uint2*>(smem + FB_TRPATCH + (c * FB_TW + lp) * 8);
          const f32x4 ta = mma16<T>(tra, make_uint4(tv.x, tv.y, bv.x, bv.y), zero4);
          if (lq == 0) *reinterpret_cast<float4*>(smem + FB_TRP + (c * FB_TW + lp) * 16) = make_float4(ta[0], ta[1], ta[2], 0.f);
        }
      }
      // halo columns 0..2, 19..21: 84 pixels in 6 groups, 1 / 1 / 2 / 2 per wave
      const int g0 = wave < 2 ? wave : 2 * wave - 2, ng = wave < 2 ? 1 : 2;
      for (int g = g0; g < g0 + ng; ++g) {
        const int q = 16 * g + lp;
        const bool ok = q < 6 * FbIn<1>::H;
        const int qq = ok ? q : 0, ry = qq / 6, cc = qq - 6 * ry, rx = cc < 3 ? cc : cc + 16;
        const int gy = y0 - 3 + ry, gx = x0 - 3 + rx;
        const bool img_ok = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        uint32_t u0, u1;
        l0(ry, rx, u0, u1);
        fb_emit<T, 0, false>(smem, u0, u1, ry, rx, img_ok, ok, lq, cs, cb);
      }
    }
    FSEG(2);
    __syncthreads();   // B2
    FSEG(3);

    // =========================== P2: layer 1 on 20 x 12 -> A_2, A_3, transition term of g1 ===========================
    {
      FB_FRESH_IDS();
      float4 cs[3], cb[3];
      cs[0] = cb[0] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int L = 2; L <= 3; ++L) { cs[L - 1] = fb_c4(smem, FB_CPS, 72 * L + 24 + 4 * lq); cb[L - 1] = fb_c4(smem, FB_CPB, 72 * L + 24 + 4 * lq); }
      const float4 ts = fb_c4(smem, FB_CTS, 24 + 4 * lq), tb = fb_c4(smem, FB_CTB, 24 + 4 * lq);
      const float4 esc = fb_c4(smem, FB_CES, 16 + 4 * lq), esh = fb_c4(smem, FB_CEB, 16 + 4 * lq);
      const uint2 trw = fb_tr_rows(smem, 24, lq, lp);
      auto finish = [&](const f32x4& acc, uint32_t& u0, uint32_t& u1) {
        const f32x2 lo = __builtin_elementwise_fma(f32x2{acc[0], acc[1]}, f32x2{esc.x, esc.y}, f32x2{esh.x, esh.y});
        const f32x2 hi = __builtin_elementwise_fma(f32x2{acc[2], acc[3]}, f32x2{esc.z, esc.w}, f32x2{esh.z, esh.w});
        u0 = __builtin_bit_cast(uint32_t, half_bits<T>(lo)); u1 = __builtin_bit_cast(uint32_t, half_bits<T>(hi));
      };
      const int row0 = 3 * wave;
      const int xb[1] = {fb_plane<1>(min(lq, 2)) + (row0 * FbIn<1>::W + 2 + lp) * 16};
      f32x4 acc[3];
      fb_band<T, 1, 1, 3>(smem, xb, w1r, acc);
      FSEG(4);
#pragma unroll
      for (int ps = 0; ps < 3; ++ps) {
        const int ry = row0 + ps, gy = y0 - 2 + ry;
        const bool img_ok = gy >= 0 && gy < a.H;
        uint32_t u0, u1;
        finish(acc[ps], u0, u1);
        fb_emit<T, 1, true>(smem, u0, u1, ry, 2 + lp, img_ok, true, lq, cs, cb);
        const int c = ry - 2;
        if (c >= 0 && c < FB_TH) {
          const uint2 tv = fb_preact4<T>(f32x2{Half<T>::lo(u0), Half<T>::hi(u0)}, f32x2{Half<T>::lo(u1), Half<T>::hi(u1)}, ts, tb);
          const f32x4 ta = mma16<T>(make_uint4(trw.x, trw.y, 0u, 0u), make_uint4(tv.x, tv.y, 0u, 0u), zero4);
          if (lq == 0) {
            float4* const pp = reinterpret_cast<float4*>(smem + FB_TRP + (c * FB_TW + lp) * 16);
            const float4 part = *pp;
            *pp = make_float4(part.x + ta[0], part.y + ta[1], part.z + ta[2], 0.f);
          }
        }
      }
      if (wave < 3) {   // halo columns 0, 1, 18, 19: 48 pixels in 3 groups
        const int q = 16 * wave + lp, ry = q >> 2, cc = q & 3, rx = cc < 2 ? cc : cc + 16;
        const int gy = y0 - 2 + ry, gx = x0 - 2 + rx;
        const bool img_ok = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        const int xg[1] = {fb_plane<1>(min(lq, 2)) + (ry * FbIn<1>::W + rx) * 16};
        const f32x4 ga = fb_group<T, 1, 1>(smem, xg, w1r);
        uint32_t u0, u1;
        finish(ga, u0, u1);
        fb_emit<T, 1, false>(smem, u0, u1, ry, rx, img_ok, true, lq, cs, cb);
      }
    }
    FSEG(9);
    __syncthreads();   // B3
    FSEG(5);

    // =========================== P3: layer 2 on 18 x 10 -> A_3 (g2 over A_1's g0 planes), transition term of g2 ===========================
    {
      FB_FRESH_IDS();
      float4 cs[3], cb[3];
      cs[0] = cb[0] = cs[1] = cb[1] = make_float4(0.f, 0.f, 0.f, 0.f);
      cs[2] = fb_c4(smem, FB_CPS, 216 + 40 + 4 * lq); cb[2] = fb_c4(smem, FB_CPB, 216 + 40 + 4 * lq);
      const float4 ts = fb_c4(smem, FB_CTS, 40 + 4 * lq), tb = fb_c4(smem, FB_CTB, 40 + 4 * lq);
      const float4 esc = fb_c4(smem, FB_CES, 32 + 4 * lq), esh = fb_c4(smem, FB_CEB, 32 + 4 * lq);
      const uint2 trw = fb_tr_rows(smem, 40, lq, lp);
      auto finish = [&](const f32x4& acc, uint32_t& u0, uint32_t& u1) {
        const f32x2 lo = __builtin_elementwise_fma(f32x2{acc[0], acc[1]}, f32x2{esc.x, esc.y}, f32x2{esh.x, esh.y});
        const f32x2 hi = __builtin_elementwise_fma(f32x2{acc[2], acc[3]}, f32x2{esc.z, esc.w}, f32x2{esh.z, esh.w});
        u0 = __builtin_bit_cast(uint32_t, half_bits<T>(lo)); u1 = __builtin_bit_cast(uint32_t, half_bits<T>(hi));
      };
      auto row_out = [&](const f32x4& acc, int ry) {
        const int gy = y0 - 1 + ry;
        const bool img_ok = gy >= 0 && gy < a.H;
        uint32_t u0, u1;
        finish(acc, u0, u1);
        fb_emit<T, 2, true>(smem, u0, u1, ry, 1 + lp, img_ok, true, lq, cs, cb);
        const int c = ry - 1;
        if (c >= 0 && c < FB_TH) {
          const uint2 tv = fb_preact4<T>(f32x2{Half<T>::lo(u0), Half<T>::hi(u0)}, f32x2{Half<T>::lo(u1), Half<T>::hi(u1)}, ts, tb);
          const f32x4 ta = mma16<T>(make_uint4(trw.x, trw.y, 0u, 0u), make_uint4(tv.x, tv.y, 0u, 0u), zero4);
          if (lq == 0) {
            float4* const pp = reinterpret_cast<float4*>(smem + FB_TRP + (c * FB_TW + lp) * 16);
            const float4 part = *pp;
            *pp = make_float4(part.x + ta[0], part.y + ta[1], part.z + ta[2], 0.f);
          }
        }
      };
      // centre strip: rows 0..9, 3 / 3 / 2 / 2 per wave
      const int row0 = wave < 2 ? 3 * wave : 6 + 2 * (wave - 2);
      int xb[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) xb[k] = fb_plane<2>(min(4 * k + lq, 4)) + (row0 * FbIn<2>::W + 1 + lp) * 16;
      if (wave < 2) {
        f32x4 acc[3];
        fb_band<T, 2, 2, 3>(smem, xb, w2r, acc);
        FSEG(6);
#pragma unroll
        for (int ps = 0; ps < 3; ++ps) row_out(acc[ps], row0 + ps);
      } else {
        f32x4 acc[2];
        fb_band<T, 2, 2, 2>(smem, xb, w2r, acc);
        FSEG(6);
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) row_out(acc[ps], row0 + ps);
        // halo columns 0, 17: 20 pixels in 2 groups (waves 2, 3)
        const int q = 16 * (wave - 2) + lp;
        const bool ok = q < 2 * FbIn<3>::H;
        const int qq = ok ? q : 0, ry = qq >> 1, rx = (qq & 1) ? 17 : 0;
        const int gy = y0 - 1 + ry, gx = x0 - 1 + rx;
        const bool img_ok = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        int xg[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) xg[k] = fb_plane<2>(min(4 * k + lq, 4)) + (ry * FbIn<2>::W + rx) * 16;
        const f32x4 ga = fb_group<T, 2, 2>(smem, xg, w2r);
        uint32_t u0, u1;
        finish(ga, u0, u1);
        fb_emit<T, 2, false>(smem, u0, u1, ry, rx, img_ok, ok, lq, cs, cb);
      }
    }
    FSEG(10);
    __syncthreads();   // B4
    FSEG(7);

    // =========================== P4: layer 3 on the tile, its transition term, bias, sigmoid -> y ===========================
    {
      FB_FRESH_IDS();
      const int wlane = (lq * 9 * 16 + lp) * 16;
      const int row0 = 2 * wave;
      int xb[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) xb[k] = fb_plane<3>(min(4 * k + lq, 6)) + (row0 * FbIn<3>::W + lp) * 16;
      f32x4 acc[2];
      fb_band_wlds<T, 3, 2, 2>(smem, xb, FB_W3 + wlane, acc);
      FSEG(8);
      const float4 ts = fb_c4(smem, FB_CTS, 56 + 4 * lq), tb = fb_c4(smem, FB_CTB, 56 + 4 * lq);
      const float4 esc = fb_c4(smem, FB_CES, 48 + 4 * lq), esh = fb_c4(smem, FB_CEB, 48 + 4 * lq);
      const uint2 trw = fb_tr_rows(smem, 56, lq, lp);
      const float4 tes = fb_c4(smem, FB_CTE, 0), teb = fb_c4(smem, FB_CTE, 4);
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) {
        const int c = row0 + ps;
        const f32x2 lo = __builtin_elementwise_fma(f32x2{acc[ps][0], acc[ps][1]}, f32x2{esc.x, esc.y}, f32x2{esh.x, esh.y});
        const f32x2 hi = __builtin_elementwise_fma(f32x2{acc[ps][2], acc[ps][3]}, f32x2{esc.z, esc.w}, f32x2{esh.z, esh.w});
        const uint32_t u0 = __builtin_bit_cast(uint32_t, half_bits<T>(lo)), u1 = __builtin_bit_cast(uint32_t, half_bits<T>(hi));
        const uint2 tv = fb_preact4<T>(f32x2{Half<T>::lo(u0), Half<T>::hi(u0)}, f32x2{Half<T>::lo(u1), Half<T>::hi(u1)}, ts, tb);
        const f32x4 ta = mma16<T>(make_uint4(trw.x, trw.y, 0u, 0u), make_uint4(tv.x, tv.y, 0u, 0u), zero4);
        if (lq == 0) {
          const float4 part = *reinterpret_cast<const float4*>(smem + FB_TRP + (c * FB_TW + lp) * 16);
          float* const ob = a.y + (size_t)img * 3 * hw + ((size_t)(y0 + c) * a.W + x0);   // wave-uniform
          const float v0 = fmaf(part.x + ta[0], tes.x, teb.x), v1 = fmaf(part.y + ta[1], tes.y, teb.y), v2 = fmaf(part.z + ta[2], tes.z, teb.z);
          ob[lp] = sigmoidf(v0);
          ob[hw + lp] = sigmoidf(v1);
          ob[2 * hw + lp] = sigmoidf(v2);
        }
      }
    }
    FSEG(11);
    if (++tx == a.tiles_x) { tx = 0; if (++ty == a.tiles_y) { ty = 0; ++img; } }
  }
#ifdef EXP_FBSTAMPS
  if (dbg && (tid_ & 63) == 0) {
    unsigned long long t1, r1; asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
    unsigned long long* o = dbg + ((size_t)blockIdx.x * 4 + wave_) * 16;
    for (int k = 0; k < 12; ++k) o[k] = facc[k];
    o[12] = item - item_first; o[13] = t1 - fstart; o[14] = rstart; o[15] = r1;
  }
#endif
}

}  // namespace mdie

using namespace mdie;

extern "C" int mdie_final_dense_fwd(const mdie_final_dense_desc* d, void* stream) {
  MDIE_REQUIRE(d != nullptr, "mdie_final_dense_fwd: null descriptor");
  MDIE_REQUIRE(d->dtype == MDIE_BF16 || d->dtype == MDIE_F16, "mdie_final_dense_fwd: 16-bit element types only (got %d)", d->dtype);
  MDIE_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->H % FB_TH == 0 && d->W % FB_TW == 0,
               "mdie_final_dense_fwd: extent %dx%dx%d (H a multiple of %d, W of %d)", d->B, d->H, d->W, FB_TH, FB_TW);
  MDIE_REQUIRE(d->lo && d->x && d->y && d->w0 && d->wt && d->tr_pre_scale && d->tr_pre_shift && d->tr_post_scale && d->tr_post_shift, "mdie_final_dense_fwd: null pointer");
  for (int l = 0; l < 4; ++l)
    MDIE_REQUIRE(d->pre_scale[l] && d->pre_shift[l] && d->post_scale[l] && d->post_shift[l] && (l == 0 || d->w[l - 1]), "mdie_final_dense_fwd: layer %d: null pointer", l);
  uintptr_t al = (uintptr_t)d->lo | (uintptr_t)d->w0 | (uintptr_t)d->wt | (uintptr_t)d->tr_pre_scale | (uintptr_t)d->tr_pre_shift;
  for (int l = 0; l < 4; ++l) al |= (uintptr_t)d->pre_scale[l] | (uintptr_t)d->pre_shift[l] | (uintptr_t)d->post_scale[l] | (uintptr_t)d->post_shift[l] | (l ? (uintptr_t)d->w[l - 1] : 0);
  MDIE_REQUIRE((al & 15) == 0, "mdie_final_dense_fwd: lo, the weights and the constant vectors must be 16-byte aligned");
  MDIE_REQUIRE(d->lo_stride >= 4 && d->lo_stride % 4 == 0, "mdie_final_dense_fwd: lo's pixel stride must be a multiple of 4 channels (%d)", d->lo_stride);
  MDIE_REQUIRE((size_t)d->H * d->W < ((size_t)1 << 24) && (size_t)(d->H / 2) * (d->W / 2) * d->lo_stride * 2 < ((size_t)1 << 32) && d->lo_stride * 2 < (1 << 24),
               "mdie_final_dense_fwd: picture too large for the kernel's 24-bit pixel / 32-bit byte offsets (%dx%d)", d->H, d->W);
  FbArgs a{};
  a.B = d->B; a.H = d->H; a.W = d->W; a.tiles_x = d->W / FB_TW; a.tiles_y = d->H / FB_TH;
  a.lo = reinterpret_cast<const char*>(d->lo); a.lo_stride = (unsigned)d->lo_stride * 2u;
  a.x = d->x;
  a.w0 = reinterpret_cast<const char*>(d->w0);
  for (int l = 0; l < 3; ++l) a.w[l] = reinterpret_cast<const char*>(d->w[l]);
  for (int l = 0; l < 4; ++l) { a.ps[l] = d->pre_scale[l]; a.pb[l] = d->pre_shift[l]; a.esc[l] = d->post_scale[l]; a.esh[l] = d->post_shift[l]; }
  a.wt = reinterpret_cast<const char*>(d->wt); a.tps = d->tr_pre_scale; a.tpb = d->tr_pre_shift; a.tes = d->tr_post_scale; a.teb = d->tr_post_shift;
  a.y = d->y;
  const long items_l = (long)d->B * a.tiles_x * a.tiles_y;
  MDIE_REQUIRE(items_l < (1l << 30), "mdie_final_dense_fwd: too many tiles (%ld)", items_l);
  const int items = (int)items_l;
  const int wgs = 8 * cdiv(std::min(items, 256 * 2), 8);
#ifdef EXP_FBSTAMPS
  a.dbg = g_fb_dbg;
#endif
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  TimedLaunch tl(MDIE_K_CONV3);
  if (d->dtype == MDIE_BF16) {
    static LdsOptIn opt;
    if (!opt.ensure(reinterpret_cast<const void*>(&final_block_kernel<bf16>), FB_LDS)) return MDIE_ELAUNCH;
    hipLaunchKernelGGL((final_block_kernel<bf16>), dim3(wgs), dim3(FB_THREADS), FB_LDS, s, a, items);
  } else {
    static LdsOptIn opt;
    if (!opt.ensure(reinterpret_cast<const void*>(&final_block_kernel<f16>), FB_LDS)) return MDIE_ELAUNCH;
    hipLaunchKernelGGL((final_block_kernel<f16>), dim3(wgs), dim3(FB_THREADS), FB_LDS, s, a, items);
  }
  MDIE_LAUNCH_CHECK("mdie_final_dense_fwd");
  return MDIE_OK;
}

#ifdef EXP_FBSTAMPS
extern "C" void mdie_exp_set_fb_dbg(void* p) { mdie::g_fb_dbg = (unsigned long long*)p; }
#endif
