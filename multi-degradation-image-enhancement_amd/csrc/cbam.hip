// CBAM (models/cbam.py:84-95) on NHWC tensors, four HBM-bound passes:
//   pool      per-(image, channel) sum / max over H*W          ChannelGate pools, cbam.py:41,44
//   gate      sigmoid(MLP(avg) + MLP(max))                     cbam.py:30-35,42-59  (folded into chanpool's prologue)
//   chanpool  per-pixel max / mean over channels of x*gate     ChannelPool, cbam.py:68-70
//   spatial   sigmoid(BN(conv7x7(map))) ; out = x*gate*s [*mul]  SpatialGate, cbam.py:72-82
// The channel-scaled tensor x*gate is never written: passes 3 and 4 recompute it from x.
// NHWC makes the channel reductions contiguous (one 16-byte vector per lane, wave shuffles to
// finish), and the global pools are deterministic two-level reductions (no float atomics).
#include <stdlib.h>

#include <algorithm>

#include "common.hpp"

// No implicit FMA contraction in this file: its loops are unrolled, and the unrolled body and the remainder loop must
// round identically -- which copy handles a pixel depends on the grid, i.e. on the batch size, and an image's result
// must not (tests/test_gpu_parity.py::test_full_batch_properties).  Fused multiply-adds are written as fmaf() where wanted.
#pragma clang fp contract(off)

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
  int gate_ready;  // pass 3: the gate was computed by cbam_gate_kernel (wide tensors), do not re-derive it per block
  const long long* delta;   // several weight sets in one launch (mdie_cbam_desc.blob_delta): per-image byte offset of w1, b1, w2, b2, w7, bn
};

// the arguments with image `img`'s weight set selected (a scalar load and six pointer additions; nothing for the usual single set)
__device__ __forceinline__ CbamArgs cbam_select(const CbamArgs& a0, int img) {
  CbamArgs a = a0;
  if (a0.delta) {
    const long long dl = a0.delta[img];
    auto sh = [&](const float* p) { return reinterpret_cast<const float*>(reinterpret_cast<const char*>(p) + dl); };
    a.w1 = sh(a0.w1); a.b1 = sh(a0.b1); a.w2 = sh(a0.w2); a.b2 = sh(a0.b2); a.w7 = sh(a0.w7); a.bn = sh(a0.bn);
  }
  return a;
}

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
#pragma unroll 4
  for (int p = p_begin + r; p < p_end; p += rows) {   // (unrolled: 4 independent 16-byte loads in flight per thread)
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
template <int W1V, int W2V>   // float4 of layer-1 / layer-2 weights a thread holds: <16, 8> covers C <= 512, <1, 2> C <= 128
__device__ __forceinline__ void cbam_gate_block(const CbamArgs& a0, int img, float* avg, float* mx, float* part, float* hid, float* gate) {
  const CbamArgs a = cbam_select(a0, img);
  const int Hd = a.C / 16;
  const int tid = threadIdx.x;
  const float inv = 1.0f / (float)(a.H * a.W);
  // Every phase of this block is a chain of dependent memory round trips unless its loads are issued together (a C = 512 gate
  // ran 17 us that way).  The MLP weights depend on nothing, so every thread requests ITS weights of both layers first --
  // into registers, up to 16 + 16 float4 -- and the phases below consume them: what is left on the chain is one round trip
  // for the pooled partials and the LDS exchanges.
  //   layer 1: thread = (unit j, part q), 256/Hd parts per unit, each a contiguous run of C/parts channels of BOTH pooled vectors
  //   layer 2: thread = channel c (+256 when C = 512), all Hd hidden units
  const int parts = CB_THREADS / Hd;                   // 8 (C=512) .. 256 (C=16)
  const int run = a.C / parts > 0 ? a.C / parts : 1;   // channels per part: 1 (C<=64), 4, 16, 64
  const int j1 = tid / parts, q1 = tid - j1 * parts;
  const int c01 = q1 * run;
  const bool l1_live = j1 < Hd && c01 < a.C;
  const float* w1p = a.w1 + (size_t)j1 * a.C + c01;
  const bool vec1 = run >= 4 && ((uintptr_t)a.w1 & 15) == 0;
  const bool vec2 = Hd >= 4 && ((uintptr_t)a.w2 & 15) == 0;
  float4 w1r[W1V], w2r[2][W2V];
  float w1s = 0.f;
  // the biases too: read where they are used they are two more dependent (cold, TLB-missing) round trips on the chain
  const float b1r = tid < Hd ? a.b1[tid] : 0.f;
  float b2r[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) b2r[h] = tid + h * CB_THREADS < a.C ? a.b2[tid + h * CB_THREADS] : 0.f;
  if (l1_live) {
    if (vec1) {
#pragma unroll
      for (int i = 0; i < W1V; ++i) if (i * 4 < run) w1r[i] = *reinterpret_cast<const float4*>(w1p + i * 4);
    } else if (run == 1) w1s = w1p[0];
  }
  if (vec2) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c = tid + h * CB_THREADS;
      if (c < a.C) {
#pragma unroll
        for (int i = 0; i < W2V; ++i) if (i * 4 < Hd) w2r[h][i] = *reinterpret_cast<const float4*>(a.w2 + (size_t)c * Hd + i * 4);
      }
    }
  }
  // Phase 1: fold the pooled partials of the image's slabs.  Thread = (channel c, slab class q): class q sums slabs
  // q, q+P, q+2P, ... in order, then the classes are combined in order -- fixed by (C, nslab) alone, i.e. by the resolution.
  {
    const int cg = a.C < CB_THREADS ? a.C : CB_THREADS;   // threads along the channels
    const int P = CB_THREADS / cg;                        // slab classes (1 when C >= 256)
    const int q = tid / cg;
    for (int c = tid - q * cg; c < a.C; c += cg) {
      float s = 0.f, m = -INFINITY;
      const float* p = a.partial + (size_t)img * a.nslab * 2 * a.C + c;
#pragma unroll 8
      for (int k = q; k < a.nslab; k += P) {
        s += p[(size_t)k * 2 * a.C];
        m = fmaxf(m, p[(size_t)k * 2 * a.C + a.C]);
      }
      if (P == 1) { avg[c] = s * inv; mx[c] = m; }
      else { part[q * cg + c] = s; part[CB_THREADS + q * cg + c] = m; }
    }
    if (P > 1) {
      __syncthreads();
      if (tid < a.C) {
        float s = 0.f, m = -INFINITY;
        for (int k = 0; k < P; ++k) { s += part[k * cg + tid]; m = fmaxf(m, part[CB_THREADS + k * cg + tid]); }
        avg[tid] = s * inv;
        mx[tid] = m;
      }
    }
  }
  __syncthreads();
  // Phase 2: hidden layer partial dot products
  {
    float sa = 0.f, sm = 0.f;
    if (l1_live) {
      if (vec1) {
#pragma unroll
        for (int i = 0; i < W1V; ++i)
          if (i * 4 < run) {
            const float4 wv = w1r[i];
            const int c = c01 + i * 4;
            sa = fmaf(wv.x, avg[c], sa); sm = fmaf(wv.x, mx[c], sm);
            sa = fmaf(wv.y, avg[c + 1], sa); sm = fmaf(wv.y, mx[c + 1], sm);
            sa = fmaf(wv.z, avg[c + 2], sa); sm = fmaf(wv.z, mx[c + 2], sm);
            sa = fmaf(wv.w, avg[c + 3], sa); sm = fmaf(wv.w, mx[c + 3], sm);
          }
      } else if (run == 1) {
        sa = w1s * avg[c01]; sm = w1s * mx[c01];
      } else {
        for (int i = 0; i < run; ++i) { const float wv = w1p[i]; sa = fmaf(wv, avg[c01 + i], sa); sm = fmaf(wv, mx[c01 + i], sm); }
      }
    }
    part[tid] = sa;
    part[CB_THREADS + tid] = sm;
  }
  __syncthreads();
  if (tid < Hd) {
    float sa = 0.f, sm = 0.f;
    for (int q = 0; q < parts; ++q) { sa += part[tid * parts + q]; sm += part[CB_THREADS + tid * parts + q]; }
    hid[tid] = fmaxf(sa + b1r, 0.f) + fmaxf(sm + b1r, 0.f);
  }
  __syncthreads();
  // Phase 3: output layer (the MLP, bias included, is applied to both pooled vectors)
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int c = tid + h * CB_THREADS;
    if (c < a.C) {
      float s = 2.0f * b2r[h];
      if (vec2) {
#pragma unroll
        for (int i = 0; i < W2V; ++i)
          if (i * 4 < Hd) {
            const float4 wv = w2r[h][i];
            const int j = i * 4;
            s = fmaf(wv.x, hid[j], s); s = fmaf(wv.y, hid[j + 1], s); s = fmaf(wv.z, hid[j + 2], s); s = fmaf(wv.w, hid[j + 3], s);
          }
      } else {
        const float* w = a.w2 + (size_t)c * Hd;
        for (int j = 0; j < Hd; ++j) s = fmaf(w[j], hid[j], s);
      }
      gate[c] = sigmoidf(s);
    }
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
  cbam_gate_block<16, 8>(a, blockIdx.x, avg, mx, part, hid, gate);
  for (int c = threadIdx.x; c < a.C; c += CB_THREADS) a.gate[(size_t)blockIdx.x * a.C + c] = gate[c];
}

template <int CTRL> __device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
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
  // The pixel loop is latency-bound (one 16-byte load per lane and pixel, then a lane reduction): U neighbouring pixels per
  // lane are fetched together (a block reads one contiguous run of U * groups pixels per iteration; U = 4 and U streams a
  // power-of-two distance apart both measured slower on the 64x64 / 128x128 maps), the first batch is requested BEFORE the
  // gate is derived so that it arrives under that prelude, and each next batch before the current one is reduced.
  // (x and map never overlap.)
  const char* __restrict__ xsrc = a.x;
  float* __restrict__ mdst = a.map;
  constexpr int U = 2;
  const int pstep = gridDim.x * groups * U;
  auto load_batch = [&](int p0, uint4 (&u)[U][NV]) {
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const int p = p0 + j;
      const char* px = xsrc + ((size_t)img * npix + (p < npix ? p : 0)) * a.x_stride * sizeof(T);
#pragma unroll
      for (int k = 0; k < NV; ++k) u[j][k] = *reinterpret_cast<const uint4*>(px + (size_t)(k * LPP + sub) * 16);
    }
  };
  uint4 u[U][NV];
  const int p_first = (blockIdx.x * groups + tid / LPP) * U;
  load_batch(p_first, u);
  if (a.gate_ready) {   // launch-uniform
    for (int c = tid; c < a.C; c += CB_THREADS) gsh[c] = a.gate[(size_t)img * a.C + c];
    __syncthreads();
  } else {
    cbam_gate_block<1, 2>(a, img, avg, mx, part, hid, gsh);   // folded only for C <= 128 (run <= 4 channels, Hd <= 8)
    if (blockIdx.x == 0)
      for (int c = tid; c < a.C; c += CB_THREADS) a.gate[(size_t)img * a.C + c] = gsh[c];
  }
  float g[NV][VEC];
#pragma unroll
  for (int k = 0; k < NV; ++k)
#pragma unroll
    for (int i = 0; i < VEC; ++i) g[k][i] = gsh[(k * LPP + sub) * VEC + i];
  const float invC = 1.0f / (float)a.C;
  for (int p0 = p_first; p0 < npix; p0 += pstep) {
    uint4 un[U][NV];
    const bool more = p0 + pstep < npix;
    if (more) load_batch(p0 + pstep, un);
    float m[U], s[U];
#pragma unroll
    for (int j = 0; j < U; ++j) {
      m[j] = -INFINITY; s[j] = 0.f;
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        float f[VEC];
        Vec16<T>::unpack(u[j][k], f);
#pragma unroll
        for (int i = 0; i < VEC; ++i) { const float y = f[i] * g[k][i]; m[j] = fmaxf(m[j], y); s[j] += y; }
      }
    }
    // all-reduce over the LPP lanes of a pixel: DPP lane swaps inside a 16-lane row (no LDS crossbar round trip per step),
    // ds_bpermute only across rows.  After each step every lane of the reduced subgroup holds the same value, so the
    // mirror swaps pair exactly the partial sums the xor butterfly pairs: bit-identical to a __shfl_xor tree.
#pragma unroll
    for (int j = 0; j < U; ++j) {
      float mm = m[j], ss = s[j];
      if (LPP >= 2) { mm = fmaxf(mm, dpp_f<0xB1>(mm)); ss += dpp_f<0xB1>(ss); }     // quad_perm [1,0,3,2]
      if (LPP >= 4) { mm = fmaxf(mm, dpp_f<0x4E>(mm)); ss += dpp_f<0x4E>(ss); }     // quad_perm [2,3,0,1]
      if (LPP >= 8) { mm = fmaxf(mm, dpp_f<0x141>(mm)); ss += dpp_f<0x141>(ss); }   // row_half_mirror
      if (LPP >= 16) { mm = fmaxf(mm, dpp_f<0x140>(mm)); ss += dpp_f<0x140>(ss); }  // row_mirror
      if (LPP >= 32) { mm = fmaxf(mm, __shfl_xor(mm, 16)); ss += __shfl_xor(ss, 16); }
      if (LPP >= 64) { mm = fmaxf(mm, __shfl_xor(mm, 32)); ss += __shfl_xor(ss, 32); }
      m[j] = mm; s[j] = ss;
    }
    if (sub == 0) {
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const int p = p0 + j;
        if (p < npix) *reinterpret_cast<float2*>(mdst + ((size_t)img * npix + p) * 2) = make_float2(m[j], s[j] * invC);
      }
    }
    if (more) {
#pragma unroll
      for (int j = 0; j < U; ++j)
#pragma unroll
        for (int k = 0; k < NV; ++k) u[j][k] = un[j][k];
    }
  }
}

// ---- pass 4 ----------------------------------------------------------------------------------------
template <typename T, int TS>
__global__ __launch_bounds__(CB_THREADS) void cbam_spatial_kernel(const CbamArgs a0) {
  constexpr int VEC = Traits<T>::VEC;
  constexpr int PW = TS + 6;
  const CbamArgs a = cbam_select(a0, blockIdx.y);
  extern __shared__ __attribute__((aligned(16))) char dyn[];
  float* gate = reinterpret_cast<float*>(dyn);     // [C]
  float* patch = gate + a.C;                       // [2][PW][PW]
  float* sg = patch + 2 * PW * PW;                 // [TS*TS]
  float* w7 = sg + TS * TS;                        // [98]
  const int tid = threadIdx.x;
  const int tiles_x = cdiv(a.W, TS);
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x, img = blockIdx.y;
  const int y0 = ty * TS, x0 = tx * TS;
  // ---- the streaming part (x * gate * s [* mul] -> out) is a chain of memory round trips unless its loads are batched:
  // U units (16 bytes of one pixel each) per thread are fetched together, clamped to the tile origin where a unit lies
  // outside the image, and the first batch is requested before the gate / 7x7 phases so that it arrives under them.
  // (out may coincide with x -- every unit is read before it is written by the same thread -- but not overlap it otherwise.)
  constexpr int U = 4;
  const int CV = a.C / VEC;                      // a power of two
  const int cv_shift = __builtin_ctz(CV);
  const int total = TS * TS * CV;
  // Addresses: wave-uniform image base + a 32-bit lane offset from 24-bit multiplies (a 32-bit integer multiply or a 64-bit
  // mad is 16 cycles of the SIMD, v_mul_u32_u24 4: this kernel spent 141 of them per thread on index arithmetic); the host
  // checks that a picture's pixels fit 24 bits and its bytes 32.
  const size_t img_pix = (size_t)img * a.H * a.W;
  const unsigned xs = (unsigned)a.x_stride * sizeof(T), ms = (unsigned)a.mul_stride * sizeof(T), os = (unsigned)a.out_stride * sizeof(T);
  const char* __restrict__ xsrc = a.x + img_pix * xs;
  const char* __restrict__ msrc = a.mul ? a.mul + img_pix * ms : nullptr;
  char* __restrict__ odst = a.out + img_pix * os;
  const unsigned p_origin = __umul24(y0, a.W) + x0;
  auto load_batch = [&](int u0, uint4 (&xv)[U], uint4 (&mv)[U]) {
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const int u = u0 + j * CB_THREADS;
      const int uu = u < total ? u : tid;
      const int pix = uu >> cv_shift, v = uu & (CV - 1);
      const int gy = y0 + pix / TS, gx = x0 + pix % TS;
      const unsigned gp = (gy < a.H && gx < a.W) ? __umul24(gy, a.W) + gx : p_origin;
      xv[j] = *reinterpret_cast<const uint4*>(xsrc + __umul24(gp, xs) + (unsigned)v * 16u);
      if (msrc) mv[j] = *reinterpret_cast<const uint4*>(msrc + __umul24(gp, ms) + (unsigned)v * 16u);
    }
  };
  uint4 xv[U], mv[U];
  load_batch(tid, xv, mv);
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
  for (int u0 = tid; u0 < total; u0 += U * CB_THREADS) {
    uint4 xn[U], mn[U];
    const bool more = u0 + U * CB_THREADS < total;   // block-uniform
    if (more) load_batch(u0 + U * CB_THREADS, xn, mn);
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const int u = u0 + j * CB_THREADS;
      if (u < total) {
        const int pix = u >> cv_shift, v = u & (CV - 1);
        const int gy = y0 + pix / TS, gx = x0 + pix % TS;
        if (gy < a.H && gx < a.W) {
          const unsigned gp = __umul24(gy, a.W) + gx;
          float f[VEC];
          Vec16<T>::unpack(xv[j], f);
          const float sv = sg[pix];
#pragma unroll
          for (int i = 0; i < VEC; ++i) f[i] = f[i] * gate[v * VEC + i] * sv;
          if (msrc) {
            float m[VEC];
            Vec16<T>::unpack(mv[j], m);
#pragma unroll
            for (int i = 0; i < VEC; ++i) f[i] *= m[i];
          }
          *reinterpret_cast<uint4*>(odst + __umul24(gp, os) + (unsigned)v * 16u) = Vec16<T>::pack(f);
        }
      }
    }
    if (more) {
#pragma unroll
      for (int j = 0; j < U; ++j) { xv[j] = xn[j]; mv[j] = mn[j]; }
    }
  }
}

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

static int nslab_for(int H, int W) { const int n = cdiv(H * W, POOL_MIN_SLAB), m = pool_max_slabs(H, W); return n < m ? n : m; }

template <typename T>
static int run_cbam(const mdie_cbam_desc* d, bool spatial, hipStream_t stream, cbam_hook_fn before_last = nullptr, void* hook_ctx = nullptr) {
  constexpr int VEC = Traits<T>::VEC;
  CbamArgs a{};
  a.B = d->B; a.H = d->H; a.W = d->W; a.C = d->C;
  a.x = reinterpret_cast<const char*>(d->x); a.x_stride = d->x_stride;
  a.w1 = d->w1; a.b1 = d->b1; a.w2 = d->w2; a.b2 = d->b2; a.w7 = d->w7;
  a.bn = d->bn;
  a.delta = d->blob_delta;
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
  // Wide tensors (C >= 256: 32-128 KB of MLP weights): one gate block per image in its own launch.  Folded into pass 3,
  // every one of its ~1000 blocks re-read those weights from L2 -- 4x the bytes of the tensor itself at C = 512.
  const bool split_gate = spatial && d->C >= 256;
  a.gate_ready = split_gate ? 1 : 0;
  if (!spatial || split_gate) {
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
    // (32 per image, not 64: a block's prelude -- fold of the pooled partials + the gate MLP, ~4 us -- is then paid for twice
    //  the pixels; cbam2 18.8 -> 14.9 us, cbam3 21.9 -> 19.2 at B = 32; 16 no better, 128 worse)
    int cap = (d->C >= 256 && !split_gate) ? 16 : 32;
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
    // (the last pass is the only one that reads `mul`: a caller whose multiplicand comes from another stream joins it HERE)
    if (before_last) if (int rc = before_last(hook_ctx)) return rc;
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

static int check_cbam(const mdie_cbam_desc* d, bool need_out = true) {
  MDIE_REQUIRE(d != nullptr, "mdie_cbam_fwd: null descriptor");
  MDIE_REQUIRE(dtype_valid(d->dtype), "mdie_cbam_fwd: bad dtype %d", d->dtype);
  MDIE_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0, "mdie_cbam_fwd: empty extent");
  MDIE_REQUIRE(d->C >= 16 && d->C <= 512 && (d->C & (d->C - 1)) == 0, "mdie_cbam_fwd: C = %d must be a power of two in [16, 512]", d->C);
  MDIE_REQUIRE(d->x && (d->out || !need_out) && d->w1 && d->b1 && d->w2 && d->b2 && d->w7 && d->bn && d->workspace, "mdie_cbam_fwd: null pointer");
  MDIE_REQUIRE(d->x_stride % 16 == 0 && d->out_stride % 16 == 0 && (!d->mul || d->mul_stride % 16 == 0), "mdie_cbam_fwd: strides must be multiples of 16");
  MDIE_REQUIRE(((uintptr_t)d->x & 15) == 0 && ((uintptr_t)d->out & 15) == 0 && ((uintptr_t)d->mul & 15) == 0, "mdie_cbam_fwd: alignment");
  MDIE_REQUIRE(!d->pool_partial || (d->pool_slabs >= 1 && d->pool_slabs <= MDIE_POOL_SLABS_MAX), "mdie_cbam_fwd: pool_slabs %d", d->pool_slabs);
  {   // the spatial pass indexes a picture with 24-bit pixel and 32-bit byte offsets
    const size_t npx = (size_t)d->H * d->W, smax = (size_t)std::max(std::max(d->x_stride, d->out_stride), d->mul ? d->mul_stride : 0) * 4;
    MDIE_REQUIRE(npx < ((size_t)1 << 24) && smax < ((size_t)1 << 24) && npx * smax < ((size_t)1 << 32), "mdie_cbam_fwd: picture too large (%dx%d)", d->H, d->W);
  }
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

namespace mdie {
// mdie_cbam_fwd with a hook run between the channel-pool pass and the spatial pass (engine.hip: the join of the DenseBlock branch whose
// output is this CBAM's multiplicand -- models/cdan.py:133,141,149 -- so that the branch has the first passes' time to finish)
int cbam_fwd_hooked(const mdie_cbam_desc* d, hipStream_t stream, cbam_hook_fn before_last, void* ctx) {
  if (int e = check_cbam(d)) return e;
  MDIE_SWITCH_T(d->dtype, return run_cbam<T>(d, true, stream, before_last, ctx));
}
}  // namespace mdie

extern "C" int mdie_cbam_fwd(const mdie_cbam_desc* d, void* stream) {
  using namespace mdie;
  if (int e = check_cbam(d)) return e;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  MDIE_SWITCH_T(d->dtype, return run_cbam<T>(d, true, s));
}

extern "C" int mdie_cbam_channel_only_fwd(const mdie_cbam_desc* d, void* stream) {
  using namespace mdie;
  if (int e = check_cbam(d)) return e;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  MDIE_SWITCH_T(d->dtype, return run_cbam<T>(d, false, s));
}
