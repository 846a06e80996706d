// conv_kernel: the general fused convolution (3x3 / 1x1, 16 or 64 outputs per workgroup, 16x16 or 8x8 pixel tiles) of the CDAN
// path -- formulation and measurements in conv.hip's header.  The template lives in this header so that the instantiations whose
// output is written plane by plane (PLANAR: the input gradients of the DenseBlock layers in training, conv_planar.hip) are
// compiled in a translation unit of their own: the inference kernels of conv.hip are not recompiled next to them.
#pragma once
#include "conv_common.hpp"

namespace mdie {

template <int KS, int BN, int TILE> struct ConvGeom {
  static constexpr int PAD = KS / 2;
  static constexpr int PW = TILE + 2 * PAD;                 // logical patch edge
  static constexpr int NTAP = KS * KS;
  static constexpr int PLANE = ((PW * PWP * 16 + 127) / 256) * 256 + 128;  // == 128 (mod 256): 2-way staging writes at worst
  static constexpr int WPLANE = NTAP * BN * 16;             // multiple of 256 for BN in {16, 64}
  static constexpr int BUF_BYTES = 4 * PLANE + 4 * WPLANE;  // one stage: patch + weights of a 64-byte K chunk
};

// minimum waves per SIMD the register allocator must allow: the thin (cout = 16) kernels live on occupancy
constexpr int conv_min_waves(int BN, int TILE) { return BN == 16 ? 4 : (TILE == 16 ? 2 : 3); }

#ifdef EXP_STAMPS
#define STAMP(i) do { if (threadIdx.x == 0 && dbg) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); dbg[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16 + (i)] = t_; } } while (0)
#else
#define STAMP(i) do {} while (0)
#endif

// The kernel is instruction-issue bound on its bookkeeping (in-kernel s_memtime stamps: ~2.4k cycles of
// setup and ~5k cycles of epilogue around ~5k cycles of MFMA work for a cout=16 layer), so everything
// that is not a load, an LDS access or an MFMA is kept off the hot path: 3-D grid instead of index
// division, incremental patch coordinates, remainder staging iterations under a wave-uniform branch,
// epilogue constants prefetched at kernel entry.
template <typename T, int KS, int BN, int TILE, bool STATS = false, bool PLANAR = false, bool BNRED = false>
__global__ __launch_bounds__(CONV_THREADS, conv_min_waves(BN, TILE)) void conv_kernel(const ConvArgs a) {
  static_assert(!(PLANAR && STATS), "planar output is the plain epilogue");
  static_assert(!BNRED || PLANAR, "the BatchNorm-backward sums ride on the planar epilogue");
  using G = ConvGeom<KS, BN, TILE>;
#ifdef EXP_STAMPS
  unsigned long long* dbg = (a.e.res_stride == -12345) ? reinterpret_cast<unsigned long long*>(const_cast<char*>(a.e.residual)) : nullptr;
  STAMP(0);
#endif
  constexpr int VEC = Traits<T>::VEC;
  constexpr int KC = Traits<T>::KC;
  constexpr int PAD = G::PAD, PW = G::PW, NTAP = G::NTAP;
  constexpr int NCS = BN / 16;                       // cout subtiles per wave
  constexpr int NPS = TILE * TILE / 64;              // pixel subtiles per wave
  constexpr int PATCH_UNITS = PW * PW * 4;           // 16-byte units
  constexpr int W_UNITS = 4 * NTAP * BN;
  constexpr int PATCH_IT = (PATCH_UNITS + CONV_THREADS - 1) / CONV_THREADS;
  constexpr int W_IT = (W_UNITS + CONV_THREADS - 1) / CONV_THREADS;
  // the last staging iteration is partial: only the first waves run it (wave-uniform branch)
  constexpr int PATCH_LAST_WAVES = ((PATCH_UNITS - (PATCH_IT - 1) * CONV_THREADS) + 63) / 64;
  constexpr int W_LAST_WAVES = ((W_UNITS - (W_IT - 1) * CONV_THREADS) + 63) / 64;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_patch = smem;
  char* lds_w = smem + 4 * G::PLANE;
  float* lds_epi = reinterpret_cast<float*>(smem + G::BUF_BYTES);   // [2][BN] post_scale, post_shift of this workgroup's output channels
  float* lds_pre = lds_epi + 2 * BN;                                // [2][nchunk * KC] pre_scale, pre_shift (when present)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lq = lane >> 4;   // 16-byte K group = LDS plane
  const int lp = lane & 15;

  // grid = (8, n_tiles, ceil(patches / 8)): workgroups are dealt round-robin over the 8 XCDs in dispatch
  // order, so the fastest grid index is the XCD slot.  All output-channel tiles (y) of one pixel patch
  // share that slot: the patch is filled into ONE XCD's L2 and re-read there by the other n-tiles
  // (speed only -- nothing depends on the placement).
#ifdef EXP_XCD_INTERLEAVE
  const int patch = blockIdx.z * 8 + blockIdx.x;
#else
  // each XCD slot owns a CONTIGUOUS run of patches (whole images where they divide evenly): neighbouring tiles share
  // their halo pixels through that XCD's L2 instead of fetching them once per XCD
  const int patch = blockIdx.x * gridDim.z + blockIdx.z;
#endif
  if (patch >= a.tiles_x * a.tiles_y * a.B) return;
  const int tpi = a.tiles_x * a.tiles_y;
  const int img = patch / tpi;
  const int trem = patch - img * tpi;
  const int ty = trem / a.tiles_x, tx = trem - ty * a.tiles_x;
  const int y0 = ty * TILE, x0 = tx * TILE, n0 = blockIdx.y * BN;
  // several weight sets in one launch: this tile's image selects its set (one scalar load; 0 for the usual single set)
  const long long dl = a.delta ? a.delta[img] : 0;
  const char* const p_weight = a.weight + dl;
  const float* const p_pre_scale = param_shift(a.pre_scale, dl), * const p_pre_shift = param_shift(a.pre_shift, dl);
  const float* const p_post_scale = param_shift(a.e.post_scale, dl), * const p_post_shift = param_shift(a.e.post_shift, dl);

  const int q = tid & 3;  // this thread's 16-byte column while staging the patch (CONV_THREADS % 4 == 0)
  const bool has_pre = a.pre_scale != nullptr;
  const int kpad = a.nchunk * KC;

  // ---- staging geometry: unit u = tid + it*256 -> patch pixel u>>2 (advances 64 pixels per iteration) ----
  // global pixel index (img*H + gy)*W + gx of the units: like the LDS offsets a start value and a running sum, plus a bit
  // mask of the iterations whose pixel lies inside the picture (and inside the patch)
  int gpix0, ginside = 0;
  // LDS byte offset of the unit: first one + a running sum of two possible strides (64 pixels further = 64/PW rows and 64%PW
  // columns, one more row when the column wraps) -- a start offset and a wrap bit mask instead of PATCH_IT registers
  constexpr int PD_STEP = ((64 / PW) * PWP + 64 % PW) * 16, PD_WRAP = (PWP - PW) * 16;
  constexpr int LAST_UNITS = PATCH_UNITS - (PATCH_IT - 1) * CONV_THREADS;   // units of the (partial) last iteration
  int pdst0, pwrap = 0;
  {
    int pix = tid >> 2;
    int py = pix / PW, px = pix - py * PW;
    const int base = img * a.H * a.W;
    pdst0 = q * G::PLANE + (py * PWP + px) * 16;
    gpix0 = base + (y0 + py - PAD) * a.W + (x0 + px - PAD);
#pragma unroll
    for (int it = 0; it < PATCH_IT; ++it) {
      const int gy = y0 + py - PAD, gx = x0 + px - PAD;
      const bool in_patch = (it < PATCH_IT - 1) || (tid < LAST_UNITS);
      const bool ok = in_patch && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      ginside |= (ok ? 1 : 0) << it;
      px += 64 % PW; py += 64 / PW;
      if (px >= PW) { px -= PW; py += 1; pwrap |= 1 << it; }
    }
  }
  // byte offset of this thread's weight unit inside one chunk: unit u = tid + it*256 -> (qt, n) = (u / BN, u % BN), so the
  // offsets of successive iterations differ by a wave-uniform constant (256 / BN rows of cout units): one register, not W_IT
  const int wsrc_off0 = ((tid / BN) * a.cout + n0 + (tid % BN)) * 16;
  const int wsrc_step = (CONV_THREADS / BN) * a.cout * 16;
  const size_t wchunk_bytes = (size_t)4 * NTAP * a.cout * 16;

  // staging registers (chunk in flight)
  uint4 pv[PATCH_IT];
  uint4 wv[W_IT];
  bool chunk_live = false;  // this thread's channel column exists in the chunk
  int chunk_c0 = 0;         // first stored channel of this thread's column in the chunk in flight

  auto load_chunk = [&](int chunk) {
    const int c0 = chunk * KC + q * VEC;  // first stored channel of this thread's column
    // branch-free over ALL segment slots (unused ones are zero): every field is read unconditionally, so the kernel-argument
    // loads batch into a few wide s_loads with one wait -- under `if (hit)` each slot cost its own scalar round trip
    // (8 dependent s_waitcnt in the prologue, ~1.5 k cycles of a 12 k-cycle tile)
    const char* sbase = nullptr;
    int sstride = 0;
#pragma unroll
    for (int s = 0; s < MDIE_MAX_SEG; ++s) {
      const int cb = a.seg[s].ch_begin, ce = a.seg[s].ch_end, st = a.seg[s].stride;
      const char* sp = a.seg[s].ptr;
      const bool hit = (s < a.nseg) & (c0 >= cb) & (c0 < ce);
      sbase = hit ? sp + (size_t)(c0 - cb) * sizeof(T) : sbase;
      sstride = hit ? st * (int)sizeof(T) : sstride;
    }
    chunk_live = sbase != nullptr;
    int gp = gpix0;
    const int g_step = (64 / PW) * a.W + 64 % PW, g_wrap = a.W - PW;
#pragma unroll
    for (int it = 0; it < PATCH_IT; ++it) {
      pv[it] = make_uint4(0, 0, 0, 0);
#ifndef EXP_NO_GLOAD
      if (it < PATCH_IT - 1 || wave < PATCH_LAST_WAVES)
        if (chunk_live && ((ginside >> it) & 1)) pv[it] = *reinterpret_cast<const uint4*>(sbase + (size_t)gp * sstride);
#endif
      gp += g_step + (((pwrap >> it) & 1) ? g_wrap : 0);
    }
    // weights of this chunk: global [chunk][q][tap][cout] x 16 B  ->  LDS [q][tap][BN] x 16 B (linear copy per (q, tap))
    const char* wsrc = p_weight + chunk * wchunk_bytes;
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
      wv[it] = make_uint4(0, 0, 0, 0);
#ifndef EXP_NO_WLOAD
      if (it < W_IT - 1 || wave < W_LAST_WAVES)
        if (tid + it * CONV_THREADS < W_UNITS) wv[it] = *reinterpret_cast<const uint4*>(wsrc + wsrc_off0 + it * wsrc_step);
#endif
    }
    chunk_c0 = c0;
  };

  auto store_chunk = [&]() {
    int pd = pdst0;
    // pre-activation constants of this thread's channels: read from their LDS copy here, not loaded with the chunk
    // (where 16 registers would stay live across the whole MFMA phase)
    f32x2 ps_[VEC / 2], pb_[VEC / 2];
    if (has_pre && chunk_live) {
#pragma unroll
      for (int i = 0; i < VEC; i += 4) {
        const float4 s4 = *reinterpret_cast<const float4*>(lds_pre + chunk_c0 + i), b4 = *reinterpret_cast<const float4*>(lds_pre + kpad + chunk_c0 + i);
        ps_[i / 2] = f32x2{s4.x, s4.y}; ps_[i / 2 + 1] = f32x2{s4.z, s4.w};
        pb_[i / 2] = f32x2{b4.x, b4.y}; pb_[i / 2 + 1] = f32x2{b4.z, b4.w};
      }
    }
#pragma unroll
    for (int it = 0; it < PATCH_IT; ++it) {
      if (it < PATCH_IT - 1 || wave < PATCH_LAST_WAVES) {
        if (it < PATCH_IT - 1 || tid < LAST_UNITS) {
          uint4 v = pv[it];
          if (has_pre && chunk_live && ((ginside >> it) & 1)) v = PreAct<T>::apply(v, ps_, pb_);
          *reinterpret_cast<uint4*>(lds_patch + pd) = v;
        }
      }
      pd += PD_STEP + (((pwrap >> it) & 1) ? PD_WRAP : 0);
    }
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
      if (it < W_IT - 1 || wave < W_LAST_WAVES) {
        const int u = tid + it * CONV_THREADS;
        if (u < W_UNITS) *reinterpret_cast<uint4*>(lds_w + u * 16) = wv[it];
      }
    }
  };

  STAMP(1);
  // Launch constants (pre-activation scale/shift of the input channels, post scale/shift of this workgroup's output
  // channels) go to LDS.  Their loads are issued BEFORE the first chunk's and their LDS writes come AFTER it is issued:
  // vector loads return in order, so the writes wait for these few dwords only (a counted vmcnt), not for the patch.
  constexpr int PRE_IT = 2;   // covers 512 input channels; more take the loop below
  float cpre[2][PRE_IT], cepi[2];
  if (has_pre) {
#pragma unroll
    for (int i = 0; i < PRE_IT; ++i) {
      const int c = tid + i * CONV_THREADS;
      cpre[0][i] = c < a.cin ? p_pre_scale[c] : 0.f;
      cpre[1][i] = c < a.cin ? p_pre_shift[c] : 0.f;
    }
  }
  if (tid < BN) { cepi[0] = p_post_scale[n0 + tid]; cepi[1] = p_post_shift[n0 + tid]; }
  load_chunk(0);
  if (has_pre) {   // launch-uniform
#pragma unroll
    for (int i = 0; i < PRE_IT; ++i) {
      const int c = tid + i * CONV_THREADS;
      if (c < kpad) { lds_pre[c] = cpre[0][i]; lds_pre[kpad + c] = cpre[1][i]; }
    }
    for (int c = tid + PRE_IT * CONV_THREADS; c < kpad; c += CONV_THREADS) {
      lds_pre[c] = c < a.cin ? p_pre_scale[c] : 0.f;
      lds_pre[kpad + c] = c < a.cin ? p_pre_shift[c] : 0.f;
    }
  }
  if (tid < BN) { lds_epi[tid] = cepi[0]; lds_epi[BN + tid] = cepi[1]; }   // read back after the last MFMA: visible after any barrier below
  if (has_pre) __syncthreads();
  STAMP(2);

  f32x4 acc[NCS][NPS];
#pragma unroll
  for (int i = 0; i < NCS; ++i)
#pragma unroll
    for (int j = 0; j < NPS; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // per-lane LDS read offset of pixel subtile 0; the other subtiles are compile-time offsets from it (TileStep)
  int xoff0;
  {
    int y, x;
    tile_pixel<TILE>(wave * NPS, lp, y, x);
    xoff0 = lq * G::PLANE + (y * PWP + x) * 16;
  }
  using TS = TileStep<TILE, NPS>;
  const int woff = lq * G::WPLANE + lp * 16;

  for (int chunk = 0; chunk < a.nchunk; ++chunk) {
    if (chunk > 0) __syncthreads();  // previous chunk's LDS reads are done
    if (chunk < 2) STAMP(3 + 4 * chunk);
    store_chunk();
    if (chunk < 2) STAMP(4 + 4 * chunk);
    __syncthreads();
    if (chunk + 1 < a.nchunk) load_chunk(chunk + 1);  // in flight during the MFMAs below
    if (chunk < 2) STAMP(5 + 4 * chunk);

#ifndef EXP_NO_MFMA
    // Operand fragments are double-buffered by hand: the reads of tap t+1 are issued before the MFMAs of tap t and pinned
    // there (sched_barrier).  Left to itself the scheduler, at the register cap, emits ds_read -> s_waitcnt lgkmcnt(0) ->
    // MFMA chains that expose an LDS round trip per MFMA (the MFMA phase of a 16-output tile took 2.9 k cycles for 36 MFMAs).
    uint4 wf[2][NCS], xf[2][NPS];
    auto read_tap = [&](int tap, int b) {
      const int kh = tap / KS, kw = tap - kh * KS;
#pragma unroll
      for (int cs = 0; cs < NCS; ++cs)
        wf[b][cs] = *reinterpret_cast<const uint4*>(lds_w + (tap * BN + cs * 16) * 16 + woff);
#pragma unroll
      for (int ps = 0; ps < NPS; ++ps)
        xf[b][ps] = *reinterpret_cast<const uint4*>(lds_patch + ((kh + TS::dy(ps)) * PWP + kw + TS::dx(ps)) * 16 + xoff0);
    };
    read_tap(0, 0);
#pragma unroll
    for (int tap = 0; tap < NTAP; ++tap) {
      if (tap + 1 < NTAP) read_tap(tap + 1, (tap + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int cs = 0; cs < NCS; ++cs)
#pragma unroll
        for (int ps = 0; ps < NPS; ++ps) acc[cs][ps] = mma16<T>(wf[tap & 1][cs], xf[tap & 1][ps], acc[cs][ps]);
      __builtin_amdgcn_sched_barrier(0);
    }
#endif
    if (chunk < 2) STAMP(6 + 4 * chunk);
  }
  STAMP(11);
  float4 esc[NCS], esh[NCS];
#pragma unroll
  for (int cs = 0; cs < NCS; ++cs) {
    esc[cs] = *reinterpret_cast<const float4*>(lds_epi + cs * 16 + lq * 4);
    esh[cs] = *reinterpret_cast<const float4*>(lds_epi + BN + cs * 16 + lq * 4);
  }
#ifdef EXP_STAMPS
  EpiArgs e2 = a.e;
  if (dbg) { e2.residual = nullptr; e2.res_stride = 0; }
  conv_epilogue<T, NCS, NPS, TILE>(e2, esc, esh, acc, img, y0, x0, n0, wave * NPS, lq, lp);
#else
  if constexpr (PLANAR && BNRED) {
    // the patch image is dead once every wave is past its last MFMA: the 4 waves' channel sums meet there, then one thread per
    // channel folds them in wave order and writes this tile's slab (fixed order: bit-reproducible)
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);           // [wave][2][BN]
    conv_epilogue_bnred<T, NCS, NPS, TILE>(a.e, esc, esh, acc, img, y0, x0, n0, wave * NPS, lq, lp, red, wave, BN);
    __syncthreads();
    if (tid < 2 * BN) {
      const int k = tid / BN, ch = tid - k * BN;
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < CONV_THREADS / 64; ++w) t += red[(w * 2 + k) * BN + ch];
      a.e.b_partial[(((size_t)img * tpi + trem) * 2 + k) * a.cout + n0 + ch] = t;
    }
  } else if constexpr (PLANAR) {   // (no activation, no pooling, no residual: checked by the host)
    conv_epilogue_t<T, NCS, NPS, TILE, MDIE_ACT_NONE, false, false, true>(a.e, esc, esh, acc, img, y0, x0, n0, wave * NPS, lq, lp);
  } else if constexpr (!STATS) {
    conv_epilogue<T, NCS, NPS, TILE>(a.e, esc, esh, acc, img, y0, x0, n0, wave * NPS, lq, lp);
  } else {
    // The tensor this convolution writes feeds a CBAM whose first pass is a global average / max pool over H*W
    // (models/cbam.py:41,44): emit this tile's per-channel sum and maximum while the values are in registers, one
    // partial per (image, tile) -- the gate folds the tiles in order (deterministic, independent of the batch).
    float st_sum[NCS][4], st_max[NCS][4];
#pragma unroll
    for (int cs = 0; cs < NCS; ++cs)
#pragma unroll
      for (int i = 0; i < 4; ++i) { st_sum[cs][i] = 0.f; st_max[cs][i] = -INFINITY; }
    conv_epilogue_t<T, NCS, NPS, TILE, MDIE_ACT_RELU, false, true>(a.e, esc, esh, acc, img, y0, x0, n0, wave * NPS, lq, lp, st_sum, st_max);
    // over the 16 pixel lanes of a row (same lq = same 4 channels) ...
#pragma unroll
    for (int cs = 0; cs < NCS; ++cs)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int d = 8; d > 0; d >>= 1) {
          st_sum[cs][i] += __shfl_xor(st_sum[cs][i], d);
          st_max[cs][i] = fmaxf(st_max[cs][i], __shfl_xor(st_max[cs][i], d));
        }
    // ... then over the 4 waves through LDS (the patch image is dead once every wave is past its last MFMA)
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);           // [wave][2][BN]
    if (lp == 0) {
#pragma unroll
      for (int cs = 0; cs < NCS; ++cs)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          red[(wave * 2 + 0) * BN + cs * 16 + lq * 4 + i] = st_sum[cs][i];
          red[(wave * 2 + 1) * BN + cs * 16 + lq * 4 + i] = st_max[cs][i];
        }
    }
    __syncthreads();
    if (tid < BN) {
      float ss = 0.f, mm = -INFINITY;
#pragma unroll
      for (int w = 0; w < CONV_THREADS / 64; ++w) { ss += red[(w * 2 + 0) * BN + tid]; mm = fmaxf(mm, red[(w * 2 + 1) * BN + tid]); }
      float* dst = a.pool_partial + ((size_t)img * tpi + trem) * 2 * a.cout + n0 + tid;
      dst[0] = ss;
      dst[a.cout] = mm;
    }
  }
#endif
  STAMP(12);
}

}  // namespace mdie
