// 3x3 convolution for the WIDE layers of the CDAN path (>= 64 channels in and out, no pre-activation): encoder.conv2-4
// (models/cdan.py:59-61,81-96) and the decoder's ConvTranspose2d+BN+ReLU stages conv1-3 (:103-111,127-147) -- 62 % of the
// network's FLOPs, the only layers whose roof is the matrix pipe (arithmetic intensity 768-1536 FLOP/B).
//
// Why a second kernel.  conv_kernel (conv.hip) stages a 16x16-pixel x 64-output tile through registers: per 32-channel K
// chunk a workgroup pulls 57.6 KB (20.7 KB of patch + 36.9 KB of weights) through the vector memory path for 576 MFMAs,
// 100 B per MFMA.  At the matrix pipe's rate (4 SIMDs x 1 MFMA / 16 cycles) that is 25 B/clk per CU, against the ~29 B/clk
// an MI355X CU takes in from its XCD's L2 (MI355X_MICROARCH.md, indexed-row gathers): the kernel can only run its load,
// LDS-write and MFMA phases one after the other, and profiles at 34-38 % of the bf16 MFMA peak (profiles/r01n_*).
// Here:
//   * tile = 32 x 16 pixels x 64 outputs per workgroup of 8 waves (each wave 64 pixels x 64 outputs, as before): the weights
//     of a chunk serve twice the pixels -> 78.8 KB per 1152 MFMAs = 68 B per MFMA (17 B/clk at the full matrix rate);
//   * staging is LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction, no VGPRs, no ds_write): two stages of
//     (patch + weights) = 154 KB of the CU's 160 KB LDS, the DMA of stage s+1 in flight during the MFMAs of stage s,
//     ONE barrier per stage;
//   * workgroups are persistent (one per CU) over a contiguous run of (pixel tile, output tile) items: the stage ring
//     runs across item boundaries, so an item's epilogue and the next item's first DMA overlap.
// LDS images.  Weights: the planar [q][tap][64 couts][16 B] image of conv_kernel (conflict-free A-operand reads); the packed
// global layout [chunk][q][tap][cout][16 B] makes every (q, tap) row of an output tile ONE contiguous 1 KiB DMA piece.
// Patch: pixel-major, 64 B per pixel, row pitch 36 pixels (34 used); an LDS-DMA piece lands lane-linear (lane i -> base +
// 16 i), so the 16-byte slot a K group goes to is chosen through the SOURCE address: pixel (row, col) keeps K group g in
// slot g ^ ((col >> 1) & 3) ^ ((row & 1) << 1), which makes every B-operand ds_read_b128 (16 pixels of two rows, one K group per
// 16 lanes) conflict-free for all taps (exhaustive check: tools/lds_swizzle_search.py).  Pixels outside the picture read a
// 64-byte zero page instead (nn.Conv2d's zero padding).
// The epilogue is conv.hip's (conv_common.hpp): affine + ReLU, residual, 2x2 max-pool, CBAM pooling partials -- results are
// bit-identical to conv_kernel's (same MFMA order per output element: chunks, then taps, in sequence).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "conv_common.hpp"

namespace mdie {

constexpr int WD_THREADS = 512;
constexpr int WD_TW = 32, WD_TH = 16;                 // pixel tile
constexpr int WD_PITCH = 36;                          // LDS row pitch in pixels (TW + 2 used)
constexpr int WD_ROWS = WD_TH + 2;
constexpr int WD_PATCH_PIECES = (WD_ROWS * WD_PITCH * 64 + 1023) / 1024;   // 41 one-KiB pieces (16 pixels each)
constexpr int WD_PATCH_BYTES = WD_PATCH_PIECES * 1024;
constexpr int WD_BN = 64;
constexpr int WD_WPLANE = 9 * WD_BN * 16;             // one K group's [tap][cout] plane
constexpr int WD_W_PIECES = 36;                       // (q, tap): 64 couts x 16 B
constexpr int WD_W_BYTES = WD_W_PIECES * 1024;
constexpr int WD_STAGE = WD_PATCH_BYTES + WD_W_BYTES; // 78848
constexpr int WD_MAX_COUT = 512;                      // post scale / shift of the whole layer live in LDS
constexpr int WD_LDS = 2 * WD_STAGE + 2 * WD_MAX_COUT * (int)sizeof(float);
constexpr int WD_PP = (WD_PATCH_PIECES + 7) / 8;      // patch pieces per wave (6)
constexpr int WD_WP = (WD_W_PIECES + 7) / 8;          // weight pieces per wave (5)
static_assert(WD_LDS <= 160 * 1024, "two stages must fit the CU's LDS");

__device__ __attribute__((aligned(64))) unsigned char g_wd_zero[64];   // zero-initialised: source of every padding pixel

struct WideArgs {
  int half_width;        // items of 32 outputs (two per staged 64-output weight tile)
  ConvArgs c;
  int tiles_x, tiles_y;     // 32x16 tiles per image
  int items;                // item = (pixel tile, output tile), output tile fastest
  int per_xcd, wgs_per_xcd; // item order, below
};

// one 1 KiB LDS-DMA piece: lane i's 16 bytes at `src` -> LDS byte address lds + 16 i.  Hidden from the compiler on purpose
// (an LDS-DMA it can see makes it wait vmcnt(0) before every later ds_read of the same array -- the DMA of the next stage
// would be drained in front of this stage's first operand read); the stage loop waits for it with its own s_waitcnt.
__device__ __forceinline__ void dma_piece(const char* src, unsigned lds) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(lds) : "memory");
}

#ifdef EXP_STAMPS   // diagnostic build only (tools/stamp_wide.py): shader-clock stamps of wave 0 per stage, into a buffer passed as `residual`
#define WSTAMP(slot) do { if (dbg && tid == 0 && s < 24) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); dbg[((size_t)blockIdx.x * 24 + s) * 4 + (slot)] = t_; } } while (0)
#else
#define WSTAMP(slot) do {} while (0)
#endif

// NCS = 16-output subtiles an item computes: 4 (the 64 staged outputs) or 2 -- the HALF-WIDTH form for layers whose 64-wide
// item count leaves CUs idle (dec.conv2 at B = 32: 128 items on 256 CUs): two workgroups stage the same 64-output weight
// tile and each runs the MFMAs of one half of it, so a stage's matrix work halves while twice the CUs are busy.
// MULTI (round 4): several weight sets in one launch (ConvArgs.delta: a byte offset per image).  An item's weights are DMA'd per
// stage anyway -- the item's image adds its offset to the source; the epilogue constants, which the single-set form parks in LDS
// once per workgroup, are read from the item's weight set when its epilogue runs.
template <typename T, int ACT, bool POOL, bool STATS, int NCS = 4, bool MULTI = false>
__global__ __launch_bounds__(WD_THREADS, 2) void conv_wide_kernel(const WideArgs w) {
#ifdef EXP_STAMPS
  WideArgs w2 = w;
  unsigned long long* dbg = (w.c.e.res_stride == -12345) ? reinterpret_cast<unsigned long long*>(const_cast<char*>(w.c.e.residual)) : nullptr;
  if (dbg) { w2.c.e.residual = nullptr; w2.c.e.res_stride = 0; }
  const ConvArgs& a = w2.c;
  if (dbg && threadIdx.x == 0) { unsigned long long r_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r_) :: "memory"); dbg[(size_t)gridDim.x * 96 + blockIdx.x * 2] = r_; }
#else
  const ConvArgs& a = w.c;
#endif
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* lds_epi = reinterpret_cast<float*>(smem + 2 * WD_STAGE);
  const unsigned smem_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;   // LDS byte address of the dynamic region

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wave >> 2, strip = wave & 3;           // 16x16 half of the tile, 4-row strip of it
  const int lq = lane >> 4, lp = lane & 15;
  constexpr int NPS = 4;
  constexpr int BNI = NCS * 16;                           // outputs of an item
  static_assert(NCS == 4 || (NCS == 2 && !POOL && !STATS), "the half-width form has the plain epilogue only");

  // Which items a workgroup takes.  Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 shares an XCD: observed
  // placement, used for speed only).  Each XCD slot owns a contiguous run of `per_xcd` items; its workgroups walk the run
  // INTERLEAVED (workgroup j takes items j, j + wgs_per_xcd, ...), so at any moment the workgroups of one XCD work on
  // neighbouring items -- the output tiles of the same few pixel tiles -- and a patch chunk is pulled into that XCD's L2
  // once for all of them instead of once per output tile (FETCH_SIZE of the step: -11 %, profiles/r02*_traffic*).
  const int xcd = blockIdx.x & 7, wj = blockIdx.x >> 3;
  const int run_lo = xcd * w.per_xcd;
  const int run_n = min(w.per_xcd, w.items - run_lo);
  if (wj >= run_n) return;
  const int first = run_lo + wj, istep = w.wgs_per_xcd;
  const int nitems = (run_n - wj + istep - 1) / istep;
  const int nstages = nitems * a.nchunk;
  const int tpi = w.tiles_x * w.tiles_y;
  const int sbytes = a.seg[0].stride * (int)sizeof(T);    // pixel stride of the input in bytes

  // ---- operand read addresses (item independent) ----
  // B operand: this lane's pixel of subtile 0, LDS column = x + 1 + kw - 1 ... the patch's column 0 is picture column x0 - 1
  int xaddr[3][2];
  {
    int y, x;
    tile_pixel<16>(strip * NPS, lp, y, x);
    x += 16 * half;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
      for (int par = 0; par < 2; ++par) {
        const int col = x + kw, row = y + par;
        const int slot = lq ^ ((col >> 1) & 3) ^ ((row & 1) << 1);
        xaddr[kw][par] = (y * WD_PITCH + col) * 64 + slot * 16;
      }
  }
  using TS = TileStep<16, NPS>;
  const int woff = WD_PATCH_BYTES + lq * WD_WPLANE + lp * 16;

  // ---- DMA: per-lane source offsets of this wave's patch pieces for the item being fetched ----
  // piece pp = wave + 8 j covers LDS pixels 16 pp .. 16 pp + 15; lane i holds slot i & 3 of pixel 16 pp + (i >> 2)
  unsigned poff[WD_PP];       // byte offset of the lane's 16 bytes in the input tensor (chunk 0)
  unsigned pvalid = 0;        // bit j: inside the picture (else the zero page)
  int d_item = first, d_chunk = 0;
  int d_n0 = 0;
  long long d_delta = 0;      // MULTI: weight-set offset of the item being fetched
  // item independent: the lane's patch pixel (row, col) per piece and its K group's byte offset
  int prow[WD_PP], pcol[WD_PP];   // row = -1: the lane's pixel is padding of the LDS image (pitch 36 > 34 columns, last piece's tail)
#pragma unroll
  for (int j = 0; j < WD_PP; ++j) {
    const int idx = (wave + 8 * j) * 16 + (lane >> 2);
    const int row = idx / WD_PITCH, col = idx - row * WD_PITCH;
    const int g = (lane & 3) ^ ((col >> 1) & 3) ^ ((row & 1) << 1);
    prow[j] = (row < WD_ROWS && col < WD_TW + 2) ? row : -1;
    pcol[j] = col | (g << 8);
  }
  auto dma_setup = [&](int item) {
    const int nt = item % a.n_tiles, patch = item / a.n_tiles;
    const int img = patch / tpi, trem = patch - img * tpi;
    const int ty = trem / w.tiles_x, tx = trem - ty * w.tiles_x;
    const int y0 = ty * WD_TH - 1, x0 = tx * WD_TW - 1;
    d_n0 = (nt * BNI) / WD_BN * WD_BN;                    // the 64-output weight tile this item's outputs live in
    if constexpr (MULTI) d_delta = a.delta[img];
    pvalid = 0;
#pragma unroll
    for (int j = 0; j < WD_PP; ++j) {
      const int gy = y0 + prow[j], gx = x0 + (pcol[j] & 255);
      const bool ok = (prow[j] >= 0) & (gy >= 0) & (gy < a.H) & (gx >= 0) & (gx < a.W);   // (bitwise: no branches)
      poff[j] = (unsigned)((img * a.H + gy) * a.W + gx) * (unsigned)sbytes + (unsigned)((pcol[j] >> 8) * 16);
      pvalid |= (ok ? 1u : 0u) << j;
    }
  };
  // piece k of this wave's share of a stage: k < WD_PP patch pieces, then WD_WP weight pieces
  auto dma_one = [&](int k, int chunk, int buf) {
    const unsigned sb = smem_base + buf * WD_STAGE;
    if (k < WD_PP) {
      const int pp = wave + 8 * k;
      if (pp < WD_PATCH_PIECES) {   // wave-uniform
        const char* src = ((pvalid >> k) & 1u) ? a.seg[0].ptr + chunk * 64 + poff[k] : reinterpret_cast<const char*>(g_wd_zero) + (lane & 3) * 16;
        dma_piece(src, sb + pp * 1024);
      }
    } else {
      const int wp = wave + 8 * (k - WD_PP);   // = q * 9 + tap
      if (wp < WD_W_PIECES)
        // POOL: LDS row cs*16 + j of the piece holds output channel 4j + cs (see the epilogue): the permutation costs nothing,
        // it is the lane's SOURCE row
        dma_piece(a.weight + d_delta + (size_t)chunk * (4 * 9 * 16) * a.cout + (size_t)d_n0 * 16 + (POOL ? 4 * (lane & 15) + (lane >> 4) : lane) * 16 + (size_t)wp * a.cout * 16,
                  sb + WD_PATCH_BYTES + wp * 1024);
    }
  };
  auto dma_issue = [&](int chunk, int buf) {
#pragma unroll
    for (int k = 0; k < WD_PP + WD_WP; ++k) dma_one(k, chunk, buf);
  };

  // the layer's epilogue constants -> LDS, before any DMA is in flight (a compiler-visible load waits vmcnt(0), which
  // would drain the asynchronous stage behind it); read back per item, many barriers later
  if constexpr (!MULTI)
    for (int c = tid; c < a.cout; c += WD_THREADS) { lds_epi[c] = a.e.post_scale[c]; lds_epi[WD_MAX_COUT + c] = a.e.post_shift[c]; }
  dma_setup(d_item);
  dma_issue(0, 0);
  if (++d_chunk == a.nchunk) { d_chunk = 0; d_item += istep; }

  f32x4 acc[NCS][NPS];
  int item = first, chunk = 0;
  for (int s = 0; s < nstages; ++s) {
    // stage s has landed (every wave waits for its own pieces, the barrier publishes them) and every wave is past the
    // operand reads of stage s-1, whose buffer the next DMA overwrites
    WSTAMP(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WSTAMP(1);
    __syncthreads();
    // The next stage's DMA is issued INSIDE this stage's MFMA phase, two pieces per tap over the first taps: the CU's
    // address path takes one 1 KiB piece per 16 cycles, so 77 pieces issued back to back by 8 waves hold every wave for
    // >= 1.2 k cycles with the matrix pipe idle (measured 1.5-2.9 k per stage, tools/stamp_wide.py); spread between the
    // MFMA groups they ride under the partner wave's matrix work, and the last three taps give the tail time to land.
    const bool fetch = s + 1 < nstages;
    const int f_chunk = d_chunk, f_buf = (s + 1) & 1;
    if (fetch) {
      if (d_chunk == 0) dma_setup(d_item);
      if (++d_chunk == a.nchunk) { d_chunk = 0; d_item += istep; }
    }
    WSTAMP(2);
    const int n0 = (item % a.n_tiles) * BNI;
    const int csb = (n0 % WD_BN) / 16;                     // first of this item's subtiles inside the staged weight tile
    if (chunk == 0) {
#pragma unroll
      for (int i = 0; i < NCS; ++i)
#pragma unroll
        for (int j = 0; j < NPS; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const char* st = smem + (s & 1) * WD_STAGE;
    // operand fragments double-buffered by hand, as in conv_kernel
    uint4 wf[2][NCS], xf[2][NPS];
    auto read_tap = [&](int tap, int b) {
      const int kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
      for (int cs = 0; cs < NCS; ++cs)
        wf[b][cs] = *reinterpret_cast<const uint4*>(st + (tap * WD_BN + (csb + cs) * 16) * 16 + woff);
#pragma unroll
      for (int ps = 0; ps < NPS; ++ps)
        xf[b][ps] = *reinterpret_cast<const uint4*>(st + ((kh + TS::dy(ps)) * WD_PITCH + TS::dx(ps)) * 64 + xaddr[kw][kh & 1]);
    };
    read_tap(0, 0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (tap + 1 < 9) read_tap(tap + 1, (tap + 1) & 1);
      if (fetch) {   // uniform
        if (2 * tap < WD_PP + WD_WP) dma_one(2 * tap, f_chunk, f_buf);
        if (2 * tap + 1 < WD_PP + WD_WP) dma_one(2 * tap + 1, f_chunk, f_buf);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int cs = 0; cs < NCS; ++cs)
#pragma unroll
        for (int ps = 0; ps < NPS; ++ps)
          acc[cs][ps] = POOL ? mma16<T>(xf[tap & 1][ps], wf[tap & 1][cs], acc[cs][ps])    // rows = pixels, columns = channels
                             : mma16<T>(wf[tap & 1][cs], xf[tap & 1][ps], acc[cs][ps]);
      __builtin_amdgcn_sched_barrier(0);
    }

    WSTAMP(3);
    if (++chunk == a.nchunk) {
      chunk = 0;
      const int patch = item / a.n_tiles;
      const int img = patch / tpi, trem = patch - img * tpi;
      const int ty = trem / w.tiles_x, tx = trem - ty * w.tiles_x;
      const int y0 = ty * WD_TH, x0 = tx * WD_TW + 16 * half;
      // the epilogue constants: the layer's, parked in LDS at kernel entry -- or (MULTI) this item's weight set's, from memory
      const float* e_sc = lds_epi; const float* e_sh = lds_epi + WD_MAX_COUT;
      if constexpr (MULTI) { const long long dl = a.delta[img]; e_sc = param_shift(a.e.post_scale, dl); e_sh = param_shift(a.e.post_shift, dl); }
      float4 esc[NCS], esh[NCS];
#pragma unroll
      for (int cs = 0; cs < NCS; ++cs) {
        esc[cs] = *reinterpret_cast<const float4*>(e_sc + n0 + cs * 16 + lq * 4);
        esh[cs] = *reinterpret_cast<const float4*>(e_sh + n0 + cs * 16 + lq * 4);
      }
      if constexpr (POOL) {
        // Pooled form (encoder.conv2/conv3), operand roles exchanged as in conv_first_pool_kernel: a lane's four accumulator
        // registers are the 2x2 window of ONE pooled pixel (tile_pixel puts a window on 4 consecutive rows) for channel
        // 4 lp + cs, so affine + ReLU + max-pool are four fma and two v_max3_f32 per subtile and channel group -- no DPP on
        // all 64 lanes for a result a quarter of them keep -- and the lane stores 4 consecutive channels (8 bytes).  The
        // epilogue was 3.1 k of the 15 k cycles conv2 spends per item (tools/stamp_wide.py).  Scalar fma on purpose: see
        // conv_first_pool_kernel about the splat v_pk_fma_f32 forms.
        const float4 p_sc = *reinterpret_cast<const float4*>(e_sc + n0 + 4 * lp), p_sh = *reinterpret_cast<const float4*>(e_sh + n0 + 4 * lp);
        const float sc4[4] = {p_sc.x, p_sc.y, p_sc.z, p_sc.w}, sh4[4] = {p_sh.x, p_sh.y, p_sh.z, p_sh.w};
        const int Ho = a.e.H >> 1, Wo = a.e.W >> 1;
#pragma unroll
        for (int ps = 0; ps < NPS; ++ps) {
          const int blk = (strip * NPS + ps) * 4 + lq;                       // this lane group's 2x2 window of the 16x16 half tile
          const int oy = (y0 >> 1) + (blk >> 3), ox = (x0 >> 1) + (blk & 7);
          float m[NCS];
#pragma unroll
          for (int cs = 0; cs < NCS; ++cs) {
            const float v0 = fmaf(acc[cs][ps][0], sc4[cs], sh4[cs]), v1 = fmaf(acc[cs][ps][1], sc4[cs], sh4[cs]);
            const float v2 = fmaf(acc[cs][ps][2], sc4[cs], sh4[cs]), v3 = fmaf(acc[cs][ps][3], sc4[cs], sh4[cs]);
            m[cs] = fmaxf(fmaxf(fmaxf(fmaxf(v0, v1), v2), v3), 0.f);
          }
          T* const o = reinterpret_cast<T*>(a.e.out) + (((size_t)img * Ho + oy) * Wo + ox) * a.e.out_stride + n0 + 4 * lp;
          *reinterpret_cast<uint2*>(o) = make_uint2(Half<T>::pack(m[0], m[1]), Half<T>::pack(m[2], m[3]));
        }
      } else if constexpr (!STATS) {
        conv_epilogue_t<T, NCS, NPS, 16, ACT, POOL>(a.e, esc, esh, acc, img, y0, x0, n0, strip * NPS, lq, lp);
      } else {
        // per-channel sum / maximum of what this 16x16 half tile stores: one slab per half tile, in the raster order of
        // 16x16 tiles -- the layout (and the summation order) conv_kernel<..., STATS> produces
        float st_sum[NCS][4], st_max[NCS][4];
#pragma unroll
        for (int cs = 0; cs < NCS; ++cs)
#pragma unroll
          for (int i = 0; i < 4; ++i) { st_sum[cs][i] = 0.f; st_max[cs][i] = -INFINITY; }
        conv_epilogue_t<T, NCS, NPS, 16, ACT, false, true>(a.e, esc, esh, acc, img, y0, x0, n0, strip * NPS, lq, lp, st_sum, st_max);
#pragma unroll
        for (int cs = 0; cs < NCS; ++cs)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int d = 8; d > 0; d >>= 1) {
              st_sum[cs][i] += __shfl_xor(st_sum[cs][i], d);
              st_max[cs][i] = fmaxf(st_max[cs][i], __shfl_xor(st_max[cs][i], d));
            }
        // across the 4 waves of the half tile: through the stage buffer this item has just finished with (the DMA in flight
        // targets the OTHER buffer; the next DMA into this one is issued only after the next stage's barrier)
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem + (s & 1) * WD_STAGE);   // [wave][2][64]
        if (lp == 0) {
#pragma unroll
          for (int cs = 0; cs < NCS; ++cs)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              red[(wave * 2 + 0) * WD_BN + cs * 16 + lq * 4 + i] = st_sum[cs][i];
              red[(wave * 2 + 1) * WD_BN + cs * 16 + lq * 4 + i] = st_max[cs][i];
            }
        }
        __syncthreads();
        if (tid < 2 * WD_BN) {
          const int h = tid >> 6, ch = tid & 63;
          float ss = 0.f, mm = -INFINITY;
#pragma unroll
          for (int k = 0; k < 4; ++k) { ss += red[((h * 4 + k) * 2 + 0) * WD_BN + ch]; mm = fmaxf(mm, red[((h * 4 + k) * 2 + 1) * WD_BN + ch]); }
          const int slab = ty * (2 * w.tiles_x) + 2 * tx + h;        // raster index of the 16x16 half tile in its picture
          float* dst = a.pool_partial + ((size_t)img * (2 * tpi) + slab) * 2 * a.cout + n0 + ch;
          dst[0] = ss;
          dst[a.cout] = mm;
        }
      }
      item += istep;
    }
  }
#ifdef EXP_STAMPS
  if (dbg && threadIdx.x == 0) {
    unsigned long long r_, t_;
    asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(r_), "=s"(t_) :: "memory");
    dbg[(size_t)gridDim.x * 96 + blockIdx.x * 2 + 1] = r_;
    dbg[(size_t)gridDim.x * 98 + blockIdx.x] = t_;
  }
#endif
}

bool conv_wide_applicable(int dtype, const ConvArgs& a, int ksize, bool has_nchw3) {
  if (dtype == MDIE_F32 || ksize != 3 || has_nchw3 || a.pre_scale || a.nseg != 1) return false;
  if (a.cin % 32 != 0 || a.cin < 64 || a.cout % WD_BN != 0 || a.cout > WD_MAX_COUT) return false;
  if (a.W % WD_TW != 0 || a.H % WD_TH != 0) return false;
  if (a.e.pool && a.e.residual) return false;                                                       // (the pooled epilogue has no residual input)
  if (a.e.act != MDIE_ACT_RELU && !(a.e.act == MDIE_ACT_NONE && !a.e.pool && !a.pool_partial)) return false;   // (NONE: training forward / dgrad)
  if (a.pool_partial && (a.e.pool || mdie_conv_tile(a.B, a.H, a.W, a.cout) != 16)) return false;   // slabs are 16x16 tiles
  if ((size_t)a.B * a.H * a.W * a.seg[0].stride * 2 >= ((size_t)1 << 32)) return false;            // 32-bit source offsets
  const long items = (long)a.B * (a.H / WD_TH) * (a.W / WD_TW) * (a.cout / WD_BN);
  return items >= 96;   // fewer: not enough persistent workgroups to use the chip
}

template <typename T>
static int launch_wide_t(WideArgs& w, hipStream_t stream) {
  const ConvArgs& a = w.c;
  const int grid = 8 * w.wgs_per_xcd;
  TimedLaunch tl(MDIE_K_CONV3);
#define MDIE_WIDE_M(ACT, POOL, STATS, NCS_, MULTI_)                                                               \
  do {                                                                                                           \
    static LdsOptIn opt;                                                                                         \
    if (!opt.ensure(reinterpret_cast<const void*>(&conv_wide_kernel<T, ACT, POOL, STATS, NCS_, MULTI_>), WD_LDS)) return MDIE_ELAUNCH; \
    hipLaunchKernelGGL((conv_wide_kernel<T, ACT, POOL, STATS, NCS_, MULTI_>), dim3(grid), dim3(WD_THREADS), WD_LDS, stream, w); \
  } while (0)
#define MDIE_WIDE(ACT, POOL, STATS) do { if (a.delta) MDIE_WIDE_M(ACT, POOL, STATS, 4, true); else MDIE_WIDE_M(ACT, POOL, STATS, 4, false); } while (0)
#define MDIE_WIDE_HALF(ACT) do { if (a.delta) MDIE_WIDE_M(ACT, false, false, 2, true); else MDIE_WIDE_M(ACT, false, false, 2, false); } while (0)
  if (w.half_width) { if (a.e.act == MDIE_ACT_NONE) MDIE_WIDE_HALF(MDIE_ACT_NONE); else MDIE_WIDE_HALF(MDIE_ACT_RELU); }
  else if (a.pool_partial) MDIE_WIDE(MDIE_ACT_RELU, false, true);
  else if (a.e.pool) MDIE_WIDE(MDIE_ACT_RELU, true, false);
  else if (a.e.act == MDIE_ACT_NONE) MDIE_WIDE(MDIE_ACT_NONE, false, false);
  else MDIE_WIDE(MDIE_ACT_RELU, false, false);
#undef MDIE_WIDE
#undef MDIE_WIDE_HALF
#undef MDIE_WIDE_M
  MDIE_LAUNCH_CHECK("mdie_conv_fwd");
  return MDIE_OK;
}

int launch_conv_wide(int dtype, ConvArgs& a, hipStream_t stream, bool yield_cu) {
  WideArgs w{};
  static int cus = 0;
  if (!cus) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus = n;
  }
  w.tiles_x = a.W / WD_TW; w.tiles_y = a.H / WD_TH;
  // half-width items when the 64-wide ones would leave more than a quarter of the CUs without work
  const long items64 = (long)a.B * w.tiles_x * w.tiles_y * (a.cout / WD_BN);
  w.half_width = !a.e.pool && !a.pool_partial && !a.e.residual && items64 * 4 < (long)cus * 3;
  a.n_tiles = a.cout / (w.half_width ? WD_BN / 2 : WD_BN);
  w.c = a;
  w.items = a.B * w.tiles_x * w.tiles_y * a.n_tiles;
  w.per_xcd = cdiv(w.items, 8);
  w.wgs_per_xcd = w.per_xcd < cus / 8 ? w.per_xcd : cus / 8;      // one persistent workgroup per CU at most
  // yield_cu (mdie_conv_desc.share_cu = 2): twice as many workgroups, each with half the run of items.  A persistent workgroup holds its CU
  // -- and all of its LDS -- until the LAUNCH ends, so the kernels of a side branch (the encoder DenseBlocks beside encoder.conv4) queue
  // behind the whole layer; with two shorter runs per CU the dispatcher gets every CU back half way and the branch's workgroups slip in.
  // 2 us slower alone (the stage ring restarts once more per CU), -29 ... -34 us for the step on the boxes where it was swept.
  if (yield_cu) { const int cap = (cus / 8) * 2; w.wgs_per_xcd = w.per_xcd < cap ? w.per_xcd : cap; }
#ifdef EXP_SCHED   // schedule-exploration builds only (tools/sched_sweep.py): MDIE_EXP_WIDE_WGS="<cin>x<cout>:<m>,..." -- m x as many workgroups as CUs
                   // for that layer (each with 1/m of the items: a CU is handed back to the dispatcher between them instead of held to the end)
  if (const char* v = getenv("MDIE_EXP_WIDE_WGS")) {
    char key[32];
    snprintf(key, sizeof key, "%dx%d:", a.cin, a.cout);
    if (const char* q = strstr(v, key)) {
      const int m = atoi(q + strlen(key));
      if (m > 1) { const int cap = (cus / 8) * m; w.wgs_per_xcd = w.per_xcd < cap ? w.per_xcd : cap; }
    }
  }
#endif
  // (fewer workgroups with several items each -- the next item's first DMA under the previous item's epilogue -- measured for the layers
  //  that have one item per CU: dec.conv2 24.5 -> 38.1 us, dec.conv3 24.9 -> 35.3 us on 128 workgroups x 2 items: profiles/r05e_ab_wide_two_items.txt)
  if (dtype == MDIE_BF16) return launch_wide_t<bf16>(w, stream);
  return launch_wide_t<f16>(w, stream);
}

}  // namespace mdie
