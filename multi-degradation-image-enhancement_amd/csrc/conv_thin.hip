// DenseLayers at full resolution (decoder.final_dense layers 1..3, models/cdan.py:35-46,155):
//     out[16] = act(conv3x3(relu(bn(concat(base, g0, ..)))) * post_scale + post_shift)
// with at most 7 live 16-byte channel columns (<= 56 stored input channels) and 16 output channels, 16-bit storage.
//
// Why a kernel of their own.  conv_kernel runs these at 40-50 % of their HBM floor (final.l1/l2/l3: 52 / 72 / 78 us against
// 21 / 29 / 38 us at B = 32, 256x256; this kernel: 47 / 61 / 73 us -- what is left is at the end of this comment).  Stamps of its sibling for the first layer (tools/stamp_first.py) showed what such
// a tile really pays for: every vector instruction is 4 cycles of a SIMD (integer multiplies and 64-bit address arithmetic
// 16), a branch around a load or a store turns the compiler's counted waits into vmcnt(0) -- the wave then also waits for the
// ~2 us acknowledgement of its own stores -- and a workgroup that lives for one tile pays weight staging, constants and
// launch for 256 pixels.  Here:
//   * persistent workgroups walk a contiguous run of tiles inside their XCD's share of the picture (increment with carry:
//     no division after the prologue); the pre-activation constants go to LDS and the weights to registers ONCE;
//   * one wave stages one 16-byte column of the input at a time (column j = wave + 4c): its segment pointer and pixel stride
//     are wave-uniform, so a patch load is "scalar base + one 24-bit multiply" and the column's BatchNorm constants are
//     read once per tile;
//   * a wave's four pixel subtiles are four ROWS of 16 consecutive pixels, so the operand fragment of input row r and
//     column shift kw serves three output rows (kh = 0, 1, 2): 18 LDS reads for a chunk's 36 MFMAs instead of 36, and the
//     weight fragments live in registers for the whole run.  With 16 outputs every pixel fragment feeds exactly one MFMA:
//     conv_kernel's 5 ds_read_b128 per 4 MFMAs are 360 LDS read instructions per tile = 1440 LDS cycles at the 256 B/clk
//     peak, more than the 1152 cycles the tile's MFMAs need (here: 144 reads);
//   * the NEXT tile's columns are loaded into registers right after the barrier and fly under this tile's MFMAs and stores;
//     every load and store is unconditional (border lanes read the tile's own first pixel and are zeroed at the LDS write;
//     a dead column slot reads one line), the first tile is peeled off the loop, and so the wait before the LDS write is
//     a counted vmcnt(stores issued after the loads), never vmcnt(0);
//   * all live columns of a tile are in LDS at once (plane = column, conv_kernel's bank-conflict-free pitch): two barriers
//     per TILE instead of two per K chunk.
// What is left (tools/stamp_thin.py): per tile a wave issues ~270 vector instructions for the pre-activation of its 12
// patch units (unpack, fused multiply-add, round, ReLU: 5 per dword) next to 72 MFMAs, and with the weights in registers
// only two waves fit a SIMD, so the phases of the CU's two workgroups overlap imperfectly: ~60 % of the issue bound.
// Arithmetic is conv_kernel's, operation for operation (PreAct on the 16-bit input, fp32 accumulate in the same K order,
// one fused multiply-add in the epilogue), so results are bit-identical to it (tests/test_gpu_parity.py).
#include <stdlib.h>
#include <algorithm>

#include "conv_common.hpp"

namespace mdie {

constexpr int TH_THREADS = 256;                                           // 4 waves: wave w owns pixel rows 4w .. 4w+3 of the tile
constexpr int TH_TILE = 16, TH_PW = TH_TILE + 2;
// Plane stride == 0 (mod 256 B).  A ds_read_b128 is served in 4 groups of 16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19,
// 28-31} and the same in the upper half -- each of which must cover 64 distinct banks.  Here a 16-lane K group reads 16
// CONSECUTIVE pixels (256 contiguous bytes) of its plane, so two K groups meet conflict-free when their planes start on the
// same bank.  (conv_kernel's planes are == 128 (mod 256) for its 2x2-block pixel order: with that pitch this kernel's reads
// were 2-way conflicted, SQ_LDS_BANK_CONFLICT = 32 % of SQ_LDS_IDX_ACTIVE.)
constexpr int TH_PLANE = TH_PW * PWP * 16;
static_assert(TH_PLANE % 256 == 0, "plane stride must keep the K groups on the same banks");
constexpr int TH_WCHUNK = 4 * 9 * 16 * 16;                                // packed weights of one 64-byte K chunk, 16 outputs
constexpr int TH_MAXCOL = 8;
constexpr int TH_PIT = (TH_PW * TH_PW + 63) / 64;                         // 6 staging iterations of a wave over the 324 patch pixels
constexpr int TH_PIT2 = (TH_PW * TH_PW + 31) / 32;                        // 11 when a lane is HALF a pixel of a 16-channel segment (PAIR form)
constexpr int TH_MAXUNIT = 4;                                             // PAIR form: one staging unit (a segment's 1 or 2 columns) per wave

struct ThinArgs {
  int B, H, W, tiles_x, tiles_y;
  int ncol, cin;                    // live 16-byte columns, stored input channels
  const char* col_ptr[TH_MAXCOL];   // first byte of column j at pixel 0 (segment base + channel offset)
  unsigned col_stride[TH_MAXCOL];   // bytes per pixel of the segment the column lives in
  const float *pre_scale, *pre_shift;   // [cin] folded BatchNorm of the input channels (stored-channel order)
  const char* weight;               // mdie_pack_conv_weight layout, cout_stored = 16
  const float *post_scale, *post_shift;   // [16]
  char* out; unsigned out_stride;   // bytes per pixel
  unsigned long long* dbg;          // (diagnostic builds)
  // TR != 0: the block's transition folded into this layer (mdie_tr_fuse, see below)
  const char* tr_w; int tr_c0;      // the transition's packed 1x1 weights (16 stored outputs); stored input channel of this layer's output 0
  const float *tr_scale, *tr_shift; // the transition's folded BatchNorm, this layer's 16 channels
  const float* tr_in; float* tr_out;    // [pixel][4] fp32 partial sums of the transition (may be the same buffer)
  int nunit; int unit_col[TH_MAXUNIT]; int unit_ncol[TH_MAXUNIT];   // PAIR form: unit u = columns unit_col[u] .. + unit_ncol[u] (1 or 2) of ONE segment
  const float *tr_post_scale, *tr_post_shift; float* tr_nchw3;   // TR == 2: the transition's epilogue constants and the network output
  const long long* delta;           // MULTI: several weight sets in one launch, a byte offset per image for every parameter pointer above
};

#ifdef EXP_TSTAMPS   // diagnostic build only (tools/stamp_thin.py): per-segment shader-clock sums of wave 0 of each workgroup
static unsigned long long* g_thin_dbg = nullptr;
#define TSEG(k) do { if (dbg) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); if ((k) >= 0) tacc[(k) < 0 ? 0 : (k)] += t_ - tprev; tprev = t_; } } while (0)
#else
#define TSEG(k) do {} while (0)
#endif

// NCHUNK = 64-byte K chunks (1: <= 4 columns, 2: <= 8); CPW = NCHUNK = column slots per wave (column j = wave + 4c)
//
// TR: the DenseBlock's transition (BN -> ReLU -> Conv1x1, models/cdan.py:48-53) folded into its producers.
//     relu(bn(cat(f0, f1, ..))) . W  ==  sum over segments of relu(bn_s(f_s)) . W_s,
// so a layer can add the term of the 16 channels it has just computed while they are still in registers: the accumulator
// layout of the MFMA (lane (lq, lp): channels 4 lq .. 4 lq + 3 of pixel lp) IS the B-operand layout of a second MFMA whose
// K axis is those channels (k = 8 lq + j, j < 4; j >= 4 meets zero weights), so the term costs 10 vector instructions and
// one MFMA per 16 pixels.  The A operand of row subtile ps holds the transition's rows at 4 ps + o, zeros elsewhere: the
// four subtiles accumulate into ONE accumulator in which lane (lq, lp) ends up with the 3 outputs of pixel (row 4 wave +
// lq, column lp) -- one 16-byte fp32 partial per lane and tile, read, added and written back (TR = 1), or finished with
// the transition's bias and the sigmoid and written as the network's fp32 NCHW output while the layer's own 16 channels
// are never stored (TR = 2: nothing reads them).  Removes the transition's launch, its re-read of all 67 channels and the
// last growth map's write from decoder.final_dense (engine.hip).
//
// PAIR (round 3): who stages what.  In the column form above a wave loads one 16-byte column, lane = pixel: for a 16-channel
// growth map (32 bytes per pixel) that is 64 lanes x 16 bytes at a 32-byte stride -- every wave instruction touches 64 half-used
// sectors and the OTHER half of each is fetched by another wave.  Stamps (tools/stamp_thin.py, 7 columns): "issue next tile's
// loads" 2.2 k of a tile's 7.9 k cycles -- twelve loads per lane -- and 3.0 k in front of the LDS writes: the vector memory path,
// not arithmetic.  In the PAIR form a wave stages one UNIT -- the one or two columns of a <= 16-channel stretch of ONE segment --
// with lane = (pixel, 16-byte half): a wave instruction reads 32 pixels x 32 bytes = 1 KiB CONTIGUOUS, 11 loads per lane cover
// the patch, and the lane writes its half into its own plane.  Single-column units (the 8-channel base) run the same 11
// iterations with both halves reading the same 16 bytes and the odd lanes writing nothing.  Needs <= 4 units (one per wave).
// MULTI (round 4): several weight sets in one launch (mdie_conv_desc.blob_delta).  A workgroup's run of tiles is contiguous, so
// with the images of a task next to each other it crosses a weight-set boundary at most a few times.  The run is cut into
// STRETCHES of tiles that share a weight set; everything the workgroup loads once -- weight fragments, pre-activation constants
// in LDS, epilogue and transition constants -- is loaded per stretch, and the tile pipeline (prefetch of the next tile under the
// current one) runs inside a stretch exactly as in the single-set form (a reload inside the tile loop, with the prefetch
// registers live, spilled 50-60 registers).
template <typename T, int NCHUNK, int ACT, int TR = 0, bool PAIR = false, bool MULTI = false>
__global__ __launch_bounds__(TH_THREADS, 2) void conv_thin_kernel(const ThinArgs a, const int n_items) {
  static_assert(sizeof(T) == 2, "16-bit storage types");
  constexpr int PW = TH_PW, PIT = PAIR ? TH_PIT2 : TH_PIT, CPW = PAIR ? 1 : NCHUNK;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const lds_patch = smem;                                             // [ncol] planes
  float* const lds_pre = reinterpret_cast<float*>(smem + a.ncol * TH_PLANE);   // [ncol * 8] scale, [ncol * 8] shift

  const int tid = threadIdx.x, lane = tid & 63, lq = lane >> 4, lp = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- this workgroup's run of tiles (see conv_first_pool_kernel) ----
  const int per_xcd = (n_items + 7) >> 3, nwg = gridDim.x >> 3;
  const int band0 = ((int)blockIdx.x & 7) * per_xcd, band1 = min(band0 + per_xcd, n_items);
  const int run = (per_xcd + nwg - 1) / nwg;
  int item = band0 + ((int)blockIdx.x >> 3) * run;
  const int run_end = min(item + run, band1);
  if (item >= run_end) return;
  int item_end = run_end;             // end of the current stretch (MULTI: where the weight set changes)
#ifdef EXP_TSTAMPS
  unsigned long long* const dbg = a.dbg;
  unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = 0, tstart = 0, rstart = 0;
  if (dbg) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(tstart), "=s"(rstart) :: "memory");
  const int item_first = item;
#endif
  int tx, ty, img;

  // ---- lane constants ----
  // staging: patch pixel p = lane + 64 it: offset in pixels from the patch corner, LDS offset inside a plane, border classes
  unsigned dp[PIT], ldst[PIT];
  unsigned m_top = 0, m_bot = 0, m_left = 0, m_right = 0, m_valid = 0;
#pragma unroll
  for (int it = 0; it < PIT; ++it) {
    const unsigned p = PAIR ? (unsigned)(lane >> 1) + 32u * it : (unsigned)lane + 64u * it, py = p / PW, px = p - py * PW;
    dp[it] = py * (unsigned)a.W + px;
    ldst[it] = (py * PWP + px) * 16;
    m_valid |= (p < PW * PW ? 1u : 0u) << it;
    m_top |= (py == 0 ? 1u : 0u) << it; m_bot |= (py == PW - 1 ? 1u : 0u) << it;
    m_left |= (px == 0 ? 1u : 0u) << it; m_right |= (px == PW - 1 ? 1u : 0u) << it;
  }
  // this wave's column slots (PAIR: its one unit)
  const char* cptr[CPW]; unsigned cstr[CPW]; bool clive[CPW];
  int ucol = 0; bool upair = false;                 // PAIR: first column of the unit, and whether it has two
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    const int j = wave + 4 * c;
    clive[c] = PAIR ? j < a.nunit : j < a.ncol;
    int jj = clive[c] ? j : 0;
    if constexpr (PAIR) { ucol = a.unit_col[jj]; upair = a.unit_ncol[jj] == 2; jj = ucol; }
    cptr[c] = a.col_ptr[jj]; cstr[c] = a.col_stride[jj];
  }
  const unsigned lhalf = PAIR && upair ? (unsigned)(lane & 1) * 16u : 0u;     // the lane's 16-byte half of its pixel
  const int mycol = PAIR ? ucol + (upair ? (lane & 1) : 0) : 0;               // PAIR: the column (= LDS plane) this lane fills
  const bool lwrite = !PAIR || upair || (lane & 1) == 0;                      // (single-column unit: the odd lanes duplicate the even ones)
  // operands: B column lp of the wave's subtile ps = pixel (4 wave + ps, lp) -- 16 consecutive pixels of one row, so the
  // fragment of input row r and column shift kw serves the three output rows r, r-1, r-2 (taps kh = 0, 1, 2): 18 LDS reads
  // for a chunk's 36 MFMAs instead of 36.  K group lq of chunk k = plane min(4k + lq, ncol - 1) (a plane past the live
  // columns meets zero weights: any finite data will do).
  int xoff[NCHUNK];
#pragma unroll
  for (int k = 0; k < NCHUNK; ++k) xoff[k] = min(4 * k + lq, a.ncol - 1) * TH_PLANE + ((4 * wave) * PWP + lp) * 16;
  // output: lane (lq, lp) stores channels 4 lq .. 4 lq + 3 of pixel lp of the row: 512 contiguous bytes per 16 lanes x 4
  const unsigned olane = (unsigned)lp * a.out_stride + (unsigned)lq * 8u;
  const unsigned plane_px = (unsigned)lq * (unsigned)a.W + (unsigned)lp;      // TR: the lane's pixel (row 4 wave + lq, column lp) from the wave's first

  uint4 pv[CPW][PIT];
  unsigned okm = 0;
  bool border = false;
  auto issue = [&](int img_, int y0, int x0) {
    border = y0 == 0 || x0 == 0 || y0 + TH_TILE == a.H || x0 + TH_TILE == a.W;
    okm = m_valid & ~((y0 == 0 ? m_top : 0u) | (y0 + TH_TILE == a.H ? m_bot : 0u) | (x0 == 0 ? m_left : 0u) | (x0 + TH_TILE == a.W ? m_right : 0u));
    const long long gp0 = ((long long)img_ * a.H + (y0 - 1)) * a.W + (x0 - 1);   // pixel index of the patch corner (may lie before the image: never read)
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
      const char* const base = cptr[c] + gp0 * (long long)cstr[c];           // wave-uniform
      const unsigned self = ((unsigned)a.W + 1u) * cstr[c] + lhalf;          // the tile's own first pixel: always inside
#pragma unroll
      for (int it = 0; it < PIT; ++it) {
        const unsigned off = clive[c] ? (((okm >> it) & 1u) ? __umul24(dp[it], cstr[c]) + lhalf : self) : self;
        pv[c][it] = *reinterpret_cast<const uint4*>(base + off);
      }
    }
  };

  do {   // MULTI: one pass per stretch of the run that shares a weight set; otherwise exactly one pass
  { int r = item; tx = r % a.tiles_x; r /= a.tiles_x; ty = r % a.tiles_y; img = r / a.tiles_y; }
  long long dl = 0;
  if constexpr (MULTI) {
    dl = a.delta[img];
    const int per_img = a.tiles_x * a.tiles_y;
    int nimg_ = img + 1;
    while (nimg_ * per_img < run_end && a.delta[nimg_] == dl) ++nimg_;       // wave-uniform, scalar: the first image of another set
    item_end = min(run_end, nimg_ * per_img);
    __syncthreads();                                                         // (the previous stretch's readers of the LDS constants are done)
  }
  // ---- once per stretch: weight fragments -> registers (A row lp = output channel, K group lq), constants -> LDS ----
  uint4 wreg[NCHUNK][9];
#pragma unroll
  for (int k = 0; k < NCHUNK; ++k)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) wreg[k][tap] = *reinterpret_cast<const uint4*>(a.weight + dl + (size_t)k * TH_WCHUNK + ((lq * 9 + tap) * 16 + lp) * 16);
  {
    const int nc = a.ncol * 8;
    const float* const psc_ = param_shift(a.pre_scale, dl), * const psh_ = param_shift(a.pre_shift, dl);
    for (int c = tid; c < nc; c += TH_THREADS) { lds_pre[c] = c < a.cin ? psc_[c] : 0.f; lds_pre[nc + c] = c < a.cin ? psh_[c] : 0.f; }
  }
  const float4 esc = *reinterpret_cast<const float4*>(param_shift(a.post_scale, dl) + 4 * lq), esh = *reinterpret_cast<const float4*>(param_shift(a.post_shift, dl) + 4 * lq);
  // transition term: A fragments (row lp of subtile ps' operand = output lp - 4 ps, K group lq = this layer's channels
  // 4 lq .. 4 lq + 3 in elements 0..3) and the transition's BatchNorm constants of the lane's 4 channels
  uint2 tra[4];
  f32x2 trs[2], trb[2];
  float4 trps = make_float4(0.f, 0.f, 0.f, 0.f), trpb = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (TR != 0) {
    const int c = a.tr_c0 + 4 * lq;                                          // stored input channel of the transition
    const char* const wrow = a.tr_w + dl + ((size_t)(c >> 5) * 4 + ((c & 31) >> 3)) * (16 * 16) + (c & 7) * 2;
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int o = lp - 4 * ps;
      const uint2 w = *reinterpret_cast<const uint2*>(wrow + (unsigned)(o & 15) * 16);   // (unconditional load, masked after)
      tra[ps] = (o >= 0 && o < 4) ? w : make_uint2(0u, 0u);
    }
    const float4 s4 = *reinterpret_cast<const float4*>(param_shift(a.tr_scale, dl) + 4 * lq), b4 = *reinterpret_cast<const float4*>(param_shift(a.tr_shift, dl) + 4 * lq);
    trs[0] = f32x2{s4.x, s4.y}; trs[1] = f32x2{s4.z, s4.w}; trb[0] = f32x2{b4.x, b4.y}; trb[1] = f32x2{b4.z, b4.w};
    if constexpr (TR == 2) { trps = *reinterpret_cast<const float4*>(param_shift(a.tr_post_scale, dl)); trpb = *reinterpret_cast<const float4*>(param_shift(a.tr_post_shift, dl)); }
  }
  int y0 = ty * TH_TILE, x0 = tx * TH_TILE;
  issue(img, y0, x0);
  __syncthreads();                                                           // the constants are in LDS

  bool last = false;
  auto tile = [&]() {
    TSEG(-1);
    // ---- this tile's columns: pre-activation, zero padding of the ACTIVATED tensor, -> LDS ----
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
      if (clive[c]) {   // wave-uniform; no global memory instruction inside
        const int j = PAIR ? mycol : wave + 4 * c;      // (PAIR: per lane -- the even lanes fill the unit's first column, the odd ones its second)
        f32x2 ps_[4], pb_[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const float4 s4 = *reinterpret_cast<const float4*>(lds_pre + j * 8 + 4 * i), b4 = *reinterpret_cast<const float4*>(lds_pre + a.ncol * 8 + j * 8 + 4 * i);
          ps_[2 * i] = f32x2{s4.x, s4.y}; ps_[2 * i + 1] = f32x2{s4.z, s4.w};
          pb_[2 * i] = f32x2{b4.x, b4.y}; pb_[2 * i + 1] = f32x2{b4.z, b4.w};
        }
        char* const plane = lds_patch + j * TH_PLANE;
        constexpr int LPI = PAIR ? 32 : 64;             // pixels per staging iteration
#pragma unroll
        for (int it = 0; it < PIT; ++it) {
          const uint4 v = PreAct<T>::apply(pv[c][it], ps_, pb_);
          if (lwrite && (it < PIT - 1 || (PAIR ? (lane >> 1) : lane) + LPI * it < PW * PW)) *reinterpret_cast<uint4*>(plane + ldst[it]) = v;
        }
        if (border) {   // wave-uniform, border tiles only: the out-of-image pixels are overwritten with zeros (same lane, in order)
#pragma unroll
          for (int it = 0; it < PIT; ++it)
            if (lwrite && !((okm >> it) & 1u) && (it < PIT - 1 || (PAIR ? (lane >> 1) : lane) + LPI * it < PW * PW)) *reinterpret_cast<uint4*>(plane + ldst[it]) = make_uint4(0u, 0u, 0u, 0u);
        }
      }
    }
    TSEG(0);
    __syncthreads();
    TSEG(1);
    // ---- TR: this tile's partial sums, requested ahead of the next tile's columns (older in the vmcnt queue: the epilogue
    //      waits for them without waiting for the prefetch) ----
    float4 part = make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t ppix = ((size_t)img * a.H + y0 + 4 * wave) * a.W + x0;      // wave-uniform
    if constexpr (TR != 0) part = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(a.tr_in + 4 * ppix) + plane_px * 16u);   // (unconditional: the host requires tr_in)
    // ---- next tile's columns into registers ----
    last = item + 1 >= item_end;
    int nimg = img, ny0 = y0, nx0 = x0;                                      // (the last tile is loaded once more: no branch around the loads)
    if (!last) {
      if (++tx == a.tiles_x) { tx = 0; if (++ty == a.tiles_y) { ty = 0; ++nimg; } }
      ny0 = ty * TH_TILE; nx0 = tx * TH_TILE;
    }
    issue(nimg, ny0, nx0);
    TSEG(2);

    // ---- MFMA phase, by input row: the three fragments of row r+1 are read before the MFMAs of row r (pinned) ----
    // Per output the products still arrive in conv_kernel's order (chunk, then tap = 3 kh + kw ascending): bit-identical sums.
    f32x4 acc[4];
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) acc[ps] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < NCHUNK; ++k) {
      const char* const xk = lds_patch + xoff[k];
      uint4 xf[2][3];
      auto read_row = [&](int r, int b) {
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) xf[b][kw] = *reinterpret_cast<const uint4*>(xk + (r * PWP + kw) * 16);
      };
      read_row(0, 0);
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        if (r + 1 < 6) read_row(r + 1, (r + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)             // (kw outside: consecutive MFMAs go to different accumulators)
#pragma unroll
          for (int kh = 2; kh >= 0; --kh) {        // output row ps = r - kh, ascending
            const int ps = r - kh;
            if (ps >= 0 && ps < 4) acc[ps] = mma16<T>(wreg[k][kh * 3 + kw], xf[r & 1][kw], acc[ps]);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    TSEG(3);
    // ---- epilogue: affine, activation, one 8-byte store per lane and row (unconditional: tiles are full) ----
    char* const obase = a.out + ((((size_t)img * a.H + y0 + 4 * wave) * a.W + x0) * a.out_stride);   // wave-uniform
    f32x4 tacc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      f32x2 lo = __builtin_elementwise_fma(f32x2{acc[ps][0], acc[ps][1]}, f32x2{esc.x, esc.y}, f32x2{esh.x, esh.y});
      f32x2 hi = __builtin_elementwise_fma(f32x2{acc[ps][2], acc[ps][3]}, f32x2{esc.z, esc.w}, f32x2{esh.z, esh.w});
      i16x2 p0 = half_bits<T>(lo), p1 = half_bits<T>(hi);
      if constexpr (ACT == MDIE_ACT_RELU) { p0 = __builtin_elementwise_max(p0, i16x2{0, 0}); p1 = __builtin_elementwise_max(p1, i16x2{0, 0}); }
      if constexpr (TR != 2) {
        char* const o = obase + (size_t)ps * a.W * a.out_stride;                                       // wave-uniform
        *reinterpret_cast<uint2*>(o + olane) = make_uint2(__builtin_bit_cast(uint32_t, p0), __builtin_bit_cast(uint32_t, p1));
      }
      if constexpr (TR != 0) {   // the transition's pre-activation of the STORED values (what its own launch would read back), then its MFMA
        const uint32_t u0 = __builtin_bit_cast(uint32_t, p0), u1 = __builtin_bit_cast(uint32_t, p1);
        const f32x2 r0 = __builtin_elementwise_fma(f32x2{Half<T>::lo(u0), Half<T>::hi(u0)}, trs[0], trb[0]);
        const f32x2 r1 = __builtin_elementwise_fma(f32x2{Half<T>::lo(u1), Half<T>::hi(u1)}, trs[1], trb[1]);
        const uint32_t t0 = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(half_bits<T>(r0), i16x2{0, 0}));
        const uint32_t t1 = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(half_bits<T>(r1), i16x2{0, 0}));
        tacc = mma16<T>(make_uint4(tra[ps].x, tra[ps].y, 0u, 0u), make_uint4(t0, t1, 0u, 0u), tacc);
      }
    }
    if constexpr (TR == 1) {
      *reinterpret_cast<float4*>(reinterpret_cast<char*>(a.tr_out + 4 * ppix) + plane_px * 16u) =
          make_float4(part.x + tacc[0], part.y + tacc[1], part.z + tacc[2], 0.f);
    } else if constexpr (TR == 2) {   // transition bias, sigmoid, fp32 NCHW: 16 consecutive floats per lane group and plane
      const size_t hw = (size_t)a.H * a.W;
      float* const ob = a.tr_nchw3 + (size_t)img * 3 * hw + ((size_t)(y0 + 4 * wave) * a.W + x0);   // wave-uniform
      const float v0 = fmaf(part.x + tacc[0], trps.x, trpb.x), v1 = fmaf(part.y + tacc[1], trps.y, trpb.y), v2 = fmaf(part.z + tacc[2], trps.z, trpb.z);
      *reinterpret_cast<float*>(reinterpret_cast<char*>(ob) + plane_px * 4u) = sigmoidf(v0);
      *reinterpret_cast<float*>(reinterpret_cast<char*>(ob + hw) + plane_px * 4u) = sigmoidf(v1);
      *reinterpret_cast<float*>(reinterpret_cast<char*>(ob + 2 * hw) + plane_px * 4u) = sigmoidf(v2);
    }
    ++item; img = nimg; y0 = ny0; x0 = nx0;
    TSEG(4);
    __syncthreads();                                                         // every wave is done reading the planes
    TSEG(5);
  };
  // (first tile peeled: the loop is entered, like its back edge, with "this wave's loads, then 4 stores" outstanding)
  tile();
  while (!last) tile();
  } while (MULTI && item < run_end);
#ifdef EXP_TSTAMPS
  if (dbg && tid == 0) {
    unsigned long long t1, r1; asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
    unsigned long long* o = dbg + (size_t)blockIdx.x * 12;
    for (int k = 0; k < 6; ++k) o[k] = tacc[k];
    o[6] = item - item_first; o[7] = t1 - tstart; o[8] = rstart; o[9] = r1;
  }
#endif
}

// ---- host -----------------------------------------------------------------------------------------------------------
// any_batch: the caller needs THIS kernel whatever the batch size (the transition fusion exists only here) -- the item
// threshold below is a speed choice, not a limit of the kernel
bool conv_thin_applicable(int dtype, const ConvArgs& a, int ksize, bool has_nchw3, bool any_batch) {
  if (dtype == MDIE_F32 || ksize != 3 || has_nchw3 || !a.pre_scale || a.cout != 16) return false;
  if (a.e.pool || a.e.residual || a.pool_partial || (a.e.act != MDIE_ACT_NONE && a.e.act != MDIE_ACT_RELU)) return false;
  if (a.H % TH_TILE != 0 || a.W % TH_TILE != 0) return false;
  const int ncol = (a.cin + 7) / 8;
  if (ncol > TH_MAXCOL - 1 || a.nchunk > 2) return false;   // (8 full columns -- dense1 layer 0, 64 input channels -- measured 7-12 % SLOWER than conv_kernel at every batch size)
  for (int s = 0; s < a.nseg; ++s)
    if (a.seg[s].ch_begin % 8 != 0 || a.seg[s].ch_end % 8 != 0 || (size_t)(TH_PW + 1) * a.W * a.seg[s].stride * 2 >= ((size_t)1 << 24) * 16) return false;
  if ((size_t)a.W * 20 >= ((size_t)1 << 24)) return false;                                      // 24-bit multiply of the pixel offset
  const long items = (long)a.B * (a.H / TH_TILE) * (a.W / TH_TILE);
  return any_batch || items >= 1024;   // fewer: one tile per workgroup, nothing to pipeline -- conv_kernel
}

template <typename T, int NCHUNK>
static int launch_thin_t(const ThinArgs& t, int act, int tr, int items, hipStream_t stream) {
  const size_t lds = (size_t)t.ncol * TH_PLANE + (size_t)2 * t.ncol * 8 * sizeof(float);
  const int per_cu = 2;                                                       // (registers: weights live in them)
  const int wgs = 8 * cdiv(std::min(items, 256 * per_cu), 8);
  TimedLaunch tl(MDIE_K_CONV3);
#define MDIE_THIN_M(ACT, TR, PAIR, MULTI_)                                                                            \
  do {                                                                                                                \
    static LdsOptIn opt;                                                                                              \
    if (!opt.ensure(reinterpret_cast<const void*>(&conv_thin_kernel<T, NCHUNK, ACT, TR, PAIR, MULTI_>), 64 * 1024)) return MDIE_ELAUNCH; \
    hipLaunchKernelGGL((conv_thin_kernel<T, NCHUNK, ACT, TR, PAIR, MULTI_>), dim3(wgs), dim3(TH_THREADS), lds, stream, t, items);      \
  } while (0)
#define MDIE_THIN_P(ACT, TR, PAIR) do { if (t.delta) MDIE_THIN_M(ACT, TR, PAIR, true); else MDIE_THIN_M(ACT, TR, PAIR, false); } while (0)
#define MDIE_THIN(ACT, TR) do { if (t.nunit > 0) MDIE_THIN_P(ACT, TR, true); else MDIE_THIN_P(ACT, TR, false); } while (0)
  if (tr == 1) MDIE_THIN(MDIE_ACT_NONE, 1);
  else if (tr == 2) MDIE_THIN(MDIE_ACT_NONE, 2);
  else if (act == MDIE_ACT_RELU) MDIE_THIN(MDIE_ACT_RELU, 0);
  else MDIE_THIN(MDIE_ACT_NONE, 0);
#undef MDIE_THIN
#undef MDIE_THIN_P
#undef MDIE_THIN_M
  MDIE_LAUNCH_CHECK("mdie_conv_fwd");
  return MDIE_OK;
}

int launch_conv_thin(int dtype, const ConvArgs& a, hipStream_t stream, const mdie_tr_fuse* tr) {
  ThinArgs t{};
  t.B = a.B; t.H = a.H; t.W = a.W; t.tiles_x = a.W / TH_TILE; t.tiles_y = a.H / TH_TILE;
  t.ncol = (a.cin + 7) / 8; t.cin = a.cin;
  for (int j = 0; j < t.ncol; ++j) {
    const int c0 = j * 8;
    t.col_ptr[j] = nullptr;
    for (int s = 0; s < a.nseg; ++s)
      if (c0 >= a.seg[s].ch_begin && c0 < a.seg[s].ch_end) { t.col_ptr[j] = a.seg[s].ptr + (size_t)(c0 - a.seg[s].ch_begin) * 2; t.col_stride[j] = (unsigned)a.seg[s].stride * 2u; }
    MDIE_REQUIRE(t.col_ptr[j] != nullptr, "mdie_conv_fwd: input channel %d belongs to no segment", c0);
  }
  // staging units of the PAIR form: every segment cut into stretches of <= 16 channels (1 or 2 columns of one pixel's contiguous bytes)
#ifndef THIN_PAIR_MIN_UNITS
#define THIN_PAIR_MIN_UNITS 3
#endif
  {
    int nu = 0;
    bool fits = true;
    for (int s = 0; s < a.nseg && fits; ++s)
      for (int c0 = a.seg[s].ch_begin; c0 < a.seg[s].ch_end; c0 += 16) {
        if (nu == TH_MAXUNIT) { fits = false; break; }
        t.unit_col[nu] = c0 / 8;
        t.unit_ncol[nu] = a.seg[s].ch_end - c0 >= 16 ? 2 : 1;
        ++nu;
      }
    t.nunit = (fits && nu >= THIN_PAIR_MIN_UNITS) ? nu : 0;      // (0: the column form)
  }
  t.pre_scale = a.pre_scale; t.pre_shift = a.pre_shift; t.weight = a.weight;
  t.post_scale = a.e.post_scale; t.post_shift = a.e.post_shift;
  t.out = a.e.out; t.out_stride = (unsigned)a.e.out_stride * 2u;
  t.delta = a.delta;
  int mode = 0;
  if (tr) {
    const bool last = tr->out_nchw3 != nullptr;
    MDIE_REQUIRE(a.e.act == MDIE_ACT_NONE, "mdie_conv_fwd: the transition fusion takes layers without activation");
    MDIE_REQUIRE(tr->weight && tr->pre_scale && tr->pre_shift && tr->partial_in && tr->c0 >= 0 && tr->c0 % 8 == 0,
                 "mdie_conv_fwd: tr needs weight, pre_scale / pre_shift, partial_in and a c0 that is a multiple of 8 (got %d)", tr->c0);
    MDIE_REQUIRE(last ? (tr->post_scale && tr->post_shift && tr->act == MDIE_ACT_SIGMOID) : (tr->partial_out != nullptr && a.e.out != nullptr),
                 "mdie_conv_fwd: tr is either a middle producer (partial_out, out) or the last one (out_nchw3, post_scale / post_shift, sigmoid)");
    MDIE_REQUIRE((((uintptr_t)tr->partial_in | (uintptr_t)tr->partial_out | (uintptr_t)tr->weight) & 15) == 0, "mdie_conv_fwd: tr buffers must be 16-byte aligned");
    MDIE_REQUIRE((size_t)a.H * a.W * 16 < ((size_t)1 << 32), "mdie_conv_fwd: picture too large for the 32-bit partial offsets");
    t.tr_w = reinterpret_cast<const char*>(tr->weight); t.tr_c0 = tr->c0;
    t.tr_scale = tr->pre_scale + tr->c0; t.tr_shift = tr->pre_shift + tr->c0;
    t.tr_in = tr->partial_in; t.tr_out = tr->partial_out;
    t.tr_post_scale = tr->post_scale; t.tr_post_shift = tr->post_shift; t.tr_nchw3 = tr->out_nchw3;
    mode = last ? 2 : 1;
  } else {
    MDIE_REQUIRE(a.e.out != nullptr, "mdie_conv_fwd: null output");
  }
  const int items = a.B * t.tiles_x * t.tiles_y;
#ifdef EXP_TSTAMPS
  t.dbg = g_thin_dbg;
#endif
  if (dtype == MDIE_BF16) return t.ncol <= 4 ? launch_thin_t<bf16, 1>(t, a.e.act, mode, items, stream) : launch_thin_t<bf16, 2>(t, a.e.act, mode, items, stream);
  return t.ncol <= 4 ? launch_thin_t<f16, 1>(t, a.e.act, mode, items, stream) : launch_thin_t<f16, 2>(t, a.e.act, mode, items, stream);
}

}  // namespace mdie

#ifdef EXP_TSTAMPS
extern "C" void mdie_exp_set_thin_dbg(void* p) { mdie::g_thin_dbg = (unsigned long long*)p; }
#endif
