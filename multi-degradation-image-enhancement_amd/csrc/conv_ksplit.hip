// 3x3 convolution for the DEEPEST DenseBlock layers (encoder.dense3: BN -> ReLU -> Conv3x3(256 + 16 i -> 16) on 32x32 maps,
// models/cdan.py:41-46,65): >= 128 input channels, 16 outputs, small maps.
//
// Why a third kernel.  conv_kernel runs these layers as a fork-join per 32-channel K chunk: all four waves of a workgroup stage one
// chunk (two barriers, an LDS write -> read turn-around), then each wave issues 9 (8x8 tile) or 36 (16x16) MFMAs -- 144..576 matrix
// cycles inside a 3.1 k-cycle iteration, 8..10 iterations in a row (tools/stamp_conv.py, profiles/LEDGER.md (rounds 1-4) section 4 finding 8): 16-21 us
// for 19 MB.  Deeper prefetch, a table of segment addresses, a K split over WORKGROUPS (partial slabs + a reduce launch) all measured
// null or worse: the chain itself is the cost.  Here the K axis is split over the four WAVES of a workgroup instead:
//   * wave w owns K chunks w, w + 4, ... of the layer and the WHOLE 8x8-pixel tile (4 subtiles x 16 outputs, 36 MFMAs per chunk):
//     its patch image is wave-private LDS, written and read by the same wave -- no barrier anywhere in the chunk loop;
//   * the A operand (weights) goes from global memory straight into the MFMA registers: in the packed layout
//     [chunk][K group][tap][cout][16 B] a lane's fragment of (tap, K group lq, output lp) is 16 contiguous bytes and a 16-lane
//     group reads 256 contiguous bytes (L2-resident: every workgroup reads the same 74-88 KB) -- no weight staging at all;
//   * the next chunk's loads (7 patch units + 9 weight fragments per lane) are issued before the current chunk's MFMAs;
//   * ONE barrier at the end: every wave leaves its four partial accumulators in its own (now dead) patch region, wave w then folds
//     subtile w of all four waves in wave order 0, 1, 2, 3 and runs the epilogue for it.
// The fold order is fixed, so results are bit-reproducible and independent of the batch; they are NOT bit-identical to conv_kernel's
// (one chain over all chunks there, four chains + a fold here) -- the kernel is therefore chosen by (layer, map) alone, never by B
// (conv_ksplit_applicable), and is held to the oracle's tolerance like every other kernel, not to conv_kernel's bits.
// LDS image: one plane per 16-byte K group, rows of 16 pixels (10 used), plane stride 2688 B: with a subtile = 2 rows x 8 pixels
// (lane lp -> row lp >> 3, column lp & 7) every B-operand ds_read_b128 of every tap is conflict-free (exhaustive search over
// pitches 10..24: this is the only pitch below 24 that is).  A fragment of input rows (r, r + 1) at column shift kw serves
// (subtile r / 2, kh 0) and (subtile r / 2 - 1, kh 2) for even r, (subtile (r - 1) / 2, kh 1) for odd r: 27 reads per 36 MFMAs.
#include "conv_common.hpp"

namespace mdie {

constexpr int KS_TILE = 8, KS_PW = 10, KS_PITCH = 16;
constexpr int KS_PLANE = KS_PW * KS_PITCH * 16 + 128;       // 2688
constexpr int KS_WAVE_LDS = 4 * KS_PLANE;                   // 10752 B per wave (>= the 4 KiB of partial accumulators it holds at the end)
constexpr int KS_MAX_CIN = 512;                             // pre-activation constants of the whole layer live in LDS
constexpr int KS_LDS = 4 * KS_WAVE_LDS + 2 * KS_MAX_CIN * (int)sizeof(float);   // 47104 B per workgroup: 3 workgroups per CU
constexpr int KS_PATCH_IT = (KS_PW * KS_PW * 4 + 63) / 64;  // 7 staging units per lane (the last one: 16 lanes)
// Maps up to 40x40 (encoder.dense3 of a 256x256 .. 320x320 picture).  Measured at B = 32 (tools/stamp_ksplit.py, profiles/r04b_*):
// 32x32 maps 9.6-12.0 us per layer against conv_kernel's 16-18; 64x64 maps (dense2) 17-26 us against 18-21 -- there a workgroup
// has 4-6 chunks for four waves (one or two steps each, nothing to pipeline) and 8x8 tiles mean four times the workgroups of
// conv_kernel's 16x16, each paying the ~3 k-cycle kernel entry: dense2 stays on conv_kernel.
constexpr int KS_MAX_PIXELS = 40 * 40;

#ifdef EXP_KSTAMPS   // diagnostic build only (tools/stamp_ksplit.py): shader-clock stamps of waves 0 and 3 into a buffer passed as `residual`
#define KSTAMP(i) do { if (dbg && lane == 0 && (wave == 0 || wave == 3)) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); dbg[((size_t)patch * 2 + (wave ? 1 : 0)) * 16 + (i)] = t_; } } while (0)
#else
#define KSTAMP(i) do {} while (0)
#endif

template <typename T>
__global__ __launch_bounds__(CONV_THREADS, 3) void conv_ksplit_kernel(const ConvArgs a) {
  constexpr int VEC = Traits<T>::VEC, KC = Traits<T>::KC;
  static_assert(sizeof(T) == 2, "16-bit storage types only (fp32 layers stay on conv_kernel)");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lq = lane >> 4, lp = lane & 15;
  char* const mine = smem + wave * KS_WAVE_LDS;

  // XCD slot (fastest grid index) owns a contiguous run of tiles: neighbouring tiles share halo pixels through that XCD's L2
  const int patch = blockIdx.x * gridDim.y + blockIdx.y;
  const int tpi = a.tiles_x * a.tiles_y;
  if (patch >= tpi * a.B) return;
  const int img = patch / tpi, trem = patch - img * tpi;
  const int ty = trem / a.tiles_x, tx = trem - ty * a.tiles_x;
  const int y0 = ty * KS_TILE, x0 = tx * KS_TILE;
#ifdef EXP_KSTAMPS
  unsigned long long* dbg = (a.e.res_stride == -12345) ? reinterpret_cast<unsigned long long*>(const_cast<char*>(a.e.residual)) : nullptr;
#endif
  KSTAMP(0);

  // ---- staging geometry of this lane: unit u = lane + 64 it -> patch pixel u >> 2, K group q = lane & 3 ----
  // Every load and every LDS write below is UNCONDITIONAL (a branch around one makes the compiler's s_waitcnt pass fall back to
  // vmcnt(0) in the middle of the load cluster -- profiles/LEDGER.md (rounds 1-4) section 4, gfx950 finding 2): a unit outside the picture reads a valid
  // (clamped) pixel and is zeroed by a select before the LDS write; the lanes without a pixel in the last iteration (100 patch
  // pixels = 6 * 16 + 4) write into the 128 unused bytes at the end of their plane.
  // One register per unit: low half = pixel index inside the image (clamped), high half = LDS byte offset inside `mine`; `inside`
  // holds one bit per unit: the pixel exists (else: zero padding).
  const int q = lane & 3;
  unsigned unit[KS_PATCH_IT];
  unsigned inside = 0;
#pragma unroll
  for (int it = 0; it < KS_PATCH_IT; ++it) {
    const int pix = (lane >> 2) + 16 * it;
    const int py = pix / KS_PW, px = pix - py * KS_PW;
    const int gy = y0 + py - 1, gx = x0 + px - 1;
    const bool in_patch = pix < KS_PW * KS_PW;
    const bool ok = in_patch && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    const int cy = min(max(gy, 0), a.H - 1), cx = min(max(gx, 0), a.W - 1);
    const int dst = in_patch ? q * KS_PLANE + (py * KS_PITCH + px) * 16 : q * KS_PLANE + KS_PW * KS_PITCH * 16 + (lane >> 3) * 16;
    unit[it] = (unsigned)(cy * a.W + cx) | ((unsigned)dst << 16);
    inside |= (ok ? 1u : 0u) << it;
  }
  float* const lds_pre = reinterpret_cast<float*>(smem + 4 * KS_WAVE_LDS);
  const size_t img_pix = (size_t)img * a.H * a.W;
  constexpr int COUT = 16;                                                        // (conv_ksplit_applicable) -- taps become immediate offsets
  constexpr int wchunk_bytes = 4 * 9 * COUT * 16;
  const long long dl = a.delta ? a.delta[img] : 0;                                // (several weight sets in one launch: this tile's image selects its set)
  const char* const wlane = a.weight + dl + (lq * 9 * COUT + lp) * 16;            // + tap * COUT * 16 + chunk * wchunk_bytes

  uint4 pv[KS_PATCH_IT], wv[9];
  bool live = false;
  int live_c0 = 0;

  auto load_chunk = [&](int chunk) {
    const int c0 = chunk * KC + q * VEC;      // first stored channel of this lane's K group
    const char* sbase = nullptr;
    int sstride = 0;
#pragma unroll
    for (int s = 0; s < MDIE_MAX_SEG; ++s) {  // branch-free over all slots (conv_kernel.hpp: load_chunk)
      const int cb = a.seg[s].ch_begin, ce = a.seg[s].ch_end, st = a.seg[s].stride;
      const char* sp = a.seg[s].ptr;
      const bool hit = (s < a.nseg) & (c0 >= cb) & (c0 < ce);
      sbase = hit ? sp + (size_t)(c0 - cb) * sizeof(T) : sbase;
      sstride = hit ? st * (int)sizeof(T) : sstride;
    }
    live = sbase != nullptr;       // (false: a K group past the layer's last channel -- its weights are zero, its pixels are zeroed below)
    live_c0 = live ? c0 : 0;
    sbase = live ? sbase : a.seg[0].ptr;
    sstride = live ? sstride : a.seg[0].stride * (int)sizeof(T);
    // one 64-bit base per chunk, then a 24-bit multiply (full rate; pixel index < 4096, pixel stride < 2^20 bytes) per unit
    const char* const pb = sbase + img_pix * sstride;
#pragma unroll
    for (int it = 0; it < KS_PATCH_IT; ++it) pv[it] = *reinterpret_cast<const uint4*>(pb + __umul24(unit[it] & 0xffffu, (unsigned)sstride));
    const char* w = wlane + (size_t)chunk * wchunk_bytes;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) wv[tap] = *reinterpret_cast<const uint4*>(w + tap * COUT * 16);
  };

  f32x4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // B-operand read address of input row pair (0, 1) at column shift 0: row r adds r * KS_PITCH * 16, shift kw adds kw * 16
  const char* const xrd = mine + lq * KS_PLANE + ((lp >> 3) * KS_PITCH + (lp & 7)) * 16;

  // One step = stage the chunk that has landed, issue the next chunk's loads (when the wave has one), 36 MFMAs.  The loop runs the
  // steps that HAVE a successor -- its prefetch is unconditional -- and the last step stands behind it: a prefetch under
  // `if (chunk + 4 < nchunk)` inside the loop meets the not-taken path in a phi, and the compiler answers that with
  // s_waitcnt vmcnt(0) right behind the loads (profiles/LEDGER.md (rounds 1-4) section 4, gfx950 finding 3): this chunk's MFMAs would wait for the
  // NEXT chunk's data.
  auto stage = [&]() __attribute__((always_inline)) {
    // pre-activation (BN + ReLU; zero padding stays zero) and LDS write of this wave's patch
    const float4 s0 = *reinterpret_cast<const float4*>(lds_pre + live_c0), s1 = *reinterpret_cast<const float4*>(lds_pre + live_c0 + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(lds_pre + KS_MAX_CIN + live_c0), b1 = *reinterpret_cast<const float4*>(lds_pre + KS_MAX_CIN + live_c0 + 4);
    const f32x2 ps_[4] = {f32x2{s0.x, s0.y}, f32x2{s0.z, s0.w}, f32x2{s1.x, s1.y}, f32x2{s1.z, s1.w}};
    const f32x2 pb_[4] = {f32x2{b0.x, b0.y}, f32x2{b0.z, b0.w}, f32x2{b1.x, b1.y}, f32x2{b1.z, b1.w}};
#pragma unroll
    for (int it = 0; it < KS_PATCH_IT; ++it) {
      uint4 v = PreAct<T>::apply(pv[it], ps_, pb_);
      const bool keep = live && ((inside >> it) & 1u);
      v.x = keep ? v.x : 0u; v.y = keep ? v.y : 0u; v.z = keep ? v.z : 0u; v.w = keep ? v.w : 0u;
      *reinterpret_cast<uint4*>(mine + (unit[it] >> 16)) = v;
    }
  };
  // the wave's own LDS writes -> its own reads: no barrier, the compiler's lgkmcnt wait orders them
  auto mfmas = [&](const uint4 (&wf)[9]) __attribute__((always_inline)) {
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      uint4 xf[2];
      xf[0] = *reinterpret_cast<const uint4*>(xrd + kw * 16);
#pragma unroll
      for (int r = 0; r < 9; ++r) {
        if (r + 1 < 9) xf[(r + 1) & 1] = *reinterpret_cast<const uint4*>(xrd + ((r + 1) * KS_PITCH + kw) * 16);
        const uint4& x = xf[r & 1];
        if ((r & 1) == 0) {
          if (r / 2 < 4) acc[r / 2] = mma16<T>(wf[0 * 3 + kw], x, acc[r / 2]);            // (subtile r/2, kh 0)
          if (r >= 2) acc[r / 2 - 1] = mma16<T>(wf[2 * 3 + kw], x, acc[r / 2 - 1]);       // (subtile r/2 - 1, kh 2)
        } else if (r < 8) {
          acc[(r - 1) / 2] = mma16<T>(wf[1 * 3 + kw], x, acc[(r - 1) / 2]);               // (subtile (r-1)/2, kh 1)
        }
      }
    }
  };
  const int n_mine = (a.nchunk - wave + 3) >> 2;            // 1..4 chunks (conv_ksplit_applicable: 4 <= nchunk <= 16)
  int chunk = wave;
  load_chunk(chunk);
  KSTAMP(1);
  // pre-activation constants of the layer -> LDS once (a wave reads 16 floats of them per chunk while staging); behind the first
  // chunk's loads, so their latency and this loop's run together
  for (int c = tid; c < a.nchunk * KC; c += CONV_THREADS) {
    lds_pre[c] = c < a.cin ? param_shift(a.pre_scale, dl)[c] : 0.f;
    lds_pre[KS_MAX_CIN + c] = c < a.cin ? param_shift(a.pre_shift, dl)[c] : 0.f;
  }
  __syncthreads();
  KSTAMP(2);
  for (int i = 1; i < n_mine; ++i) {
    stage();
    uint4 wf[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) wf[tap] = wv[tap];
    __builtin_amdgcn_sched_barrier(0);
    if (i == 1) KSTAMP(3);
    chunk += 4;
    load_chunk(chunk);                                      // in flight during the MFMAs below
    __builtin_amdgcn_sched_barrier(0);
    if (i == 1) KSTAMP(4);
    mfmas(wf);
    __builtin_amdgcn_sched_barrier(0);
    if (i == 1) KSTAMP(5);
  }
  KSTAMP(6);
  stage();
  KSTAMP(7);
  mfmas(wv);
  KSTAMP(8);

  // ---- fold the four waves' partial sums (fixed order), wave w finishes subtile w ----
  {
    f32x4* red = reinterpret_cast<f32x4*>(mine);           // [subtile][lane]: this wave's patch region is dead
#pragma unroll
    for (int i = 0; i < 4; ++i) red[i * 64 + lane] = acc[i];
  }
  KSTAMP(9);
  __syncthreads();
  KSTAMP(10);
  f32x4 sum = *reinterpret_cast<const f32x4*>(smem + 0 * KS_WAVE_LDS + (wave * 64 + lane) * 16);
#pragma unroll
  for (int w = 1; w < 4; ++w) sum += *reinterpret_cast<const f32x4*>(smem + w * KS_WAVE_LDS + (wave * 64 + lane) * 16);

  const int gy = y0 + 2 * wave + (lp >> 3), gx = x0 + (lp & 7);
  const float4 sc = *reinterpret_cast<const float4*>(param_shift(a.e.post_scale, dl) + lq * 4), sh = *reinterpret_cast<const float4*>(param_shift(a.e.post_shift, dl) + lq * 4);
  float v[4] = {fmaf(sum[0], sc.x, sh.x), fmaf(sum[1], sc.y, sh.y), fmaf(sum[2], sc.z, sh.z), fmaf(sum[3], sc.w, sh.w)};
  if (a.e.act == MDIE_ACT_RELU) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
  }
  if (gy < a.H && gx < a.W) {
    T* o = reinterpret_cast<T*>(a.e.out) + (img_pix + (size_t)gy * a.W + gx) * a.e.out_stride + lq * 4;
    *reinterpret_cast<uint2*>(o) = make_uint2(Half<T>::pack(v[0], v[1]), Half<T>::pack(v[2], v[3]));
  }
  KSTAMP(11);
}

// (layer, map) only -- never the batch: the summation order differs from conv_kernel's, so one image must take the same kernel in
// every batch it is part of
bool conv_ksplit_applicable(int dtype, const ConvArgs& a, int ksize, bool has_nchw3) {
  if (dtype == MDIE_F32 || ksize != 3 || has_nchw3 || !a.pre_scale || a.cout != 16) return false;
  if (a.nchunk < 4 || a.nchunk * dtype_kc(dtype) > KS_MAX_CIN || (long)a.H * a.W > KS_MAX_PIXELS) return false;
  for (int s = 0; s < a.nseg; ++s)
    if (a.seg[s].stride * 2 >= (1 << 20)) return false;          // (24-bit multiply of pixel index and pixel stride in bytes)
#ifdef EXP_KSTAMPS
  if (a.e.pool || (a.e.residual && a.e.res_stride != -12345) || a.pool_partial || a.e.out_gs != 16) return false;
#else
  if (a.e.pool || a.e.residual || a.pool_partial || a.e.out_gs != 16) return false;
#endif
  if (a.e.act != MDIE_ACT_NONE && a.e.act != MDIE_ACT_RELU) return false;
  return true;
}

int launch_conv_ksplit(int dtype, ConvArgs& a, hipStream_t stream) {
  a.tiles_x = cdiv(a.W, KS_TILE); a.tiles_y = cdiv(a.H, KS_TILE); a.n_tiles = 1;
  const int tiles = a.tiles_x * a.tiles_y * a.B;
  const dim3 grid(8, cdiv(tiles, 8));
  TimedLaunch tl(MDIE_K_CONV3);
  if (dtype == MDIE_BF16) hipLaunchKernelGGL((conv_ksplit_kernel<bf16>), grid, dim3(CONV_THREADS), KS_LDS, stream, a);
  else hipLaunchKernelGGL((conv_ksplit_kernel<f16>), grid, dim3(CONV_THREADS), KS_LDS, stream, a);
  MDIE_LAUNCH_CHECK("mdie_conv_fwd");
  return MDIE_OK;
}

}  // namespace mdie
