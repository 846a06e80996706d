// conv_kernel with its output written ONE PLANE PER 16 CHANNELS (mdie_conv_desc.out_group_stride) -- the input gradient of a
// DenseBlock layer in training (models/cdan.py:32-53 under scaler.scale(loss).backward(), models/model.py:160-164).
//
// Why.  The gradient w.r.t. a DenseBlock feature segment is the sum of the BatchNorm-ReLU backward terms of every layer that
// consumed it (up to five).  Each term needs the segment's slice of that layer's `da` (the gradient w.r.t. the layer's activated
// input, C_l = c0 + 16 l channels per pixel).  Interleaved [pixel][C_l] rows make such a slice 32 bytes out of 64-600: adding the
// terms layer by layer (round 1-2: read x, da, the running sum; write the sum -- 4 passes over C_l channels per layer, 1.6 ms of a
// 10.9 ms step at 512x512, B = 8) moves 2.3x the bytes of forming every segment's gradient ONCE from all its consumers, and
// gathering the slices out of interleaved rows measured slower still (profiles/LEDGER.md (rounds 1-4) section 5b).  With da stored plane by plane every
// slice is a dense [pixels][16] stream, and mdie_bn_bwd_apply_multi reads x, each consumer's plane and writes the sum once.
//
// The kernels are conv_kernel's template (conv_kernel.hpp) with PLANAR = true: same staging, same MFMA order, the plain epilogue
// (no activation / pooling / residual) with the channel-group offset multiplied out.  A translation unit of its own, so the
// inference instantiations in conv.hip are not compiled next to these.
#include "conv_kernel.hpp"

namespace mdie {

template <typename T, int KS, int BN, int TILE, bool BNRED>
static int launch_planar_k(ConvArgs& a, hipStream_t stream) {
  using G = ConvGeom<KS, BN, TILE>;
  a.tiles_x = cdiv(a.W, TILE); a.tiles_y = cdiv(a.H, TILE);
  const dim3 grid(8, a.n_tiles, cdiv(a.tiles_x * a.tiles_y * a.B, 8));
  static LdsOptIn opt;
  if (!opt.ensure(reinterpret_cast<const void*>(&conv_kernel<T, KS, BN, TILE, false, true, BNRED>), G::BUF_BYTES + 8 * 1024)) return MDIE_ELAUNCH;
  TimedLaunch tl(KS == 3 ? MDIE_K_CONV3 : MDIE_K_CONV1);
  const size_t lds = G::BUF_BYTES + 2 * BN * sizeof(float) + (a.pre_scale ? (size_t)2 * a.nchunk * Traits<T>::KC * sizeof(float) : 0);
  hipLaunchKernelGGL((conv_kernel<T, KS, BN, TILE, false, true, BNRED>), grid, dim3(CONV_THREADS), lds, stream, a);
  MDIE_LAUNCH_CHECK("mdie_conv_fwd");
  return MDIE_OK;
}

template <typename T, int KS, int BN, int TILE>
static int launch_planar_t(ConvArgs& a, hipStream_t stream) {
  static_assert(ConvGeom<KS, BN, TILE>::BUF_BYTES >= (CONV_THREADS / 64) * 2 * BN * (int)sizeof(float), "the wave sums of the BNRED epilogue fit the dead patch image");
  return a.e.b_partial ? launch_planar_k<T, KS, BN, TILE, true>(a, stream) : launch_planar_k<T, KS, BN, TILE, false>(a, stream);
}

// the tile edge launch_planar_dt picks: mdie_conv_bnred_slabs (conv.hip) sizes the partial sums with it
int conv_planar_tile(int B, int H, int W, int cout) {
  const int bn = (cout % 64 == 0) ? 64 : 16;
  const long wgs16 = (long)cdiv(H, 16) * cdiv(W, 16) * B * (cout / bn);
  return wgs16 < SMALL_GRID_WGS ? 8 : 16;
}

template <typename T>
static int launch_planar_dt(ConvArgs& a, int ksize, hipStream_t stream) {
  const int bn = (a.cout % 64 == 0) ? 64 : 16;
  a.n_tiles = a.cout / bn;
  const long wgs16 = (long)cdiv(a.H, 16) * cdiv(a.W, 16) * a.B * a.n_tiles;
  const bool small = wgs16 < SMALL_GRID_WGS;
  if (ksize == 3) {
    if (bn == 64) return small ? launch_planar_t<T, 3, 64, 8>(a, stream) : launch_planar_t<T, 3, 64, 16>(a, stream);
    return small ? launch_planar_t<T, 3, 16, 8>(a, stream) : launch_planar_t<T, 3, 16, 16>(a, stream);
  }
  if (bn == 64) return small ? launch_planar_t<T, 1, 64, 8>(a, stream) : launch_planar_t<T, 1, 64, 16>(a, stream);
  return small ? launch_planar_t<T, 1, 16, 8>(a, stream) : launch_planar_t<T, 1, 16, 16>(a, stream);
}

int launch_conv_planar(int dtype, ConvArgs& a, int ksize, hipStream_t stream) {
  MDIE_REQUIRE(a.e.act == MDIE_ACT_NONE && !a.e.pool && !a.e.residual && !a.e.nchw3 && !a.pool_partial,
               "mdie_conv_fwd: out_group_stride takes the plain epilogue (no activation, pooling, residual, out_nchw3, pool_partial)");
  MDIE_REQUIRE(a.e.out_gs > 0 && a.e.out_gs % 16 == 0 && a.e.out_stride >= 16, "mdie_conv_fwd: out_group_stride %ld / out_stride %d", a.e.out_gs, a.e.out_stride);
  MDIE_SWITCH_T(dtype, return launch_planar_dt<T>(a, ksize, stream));
}

}  // namespace mdie
