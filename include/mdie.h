/*
 * mdie.h -- C ABI of libmdie_hip.so: the MI355X (gfx950) engine for the CDAN/CBAM
 * restoration path of danielluca00/Multi-Degradation-Image-Enhancement.
 *
 * The reference has no native interface of its own (it is pure PyTorch); its
 * extension point is the JSON-named class factory utils/parser.py:42-73, which
 * instantiates models.cdan.CDAN (config/low_light.json:10-15) and calls
 * nn.Module.forward (models/model.py:160,252,342).  This header is the boundary a
 * binding for that path talks to: plain pointers, sizes and small POD structs, no
 * C++ or torch types.  Every entry point names the reference code it replaces.
 *
 * Conventions
 *   - All device tensors are NHWC ("channels last"), element type `dtype`
 *     (MDIE_F32, MDIE_BF16 or MDIE_F16), pixel stride given in ELEMENTS.  Channel counts
 *     of internal tensors are multiples of 16; the 3-channel tensors of the path
 *     (input x, decoder.conv4 output, final output) are stored with 16 channels,
 *     channels 3..15 zero.
 *   - Entry points are asynchronous on `stream` (a hipStream_t passed as void*),
 *     never allocate, never synchronise; the caller owns every buffer.
 *   - Return 0 on success, a negative MDIE_E* code on error;
 *     mdie_last_error() returns a thread-local message.
 *   - Accumulation is always fp32.  MDIE_F32 uses the exact-f32 MFMA
 *     (v_mfma_f32_16x16x4_f32), MDIE_BF16 the bf16 MFMA (v_mfma_f32_16x16x32_bf16),
 *     MDIE_F16 the fp16 MFMA (v_mfma_f32_16x16x32_f16); the two 16-bit types share every
 *     layout rule stated for "bf16" below (8 elements per 16 bytes, 32-channel K chunks).
 */
#ifndef MDIE_H
#define MDIE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDIE_ABI_VERSION 27

enum { MDIE_F32 = 0, MDIE_BF16 = 1,
       MDIE_F16 = 2 /* IEEE half: the reference's mixed-precision dtype (torch.cuda.amp.autocast, models/model.py:15,159) */ };
enum { MDIE_ACT_NONE = 0, MDIE_ACT_RELU = 1, MDIE_ACT_SIGMOID = 2 };
enum {
  MDIE_OK = 0,
  MDIE_EINVAL = -1,   /* bad argument / unsupported shape */
  MDIE_ELAUNCH = -2,  /* HIP launch error */
  MDIE_ENOSPC = -3,   /* workspace too small */
  MDIE_ENOENT = -4    /* missing checkpoint entry */
};

#define MDIE_MAX_SEG 5

/* One channel segment of a concatenated NHWC view (a DenseBlock's `torch.cat(features)`,
 * models/cdan.py:35,38, is never materialised: each growth layer writes its own segment). */
typedef struct {
  const void* ptr;  /* first element of pixel (0,0,0) */
  int channels;     /* whole 16-byte groups: multiple of 8 (bf16) / 4 (f32); multiple of 16 for the training kernels */
  int stride;       /* elements between consecutive pixels (>= channels).  One exception, 16-bit types, a mdie_conv_desc that carries
                       `tr` (decoder.final_dense's folded chain): channels == 8 with stride == 4 -- a HALF group: 4 stored channels
                       per pixel (the block's 3-channel base), read with the group's 16-byte load whose upper half is the next
                       pixel's bytes; the weights of channels 4..7 must be zero (they are: the base has 3 real channels) and the
                       buffer must extend 8 bytes past its last pixel, those bytes holding FINITE values (0 * NaN is NaN in the
                       pre-activation and in the MFMA): mdie_up_add_dense0_fwd, the producer of such a buffer (base_stride 4),
                       writes zeros there, so the caller only provides the room.  Halves the base's bytes in all four launches. */
} mdie_seg;

/* ---------------------------------------------------------------------------------
 * Fused convolution:  out = pool2x2?( act( conv_k(pre(in)) * post_scale + post_shift ) + residual? )
 *   pre(x) = relu(x * pre_scale + pre_shift) when pre_scale != NULL (pre-activation
 *            BN -> ReLU of a dense layer, models/cdan.py:41-53; zero padding is applied
 *            AFTER pre(), as nn.Conv2d pads the activated tensor)
 *   conv_k = 3x3 / pad 1 / stride 1 or 1x1, weights packed by mdie_pack_conv_weight
 *   post   = conv bias + eval-mode BatchNorm folded to scale/shift
 *            (ConvBlock, models/cdan.py:15-19; decoder ConvTranspose2d+BN+ReLU, :127-129)
 *   pool   = nn.MaxPool2d(2,2) (models/cdan.py:67,75,82,89) fused into the epilogue
 * ConvTranspose2d(k3,s1,p1) is the same kernel with flipped/transposed weights
 * (mdie_pack_conv_weight(..., transposed=1)).
 * --------------------------------------------------------------------------------- */
/* The DenseBlock's transition (BatchNorm -> ReLU -> Conv1x1, models/cdan.py:48-53) folded into the producers of its input:
 *     relu(bn(cat(f0, f1, ..))) . W  ==  sum over the segments f_s of relu(bn_s(f_s)) . W_s,
 * so the kernel that has just computed a segment adds that segment's term to a running fp32 partial sum of the transition's
 * (<= 3) outputs, [B,H,W,4] floats, while the segment is still in registers; the last producer finishes the sum with the
 * transition's epilogue and never stores its own segment.  The engine does this for decoder.final_dense (models/cdan.py:119,
 * 155-157): one launch, the re-read of all 67 channels and the last growth map's write disappear.  16-bit element types, H and W
 * multiples of 16 (mdie_conv_fwd returns MDIE_EINVAL otherwise; the unfused chain -- four 3x3 layers, then the 1x1 -- is the
 * general form).  Arithmetic: the term is formed from the STORED (rounded) segment, pre-activated and rounded exactly as the
 * transition's own launch would, multiplied on the matrix pipe with fp32 accumulation; only the order of the fp32 sum over
 * segments differs from the single launch. */
typedef struct {
  const void* weight;        /* the transition's 1x1 weights as mdie_pack_conv_weight packs them (cout_stored = 16) */
  int c0;                    /* stored input channel OF THE TRANSITION at which this layer's output channel 0 sits (multiple of 8) */
  const float* pre_scale;    /* the transition's folded BatchNorm, indexed by ITS stored input channel */
  const float* pre_shift;
  const float* partial_in;   /* partial sums so far (written by the previous producer) */
  float* partial_out;        /* middle producer: partial sums out (may be partial_in) */
  const float* post_scale;   /* last producer: the transition's epilogue, [>= 4] each ... */
  const float* post_shift;
  int act;                   /* ... MDIE_ACT_SIGMOID ... */
  float* out_nchw3;          /* ... and the fp32 NCHW [B,3,H,W] destination; NULL marks a middle producer */
} mdie_tr_fuse;

typedef struct {
  int dtype;
  int B, H, W;             /* input extent; output is H x W, or H/2 x W/2 with pool */
  int ksize;               /* 3 or 1 */
  int nseg;
  mdie_seg in[MDIE_MAX_SEG];
  int cin;                 /* sum of segment channels */
  int cout;                /* stored output channels, multiple of 16 */
  const float* pre_scale;  /* [cin] or NULL */
  const float* pre_shift;  /* [cin] or NULL */
  const void* weight;      /* packed, element type = dtype */
  const float* post_scale; /* [cout] */
  const float* post_shift; /* [cout] */
  int act;
  int pool;
  const void* residual;    /* NHWC at output resolution, or NULL */
  int res_stride;
  void* out;
  int out_stride;
  float* out_nchw3;        /* optional: instead of `out`, write output channels 0..2 as fp32 NCHW [B,3,Ho,Wo]
                              (the network's final tensor, models/cdan.py:157); cout must be 16 */
  float* pool_partial;     /* optional (3x3, cout % 64 == 0, ReLU, no max-pool): additionally emit the channel sums and
                              maxima of the tensor written, [B][tiles per image][2][cout] with tiles of edge
                              mdie_conv_tile(B,H,W,cout) in raster order -- the global pools of the CBAM that consumes
                              it (models/cbam.py:41,44; mdie_cbam_desc.pool_partial) fused into their producer */
  const mdie_tr_fuse* tr;  /* optional: fold the consuming transition into this layer (above); `out` may be NULL for the last producer */
  long out_group_stride;   /* 0 or 16: channel n of pixel p at out[p * out_stride + n] (NHWC).  Otherwise ONE PLANE PER 16 CHANNELS: channel n at
                              out[(n / 16) * out_group_stride + p * out_stride + n % 16] (out_stride >= 16; plain epilogue only: no activation,
                              pooling, residual) -- the layout mdie_bn_bwd_reduce / mdie_bn_bwd_apply_multi take `da` in (da_plane), so that
                              one feature segment of several layers' input gradients is a set of dense streams */
  const struct mdie_bn_reduce_fuse* bnred;   /* optional, with out_group_stride only: see below */
  const long long* blob_delta;   /* optional, DEVICE array [B]: SEVERAL WEIGHT SETS IN ONE LAUNCH.  Image b adds blob_delta[b] bytes to every
                                    parameter pointer of this descriptor (weight, pre_* / post_* vectors, the members of `tr`): all
                                    parameter blobs of one architecture share one layout (mdie_cdan_pack_params), so one offset per image
                                    selects its weight set.  NULL: one weight set.  Not with out_group_stride / bnred (training). */
  int share_cu;            /* How the layer treats the CUs it runs on, among forms that are BIT-IDENTICAL for it (ABI 26).  0: the library's choice -- for
                              the wide layers conv_wide as ONE persistent workgroup per CU, which holds the CU and all of its LDS until the launch
                              ends: kernels of a side branch queue behind the whole layer.  2: conv_wide with twice the workgroups, each with half
                              the run of items -- every CU returns to the dispatcher half way (2 us slower alone; encoder.conv4: -29 ... -34 us for
                              the step, the DenseBlock branches slip in).  1: conv_kernel instead (three ~50 KB workgroups per CU; 89 against 80 us
                              alone; -14 ... -30 us for the step on some boxes, +10 on the fastest).  Which form makes the STEP fastest depends on
                              the box (how far its clock management pulls the matrix-dense kernel down): the host times them (CdanEngine.tune). */
} mdie_conv_desc;

/* The BatchNorm-ReLU backward SUMS fused into the input-gradient convolution of a DenseBlock layer (training).  The convolution's
 * output `da` is the gradient w.r.t. relu(bn(x)) where x = cat(x segments) (models/cdan.py:35-46); BatchNorm backward needs
 *   sum dz  and  sum dz * xhat   per channel,   dz = da * [x * scale + shift > 0]
 * over the whole batch before anything can be applied -- a pass of its own over da and x (mdie_bn_bwd_reduce: 0.7 ms of an 8 ms
 * step).  With `bnred` the convolution reads x at the pixels it has just produced and leaves, per tile, sum dz and sum dz * x in
 * partial[slab][2][cout] (mdie_conv_bnred_slabs() slabs); mdie_bn_bwd_finish folds the slabs and forms
 * sum dz * xhat = invstd * (sum dz * x - mean * sum dz): da is written once and read once (by mdie_bn_bwd_apply_multi). */
typedef struct mdie_bn_reduce_fuse {
  int nseg; mdie_seg x[MDIE_MAX_SEG];   /* the layer's input, stored channels: together exactly `cout` of the convolution */
  const float* scale; const float* shift;   /* [cout] of that layer's BatchNorm (mdie_bn_fold) */
  float* partial; size_t partial_bytes;     /* >= mdie_conv_bnred_slabs * 2 * cout floats */
} mdie_bn_reduce_fuse;
int mdie_conv_bnred_slabs(int B, int H, int W, int cout);

int mdie_conv_fwd(const mdie_conv_desc* d, void* stream);
/* tile edge (8 or 16) mdie_conv_fwd picks for this shape: pool_partial has ceil(H/t) * ceil(W/t) slabs per image */
int mdie_conv_tile(int B, int H, int W, int cout);   /* (a function of H and W alone since ABI 17: an image's pooled sums must not depend on its batch) */

/* Host-side packing of one convolution weight (fp32, PyTorch layout) into the layout
 * mdie_conv_fwd reads: [cin_chunk][q][tap][cout_pad][16 bytes], where a chunk is 64 bytes of
 * input channels (KC = 16 f32 / 32 bf16) and q = 0..3 selects its 16-byte quarter -- exactly the
 * planar LDS image of one K chunk (plane q = the `lane>>4` K group of an MFMA operand); zero padded.
 *   transposed = 0: w is [cout][cin][k][k]   (nn.Conv2d)
 *   transposed = 1: w is [cin][cout][k][k]   (nn.ConvTranspose2d, models/cdan.py:103-115);
 *                   the spatial flip is applied here.
 * cin_off / cin_pad place the `cin` real channels inside a wider stored input:
 * real channel c reads stored channel c + (c >= split ? gap : 0)  (the 3-channel base of
 * decoder.final_dense is stored in 16 channels: split = 3, gap = 13). */
size_t mdie_conv_weight_bytes(int dtype, int ksize, int cin_stored, int cout_stored);
int mdie_pack_conv_weight(int dtype, int ksize, int transposed, const float* w, int cout, int cin,
                          int cout_stored, int cin_stored, int split, int gap, void* dst);

/* ---------------------------------------------------------------------------------
 * Training (models/model.py:159-166: forward, loss, backward, Adam).  The convolutions carry ~97 % of the
 * FLOPs of a step; their three GEMMs run here:
 *   forward  mdie_conv_fwd
 *   dgrad    mdie_conv_fwd on weights repacked with the opposite `transposed` flag (the input gradient of a
 *            3x3 / pad 1 convolution is a convolution with the flipped, in/out-swapped kernel)
 *   wgrad    mdie_conv_wgrad: dW = sum over pixels of X (shifted per tap) x dY, exact-f32 MFMA, deterministic
 * mdie_pack_conv_weight_dev is mdie_pack_conv_weight on the GPU (weights change every step).
 * --------------------------------------------------------------------------------- */
int mdie_pack_conv_weight_dev(int dtype, int ksize, int transposed, const float* w_dev, int cout, int cin,
                              int cout_stored, int cin_stored, int split, int gap, void* dst_dev, void* stream);
/* All weight repacks of a training step in one launch: `jobs_dev` is a DEVICE array of n_jobs descriptors (same meaning as the
 * arguments of mdie_pack_conv_weight_dev; w and dst are device pointers that stay valid, so the table is built once). */
typedef struct {
  const float* w; void* dst;
  int ksize, transposed, cout, cin, cout_stored, cin_stored, split, gap;
  int out_split, out_gap;  /* the same placement for the OUTPUT channels: real o >= out_split is stored at o + out_gap (the input-gradient
                              form of a layer whose input has a gap: decoder.final_dense).  No gap: out_split = cout, out_gap = 0 */
} mdie_pack_job;
int mdie_pack_conv_weights_batch(int dtype, const mdie_pack_job* jobs_dev, int n_jobs, void* stream);
/* one job, given on the HOST (w and dst inside it are device pointers) */
int mdie_pack_conv_weight_job(int dtype, const mdie_pack_job* job, void* stream);

typedef struct {
  int dtype;               /* element type of x segments and dy */
  int B, H, W;
  int ksize;               /* 3 or 1 */
  int transposed;          /* layout of dw: 0 = [cout][cin][k][k] (nn.Conv2d), 1 = [cin][cout][k][k] flipped (nn.ConvTranspose2d) */
  int nseg;
  mdie_seg in[MDIE_MAX_SEG];   /* the convolution's input (stored channels, NHWC) */
  int cin, cout;           /* real channel counts of dw */
  int cout_stored;         /* stored channels of dy */
  int split, gap;          /* real input channel c >= split is stored at c + gap */
  const void* dy; int dy_stride;
  float* dw;               /* fp32, PyTorch layout */
  void* workspace; size_t workspace_bytes;   /* >= mdie_conv_wgrad_workspace_bytes */
  const float* pre_scale;  /* optional, per stored input channel: the convolution's input was relu(x * pre_scale + pre_shift), */
  const float* pre_shift;  /* applied while staging exactly as mdie_conv_fwd does (dense layers, models/cdan.py:41-46) */
} mdie_wgrad_desc;

size_t mdie_conv_wgrad_workspace_bytes(int B, int H, int W, int ksize, int cin_stored, int cout_stored);
int mdie_conv_wgrad(const mdie_wgrad_desc* d, void* stream);

/* First layer, straight from the network input: out = pool2x2?(act(conv3x3(x) * post_scale + post_shift))
 * with x fp32 NCHW [B,3,H,W] (encoder.conv1 + maxpool, models/cdan.py:58,74-75).  K = 27 is im2col'ed
 * into two MFMA steps; weights packed by mdie_pack_conv_first_weight:
 * [step][cout_stored][64 bytes], two steps.  fp32: element k = tap*3 + c (k < 27), 16 per step.  16-bit types: element
 * k' = tap*4 + c (c < 3; a pixel of the kernel's [pixel][4] LDS patch is then 8 aligned bytes of the operand), 32 per step:
 * taps 0..7 in step 0, tap 8 in step 1. */
typedef struct {
  int dtype;
  int B, H, W;
  const float* x;
  const void* weight;
  const float* post_scale;
  const float* post_shift;
  int cout;                /* stored output channels, multiple of 16 */
  int act;
  int pool;
  void* out;
  int out_stride;
  const long long* blob_delta;   /* optional, device [B]: per-image byte offset of the parameter pointers (see mdie_conv_desc) */
} mdie_conv_first_desc;

int mdie_conv_first_fwd(const mdie_conv_first_desc* d, void* stream);
size_t mdie_conv_first_weight_bytes(int dtype, int cout_stored);
/* w: [cout][3][3][3] fp32 (nn.Conv2d) */
int mdie_pack_conv_first_weight(int dtype, const float* w, int cout, int cout_stored, void* dst);

/* ---------------------------------------------------------------------------------
 * CBAM (models/cbam.py:84-95) as four passes over an NHWC tensor x[B,H,W,C]:
 *   1. mdie_cbam_pool      per-(image, channel) sum and max over H*W  (cbam.py:41,44)
 *   2. mdie_cbam_gate      att = MLP(avg) + MLP(max); gate = sigmoid(att)   (cbam.py:30-35,42-59)
 *   3. mdie_cbam_chanpool  y = x*gate; map = (max_c y, mean_c y)            (cbam.py:59,68-70)
 *   4. mdie_cbam_spatial   s = sigmoid(BN(conv7x7(map))); out = x*gate*s [* mul]
 *                          (cbam.py:72-82; `out *= denses[k]`, models/cdan.py:133,141,149)
 * `out` may be `x` itself (in place); otherwise it must not overlap `x` or `mul` (pass 4 batches its loads ahead of
 * its stores).  C >= 256 runs pass 2 as its own launch, smaller C folds it into pass 3.
 * --------------------------------------------------------------------------------- */
typedef struct {
  int dtype;
  int B, H, W, C;
  const void* x;       int x_stride;
  const float* w1;     /* [C/16][C]   ChannelGate.mlp.1.weight */
  const float* b1;     /* [C/16] */
  const float* w2;     /* [C][C/16]   ChannelGate.mlp.3.weight */
  const float* b2;     /* [C] */
  const float* w7;     /* [2][7][7]   SpatialGate.spatial.conv.weight (max plane first) */
  const float* bn;     /* device [2]: eval-mode BN(1) folded to (scale, shift) */
  const void* mul;     int mul_stride;  /* optional elementwise multiplicand, or NULL */
  void* out;           int out_stride;
  void* workspace;     size_t workspace_bytes; /* >= mdie_cbam_workspace_bytes */
  const float* pool_partial; int pool_slabs; /* optional: per-(image, slab) channel sums / maxima of x,
                                                [B][pool_slabs][2][C], already produced by the kernel that wrote x
                                                (mdie_upsample2x_add); pass 1 is skipped */
  const long long* blob_delta;   /* optional, device [B]: per-image byte offset of w1, b1, w2, b2, w7, bn (see mdie_conv_desc) */
} mdie_cbam_desc;

size_t mdie_cbam_workspace_bytes(int B, int H, int W, int C);
int mdie_cbam_fwd(const mdie_cbam_desc* d, void* stream);

/* the same pipeline stopped after pass 2 + channel scaling only (for stage-wise parity tests) */
int mdie_cbam_channel_only_fwd(const mdie_cbam_desc* d, void* stream);

/* One named fp32 tensor of a checkpoint (host memory). */
typedef struct {
  const char* name;   /* state_dict key, e.g. "encoder.conv1.conv.weight" */
  const float* data;  /* host pointer, fp32, PyTorch layout (int64 counters may be omitted) */
  int64_t numel;
} mdie_tensor;

/* out[B,2H,2W,C] = bilinear_x2(lo[B,H,W,C]) + skip  (F.interpolate(scale_factor=2, 'bilinear',
 * align_corners=False) + torch.add, models/cdan.py:137-138,145-146,153-154) */
int mdie_upsample2x_add(int dtype, int B, int H, int W, int C, const void* lo, int lo_stride,
                        const void* skip, int skip_stride, void* out, int out_stride, void* stream);
/* Same, additionally reducing the tensor it writes for the CBAM that consumes it next (models/cdan.py:139,147):
 * pool_partial[B][pool_slabs][2][C] receives per-slab channel sums and maxima (fp32); pool_slabs in [1, 256] is the
 * number of workgroups per image.  mdie_pool_slabs(H_out, W_out) is what the engine uses: a function of the
 * resolution ONLY (32, or 128 from 256x256 maps up), never of the batch size, so that the fp32 summation order -- and
 * with it every output bit -- of an image does not depend on which batch it is in. */
#define MDIE_POOL_SLABS_MAX 256
int mdie_pool_slabs(int H_out, int W_out);
int mdie_upsample2x_add_pool(int dtype, int B, int H, int W, int C, const void* lo, int lo_stride,
                             const void* skip, int skip_stride, void* out, int out_stride, float* pool_partial,
                             int pool_slabs, void* stream);

/* Same, for the last decoder stage where the skip is the network input itself (`torch.add(out, x)`,
 * models/cdan.py:153-154): lo NHWC [B,H,W,lo_stride] (channels 0..2; 16-byte aligned, lo_stride a multiple of 4:
 * channels 0..3 of a tap are read with one load), x fp32 NCHW [B,3,2H,2W],
 * out NHWC [B,2H,2W,out_channels] (channels 3.. zero); out_channels = 16, or one 16-byte group per pixel (8 bf16 /
 * 4 f32): the engine stores this 3-channel tensor -- read five times by decoder.final_dense -- that narrow. */
int mdie_upsample2x_add_nchw3(int dtype, int B, int H, int W, const void* lo, int lo_stride, const float* x_nchw,
                              void* out, int out_channels, void* stream);

/* Boundary layout changes: fp32 NCHW [B,3,H,W] <-> NHWC with 16 stored channels. */
int mdie_nchw3_to_nhwc16(int dtype, int B, int H, int W, const float* x_nchw, void* out, void* stream);
int mdie_nhwc16_to_nchw3(int dtype, int B, int H, int W, const void* in, float* y_nchw, void* stream);
/* Generic converters used by the per-op tests: fp32 NCHW [B,C,H,W] <-> NHWC dtype, C % 16 == 0 */
int mdie_nchw_to_nhwc(int dtype, int B, int C, int H, int W, const float* x_nchw, void* out, void* stream);
int mdie_nhwc_to_nchw(int dtype, int B, int C, int H, int W, const void* in, float* y_nchw, void* stream);

/* ---------------------------------------------------------------------------------
 * Whole network: CDAN.forward (models/cdan.py:171-176) in eval mode
 * (network.eval(), models/model.py:232: running-stat BatchNorm, dropout = identity).
 *
 *   blob   = mdie_cdan_pack_params(...)  host side, once per checkpoint
 *   y      = mdie_cdan_forward(blob_dev, x, ...)  one call enqueues every kernel
 * --------------------------------------------------------------------------------- */
size_t mdie_cdan_param_bytes(int dtype);
/* Packs the reference checkpoint (the 236-entry state_dict written by models/base.py:52-55)
 * into one relocatable blob: packed conv weights, folded BN scale/shift vectors, CBAM MLPs. */
int mdie_cdan_pack_params(int dtype, const mdie_tensor* tensors, int n, void* blob_host, size_t blob_bytes);

size_t mdie_cdan_workspace_bytes(int dtype, int B, int H, int W);

/* names of intermediate tensors mdie_cdan_forward can expose for stage-wise parity
 * (NHWC in workspace; the host copies them out through mdie_nhwc_to_nchw) */
typedef struct {
  const void* ptr; int channels; int stride; int H, W;
} mdie_tap;
enum { MDIE_TAP_SKIP0 = 0, MDIE_TAP_SKIP1, MDIE_TAP_SKIP2, MDIE_TAP_DENSE0, MDIE_TAP_DENSE1,
       MDIE_TAP_DENSE2, MDIE_TAP_ENC, MDIE_TAP_BOTT, MDIE_TAP_DEC1, MDIE_TAP_DEC2, MDIE_TAP_DEC3,
       MDIE_TAP_DEC4, MDIE_TAP_COUNT };

/* Last decoder stage + first layer of decoder.final_dense in one launch (models/cdan.py:153-155 and DenseBlock layer 0,
 * :35-36,41-46):  base = bilinear_x2(lo)[:, :3] + x ;  g0 = conv3x3(relu(base * pre_scale + pre_shift)) + bias.
 * K = 27 is im2col-ed into two MFMA steps per 16 pixels (csrc/updense0.hip).  `weight` is the layer's [16,3,3,3] weight in
 * mdie_pack_conv_first_weight's layout (cout_stored = 16); `base` gets base_channels (16, or one 16-byte group) stored
 * channels per pixel, channels 3.. zero; g0 is written with 16 channels at pixel stride g0_stride. */
typedef struct {
  int dtype;
  int B, H, W;                 /* output extent; lo is [B, H/2, W/2, >= 4 channels] */
  const void* lo; int lo_stride;
  const float* x;              /* fp32 NCHW [B,3,H,W] */
  void* base; int base_channels;
  int base_stride;             /* elements between consecutive pixels of `base`; 0 = base_channels.  With `tr` (16-bit types) 4 is
                                  accepted for base_channels == 8: only the first 4 channels of the group are stored, and 8 zero bytes are written behind the
                                  last pixel of the buffer, which must have room for them (mdie_seg) */
  const void* weight;
  const float* pre_scale; const float* pre_shift;   /* >= 3 entries */
  const float* bias;           /* [16] */
  void* g0; int g0_stride;
  const mdie_tr_fuse* tr;      /* optional (16-bit types): start the transition's partial sums with the terms of `base` (the transition's
                                  stored channels 0..2) and of g0 (stored channels tr->c0 ..); partial_in is ignored, partial_out written */
  const long long* blob_delta; /* optional, device [B]: per-image byte offset of the parameter pointers (see mdie_conv_desc) */
} mdie_up_dense0_desc;
int mdie_up_add_dense0_fwd(const mdie_up_dense0_desc* d, void* stream);

/* decoder.final_dense as ONE launch (models/cdan.py:22-53,119,153-157; csrc/final_block.hip, ABI 27):
 *     base = bilinear_x2(lo)[:, :3] + x ;  four DenseBlock layers (BN -> ReLU -> Conv3x3, 16 channels each) ;
 *     y = sigmoid(Conv1x1(relu(bn_t(cat(base, g0..g3)))))
 * A workgroup computes a 16 x 8 tile of y from the 24 x 16 base patch around it; the growth maps stay in LDS (the halo rings are
 * recomputed by the neighbouring tiles), nothing but `lo`, `x` and `y` touches HBM.  16-bit element types, H a multiple of 8, W of 16.
 * Parameters are exactly the chain's (mdie_up_add_dense0_fwd + three mdie_conv_fwd with mdie_tr_fuse): layer 0's weight in
 * mdie_pack_conv_first_weight's layout, layers 1..3 as mdie_pack_conv_weight packs them (ksize 3, cout_stored 16, cin_stored
 * 8 + 16 l, split 3, gap 5: the base is one 8-channel group), the transition likewise (ksize 1, cin_stored 72); pre_scale / pre_shift
 * by STORED input channel.  Arithmetic is the chain's operation for operation: `y` is bit-identical to it.
 * Measured (round 6, B = 32, 256 x 256, bf16): 292-319 us against 235 us for the chain's four launches -- the block removes ~1 GB of HBM traffic per
 * step and is bound by vector-instruction issue and per-wave latency chains instead (2 waves per SIMD: 218-256 registers, 78 KB of LDS).  The engine
 * therefore runs the chain by default (MDIE_FWD_BLOCK_TAIL selects this entry point). */
typedef struct {
  int dtype;
  int B, H, W;                   /* output extent; lo is [B, H/2, W/2, >= 4 channels] */
  const void* lo; int lo_stride; /* decoder.conv4's output (NHWC), elements between pixels */
  const float* x;                /* fp32 NCHW [B,3,H,W] */
  const void* w0;                /* layer 0 */
  const void* w[3];              /* layers 1..3 */
  const float* pre_scale[4]; const float* pre_shift[4];     /* folded pre-activation BatchNorm of layers 0..3 */
  const float* post_scale[4]; const float* post_shift[4];   /* [16] each: (1, bias); layer 0 uses post_shift only */
  const void* wt;                /* the transition's 1x1 weights */
  const float* tr_pre_scale; const float* tr_pre_shift;     /* [72] */
  const float* tr_post_scale; const float* tr_post_shift;   /* [>= 3]: (1, bias) */
  float* y;                      /* fp32 NCHW [B,3,H,W] */
} mdie_final_dense_desc;
int mdie_final_dense_fwd(const mdie_final_dense_desc* d, void* stream);

/* Concurrency of the three encoder DenseBlocks.  dense_k depends only on the pooled block output
 * o_k and is first consumed by the decoder (`out *= denses[k]`, models/cdan.py:133,141,149), so the
 * plan runs it beside the main chain from right after conv_k until the matching decoder CBAM:
 * the thin cout=16 launches then overlap the MFMA-heavy main chain.
 *   - `stream` NOT capturing: the block goes to a side stream of `aux` (event fork / event join); without `aux`
 *     everything stays on `stream`.
 *   - `stream` capturing (hipStreamBeginCapture on it, or on a stream it was forked from -- any depth): the block
 *     becomes a parallel BRANCH OF THE GRAPH through the stream's capture dependency set
 *     (hipStreamUpdateCaptureDependencies); `aux` is not used and no stream of the library ever joins the caller's
 *     capture, so any legal capture pattern of the caller stays legal (two engines on two forked streams, several
 *     captures in a row, engines destroyed in between: tests/test_gpu_parity.py::test_capture_*).
 * An mdie_aux belongs to ONE in-flight forward at a time: two forwards that may overlap (different streams) need two
 * handles and two workspaces.  Create/destroy outside graph capture.  Every fork is joined before
 * mdie_cdan_forward returns, also on an error return. */
int mdie_aux_create(void** aux);
void mdie_aux_destroy(void* aux);

/* x, y: fp32 NCHW [B,3,H,W] device pointers; H, W multiples of 8.
 * taps (optional, host array of MDIE_TAP_COUNT) is filled with workspace views.
 * launch_ms (optional, host array of capacity max_launches) turns on the instrumented mode:
 * a hipEvent pair around every kernel launch, synchronised at the end (NOT capturable);
 * n_launches receives the count, launch_kind[i] an MDIE_K* id; launch_info (optional, capacity max_launches) receives, per
 * launch, the layer it belongs to and ITS share of the fused-schedule model of SURVEY.md section 8d (the shares of one forward
 * sum to mdie_cdan_algorithmic_bytes / mdie_cdan_flops: a transition folded into its producers is booked with them). */
typedef struct {
  char label[40];          /* e.g. "enc.conv2+pool", "dense3.l1", "cbam2.spatial*d2", "final.l3+tr+sigmoid" */
  double alg_bytes;        /* algorithmic HBM bytes of this launch (activations + the parameters it reads) */
  double flops;            /* 2 * MAC */
} mdie_launch_info;

typedef struct {
  int dtype;
  int B, H, W;
  const void* params;      /* device copy of the packed blob */
  const float* x;
  float* y;
  void* workspace;  size_t workspace_bytes;
  mdie_tap* taps;
  int flags;                /* MDIE_FWD_* bits */
  void* aux;                /* mdie_aux_create handle or NULL; used only when `stream` is not capturing (see above) */
  float* launch_ms; int* launch_kind; int max_launches; int* n_launches;
  mdie_launch_info* launch_info;
  const long long* blob_delta;  /* optional, device [B]: image b runs with the parameter blob at (char*)params + blob_delta[b] -- a batch of
                                   mixed tasks (BASELINE configs[3]) as ONE launch chain instead of one chain per task group.  Every blob
                                   must be a mdie_cdan_pack_params blob of this dtype; images of one task should be contiguous (the
                                   persistent kernels reload their weights where the offset changes along their run of tiles). */
} mdie_cdan_fwd_desc;

enum { /* 1: was MDIE_FWD_FUSED_TAIL (the whole decoder tail as one launch, round 1): 517 us against 387 for the chain it replaced,
          removed in round 5 (ABI 25) */
       MDIE_FWD_SERIAL = 2     /* keep the encoder DenseBlocks in line with the main chain (no side streams, no graph branches) */,
       MDIE_FWD_GENERAL_TAIL = 4 /* decoder.final_dense as the general chain (3x3 layers, then the 1x1 launch) also where the transition
                                    could be folded into its producers (mdie_tr_fuse): the form fp32 and ragged extents always take */,
       MDIE_FWD_SHARE_CU_CONV4 = 16 /* encoder.conv4 with mdie_conv_desc.share_cu = 1: the layer the three DenseBlock branches run beside.  Results are
                                       bit-identical either way; which is faster depends on the box (CdanEngine.tune times the forms) */,
       MDIE_FWD_YIELD_CU_CONV4 = 32 /* encoder.conv4 with mdie_conv_desc.share_cu = 2 (wins over 16 when both are set) */,
       MDIE_FWD_LATE_DENSE1 = 64    /* the dense1 branch (needed last, by cbam3) starts behind decoder.conv1 instead of behind encoder.conv4: a schedule,
                                       not arithmetic -- bit-identical; another candidate of CdanEngine.tune */,
       MDIE_FWD_BLOCK_TAIL = 128    /* decoder.final_dense as ONE launch (mdie_final_dense_fwd, ABI 27) wherever the folded chain of four launches
                                       would run with one weight set and no taps: BIT-IDENTICAL to the chain, and slower -- 292-319 us against 235 us at
                                       B = 32, 256x256, bf16 (round 6: instruction issue and per-wave latency chains, profiles/r06*_final_block_*) --
                                       so it is opt-in: A/B runs, tests, and the starting point of whoever takes the block's VALU work down further */
       /* 8: was MDIE_FWD_FUSED_CBAM3 (cbam3's last pass fused into decoder.conv4, round 4): 68 us against 41 + 27, removed in round 5 */ };

enum { MDIE_K_LAYOUT = 0, MDIE_K_CONV3 = 1, MDIE_K_CONV1 = 2, MDIE_K_CBAM_POOL = 3, MDIE_K_CBAM_GATE = 4,
       MDIE_K_CBAM_CHANPOOL = 5, MDIE_K_CBAM_SPATIAL = 6, MDIE_K_UPSAMPLE = 7, MDIE_K_COUNT = 8 };

int mdie_cdan_forward(const mdie_cdan_fwd_desc* d, void* stream);

/* algorithmic work of one forward: FLOPs (2*MAC of every conv/linear)
 * and fused-schedule activation bytes for element size `esize` */
double mdie_cdan_flops(int B, int H, int W);
double mdie_cdan_algorithmic_bytes(int B, int H, int W, int esize);

/* ---------------------------------------------------------------------------------
 * Either side of the network in the reference's inference loop.
 * Images are 3-channel; fp32 NCHW [B,3,H,W] in [0,1] unless noted.
 * --------------------------------------------------------------------------------- */
/* feed: uint8 HWC [B,H,W,3] -> fp32 NCHW / 255  (albumentations Normalize(mean 0, std 1, max 255) + ToTensorV2,
 * utils/transforms_factory.py:78-81) */
int mdie_u8hwc_to_f32nchw(int B, int H, int W, const uint8_t* in, float* out, void* stream);
/* output: (img * 255).clip(0, 255).astype(uint8), CHW -> HWC  (models/model.py:80-84) */
int mdie_f32nchw_to_u8hwc(int B, int H, int W, const float* in, uint8_t* out, void* stream);

/* utils/post_processing.py ops, applied in order (utils/postprocessing_factory.py:19-41).
 * The reference's `if images.max() > 1: images /= 255` guard is unreachable on this path
 * (inputs are sigmoid outputs; every op clamps to [0,1]) and is not implemented. */
enum { MDIE_PP_CONTRAST = 0 /* param = contrast_factor, post_processing.py:5-15 */,
       MDIE_PP_COLOR = 1    /* param = saturation_factor, :18-30 */,
       MDIE_PP_SHARPEN = 2  /* param = strength, :33-54 */,
       MDIE_PP_DENOISE = 3  /* param = sigma, :57-77 */ };
typedef struct { int kind; float param; } mdie_pp_op;
size_t mdie_postprocess_workspace_bytes(int B, int H, int W);
/* ops: HOST array.  out_f32 (NCHW) and/or out_u8_hwc ([B,H,W,3]) receive the result; either may be NULL. */
int mdie_postprocess(int B, int H, int W, const float* y, const mdie_pp_op* ops, int nops, float* out_f32,
                     uint8_t* out_u8_hwc, void* workspace, size_t workspace_bytes, void* stream);

/* Batch PSNR and SSIM as torchmetrics computes them with default arguments
 * (utils/metrics_factory.py:76,87; restated, torchmetrics is not available offline):
 *   out2[0] = 10 log10(R^2 / MSE), R = max(target, 0) - min(target, 0), MSE over every element
 *   out2[1] = mean SSIM, 11x11 Gaussian (sigma 1.5), k1 .01, k2 .03, data range = max(range(pred), range(target)),
 *             windows fully inside the picture (reflect-pad 5, filter, crop 5)
 * out2: device float[2].  H, W > 10. */
size_t mdie_metrics_workspace_bytes(int B, int H, int W);
int mdie_psnr_ssim(int B, int H, int W, const float* pred, const float* target, float* out2, void* workspace,
                   size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------
 * Training-mode glue (csrc/bn.hip): batch-statistic BatchNorm (models/cdan.py:12,43,50,105-116) split into
 * statistics / fold / fused apply, its backward, and the ops fused with it (ReLU, nn.MaxPool2d(2,2) :67,
 * nn.Dropout(0.2) :68, F.interpolate x2 + add :137-138, the final sigmoid :157).  All tensors NHWC with a
 * pixel stride in elements; C a multiple of 16.
 * --------------------------------------------------------------------------------- */
size_t mdie_bn_workspace_bytes(int C);
/* mean[C], var[C] (biased) over N pixels */
int mdie_bn_stats(int dtype, long N, const void* x, int C, int stride, float* mean, float* var, void* workspace,
                  size_t workspace_bytes, void* stream);
/* mdie_bn_stats and mdie_bn_fold in two launches instead of three: the statistics of x's C channels go to mean / var, then
 * (gamma non-null) the fold runs over C_fold stored channels whose statistics are fold_mean / fold_var[0 .. C_fold) -- mean / var
 * must point INSIDE that range (a DenseBlock layer normalises the concatenation of earlier maps, whose statistics are already
 * there, and the map just written).  x == NULL: the partial sums [n_partial][2][C] (channel sums, sums of squares) are already
 * in `workspace`, left there by the convolution that produced the tensor (mdie_conv_desc.bn_partial): one launch. */
typedef struct {
  int dtype; long N;                 /* pixels */
  const void* x; int C, stride;      /* the tensor whose statistics are new (or NULL) */
  float* mean; float* var;
  void* workspace; size_t workspace_bytes; int n_partial;
  int C_fold, C_real, split, gap;    /* as mdie_bn_fold */
  const float* fold_mean; const float* fold_var; const float* gamma; const float* beta; float eps, momentum;
  float* running_mean; float* running_var; float* scale; float* shift; float* invstd;
} mdie_bn_stats_fold_desc;
int mdie_bn_stats_fold(const mdie_bn_stats_fold_desc* d, void* stream);
/* scale = gamma / sqrt(var + eps), shift = beta - mean * scale, invstd, per STORED channel (mean / var / outputs are
 * indexed by stored channel; gamma / beta / running_* by real channel: real c >= split is stored at c + gap; padding
 * gets scale = shift = 0).  running_* (nullable) are updated with `momentum` and the unbiased variance. */
int mdie_bn_fold(int C_stored, int C_real, int split, int gap, const float* mean, const float* var, const float* gamma,
                 const float* beta, float eps, float momentum, long count, float* running_mean, float* running_var,
                 float* scale, float* shift, float* invstd, void* stream);
/* out = pool2x2?(relu(y * scale + shift)); out_drop = dropout_p(out) (mask = hash(seed, element index), recomputed by
 * the backward); either output may be NULL */
int mdie_bn_act_pool_fwd(int dtype, int B, int H, int W, int C, const void* y, int y_stride, const float* scale,
                         const float* shift, int pool, void* out, int out_stride, void* out_drop, int drop_stride,
                         float p, unsigned seed, const unsigned* seed_dev, void* stream);
typedef struct {
  int dtype, B, H, W, C, c_real;          /* H, W: resolution of y */
  const void* y; int y_stride;
  const float *scale, *shift, *mean, *invstd;
  int pool;
  const void* d_out; int d_out_stride;    /* gradient w.r.t. `out` (or NULL) */
  const void* d_drop; int d_drop_stride;  /* gradient w.r.t. `out_drop` (or NULL) */
  float p; unsigned seed;
  const unsigned* seed_dev;               /* as in mdie_bn_act_pool_fwd: the SAME counter value must be in place for forward and backward */
  void* dz; int dz_stride;                /* out: masked, pool-routed gradient at y's resolution */
  float* dgamma; float* dbeta;            /* out: [c_real] */
  float* coef;                            /* out: [2][C], consumed by mdie_bn_bwd_apply */
  void* workspace; size_t workspace_bytes;
  int two_pass;                           /* 1: dz receives dL/dy itself (BatchNorm backward applied: no mdie_bn_bwd_apply call follows) -- a sums-only
                                             pass, the fold, and a pass that forms the masked gradient again and stores the finished value: the
                                             full-resolution gradient is written once and never read back */
} mdie_bn_pool_bwd_desc;
int mdie_bn_act_pool_bwd(const mdie_bn_pool_bwd_desc* d, void* stream);
/* out = up2x?(relu(y * scale + shift)) + skip (skip nullable); y at [B,H,W], out / skip at [B,2H,2W] when up */
int mdie_bn_act_up_add_fwd(int dtype, int B, int H, int W, int C, const void* y, int y_stride, const float* scale,
                           const float* shift, int up, const void* skip, int skip_stride, void* out, int out_stride,
                           void* stream);
typedef struct {
  int dtype, B, H, W, C, c_real;          /* H, W: resolution of y (low) */
  const void* y; int y_stride;
  const float *scale, *shift, *mean, *invstd;
  const void* dout; int dout_stride;      /* [B,2H,2W,C] */
  void* dz; int dz_stride;
  float* dgamma; float* dbeta;
  float* coef;
  void* workspace; size_t workspace_bytes;
} mdie_bn_up_bwd_desc;
int mdie_bn_act_up_bwd(const mdie_bn_up_bwd_desc* d, void* stream);
/* BatchNorm backward proper.  reduce: dz = da * [x * scale + shift > 0] (relu != 0) -> dgamma, dbeta, coef.
 * apply: g (=|+=, bit s of `accumulate` per segment) scale * (dz - coef[0] - xhat * coef[1]). */
typedef struct {
  int dtype; long N;
  int nseg; mdie_seg x[MDIE_MAX_SEG];     /* the normalised tensor (a concatenation of segments) */
  mdie_seg g[MDIE_MAX_SEG];               /* apply: destination, same partition */
  unsigned accumulate;
  /* optional fp32 accumulators for 16-bit tensors (a DenseBlock segment collects the gradients of up to five consuming
   * layers: summing them in bf16 would round five times).  acc32[s].ptr != NULL: running sums of segment s live in
   * acc32[s] (fp32, same channel count); `accumulate` then says whether acc32[s] already holds earlier contributions.
   * Channels >= final_from[s] of the segment receive their LAST contribution in this call and are written to g[s] in
   * `dtype`; the channels below stay in acc32[s]. */
  mdie_seg acc32[MDIE_MAX_SEG];
  int final_from[MDIE_MAX_SEG];
  const void* da; int da_stride;
  const float *mean, *invstd, *scale, *shift;
  int relu;
  int c_real, split, gap;                 /* reduce: layout of dgamma / dbeta */
  float* dgamma; float* dbeta;
  float* coef;                            /* [2][C] */
  void* workspace; size_t workspace_bytes;
  int coef_stride;                        /* apply: floats between coef's two rows; 0 = C (a descriptor that covers only the trailing
                                             segments of the tensor mdie_bn_bwd_reduce ran over passes its pointers advanced and this stride) */
  long da_plane;                          /* 0: da is [N][da_stride] rows.  Otherwise da is stored one plane per 16 channels (mdie_conv_desc.
                                             out_group_stride): channel c of pixel p at da[(c / 16) * da_plane + p * da_stride + c % 16] */
} mdie_bn_bwd_desc;
int mdie_bn_bwd_reduce(const mdie_bn_bwd_desc* d, void* stream);
int mdie_bn_bwd_apply(const mdie_bn_bwd_desc* d, void* stream);
/* mdie_bn_bwd_reduce's second half alone, for sums an input-gradient convolution left (mdie_conv_desc.bnred: per slab, sum dz and
 * sum dz * x): folds the slabs in order and writes dgamma, dbeta (real-channel layout: c_real / split / gap) and coef [2][C]. */
typedef struct {
  int C; long N;                          /* stored channels, pixels */
  const float* partial; int n_partial;    /* [n_partial][2][C] */
  const float* mean; const float* invstd; /* [C] */
  int c_real, split, gap;
  float* dgamma; float* dbeta; float* coef;
} mdie_bn_bwd_finish_desc;
int mdie_bn_bwd_finish(const mdie_bn_bwd_finish_desc* d, void* stream);
/* The same backward for a tensor that SEVERAL BatchNorm layers normalise -- the input of a DenseBlock, which each of its four
 * layers and its transition see through cat(features) (models/cdan.py:35,38) -- in ONE pass:
 *     g = sum_j scale_j * (da_j * [x * scale_j + shift_j > 0] - coef_j[0] - xhat * coef_j[1])      (written, rounded once)
 * over channels [0, C) of x; da_j is the gradient w.r.t. layer j's activated input, whose row begins with these C channels;
 * scale_j / shift_j / coef_j are that layer's folded constants and mdie_bn_bwd_reduce's output, indexed from the tensor's
 * channel 0 (coef_j: [2][coef_stride_j]); mean / invstd are the tensor's batch statistics.  Replaces nlayer read-modify-write
 * passes of mdie_bn_bwd_apply over the tensor (4 * nlayer passes over its channels) by nlayer + 2. */
typedef struct {
  int dtype; long N; int C;
  const void* x; int x_stride;
  void* g; int g_stride;
  const float* mean; const float* invstd;
  int nlayer;                                 /* 1..5 */
  const void* da[5]; int da_stride[5];
  const float* scale[5]; const float* shift[5]; const float* coef[5]; int coef_stride[5];
  long da_plane[5];                           /* 0, or the plane stride of da[j] (one plane per 16 channels, as mdie_bn_bwd_desc.da_plane);
                                                 da[j] then points at the plane of the tensor's channel 0 */
} mdie_bn_bwd_multi_desc;
int mdie_bn_bwd_apply_multi(const mdie_bn_bwd_multi_desc* d, void* stream);
/* dz[NHWC16] = grad[NCHW3] * y * (1 - y), padding channels zero (torch.sigmoid, models/cdan.py:157) */
int mdie_sigmoid_bwd_nchw3(int dtype, int B, int H, int W, const float* grad_nchw, const float* y_nchw, void* dz_nhwc16,
                           int dz_stride, void* stream);

/* CBAM in training mode (models/cbam.py:37-60,68-82,91-95): out = x * g * s (* mul), the spatial gate's
 * BatchNorm2d(1) on batch statistics (momentum / eps given, running statistics updated when non-NULL), and the full
 * backward.  The forward fills the saved-state buffers the backward reads (caller-owned, all fp32 / int32):
 *   gate [B][C], amax_idx [B][C], pooled [B][2][C], comp [B][H][W][2], smap [B][H][W], bnc [4].
 * arg-max ties go to the first index in scan order (F.max_pool2d / torch.max).  C: power of two in [16, 512]. */
typedef struct {
  int dtype, B, H, W, C;
  const void* x; int x_stride;
  const void* mul; int mul_stride;          /* optional multiplicand (`out *= dense_k`, models/cdan.py:133,141,149) */
  void* out; int out_stride;                /* forward only */
  const float *w1, *b1, *w2, *b2;           /* ChannelGate.mlp: [C/16][C], [C/16], [C][C/16], [C] */
  const float* w7;                          /* SpatialGate conv [1][2][7][7] */
  const float *gamma, *beta;                /* SpatialGate BatchNorm2d(1) */
  float *running_mean, *running_var;        /* forward: updated in place; may be NULL */
  float momentum, eps;
  float* gate; int* amax_idx; float* pooled; float* comp; float* smap; float* bnc;   /* saved state */
  const void* dout; int dout_stride;        /* backward: gradient of out */
  void* dx; int dx_stride;                  /* backward outputs */
  void* dmul; int dmul_stride;              /* NULL when mul is NULL */
  float *dw1, *db1, *dw2, *db2, *dw7, *dgamma, *dbeta;
  void* workspace; size_t workspace_bytes;  /* >= mdie_cbam_train_workspace_bytes */
} mdie_cbam_train_desc;
size_t mdie_cbam_train_workspace_bytes(int B, int H, int W, int C);
int mdie_cbam_train_fwd(const mdie_cbam_train_desc* d, void* stream);
int mdie_cbam_train_bwd(const mdie_cbam_train_desc* d, void* stream);

/* ---------------------------------------------------------------------------------
 * Degradation classifier (router) pieces, classification/train_multilabel_classifier.py:117-131 -- a torchvision
 * ResNet18 backbone with two nn.Linear heads.  BasicBlock convolutions run on mdie_conv_fwd (BatchNorm folded, identity
 * branch as `residual` with act = NONE, then mdie_relu_inplace: the hot epilogue adds its residual AFTER the activation,
 * which is what the CDAN decoder needs); stride-2 convolutions as stride-1 + mdie_subsample2.
 * --------------------------------------------------------------------------------- */
/* relu(bn(conv7x7/s2/p3(normalise(x)))): x fp32 NCHW [B,3,H,W]; mean3/std3 HOST float[3] (NULL = no normalisation,
 * :760 uses the ImageNet constants); weight packed by mdie_pack_stem7_weight from [64][3][7][7]; out NHWC
 * [B, ceil(H/2), ceil(W/2), 64]. */
size_t mdie_stem7_weight_bytes(int dtype);
int mdie_pack_stem7_weight(int dtype, const float* w, void* dst);
int mdie_stem7_fwd(int dtype, int B, int H, int W, const float* x_nchw, const float* mean3, const float* std3,
                   const void* weight, const float* post_scale, const float* post_shift, void* out, int out_stride,
                   void* stream);
/* nn.MaxPool2d(3, stride 2, padding 1): [B,H,W,C] -> [B,ceil(H/2),ceil(W/2),C] */
int mdie_maxpool3x3s2(int dtype, int B, int H, int W, int C, const void* in, int in_stride, void* out, int out_stride,
                      void* stream);
/* x = max(x, 0) in place, NHWC [B,H,W,C] with pixel stride */
int mdie_relu_inplace(int dtype, long npix, int C, void* x, int stride, void* stream);
/* out[b,y,x,:] = in[b,2y,2x,:] */
int mdie_subsample2(int dtype, int B, int H, int W, int C, const void* in, int in_stride, void* out, int out_stride,
                    void* stream);
/* AdaptiveAvgPool2d(1) + head_cls / head_sev (weights [ncls][C] fp32) + sigmoid -> prob_cls, sev: [B][ncls];
 * feat (nullable): [B][C] pooled features */
int mdie_avgpool_heads(int dtype, int B, int H, int W, int C, const void* x, int stride, const float* w_cls,
                       const float* b_cls, const float* w_sev, const float* b_sev, int ncls, float* feat,
                       float* prob_cls, float* sev, void* stream);

/* Training loss, value and gradient in one call (utils/loss_factory.py:146-230; models/model.py:161-164 evaluates the
 * pipeline and calls backward on it every step).  Terms that need downloaded networks (vgg_perceptual, lpips) are
 * not here.  ssim is 1 - SSIM with the torchmetrics defaults restated as for mdie_psnr_ssim; the data range is
 * taken from the tensors and treated as a constant in the gradient.
 *   values: device float[nterms + 1]: each term unweighted, then sum_k weight_k * term_k
 *   grad:   device fp32 NCHW [B,3,H,W] = d values[nterms] / d pred, or NULL (values only)
 * terms: HOST array, each kind at most once. */
enum { MDIE_LOSS_MSE = 0          /* loss_factory.py:146-151 */,
       MDIE_LOSS_L1 = 1           /* :153-158 */,
       MDIE_LOSS_CHARBONNIER = 2  /* param = eps, :160-167 */,
       MDIE_LOSS_SSIM = 3         /* :180-189 */,
       MDIE_LOSS_GRADIENT_L1 = 4  /* param = to_gray (0/1), :90-103, 203-230 */ };
#define MDIE_LOSS_MAX_TERMS 5
typedef struct { int kind; float weight; float param; } mdie_loss_term;
size_t mdie_loss_workspace_bytes(int B, int H, int W);
int mdie_loss_fwd_bwd(int B, int H, int W, const float* pred, const float* target, const mdie_loss_term* terms,
                      int nterms, float* values, float* grad, void* workspace, size_t workspace_bytes, void* stream);

const char* mdie_last_error(void);
int mdie_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MDIE_H */
