// Whole-network plan: CDAN.forward (models/cdan.py:171-176) in eval mode as one host call that
// enqueues every kernel on the caller's stream (capturable into a hipGraph: no allocation, no
// synchronisation), plus the host-side packing of the reference checkpoint into the parameter blob.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "common.hpp"

namespace mdie {

// ---- error plumbing -------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
LaunchTimer*& current_timer() {
  static thread_local LaunchTimer* t = nullptr;
  return t;
}

// ---- architecture table -----------------------------------------------------------------------------------
// Everything below is derived from the layer widths of models/cdan.py:58-65,103-119.
constexpr float BN_EPS = 1e-5f;
constexpr int GROWTH = 16, NLAYERS = 4;

struct ConvSpec {
  std::string w_key, b_key, bn_pre, bn_post;
  int ks, transposed;
  int cin, cout;        // real channels
  int cin_st, cout_st;  // stored channels (multiples of 16)
  int split, gap;       // real channel c >= split is stored at c + gap
};
struct CbamSpec {
  std::string prefix;
  int C;
};

static int st16(int c) { return (c + 15) / 16 * 16; }

enum { CV_E1 = 0, CV_E2, CV_E3, CV_E4, CV_DENSE0 = 4 /* 4 blocks x 5 */, CV_D1 = 24, CV_D2, CV_D3, CV_D4, CV_COUNT };
enum { CB_BOTT = 0, CB_1, CB_2, CB_3, CB_COUNT };

struct Arch {
  std::vector<ConvSpec> conv;
  std::vector<CbamSpec> cbam;
  int base3;   // stored channels of the 3-channel input of decoder.final_dense: ONE 16-byte K group per pixel
               // (8 bf16 / 4 f32), not 16 -- the five layers of the block read it five times at 256x256
  explicit Arch(int dtype) {
    base3 = dtype_vec(dtype);
    conv.resize(CV_COUNT);
    const int enc[5] = {3, 64, 128, 256, 512};
    for (int i = 0; i < 4; ++i) {
      const std::string p = "encoder.conv" + std::to_string(i + 1);
      conv[CV_E1 + i] = {p + ".conv.weight", p + ".conv.bias", "", p + ".bn", 3, 0, enc[i], enc[i + 1], st16(enc[i]), st16(enc[i + 1]), enc[i], 0};
    }
    const char* dn[4] = {"encoder.dense1", "encoder.dense2", "encoder.dense3", "decoder.final_dense"};
    const int dc[4] = {64, 128, 256, 3};
    for (int b = 0; b < 4; ++b) {
      const int c0 = dc[b], c0s = b == 3 ? base3 : st16(c0), gap = c0s - c0;
      for (int l = 0; l <= NLAYERS; ++l) {
        const bool tr = l == NLAYERS;
        const std::string p = std::string(dn[b]) + (tr ? ".transition_layer" : ".layers." + std::to_string(l));
        const int cin = c0 + GROWTH * l, cout = tr ? c0 : GROWTH;
        conv[CV_DENSE0 + b * 5 + l] = {p + ".2.weight", p + ".2.bias", p + ".0", "", tr ? 1 : 3, 0, cin, cout, c0s + GROWTH * l, st16(cout), c0, gap};
      }
    }
    const int dec[5] = {512, 256, 128, 64, 3};
    for (int i = 0; i < 4; ++i) {
      const std::string p = "decoder.conv" + std::to_string(i + 1);
      conv[CV_D1 + i] = {p + ".weight", p + ".bias", "", "decoder.bn" + std::to_string(i + 1), 3, 1, dec[i], dec[i + 1], st16(dec[i]), st16(dec[i + 1]), dec[i], 0};
    }
    cbam = {{"bottleneck", 512}, {"decoder.cbam1", 256}, {"decoder.cbam2", 128}, {"decoder.cbam3", 64}};
  }
};
static const Arch& arch(int dtype) {
  static const Arch f32(MDIE_F32), b16(MDIE_BF16);   // the two 16-bit types share one table (same channel grouping)
  return dtype == MDIE_F32 ? f32 : b16;
}

// ---- parameter blob layout ----------------------------------------------------------------------------------
static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct ConvBlob { size_t w, post_scale, post_shift, pre_scale, pre_shift; };
struct CbamBlob { size_t w1, b1, w2, b2, w7, bn; };
struct BlobLayout {
  ConvBlob conv[CV_COUNT];
  CbamBlob cbam[CB_COUNT];
  size_t fl0_w;  // decoder.final_dense layer 0 once more, im2col-packed (k = tap*3 + c) for mdie_up_add_dense0_fwd
  size_t total;
};

static size_t conv_weight_bytes(int dtype, int ks, int cin_st, int cout_st) {
  const int kc = dtype_kc(dtype);
  return (size_t)cdiv(cin_st, kc) * ks * ks * cout_st * 64;
}

static size_t first_weight_bytes(int dtype, int cout_st) { (void)dtype; return (size_t)2 * cout_st * 64; }   // two MFMA steps for every type

static BlobLayout blob_layout(int dtype) {
  BlobLayout L{};
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += align256(bytes); return o; };
  const Arch& A = arch(dtype);
  for (int i = 0; i < CV_COUNT; ++i) {
    const ConvSpec& s = A.conv[i];
    L.conv[i].w = take(i == CV_E1 ? first_weight_bytes(dtype, s.cout_st) : conv_weight_bytes(dtype, s.ks, s.cin_st, s.cout_st));
    L.conv[i].post_scale = take(s.cout_st * sizeof(float));
    L.conv[i].post_shift = take(s.cout_st * sizeof(float));
    L.conv[i].pre_scale = take(s.cin_st * sizeof(float));
    L.conv[i].pre_shift = take(s.cin_st * sizeof(float));
  }
  for (int i = 0; i < CB_COUNT; ++i) {
    const int C = A.cbam[i].C, Hd = C / 16;
    L.cbam[i].w1 = take((size_t)Hd * C * sizeof(float));
    L.cbam[i].b1 = take(Hd * sizeof(float));
    L.cbam[i].w2 = take((size_t)C * Hd * sizeof(float));
    L.cbam[i].b2 = take(C * sizeof(float));
    L.cbam[i].w7 = take(98 * sizeof(float));
    L.cbam[i].bn = take(2 * sizeof(float));
  }
  L.fl0_w = take(first_weight_bytes(dtype, 16));
  L.total = off;
  return L;
}

// ---- host-side packing ----------------------------------------------------------------------------------------
static int pack_conv_weight(int dtype, int ks, int transposed, const float* w, int cout, int cin, int cout_st, int cin_st, int split,
                            int gap, void* dst) {
  const int kc = dtype_kc(dtype);
  const int ntap = ks * ks;
  const size_t esz = dtype_size(dtype);
  memset(dst, 0, conv_weight_bytes(dtype, ks, cin_st, cout_st));
  for (int o = 0; o < cout; ++o)
    for (int c = 0; c < cin; ++c) {
      const int cs = c + (c >= split ? gap : 0);
      const int chunk = cs / kc, k = cs % kc;
      for (int kh = 0; kh < ks; ++kh)
        for (int kw = 0; kw < ks; ++kw) {
          // ConvTranspose2d(k3,s1,p1) == Conv2d with W'[o][c][kh][kw] = W[c][o][k-1-kh][k-1-kw]
          const float v = transposed ? w[(((size_t)c * cout + o) * ks + (ks - 1 - kh)) * ks + (ks - 1 - kw)]
                                     : w[(((size_t)o * cin + c) * ks + kh) * ks + kw];
          const int vec = kc / 4, q = k / vec, i = k % vec;  // 16-byte quarter q of the chunk, element i in it
          const size_t idx = ((((size_t)chunk * 4 + q) * ntap + kh * ks + kw) * cout_st + o) * vec + i;
          if (esz == 4) reinterpret_cast<float*>(dst)[idx] = v;
          else reinterpret_cast<uint16_t*>(dst)[idx] = f32_to_half_bits(dtype, v);
        }
    }
  return MDIE_OK;
}


// encoder.conv1 weight [cout][3][3][3] -> [step][cout_st][64 B] with k = tap*3 + c
static void pack_first_weight(int dtype, const float* w, int cout, int cout_st, void* dst) {
  memset(dst, 0, first_weight_bytes(dtype, cout_st));
  for (int o = 0; o < cout; ++o)
    for (int k = 0; k < 27; ++k) {
      const int tap = k / 3, c = k % 3;
      const float v = w[((size_t)o * 3 + c) * 9 + tap];
      if (dtype == MDIE_F32) reinterpret_cast<float*>(dst)[((size_t)(k / 16) * cout_st + o) * 16 + k % 16] = v;
      else {   // 16-bit: k' = tap*4 + c (a patch pixel's 4 stored channels = 8 aligned bytes), step = k' / 32: taps 0..7 | tap 8
        const int kp = tap * 4 + c;
        reinterpret_cast<uint16_t*>(dst)[((size_t)(kp / 32) * cout_st + o) * 32 + kp % 32] = f32_to_half_bits(dtype, v);
      }
    }
}

struct TensorMap {
  std::unordered_map<std::string, const mdie_tensor*> m;
  const float* get(const std::string& k, int64_t numel) const {
    auto it = m.find(k);
    if (it == m.end() || !it->second->data) { set_error("mdie_cdan_pack_params: checkpoint entry '%s' missing", k.c_str()); return nullptr; }
    if (it->second->numel != numel) {
      set_error("mdie_cdan_pack_params: '%s' has %lld elements, expected %lld", k.c_str(), (long long)it->second->numel, (long long)numel);
      return nullptr;
    }
    return it->second->data;
  }
};

// eval-mode BatchNorm as y = x * s + t  (models/cdan.py:12; running statistics, eps 1e-5)
static bool bn_fold(const TensorMap& T, const std::string& p, int c, std::vector<float>& s, std::vector<float>& t) {
  const float *g = T.get(p + ".weight", c), *b = T.get(p + ".bias", c), *m = T.get(p + ".running_mean", c), *v = T.get(p + ".running_var", c);
  if (!g || !b || !m || !v) return false;
  s.resize(c); t.resize(c);
  for (int i = 0; i < c; ++i) {
    const double inv = 1.0 / sqrt((double)v[i] + (double)BN_EPS);
    s[i] = (float)(g[i] * inv);
    t[i] = (float)(b[i] - m[i] * g[i] * inv);
  }
  return true;
}

// ---- workspace plan ---------------------------------------------------------------------------------------------------
struct Buf { size_t off; int C; int st = 0; };  // NHWC, C stored channels, pixel stride st (0: == C)
struct Plan {
  Buf x16, o[3], g[3][4], d[3], e, bott, t1, u1, t2lo, t2, u2, t3lo, t3, u3, t4lo, t4, fg[4], out16;
  size_t cbam_ws, cbam_ws_bytes;
  size_t pool_ws;   // [B][<= 128 slabs][2][128] floats: pooled partials written by upsample+skip
  size_t tr_ws;     // [B][H][W][4] floats: partial sums of decoder.final_dense's transition (fused chain, forward_impl)
  size_t total;
};

static Plan make_plan(int dtype, int B, int H, int W) {
  Plan P{};
  size_t off = 0;
  const size_t esz = dtype_size(dtype);
  auto take = [&](int C, int h, int w) { Buf b{off, C}; off += align256((size_t)B * h * w * C * esz); return b; };
  const int h1 = H / 2, w1 = W / 2, h2 = H / 4, w2 = W / 4, h3 = H / 8, w3 = W / 8;
  const int hs[3] = {h1, h2, h3}, ws[3] = {w1, w2, w3}, cs[3] = {64, 128, 256};
  P.x16 = take(16, H, W);
  for (int i = 0; i < 3; ++i) {
    P.o[i] = take(cs[i], hs[i], ws[i]);
    for (int l = 0; l < 4; ++l) P.g[i][l] = take(16, hs[i], ws[i]);
    P.d[i] = take(cs[i], hs[i], ws[i]);
  }
  P.e = take(512, h3, w3);
  P.bott = take(512, h3, w3);
  P.t1 = take(256, h3, w3); P.u1 = take(256, h3, w3);
  P.t2lo = take(128, h3, w3); P.t2 = take(128, h2, w2); P.u2 = take(128, h2, w2);
  P.t3lo = take(64, h2, w2); P.t3 = take(64, h1, w1); P.u3 = take(64, h1, w1);
  P.t4lo = take(16, h1, w1); P.t4 = take(arch(dtype).base3, H, W);
  for (int l = 0; l < 4; ++l) P.fg[l] = take(16, H, W);
  P.out16 = take(16, H, W);
  size_t cb = 0;
  const int ch[4] = {512, 256, 128, 64}, chh[4] = {h3, h3, h2, h1}, cww[4] = {w3, w3, w2, w1};
  for (int i = 0; i < 4; ++i) { size_t b = mdie_cbam_workspace_bytes(B, chh[i], cww[i], ch[i]); cb = b > cb ? b : cb; }
  P.cbam_ws = off; P.cbam_ws_bytes = cb; off += align256(cb);
  P.pool_ws = off; {
    // pooling partials written by the producer of a CBAM input: upsample+skip (<= 128 slabs x 128 channels) or the
    // 512- / 256-channel convolutions in front of the bottleneck CBAM and cbam1 (one slab per conv tile, <= 256)
    size_t floats = (size_t)128 * 2 * 128;
    const int widths[2] = {512, 256};
    for (int i = 0; i < 2; ++i) {
      const int t = mdie_conv_tile(B, h3, w3, widths[i]);
      const size_t slabs = (size_t)cdiv(h3, t) * cdiv(w3, t);
      if (slabs <= MDIE_POOL_SLABS_MAX && slabs * 2 * widths[i] > floats) floats = slabs * 2 * widths[i];   // (more tiles: that CBAM pools on its own)
    }
    off += align256((size_t)B * floats * sizeof(float));
  }
  P.tr_ws = off; off += align256((size_t)B * H * W * 4 * sizeof(float));
  P.total = off;
  return P;
}

}  // namespace mdie

using namespace mdie;

extern "C" const char* mdie_last_error(void) { return g_err; }
extern "C" int mdie_abi_version(void) { return MDIE_ABI_VERSION; }

extern "C" size_t mdie_conv_weight_bytes(int dtype, int ksize, int cin_stored, int cout_stored) {
  return conv_weight_bytes(dtype, ksize, cin_stored, cout_stored);
}

extern "C" int mdie_pack_conv_weight(int dtype, int ksize, int transposed, const float* w, int cout, int cin, int cout_stored,
                                     int cin_stored, int split, int gap, void* dst) {
  MDIE_REQUIRE(dtype_valid(dtype), "mdie_pack_conv_weight: bad dtype %d", dtype);
  MDIE_REQUIRE(ksize == 1 || ksize == 3, "mdie_pack_conv_weight: ksize %d", ksize);
  MDIE_REQUIRE(w && dst && cout > 0 && cin > 0, "mdie_pack_conv_weight: null/empty");
  MDIE_REQUIRE(cout_stored >= cout && cout_stored % 16 == 0, "mdie_pack_conv_weight: cout_stored %d", cout_stored);
  MDIE_REQUIRE(cin_stored % dtype_vec(dtype) == 0 && cin_stored >= cin + (split < cin ? gap : 0) && gap >= 0 && split >= 0,
               "mdie_pack_conv_weight: cin_stored %d invalid / too small for cin %d split %d gap %d", cin_stored, cin, split, gap);
  return pack_conv_weight(dtype, ksize, transposed, w, cout, cin, cout_stored, cin_stored, split, gap, dst);
}

extern "C" size_t mdie_conv_first_weight_bytes(int dtype, int cout_stored) {
  if ((!dtype_valid(dtype)) || cout_stored <= 0) return 0;
  return first_weight_bytes(dtype, cout_stored);
}

extern "C" int mdie_pack_conv_first_weight(int dtype, const float* w, int cout, int cout_stored, void* dst) {
  MDIE_REQUIRE(dtype_valid(dtype), "mdie_pack_conv_first_weight: bad dtype %d", dtype);
  MDIE_REQUIRE(w && dst && cout > 0 && cout_stored >= cout && cout_stored % 16 == 0, "mdie_pack_conv_first_weight: bad argument");
  pack_first_weight(dtype, w, cout, cout_stored, dst);
  return MDIE_OK;
}

extern "C" size_t mdie_cdan_param_bytes(int dtype) {
  if (!dtype_valid(dtype)) return 0;
  return blob_layout(dtype).total;
}

extern "C" int mdie_cdan_pack_params(int dtype, const mdie_tensor* tensors, int n, void* blob_host, size_t blob_bytes) {
  MDIE_REQUIRE(dtype_valid(dtype), "mdie_cdan_pack_params: bad dtype %d", dtype);
  MDIE_REQUIRE(tensors && n > 0 && blob_host, "mdie_cdan_pack_params: null argument");
  const BlobLayout L = blob_layout(dtype);
  if (blob_bytes < L.total) { set_error("mdie_cdan_pack_params: blob %zu < %zu bytes", blob_bytes, L.total); return MDIE_ENOSPC; }
  TensorMap T;
  for (int i = 0; i < n; ++i)
    if (tensors[i].name) T.m[tensors[i].name] = &tensors[i];
  char* blob = reinterpret_cast<char*>(blob_host);
  memset(blob, 0, L.total);
  const Arch& A = arch(dtype);
  for (int i = 0; i < CV_COUNT; ++i) {
    const ConvSpec& s = A.conv[i];
    const float* w = T.get(s.w_key, (int64_t)s.cin * s.cout * s.ks * s.ks);
    const float* b = T.get(s.b_key, s.cout);
    if (!w || !b) return MDIE_ENOENT;
    if (i == CV_E1) pack_first_weight(dtype, w, s.cout, s.cout_st, blob + L.conv[i].w);
    else if (i == CV_DENSE0 + 3 * 5) { pack_first_weight(dtype, w, s.cout, s.cout_st, blob + L.fl0_w); pack_conv_weight(dtype, s.ks, s.transposed, w, s.cout, s.cin, s.cout_st, s.cin_st, s.split, s.gap, blob + L.conv[i].w); }
    else pack_conv_weight(dtype, s.ks, s.transposed, w, s.cout, s.cin, s.cout_st, s.cin_st, s.split, s.gap, blob + L.conv[i].w);
    float* ps = reinterpret_cast<float*>(blob + L.conv[i].post_scale);
    float* pt = reinterpret_cast<float*>(blob + L.conv[i].post_shift);
    if (!s.bn_post.empty()) {
      std::vector<float> sc, sh;
      if (!bn_fold(T, s.bn_post, s.cout, sc, sh)) return MDIE_ENOENT;
      for (int c = 0; c < s.cout; ++c) { ps[c] = sc[c]; pt[c] = (float)((double)b[c] * sc[c] + sh[c]); }
    } else {
      for (int c = 0; c < s.cout; ++c) { ps[c] = 1.0f; pt[c] = b[c]; }
    }
    if (!s.bn_pre.empty()) {
      std::vector<float> sc, sh;
      if (!bn_fold(T, s.bn_pre, s.cin, sc, sh)) return MDIE_ENOENT;
      float* qs = reinterpret_cast<float*>(blob + L.conv[i].pre_scale);
      float* qt = reinterpret_cast<float*>(blob + L.conv[i].pre_shift);
      for (int c = 0; c < s.cin; ++c) {
        const int cs = c + (c >= s.split ? s.gap : 0);
        qs[cs] = sc[c]; qt[cs] = sh[c];
      }
    }
  }
  for (int i = 0; i < CB_COUNT; ++i) {
    const std::string& p = A.cbam[i].prefix;
    const int C = A.cbam[i].C, Hd = C / 16;
    const float* w1 = T.get(p + ".ChannelGate.mlp.1.weight", (int64_t)Hd * C);
    const float* b1 = T.get(p + ".ChannelGate.mlp.1.bias", Hd);
    const float* w2 = T.get(p + ".ChannelGate.mlp.3.weight", (int64_t)C * Hd);
    const float* b2 = T.get(p + ".ChannelGate.mlp.3.bias", C);
    const float* w7 = T.get(p + ".SpatialGate.spatial.conv.weight", 98);
    std::vector<float> sc, sh;
    if (!w1 || !b1 || !w2 || !b2 || !w7 || !bn_fold(T, p + ".SpatialGate.spatial.bn", 1, sc, sh)) return MDIE_ENOENT;
    memcpy(blob + L.cbam[i].w1, w1, (size_t)Hd * C * 4);
    memcpy(blob + L.cbam[i].b1, b1, Hd * 4);
    memcpy(blob + L.cbam[i].w2, w2, (size_t)C * Hd * 4);
    memcpy(blob + L.cbam[i].b2, b2, C * 4);
    memcpy(blob + L.cbam[i].w7, w7, 98 * 4);
    float bn[2] = {sc[0], sh[0]};
    memcpy(blob + L.cbam[i].bn, bn, 8);
  }
  return MDIE_OK;
}

extern "C" size_t mdie_cdan_workspace_bytes(int dtype, int B, int H, int W) {
  if ((!dtype_valid(dtype)) || B <= 0 || H <= 0 || W <= 0 || H % 8 || W % 8) return 0;
  return make_plan(dtype, B, H, W).total;
}

extern "C" double mdie_cdan_flops(int B, int H, int W) {
  // 2*MAC over the 35 conv / 16 linear applications (hook-traced law of SURVEY.md section 6)
  return (double)B * (252837.375 * (double)H * (double)W + 174080.0);
}

extern "C" double mdie_cdan_algorithmic_bytes(int B, int H, int W, int esize) {
  // Fused-schedule model of SURVEY.md 8(d): each conv reads its input and writes its output once
  // (BN/ReLU/bias/sigmoid/maxpool free, torch.cat free); a CBAM reads its tensor twice, writes it
  // once, reads the multiplicand and round-trips the 2-channel map; upsample+add reads low-res and
  // skip and writes the sum.  Weights (3.53 M elements) are counted once per batch.
  const double P = (double)H * W;
  double el = 0;
  auto conv = [&](double cin, double cout, double pin, double pout) { el += cin * pin + cout * pout; };
  auto dense = [&](double c, double p, double cout) {
    for (int i = 0; i < 4; ++i) el += (c + 16 * i) * p + 16 * p;
    el += (c + 64) * p + cout * p;
  };
  auto cbam = [&](double c, double p, bool mul) { el += 3 * c * p + (mul ? c * p : 0) + 4 * p; };
  auto up = [&](double c, double plo) { el += c * plo + 8 * c * plo; };
  conv(3, 64, P, P / 4); dense(64, P / 4, 64);
  conv(64, 128, P / 4, P / 16); dense(128, P / 16, 128);
  conv(128, 256, P / 16, P / 64); dense(256, P / 64, 256);
  conv(256, 512, P / 64, P / 64);
  cbam(512, P / 64, false);
  conv(512, 256, P / 64, P / 64); el += 256 * P / 64; cbam(256, P / 64, true);
  conv(256, 128, P / 64, P / 64); up(128, P / 64); cbam(128, P / 16, true);
  conv(128, 64, P / 16, P / 16); up(64, P / 16); cbam(64, P / 4, true);
  conv(64, 3, P / 4, P / 4); up(3, P / 4);
  dense(3, P, 3);
  const double weights = 3585663.0 - 2.0 * 0;  // parameters, read once per batch
  return ((double)B * el + weights) * esize;
}

// ---- the forward plan ---------------------------------------------------------------------------------------------------
namespace {

struct Aux {
  hipStream_t side[3];
  hipEvent_t fork[3], join[3];
};

// Instrumented mode only (mdie_cdan_fwd_desc.launch_info): which layer a launch belongs to and its share of the fused-schedule
// model of SURVEY.md 8d -- elements per image as mdie_cdan_algorithmic_bytes counts them, plus the parameters the launch reads
// (the model's "weights once per batch" term, 3 585 663 elements, split over the launches that read them).  A call's launches
// are [from, timer->n); the share goes to the first of them unless a caller splits it (CBAM stages).
struct Notes {
  LaunchTimer* lt = nullptr;
  double esz = 2, Bn = 1;
  int dtype = MDIE_BF16;
  int mark() const { return lt ? lt->n : 0; }
  bool on() const { return lt && lt->info; }
  void put(int i, const char* label, double elems, double params, double flops) const {
    if (!on() || i >= lt->n || i >= lt->cap) return;
    mdie_launch_info& li = lt->info[i];
    snprintf(li.label, sizeof li.label, "%s", label);
    li.alg_bytes = (elems * Bn + params) * esz;
    li.flops = flops * Bn;
  }
  void add(int i, const char* label, double elems, double params, double flops) const {     // a second share for a launch that already has one
    if (!on() || i < 0 || i >= lt->n || i >= lt->cap) return;
    mdie_launch_info& li = lt->info[i];
    snprintf(li.label, sizeof li.label, "%s", label);
    li.alg_bytes += (elems * Bn + params) * esz;
    li.flops += flops * Bn;
  }
  void note(int from, const char* label, double elems, double params, double flops) const {
    if (!on()) return;
    for (int i = from; i < lt->n; ++i) put(i, label, i == from ? elems : 0.0, i == from ? params : 0.0, i == from ? flops : 0.0);
  }
  // one convolution: reads cin x pin, writes cout x pout (+ extra elements: a residual read), real channel counts
  void conv(int from, const char* label, int id, double pin, double pout, double extra = 0.0) const {
    if (!on()) return;
    const ConvSpec& s = arch(dtype).conv[id];
    const double params = (double)s.cin * s.cout * s.ks * s.ks + s.cout + 2.0 * (!s.bn_post.empty() ? s.cout : !s.bn_pre.empty() ? s.cin : 0);
    note(from, label, s.cin * pin + s.cout * pout + extra, params, 2.0 * s.cin * s.cout * s.ks * s.ks * pin);
  }
  // a CBAM stage: 2 reads + 1 write of the tensor, the multiplicand, the 2-channel map both ways; the global pool rides on the
  // producer (a standalone pool launch is traffic beyond the model: 0 bytes)
  void cbam(int from, const char* name, int C, double p, bool mul, const char* mulname) const {
    if (!on()) return;
    const double mlp = 2.0 * C * (C / 16) + C / 16 + C;
    bool gate_seen = false;
    for (int i = from; i < lt->n && i < lt->cap; ++i) {
      char label[40];
      switch (lt->kind[i]) {
        case MDIE_K_CBAM_POOL: snprintf(label, sizeof label, "%s.pool", name); put(i, label, 0, 0, 0); break;
        case MDIE_K_CBAM_GATE: gate_seen = true; snprintf(label, sizeof label, "%s.gate", name); put(i, label, 0, mlp, 8.0 * C * (C / 16)); break;
        case MDIE_K_CBAM_CHANPOOL:
          snprintf(label, sizeof label, gate_seen ? "%s.chanpool" : "%s.gate+chanpool", name);
          put(i, label, C * p + 2 * p, gate_seen ? 0 : mlp, gate_seen ? 0 : 8.0 * C * (C / 16));
          break;
        default:
          snprintf(label, sizeof label, "%s.spatial%s%s", name, mul ? "*" : "", mul ? mulname : "");
          put(i, label, (mul ? 3.0 : 2.0) * C * p + 2 * p, 100, 2.0 * 98 * p);
      }
    }
  }
};

struct Ctx {
  int dtype, B;
  const char* params;
  char* ws;
  BlobLayout L;
  hipStream_t stream;
  size_t esz;
  const Notes* notes = nullptr;
  const long long* delta = nullptr;   // several weight sets in one launch chain (mdie_cdan_fwd_desc.blob_delta)
};

static mdie_seg seg(const Ctx& c, const Buf& b) { return mdie_seg{c.ws + b.off, b.C, b.st ? b.st : b.C}; }

// Diagnostic build only (-DEXP_ABLATE, tools/ablate.sh): MDIE_ABLATE = comma-separated label prefixes whose launches are LEFT OUT
// (results are garbage; the step time without a stage is what the stage costs the step, as opposed to its serial time).
#ifdef EXP_ABLATE
static bool ablated(const char* label) {
  const char* e = getenv("MDIE_ABLATE");
  if (!e || !label) return false;
  const size_t n = strlen(label);
  for (const char* p = e; *p;) {
    const char* q = strchr(p, ',');
    const size_t k = q ? (size_t)(q - p) : strlen(p);
    if (k > 0 && k <= n && strncmp(p, label, k) == 0) return true;
    if (!q) break;
    p = q + 1;
  }
  return false;
}
#else
static inline bool ablated(const char*) { return false; }
#endif

#ifdef EXP_SCHED   // schedule-exploration builds only (tools/sched_sweep.py): per-layer kernel choice and fork points from the environment
static bool exp_listed(const char* var, const char* label) {
  const char* v = getenv(var);
  if (!v || !label) return false;
  const size_t n = strlen(label);
  for (const char* p = v; (p = strstr(p, label)); p += n)
    if ((p == v || p[-1] == ',') && (p[n] == 0 || p[n] == ',')) return true;
  return false;
}
static int exp_fork_pos(int block, int dflt) {      // MDIE_EXP_FORK = "p1,p2,p3": where dense1 / dense2 / dense3 start
  const char* v = getenv("MDIE_EXP_FORK");
  if (!v) return dflt;
  int p[3] = {-1, -1, -1};
  sscanf(v, "%d,%d,%d", &p[0], &p[1], &p[2]);
  return p[block] >= 0 ? p[block] : dflt;
}
#endif
static int run_conv(const Ctx& c, const char* label, int id, int H, int W, std::initializer_list<Buf> in, const Buf& out, int act, int pool,
                    const Buf* residual, float* out_nchw3 = nullptr, float* pool_partial = nullptr, const mdie_tr_fuse* tr = nullptr, int share_cu = 0) {
  const ConvSpec& s = arch(c.dtype).conv[id];
  if (ablated(label)) return MDIE_OK;
  const int from = c.notes ? c.notes->mark() : 0;
  mdie_conv_desc d{};
  d.dtype = c.dtype; d.B = c.B; d.H = H; d.W = W; d.ksize = s.ks;
  d.nseg = 0; d.cin = 0;
  for (const Buf& b : in) { d.in[d.nseg++] = seg(c, b); d.cin += b.C; }
  d.cout = s.cout_st;
  if (d.cin != s.cin_st) { set_error("plan: conv %d fed %d channels, expects %d", id, d.cin, s.cin_st); return MDIE_EINVAL; }
  if (!s.bn_pre.empty()) {
    d.pre_scale = reinterpret_cast<const float*>(c.params + c.L.conv[id].pre_scale);
    d.pre_shift = reinterpret_cast<const float*>(c.params + c.L.conv[id].pre_shift);
  }
  d.weight = c.params + c.L.conv[id].w;
  d.post_scale = reinterpret_cast<const float*>(c.params + c.L.conv[id].post_scale);
  d.post_shift = reinterpret_cast<const float*>(c.params + c.L.conv[id].post_shift);
  d.act = act; d.pool = pool;
  if (residual) { d.residual = c.ws + residual->off; d.res_stride = residual->C; }
  d.out = c.ws + out.off; d.out_stride = out.C;
  d.out_nchw3 = out_nchw3;
  d.pool_partial = pool_partial;
  d.tr = tr;
  d.blob_delta = c.delta;
  d.share_cu = share_cu;
#ifdef EXP_SCHED
  if (getenv("MDIE_EXP_NOWIDE")) d.share_cu = label && exp_listed("MDIE_EXP_NOWIDE", label);
#endif
  const int rc = mdie_conv_fwd(&d, c.stream);
  if (c.notes && label && !tr) {
    const double pin = (double)H * W;
    c.notes->conv(from, label, id, pin, pool ? pin / 4 : pin, residual ? (double)s.cout * pin : 0.0);
  }
  return rc;
}

static int run_dense(const Ctx& c, int block, int H, int W, const Buf& base, const Buf* g, const Buf& out, int act,
                     float* out_nchw3 = nullptr, bool have_g0 = false) {
  const int id0 = CV_DENSE0 + block * 5;
  static const char* const names[4][5] = {{"dense1.l0", "dense1.l1", "dense1.l2", "dense1.l3", "dense1.tr"},
                                          {"dense2.l0", "dense2.l1", "dense2.l2", "dense2.l3", "dense2.tr"},
                                          {"dense3.l0", "dense3.l1", "dense3.l2", "dense3.l3", "dense3.tr"},
                                          {"final.l0", "final.l1", "final.l2", "final.l3", "final.tr+sigmoid->nchw"}};
  const char* const* nm = names[block];
  int e;
  if (!have_g0 && (e = run_conv(c, nm[0], id0 + 0, H, W, {base}, g[0], MDIE_ACT_NONE, 0, nullptr))) return e;
  if ((e = run_conv(c, nm[1], id0 + 1, H, W, {base, g[0]}, g[1], MDIE_ACT_NONE, 0, nullptr))) return e;
  if ((e = run_conv(c, nm[2], id0 + 2, H, W, {base, g[0], g[1]}, g[2], MDIE_ACT_NONE, 0, nullptr))) return e;
  if ((e = run_conv(c, nm[3], id0 + 3, H, W, {base, g[0], g[1], g[2]}, g[3], MDIE_ACT_NONE, 0, nullptr))) return e;
  return run_conv(c, nm[4], id0 + 4, H, W, {base, g[0], g[1], g[2], g[3]}, out, act, 0, nullptr, out_nchw3);
}

// pooled_slabs > 0: the producer of x already wrote that many pooling partials per image into the plan's pool buffer
static int run_cbam_stage(const Ctx& c, const Plan& P, int id, int H, int W, const Buf& x, const Buf* mul, const Buf& out,
                          int pooled_slabs = 0, cbam_hook_fn before_last = nullptr, void* hook_ctx = nullptr) {
  const CbamBlob& o = c.L.cbam[id];
  { static const char* const nm[4] = {"bott", "cbam1", "cbam2", "cbam3"}; if (id >= 0 && id < 4 && ablated(nm[id])) return MDIE_OK; }
  const int from = c.notes ? c.notes->mark() : 0;
  mdie_cbam_desc d{};
  d.dtype = c.dtype; d.B = c.B; d.H = H; d.W = W; d.C = arch(c.dtype).cbam[id].C;
  d.x = c.ws + x.off; d.x_stride = x.C;
  d.w1 = reinterpret_cast<const float*>(c.params + o.w1); d.b1 = reinterpret_cast<const float*>(c.params + o.b1);
  d.w2 = reinterpret_cast<const float*>(c.params + o.w2); d.b2 = reinterpret_cast<const float*>(c.params + o.b2);
  d.w7 = reinterpret_cast<const float*>(c.params + o.w7);
  d.bn = reinterpret_cast<const float*>(c.params + o.bn);
  if (mul) { d.mul = c.ws + mul->off; d.mul_stride = mul->C; }
  d.out = c.ws + out.off; d.out_stride = out.C;
  d.workspace = c.ws + P.cbam_ws; d.workspace_bytes = P.cbam_ws_bytes;
  if (pooled_slabs > 0) { d.pool_partial = reinterpret_cast<const float*>(c.ws + P.pool_ws); d.pool_slabs = pooled_slabs; }
  d.blob_delta = c.delta;
  const int rc = before_last ? cbam_fwd_hooked(&d, c.stream, before_last, hook_ctx) : mdie_cbam_fwd(&d, c.stream);
  if (c.notes) {
    static const char* const names[4] = {"bott", "cbam1", "cbam2", "cbam3"}, * const muls[4] = {"", "d3", "d2", "d1"};
    c.notes->cbam(from, names[id], d.C, (double)H * W, mul != nullptr, muls[id]);
  }
  return rc;
}

static int run_up(const Ctx& c, const Plan& P, const char* label, int H, int W, const Buf& lo, const Buf& skip, const Buf& out) {
  // the upsampled + skip tensor feeds a CBAM: reduce it for the channel gate while writing it
  if (ablated(label)) return MDIE_OK;
  const int from = c.notes ? c.notes->mark() : 0;
  const int rc = mdie_upsample2x_add_pool(c.dtype, c.B, H, W, lo.C, c.ws + lo.off, lo.C, c.ws + skip.off, skip.C, c.ws + out.off, out.C,
                                          reinterpret_cast<float*>(c.ws + P.pool_ws), mdie_pool_slabs(2 * H, 2 * W), c.stream);
  if (c.notes) c.notes->note(from, label, 9.0 * lo.C * H * W, 0, 0);      // read lo, read skip (4x), write the sum (4x)
  return rc;
}

// Concurrency of the encoder DenseBlocks with the main path, in the two forms a caller's stream can be in:
//   STREAMS  eager launches: the block is enqueued on a side stream of the caller's mdie_aux (event fork / event join);
//   GRAPH    `stream` is being captured: NO other stream joins the capture.  The block's launches are captured on `stream`
//            itself, then the stream's capture dependency set is put back to the fork point
//            (hipStreamUpdateCaptureDependencies, SET), so what follows is a parallel branch of the graph; join() adds the
//            block's last node to the dependency set again (ADD).  This is legal on any capturing stream, origin or forked,
//            and leaves no foreign stream in capture state: round 1 forked its own non-blocking side streams into the
//            caller's capture, and a capture that reached them through a stream which was itself a fork took the process
//            down in hipStreamEndCapture once engines (and their side streams) of an earlier capture had been destroyed
//            (tools/capture_probe.hip reproduces the shapes; profiles/r02b_capture_probe.txt).
// Every fork is joined exactly once, also when a launch in between fails (the destructor joins what is still open), so an
// error return never leaves a side stream running ahead of -- or a graph branch dangling from -- the caller's stream.
struct Branches {
  enum Mode { SERIAL, STREAMS, GRAPH } mode = SERIAL;
  hipStream_t stream = nullptr;
  Aux* aux = nullptr;
  bool open[3] = {false, false, false};
  std::vector<hipGraphNode_t> at_fork[3], tail[3];

  static int deps_of(hipStream_t s, std::vector<hipGraphNode_t>& out) {
    hipStreamCaptureStatus st; unsigned long long id; hipGraph_t g; const hipGraphNode_t* deps = nullptr; size_t n = 0;
    if (hipStreamGetCaptureInfo_v2(s, &st, &id, &g, &deps, &n) != hipSuccess || st != hipStreamCaptureStatusActive) {
      (void)hipGetLastError();
      set_error("mdie_cdan_forward: hipStreamGetCaptureInfo_v2 failed on a capturing stream");
      return MDIE_ELAUNCH;
    }
    out.assign(deps, deps + n);
    return MDIE_OK;
  }
  // start branch k; *bs = the stream its launches go to
  int fork(int k, hipStream_t* bs) {
    *bs = stream;
    if (mode == STREAMS) {
      if (hipEventRecord(aux->fork[k], stream) != hipSuccess || hipStreamWaitEvent(aux->side[k], aux->fork[k], 0) != hipSuccess) {
        (void)hipGetLastError();
        set_error("mdie_cdan_forward: fork to side stream %d failed", k);
        return MDIE_ELAUNCH;
      }
      *bs = aux->side[k];
      open[k] = true;
    } else if (mode == GRAPH) {
      if (int rc = deps_of(stream, at_fork[k])) return rc;
      if (at_fork[k].empty()) return MDIE_OK;   // nothing captured before the block: it IS the head of the stream, stay serial
      open[k] = true;
    }
    return MDIE_OK;
  }
  // the branch's launches are enqueued: remember its end, give the main path its fork point back
  int end_branch(int k) {
    if (!open[k]) return MDIE_OK;
    if (mode == STREAMS) {
      if (hipEventRecord(aux->join[k], aux->side[k]) != hipSuccess) { (void)hipGetLastError(); set_error("mdie_cdan_forward: join record %d failed", k); return MDIE_ELAUNCH; }
    } else if (mode == GRAPH) {
      if (int rc = deps_of(stream, tail[k])) return rc;
      if (hipStreamUpdateCaptureDependencies(stream, at_fork[k].data(), at_fork[k].size(), hipStreamSetCaptureDependencies) != hipSuccess) {
        (void)hipGetLastError();
        open[k] = false;                         // the block stays in line with the main path: correct, merely serial
        set_error("mdie_cdan_forward: hipStreamUpdateCaptureDependencies(SET) failed");
        return MDIE_ELAUNCH;
      }
    }
    return MDIE_OK;
  }
  int join(int k) {
    if (!open[k]) return MDIE_OK;
    open[k] = false;
    if (mode == STREAMS) {
      if (hipStreamWaitEvent(stream, aux->join[k], 0) != hipSuccess) { (void)hipGetLastError(); set_error("mdie_cdan_forward: join wait %d failed", k); return MDIE_ELAUNCH; }
    } else if (mode == GRAPH && !tail[k].empty()) {
      if (hipStreamUpdateCaptureDependencies(stream, tail[k].data(), tail[k].size(), hipStreamAddCaptureDependencies) != hipSuccess) {
        (void)hipGetLastError();
        set_error("mdie_cdan_forward: hipStreamUpdateCaptureDependencies(ADD) failed");
        return MDIE_ELAUNCH;
      }
    }
    return MDIE_OK;
  }
  ~Branches() {
    for (int k = 0; k < 3; ++k)
      if (open[k]) {
        if (mode == STREAMS) (void)hipEventRecord(aux->join[k], aux->side[k]);   // (idempotent when end_branch already did)
        (void)join(k);
      }
  }
};

static int forward_impl(const mdie_cdan_fwd_desc* d, hipStream_t stream) {
  const int B = d->B, H = d->H, W = d->W;
  const Plan P = make_plan(d->dtype, B, H, W);
  if (d->workspace_bytes < P.total) { set_error("mdie_cdan_forward: workspace %zu < %zu bytes", d->workspace_bytes, P.total); return MDIE_ENOSPC; }
  Ctx c{d->dtype, B, reinterpret_cast<const char*>(d->params), reinterpret_cast<char*>(d->workspace), blob_layout(d->dtype), stream,
        dtype_size(d->dtype)};
  Notes notes;
  notes.lt = current_timer(); notes.esz = (double)c.esz; notes.Bn = (double)B; notes.dtype = d->dtype;
  if (notes.on()) c.notes = &notes;
  c.delta = d->blob_delta;
  const double PX = (double)H * W;
  const int h1 = H / 2, w1 = W / 2, h2 = H / 4, w2 = W / 4, h3 = H / 8, w3 = W / 8;
  int e;
#define RUN(call) do { if ((e = (call))) return e; } while (0)
  // The three encoder DenseBlocks only meet the main path again in the decoder (models/cdan.py:133,141,149), so each runs
  // beside it (Branches, above): side streams when launching eagerly, explicit graph branches when `stream` is capturing.
  Branches br;
  br.stream = stream;
  if (d->launch_ms == nullptr && !(d->flags & MDIE_FWD_SERIAL)) {   // instrumented mode stays serial
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &st) != hipSuccess) { (void)hipGetLastError(); st = hipStreamCaptureStatusNone; }
    if (st == hipStreamCaptureStatusActive) br.mode = Branches::GRAPH;   // never a library stream inside a caller's capture
    else if (st == hipStreamCaptureStatusNone && d->aux) { br.mode = Branches::STREAMS; br.aux = reinterpret_cast<Aux*>(d->aux); }
    else if (st != hipStreamCaptureStatusNone) { set_error("mdie_cdan_forward: the stream's capture has been invalidated"); return MDIE_EINVAL; }
  }
  auto side_dense = [&](int k, int hh, int ww) -> int {
    hipStream_t bs = stream;
    if (int rc = br.fork(k, &bs)) return rc;
    Ctx cs = c;
    cs.stream = bs;
    const int rc = run_dense(cs, k, hh, ww, P.o[k], P.g[k], P.d[k], MDIE_ACT_NONE);
    const int rc2 = br.end_branch(k);     // also after a failed launch: the branch must stay joinable
    return rc ? rc : rc2;
  };
  auto join_dense = [&](int k) -> int { return br.join(k); };
  // Encoder.forward, models/cdan.py:70-98 (dropout = identity in eval)
  {
    mdie_conv_first_desc f{};  // encoder.conv1 + BN + ReLU + maxpool straight from the fp32 NCHW input
    f.dtype = d->dtype; f.B = B; f.H = H; f.W = W; f.x = d->x;
    f.weight = c.params + c.L.conv[CV_E1].w;
    f.post_scale = reinterpret_cast<const float*>(c.params + c.L.conv[CV_E1].post_scale);
    f.post_shift = reinterpret_cast<const float*>(c.params + c.L.conv[CV_E1].post_shift);
    f.cout = 64; f.act = MDIE_ACT_RELU; f.pool = 1;
    f.out = c.ws + P.o[0].off; f.out_stride = P.o[0].C;
    f.blob_delta = c.delta;
    const int from = notes.mark();
    if (!ablated("enc.conv1")) RUN(mdie_conv_first_fwd(&f, stream));
    notes.conv(from, "enc.conv1+pool", CV_E1, PX, PX / 4);
  }
  // WHERE the side branches start (round 3, same-box sweep over fork points, eager launches, B = 32 at 256x256): forking each block
  // as soon as its input exists (dense1 after conv1, dense2 after conv2, dense3 after conv3) puts all three next to conv2 / conv3 /
  // conv4 -- the MFMA-bound layers of the main chain ran 1.3-2x their serial time with the HBM-bound dense kernels taking CUs and LDS
  // from them (profiles/r03y_infer_timeline_graph.txt), while the decoder's first half (gates, small maps) left the chip idle.
  // dense3 is needed first (cbam1) and stays where it was; dense1 and dense2 start after conv4 and run beside the bottleneck CBAM,
  // dec.conv1 and the decoder's small kernels: 30.2 k -> 31.0 k images/s (+2.6 %; all three after conv4 +2.3 %, only dense1 late
  // +2.2 %, all after the bottleneck +1 %, all three in line on ONE side stream -2 %; stream priorities: nothing).
#ifdef EXP_SCHED
  // fork positions: 0 after conv1, 1 after conv2, 2 after conv3, 3 after conv4, 4 after the bottleneck, 5 after dec.conv1 (dense1 / dense2 only),
  // 6 after cbam1, 7 after dec.conv2, 8 after up2 (dense1 only), 9 after cbam2, 10 after dec.conv3
  const int fpos[3] = {exp_fork_pos(0, 3), exp_fork_pos(1, 3), exp_fork_pos(2, 2)};
  const int fhh[3] = {h1, h2, h3}, fww[3] = {w1, w2, w3};
  auto fork_here = [&](int pos) -> int {
    for (int k = 2; k >= 0; --k) if (fpos[k] == pos) if (int rc = side_dense(k, fhh[k], fww[k])) return rc;
    return MDIE_OK;
  };
#define FORK_AT(p) RUN(fork_here(p))
#else
#define FORK_AT(p)
#endif
  FORK_AT(0);
  RUN(run_conv(c, "enc.conv2+pool", CV_E2, h1, w1, {P.o[0]}, P.o[1], MDIE_ACT_RELU, 1, nullptr));
  FORK_AT(1);
  RUN(run_conv(c, "enc.conv3+pool", CV_E3, h2, w2, {P.o[1]}, P.o[2], MDIE_ACT_RELU, 1, nullptr));
#ifdef EXP_SCHED
  FORK_AT(2);
#else
  RUN(side_dense(2, h3, w3));
#endif
  // the two CBAMs at the deep end pool tensors a 64-wide convolution has just written: that convolution emits the
  // per-tile channel sums / maxima itself (one slab per tile), unless the picture is so large that a gate would
  // have to fold more than MDIE_POOL_SLABS_MAX of them
  float* pool_buf = reinterpret_cast<float*>(c.ws + P.pool_ws);
  const int te = mdie_conv_tile(B, h3, w3, 512), td = mdie_conv_tile(B, h3, w3, 256);
  const int slabs_e = cdiv(h3, te) * cdiv(w3, te), slabs_d = cdiv(h3, td) * cdiv(w3, td);
  const bool fuse_e = slabs_e <= MDIE_POOL_SLABS_MAX, fuse_d = slabs_d <= MDIE_POOL_SLABS_MAX;
  // (MDIE_FWD_SHARE_CU_CONV4: the one layer whose kernel choice moves the step -- dense3 runs beside it, dense1 / dense2 start behind it)
  RUN(run_conv(c, fuse_e ? "enc.conv4+pool-stats" : "enc.conv4", CV_E4, h3, w3, {P.o[2]}, P.e, MDIE_ACT_RELU, 0, nullptr, nullptr, fuse_e ? pool_buf : nullptr, nullptr,
               (d->flags & MDIE_FWD_YIELD_CU_CONV4) ? 2 : (d->flags & MDIE_FWD_SHARE_CU_CONV4) ? 1 : 0));
#ifdef EXP_SCHED
  FORK_AT(3);
#else
  // dense2 starts here; dense1 -- the largest block and the one needed last (cbam3) -- here too, or behind dec.conv1 (MDIE_FWD_LATE_DENSE1: 1-4 us
  // less for the step with the two-run form of conv4 on every box where it was swept; the host times it with the other forms)
  const bool late_dense1 = (d->flags & MDIE_FWD_LATE_DENSE1) != 0;
  RUN(side_dense(1, h2, w2));
  if (!late_dense1) RUN(side_dense(0, h1, w1));
#endif
  // bottleneck, models/cdan.py:173
  RUN(run_cbam_stage(c, P, CB_BOTT, h3, w3, P.e, nullptr, P.bott, fuse_e ? slabs_e : 0));
  FORK_AT(4);
  // Decoder.forward, models/cdan.py:126-159
  RUN(run_conv(c, fuse_d ? "dec.conv1+skip2+pool-stats" : "dec.conv1+skip2", CV_D1, h3, w3, {P.bott}, P.t1, MDIE_ACT_RELU, 0, &P.o[2], nullptr, fuse_d ? pool_buf : nullptr));   // convT+BN+ReLU, + skip2
#ifdef EXP_SCHED
  FORK_AT(5);
#else
  if (late_dense1) RUN(side_dense(0, h1, w1));
#endif
  // (a DenseBlock branch COULD be joined in front of the last pass of the CBAM that multiplies with it -- the gate and channel-pool passes do
  //  not read it -- through cbam_fwd_hooked: built in round 5, -0.5 % ... 0 on a fast box; kept as an exploration switch)
  struct JoinCtx { Branches* br; int k; } jc[3] = {{&br, 0}, {&br, 1}, {&br, 2}};
  cbam_hook_fn join_hook = nullptr;
#ifdef EXP_SCHED   // schedule-exploration builds only: MDIE_EXP_LATE_JOIN=1 (measured on a fast box: -0.5 % ... 0, the CBAM's first passes then run against the branch)
  if (getenv("MDIE_EXP_LATE_JOIN")) join_hook = [](void* p) -> int { JoinCtx* j = static_cast<JoinCtx*>(p); return j->br->join(j->k); };
#endif
  if (!join_hook) RUN(join_dense(2));
  RUN(run_cbam_stage(c, P, CB_1, h3, w3, P.t1, &P.d[2], P.u1, fuse_d ? slabs_d : 0, join_hook, &jc[2]));  // cbam1, *= dense3
  RUN(join_dense(2));      // (no-op when joined already)
  FORK_AT(6);
  RUN(run_conv(c, "dec.conv2", CV_D2, h3, w3, {P.u1}, P.t2lo, MDIE_ACT_RELU, 0, nullptr));
  FORK_AT(7);
  RUN(run_up(c, P, "up2+skip1+pool", h3, w3, P.t2lo, P.o[1], P.t2));                    // bilinear x2 + skip1
  FORK_AT(8);
  if (!join_hook) RUN(join_dense(1));
  RUN(run_cbam_stage(c, P, CB_2, h2, w2, P.t2, &P.d[1], P.u2, mdie_pool_slabs(h2, w2), join_hook, &jc[1]));
  RUN(join_dense(1));
  FORK_AT(9);
  RUN(run_conv(c, "dec.conv3", CV_D3, h2, w2, {P.u2}, P.t3lo, MDIE_ACT_RELU, 0, nullptr));
  FORK_AT(10);
  RUN(run_up(c, P, "up3+skip0+pool", h2, w2, P.t3lo, P.o[0], P.t3));
  if (!join_hook) RUN(join_dense(0));
  RUN(run_cbam_stage(c, P, CB_3, h1, w1, P.t3, &P.d[0], P.u3, mdie_pool_slabs(h1, w1), join_hook, &jc[0]));
  RUN(join_dense(0));
  // (cbam3's last pass as decoder.conv4's staging prologue -- mdie_cbam_conv_fwd, rounds 4 -- measured 68-69 us against 41 + 27 for the two
  //  launches and was removed in round 5; profiles/LEDGER.md has the stamps and what a form that could win would look like)
  RUN(run_conv(c, "dec.conv4", CV_D4, h1, w1, {P.u3}, P.t4lo, MDIE_ACT_RELU, 0, nullptr));
  bool half_base = false;
  {
    // bilinear x2 + x (x read from its fp32 NCHW planes), final_dense, sigmoid written straight to NCHW
    // upsample + x and final_dense layer 0 in one launch (csrc/updense0.hip); then layers 1..3 and the transition.
    // 16-bit types on pictures of whole 16x16 tiles: the transition (BN -> ReLU -> Conv1x1 67 -> 3 -> sigmoid) is FOLDED into the
    // four producers of its input (mdie_tr_fuse, include/mdie.h): each adds the term of the channels it has just computed to
    // a 16-byte fp32 partial per pixel, the last one applies bias + sigmoid, writes NCHW and never stores its growth map.
    // The choice depends on the element type and the picture size only, never on the batch: an image's bits do not
    // depend on what it is batched with.  Everything else (fp32, ragged extents) runs the general chain.
    const int id0 = CV_DENSE0 + 3 * 5;
    const bool fold_tr = !(d->flags & MDIE_FWD_GENERAL_TAIL) && d->dtype != MDIE_F32 && H % 16 == 0 && W % 16 == 0 && (size_t)W * 20 < ((size_t)1 << 24) && (size_t)H * W * 16 < ((size_t)1 << 32);
    // the block's base has 3 real channels: in the folded chain it is stored as HALF a 16-byte group (4 channels, 8 bytes per pixel;
    // mdie_seg) -- written once and read by three launches with an 18x18 halo, 38 of the chain's ~470 bytes per pixel
    Buf t4 = P.t4;
#ifndef EXP_FULL_BASE   // (A/B builds only: the base as a whole 16-byte group)
    if (fold_tr && t4.C == 8) { t4.st = 4; half_base = true; }
#endif
    // ONE launch for the whole block (csrc/final_block.hip, ABI 27; MDIE_FWD_BLOCK_TAIL) wherever the folded chain would run with one weight set
    // and nobody asks for the base tensor (taps): growth maps in LDS, halo rings recomputed, y bit-identical to the chain's.  OPT-IN: measured
    // 292-319 us against the chain's 235 us at B = 32, 256x256, bf16 (round 6; profiles/r06*_final_block_*, LEDGER round 6)
    const bool one_launch = fold_tr && (d->flags & MDIE_FWD_BLOCK_TAIL) && !d->taps && !c.delta;
    if (one_launch) {
      mdie_final_dense_desc f{};
      f.dtype = d->dtype; f.B = B; f.H = H; f.W = W;
      f.lo = c.ws + P.t4lo.off; f.lo_stride = P.t4lo.C; f.x = d->x;
      f.w0 = c.params + c.L.fl0_w;
      for (int l = 0; l < 4; ++l) {
        if (l) f.w[l - 1] = c.params + c.L.conv[id0 + l].w;
        f.pre_scale[l] = reinterpret_cast<const float*>(c.params + c.L.conv[id0 + l].pre_scale);
        f.pre_shift[l] = reinterpret_cast<const float*>(c.params + c.L.conv[id0 + l].pre_shift);
        f.post_scale[l] = reinterpret_cast<const float*>(c.params + c.L.conv[id0 + l].post_scale);
        f.post_shift[l] = reinterpret_cast<const float*>(c.params + c.L.conv[id0 + l].post_shift);
      }
      f.wt = c.params + c.L.conv[id0 + 4].w;
      f.tr_pre_scale = reinterpret_cast<const float*>(c.params + c.L.conv[id0 + 4].pre_scale);
      f.tr_pre_shift = reinterpret_cast<const float*>(c.params + c.L.conv[id0 + 4].pre_shift);
      f.tr_post_scale = reinterpret_cast<const float*>(c.params + c.L.conv[id0 + 4].post_scale);
      f.tr_post_shift = reinterpret_cast<const float*>(c.params + c.L.conv[id0 + 4].post_shift);
      f.y = d->y;
      const int from = notes.mark();
      if (!ablated("final.block")) RUN(mdie_final_dense_fwd(&f, stream));
      if (notes.on()) {   // booked at the block's SURVEY 8d share: what the four launches of the chain are booked at, summed
        double el = 9.0 * 3 * PX / 4 + 3 * PX + 16 * PX + (3 + 16) * PX, par = 3.0 * 16 * 9 + 16 + 2 * 3, fl = 2.0 * 3 * 16 * 9 * PX + 2.0 * (3 + 16) * 3 * PX;
        for (int l = 1; l <= 3; ++l) {
          const double cin = 3 + 16.0 * l;
          el += cin * PX + 16 * PX + 16 * PX + (l == 3 ? 3 * PX : 0);
          par += cin * 16 * 9 + 16 + 2 * cin + (l == 3 ? 67.0 * 3 + 3 + 2 * 67 : 0);
          fl += 2.0 * cin * 16 * 9 * PX + 2.0 * 16 * 3 * PX;
        }
        notes.note(from, "up4+x+final_dense+sigmoid->nchw", el, par, fl);
      }
    } else {
    mdie_tr_fuse tr{};
    tr.weight = c.params + c.L.conv[id0 + 4].w;
    tr.pre_scale = reinterpret_cast<const float*>(c.params + c.L.conv[id0 + 4].pre_scale);
    tr.pre_shift = reinterpret_cast<const float*>(c.params + c.L.conv[id0 + 4].pre_shift);
    float* const trp = reinterpret_cast<float*>(c.ws + P.tr_ws);
    tr.partial_in = trp; tr.partial_out = trp;
    {
      mdie_up_dense0_desc u{};
      u.dtype = d->dtype; u.B = B; u.H = H; u.W = W;
      u.lo = c.ws + P.t4lo.off; u.lo_stride = P.t4lo.C; u.x = d->x;
      u.base = c.ws + t4.off; u.base_channels = t4.C; u.base_stride = t4.st;
      u.weight = c.params + c.L.fl0_w;
      u.pre_scale = reinterpret_cast<const float*>(c.params + c.L.conv[id0].pre_scale);
      u.pre_shift = reinterpret_cast<const float*>(c.params + c.L.conv[id0].pre_shift);
      u.bias = reinterpret_cast<const float*>(c.params + c.L.conv[id0].post_shift);
      u.g0 = c.ws + P.fg[0].off; u.g0_stride = P.fg[0].C;
      mdie_tr_fuse t0 = tr;
      t0.c0 = P.t4.C;                                       // g0 follows the base group in the transition's stored input
      if (fold_tr) u.tr = &t0;
      u.blob_delta = c.delta;
      const int from = notes.mark();
      if (!ablated("final.l0")) RUN(mdie_up_add_dense0_fwd(&u, stream));
      if (notes.on()) {
        // upsample + x (read lo, read x, the sum written once as the block's base), layer 0 (reads the base, writes g0); folded: + the
        // transition's terms of the base and of g0
        const ConvSpec& l0 = arch(d->dtype).conv[id0];
        const double up_el = 9.0 * 3 * PX / 4, l0_el = 3 * PX + 16 * PX, l0_par = 3.0 * 16 * 9 + 16 + 2 * 3, l0_fl = 2.0 * 3 * 16 * 9 * PX;
        (void)l0;
        if (fold_tr) notes.note(from, "up4+x+final.l0+tr", up_el + l0_el + (3 + 16) * PX, l0_par, l0_fl + 2.0 * (3 + 16) * 3 * PX);
        else notes.note(from, "up4+x+final.l0", up_el + l0_el, l0_par, l0_fl);
      }
    }
    if (fold_tr) {
      mdie_tr_fuse t1 = tr, t2 = tr, t3 = tr;
      t1.c0 = P.t4.C + 16; t2.c0 = P.t4.C + 32; t3.c0 = P.t4.C + 48;
      t3.partial_out = nullptr;
      t3.post_scale = reinterpret_cast<const float*>(c.params + c.L.conv[id0 + 4].post_scale);
      t3.post_shift = reinterpret_cast<const float*>(c.params + c.L.conv[id0 + 4].post_shift);
      t3.act = MDIE_ACT_SIGMOID; t3.out_nchw3 = d->y;
      // (the folded transition is booked with its producers: each reads its 16 new channels once more in the model's terms, the
      //  last one writes the 3 outputs and carries the transition's parameters)
      const double tr_par = 67.0 * 3 + 3 + 2 * 67;
      for (int l = 1; l <= 3; ++l) {
        const int from = notes.mark();
        const mdie_tr_fuse* tl = l == 1 ? &t1 : l == 2 ? &t2 : &t3;
        { static const char* const an[4] = {"", "final.l1", "final.l2", "final.l3"}; if (ablated(an[l])) continue; }
        if (l == 1) RUN(run_conv(c, nullptr, id0 + 1, H, W, {t4, P.fg[0]}, P.fg[1], MDIE_ACT_NONE, 0, nullptr, nullptr, nullptr, tl));
        else if (l == 2) RUN(run_conv(c, nullptr, id0 + 2, H, W, {t4, P.fg[0], P.fg[1]}, P.fg[2], MDIE_ACT_NONE, 0, nullptr, nullptr, nullptr, tl));
        else RUN(run_conv(c, nullptr, id0 + 3, H, W, {t4, P.fg[0], P.fg[1], P.fg[2]}, P.fg[3], MDIE_ACT_NONE, 0, nullptr, nullptr, nullptr, tl));
        if (notes.on()) {
          const double cin = 3 + 16.0 * l;
          static const char* const nm[4] = {"", "final.l1+tr", "final.l2+tr", "final.l3+tr+sigmoid->nchw"};
          notes.note(from, nm[l], cin * PX + 16 * PX + 16 * PX + (l == 3 ? 3 * PX : 0), cin * 16 * 9 + 16 + 2 * cin + (l == 3 ? tr_par : 0),
                     2.0 * cin * 16 * 9 * PX + 2.0 * 16 * 3 * PX);
        }
      }
    } else {
      RUN(run_dense(c, 3, H, W, P.t4, P.fg, P.out16, MDIE_ACT_SIGMOID, d->y, true));
    }
    }
  }
#undef RUN
  if (d->taps) {
    auto tap = [&](int id, const Buf& b, int h, int w) { d->taps[id] = mdie_tap{c.ws + b.off, b.C, b.C, h, w}; };
    tap(MDIE_TAP_SKIP0, P.o[0], h1, w1); tap(MDIE_TAP_SKIP1, P.o[1], h2, w2); tap(MDIE_TAP_SKIP2, P.o[2], h3, w3);
    tap(MDIE_TAP_DENSE0, P.d[0], h1, w1); tap(MDIE_TAP_DENSE1, P.d[1], h2, w2); tap(MDIE_TAP_DENSE2, P.d[2], h3, w3);
    tap(MDIE_TAP_ENC, P.e, h3, w3); tap(MDIE_TAP_BOTT, P.bott, h3, w3);
    tap(MDIE_TAP_DEC1, P.u1, h3, w3); tap(MDIE_TAP_DEC2, P.u2, h2, w2); tap(MDIE_TAP_DEC3, P.u3, h1, w1);
    tap(MDIE_TAP_DEC4, P.t4, H, W);
    if (half_base) d->taps[MDIE_TAP_DEC4] = mdie_tap{c.ws + P.t4.off, 4, 4, H, W};   // (the folded chain stores 4 of the base group's 8 channels)
  }
  return MDIE_OK;
}

}  // namespace

extern "C" int mdie_aux_create(void** out) {
  MDIE_REQUIRE(out != nullptr, "mdie_aux_create: null argument");
  Aux* a = new Aux();
  for (int i = 0; i < 3; ++i) { a->side[i] = nullptr; a->fork[i] = nullptr; a->join[i] = nullptr; }
  hipError_t err = hipSuccess;
  for (int i = 0; i < 3 && err == hipSuccess; ++i) {
#ifdef EXP_SCHED   // schedule-exploration builds only: MDIE_EXP_SIDE_PRIO = low | high -- the side streams' priority against the caller's stream
    if (const char* v = getenv("MDIE_EXP_SIDE_PRIO")) {
      int least = 0, greatest = 0;
      (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
      err = hipStreamCreateWithPriority(&a->side[i], hipStreamNonBlocking, v[0] == 'l' ? least : greatest);
    } else
#endif
    err = hipStreamCreateWithFlags(&a->side[i], hipStreamNonBlocking);
    // Fork / join events with a DEVICE-scope release (hipEventReleaseToDevice): they order the caller's stream and the library's side
    // streams on ONE GPU, and the default -- a system-scope release, i.e. a write-back towards the host at every record -- sits on the
    // critical path three times per forward at each end of a branch.  Same-box A/B (tools/sched_sweep.py, profiles/r05ak_event_scope.txt):
    // -5 ... -11 us per step in every schedule, outputs bit-identical.  (Nothing outside the device ever waits on these events.)
    const unsigned evf = hipEventDisableTiming | hipEventReleaseToDevice;
    if (err == hipSuccess) err = hipEventCreateWithFlags(&a->fork[i], evf);
    if (err == hipSuccess) err = hipEventCreateWithFlags(&a->join[i], evf);
  }
  if (err != hipSuccess) {   // give back what was created before the failure
    set_error("mdie_aux_create: HIP stream/event creation failed: %s", hipGetErrorString(err));
    mdie_aux_destroy(a);
    return MDIE_ELAUNCH;
  }
  *out = a;
  return MDIE_OK;
}

extern "C" void mdie_aux_destroy(void* aux) {
  Aux* a = reinterpret_cast<Aux*>(aux);
  if (!a) return;
  for (int i = 0; i < 3; ++i) {
    if (a->side[i]) (void)hipStreamDestroy(a->side[i]);
    if (a->fork[i]) (void)hipEventDestroy(a->fork[i]);
    if (a->join[i]) (void)hipEventDestroy(a->join[i]);
  }
  delete a;
}

extern "C" int mdie_cdan_forward(const mdie_cdan_fwd_desc* d, void* stream) {
  MDIE_REQUIRE(d != nullptr, "mdie_cdan_forward: null descriptor");
  MDIE_REQUIRE(dtype_valid(d->dtype), "mdie_cdan_forward: bad dtype %d", d->dtype);
  MDIE_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0, "mdie_cdan_forward: empty batch %dx%dx%d", d->B, d->H, d->W);
  MDIE_REQUIRE(d->H % 8 == 0 && d->W % 8 == 0,
               "mdie_cdan_forward: H, W must be multiples of 8 (three 2x2 pools + three x2 upsamples with skip adds), got %dx%d", d->H, d->W);
  MDIE_REQUIRE(d->params && d->x && d->y && d->workspace, "mdie_cdan_forward: null pointer");
  MDIE_REQUIRE((((uintptr_t)d->params | (uintptr_t)d->workspace) & 255) == 0, "mdie_cdan_forward: params/workspace must be 256-byte aligned");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (!d->launch_ms) return forward_impl(d, s);

  // instrumented mode: one event pair per launch (not capturable; synchronises at the end)
  MDIE_REQUIRE(d->launch_kind && d->n_launches && d->max_launches > 0, "mdie_cdan_forward: instrumentation buffers missing");
  LaunchTimer t;
  t.stream = s; t.cap = d->max_launches; t.kind = d->launch_kind; t.info = d->launch_info;
  if (t.info) memset(t.info, 0, sizeof(mdie_launch_info) * (size_t)t.cap);
  std::vector<hipEvent_t> ev(2 * (size_t)t.cap);
  for (auto& x : ev) (void)hipEventCreate(&x);
  t.ev = ev.data();
  current_timer() = &t;
  const int rc = forward_impl(d, s);
  current_timer() = nullptr;
  (void)hipStreamSynchronize(s);
  for (int i = 0; i < t.n; ++i) (void)hipEventElapsedTime(&d->launch_ms[i], ev[2 * i], ev[2 * i + 1]);
  *d->n_launches = t.n;
  for (auto& x : ev) (void)hipEventDestroy(x);
  return rc;
}
