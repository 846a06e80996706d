// Hardware probe for DESIGN.md section 4, finding 6: does `v_pk_fma_f32` reading a VGPR pair through op_sel / op_sel_hi return
// wrong values on gfx950 while ANOTHER stream's MFMA kernel is resident on the same CUs?
//
// Round 3 saw the bottleneck CBAM's MLP gradients differ run to run once weight-gradient MFMA kernels ran beside the MFMA-free
// cbt_gate_bwd_kernel (the only kernel of that path whose inner loops are the swizzled form, 10 of its 11 packed instructions:
// `v_pk_fma_f32 v[4:5], v[20:21], v[26:27], v[4:5] op_sel_hi:[0,1,1]`, fed by global loads under partial s_waitcnt vmcnt).
// This is the minimal form of that situation, ONE process, each victim launch checking itself bit for bit in registers:
//
//   victim "alu"    : operands in registers; swizzled packed fma vs two scalar v_fma_f32 of the same operands
//   victim "stream" : the loop of cbt_gate_bwd_kernel -- w[c] from global memory (4 loads in flight, counted waits), (avg, mx)
//                     pairs from LDS, accumulate (sa, sm) with the swizzled packed form AND with scalar fmas; compare the sums
//   victim "plain"  : control -- packed fma on distinct register pairs without op_sel (the form conv*.hip keeps)
//   aggressor       : v_mfma_f32_16x16x32_bf16 back to back, one wave per SIMD on every CU, on a second stream
//
// Each victim runs alone and under the aggressor; a mismatch is counted per lane (the round-2 case was lanes 48..63).
//   hipcc --offload-arch=gfx950 -O2 tools/pkfma_probe.hip -o build/pkfma_probe && build/pkfma_probe [launches]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Result { unsigned long long checks; unsigned int bad; unsigned int lane_hist[64]; unsigned int first[4][6]; };

__device__ __forceinline__ void report(Result* r, unsigned it, float got0, float got1, float exp0, float exp1) {
  const unsigned k = atomicAdd(&r->bad, 1u);
  atomicAdd(&r->lane_hist[threadIdx.x & 63], 1u);
  if (k < 4) {
    r->first[k][0] = it; r->first[k][1] = threadIdx.x;
    r->first[k][2] = __float_as_uint(got0); r->first[k][3] = __float_as_uint(exp0);
    r->first[k][4] = __float_as_uint(got1); r->first[k][5] = __float_as_uint(exp1);
  }
}

// d = {a.x * s.x + t.x, a.y * s.x + t.y}  (src1 splat of its LOW register: op_sel_hi:[1,0,1])
__device__ __forceinline__ f32x2 pk_fma_splat1(f32x2 a, f32x2 s, f32x2 t) {
  f32x2 d;
  asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(s), "v"(t));
  return d;
}
// d = {w.x * v.x + acc.x, w.x * v.y + acc.y}  (src0 splat: the exact form in cbt_gate_bwd_kernel)
__device__ __forceinline__ f32x2 pk_fma_splat0(f32x2 w, f32x2 v, f32x2 acc) {
  f32x2 d;
  asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(d) : "v"(w), "v"(v), "v"(acc));
  return d;
}
__device__ __forceinline__ f32x2 pk_fma_plain(f32x2 a, f32x2 b, f32x2 c) {
  f32x2 d;
  asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ float s_fma(float a, float b, float c) {
  float d;
  asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

template <int FORM>   // 0: src1 splat, 1: src0 splat, 2: plain
__global__ __launch_bounds__(256) void victim_alu(Result* r, int iters, unsigned seed) {
  const unsigned tid = blockIdx.x * 256 + threadIdx.x;
  float x = 1.0f + (float)((tid * 2654435761u + seed) >> 9) * (1.0f / 8388608.0f);
  float y = 1.5f - (float)((tid * 40503u + seed * 7u) & 0xffff) * (1.0f / 131072.0f);
  float s = 0.999f + (float)(threadIdx.x & 63) * 1e-5f, t = 1e-3f;
  unsigned long long n = 0;
  for (int it = 0; it < iters; ++it) {
    f32x2 got;
    float e0, e1;
    if (FORM == 0) { got = pk_fma_splat1(f32x2{x, y}, f32x2{s, __uint_as_float(0x7fc00000u ^ it)}, f32x2{t, t}); e0 = s_fma(x, s, t); e1 = s_fma(y, s, t); }
    else if (FORM == 1) { got = pk_fma_splat0(f32x2{s, __uint_as_float(0x7fc00000u ^ it)}, f32x2{x, y}, f32x2{t, t}); e0 = s_fma(s, x, t); e1 = s_fma(s, y, t); }
    else { got = pk_fma_plain(f32x2{x, y}, f32x2{s, s}, f32x2{t, t}); e0 = s_fma(x, s, t); e1 = s_fma(y, s, t); }
    if (__float_as_uint(got.x) != __float_as_uint(e0) || __float_as_uint(got.y) != __float_as_uint(e1)) report(r, it, got.x, got.y, e0, e1);
    x = e0 > 2.0f ? s_fma(e0, 1.0f, -1.0f) : e0;     // keep the operands moving and in range (scalar forms only: no compiler-made packed math)
    y = e1 > 2.0f ? s_fma(e1, 1.0f, -1.0f) : e1;
    ++n;
  }
  if (threadIdx.x == 0) atomicAdd(&r->checks, n * 256ull);
}

// the inner loop of cbt_gate_bwd_kernel: one block per "image", C weights per hidden unit row, LPJ lanes per row
__global__ __launch_bounds__(256) void victim_stream(Result* r, const float* __restrict__ w, int C, int rows, int reps, unsigned seed) {
  extern __shared__ float lds[];       // avg[C], mx[C]
  float* avg = lds; float* mx = lds + C;
  for (int c = threadIdx.x; c < C; c += 256) { avg[c] = 0.25f + (float)((c * 2654435761u + seed) >> 10) * (1.0f / 4194304.0f); mx[c] = 1.0f - avg[c] * 0.5f; }
  __syncthreads();
  const int LPJ = 8, j = threadIdx.x / LPJ, q = threadIdx.x % LPJ;
  unsigned long long n = 0;
  for (int rep = 0; rep < reps; ++rep) {
    const float* wr = w + ((size_t)((blockIdx.x * 32 + j + rep * 7) % rows)) * C;
    f32x2 acc = {0.f, 0.f};
    float ra = 0.f, rm = 0.f;
    for (int c = q; c < C; c += 4 * LPJ) {       // four loads in flight, as the compiler unrolls the original (C is a multiple of 32)
      float wv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) wv[u] = __builtin_nontemporal_load(wr + c + u * LPJ);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float av = avg[c + u * LPJ], mv = mx[c + u * LPJ];
        acc = pk_fma_splat0(f32x2{wv[u], 0.f}, f32x2{av, mv}, acc);
        ra = s_fma(wv[u], av, ra); rm = s_fma(wv[u], mv, rm);
      }
      n += 4;
    }
    if (__float_as_uint(acc.x) != __float_as_uint(ra) || __float_as_uint(acc.y) != __float_as_uint(rm)) report(r, rep, acc.x, acc.y, ra, rm);
  }
  if (threadIdx.x == 0) atomicAdd(&r->checks, n * 256ull);
}

__global__ __launch_bounds__(256) void aggressor_mfma(float* sink, int iters) {
  bf16x8 a, b0, b1, b2, b3;
  for (int i = 0; i < 8; ++i) {
    a[i] = (__bf16)(0.001f * (float)(threadIdx.x + i));
    b0[i] = (__bf16)(0.002f * (float)(i + 1)); b1[i] = (__bf16)(0.003f * (float)(i + 1)); b2[i] = (__bf16)(0.004f * (float)(i + 1)); b3[i] = (__bf16)(0.005f * (float)(i + 1));
  }
  f32x4 c0 = {0, 0, 0, 0}, c1 = {1, 0, 0, 0}, c2 = {0, 1, 0, 0}, c3 = {0, 0, 1, 0};
  for (int it = 0; it < iters; ++it) {      // four independent accumulator chains: the matrix pipe never waits
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %4, %5, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %4, %6, %1\n\t"
                 "v_mfma_f32_16x16x32_bf16 %2, %4, %7, %2\n\tv_mfma_f32_16x16x32_bf16 %3, %4, %8, %3"
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b0), "v"(b1), "v"(b2), "v"(b3));
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  const f32x4 s = c0 + c1 + c2 + c3;
  if (s[0] == 12345.678f) sink[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

static void show(const char* what, const Result& h, float ms) {
  unsigned q[4] = {0, 0, 0, 0};
  for (int l = 0; l < 64; ++l) q[l / 16] += h.lane_hist[l];
  printf("%-46s checks %12llu  mismatches %8u  by lane quarter [%u %u %u %u]  %.2f ms\n", what, h.checks, h.bad, q[0], q[1], q[2], q[3], ms);
  for (unsigned k = 0; k < (h.bad < 4 ? h.bad : 4); ++k)
    printf("    e.g. iter %u thread %u: got (%08x, %08x) expected (%08x, %08x)\n", h.first[k][0], h.first[k][1], h.first[k][2], h.first[k][4], h.first[k][3], h.first[k][5]);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 40;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s, %d CUs\n", prop.gcnArchName, prop.multiProcessorCount);
  hipStream_t sv, sa;
  CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  Result* dr; float* sink; float* w;
  const int C = 512, rows = 4096;
  CK(hipMalloc(&dr, sizeof(Result)));
  CK(hipMalloc(&sink, 4096 * 256 * 4));
  CK(hipMalloc(&w, (size_t)rows * C * 4));
  {
    float* hw = (float*)malloc((size_t)rows * C * 4);
    unsigned s = 12345;
    for (size_t i = 0; i < (size_t)rows * C; ++i) { s = s * 1664525u + 1013904223u; hw[i] = ((float)(s >> 8) * (1.0f / 16777216.0f) - 0.5f) * 0.1f; }
    CK(hipMemcpy(w, hw, (size_t)rows * C * 4, hipMemcpyHostToDevice));
    free(hw);
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int ncu = prop.multiProcessorCount;
  // aggressor length: calibrate ~ 4 MFMA x iters; 16 cycles each at ~2 GHz -> 20000 iters ~ 0.7 ms
  const int agg_iters = 60000;
  for (int pass = 0; pass < 2; ++pass) {
    const bool with_agg = pass == 1;
    for (int victim = 0; victim < 5; ++victim) {
      CK(hipMemset(dr, 0, sizeof(Result)));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, sv));
      for (int l = 0; l < launches; ++l) {
        if (with_agg) hipLaunchKernelGGL(aggressor_mfma, dim3(ncu), dim3(256), 0, sa, sink, agg_iters);
        for (int sub = 0; sub < 4; ++sub) {
          const unsigned seed = (unsigned)(l * 4 + sub);
          switch (victim) {
            case 0: hipLaunchKernelGGL(victim_alu<0>, dim3(ncu * 2), dim3(256), 0, sv, dr, 4096, seed); break;
            case 1: hipLaunchKernelGGL(victim_alu<1>, dim3(ncu * 2), dim3(256), 0, sv, dr, 4096, seed); break;
            case 2: hipLaunchKernelGGL(victim_alu<2>, dim3(ncu * 2), dim3(256), 0, sv, dr, 4096, seed); break;
            case 3: hipLaunchKernelGGL(victim_stream, dim3(8), dim3(256), 2 * C * 4, sv, dr, w, C, rows, 64, seed); break;
            case 4: hipLaunchKernelGGL(victim_stream, dim3(ncu * 2), dim3(256), 2 * C * 4, sv, dr, w, C, rows, 16, seed); break;
          }
        }
        if (with_agg) CK(hipDeviceSynchronize());    // next aggressor starts with the next batch of victims
      }
      CK(hipEventRecord(e1, sv));
      CK(hipDeviceSynchronize());
      CK(hipGetLastError());
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      Result h;
      CK(hipMemcpy(&h, dr, sizeof(Result), hipMemcpyDeviceToHost));
      static const char* names[5] = {"alu, src1 splat op_sel_hi:[1,0,1]", "alu, src0 splat op_sel_hi:[0,1,1]", "alu, plain packed (control)",
                                     "gate_bwd loop, 8 blocks", "gate_bwd loop, 2 blocks per CU"};
      char what[128];
      snprintf(what, sizeof what, "%s | %s", with_agg ? "MFMA beside" : "alone      ", names[victim]);
      show(what, h, ms);
    }
  }
  return 0;
}
