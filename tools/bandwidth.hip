// What the memory system of an MI355X delivers to plain streaming HIP kernels -- the calibration the per-kernel TB/s figures
// of DESIGN.md are read against (next to the contract's 8 TB/s spec peak).  Own kernels, 16 bytes per lane per access,
// grid-stride over persistent workgroups; no library call in the timed region.
//   copy   : 1 read + 1 write            (float4 copy, what MI355X_MICROARCH.md quotes 6.29 TB/s for)
//   dense  : 2 reads + 1 write           (what a DenseLayer looks like to HBM: two input segments in, one growth map out)
//   wide   : 4 reads + 1 write (1/4 size) (a 64-channel input read for a 16-channel output)
//   read   : reads only (xor-folded into one word per thread so nothing is dead)
//   write  : writes only
// Build: hipcc -O3 --offload-arch=gfx950 tools/bandwidth.hip -o build/bandwidth      Run: build/bandwidth [MiB ...]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int THREADS = 256, UNROLL = 4;

__global__ __launch_bounds__(THREADS) void k_copy(const uint4* __restrict__ a, uint4* __restrict__ o, size_t n) {
  const size_t stride = (size_t)gridDim.x * THREADS * UNROLL;
  for (size_t i = (size_t)blockIdx.x * THREADS * UNROLL + threadIdx.x; i < n; i += stride) {
    uint4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = a[i + (size_t)u * THREADS];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) o[i + (size_t)u * THREADS] = v[u];
  }
}

__global__ __launch_bounds__(THREADS) void k_dense(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ o, size_t n) {
  const size_t stride = (size_t)gridDim.x * THREADS * UNROLL;
  for (size_t i = (size_t)blockIdx.x * THREADS * UNROLL + threadIdx.x; i < n; i += stride) {
    uint4 v[UNROLL], w[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) { v[u] = a[i + (size_t)u * THREADS]; w[u] = b[i + (size_t)u * THREADS]; }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) o[i + (size_t)u * THREADS] = make_uint4(v[u].x ^ w[u].x, v[u].y ^ w[u].y, v[u].z ^ w[u].z, v[u].w ^ w[u].w);
  }
}

// n = number of OUTPUT uint4; the input holds 4 n
__global__ __launch_bounds__(THREADS) void k_wide(const uint4* __restrict__ a, uint4* __restrict__ o, size_t n) {
  const size_t stride = (size_t)gridDim.x * THREADS;
  for (size_t i = (size_t)blockIdx.x * THREADS + threadIdx.x; i < n; i += stride) {
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = a[(size_t)u * n + i];
    o[i] = make_uint4(v[0].x ^ v[1].x ^ v[2].x ^ v[3].x, v[0].y ^ v[1].y ^ v[2].y ^ v[3].y, v[0].z ^ v[1].z ^ v[2].z ^ v[3].z, v[0].w ^ v[1].w ^ v[2].w ^ v[3].w);
  }
}

__global__ __launch_bounds__(THREADS) void k_read(const uint4* __restrict__ a, unsigned* __restrict__ sink, size_t n) {
  const size_t stride = (size_t)gridDim.x * THREADS * UNROLL;
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * THREADS * UNROLL + threadIdx.x; i < n; i += stride) {
    uint4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = a[i + (size_t)u * THREADS];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  if (acc == 0x12345678u) sink[0] = acc;     // (never true on random data: keeps the loads alive without a store per thread)
}

__global__ __launch_bounds__(THREADS) void k_write(uint4* __restrict__ o, size_t n, unsigned seed) {
  const size_t stride = (size_t)gridDim.x * THREADS * UNROLL;
  for (size_t i = (size_t)blockIdx.x * THREADS * UNROLL + threadIdx.x; i < n; i += stride) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) o[i + (size_t)u * THREADS] = make_uint4(seed, (unsigned)i, seed ^ u, 7u);
  }
}

__global__ void k_fill(unsigned* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = x;
  }
}

template <typename F>
static float time_us(F&& launch, int reps) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) launch();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, nullptr);
  for (int i = 0; i < reps; ++i) launch();
  (void)hipEventRecord(e1, nullptr);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return ms * 1e3f / reps;
}

int main(int argc, char** argv) {
  std::vector<size_t> mibs;
  for (int i = 1; i < argc; ++i) mibs.push_back((size_t)atol(argv[i]));
  if (mibs.empty()) mibs = {64, 256, 1024, 2048};
  size_t max_mib = 0;
  for (size_t m : mibs) max_mib = m > max_mib ? m : max_mib;
  int cus = 256;
  CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  const size_t bytes = max_mib << 20;
  uint4 *a, *b, *o;
  unsigned* sink;
  CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes)); CHECK(hipMalloc(&o, bytes)); CHECK(hipMalloc(&sink, 256));
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, nullptr, reinterpret_cast<unsigned*>(a), bytes / 4);
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, nullptr, reinterpret_cast<unsigned*>(b), bytes / 4);
  CHECK(hipDeviceSynchronize());
  printf("# MI355X streaming bandwidth, own HIP kernels (16 B per lane, %d CUs); sizes are PER BUFFER; TB/s = all bytes moved / time\n", cus);
  printf("# %8s %6s | %10s %10s %10s %10s %10s\n", "MiB", "wg/CU", "copy 1r1w", "dense 2r1w", "wide 4r1w", "read", "write");
  for (size_t mib : mibs) {
    const size_t n = (mib << 20) / 16;             // uint4 per buffer (a multiple of THREADS * UNROLL for every size used)
    for (int per_cu : {4, 8, 16}) {
      const int grid = cus * per_cu;
      const int reps = mib >= 1024 ? 10 : 30;
      const float t_copy = time_us([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(THREADS), 0, nullptr, a, o, n); }, reps);
      const float t_dense = time_us([&] { hipLaunchKernelGGL(k_dense, dim3(grid), dim3(THREADS), 0, nullptr, a, b, o, n); }, reps);
      const float t_wide = time_us([&] { hipLaunchKernelGGL(k_wide, dim3(grid), dim3(THREADS), 0, nullptr, a, o, n / 4); }, reps);
      const float t_read = time_us([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(THREADS), 0, nullptr, a, sink, n); }, reps);
      const float t_write = time_us([&] { hipLaunchKernelGGL(k_write, dim3(grid), dim3(THREADS), 0, nullptr, o, n, 3u); }, reps);
      const double B = (double)(mib << 20);
      printf("  %8zu %6d | %10.2f %10.2f %10.2f %10.2f %10.2f   TB/s   (copy %.1f us)\n", mib, per_cu, 2 * B / t_copy * 1e-6, 3 * B / t_dense * 1e-6,
             1.25 * B / t_wide * 1e-6, B / t_read * 1e-6, B / t_write * 1e-6, t_copy);
    }
  }
  CHECK(hipDeviceSynchronize());
  return 0;
}
