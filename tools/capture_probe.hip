// Stream-capture probe for ROCm 7.2 / gfx950: which fork/join shapes survive hipStreamEndCapture -- and (round 5) hipGraphLaunch?
// Background: capturing mdie_cdan_forward (which forks its encoder DenseBlocks onto side streams) from a stream that is
// itself a fork inside a capture took the process down in hipStreamEndCapture (round 1, tools/bench_streams.py).  Each
// variant below runs in its own child process (forked BEFORE any HIP call), so a crash in one is just a result line.
//
//   hipcc --offload-arch=gfx950 -O2 tools/capture_probe.hip -o build/capture_probe && build/capture_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <unistd.h>

#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("    %s -> %s\n", #x, hipGetErrorString(e_)); fflush(stdout); return 10; } } while (0)

__global__ void fill(float* p, float v, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = v; }
__global__ void sum3(const float* a, const float* b, const float* c, float* o, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) o[i] = a[i] + b[i] + c[i]; }

constexpr int N = 1 << 16;

// work on `s`: a, then a nested fork for b (mode), c on `s` meanwhile, join, out = a + b + c
//   mode 0: nested fork through a second stream `side` (hipStreamWaitEvent both ways)
//   mode 1: no second stream: the b launch is captured on `s`, then the stream's dependency set is put back to the fork
//           point (hipStreamUpdateCaptureDependencies SET) and b's node is added again before the join (ADD)
static int body(hipStream_t s, hipStream_t side, hipEvent_t fork, hipEvent_t join, float* a, float* b, float* c, float* o, int mode) {
  hipLaunchKernelGGL(fill, dim3(N / 256), dim3(256), 0, s, a, 1.f, N);
  if (mode == 0) {
    CK(hipEventRecord(fork, s));
    CK(hipStreamWaitEvent(side, fork, 0));
    hipLaunchKernelGGL(fill, dim3(N / 256), dim3(256), 0, side, b, 2.f, N);
    CK(hipEventRecord(join, side));
    hipLaunchKernelGGL(fill, dim3(N / 256), dim3(256), 0, s, c, 4.f, N);
    CK(hipStreamWaitEvent(s, join, 0));
  } else {
    hipStreamCaptureStatus st; unsigned long long id; hipGraph_t g; const hipGraphNode_t* deps; size_t nd;
    CK(hipStreamGetCaptureInfo_v2(s, &st, &id, &g, &deps, &nd));
    if (st != hipStreamCaptureStatusActive) { printf("    not capturing\n"); return 11; }
    std::vector<hipGraphNode_t> at_fork(deps, deps + nd);
    hipLaunchKernelGGL(fill, dim3(N / 256), dim3(256), 0, s, b, 2.f, N);
    CK(hipStreamGetCaptureInfo_v2(s, &st, &id, &g, &deps, &nd));
    std::vector<hipGraphNode_t> branch(deps, deps + nd);
    CK(hipStreamUpdateCaptureDependencies(s, at_fork.data(), at_fork.size(), hipStreamSetCaptureDependencies));
    hipLaunchKernelGGL(fill, dim3(N / 256), dim3(256), 0, s, c, 4.f, N);
    CK(hipStreamUpdateCaptureDependencies(s, branch.data(), branch.size(), hipStreamAddCaptureDependencies));
  }
  hipLaunchKernelGGL(sum3, dim3(N / 256), dim3(256), 0, s, a, b, c, o, N);
  return 0;
}

// variant: (outer fork levels, inner mode, side stream flags)
//   outer = 0: body runs on the capture's origin stream;  outer = 1: on a stream forked from the origin (two of them, like
//   two engines on two streams);  nonblocking: side/outer streams created with hipStreamNonBlocking
static int variant(int outer, int mode, int nonblocking) {
  CK(hipSetDevice(0));
  const unsigned fl = nonblocking ? hipStreamNonBlocking : hipStreamDefault;
  hipStream_t origin, mid[2], side[2];
  hipEvent_t ev[8];
  CK(hipStreamCreateWithFlags(&origin, hipStreamNonBlocking));
  for (int i = 0; i < 2; ++i) { CK(hipStreamCreateWithFlags(&mid[i], fl)); CK(hipStreamCreateWithFlags(&side[i], fl)); }
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  float* buf; CK(hipMalloc(&buf, (size_t)8 * N * sizeof(float)));
  CK(hipMemset(buf, 0, (size_t)8 * N * sizeof(float)));
  CK(hipDeviceSynchronize());
  CK(hipStreamBeginCapture(origin, hipStreamCaptureModeGlobal));
  if (outer == 0) {
    if (int e = body(origin, side[0], ev[0], ev[1], buf, buf + N, buf + 2 * N, buf + 3 * N, mode)) return e;
  } else {
    CK(hipEventRecord(ev[4], origin));
    for (int i = 0; i < 2; ++i) {
      CK(hipStreamWaitEvent(mid[i], ev[4], 0));
      float* b = buf + (size_t)i * 4 * N;
      if (int e = body(mid[i], side[i], ev[2 * i], ev[2 * i + 1], b, b + N, b + 2 * N, b + 3 * N, mode)) return e;
      CK(hipEventRecord(ev[5 + i], mid[i]));
    }
    for (int i = 0; i < 2; ++i) CK(hipStreamWaitEvent(origin, ev[5 + i], 0));
  }
  hipGraph_t graph;
  printf("    end capture...\n"); fflush(stdout);
  CK(hipStreamEndCapture(origin, &graph));
  size_t nn = 0; CK(hipGraphGetNodes(graph, nullptr, &nn));
  hipGraphExec_t exec;
  CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(exec, origin));
  CK(hipStreamSynchronize(origin));
  std::vector<float> h(N);
  const int outs = outer ? 2 : 1;
  for (int i = 0; i < outs; ++i) {
    CK(hipMemcpy(h.data(), buf + (size_t)i * 4 * N + 3 * N, N * sizeof(float), hipMemcpyDeviceToHost));
    for (int j = 0; j < N; ++j) if (h[j] != 7.f) { printf("    wrong value %f at %d (branch %d)\n", h[j], j, i); return 12; }
  }
  printf("    ok: %zu nodes\n", nn);
  return 0;
}

// Two captures in a row from the SAME origin and mid streams (a serving process captures more than one graph; torch keeps
// its capture stream for the life of the process), with the inner side streams of the first capture destroyed before the
// second one -- what happened in tools/bench_streams.py when the k = 1 engines were released before the k = 2 capture.
//   destroy_side = 1: hipStreamDestroy(side) between the captures;  mode as in body()
static int twice(int mode, int destroy_side) {
  CK(hipSetDevice(0));
  hipStream_t origin, mid;
  hipEvent_t ev[6];
  CK(hipStreamCreateWithFlags(&origin, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&mid, hipStreamNonBlocking));
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  float* buf; CK(hipMalloc(&buf, (size_t)4 * N * sizeof(float)));
  CK(hipDeviceSynchronize());
  for (int round = 0; round < 3; ++round) {
    hipStream_t side;
    CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    CK(hipStreamBeginCapture(origin, hipStreamCaptureModeGlobal));
    CK(hipEventRecord(ev[4], origin));
    CK(hipStreamWaitEvent(mid, ev[4], 0));
    if (int e = body(mid, side, ev[0], ev[1], buf, buf + N, buf + 2 * N, buf + 3 * N, mode)) return e;
    CK(hipEventRecord(ev[5], mid));
    CK(hipStreamWaitEvent(origin, ev[5], 0));
    hipGraph_t graph;
    printf("    round %d: end capture...\n", round); fflush(stdout);
    CK(hipStreamEndCapture(origin, &graph));
    hipGraphExec_t exec;
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    CK(hipGraphLaunch(exec, origin));
    CK(hipStreamSynchronize(origin));
    hipStreamCaptureStatus st;
    CK(hipStreamIsCapturing(side, &st));
    printf("    round %d: ok; side stream capture status after EndCapture = %d (0 = none)\n", round, (int)st); fflush(stdout);
    if (destroy_side) CK(hipStreamDestroy(side));
    CK(hipGraphExecDestroy(exec));
    CK(hipGraphDestroy(graph));
  }
  return 0;
}

// Round 4's hipGraphLaunch segfault (profiles/r04e_graph_replay_segfault.log): RoutedEngine's "groups" mode had captured ONE graph per
// task group -- nine graphs of ~39 kernel nodes, each with the engine's fork-join expressed through hipStreamUpdateCaptureDependencies
// -- and replayed them in a row on nine streams (torch.cuda.CUDAGraph.replay from worker threads).  The pure-HIP shape of that:
//   threads = 0: the nine graphs launched round-robin from ONE host thread, 200 rounds;  threads = 1: nine host threads, one graph each.
__global__ void inc(float* p, float v, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += v; }

static int nine(int threads) {
  CK(hipSetDevice(0));
  constexpr int G = 9, ROUNDS = 200;
  hipStream_t st[G]; hipGraphExec_t ex[G]; float* buf[G];
  size_t nodes = 0;
  for (int g = 0; g < G; ++g) {
    CK(hipStreamCreateWithFlags(&st[g], hipStreamNonBlocking));
    CK(hipMalloc(&buf[g], (size_t)4 * N * sizeof(float)));
    CK(hipMemset(buf[g], 0, (size_t)4 * N * sizeof(float)));
  }
  CK(hipDeviceSynchronize());
  for (int g = 0; g < G; ++g) {
    hipStream_t s = st[g];
    float *m = buf[g], *b0 = buf[g] + N, *b1 = buf[g] + 2 * N, *b2 = buf[g] + 3 * N;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < 10; ++k) hipLaunchKernelGGL(inc, dim3(N / 256), dim3(256), 0, s, m, 1.f, N);
    hipStreamCaptureStatus cs; unsigned long long id; hipGraph_t gr; const hipGraphNode_t* deps; size_t nd;
    CK(hipStreamGetCaptureInfo_v2(s, &cs, &id, &gr, &deps, &nd));
    std::vector<hipGraphNode_t> at_fork(deps, deps + nd), tails;
    float* br[3] = {b0, b1, b2};
    for (int b = 0; b < 3; ++b) {       // three side branches (the encoder DenseBlocks), each captured on s and then detached
      for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(inc, dim3(N / 256), dim3(256), 0, s, br[b], (float)(b + 1), N);
      CK(hipStreamGetCaptureInfo_v2(s, &cs, &id, &gr, &deps, &nd));
      tails.insert(tails.end(), deps, deps + nd);
      CK(hipStreamUpdateCaptureDependencies(s, at_fork.data(), at_fork.size(), hipStreamSetCaptureDependencies));
    }
    for (int k = 0; k < 9; ++k) hipLaunchKernelGGL(inc, dim3(N / 256), dim3(256), 0, s, m, 1.f, N);
    CK(hipStreamUpdateCaptureDependencies(s, tails.data(), tails.size(), hipStreamAddCaptureDependencies));
    for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(sum3, dim3(N / 256), dim3(256), 0, s, b0, b1, b2, m, N);   // m = 5 + 10 + 15 = 30 (after the join)
    hipGraph_t graph;
    CK(hipStreamEndCapture(s, &graph));
    size_t nn = 0; CK(hipGraphGetNodes(graph, nullptr, &nn)); nodes = nn;
    CK(hipGraphInstantiate(&ex[g], graph, nullptr, nullptr, 0));
  }
  printf("    nine graphs of %zu nodes instantiated\n", nodes); fflush(stdout);
  // (the branch buffers accumulate over replays: a replay r leaves b = 5 (b+1) r, m = 30 r -- any lost dependency shows)
  if (!threads) {
    for (int r = 0; r < ROUNDS; ++r) for (int g = 0; g < G; ++g) CK(hipGraphLaunch(ex[g], st[g]));
  } else {
    std::vector<std::thread> th; int rc[G] = {};
    for (int g = 0; g < G; ++g) th.emplace_back([&, g] { for (int r = 0; r < ROUNDS; ++r) if (hipGraphLaunch(ex[g], st[g]) != hipSuccess) { rc[g] = 1; return; } });
    for (auto& t : th) t.join();
    for (int g = 0; g < G; ++g) if (rc[g]) { printf("    hipGraphLaunch failed in thread %d\n", g); return 13; }
  }
  CK(hipDeviceSynchronize());
  std::vector<float> h(N);
  for (int g = 0; g < G; ++g) {
    CK(hipMemcpy(h.data(), buf[g], N * sizeof(float), hipMemcpyDeviceToHost));
    for (int j = 0; j < N; ++j) if (h[j] != 30.f * ROUNDS) { printf("    graph %d: wrong value %f at %d (want %f)\n", g, h[j], j, 30.f * ROUNDS); return 12; }
  }
  printf("    ok: %d rounds of nine replays, values right\n", ROUNDS);
  return 0;
}

int main() {
  struct V { int outer, mode, nb; const char* what; } vs[] = {
    {0, 0, 1, "fork from the ORIGIN stream through a non-blocking side stream (what bench.py captures)"},
    {1, 0, 0, "nested: two forked streams, each forks again through a BLOCKING side stream"},
    {1, 0, 1, "nested: two forked streams, each forks again through a NON-BLOCKING side stream (round 1's crash shape)"},
    {0, 1, 1, "origin stream, branch expressed with hipStreamUpdateCaptureDependencies (no second stream)"},
    {1, 1, 1, "two forked streams, each branches with hipStreamUpdateCaptureDependencies (no nested stream fork)"},
  };
  struct W { int mode, destroy; const char* what; } ws[] = {
    {0, 0, "three captures in a row, nested stream fork, side streams kept"},
    {0, 1, "three captures in a row, nested stream fork, the side stream DESTROYED after each capture"},
    {1, 1, "three captures in a row, branch by capture dependencies (side stream never joins a capture), destroyed after each"},
  };
  int k = 0;
  for (const W& w : ws) {
    printf("[T%d] %s\n", k++, w.what); fflush(stdout);
    const pid_t pid = fork();
    if (pid == 0) { const int rc = twice(w.mode, w.destroy); fflush(stdout); _exit(rc); }
    int st = 0; waitpid(pid, &st, 0);
    if (WIFSIGNALED(st)) printf("    => KILLED by signal %d (%s)\n", WTERMSIG(st), strsignal(WTERMSIG(st)));
    else printf("    => exit code %d\n", WEXITSTATUS(st));
    fflush(stdout);
  }
  for (int threads = 0; threads < 2; ++threads) {
    printf("[N%d] nine ~39-node graphs with capture-dependency fork-joins, replayed on nine streams from %s\n", threads, threads ? "NINE host threads" : "ONE host thread"); fflush(stdout);
    const pid_t pid = fork();
    if (pid == 0) { const int rc = nine(threads); fflush(stdout); _exit(rc); }
    int st = 0; waitpid(pid, &st, 0);
    if (WIFSIGNALED(st)) printf("    => KILLED by signal %d (%s)\n", WTERMSIG(st), strsignal(WTERMSIG(st)));
    else printf("    => exit code %d\n", WEXITSTATUS(st));
    fflush(stdout);
  }
  k = 0;
  for (const V& v : vs) {
    printf("[%d] %s\n", k++, v.what); fflush(stdout);
    const pid_t pid = fork();
    if (pid == 0) { const int rc = variant(v.outer, v.mode, v.nb); fflush(stdout); _exit(rc); }
    int st = 0; waitpid(pid, &st, 0);
    if (WIFSIGNALED(st)) printf("    => KILLED by signal %d (%s)\n", WTERMSIG(st), strsignal(WTERMSIG(st)));
    else printf("    => exit code %d\n", WEXITSTATUS(st));
    fflush(stdout);
  }
  return 0;
}
