// Developer probe: what a chain of N dependent tiny kernels costs per launch on this GPU / runtime, eager and as a replayed hipGraph,
// on 1 and on 4 streams, for several kernel bodies:
//   empty      no memory access
//   touch      every workgroup loads 16 bytes per lane from the previous kernel's output and stores 16 bytes per lane (a "trivial layer")
//   touch + p  the same, its pointers behind one more dependent load (an argument table in device memory, like a by-pointer argument struct)
// and, for the in-kernel alternative, a persistent kernel of G workgroups that runs N "layers" separated by a counter barrier
// (sc1 stores / sc1 loads for the hand-off: no fences), reported per layer.
// hipcc -O3 --offload-arch=gfx950 tools/probes/launch_floor.hip -o tools/probes/bin/launch_floor
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_empty() {}
__global__ __launch_bounds__(256) void k_touch(const uint4* __restrict__ in, uint4* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  uint4 v = in[i];
  v.x += 1;
  out[i] = v;
}
struct Tab { const uint4* in; uint4* out; };
__global__ __launch_bounds__(256) void k_touch_p(const Tab* __restrict__ t) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  uint4 v = t->in[i];
  v.x += 1;
  t->out[i] = v;
}

typedef __attribute__((address_space(1))) unsigned gu32;
// persistent: G workgroups, N layers; layer l reads buffer (l & 1) written by layer l - 1 (all sc1), counter barrier between layers
__global__ __launch_bounds__(256) void k_persist(uint4* a, uint4* b, unsigned* counter, int N, int G, unsigned* tmo) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int j = ((blockIdx.x + 1) % G) * 256 + threadIdx.x;     // read the neighbour workgroup's data: a real cross-CU dependency
  __shared__ int fail;
  if (threadIdx.x == 0) fail = 0;
  __syncthreads();
  for (int l = 0; l < N; l++) {
    uint4* src = (l & 1) ? b : a;
    uint4* dst = (l & 1) ? a : b;
    unsigned x = __hip_atomic_load((gu32*)&src[j].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store((gu32*)&dst[i].x, x + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add((gu32*)counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = (unsigned)(l + 1) * (unsigned)G;
      long spins = 0;
      while (__hip_atomic_load((gu32*)counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 2000000) { fail = 1; atomicExch(tmo, 1u); break; }
      }
    }
    __syncthreads();
    if (fail) return;
  }
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  const int N = 64, REP = 200;
  uint4 *a, *b;
  CK(hipMalloc(&a, 256 * 256 * 16));
  CK(hipMalloc(&b, 256 * 256 * 16));
  CK(hipMemset(a, 0, 256 * 256 * 16));
  CK(hipMemset(b, 0, 256 * 256 * 16));
  Tab ht[2] = {{a, b}, {b, a}};
  Tab* dt;
  CK(hipMalloc(&dt, sizeof(ht)));
  CK(hipMemcpy(dt, ht, sizeof(ht), hipMemcpyHostToDevice));
  hipStream_t st[4];
  for (int i = 0; i < 4; i++) CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
  for (int body = 0; body < 3; body++)
    for (int G : {1, 16, 256}) {
      auto chain = [&](hipStream_t s) {
        for (int l = 0; l < N; l++) {
          if (body == 0) k_empty<<<G, 256, 0, s>>>();
          else if (body == 1) k_touch<<<G, 256, 0, s>>>((l & 1) ? b : a, (l & 1) ? a : b);
          else k_touch_p<<<G, 256, 0, s>>>(dt + (l & 1));
        }
      };
      // eager, one stream
      chain(st[0]);
      CK(hipStreamSynchronize(st[0]));
      double t0 = now();
      for (int r = 0; r < REP; r++) chain(st[0]);
      CK(hipStreamSynchronize(st[0]));
      const double eager = (now() - t0) / (REP * N) * 1e6;
      // graph, one stream and four streams
      hipGraph_t g;
      hipGraphExec_t ge[4];
      for (int i = 0; i < 4; i++) {
        CK(hipStreamBeginCapture(st[i], hipStreamCaptureModeThreadLocal));
        chain(st[i]);
        CK(hipStreamEndCapture(st[i], &g));
        CK(hipGraphInstantiate(&ge[i], g, nullptr, nullptr, 0));
        CK(hipGraphDestroy(g));
      }
      double res[2];
      for (int ns : {1, 4}) {
        for (int i = 0; i < ns; i++) CK(hipGraphLaunch(ge[i], st[i]));
        CK(hipDeviceSynchronize());
        t0 = now();
        for (int r = 0; r < REP; r++)
          for (int i = 0; i < ns; i++) CK(hipGraphLaunch(ge[i], st[i]));
        CK(hipDeviceSynchronize());
        res[ns == 4] = (now() - t0) / (REP * N) * 1e6;     // wall time per launch position (4 streams: four launches share it)
      }
      printf("body %s grid %3d: eager %.2f us/launch, graph %.2f us/launch, 4 graphs side by side %.2f us per launch position\n",
             body == 0 ? "empty  " : body == 1 ? "touch  " : "touch+p", G, eager, res[0], res[1]);
      for (int i = 0; i < 4; i++) CK(hipGraphExecDestroy(ge[i]));
    }
  unsigned *counter, *tmo;
  CK(hipMalloc(&counter, 64));
  CK(hipMalloc(&tmo, 64));
  for (int G : {8, 16, 32, 64, 128, 256}) {
    CK(hipMemset(tmo, 0, 4));
    CK(hipMemset(a, 0, 256 * 256 * 16));
    CK(hipMemset(b, 0, 256 * 256 * 16));
    double best = 1e30;
    for (int r = 0; r < 20; r++) {
      CK(hipMemsetAsync(counter, 0, 4, st[0]));
      CK(hipStreamSynchronize(st[0]));
      const double t0 = now();
      k_persist<<<G, 256, 0, st[0]>>>(a, b, counter, N, G, tmo);
      CK(hipStreamSynchronize(st[0]));
      best = std::min(best, now() - t0);
    }
    unsigned h = 0;
    CK(hipMemcpy(&h, tmo, 4, hipMemcpyDeviceToHost));
    // (the buffers are not reset between repetitions: after r runs of N layers every element of the last written buffer holds
    //  a multiple of N plus what the other buffer started with; a stale read shows as a value that is not uniform)
    std::vector<uint4> hb((size_t)G * 256);
    CK(hipMemcpy(hb.data(), (N & 1) ? b : a, hb.size() * 16, hipMemcpyDeviceToHost));
    int bad = 0;
    for (auto& v : hb) bad += v.x != hb[0].x;
    if (bad) printf("  STALE: %d of %zu elements differ from element 0 (%u)\n", bad, hb.size(), hb[0].x);
    printf("persistent %3d workgroups: %.2f us per layer (launch + sync included, best of 20)%s\n", G, best / N * 1e6, h ? "  TIMEOUT" : "");
  }
  return 0;
}
