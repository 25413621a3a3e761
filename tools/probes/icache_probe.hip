// Developer probe: what once-executed straight-line code costs a latency-bound kernel.  A single-wave kernel executes N independent-ish
// v_add_f32 (4 bytes each) once; the same kernel back to back (instruction cache warm) against eight different kernels of the same size
// in rotation (every launch finds its code cold in the 64 KB instruction cache two CUs share), and against a loop over 64 instructions with
// the same instruction count.  hipcc -O3 --offload-arch=gfx950 tools/probes/icache_probe.hip -o tools/probes/bin/icache_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define A8(s) s s s s s s s s
#define A64(s) A8(A8(s))
#define A512(s) A8(A64(s))
#define A2048(s) A512(s) A512(s) A512(s) A512(s)
#define A8192(s) A2048(s) A2048(s) A2048(s) A2048(s)
#define ADD "v_add_f32 %0, %0, %1\n"

template <int ID, int N>
__global__ void k_line(float* out) {
  float a = threadIdx.x, b = ID + 1;
  if (N == 512) asm volatile(A512(ADD) : "+v"(a) : "v"(b));
  if (N == 2048) asm volatile(A2048(ADD) : "+v"(a) : "v"(b));
  if (N == 8192) asm volatile(A8192(ADD) : "+v"(a) : "v"(b));
  out[threadIdx.x] = a;
}
template <int N>
__global__ void k_loop(float* out) {
  float a = threadIdx.x, b = 1;
  for (int i = 0; i < N / 64; i++) asm volatile(A64(ADD) : "+v"(a) : "v"(b));
  out[threadIdx.x] = a;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <int N>
static void run(float* out, hipStream_t st) {
  const int REP = 400;
  auto launch = [&](int id) {
    switch (id & 7) {
      case 0: k_line<0, N><<<1, 64, 0, st>>>(out); break;
      case 1: k_line<1, N><<<1, 64, 0, st>>>(out); break;
      case 2: k_line<2, N><<<1, 64, 0, st>>>(out); break;
      case 3: k_line<3, N><<<1, 64, 0, st>>>(out); break;
      case 4: k_line<4, N><<<1, 64, 0, st>>>(out); break;
      case 5: k_line<5, N><<<1, 64, 0, st>>>(out); break;
      case 6: k_line<6, N><<<1, 64, 0, st>>>(out); break;
      default: k_line<7, N><<<1, 64, 0, st>>>(out); break;
    }
  };
  double res[3];
  for (int mode = 0; mode < 3; mode++) {
    for (int r = 0; r < 16; r++) { if (mode == 2) k_loop<N><<<1, 64, 0, st>>>(out); else launch(mode == 0 ? 0 : r); }
    CK(hipStreamSynchronize(st));
    const double t0 = now();
    for (int r = 0; r < REP; r++) { if (mode == 2) k_loop<N><<<1, 64, 0, st>>>(out); else launch(mode == 0 ? 0 : r); }
    CK(hipStreamSynchronize(st));
    res[mode] = (now() - t0) / REP * 1e6;
  }
  printf("%5d instructions (%3d KB): same kernel %.2f us, eight kernels in rotation %.2f us, loop over 64 instructions %.2f us  (pure issue at 4 cycles: %.2f us)\n",
         N, N * 4 / 1024, res[0], res[1], res[2], N * 4 / 2400.0);
}

int main() {
  float* out;
  CK(hipMalloc(&out, 4096));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  run<512>(out, st);
  run<2048>(out, st);
  run<8192>(out, st);
  return 0;
}
