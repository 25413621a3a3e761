// Developer probe: issue cost of the vector instructions the requantisation epilogues are made of, per SIMD, with 1, 2 and 4 waves per SIMD.
// hipcc -O3 --offload-arch=gfx950 tools/probes/valu_rate.hip -o tools/probes/bin/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define REP8(s) s s s s s s s s
#define REP64(s) REP8(REP8(s))

template <int OP>
__global__ void k(unsigned long long* out, float* sink, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
  int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3;
  unsigned u0 = 0, u1 = 0;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
    if (OP == 0) { REP8(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));) }
    if (OP == 1) { REP8(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p0));) }
    if (OP == 2) { REP8(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p0));) }
    if (OP == 3) { REP8(asm volatile("v_cvt_f32_i32 %0, %4\n v_cvt_f32_i32 %1, %5\n v_cvt_f32_i32 %2, %6\n v_cvt_f32_i32 %3, %7\n v_cvt_f32_i32 %0, %4\n v_cvt_f32_i32 %1, %5\n v_cvt_f32_i32 %2, %6\n v_cvt_f32_i32 %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(i0), "v"(i1), "v"(i2), "v"(i3));) }
    if (OP == 4) { REP8(asm volatile("v_cvt_pk_u8_f32 %0, %2, 0, %0\n v_cvt_pk_u8_f32 %1, %3, 1, %1\n v_cvt_pk_u8_f32 %0, %4, 2, %0\n v_cvt_pk_u8_f32 %1, %5, 3, %1\n v_cvt_pk_u8_f32 %0, %2, 0, %0\n v_cvt_pk_u8_f32 %1, %3, 1, %1\n v_cvt_pk_u8_f32 %0, %4, 2, %0\n v_cvt_pk_u8_f32 %1, %5, 3, %1" : "+v"(u0), "+v"(u1) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));) }
    if (OP == 5) { REP8(asm volatile("v_rndne_f32 %0, %0\n v_rndne_f32 %1, %1\n v_rndne_f32 %2, %2\n v_rndne_f32 %3, %3\n v_rndne_f32 %0, %0\n v_rndne_f32 %1, %1\n v_rndne_f32 %2, %2\n v_rndne_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if (OP == 6) { REP8(asm volatile("v_med3_f32 %0, %0, %4, %5\n v_med3_f32 %1, %1, %4, %5\n v_med3_f32 %2, %2, %4, %5\n v_med3_f32 %3, %3, %4, %5\n v_med3_f32 %0, %0, %4, %5\n v_med3_f32 %1, %1, %4, %5\n v_med3_f32 %2, %2, %4, %5\n v_med3_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5));) }
    if (OP == 7) { REP8(asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));) }
    if (OP == 8) { REP8(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4\n v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(i0));) }
    if (OP == 9) { REP8(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(i0));) }
    if (OP == 10) { REP8(asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1\n v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1\n v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1\n v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1" : "+v"(p0), "+v"(p1) : "v"(i0), "v"(i1) : "vcc");) }
    if (OP == 11) { REP8(asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4\n v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));) }
    if (OP == 12) { REP8(asm volatile("v_mad_i32_i24 %0, %0, %4, %0\n v_mad_i32_i24 %1, %1, %4, %1\n v_mad_i32_i24 %2, %2, %4, %2\n v_mad_i32_i24 %3, %3, %4, %3\n v_mad_i32_i24 %0, %0, %4, %0\n v_mad_i32_i24 %1, %1, %4, %1\n v_mad_i32_i24 %2, %2, %4, %2\n v_mad_i32_i24 %3, %3, %4, %3" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(i0));) }
    if (OP == 13) { REP8(asm volatile("v_bfe_i32 %0, %4, 8, 8\n v_bfe_i32 %1, %4, 8, 8\n v_bfe_i32 %2, %4, 8, 8\n v_bfe_i32 %3, %4, 8, 8\n v_bfe_i32 %0, %4, 8, 8\n v_bfe_i32 %1, %4, 8, 8\n v_bfe_i32 %2, %4, 8, 8\n v_bfe_i32 %3, %4, 8, 8" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(i0));) }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + p0.x + p1.y + p2.x + p3.y + (float)(i0 + i1 + i2 + i3) + (float)(u0 + u1);
}

template <int OP>
static int run(const char* name) {
  unsigned long long* d; float* sink;
  CK(hipMalloc(&d, 256 * 16 * 8)); CK(hipMalloc(&sink, 256 * 1024 * 4));
  const int iters = 50;
  printf("%-18s", name);
  for (int wps : {1, 2, 4}) {
    CK(hipMemset(d, 0, 256 * 16 * 8));
    k<OP><<<256, 256 * wps>>>(d, sink, iters);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(256 * 16);
    CK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
    double s = 0; int n = 0;
    for (int b = 0; b < 256; b++) for (int w = 0; w < 4 * wps; w++) { s += (double)h[b * 16 + w]; n++; }
    const double per_wave = s / n / (iters * 64.0);            // cycles per instruction as one wave sees it
    printf("  %d wave/SIMD: %5.2f cyc/instr/wave = %5.2f cyc/instr/SIMD", wps, per_wave, per_wave / wps);
  }
  printf("\n");
  return 0;
}
int main() {
  run<0>("v_add_f32"); run<7>("v_mul_f32"); run<11>("v_fma_f32"); run<1>("v_pk_add_f32"); run<2>("v_pk_mul_f32"); run<3>("v_cvt_f32_i32"); run<4>("v_cvt_pk_u8_f32");
  run<5>("v_rndne_f32"); run<6>("v_med3_f32"); run<8>("v_mul_lo_u32"); run<9>("v_add_u32"); run<10>("v_mad_u64_u32"); run<12>("v_mad_i32_i24"); run<13>("v_bfe_i32");
  return 0;
}
