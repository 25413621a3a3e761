// Developer probe: the row-band SeparableConv kernel (vbt_amd/csrc/band_block.h) alone on synthetic operands, with s_memtime stamps at its
// stage boundaries.  hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -DVBT_BD_PROF tools/probes/bd_probe.hip -o tools/probes/bin/bd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "../../vbt_amd/csrc/dev_common.h"
namespace vbt {
void set_error(const char*, ...) {}

#include "../../vbt_amd/csrc/fused_block.h"
#include "../../vbt_amd/csrc/band_block.h"
template <int NW, bool C64>
__global__ __launch_bounds__(64 * NW) void band_probe_kernel(BandArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  sepconv_band_body<NW, C64>(a, (int)blockIdx.x, smem);
}
}
using namespace vbt;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int NW, bool C64 = true>
static void run(const char* name, int H, int rows, int Cout, int n_src, int B, int C = 64) {
  BandArgs a{};
  a.H = H; a.W = H; a.Cout = Cout; a.rows = rows; a.nbands = (H + rows - 1) / rows;
  a.C = C; a.CS = C64 ? 80 : (((C + 15) / 16) | 1) * 16; a.NCG = (C + 15) / 16; a.KS = (C + 63) / 64; a.zx4 = 0x80808080u;
  auto dalloc = [&](size_t bytes, int fill) { void* p; CK(hipMalloc(&p, bytes)); std::vector<unsigned char> h(bytes); for (auto& v : h) v = fill < 0 ? (unsigned char)(rand() & 255) : (unsigned char)fill; CK(hipMemcpy(p, h.data(), bytes, hipMemcpyHostToDevice)); return p; };
  a.x = (const int8_t*)dalloc((size_t)B * H * H * C + 256, -1);
  a.out = (int8_t*)dalloc((size_t)B * H * H * Cout + 256, 0);
  a.wd = (const v4i*)dalloc((size_t)a.NCG * 3 * 64 * 16, 1);
  a.bd = (const int*)dalloc(128 * 4, 0); a.md = (const float*)dalloc(128 * 4, 1);
  const int NT = (Cout + 15) / 16;
  a.wp = (const v4i*)dalloc((size_t)NT * a.KS * 64 * 16, 1);
  a.bp = (const int*)dalloc(128 * 4, 0); a.mp = (const float*)dalloc(128 * 4, 1);
  a.rqd = make_rq(-128, -128, 127); a.rqp = make_rq(-128, -128, 127);
  a.n_src = n_src; a.chain = n_src == 3 ? 1 : 0;
  for (int j = 0; j < n_src; j++) { a.src[j] = a.x; a.sh[j] = H; a.sw[j] = H; a.smode[j] = 0; }
  a.sumq = AddQ{1 << 19, 1 << 20, 1 << 19, 20, -128, 127, 128}; a.preq = a.sumq;
  const int lds = (rows + 2) * (H + 2) * a.CS + (((rows * H + 15) >> 4) << 4) * a.CS + NT * a.KS * 1024 + BD_WP_TAIL;
  const int grid = B * a.nbands;
  CK(hipMalloc(&a.prof, (size_t)grid * 8 * 8)); CK(hipMemset(a.prof, 0, (size_t)grid * 8 * 8));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&band_probe_kernel<NW, C64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int it = 0; it < 6; it++) {
    CK(hipEventRecord(e0));
    band_probe_kernel<NW, C64><<<grid, 64 * NW, lds>>>(a);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
  }
  std::vector<unsigned long long> st((size_t)grid * 8);
  CK(hipMemcpy(st.data(), a.prof, st.size() * 8, hipMemcpyDeviceToHost));
  double avg[4] = {0, 0, 0, 0};
  for (int w = 0; w < grid; w++) for (int k = 1; k < 4; k++) avg[k] += (double)(st[(size_t)w * 8 + k] - st[(size_t)w * 8 + k - 1]);
  printf("%s %dx%d rows=%d Cout=%d n_src=%d waves=%d grid=%d lds=%d KB: %.1f us (event); stage ticks (avg): L %.0f | D %.0f | P %.0f\n", name, H, H, rows, Cout, n_src, NW, grid,
         lds / 1024, best * 1e3, avg[1] / grid, avg[2] / grid, avg[3] / grid);
}

int main() {
  run<8>("head 40x40", 40, 6, 64, 0, 128);    // a head layer's 40x40 level: both heads x 64 images, 240-pixel bands
  run<8>("head 20x20", 20, 10, 64, 0, 128);
  run<16>("node 20x20", 20, 10, 64, 2, 64);    // a BiFPN node, two sources, two bands per image
  run<16>("node 20x20", 20, 10, 64, 3, 64);
  run<16>("node 10x10", 10, 10, 64, 2, 64);
  run<16, false>("lite2 head 56x56", 56, 4, 112, 0, 128, 112);
  run<16, false>("lite2 head 28x28", 28, 8, 112, 0, 128, 112);
  run<16, false>("lite2 node 28x28", 28, 14, 112, 2, 64, 112);
  return 0;
}
