// Developer probe: the expand + depthwise kernel (vbt_amd/csrc/expdw_block.h) alone on synthetic operands, with s_memtime stamps at
// its stage boundaries.  hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -DVBT_XD_PROF tools/probes/xd_probe.hip -o /tmp/xd_probe
#include <hip/hip_runtime.h>
#ifndef XD2_PAD
#define XD2_PAD 80
#endif
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "../../vbt_amd/csrc/dev_common.h"
namespace vbt {
void set_error(const char*, ...) {}
#include "../../vbt_amd/csrc/expdw2_block.h"
}
using namespace vbt;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int KK, int KS64, int GPW>
static void run2(const char* name, int H, int Cin, int Ce, int cpw, int B, int eys_pad) {
  ExpDw2Args a{};
  const int W = H, OH = H, OW = H, KT2 = (KK + 1) / 2;
  a.H = H; a.W = W; a.Cin = Cin; a.OH = OH; a.OW = OW; a.Ce = Ce;
  a.pad_t = a.pad_l = (KK - 1) / 2;
  a.nchunks = (Ce + 63) / 64; a.cpw = cpw; a.nbands = 1; a.brows = OH;
  a.XB = (OW + 3) / 4;
  a.EQS = (4 * (OW - 1 + KK) + 15) & ~15;
  a.EYS = 16 * a.EQS + eys_pad;
  a.e_bytes = ((OH - 1 + 2 * KT2) * a.EYS + 32 + 15) & ~15;
  int ps4 = OH * OW; while ((ps4 & 31) != 2) ps4++;
  a.PS = 4 * ps4;
  a.pe_off = a.e_bytes + 16 * a.PS;
  a.pd_off = a.pe_off + 16 * (KS64 * 256 + 32);
  const int lds = a.pd_off + 16 * (KT2 * 256 + 32);
  const int ngroups = (a.nchunks + cpw - 1) / cpw, grid = B * ngroups;
  auto dalloc = [&](size_t bytes, int fill) { void* p; CK(hipMalloc(&p, bytes)); std::vector<unsigned char> h(bytes); for (auto& v : h) v = fill < 0 ? (unsigned char)(rand() & 255) : (unsigned char)fill; CK(hipMemcpy(p, h.data(), bytes, hipMemcpyHostToDevice)); return p; };
  a.x = (const int8_t*)dalloc((size_t)B * H * W * Cin + 256, -1);
  a.out = (int8_t*)dalloc((size_t)B * OH * OW * Ce + 256, 0);
  a.pe = (const v4i*)dalloc((size_t)a.nchunks * (KS64 * 256 + 32) * 16, 1);   // timing only: weights 0x01.., biases / multipliers tiny denormal-free floats
  a.pd = (const v4i*)dalloc((size_t)a.nchunks * (KT2 * 256 + 32) * 16, 1);
  a.rqe = make_rq(-128, -128, 127); a.rqd = make_rq(-128, -128, 127);
  a.zeb = 0x80808080u;
  CK(hipMalloc(&a.prof, (size_t)grid * 32 * 8)); CK(hipMemset(a.prof, 0, (size_t)grid * 32 * 8));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&expdw2_kernel<KK, KS64, GPW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int it = 0; it < 6; it++) {
    CK(hipEventRecord(e0));
    expdw2_kernel<KK, KS64, GPW><<<grid, XD2_THREADS, lds>>>(a);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
  }
  std::vector<unsigned long long> st((size_t)grid * 32);
  CK(hipMemcpy(st.data(), a.prof, st.size() * 8, hipMemcpyDeviceToHost));
  const int ns = 3 + 2 * cpw;
  std::vector<double> avg(ns, 0.0);
  int full = 0;
  for (int w = 0; w < grid; w++) {
    const unsigned long long* s = &st[(size_t)w * 32];
    if (s[ns - 1] == 0) continue;
    full++;
    for (int k = 1; k < ns; k++) avg[k] += (double)(s[k] - s[k - 1]);
  }
  printf("v2 %s H=%d Cin=%d Ce=%d k%d cpw=%d grid=%d lds=%d KB EYS=%d: %.1f us (event)\n  stage ticks (avg over %d wgs): prologue %.0f |", name, H, Cin, Ce, KK, cpw, grid, lds / 1024, a.EYS, best * 1e3, full, avg[1] / full);
  for (int c = 0; c < cpw; c++) printf(" E+O %.0f D %.0f |", avg[2 + 2 * c] / full, avg[3 + 2 * c] / full);
  printf(" O %.0f\n", avg[ns - 1] / full);
}

int main() {
  run2<3, 2, 7>("b6 ", 20, 80, 480, 2, 64, XD2_PAD);
  run2<5, 2, 7>("b9 ", 20, 112, 672, 3, 64, XD2_PAD);
  run2<5, 3, 2>("b12", 10, 192, 1152, 3, 64, XD2_PAD);
  return 0;
}
