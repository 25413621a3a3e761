// Fused MBConv blocks on LDS tiles (fused_block.h, EXPAND = true): expand 1x1 -> depthwise -> project 1x1 [+ residual].
#include "launchers.h"

namespace vbt {

#define FB_P2(KK, S, NBP, KSE)                                                                                      \
  do {                                                                                                              \
    if (L.nt3) fused_block_kernel<KK, S, NBP, true, true, KSE, 3, 2, true><<<grid, 256, L.lds_bytes, st>>>(a);      \
    else fused_block_kernel<KK, S, NBP, true, true, KSE, 4, 2, true><<<grid, 256, L.lds_bytes, st>>>(a);            \
  } while (0)
#define FB_P2K(KK, S)                                                                                               \
  do {                                                                                                              \
    if (L.nbp == 1 && a.KSe == 1) FB_P2(KK, S, 1, 1);                                                               \
    else if (L.nbp == 1) FB_P2(KK, S, 1, 2);                                                                        \
    else if (a.KSe == 1) FB_P2(KK, S, 2, 1);                                                                        \
    else FB_P2(KK, S, 2, 2);                                                                                        \
  } while (0)
#define FB_DW64(KK, S, NBP)                                                                                                                    \
  do {                                                                                                                                         \
    if (L.nt3 && a.KSe == 1) fused_block_kernel<KK, S, (NBP <= 2 ? NBP : 1), true, true, 1, 3, 1, true><<<grid, 256, L.lds_bytes, st>>>(a);    \
    else if (L.nt3) fused_block_kernel<KK, S, (NBP <= 2 ? NBP : 1), true, true, 2, 3, 1, true><<<grid, 256, L.lds_bytes, st>>>(a);            \
    else if (a.KSe == 1) fused_block_kernel<KK, S, (NBP <= 2 ? NBP : 1), true, true, 1, 4, 1, true><<<grid, 256, L.lds_bytes, st>>>(a);       \
    else fused_block_kernel<KK, S, (NBP <= 2 ? NBP : 1), true, true, 2, 4, 1, true><<<grid, 256, L.lds_bytes, st>>>(a);                       \
  } while (0)
#define FB_LAUNCH(KK, S, NBP)                                                                                                                          \
  do {                                                                                                                                                 \
    if (L.dw64) FB_DW64(KK, S, NBP);                                                                                                                   \
    else if (L.mdw && L.nt3 && NBP <= 2 && a.KSe == 1) fused_block_kernel<KK, S, (NBP <= 2 ? NBP : 1), true, true, 1, 3><<<grid, 256, L.lds_bytes, st>>>(a); \
    else if (L.mdw && L.nt3 && NBP <= 2 && a.KSe == 2) fused_block_kernel<KK, S, (NBP <= 2 ? NBP : 1), true, true, 2, 3><<<grid, 256, L.lds_bytes, st>>>(a); \
    else if (L.mdw && L.nt3 && NBP <= 2 && a.KSe == 3) fused_block_kernel<KK, S, (NBP <= 2 ? NBP : 1), true, true, 3, 3><<<grid, 256, L.lds_bytes, st>>>(a); \
    else if (L.mdw && L.nt3 && NBP <= 2 && a.KSe == 4) fused_block_kernel<KK, S, (NBP <= 2 ? NBP : 1), true, true, 4, 3><<<grid, 256, L.lds_bytes, st>>>(a); \
    else if (L.mdw && a.KSe == 1) fused_block_kernel<KK, S, NBP, true, true, 1><<<grid, 256, L.lds_bytes, st>>>(a);                                   \
    else if (L.mdw && a.KSe == 2) fused_block_kernel<KK, S, NBP, true, true, 2><<<grid, 256, L.lds_bytes, st>>>(a);                                   \
    else if (L.mdw && a.KSe == 3) fused_block_kernel<KK, S, NBP, true, true, 3><<<grid, 256, L.lds_bytes, st>>>(a);                                   \
    else if (L.mdw && a.KSe == 4) fused_block_kernel<KK, S, NBP, true, true, 4><<<grid, 256, L.lds_bytes, st>>>(a);                                   \
    else if (L.mdw) fused_block_kernel<KK, S, NBP, true, true><<<grid, 256, L.lds_bytes, st>>>(a);                                                    \
    else fused_block_kernel<KK, S, NBP, true, false><<<grid, 256, L.lds_bytes, st>>>(a);                                                              \
  } while (0)
#define FB_NBP(KK, S)                                      \
  do {                                                     \
    switch (L.nbp) {                                       \
      case 1: FB_LAUNCH(KK, S, 1); break;                  \
      case 2: FB_LAUNCH(KK, S, 2); break;                  \
      case 3: FB_LAUNCH(KK, S, 3); break;                  \
      default: FB_LAUNCH(KK, S, 5); break;                 \
    }                                                      \
  } while (0)

int launch_fused_mbconv(const FusedArgs& a, const FusedLaunch& L, hipStream_t st) {
  const dim3 grid(L.grid);
  if (L.ppw2) {
    if (L.k == 3 && L.stride == 1) FB_P2K(3, 1);
    else if (L.k == 3 && L.stride == 2) FB_P2K(3, 2);
    else if (L.k == 5 && L.stride == 1) FB_P2K(5, 1);
    else FB_P2K(5, 2);
    return VBT_OK;
  }
  if (L.k == 3 && L.stride == 1) FB_NBP(3, 1);
  else if (L.k == 3 && L.stride == 2) FB_NBP(3, 2);
  else if (L.k == 5 && L.stride == 1) FB_NBP(5, 1);
  else FB_NBP(5, 2);
  return VBT_OK;
}

}  // namespace vbt
