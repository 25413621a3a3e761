// SeparableConv / BiFPN node on LDS tiles (fused_block.h, EXPAND = false), the head layers of all levels in one grid
// (fused_block_multi_kernel) and the stand-alone LDS-tiled depthwise conv (dw_tile_kernel).
#include "launchers.h"

namespace vbt {

int launch_fused_mbconv(const FusedArgs& a, const FusedLaunch& L, hipStream_t st);   // k_fused_mbconv.hip

#define FB_LAUNCH(KK, S, NBP)                                                                              \
  do {                                                                                                     \
    if (L.mdw) fused_block_kernel<KK, S, NBP, false, true><<<grid, 256, L.lds_bytes, st>>>(a);             \
    else fused_block_kernel<KK, S, NBP, false, false><<<grid, 256, L.lds_bytes, st>>>(a);                  \
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

int launch_fused_block(const FusedArgs& a, const FusedLaunch& L, hipStream_t st) {
  if (L.expand) return launch_fused_mbconv(a, L, st);
  const dim3 grid(L.grid);
  if (L.k == 3 && L.stride == 1) FB_NBP(3, 1);
  else if (L.k == 3 && L.stride == 2) FB_NBP(3, 2);
  else if (L.k == 5 && L.stride == 1) FB_NBP(5, 1);
  else FB_NBP(5, 2);
  return VBT_OK;
}

int launch_fused_multi(const FusedArgs* d_args, const MultiTiles& mt, int k, int stride, int nbp, bool mdw, int lds_bytes, unsigned grid_x,
                       hipStream_t st) {
  const dim3 grid(grid_x);
  if (k == 3 && stride == 1 && nbp == 1) {
    if (mdw) fused_block_multi_kernel<3, 1, 1, false, true><<<grid, 256, lds_bytes, st>>>(d_args, mt);
    else fused_block_multi_kernel<3, 1, 1, false, false><<<grid, 256, lds_bytes, st>>>(d_args, mt);
  } else if (k == 3 && stride == 1 && nbp == 2) {  // BiFPN width 65..128 (Lite1 / Lite2)
    if (mdw) fused_block_multi_kernel<3, 1, 2, false, true><<<grid, 256, lds_bytes, st>>>(d_args, mt);
    else fused_block_multi_kernel<3, 1, 2, false, false><<<grid, 256, lds_bytes, st>>>(d_args, mt);
  } else {
    set_error("fused_heads_multi: unsupported instantiation (k=%d s=%d nbp=%d)", k, stride, nbp);
    return VBT_ERR_ARG;
  }
  return VBT_OK;
}

int launch_dw_tile(const DwTileArgs& a, int k, int stride, bool mdw, dim3 grid, int lds_bytes, hipStream_t st) {
#define DW_LAUNCH(KK, S)                                                        \
  do {                                                                          \
    if (mdw) dw_tile_kernel<KK, S, true><<<grid, 256, lds_bytes, st>>>(a);      \
    else dw_tile_kernel<KK, S, false><<<grid, 256, lds_bytes, st>>>(a);         \
  } while (0)
  if (k == 3 && stride == 1) DW_LAUNCH(3, 1);
  else if (k == 3 && stride == 2) DW_LAUNCH(3, 2);
  else if (k == 5 && stride == 1) DW_LAUNCH(5, 1);
  else DW_LAUNCH(5, 2);
#undef DW_LAUNCH
  return VBT_OK;
}

}  // namespace vbt
