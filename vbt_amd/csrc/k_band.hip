// SeparableConv / BiFPN node / head layer on row bands (band_block.h), whole-image expand + depthwise of the low-resolution
// MBConv blocks (expdw_block.h) and the network entry (stem_block.h).
#define VBT_DEFINE_BAND_KERNELS 1
#include "launchers.h"

namespace vbt {

static void band_attrs() {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sepconv_band_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sepconv_band_one_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sepconv_band_wide_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sepconv_band_one_wide_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
}

int launch_band_one(const BandArgs& a, unsigned grid, int lds_bytes, hipStream_t st) {
  band_attrs();
  if (a.C == 64) sepconv_band_one_kernel<<<dim3(grid), BD_THREADS, lds_bytes, st>>>(a);
  else sepconv_band_one_wide_kernel<<<dim3(grid), BD_THREADS, lds_bytes, st>>>(a);
  return VBT_OK;
}

int launch_band_multi(const BandArgs* d_probs, const MultiTiles& mt, int C, unsigned grid, int lds_bytes, hipStream_t st) {
  band_attrs();
  if (C == 64) sepconv_band_kernel<<<dim3(grid), 64 * BD_HEAD_WAVES, lds_bytes, st>>>(d_probs, mt);
  else sepconv_band_wide_kernel<<<dim3(grid), 64 * BD_HEAD_WAVES, lds_bytes, st>>>(d_probs, mt);
  return VBT_OK;
}

int launch_expdw(const ExpDwArgs& a, int k, int stride, int KS64, unsigned grid_x, int lds_bytes, hipStream_t st) {
  const dim3 grid(grid_x);
#define XD_LAUNCH(KK, S)                                                                                             \
  do {                                                                                                               \
    static bool attr_set = false;                                                                                    \
    if (!attr_set) {                                                                                                 \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&expdw_image_kernel<KK, S, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&expdw_image_kernel<KK, S, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&expdw_image_kernel<KK, S, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      attr_set = true;                                                                                               \
    }                                                                                                                \
    if (KS64 == 2) expdw_image_kernel<KK, S, 2><<<grid, XD_THREADS, lds_bytes, st>>>(a);                             \
    else if (KS64 == 3) expdw_image_kernel<KK, S, 3><<<grid, XD_THREADS, lds_bytes, st>>>(a);                        \
    else expdw_image_kernel<KK, S, 4><<<grid, XD_THREADS, lds_bytes, st>>>(a);                                       \
  } while (0)
  if (k == 3 && stride == 1) XD_LAUNCH(3, 1);
  else if (k == 5 && stride == 1) XD_LAUNCH(5, 1);
  else XD_LAUNCH(5, 2);
#undef XD_LAUNCH
  return VBT_OK;
}

int launch_stem_block(const StemBlockArgs& a, bool full_range, unsigned grid, hipStream_t st) {
  if (full_range) stem_block_kernel<true><<<dim3(grid), 256, 0, st>>>(a);
  else stem_block_kernel<false><<<dim3(grid), 256, 0, st>>>(a);
  return VBT_OK;
}

}  // namespace vbt
