// One workgroup per image: whole-image MBConv blocks on low-resolution maps (image_block.h).
#include "launchers.h"

namespace vbt {

template <int KK, int S, int MAXU>
static void launch_image_t(const FusedArgs& a, const ImageBundle& wb, int PW, int PH, int NB, int lds_bytes, int B, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mbconv_image_kernel<KK, S, MAXU>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  mbconv_image_kernel<KK, S, MAXU><<<dim3((unsigned)B), IB_THREADS, lds_bytes, st>>>(a, wb, PW, PH, NB);
}

int launch_mbconv_image(const FusedArgs& a, const ImageBundle& wb, int k, int stride, int maxu, int PW, int PH, int NB, int lds_bytes, int B,
                        hipStream_t st) {
#define IB_LAUNCH(KK, S)                                                            \
  do {                                                                              \
    if (maxu == 2) launch_image_t<KK, S, 2>(a, wb, PW, PH, NB, lds_bytes, B, st);        \
    else if (maxu == 3) launch_image_t<KK, S, 3>(a, wb, PW, PH, NB, lds_bytes, B, st);   \
    else launch_image_t<KK, S, 4>(a, wb, PW, PH, NB, lds_bytes, B, st);                  \
  } while (0)
  if (k == 3 && stride == 1) IB_LAUNCH(3, 1);
  else if (k == 3 && stride == 2) IB_LAUNCH(3, 2);
  else if (k == 5 && stride == 1) IB_LAUNCH(5, 1);
  else IB_LAUNCH(5, 2);
#undef IB_LAUNCH
  return VBT_OK;
}

}  // namespace vbt
