// One workgroup per image: whole-image MBConv blocks on low-resolution maps (image_block.h).
#include "launchers.h"

namespace vbt {

template <int KK, int S, int MAXU>
static int launch_image_t(const FusedArgs& a, const ImageBundle& wb, int PW, int PH, int NB, int lds_bytes, int B, hipStream_t st) {
  VBT_LDS_OPT_IN(mbconv_image_kernel<KK, S, MAXU>);
  mbconv_image_kernel<KK, S, MAXU><<<dim3((unsigned)B), IB_THREADS, lds_bytes, st>>>(a, wb, PW, PH, NB);
  return VBT_OK;
}

int launch_mbconv_image(const FusedArgs& a, const ImageBundle& wb, int k, int stride, int maxu, int PW, int PH, int NB, int lds_bytes, int B,
                        hipStream_t st) {
#define IB_LAUNCH(KK, S)                                                            \
  do {                                                                              \
    if (maxu == 2) { if (launch_image_t<KK, S, 2>(a, wb, PW, PH, NB, lds_bytes, B, st)) return VBT_ERR_HIP; }        \
    else if (maxu == 3) { if (launch_image_t<KK, S, 3>(a, wb, PW, PH, NB, lds_bytes, B, st)) return VBT_ERR_HIP; }   \
    else { if (launch_image_t<KK, S, 4>(a, wb, PW, PH, NB, lds_bytes, B, st)) return VBT_ERR_HIP; }                  \
  } while (0)
  if (k == 3 && stride == 1) IB_LAUNCH(3, 1);
  else if (k == 3 && stride == 2) IB_LAUNCH(3, 2);
  else if (k == 5 && stride == 1) IB_LAUNCH(5, 1);
  else IB_LAUNCH(5, 2);
#undef IB_LAUNCH
  return VBT_OK;
}

}  // namespace vbt
