// SeparableConv / BiFPN node / head layer on row bands (band_block.h), whole-image expand + depthwise of the low-resolution
// MBConv blocks (expdw_block.h) and the network entry (stem_block.h).
#define VBT_DEFINE_BAND_KERNELS 1
#include "launchers.h"

namespace vbt {

static int band_attrs() {
  VBT_LDS_OPT_IN(sepconv_band_kernel);
  VBT_LDS_OPT_IN(sepconv_band_one_kernel);
  VBT_LDS_OPT_IN(sepconv_band_wide_kernel);
  VBT_LDS_OPT_IN(sepconv_band_one_wide_kernel);
  VBT_LDS_OPT_IN(sepconv_band_c112_kernel);
  VBT_LDS_OPT_IN(sepconv_band_one_c112_kernel);
  return VBT_OK;
}

int launch_band_one(const BandArgs& a, unsigned grid, int lds_bytes, hipStream_t st) {
  if (band_attrs()) return VBT_ERR_HIP;
  static const bool generic = getenv("VBT_BAND_GENERIC") != nullptr;   // (A/B knob: the any-width kernel on 112-channel maps)
  if (a.C == 64) sepconv_band_one_kernel<<<dim3(grid), BD_THREADS, lds_bytes, st>>>(a);
  else if (a.C == 112 && a.CS == 112 && !generic) sepconv_band_one_c112_kernel<<<dim3(grid), BD_THREADS, lds_bytes, st>>>(a);
  else sepconv_band_one_wide_kernel<<<dim3(grid), BD_THREADS, lds_bytes, st>>>(a);
  return VBT_OK;
}

int launch_band_multi(const BandArgs* d_probs, const MultiTiles& mt, int C, unsigned grid, int lds_bytes, hipStream_t st) {
  if (band_attrs()) return VBT_ERR_HIP;
  static const bool generic = getenv("VBT_BAND_GENERIC") != nullptr;
  if (C == 64) sepconv_band_kernel<<<dim3(grid), 64 * BD_HEAD_WAVES, lds_bytes, st>>>(d_probs, mt);
  else if (C == 112 && !generic) sepconv_band_c112_kernel<<<dim3(grid), 64 * BD_HEAD_WAVES_WIDE, lds_bytes, st>>>(d_probs, mt);
  else sepconv_band_wide_kernel<<<dim3(grid), 64 * BD_HEAD_WAVES_WIDE, lds_bytes, st>>>(d_probs, mt);
  return VBT_OK;
}

int launch_expdw(const ExpDwArgs& a, int k, int stride, int KS64, unsigned grid_x, int lds_bytes, hipStream_t st) {
  const dim3 grid(grid_x);
#define XD_LAUNCH(KK, S)                                                                                             \
  do {                                                                                                               \
    VBT_LDS_OPT_IN(expdw_image_kernel<KK, S, 2>);                                                                    \
    VBT_LDS_OPT_IN(expdw_image_kernel<KK, S, 3>);                                                                    \
    VBT_LDS_OPT_IN(expdw_image_kernel<KK, S, 4>);                                                                    \
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

template <int KK, int S, int KS64, int NW, int GPW>
static int launch_expdw2_t(const ExpDw2Args& a, unsigned grid, int lds_bytes, hipStream_t st) {
  VBT_LDS_OPT_IN(expdw2_kernel<KK, S, KS64, NW, GPW>);
  expdw2_kernel<KK, S, KS64, NW, GPW><<<dim3(grid), 64 * NW, lds_bytes, st>>>(a);
  return VBT_OK;
}

int launch_expdw2(const ExpDw2Args& a, int k, int stride, int KS64, int nw, int gpw, unsigned grid, int lds_bytes, hipStream_t st) {
#define XD2_GPW(KK, KS)                                                                    \
  do {                                                                                     \
    if (stride == 2) {   /* 16 waves only */                                               \
      if (gpw == 1) { if (launch_expdw2_t<KK, 2, KS, 16, 1>(a, grid, lds_bytes, st)) return VBT_ERR_HIP; }             \
      else if (gpw == 2) { if (launch_expdw2_t<KK, 2, KS, 16, 2>(a, grid, lds_bytes, st)) return VBT_ERR_HIP; }        \
      else { if (launch_expdw2_t<KK, 2, KS, 16, 4>(a, grid, lds_bytes, st)) return VBT_ERR_HIP; }                      \
    } else if (nw == 16) {                                                                 \
      if (gpw == 1) { if (launch_expdw2_t<KK, 1, KS, 16, 1>(a, grid, lds_bytes, st)) return VBT_ERR_HIP; }             \
      else if (gpw == 2) { if (launch_expdw2_t<KK, 1, KS, 16, 2>(a, grid, lds_bytes, st)) return VBT_ERR_HIP; }        \
      else { if (launch_expdw2_t<KK, 1, KS, 16, 4>(a, grid, lds_bytes, st)) return VBT_ERR_HIP; }                      \
    } else {                                                                               \
      if (gpw == 2) { if (launch_expdw2_t<KK, 1, KS, 8, 2>(a, grid, lds_bytes, st)) return VBT_ERR_HIP; }              \
      else if (gpw == 4) { if (launch_expdw2_t<KK, 1, KS, 8, 4>(a, grid, lds_bytes, st)) return VBT_ERR_HIP; }         \
      else { if (launch_expdw2_t<KK, 1, KS, 8, 7>(a, grid, lds_bytes, st)) return VBT_ERR_HIP; }                       \
    }                                                                                      \
  } while (0)
#define XD2_KS(KK)                  \
  do {                              \
    if (KS64 == 2) XD2_GPW(KK, 2);  \
    else if (KS64 == 3) XD2_GPW(KK, 3); \
    else XD2_GPW(KK, 4);            \
  } while (0)
  const bool gpw_ok = nw == 16 ? (gpw == 1 || gpw == 2 || gpw == 4) : (gpw == 2 || gpw == 4 || gpw == 7);
  if ((k != 3 && k != 5) || (stride != 1 && stride != 2) || (stride == 2 && nw != 16) || KS64 < 2 || KS64 > 4 || (nw != 8 && nw != 16) || !gpw_ok) {
    set_error("expdw2: no kernel for k %d, stride %d, %d K steps, %d waves, %d groups per wave", k, stride, KS64, nw, gpw);
    return VBT_ERR_ARG;
  }
  if (k == 3) XD2_KS(3);
  else XD2_KS(5);
#undef XD2_KS
#undef XD2_GPW
  return VBT_OK;
}

int launch_stem_block(const StemBlockArgs& a, bool full_range, unsigned grid, hipStream_t st) {
  if (full_range && a.rqs.kb && a.rqd.kb && a.rqp.kb) stem_block_kernel<3><<<dim3(grid), 256, 0, st>>>(a);
  else if (full_range) stem_block_kernel<1><<<dim3(grid), 256, 0, st>>>(a);
  else stem_block_kernel<0><<<dim3(grid), 256, 0, st>>>(a);
  return VBT_OK;
}

}  // namespace vbt
