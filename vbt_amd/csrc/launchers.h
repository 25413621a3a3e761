// Host-side entry points of the kernel translation units (k_*.hip).  Each kernel family is compiled in its own translation
// unit - the template instantiations of the fused MBConv / SeparableConv tiles alone are most of the library's build time -
// and the planner (detector.hip) reaches the kernels only through these launchers: plain argument structs in, one launch
// enqueued on `st` out.  A launcher returns VBT_OK or VBT_ERR_ARG (no such instantiation) and sets the error text.
#pragma once
#include "dev_common.h"

namespace vbt {
#include "fused_block.h"   // FusedArgs, MultiTiles, DwTileArgs, FB_* tile constants (+ the kernel templates)
#include "stem_block.h"    // StemBlockArgs
#include "image_block.h"   // ImageBundle, IB_*
#include "expdw_block.h"   // ExpDwArgs, XD_*
#include "expdw2_block.h"  // ExpDw2Args, XD2_*
#include "band_block.h"    // BandArgs, BD_*

// fused MBConv / SeparableConv / BiFPN node on LDS tiles (fused_block.h): which instantiation to run
struct FusedLaunch {
  int k, stride, nbp;   // depthwise kernel size / stride, 64-channel output blocks per pass
  bool expand;          // MBConv (expand -> depthwise -> project) or SeparableConv / node (depthwise -> project)
  bool mdw;             // depthwise on the matrix pipe
  bool nt3;             // 48-channel chunks
  bool ppw2;            // 128-pixel (16 x 8) tiles
  bool dw64;            // depthwise on the 16x16x64 MFMA with immediate LDS offsets (8 x 8 or 16 x 8 tiles)
  int lds_bytes;
  unsigned grid;
};
int launch_fused_block(const FusedArgs& a, const FusedLaunch& L, hipStream_t st);
int launch_fused_multi(const FusedArgs* d_args, const MultiTiles& mt, int k, int stride, int nbp, bool mdw, int lds_bytes, unsigned grid,
                       hipStream_t st);
int launch_dw_tile(const DwTileArgs& a, int k, int stride, bool mdw, dim3 grid, int lds_bytes, hipStream_t st);
// one workgroup per image (image_block.h)
int launch_mbconv_image(const FusedArgs& a, const ImageBundle& wb, int k, int stride, int maxu, int PW, int PH, int NB, int lds_bytes, int B,
                        hipStream_t st);
// SeparableConv / BiFPN node on row bands (band_block.h): one problem by value, or several problems in one grid
int launch_band_one(const BandArgs& a, unsigned grid, int lds_bytes, hipStream_t st);
int launch_band_multi(const BandArgs* d_probs, const MultiTiles& mt, int C, unsigned grid, int lds_bytes, hipStream_t st);   // C: channels of the maps (all problems alike)
// whole-image expand + depthwise (expdw_block.h)
int launch_expdw(const ExpDwArgs& a, int k, int stride, int KS64, unsigned grid, int lds_bytes, hipStream_t st);
// the same on the second form of the kernel (expdw2_block.h); nw = waves per workgroup (8 or 16; stride 2: 16), gpw = input pixel groups
// per wave (8 waves: 2, 4 or 7; 16 waves: 1, 2 or 4)
int launch_expdw2(const ExpDw2Args& a, int k, int stride, int KS64, int nw, int gpw, unsigned grid, int lds_bytes, hipStream_t st);
// network entry: stem 3x3/2 + first SeparableConv (stem_block.h)
int launch_stem_block(const StemBlockArgs& a, bool full_range, unsigned grid, hipStream_t st);

}  // namespace vbt
