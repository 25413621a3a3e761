// SeparableConv (depthwise 3x3 / 1 SAME -> pointwise) on ROW BANDS of a C-channel map (C = the BiFPN width: 64 / 88 / 112 for
// EfficientDet-Lite0 / 1 / 2): the BiFPN nodes and the box / class head layers.  Optional BiFPN node input: the band is the sum (+ReLU6) of two or
// three resampled sources (binary integer ADDs, node_sum4 of fused_block.h).
//
// Why not the 64-pixel tile kernel (fused_block.h) here: on these layers a tile does ~120 wave-instructions of
// requantisation but ~550 in all - tile decode, halo addressing, parameter loads, the epilogue's addressing are paid per 64
// pixels, and every tile re-reads a 1-pixel halo.  A band is `rows` full-width rows of one image (up to 320 pixels): the
// per-workgroup costs are paid once per band, the halo is two rows, the projection weights / biases are staged in LDS
// once, and the projection is dealt in (pixel group, 16-channel tile) units, so a 9- or 36-channel head output costs a
// quarter / three quarters of a 64-channel one instead of the same.
// LDS:  T0 [(rows+2)*(W+2)][CS]  input band with its 1-pixel border (zero point outside the image); CS = odd multiple of 16 >= C
//       D  [rows*W (16-padded)][CS]  depthwise output
//       WP [NT][KS][64] x 16 B  projection weights, natural channel order | bias int[16 NT] | mult float[16 NT]
// 16 wavefronts; both stages on the 16x16x64 int8 MFMA (depthwise as a diagonal-embedded matrix product, four taps per
// instruction: three instructions for 3x3).  Arithmetic identical to the per-op kernels.
#pragma once

#ifndef VBT_BD_WAVES
#define VBT_BD_WAVES 16
#endif
constexpr int BD_WAVES = VBT_BD_WAVES, BD_THREADS = 64 * BD_WAVES;
constexpr int BD_WP_TAIL = 1024;   // bias (512 B) | multipliers (512 B) behind the projection weights in LDS

struct BandArgs {
  const int8_t* x;   // [B][H][W][C] (plain input; unused when n_src > 0)
  int8_t* out;       // [B][H][W][Cout]
  int H, W, Cout, rows, nbands;
  int C, CS;         // input channels (% 8 == 0) / bytes per pixel row of T0 and D: odd multiple of 16 >= C
  int NCG, KS;       // 16-channel groups of the depthwise: ceil(C / 16) / projection K-steps of 64: ceil(C / 64)
  unsigned zx4;      // zero point of the depthwise input x4
  const v4i* wd;     // [cg][m][lane] x 16 B: row i = channel 16cg + i, k = 16g + j -> tap 4m + g, diagonal j == i
  const int* bd;     // bias with the input zero point folded (16 NCG)
  const float* md;
  Rq rqd;
  const v4i* wp;     // [t][ks][lane] x 16 B: row i = output channel 16t + i, k = 64ks + 16g + j
  const int* bp;     // bias with the depthwise output's zero point folded, padded to 64-channel blocks
  const float* mp;
  Rq rqp;
  // BiFPN node: see FusedArgs
  int n_src, chain;
  const int8_t* src[3];
  int sh[3], sw[3], smode[3], spt[3], spl[3];
  AddQ sumq, preq;
};

__device__ __forceinline__ unsigned band_source4(const BandArgs& a, int j, long b, int iy, int ix, int cd, bool up2) {
  const int C = a.C;
  const int8_t* sb = a.src[j] + b * (long)a.sh[j] * a.sw[j] * C + 4 * cd;
  if (a.smode[j] == 0) return *(const unsigned*)(sb + (iy * a.sw[j] + ix) * C);
  if (a.smode[j] == 1) {
    int yy, xx;
    if (up2) { yy = iy >> 1; xx = ix >> 1; }
    else { yy = (iy * a.sh[j]) / a.H; xx = (ix * a.sw[j]) / a.W; }
    return *(const unsigned*)(sb + (yy * a.sw[j] + xx) * C);
  }
  unsigned lo = 0u, hi = 0u;   // 3x3/2 max pool read in place on the u8 image of the bytes (out-of-map taps = -128)
#pragma unroll
  for (int ky = 0; ky < 3; ky++) {
    const int yy = iy * 2 + ky - a.spt[j], yc = min(max(yy, 0), a.sh[j] - 1);
#pragma unroll
    for (int kx = 0; kx < 3; kx++) {
      const int xx = ix * 2 + kx - a.spl[j], xc = min(max(xx, 0), a.sw[j] - 1);
      unsigned t = *(const unsigned*)(sb + (yc * a.sw[j] + xc) * C);
      t = (yy == yc && xx == xc) ? (t ^ 0x80808080u) : 0u;
      lo = pk_max_u16(lo, t & 0x00FF00FFu);
      hi = pk_max_u16(hi, (t >> 8) & 0x00FF00FFu);
    }
  }
  return (lo | (hi << 8)) ^ 0x80808080u;
}

// NW waves per workgroup: 16 for a BiFPN node (one or two workgroups per image: the per-wave chain of units must be short), 8 for
// the head layers (1280+ bands per launch: 16-wave workgroups fill every wave slot of a CU with two of them, so a third forward
// in flight cannot co-reside; 8 waves on bands of <= 240 pixels interleave twice as many phases - +1.3 % end to end).
// C64: the map has 64 channels (Lite0): row stride, channel groups and K-steps are compile-time constants (the generic
// form costs the Lite0 pipeline 3 % end to end: 95.0 k vs 98.0 k frames/s).
template <int NW, bool C64>
__device__ __forceinline__ void sepconv_band_body(const BandArgs& a, int local, unsigned char* bd_smem) {
  constexpr int nwaves = NW, nthreads = 64 * NW;
  const long b = fdiv_small(local, frcp(a.nbands));
  const int band = local - (int)b * a.nbands;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 15, g = lane >> 4;
  const int y0 = band * a.rows, nr = min(a.rows, a.H - y0);
  const int PW = a.W + 2, NPh = (nr + 2) * PW, NPo = nr * a.W, NPG = (NPo + 15) >> 4;
  const int C = C64 ? 64 : a.C, CS = C64 ? 80 : a.CS;
  unsigned char* T0 = bd_smem;
  unsigned char* D = T0 + (a.rows + 2) * PW * CS;
  unsigned char* WP = D + (((a.rows * a.W + 15) >> 4) << 4) * CS;   // NT x KS KB of weights | 512 B bias | 512 B mult
  const int NT = (a.Cout + 15) >> 4, KS = C64 ? 1 : a.KS, NCG = C64 ? 4 : a.NCG;
  unsigned char* WB = WP + NT * KS * 1024;

  // ---- stage L: band + border -> T0; projection weights / bias / multipliers -> LDS ----
  for (int i = tid; i < NT * KS * 64; i += nthreads) *(v4i*)(WP + 16 * i) = a.wp[i];
  if (tid < 4 * NT) *(uint4*)(WB + 16 * tid) = *(const uint4*)((const unsigned char*)a.bp + 16 * tid);
  else if (tid >= 32 && tid < 32 + 4 * NT) *(uint4*)(WB + 512 + 16 * (tid - 32)) = *(const uint4*)((const unsigned char*)a.mp + 16 * (tid - 32));
  const float rcp_pw = frcp(PW);
  if (a.n_src > 0) {
    const bool up2[3] = {a.H == 2 * a.sh[0] && a.W == 2 * a.sw[0], a.H == 2 * a.sh[1] && a.W == 2 * a.sw[1], a.H == 2 * a.sh[2] && a.W == 2 * a.sw[2]};
    const int ndp = C64 ? 16 : CS >> 2, nd = C >> 2;   // dwords per pixel visited (Lite0: the 64 real channels only) / of real channels
    const float rcp_ndp = frcp(ndp);
    for (int i = tid; i < NPh * ndp; i += nthreads) {   // 4 channels per lane-iteration
      const int p = C64 ? i >> 4 : fdiv_small(i, rcp_ndp), cd = i - p * ndp;
      const int hy = fdiv_small(p, rcp_pw), hx = p - hy * PW;
      const int iy = y0 + hy - 1, ix = hx - 1;
      unsigned v = a.zx4;
      if (cd < nd && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) {
        unsigned us[3] = {0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < 3; j++)
          if (j < a.n_src) us[j] = band_source4(a, j, b, iy, ix, cd, up2[j]);
        if (a.chain == 0) v = addq4(us[0], us[1], a.sumq);
        else {
          const unsigned pp = addq4(us[0], us[1], a.preq);
          v = a.chain == 1 ? addq4(pp, us[2], a.sumq) : addq4(us[2], pp, a.sumq);
        }
      }
      *(unsigned*)(T0 + p * CS + 4 * cd) = v;
    }
  } else {
    const int8_t* xb = a.x + b * (long)a.H * a.W * C;
    if constexpr (C64) {
      const uint4 z4 = make_uint4(a.zx4, a.zx4, a.zx4, a.zx4);
      for (int i = tid; i < NPh * 4; i += nthreads) {    // 16 bytes per lane-iteration
        const int p = i >> 2, sg = i & 3;
        const int hy = fdiv_small(p, rcp_pw), hx = p - hy * PW;
        const int iy = y0 + hy - 1, ix = hx - 1;
        uint4 v = z4;
        if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) v = *(const uint4*)(xb + (iy * a.W + ix) * 64 + 16 * sg);
        *(uint4*)(T0 + p * 80 + 16 * sg) = v;
      }
    } else {
    const int ngp = CS >> 3, ng = C >> 3;              // 8-byte granules per LDS row / of real channels
    const float rcp_ngp = frcp(ngp);
    const uint2 z2 = make_uint2(a.zx4, a.zx4);
    for (int i = tid; i < NPh * ngp; i += nthreads) {
      const int p = fdiv_small(i, rcp_ngp), sg = i - p * ngp;
      const int hy = fdiv_small(p, rcp_pw), hx = p - hy * PW;
      const int iy = y0 + hy - 1, ix = hx - 1;
      uint2 v = z2;
      if (sg < ng && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) v = *(const uint2*)(xb + (iy * a.W + ix) * C + 8 * sg);
      *(uint2*)(T0 + p * CS + 8 * sg) = v;
    }
    }
  }
  // depthwise operands of this wave's channel group (requested before the barrier): wave w owns group w % NCG and, of its
  // pixel groups, every (NW / NCG)-th one; waves beyond NCG * (NW / NCG) sit the stage out (7 groups on 8 / 16 waves: one / two)
  const int cg = wave % NCG, sub = wave / NCG, nsub = nwaves / NCG;
  v4i wdv[3];
#pragma unroll
  for (int m = 0; m < 3; m++) wdv[m] = a.wd[(cg * 3 + m) * 64 + lane];
  const int4 bq = *(const int4*)(a.bd + 16 * cg + 4 * g);
  const float4 mu = *(const float4*)(a.md + 16 * cg + 4 * g);
  int tapoff[3];
#pragma unroll
  for (int m = 0; m < 3; m++) {
    const int tap = min(4 * m + g, 8);
    tapoff[m] = ((tap / 3) * PW + (tap % 3)) * CS;
  }
  const float rcp_w = frcp(a.W);
  __syncthreads();
  // ---- stage D: depthwise; unit = (output pixel group, channel group cg) ----
  if (sub < nsub)
    for (int pg = sub; pg < NPG; pg += nsub) {
      const int slot = pg * 16 + r, sc = min(slot, NPo - 1);
      const int py = fdiv_small(sc, rcp_w), px = sc - py * a.W;
      const unsigned char* pb = T0 + (py * PW + px) * CS + 16 * cg;
      v4i acc = v4i_from(bq);
#pragma unroll
      for (int m = 0; m < 3; m++) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(wdv[m], *(const v4i*)(pb + tapoff[m]), acc, 0, 0, 0);
      *(unsigned*)(D + slot * CS + 16 * cg + 4 * g) = rq_pack_b(acc, mu, a.rqd);
    }
  __syncthreads();
  // ---- stage P: projection; unit = (pixel group, 16-channel output tile) ----
  const int NU = NPG * NT;
  const float rcp_nt = frcp(NT);
  auto store_unit = [&](int pg, int t, const v4i& acc, const float4& mm) {
    const int slot = pg * 16 + r;
    const int c0 = 16 * t + 4 * g;
    const unsigned d = rq_pack_b(acc, mm, a.rqp);
    if (slot < NPo && c0 < a.Cout) {
      int8_t* o = a.out + ((b * a.H + y0) * (long)a.W + slot) * a.Cout + c0;   // the band's pixels are contiguous: (y0 + py) * W + px = y0 * W + slot
      if ((a.Cout & 3) == 0) *(unsigned*)o = d;
      else
        for (int j = 0; j < 4; j++)
          if (c0 + j < a.Cout) o[j] = (int8_t)(d >> (8 * j));
    }
  };
  if (NT == 4 && KS == 1) {
    // 64 -> 64 channels (every Lite0 layer but the heads' last): unit u = wave + 16 i is tile t = wave & 3 of pixel group
    // (wave >> 2) + 4 i, so the wave's weights / bias / multipliers are loop invariants
    const int t = wave & 3, c0 = 16 * t + 4 * g;
    const v4i wv = *(const v4i*)(WP + (t * 64 + lane) * 16);
    const int4 bb = *(const int4*)(WB + 4 * c0);
    const float4 mm = *(const float4*)(WB + 512 + 4 * c0);
    for (int pg = wave >> 2; pg < NPG; pg += nwaves / 4) {
      v4i acc = v4i_from(bb);
      acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(wv, *(const v4i*)(D + (pg * 16 + r) * CS + 16 * g), acc, 0, 0, 0);
      store_unit(pg, t, acc, mm);
    }
  } else {
    for (int u = wave; u < NU; u += nwaves) {
      const int pg = fdiv_small(u, rcp_nt), t = u - pg * NT;
      const int c0 = 16 * t + 4 * g;
      v4i acc = v4i_from(*(const int4*)(WB + 4 * c0));
      const unsigned char* drow = D + (pg * 16 + r) * CS + 16 * g;
      for (int ks = 0; ks < KS; ks++)
        acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(*(const v4i*)(WP + ((t * KS + ks) * 64 + lane) * 16), *(const v4i*)(drow + 64 * ks), acc, 0, 0, 0);
      store_unit(pg, t, acc, *(const float4*)(WB + 512 + 4 * c0));
    }
  }
}

// Several problems in one grid (the same head layer on all pyramid levels of both heads): the problem list lives in HBM.
#ifndef VBT_BD_HEAD_WAVES
#define VBT_BD_HEAD_WAVES 8
#endif
#ifndef VBT_BD_HEAD_MAXPX
#define VBT_BD_HEAD_MAXPX 240
#endif
constexpr int BD_HEAD_WAVES = VBT_BD_HEAD_WAVES, BD_HEAD_MAXPX = VBT_BD_HEAD_MAXPX;
#ifdef VBT_DEFINE_BAND_KERNELS   // the entry points are not templates: exactly one translation unit (k_band.hip) defines them
__device__ __forceinline__ int band_problem(const MultiTiles& mt) {
  int pi = 0;
#pragma unroll
  for (int i = 1; i < 12; i++)
    if (i < mt.n && (int)blockIdx.x >= mt.start[i]) pi = i;
  return pi;
}
__global__ __launch_bounds__(64 * BD_HEAD_WAVES) void sepconv_band_kernel(const BandArgs* __restrict__ probs, MultiTiles mt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char bd_smem_dyn[];
  const int pi = band_problem(mt);
  sepconv_band_body<BD_HEAD_WAVES, true>(probs[pi], (int)blockIdx.x - mt.start[pi], bd_smem_dyn);
}
__global__ __launch_bounds__(64 * BD_HEAD_WAVES) void sepconv_band_wide_kernel(const BandArgs* __restrict__ probs, MultiTiles mt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char bd_smem_dyn[];
  const int pi = band_problem(mt);
  sepconv_band_body<BD_HEAD_WAVES, false>(probs[pi], (int)blockIdx.x - mt.start[pi], bd_smem_dyn);
}
// One problem (a BiFPN node): the arguments travel in the kernel-argument segment, one dependent memory round trip
// less at the head of a kernel that is a chain of round trips.
__global__ __launch_bounds__(BD_THREADS) void sepconv_band_one_kernel(BandArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char bd_smem_dyn[];
  sepconv_band_body<BD_WAVES, true>(a, (int)blockIdx.x, bd_smem_dyn);
}
__global__ __launch_bounds__(BD_THREADS) void sepconv_band_one_wide_kernel(BandArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char bd_smem_dyn[];
  sepconv_band_body<BD_WAVES, false>(a, (int)blockIdx.x, bd_smem_dyn);
}
#endif  // VBT_DEFINE_BAND_KERNELS
