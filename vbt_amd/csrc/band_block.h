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
#include <type_traits>

#ifndef VBT_BD_WAVES
#define VBT_BD_WAVES 16
#endif
constexpr int BD_WAVES = VBT_BD_WAVES, BD_THREADS = 64 * BD_WAVES;
constexpr int BD_LIT = 3;           // lane-iterations of the load stage whose global loads are in flight together (plain input)
constexpr int BD_LIT_NODE = 1;      // the same for a node's source loads (up to three 16-byte loads per iteration; registers: the head layers share this kernel)
constexpr int BD_LIT_NODE_WIDE = 3; // maps of more than 64 channels: LDS leaves at most four waves per SIMD, so 128 registers are free to use
constexpr int BD_WP_TAIL = 1024;   // bias (512 B) | multipliers (512 B) behind the projection weights in LDS

// Developer build (tools/probes/bd_probe.hip): s_memtime stamps of wave 0 at the stage boundaries of every workgroup.
#ifdef VBT_BD_PROF
#define BD_STAMP(k) do { if (threadIdx.x == 0) a.prof[(long)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define BD_STAMP(k) do { } while (0)
#endif

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
#ifdef VBT_BD_PROF
  unsigned long long* prof;
#endif
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

// the same for 16 channels (4 dwords) of one pixel: one address computation per source and pixel instead of one per dword
__device__ __forceinline__ uint4 band_source16(const BandArgs& a, int j, long b, int iy, int ix, int sg, bool up2) {
  const int C = a.C;
  const int8_t* sb = a.src[j] + b * (long)a.sh[j] * a.sw[j] * C + 16 * sg;
  if (a.smode[j] == 0) return *(const uint4*)(sb + (iy * a.sw[j] + ix) * C);
  if (a.smode[j] == 1) {
    int yy, xx;
    if (up2) { yy = iy >> 1; xx = ix >> 1; }
    else { yy = (iy * a.sh[j]) / a.H; xx = (ix * a.sw[j]) / a.W; }
    return *(const uint4*)(sb + (yy * a.sw[j] + xx) * C);
  }
  unsigned lo[4] = {0u, 0u, 0u, 0u}, hi[4] = {0u, 0u, 0u, 0u};   // 3x3/2 max pool on the u8 image of the bytes (out-of-map taps = -128)
#pragma unroll
  for (int ky = 0; ky < 3; ky++) {
    const int yy = iy * 2 + ky - a.spt[j], yc = min(max(yy, 0), a.sh[j] - 1);
#pragma unroll
    for (int kx = 0; kx < 3; kx++) {
      const int xx = ix * 2 + kx - a.spl[j], xc = min(max(xx, 0), a.sw[j] - 1);
      const uint4 t4 = *(const uint4*)(sb + (yc * a.sw[j] + xc) * C);
      const bool in = yy == yc && xx == xc;
      const unsigned tt[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
      for (int d = 0; d < 4; d++) {
        const unsigned t = in ? (tt[d] ^ 0x80808080u) : 0u;
        lo[d] = pk_max_u16(lo[d], t & 0x00FF00FFu);
        hi[d] = pk_max_u16(hi[d], (t >> 8) & 0x00FF00FFu);
      }
    }
  }
  return make_uint4((lo[0] | (hi[0] << 8)) ^ 0x80808080u, (lo[1] | (hi[1] << 8)) ^ 0x80808080u, (lo[2] | (hi[2] << 8)) ^ 0x80808080u,
                    (lo[3] | (hi[3] << 8)) ^ 0x80808080u);
}

// NW waves per workgroup: 16 for a BiFPN node (one or two workgroups per image: the per-wave chain of units must be short), 8 for
// the head layers (1280+ bands per launch: 16-wave workgroups fill every wave slot of a CU with two of them, so a third forward
// in flight cannot co-reside; 8 waves on bands of <= 240 pixels interleave twice as many phases - +1.3 % end to end).
// CT: channels of the map as a compile-time constant - 64 (Lite0) or 112 (Lite2); 0 = any width (C % 8 == 0) from the arguments.  With CT the
// row stride, channel groups, K-steps and 16-byte pieces per pixel are constants (the generic form costs the Lite0 pipeline 3 % end to end:
// 95.0 k vs 98.0 k frames/s; Lite2's 112-channel kernels ran 116 M scalar and 248 M vector instructions per forward on it, against a
// requantisation floor of about 50 M).
// NODES: the input may be a BiFPN node's sum of sources; false (the head-layer kernels) compiles that path - and its registers - out.
template <int NW, int CT, bool NODES = true>
__device__ __forceinline__ void sepconv_band_body(const BandArgs& a, int local, unsigned char* bd_smem) {
  constexpr int nwaves = NW, nthreads = 64 * NW;
  constexpr bool C64 = CT == 64, CK = CT != 0;                 // CK: the width is known at compile time
  constexpr int CS_T = CT == 64 ? 80 : CT;                      // bytes per pixel row in LDS: an odd multiple of 16 >= C (64 -> 80, 112 -> 112)
  static_assert(CT == 0 || CT == 64 || CT == 112, "compile-time widths: 64, 112");
  const long b = fdiv_small(local, frcp(a.nbands));
  const int band = local - (int)b * a.nbands;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 15, g = lane >> 4;
  const int y0 = band * a.rows, nr = min(a.rows, a.H - y0);
  const int PW = a.W + 2, NPh = (nr + 2) * PW, NPo = nr * a.W, NPG = (NPo + 15) >> 4;
  const int C = CK ? CT : a.C, CS = CK ? CS_T : a.CS;
  unsigned char* T0 = bd_smem;
  unsigned char* D = T0 + (a.rows + 2) * PW * CS;
  unsigned char* WP = D + (((a.rows * a.W + 15) >> 4) << 4) * CS;   // NT x KS KB of weights | 512 B bias | 512 B mult
  const int NT = (a.Cout + 15) >> 4, KS = CK ? (CT + 63) / 64 : a.KS, NCG = CK ? (CT + 15) / 16 : a.NCG;
  unsigned char* WB = WP + NT * KS * 1024;
  BD_STAMP(0);

  // ---- stage L: band + border -> T0; projection weights / bias / multipliers -> LDS ----
  for (int i = tid; i < NT * KS * 64; i += nthreads) *(v4i*)(WP + 16 * i) = a.wp[i];
  if (tid < 4 * NT) *(uint4*)(WB + 16 * tid) = *(const uint4*)((const unsigned char*)a.bp + 16 * tid);
  else if (tid >= 32 && tid < 32 + 4 * NT) *(uint4*)(WB + 512 + 16 * (tid - 32)) = *(const uint4*)((const unsigned char*)a.mp + 16 * (tid - 32));
  const float rcp_pw = frcp(PW);
  if (NODES && a.n_src > 0) {
    const bool up2[3] = {a.H == 2 * a.sh[0] && a.W == 2 * a.sw[0], a.H == 2 * a.sh[1] && a.W == 2 * a.sw[1], a.H == 2 * a.sh[2] && a.W == 2 * a.sw[2]};
    if (CK || (C & 15) == 0) {
      // 16 channels of one pixel per lane-iteration (the stage used to walk dwords: four times the address arithmetic and four
      // times the dependent rounds of global loads - it was two thirds of a node kernel's time, tools/probes/bd_probe.hip); the
      // source loads of an iteration are all requested before its sums are formed
      constexpr int LITN = C64 ? BD_LIT_NODE : BD_LIT_NODE_WIDE;
      const int npp = C64 ? 4 : CS >> 4, nreal = C64 ? 4 : C >> 4;   // 16-byte pieces per LDS pixel row / of real channels (constants under CT)
      const float rcp_npp = frcp(npp);
      const int total = NPh * npp;
      for (int i0 = tid; i0 < total; i0 += LITN * nthreads) {
        uint4 us[LITN][3];
        int pofs[LITN];   // LDS byte offset of the piece, -1: past the band; bit 30: outside the image or padding channels (zero point)
#pragma unroll
        for (int k = 0; k < LITN; k++) {
          const int i = i0 + k * nthreads, ic = min(i, total - 1);
          const int p = C64 ? ic >> 2 : (CK ? ic / (CS_T >> 4) : fdiv_small(ic, rcp_npp)), sg = ic - p * npp;
          const int hy = fdiv_small(p, rcp_pw), hx = p - hy * PW;
          const int iy = y0 + hy - 1, ix = hx - 1;
          const bool in = i < total && sg < nreal && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
          pofs[k] = i < total ? (p * CS + 16 * sg) | (in ? 0 : (1 << 30)) : -1;
#pragma unroll
          for (int j = 0; j < 3; j++) {
            us[k][j] = make_uint4(0u, 0u, 0u, 0u);
            if (j < a.n_src && in) us[k][j] = band_source16(a, j, b, iy, ix, sg, up2[j]);
          }
        }
#pragma unroll
        for (int k = 0; k < LITN; k++) {
          if (pofs[k] < 0) continue;
          uint4 v = make_uint4(a.zx4, a.zx4, a.zx4, a.zx4);
          if (!(pofs[k] >> 30)) {
            const unsigned u0[4] = {us[k][0].x, us[k][0].y, us[k][0].z, us[k][0].w}, u1[4] = {us[k][1].x, us[k][1].y, us[k][1].z, us[k][1].w},
                           u2[4] = {us[k][2].x, us[k][2].y, us[k][2].z, us[k][2].w};
            unsigned o[4];
#pragma unroll
            for (int d = 0; d < 4; d++) {
              if (a.chain == 0) o[d] = addq4(u0[d], u1[d], a.sumq);
              else {
                const unsigned pp = addq4(u0[d], u1[d], a.preq);
                o[d] = a.chain == 1 ? addq4(pp, u2[d], a.sumq) : addq4(u2[d], pp, a.sumq);
              }
            }
            v = make_uint4(o[0], o[1], o[2], o[3]);
          }
          *(uint4*)(T0 + (pofs[k] & 0xFFFFFF)) = v;
        }
      }
    } else {
    const int ndp = CS >> 2, nd = C >> 2;   // dwords per pixel visited / of real channels
    const float rcp_ndp = frcp(ndp);
    for (int i = tid; i < NPh * ndp; i += nthreads) {   // 4 channels per lane-iteration
      const int p = fdiv_small(i, rcp_ndp), cd = i - p * ndp;
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
    }
  } else {
    const int8_t* xb = a.x + b * (long)a.H * a.W * C;
    if constexpr (C64) {
      const uint4 z4 = make_uint4(a.zx4, a.zx4, a.zx4, a.zx4);
      const int total = NPh * 4;
      for (int i0 = tid; i0 < total; i0 += BD_LIT * nthreads) {    // 16 bytes per lane-iteration, BD_LIT loads in flight
        uint4 v[BD_LIT];
        int pofs[BD_LIT];
#pragma unroll
        for (int k = 0; k < BD_LIT; k++) {
          const int i = i0 + k * nthreads;
          const int p = i >> 2, sg = i & 3;
          const int hy = fdiv_small(min(p, NPh - 1), rcp_pw), hx = p - hy * PW;
          const int iy = y0 + hy - 1, ix = hx - 1;
          pofs[k] = i < total ? p * 80 + 16 * sg : -1;
          v[k] = z4;
          if (i < total && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) v[k] = *(const uint4*)(xb + (iy * a.W + ix) * 64 + 16 * sg);
        }
#pragma unroll
        for (int k = 0; k < BD_LIT; k++)
          if (pofs[k] >= 0) *(uint4*)(T0 + pofs[k]) = v[k];
      }
    } else {
    if (CK || (C & 15) == 0) {
      // 16-byte pieces, BD_LIT loads in flight (Lite2's 112-channel maps: the 8-byte walk below was nine dependent rounds of loads per band)
      const int npp = CS >> 4, nreal = C >> 4;
      const float rcp_npp = frcp(npp);
      const uint4 z4 = make_uint4(a.zx4, a.zx4, a.zx4, a.zx4);
      const int total = NPh * npp;
      for (int i0 = tid; i0 < total; i0 += BD_LIT * nthreads) {
        uint4 v[BD_LIT];
        int pofs[BD_LIT];
#pragma unroll
        for (int k = 0; k < BD_LIT; k++) {
          const int i = i0 + k * nthreads, ic = min(i, total - 1);
          const int p = CK ? ic / (CS_T >> 4) : fdiv_small(ic, rcp_npp), sg = ic - p * npp;
          const int hy = fdiv_small(p, rcp_pw), hx = p - hy * PW;
          const int iy = y0 + hy - 1, ix = hx - 1;
          pofs[k] = i < total ? p * CS + 16 * sg : -1;
          v[k] = z4;
          if (i < total && sg < nreal && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) v[k] = *(const uint4*)(xb + (iy * a.W + ix) * C + 16 * sg);
        }
#pragma unroll
        for (int k = 0; k < BD_LIT; k++)
          if (pofs[k] >= 0) *(uint4*)(T0 + pofs[k]) = v[k];
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
  BD_STAMP(1);
  // ---- stage D: depthwise; unit = (output pixel group, channel group cg).  Two units of a wave in flight: their operand reads are
  // requested together and their MFMA chains interleave (one unit at a time, every MFMA waited for the one before it and every
  // unit for its own LDS round trip) ----
  if (sub < nsub) {
    auto d_units = [&](auto full_c, auto u_c, int pg0) {
      constexpr int U = decltype(u_c)::value, FULL = decltype(full_c)::value;
      const unsigned char* pb[U];
      int slot[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        slot[u] = (pg0 + u * nsub) * 16 + r;
        const int sc = min(slot[u], NPo - 1);
        const int py = fdiv_small(sc, rcp_w);
        pb[u] = T0 + (sc + 2 * py) * CS + 16 * cg;   // (py * PW + px) with PW = W + 2 and px = sc - py * W
      }
      v4i bv[U][3], acc[U];
#pragma unroll
      for (int u = 0; u < U; u++)
#pragma unroll
        for (int m = 0; m < 3; m++) bv[u][m] = *(const v4i*)(pb[u] + tapoff[m]);
#pragma unroll
      for (int u = 0; u < U; u++) acc[u] = v4i_from(int4_plus(bq, FULL >= 2 ? RQ_KBIAS : 0));
#pragma unroll
      for (int m = 0; m < 3; m++)
#pragma unroll
        for (int u = 0; u < U; u++) acc[u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wdv[m], bv[u][m], acc[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < U; u++) *(unsigned*)(D + slot[u] * CS + 16 * cg + 4 * g) = rq_pack_b<FULL>(acc[u], mu, a.rqd);
    };
    // the requantisation flavour (saturating / explicitly clamped) is uniform: picked once per stage, not once per unit
    auto d_walk = [&](auto full_c) {
      int pg = sub;
      for (; pg + nsub < NPG; pg += 2 * nsub) d_units(full_c, std::integral_constant<int, 2>{}, pg);
      if (pg < NPG) d_units(full_c, std::integral_constant<int, 1>{}, pg);
    };
    rq_dispatch(a.rqd, d_walk);
  }
  __syncthreads();
  BD_STAMP(2);
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
  if (NT == 4 && KS == 1 && a.Cout == 64) {
    // 64 -> 64 channels (every Lite0 layer but the heads' last): unit u = wave + 16 i is tile t = wave & 3 of pixel group
    // (wave >> 2) + 4 i, so the wave's weights / bias / multipliers are loop invariants
    const int t = wave & 3, c0 = 16 * t + 4 * g;
    const v4i wv = *(const v4i*)(WP + (t * 64 + lane) * 16);
    const int4 bb = *(const int4*)(WB + 4 * c0);
    const float4 mm = *(const float4*)(WB + 512 + 4 * c0);
    // two units in flight, the output address one 64-bit multiply-add per unit from a base formed once
    int8_t* ob = a.out + ((b * a.H + y0) * (long)a.W) * a.Cout + c0;   // the band's pixels are contiguous: (y0 + py) * W + px = y0 * W + slot
    auto p_units = [&](auto full_c, auto u_c, int pg0) {
      constexpr int U = decltype(u_c)::value, FULL = decltype(full_c)::value;
      v4i dv[U], acc[U];
#pragma unroll
      for (int u = 0; u < U; u++) dv[u] = *(const v4i*)(D + ((pg0 + u * (nwaves / 4)) * 16 + r) * CS + 16 * g);
#pragma unroll
      for (int u = 0; u < U; u++) acc[u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wv, dv[u], v4i_from(int4_plus(bb, FULL >= 2 ? RQ_KBIAS : 0)), 0, 0, 0);
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int slot = (pg0 + u * (nwaves / 4)) * 16 + r;
        const unsigned d = rq_pack_b<FULL>(acc[u], mm, a.rqp);
        if (slot < NPo) *(unsigned*)(ob + (long)slot * a.Cout) = d;
      }
    };
    auto p_walk = [&](auto full_c) {
      int pg = wave >> 2;
      for (; pg + (nwaves / 4) < NPG; pg += 2 * (nwaves / 4)) p_units(full_c, std::integral_constant<int, 2>{}, pg);
      if (pg < NPG) p_units(full_c, std::integral_constant<int, 1>{}, pg);
    };
    rq_dispatch(a.rqp, p_walk);
  } else if (!C64 && NT <= nwaves && KS <= 2 && (nwaves - NT * (nwaves / NT)) * 8 <= nwaves) {   // (KS is a constant under CT)   // (maps of more than 64 channels; at most an eighth of the waves without a tile)
    // any other width (Lite1 / Lite2 maps, the heads' 9- / 36-channel outputs): wave w owns output tile w % NT and, of the pixel groups,
    // every (NW / NT)-th one, so its weights / bias / multipliers are loop invariants held in registers (the unit loop below re-read
    // them from LDS and re-derived (pixel group, tile) per unit: 40 vector instructions and 6 LDS reads per unit against 22 and 2)
    const int t = wave % NT, sub = wave / NT, nsub = nwaves / NT;
    if (sub < nsub) {
      const int c0 = 16 * t + 4 * g;
      v4i wv[2];
#pragma unroll
      for (int ks = 0; ks < 2; ks++) wv[ks] = ks < KS ? *(const v4i*)(WP + ((t * KS + ks) * 64 + lane) * 16) : (v4i){0, 0, 0, 0};
      const int4 bb = *(const int4*)(WB + 4 * c0);
      const float4 mm = *(const float4*)(WB + 512 + 4 * c0);
      auto p_units = [&](auto u_c, int pg0) {
        constexpr int U = decltype(u_c)::value;
        v4i dv[U][2], acc[U];
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
          for (int ks = 0; ks < 2; ks++)
            if (ks < KS) dv[u][ks] = *(const v4i*)(D + ((pg0 + u * nsub) * 16 + r) * CS + 16 * g + 64 * ks);
#pragma unroll
        for (int u = 0; u < U; u++) {
          acc[u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wv[0], dv[u][0], v4i_from(bb), 0, 0, 0);
          if (KS > 1) acc[u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wv[1], dv[u][1], acc[u], 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < U; u++) store_unit(pg0 + u * nsub, t, acc[u], mm);
      };
      int pg = sub;
      for (; pg + nsub < NPG; pg += 2 * nsub) p_units(std::integral_constant<int, 2>{}, pg);
      if (pg < NPG) p_units(std::integral_constant<int, 1>{}, pg);
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
  BD_STAMP(3);
}

// Several problems in one grid (the same head layer on all pyramid levels of both heads): the problem list lives in HBM.
#ifndef VBT_BD_HEAD_WAVES
#define VBT_BD_HEAD_WAVES 8
#endif
#ifndef VBT_BD_HEAD_MAXPX
#define VBT_BD_HEAD_MAXPX 240
#endif
constexpr int BD_HEAD_WAVES = VBT_BD_HEAD_WAVES, BD_HEAD_MAXPX = VBT_BD_HEAD_MAXPX;
// Maps of more than 64 channels (Lite1 / Lite2): 16 waves, held to 64 registers so that two workgroups still share a CU.  With 6 or 7
// channel groups only NCG * (NW / NCG) waves work in the depthwise stage - 7 of 8, each with every pixel group of the band, against 14 of
// 16 with half of them (tools/probes/bd_probe.hip, Lite2 head layer at 56x56: 24 000 cycles per band on 8 waves, 16 300 on 16; with
// one 16-wave workgroup per CU - what an 84-register build gets - the launch was slower all the same).
constexpr int BD_HEAD_WAVES_WIDE = 16;
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
  sepconv_band_body<BD_HEAD_WAVES, 64, false>(probs[pi], (int)blockIdx.x - mt.start[pi], bd_smem_dyn);
}
__global__ __launch_bounds__(64 * BD_HEAD_WAVES_WIDE, 2) void sepconv_band_wide_kernel(const BandArgs* __restrict__ probs, MultiTiles mt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char bd_smem_dyn[];
  const int pi = band_problem(mt);
  sepconv_band_body<BD_HEAD_WAVES_WIDE, 0, false>(probs[pi], (int)blockIdx.x - mt.start[pi], bd_smem_dyn);
}
__global__ __launch_bounds__(64 * BD_HEAD_WAVES_WIDE, 2) void sepconv_band_c112_kernel(const BandArgs* __restrict__ probs, MultiTiles mt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char bd_smem_dyn[];
  const int pi = band_problem(mt);
  sepconv_band_body<BD_HEAD_WAVES_WIDE, 112, false>(probs[pi], (int)blockIdx.x - mt.start[pi], bd_smem_dyn);
}
// One problem (a BiFPN node): the arguments travel in the kernel-argument segment, one dependent memory round trip
// less at the head of a kernel that is a chain of round trips.
__global__ __launch_bounds__(BD_THREADS) void sepconv_band_one_kernel(BandArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char bd_smem_dyn[];
  sepconv_band_body<BD_WAVES, 64>(a, (int)blockIdx.x, bd_smem_dyn);
}
__global__ __launch_bounds__(BD_THREADS) void sepconv_band_one_wide_kernel(BandArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char bd_smem_dyn[];
  sepconv_band_body<BD_WAVES, 0>(a, (int)blockIdx.x, bd_smem_dyn);
}
__global__ __launch_bounds__(BD_THREADS) void sepconv_band_one_c112_kernel(BandArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char bd_smem_dyn[];
  sepconv_band_body<BD_WAVES, 112>(a, (int)blockIdx.x, bd_smem_dyn);
}
#endif  // VBT_DEFINE_BAND_KERNELS
