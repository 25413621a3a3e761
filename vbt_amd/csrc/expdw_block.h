// Low-resolution MBConv, first half: expand 1x1 + ReLU6 -> depthwise kxk as ONE kernel whose tile is the whole image
// and whose grid splits the EXPANDED CHANNELS: workgroup (image b, channel group q) produces `cpw` 64-channel chunks of
// the depthwise output for all pixels of image b.  The projection then runs as a plain pointwise GEMM over that tensor
// (pw_*_kernel, residual ADD in its epilogue).
//
// Why this shape (maps of at most 20x20 pixels, blocks b6..b15 of Lite0):
//   * the tile kernel (fused_block.h) recomputes the expand on every 64-pixel tile's halo: 1.9x (3x3) to 2.9x (5x5) the
//     useful work on a 20x20 map, and its LDS tiles do not fit the 10x10 x 1152-channel blocks at all;
//   * expand and depthwise are independent per channel, so splitting the channels over workgroups needs no reduction
//     (the projection, which sums over channels, is the part that is NOT fused here) and fills the chip at batch 64:
//     64 images x (7.5 .. 18 chunks / cpw) workgroups;
//   * the expand runs on the real pixels only and lands in an LDS copy of the image bordered by its zero point, so SAME
//     padding costs nothing; the 6x expanded tensor never reaches HBM, the depthwise output (1/k^2 of the MACs) does.
// LDS:  T0 [H*W][T0S]      block input, real channels only (the reads of the K padding run into the next pixel: zero weights)
//       E  [PH*PW][80]     one 64-channel chunk of the expanded tensor inside a border of its zero point
//       D  [OH*OW (16-padded)][80]  one chunk of the depthwise output, copied out with 16-byte stores
// 8 wavefronts (XD_WAVES).  Both stages run on the 16x16x64 int8 MFMA; wave w keeps the operands of 16-channel tile (w & 3) in
// registers and walks the pixel groups (w >> 2) + 2i.  Depthwise on the matrix pipe as in fused_block.h, four taps per
// instruction: out[c][p] = sum_t W'[c][(t,c')] X[(t,c')][p], W' = w[t][c] delta(c,c'), exact int32.
// Arithmetic identical to the per-op kernels: same integer accumulations, same single-op float requantisation.
#pragma once

// Waves per workgroup.  Measured end to end with three forwards in flight (A/B of two builds on one box): 16 waves 90.2 k
// frames/s, 8 waves 91.9 k (92.5 k with the chunks per workgroup re-tuned), 4 waves 86.6 k.  A 16-wave workgroup is faster alone
// (15 vs 21 us on b6) but holds every SIMD of its CU at each of its barriers; 8 waves leave issue slots to the other forwards.
#ifndef VBT_XD_WAVES
#define VBT_XD_WAVES 8
#endif
constexpr int XD_WAVES = VBT_XD_WAVES, XD_THREADS = 64 * XD_WAVES;
constexpr int XD_EST = 80;   // bytes per pixel of E and D rows (64 + 16: bank spread, 16-byte aligned)

struct ExpDwArgs {
  const int8_t* x;   // [B][H][W][Cin]
  int8_t* out;       // [B][OH][OW][Ce]: the graph's depthwise output tensor
  int H, W, Cin, OH, OW, Ce;
  int PW, PH;        // bordered E image (PH: of the tallest band)
  int t0_bytes, e_bytes;   // LDS bytes of T0 / E (sized for the tallest band)
  int pad_t, pad_l;
  int T0S;           // odd multiple of 16 bytes >= Cin
  int nchunks, cpw;  // 64-channel chunks in all / per workgroup
  int nbands, brows; // row bands per image (1: the whole image) / output rows per band: maps of more than 400 pixels (Lite1 / Lite2)
                     // do not fit LDS as a whole - a workgroup then owns `brows` output rows and expands the input rows they need
  const v4i* we;     // expand weights [chunk][ks][t][lane] x 16 B: row i of tile t = channel 64c + 16t + i, k = 64ks + 16g + j
  const int* be;     // bias with the input zero point folded, padded to 64 * nchunks
  const float* me;
  Rq rqe;
  unsigned zeb;      // zero point of the expanded tensor x4
  const long* wdc;   // depthwise [chunk][cg][lane] x 8 B: byte m = weight of tap 4m + g (0 past the kernel) for channel 64c + 16cg + (lane & 15);
                     // the MFMA operand (that byte on the diagonal of 16) is rebuilt in registers (diag_operand)
  const int* bd;     // bias with the expanded tensor's zero point folded
  const float* md;
  Rq rqd;
};

template <int KK, int S, int KS64>
__global__ __launch_bounds__(XD_THREADS) void expdw_image_kernel(ExpDwArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char xd_smem[];
  constexpr int KT = (KK * KK + 3) / 4;   // depthwise MFMAs per unit: four taps each
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 15, g = lane >> 4;
  const int ngroups = fdiv_small(a.nchunks + a.cpw - 1, frcp(a.cpw));
  const int per_image = ngroups * a.nbands;
  const long b = fdiv_small((int)blockIdx.x, frcp(per_image));
  const int rem = blockIdx.x - (int)b * per_image;
  const int band = fdiv_small(rem, frcp(ngroups));
  const int grp = rem - band * ngroups;
  // band geometry: output rows [oy0, oy1), the rows of the padded expanded image they read, the input rows behind those
  const int oy0 = band * a.brows, oy1 = min(oy0 + a.brows, a.OH);
  const int PHb = (oy1 - oy0 - 1) * S + KK;                        // rows of this band's E image
  const int iy_lo = max(oy0 * S - a.pad_t, 0), iy_hi = min(oy0 * S - a.pad_t + PHb, a.H);
  const int erow0 = iy_lo + a.pad_t - oy0 * S;                     // E row of input row iy_lo
  const int HW = (iy_hi - iy_lo) * a.W, OHW = (oy1 - oy0) * a.OW;  // pixels of the band: input / output
  const int NPGi = (HW + 15) >> 4, NPGo = (OHW + 15) >> 4;
  unsigned char* T0 = xd_smem;
  unsigned char* E = T0 + a.t0_bytes;
  unsigned char* D = E + a.e_bytes;
  const float rcp_w = frcp(a.W), rcp_ow = frcp(a.OW);

  // ---- input rows -> T0 (16-byte granules, 8-byte ones when Cin % 16 != 0), E <- zero point everywhere (border and rows outside the image keep it) ----
  {
    const int8_t* xb = a.x + (b * (long)a.H + iy_lo) * a.W * a.Cin;
    if ((a.Cin & 15) == 0) {
      const int ng = a.Cin >> 4;
      const float rcp_ng = frcp(ng);
      for (int i = tid; i < HW * ng; i += XD_THREADS) {
        const int p = fdiv_small(i, rcp_ng), sg = i - p * ng;
        *(uint4*)(T0 + p * a.T0S + 16 * sg) = *(const uint4*)(xb + p * a.Cin + 16 * sg);
      }
    } else {   // 88 / 120 input channels (Lite2): 8-byte granules (Cin % 8 == 0)
      const int ng = a.Cin >> 3;
      const float rcp_ng = frcp(ng);
      for (int i = tid; i < HW * ng; i += XD_THREADS) {
        const int p = fdiv_small(i, rcp_ng), sg = i - p * ng;
        *(uint2*)(T0 + p * a.T0S + 8 * sg) = *(const uint2*)(xb + p * a.Cin + 8 * sg);
      }
    }
    const uint4 z4 = make_uint4(a.zeb, a.zeb, a.zeb, a.zeb);
    for (int i = tid; i < PHb * a.PW * (XD_EST / 16); i += XD_THREADS) *(uint4*)(E + 16 * i) = z4;
  }
  const int tq = wave & 3;   // this wave's 16-channel tile of the chunk, in both stages
  // depthwise lane geometry: lane (r, g) reads, for MFMA m, the 16 channels of tile tq at input pixel
  // (oy*S + ty, ox*S + tx) with tap 4m + g = ty*KK + tx (taps past the kernel re-read a valid one: zero weights)
  int tapoff[KT];
#pragma unroll
  for (int m = 0; m < KT; m++) {
    const int tap = min(4 * m + g, KK * KK - 1);
    tapoff[m] = ((tap / KK) * a.PW + (tap % KK)) * XD_EST;
  }
  const int c_first = grp * a.cpw, c_last = min(c_first + a.cpw, a.nchunks);
  // Operand flow: every global load is requested one stage before its use, so no stage starts with an exposed L2 round trip:
  // chunk c's expand operands arrive during the previous chunk's depthwise (the first ones during the image load above),
  // its depthwise operands during its own expand.
  v4i ew[KS64];
  int4 eb;
  float4 em;
  {
    const v4i* w = a.we + ((long)c_first * KS64 * 4 + tq) * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < KS64; ks++) ew[ks] = w[ks * 4 * 64];
    eb = *(const int4*)(a.be + c_first * 64 + 16 * tq + 4 * g);
    em = *(const float4*)(a.me + c_first * 64 + 16 * tq + 4 * g);
  }
  __syncthreads();
  for (int c = c_first; c < c_last; c++) {
    v4i dwv[KT];
    {
      const unsigned long long wc = (unsigned long long)a.wdc[(long)(c * 4 + tq) * 64 + lane];
#pragma unroll
      for (int m = 0; m < KT; m++) dwv[m] = diag_operand((unsigned)(wc >> (8 * m)) & 0xffu, r);
    }
    const int4 bq = *(const int4*)(a.bd + c * 64 + 16 * tq + 4 * g);
    const float4 mu = *(const float4*)(a.md + c * 64 + 16 * tq + 4 * g);
    // ---- stage E: expand chunk c on the real pixels; unit = (pixel group, tile tq) ----
    for (int pg = wave >> 2; pg < NPGi; pg += XD_WAVES / 4) {
      const int p = pg * 16 + r, pc = min(p, HW - 1);
      v4i acc = v4i_from(eb);
      const unsigned char* brow = T0 + pc * a.T0S + 16 * g;
#pragma unroll
      for (int ks = 0; ks < KS64; ks++) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(ew[ks], *(const v4i*)(brow + 64 * ks), acc, 0, 0, 0);
      if (p < HW) {
        const int py = fdiv_small(p, rcp_w), px = p - py * a.W;
        *(unsigned*)(E + ((py + erow0) * a.PW + px + a.pad_l) * XD_EST + 16 * tq + 4 * g) = rq_pack_b(acc, em, a.rqe);
      }
    }
    __syncthreads();   // E complete; the previous chunk's D has been copied out (that copy precedes this barrier)
    if (c + 1 < c_last) {   // next chunk's expand operands
      const v4i* w = a.we + ((long)(c + 1) * KS64 * 4 + tq) * 64 + lane;
#pragma unroll
      for (int ks = 0; ks < KS64; ks++) ew[ks] = w[ks * 4 * 64];
      eb = *(const int4*)(a.be + (c + 1) * 64 + 16 * tq + 4 * g);
      em = *(const float4*)(a.me + (c + 1) * 64 + 16 * tq + 4 * g);
    }
    // ---- stage D: depthwise on chunk c; unit = (output pixel group, channel tile tq) ----
    for (int pg = wave >> 2; pg < NPGo; pg += XD_WAVES / 4) {
      const int slot = pg * 16 + r, sc = min(slot, OHW - 1);
      const int oy = fdiv_small(sc, rcp_ow), ox = sc - oy * a.OW;
      const unsigned char* pb = E + ((oy * S) * a.PW + ox * S) * XD_EST + 16 * tq;
      v4i acc = v4i_from(bq);
#pragma unroll
      for (int m = 0; m < KT; m++) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(dwv[m], *(const v4i*)(pb + tapoff[m]), acc, 0, 0, 0);
      *(unsigned*)(D + slot * XD_EST + 16 * tq + 4 * g) = rq_pack_b(acc, mu, a.rqd);
    }
    __syncthreads();   // D complete, E free for the next chunk's expand
    // ---- stage O: D -> the depthwise output tensor, 16 bytes per lane, only the chunk's real channels ----
    {
      const int nv = min(64, a.Ce - 64 * c) >> 4;   // 16-byte parts of this chunk (Ce % 16 == 0)
      const float rcp_nv = frcp(nv);
      int8_t* ob = a.out + (b * (long)a.OH + oy0) * a.OW * a.Ce + 64 * c;
      for (int i = tid; i < OHW * nv; i += XD_THREADS) {
        const int slot = fdiv_small(i, rcp_nv), part = i - slot * nv;
        *(uint4*)(ob + slot * a.Ce + 16 * part) = *(const uint4*)(D + slot * XD_EST + 16 * part);
      }
    }
  }
}
