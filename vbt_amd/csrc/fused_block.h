// Fused block kernel: [expand 1x1 + ReLU6 ->] depthwise kxk (stride 1|2) -> project 1x1 [-> residual ADD]
// on LDS-resident tiles.  Included by detector.hip after the requantisation helpers.
//
// Covers (a) every MBConv block of the EfficientNet-Lite backbone (the 6x expanded tensor and the
// depthwise output never touch HBM) and, with EXPAND = false, (b) every SeparableConv of the BiFPN and
// of the box/class heads (dw3x3 -> pw1x1).  Results are bit-identical to running the constituent graph
// ops one by one (same integer accumulations, same single-op float requantisation).
//
// One workgroup (4 wavefronts) = one output tile of <= 64 pixels of one image:
//   T0  [NPh][T0S]   input halo tile (int8 NHWC rows), out-of-image pixels = input zero point
//   E   [NPh][80]    one 64-channel chunk of the expanded tensor on the halo (EXPAND only)
//   D   [64][72]     one 64-channel chunk of the depthwise output
// per 64-channel chunk: expand (int8 MFMA, all halo pixels; out-of-image halo pixels are forced to the
// zero point of E because the depthwise pads ITS input) -> barrier -> depthwise (lane = 4 channels x 4
// output columns, weights in registers, cvt_f32_ubyte + fma, exact) -> barrier -> project accumulation
// (int8 MFMA, wave w owns pixels 16w..16w+15, accumulators stay in registers across chunks).
#pragma once

struct FusedArgs {
  const int8_t* x;
  int8_t* out;
  int H, W, Cin, OH, OW, Cout;
  int pad_t, pad_l;
  int TX, TY, tiles_x, tiles_y;
  int T0S;       // bytes per halo pixel row in LDS
  int nchunks;   // Ce_pad / 64
  int zx;        // zero point of the block input
  // expand (EXPAND only)
  const long* we;
  const int* be;
  const float* me;
  int KSe, ze, loe, hie;
  Rq rqe;
  // 48-channel chunking (template NT = 3) of the same block, for expanded widths that are multiples of 48 but not of 64
  // (6 x {16, 24, 40, 80, 112}): no padded channels in the expand / depthwise stages.  nch3 == 0: not available.
  const long* we3;   // [(c*KSe + ks)*3 + t][lane]: cout = 48c + 12(i>>2) + 4t + (i&3)
  const long* wdm3;  // [(c*3 + cg)*KT + m][lane]: channel 48c + 16cg + i
  const v4i* wp3;    // project weights, K-step c (16x16x64 layout) = channels 48c..48c+47 (+16 zero columns)
  int nch3, KSp3;
  // depthwise
  const float* wd;  // [k*k][Ce_pad]
  const int* bd;    // folded, padded to Ce_pad
  const float* md;
  int zd, lod, hid;
  Rq rqd;
  // depthwise on the matrix pipe (MDW): diagonal-embedded weights [chunk][cg][m][lane] x 8 B and the bias
  // folded for raw int8 inputs (bias - z_x * sum(w))
  const long* wdm;
  const int* bdm;
  // the same for the 16x16x64 MFMA (DW64): [16-channel group q][m][lane] x 16 B, lane (i = lane & 15 -> channel 16q + i,
  // g = lane >> 4): byte j is non-zero only for j == i and holds the weight of tap (m, g):
  //   3x3: (row m, column g), g = 3 is a zero column           -> 3 instructions
  //   5x5: m < 5: (row m, column g); m = 5: (row g, column 4); m = 6: g = 0 -> (4, 4), the rest zero -> 7 instructions
  // so the B operand of (m, g) sits at a compile-time offset from one of two per-lane base addresses
  const v4i* wd64;
  // the same operands as one byte each: [16-channel group q][lane] x 8 B, byte m = the non-zero byte of operand (q, m) of that
  // lane; the kernels rebuild the 16-byte operand in registers (diag_operand)
  const long* wd64c;
  // project
  const v4i* wp;    // packed for the 16x16x64 MFMA (pack_weights64), K = Ce_pad, one K-step per 64-channel chunk
  const int* bp;
  const float* mp;
  int KSp, zo, lop, hip;
  Rq rqp;
  // residual ADD: out = ADD(project output, block input), the graph's input order (AddQ: XNNPACK qs8-vadd, detector.hip)
  int has_res;
  AddQ resq;
  // BiFPN node: the tile is the sum (+ReLU6) of two or three sources, each read through a resampling map
  // (0 identity, 1 nearest-neighbour up: src = floor(dst*in/out), 2 max-pool 3x3/2 SAME).  ADD is binary:
  //   n_src == 2: tile = ADD(src0, src1; sumq)
  //   n_src == 3: p = ADD(src0, src1; preq) - the partial sum with its own quantisation - then
  //               tile = ADD(p, src2; sumq) (chain == 1) or ADD(src2, p; sumq) (chain == 2)
  int n_src;
  const int8_t* src[3];
  int sh[3], sw[3], smode[3], spt[3], spl[3];
  int chain;
  AddQ sumq, preq;
};

__device__ __forceinline__ unsigned node_sum4(const unsigned u[3], const FusedArgs& a) {
  if (a.chain == 0) return addq4(u[0], u[1], a.sumq);
  const unsigned p = addq4(u[0], u[1], a.preq);
  return a.chain == 1 ? addq4(p, u[2], a.sumq) : addq4(u[2], p, a.sumq);
}

// two unsigned 16-bit maxima in one VALU op (v_pk_max_u16)
typedef unsigned short v2u16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_max_u16(unsigned a, unsigned b) {
  v2u16 x, y;
  __builtin_memcpy(&x, &a, 4);
  __builtin_memcpy(&y, &b, 4);
  v2u16 r = __builtin_elementwise_max(x, y);
  unsigned o;
  __builtin_memcpy(&o, &r, 4);
  return o;
}

constexpr int FB_EST = 80;  // E tile bytes per pixel (64 + 16: bank spread, 16-B aligned)
constexpr int FB_DST = 80;  // D tile bytes per pixel: 16-byte aligned rows (the projection reads 16 bytes per lane), 20-dword pitch: conflict-free

// KSE > 0: the expand has exactly KSE K-steps and its weights stay in registers for the whole chunk (the pixel-group
// loop then issues no global loads); KSE == 0: K-steps are a runtime loop that streams the weights.
// PPW: 16-pixel output slot groups per wave.  2 -> 128-pixel tiles (matrix-pipe depthwise only): the per-tile prologue /
// epilogue addressing is paid once per 128 pixels, the halo shrinks (3x3: 1.41x instead of 1.56x, 5x5: 1.88x / 2.25x)
// and 12 halo pixel groups split evenly over the 4 waves of the expand stage.
// DW64: depthwise on the 16x16x64 MFMA, four taps per instruction, for tiles of exactly (8 * PPW) x 8 output pixels: the halo
// row length is then a compile-time constant and every LDS address of the stage is `lane base + immediate` - no address
// arithmetic in the stage at all (the 16x16x32 form spends one v_add per ds_read_b64 and needs 5 / 13 instructions per
// pixel group instead of 3 / 7).
template <int KK, int S, int NBP, bool EXPAND, bool MDW, int KSE = 0, int NT = 4, int PPW = 1, bool DW64 = false>
__device__ __forceinline__ void fused_block_body(const FusedArgs& a, int tile, unsigned char* fb_smem) {
  static_assert(PPW == 1 || MDW, "128-pixel tiles exist for the matrix-pipe depthwise only");
  static_assert(!DW64 || (EXPAND && MDW && (KK == 3 || KK == 5)), "DW64: fused expand blocks, 3x3 / 5x5");
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 15, g = lane >> 4;
  // tile -> (image, ty, tx) and every later pixel -> (row, column) split go through fdiv_small: no integer division
  // (~40 VALU instructions each) anywhere in the prologue
  const int trow = fdiv_small(tile, frcp(a.tiles_x));
  const int tx = tile - trow * a.tiles_x;
  const int bimg = fdiv_small(trow, frcp(a.tiles_y));
  const int ty = trow - bimg * a.tiles_y;
  const long b = bimg;
  const int oy0 = ty * a.TY, ox0 = tx * a.TX;
  const int TXp = (a.TX + 3) & ~3;
  const int HWx = (TXp - 1) * S + KK, HWy = (a.TY - 1) * S + KK, NPh = HWx * HWy;
  const int iy0 = oy0 * S - a.pad_t, ix0 = ox0 * S - a.pad_l;
  const float rcp_hwx = frcp(HWx), rcp_txp = frcp(TXp);
  unsigned char* T0 = fb_smem;
  unsigned char* E = T0 + ((NPh * a.T0S + 15) & ~15);
  // 48-channel chunks keep E rows at 72 bytes: fewer bank conflicts in the depthwise reads (18-dword pixel stride: conflict-
  // free at stride 1, 2-way at stride 2; 80 bytes gives 2-way / 4-way) and less LDS per workgroup
  // DW64 reads 16-byte groups, so its rows stay 16-byte aligned: 48 bytes for 48-channel chunks (conflict-free b128 reads at
  // stride 1, 3-dword-skewed conflict-free writes, and 40 % less LDS than 80-byte rows: one more workgroup per CU on b1-b3)
  constexpr int EST = (EXPAND && NT == 3) ? (DW64 ? 48 : 72) : FB_EST;
  unsigned char* D = E + (EXPAND ? ((NPh * EST + 15) & ~15) : 0);
  // SeparableConv / node / head tiles whose depthwise input is ONE 64-channel chunk (BiFPN width 64): the projection
  // weights and its bias / multipliers are copied into LDS while the input tile loads, so the projection and the epilogue
  // start from LDS instead of from two more exposed L2 round trips (these kernels are latency chains, not bandwidth)
  bool stage_p = false;
  if constexpr (!EXPAND && NBP <= 2) stage_p = a.nchunks == 1;
  unsigned char* WPS = D + 64 * PPW * FB_DST;    // [NBP][4][64] x 16 B
  unsigned char* BPS = WPS + NBP * 4096;         // bias int[NBP*64] | mult float[NBP*64]

  // ---- stage L: input halo tile -> LDS ----
  bool summed = false;
  if constexpr (!EXPAND) summed = a.n_src > 0;
  if (summed) {
    // BiFPN node input: resample + binary integer ADD(s) + clamp, 4 channels per lane-iteration (same arithmetic as add_kernel)
    const int nd = a.Cin >> 2;
    const unsigned zb4 = (unsigned)(a.zx & 255) * 0x01010101u;
    const int pstep = fdiv_small(256, frcp(nd));      // pixels advanced per round (planner guarantees nd <= 256)
    int p = fdiv_small(tid, frcp(nd));
    const int cd = tid - p * nd;
    const bool up2[3] = {a.H == 2 * a.sh[0] && a.W == 2 * a.sw[0], a.H == 2 * a.sh[1] && a.W == 2 * a.sw[1], a.H == 2 * a.sh[2] && a.W == 2 * a.sw[2]};
    for (; tid < pstep * nd && p < NPh; p += pstep) {
      const int hy = fdiv_small(p, rcp_hwx), hx = p - hy * HWx;
      const int iy = iy0 + hy, ix = ix0 + hx;
      unsigned v = zb4;
      if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) {
        unsigned us[3] = {0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < 3; j++) {
          if (j < a.n_src) {
            const int8_t* sb = a.src[j] + b * (long)a.sh[j] * a.sw[j] * a.Cin + 4 * cd;
            unsigned u;
            if (a.smode[j] == 0) {
              u = *(const unsigned*)(sb + ((long)iy * a.sw[j] + ix) * a.Cin);
            } else if (a.smode[j] == 1) {
              int yy, xx;
              if (up2[j]) { yy = iy >> 1; xx = ix >> 1; }
              else { yy = (iy * a.sh[j]) / a.H; xx = (ix * a.sw[j]) / a.W; }
              u = *(const unsigned*)(sb + ((long)yy * a.sw[j] + xx) * a.Cin);
            } else {
              // 3x3/2 max pool read in place: nine independent loads (clamped addresses, out-of-map taps replaced by
              // -128) so they are all in flight together; the byte-wise signed max runs on the u8 image of the values
              // (x ^ 0x80) as two packed-u16 maxima per tap.
              unsigned tv[9];
#pragma unroll
              for (int ky = 0; ky < 3; ky++) {
                const int yy = iy * 2 + ky - a.spt[j];
                const int yc = min(max(yy, 0), a.sh[j] - 1);
#pragma unroll
                for (int kx = 0; kx < 3; kx++) {
                  const int xx = ix * 2 + kx - a.spl[j];
                  const int xc = min(max(xx, 0), a.sw[j] - 1);
                  const unsigned t = *(const unsigned*)(sb + ((long)yc * a.sw[j] + xc) * a.Cin);
                  tv[ky * 3 + kx] = (yy == yc && xx == xc) ? (t ^ 0x80808080u) : 0u;
                }
              }
              unsigned lo = 0u, hi = 0u;   // bytes 0,2 and bytes 1,3 as u16 lanes
#pragma unroll
              for (int q = 0; q < 9; q++) {
                const unsigned l = tv[q] & 0x00FF00FFu, h = (tv[q] >> 8) & 0x00FF00FFu;
                lo = pk_max_u16(lo, l);
                hi = pk_max_u16(hi, h);
              }
              u = (lo | (hi << 8)) ^ 0x80808080u;
            }
            us[j] = u;
          }
        }
        v = node_sum4(us, a);   // binary integer ADDs (+ fused ReLU6 clamp), exactly the standalone add_kernel arithmetic
      }
      *(unsigned*)(T0 + p * a.T0S + 4 * cd) = v;
    }
  } else {
    // plain tensor, coalesced along channels; (hy,hx) advance incrementally (no divisions in the loop).
    // SeparableConv tiles (rows of Cp + 16 bytes, 16-byte aligned) with Cin % 16 == 0 move 16 bytes per lane-iteration.
    if (!EXPAND && (a.Cin & 15) == 0) {
      const int ng = a.Cin >> 4;
      const unsigned zb1 = (unsigned)(a.zx & 255) * 0x01010101u;
      const uint4 zb = make_uint4(zb1, zb1, zb1, zb1);
      const int8_t* xb = a.x + b * (long)a.H * a.W * a.Cin;
      const int pstep = fdiv_small(256, frcp(ng));
      int p = fdiv_small(tid, frcp(ng));
      const int sg = tid - p * ng;
      for (; tid < pstep * ng && p < NPh; p += pstep) {
        const int hy = fdiv_small(p, rcp_hwx), hx = p - hy * HWx;
        const int iy = iy0 + hy, ix = ix0 + hx;
        uint4 v = zb;
        if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) v = *(const uint4*)(xb + (iy * a.W + ix) * a.Cin + 16 * sg);
        *(uint4*)(T0 + p * a.T0S + 16 * sg) = v;
      }
    } else {
    const int ng = a.Cin >> 3;
    const unsigned long long zb = (unsigned long long)(a.zx & 255) * 0x0101010101010101ull;
    const int8_t* xb = a.x + b * (long)a.H * a.W * a.Cin;
    const int pstep = fdiv_small(256, frcp(ng));
    int p = fdiv_small(tid, frcp(ng));
    const int sg = tid - p * ng;
    for (; tid < pstep * ng && p < NPh; p += pstep) {
      const int hy = fdiv_small(p, rcp_hwx), hx = p - hy * HWx;
      const int iy = iy0 + hy, ix = ix0 + hx;
      unsigned long long v = zb;
      if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) v = *(const unsigned long long*)(xb + (iy * a.W + ix) * a.Cin + 8 * sg);
      *(unsigned long long*)(T0 + p * a.T0S + 8 * sg) = v;
    }
    }
  }
  if (stage_p) {
    for (int i = tid; i < NBP * 256; i += 256) *(uint4*)(WPS + 16 * i) = *(const uint4*)((const unsigned char*)a.wp + 16 * (long)i);
    if (tid < NBP * 16) *(uint4*)(BPS + 16 * tid) = *(const uint4*)((const unsigned char*)a.bp + 16 * tid);
    else if (tid < NBP * 32) *(uint4*)(BPS + 16 * tid) = *(const uint4*)((const unsigned char*)a.mp + 16 * (tid - NBP * 16));
  }
  // which of this wave's halo pixel groups hold out-of-image pixels (expand epilogue), bit i <-> pg = wave + 4i
  const int NPG = (NPh + 15) >> 4;
  unsigned oob_mask = 0, tail_mask = 0;
  if (EXPAND) {
    // interior tiles (halo completely inside the image - most tiles of the large maps) have no out-of-image pixel:
    // the per-pixel coordinate test is skipped for the whole workgroup
    const bool border = iy0 < 0 || ix0 < 0 || iy0 + HWy > a.H || ix0 + HWx > a.W;
    for (int i = 0, pg = wave; pg < NPG; pg += 4, i++) {
      int p = pg * 16 + r;
      if (border) {
        int pc = min(p, NPh - 1);
        int hy = fdiv_small(pc, rcp_hwx), hx = pc - hy * HWx;
        int iy = iy0 + hy, ix = ix0 + hx;
        if (!(iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)) oob_mask |= 1u << i;
      }
      if (p >= NPh) tail_mask |= 1u << i;
    }
  }
  v4i acc[PPW][NBP][4];
#pragma unroll
  for (int pp = 0; pp < PPW; pp++)
#pragma unroll
    for (int nb = 0; nb < NBP; nb++)
#pragma unroll
      for (int t = 0; t < 4; t++) acc[pp][nb][t] = (v4i){0, 0, 0, 0};
  // depthwise lane geometry
  const int cq = tid & 15, strip = tid >> 4;
  const int nsx = TXp >> 2;
  const int sy = fdiv_small(strip, frcp(nsx)), sx = (strip - sy * nsx) * 4;
  const bool dw_active = sy < a.TY;
  // MDW: lane (r = pixel of a 16-slot group, g) -> halo offset of the slot's window origin, per slot group
  int hbase[4 * PPW];
#pragma unroll
  for (int pg = 0; pg < 4 * PPW; pg++) hbase[pg] = 0;
  if constexpr (MDW && !DW64) {
#pragma unroll
    for (int pg = 0; pg < 4 * PPW; pg++) {
      int slot = pg * 16 + r;
      int sq = TXp == 8 ? (slot >> 3) : fdiv_small(slot, rcp_txp);   // 8-wide tiles (the common case): shifts only
      int py_ = min(sq, a.TY - 1), px_ = slot - sq * TXp;
      hbase[pg] = (py_ * S) * HWx + px_ * S;
    }
  }
  // SeparableConv / node / head tiles (no expand): the first chunk's depthwise operands do not depend on the tile, so
  // they are requested before the barrier and their latency overlaps the input tile loads
  constexpr int KTP = (KK * KK + 1) / 2;
  long wpre[KTP];
  int4 bpre = make_int4(0, 0, 0, 0);
  float4 mpre = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (!EXPAND && MDW) {
    const long* wm0 = a.wdm + ((long)wave * KTP) * 64 + lane;
#pragma unroll
    for (int mi = 0; mi < KTP; mi++) wpre[mi] = wm0[mi * 64];
    bpre = *(const int4*)(a.bdm + 16 * wave + 4 * g);
    mpre = *(const float4*)(a.md + 16 * wave + 4 * g);
  }
  __syncthreads();

  constexpr int CH = 16 * NT;                       // channels per chunk
  const int nch = NT == 3 ? a.nch3 : a.nchunks;
  const long* const wdm_ = NT == 3 ? a.wdm3 : a.wdm;
  for (int c = 0; c < nch; c++) {
    // low-resolution blocks (KSE >= 3 <=> Cin >= 80: few workgroups, latency-bound): this chunk's depthwise operands are
    // requested now and consumed after the expand stage.  On the high-resolution blocks the extra live registers cost
    // more than the hidden latency gains (b1: 100 -> 114 us), so they keep loading after the barrier.
    constexpr bool PREF_DW = EXPAND && MDW && KSE >= 3;
    if constexpr (PREF_DW) {
      if (NT == 4 || wave < NT) {
        const long* wmc = wdm_ + ((long)(c * NT + wave) * KTP) * 64 + lane;
#pragma unroll
        for (int mi = 0; mi < KTP; mi++) wpre[mi] = wmc[mi * 64];
        bpre = *(const int4*)(a.bdm + c * CH + 16 * wave + 4 * g);
        mpre = *(const float4*)(a.md + c * CH + 16 * wave + 4 * g);
      }
    }
    if (EXPAND) {
      // ---- stage E: expand chunk c on every halo pixel ----
      // lane (r, g) owns the 4*NT consecutive channels c*CH + 4*NT*g + 4t + j, t < NT
      int4 eb[NT];
      float4 em[NT];
#pragma unroll
      for (int t = 0; t < NT; t++) {
        eb[t] = *(const int4*)(a.be + c * CH + 4 * NT * g + 4 * t);
        em[t] = *(const float4*)(a.me + c * CH + 4 * NT * g + 4 * t);
      }
      const unsigned zeb = (unsigned)(a.ze & 255) * 0x01010101u;
      const long* w = (NT == 3 ? a.we3 : a.we) + (long)c * a.KSe * NT * 64 + lane;
      long wreg[KSE > 0 ? KSE : 1][NT];
      if constexpr (KSE > 0) {
#pragma unroll
        for (int ks = 0; ks < KSE; ks++)
#pragma unroll
          for (int t = 0; t < NT; t++) wreg[ks][t] = w[(ks * NT + t) * 64];
      }
      // the requantisation flavour (saturating / explicitly clamped) is uniform: pick it once, outside the loop, so the
      // loop body stays one basic block and the scheduler can overlap the four MFMAs with the previous tile's epilogue
      auto expand_loop = [&](auto full_tag) {
        constexpr int FULLK = decltype(full_tag)::value;
        int4 ebk[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) ebk[t] = int4_plus(eb[t], FULLK >= 2 ? RQ_KBIAS : 0);
        for (int i = 0, pg = wave; pg < NPG; pg += 4, i++) {
          const int p = pg * 16 + r;
          const int pc = min(p, NPh - 1);
          v4i ea[NT];
#pragma unroll
          for (int t = 0; t < NT; t++) ea[t] = v4i_from(ebk[t]);
          const unsigned char* brow = T0 + pc * a.T0S + 8 * g;
          if constexpr (KSE > 0) {
#pragma unroll
            for (int ks = 0; ks < KSE; ks++) {
              long bv = *(const long*)(brow + 32 * ks);
#pragma unroll
              for (int t = 0; t < NT; t++) ea[t] = __builtin_amdgcn_mfma_i32_16x16x32_i8(wreg[ks][t], bv, ea[t], 0, 0, 0);
            }
          } else {
#pragma unroll 2
            for (int ks = 0; ks < a.KSe; ks++) {
              long bv = *(const long*)(brow + 32 * ks);
#pragma unroll
              for (int t = 0; t < NT; t++) ea[t] = __builtin_amdgcn_mfma_i32_16x16x32_i8(w[(ks * NT + t) * 64], bv, ea[t], 0, 0, 0);
            }
          }
          unsigned d[NT];
#pragma unroll
          for (int t = 0; t < NT; t++) {
            d[t] = rq_pack_b<FULLK>(ea[t], em[t], a.rqe);
            if ((oob_mask >> i) & 1u) d[t] = zeb;
          }
          if (!((tail_mask >> i) & 1u)) {
            if constexpr (NT == 4) {
              *(uint4*)(E + p * FB_EST + 16 * g) = make_uint4(d[0], d[1], d[2], d[3]);
            } else {
#pragma unroll
              for (int t = 0; t < NT; t++) *(unsigned*)(E + p * EST + 4 * NT * g + 4 * t) = d[t];
            }
          }
        }
      };
      rq_dispatch(a.rqe, expand_loop);
      __syncthreads();
    }
    // ---- stage D: depthwise on chunk c ----
    if constexpr (DW64) {
      constexpr int TXP = 8 * PPW, HWX = (TXP - 1) * S + KK;
      constexpr int KT64 = KK == 3 ? 3 : 7;
      constexpr int PGS = (16 / TXP) * S * HWX * EST;   // slot group pg -> pg + 1
      if (NT == 4 || wave < NT) {   // wave = 16-channel group of the chunk (48-channel chunks: wave 3 sits this stage out)
        const unsigned long long wc = (unsigned long long)a.wd64c[(long)(c * NT + wave) * 64 + lane];
        v4i wreg[KT64];
#pragma unroll
        for (int mi = 0; mi < KT64; mi++) wreg[mi] = diag_operand((unsigned)(wc >> (8 * mi)) & 0xffu, r);
        const int4 bq0 = *(const int4*)(a.bdm + c * CH + 16 * wave + 4 * g);
        const float4 mum = *(const float4*)(a.md + c * CH + 16 * wave + 4 * g);
        const int hb = TXP == 8 ? ((r >> 3) * S * HWX + (r & 7) * S) : r * S;   // window origin of slot r of group 0
        const unsigned char* baseH = E + hb * EST + 16 * wave + g * EST;          // g = column of the tap
        const unsigned char* baseV = E + hb * EST + 16 * wave + g * (HWX * EST);  // g = row of the tap (5x5, column 4)
        auto dw_walk = [&](auto mode_tag) {
          constexpr int RM = decltype(mode_tag)::value;
          const int4 bqm = int4_plus(bq0, RM >= 2 ? RQ_KBIAS : 0);
#pragma unroll
          for (int pg = 0; pg < 4 * PPW; pg += 2) {   // two slot groups at a time: independent accumulate chains
            v4i dqa = v4i_from(bqm), dqb = v4i_from(bqm);
#pragma unroll
            for (int mi = 0; mi < KT64; mi++) {
              const unsigned char* bp = (KK == 5 && mi == 5) ? baseV : baseH;
              const int off = KK == 3 ? mi * HWX * EST : (mi < 5 ? mi * HWX * EST : (mi == 5 ? 4 * EST : (4 * HWX + 4) * EST));
              const v4i bva = *(const v4i*)(bp + pg * PGS + off);
              const v4i bvb = *(const v4i*)(bp + (pg + 1) * PGS + off);
              dqa = __builtin_amdgcn_mfma_i32_16x16x64_i8(wreg[mi], bva, dqa, 0, 0, 0);
              dqb = __builtin_amdgcn_mfma_i32_16x16x64_i8(wreg[mi], bvb, dqb, 0, 0, 0);
            }
            *(unsigned*)(D + (pg * 16 + r) * FB_DST + 16 * wave + 4 * g) = rq_pack_b<RM>(dqa, mum, a.rqd);
            *(unsigned*)(D + ((pg + 1) * 16 + r) * FB_DST + 16 * wave + 4 * g) = rq_pack_b<RM>(dqb, mum, a.rqd);
          }
        };
        rq_dispatch(a.rqd, dw_walk);
      }
    } else if constexpr (MDW) {
      // Matrix-pipe depthwise: out[c][p] = sum_t W'[c][(t,c')] * X[(t,c')][p] with W' = w[t][c] * delta(c,c').
      // One 16x16x32 MFMA covers 2 taps x 16 channels; wave w owns channel group w of the chunk, the B operand
      // (8 consecutive channels of pixel p + tap) is a single ds_read_b64 from the NHWC tile.  Exact int32.
      constexpr int KT = (KK * KK + 1) / 2;
      const unsigned char* Ein = EXPAND ? E : T0 + 64 * c;
      const int est = EXPAND ? EST : a.T0S;
      const long* wm = wdm_ + ((long)(c * NT + wave) * KT) * 64 + lane;
      long wreg[KT];
      int4 bqm = make_int4(0, 0, 0, 0);
      float4 mum = make_float4(0.f, 0.f, 0.f, 0.f);
      const bool cg_active = NT == 4 || wave < NT;   // 48-channel chunks have three channel groups: wave 3 sits this stage out
      if (PREF_DW || (!EXPAND && c == 0)) {
#pragma unroll
        for (int mi = 0; mi < KT; mi++) wreg[mi] = wpre[mi];
        bqm = bpre;
        mum = mpre;
      } else if (cg_active) {
#pragma unroll
        for (int mi = 0; mi < KT; mi++) wreg[mi] = wm[mi * 64];
        bqm = *(const int4*)(a.bdm + c * CH + 16 * wave + 4 * g);
        mum = *(const float4*)(a.md + c * CH + 16 * wave + 4 * g);
      }
      const unsigned char* lane_base = Ein + 16 * wave + 8 * (g & 1);
      const int hi_half = g >> 1;
      // two slot groups at a time: their MFMA chains are independent, so one hides the other's accumulate latency
#pragma unroll
      for (int pg = 0; pg < 4 * PPW; pg += 2) {
        if (!cg_active) break;
        v4i dqa = v4i_from(bqm), dqb = v4i_from(bqm);
        const unsigned char* pba = lane_base + hbase[pg] * est;
        const unsigned char* pbb = lane_base + hbase[pg + 1] * est;
#pragma unroll
        for (int mi = 0; mi < KT; mi++) {
          const int ta = 2 * mi, tb = (2 * mi + 1 < KK * KK) ? 2 * mi + 1 : 2 * mi;  // odd tap count: the pad tap has zero weights
          const int offa = ((ta / KK) * HWx + (ta % KK)) * est, offb = ((tb / KK) * HWx + (tb % KK)) * est;
          const int off = hi_half ? offb : offa;
          long bva = *(const long*)(pba + off);
          long bvb = *(const long*)(pbb + off);
          dqa = __builtin_amdgcn_mfma_i32_16x16x32_i8(wreg[mi], bva, dqa, 0, 0, 0);
          dqb = __builtin_amdgcn_mfma_i32_16x16x32_i8(wreg[mi], bvb, dqb, 0, 0, 0);
        }
        *(unsigned*)(D + (pg * 16 + r) * FB_DST + 16 * wave + 4 * g) = rq_pack_b(dqa, mum, a.rqd);
        *(unsigned*)(D + ((pg + 1) * 16 + r) * FB_DST + 16 * wave + 4 * g) = rq_pack_b(dqb, mum, a.rqd);
      }
    } else if (dw_active) {
      const unsigned char* Ein = EXPAND ? E + 4 * cq : T0 + 64 * c + 4 * cq;
      const int est = EXPAND ? EST : a.T0S;
      constexpr int IW = 3 * S + KK;
      const int Cp = a.nchunks * 64;
      const float* wd = a.wd + c * 64 + 4 * cq;
      float dacc[4][4];
#pragma unroll
      for (int o = 0; o < 4; o++)
#pragma unroll
        for (int j = 0; j < 4; j++) dacc[o][j] = 0.0f;
#pragma unroll
      for (int ky = 0; ky < KK; ky++) {
        float4 wr[KK];
#pragma unroll
        for (int kx = 0; kx < KK; kx++) wr[kx] = *(const float4*)(wd + (long)(ky * KK + kx) * Cp);
        const unsigned char* rowp = Ein + ((sy * S + ky) * HWx + sx * S) * est;
#pragma unroll
        for (int j = 0; j < IW; j++) {
          unsigned u = *(const unsigned*)(rowp + j * est) ^ 0x80808080u;
          float f0 = (float)(u & 255u), f1 = (float)((u >> 8) & 255u), f2 = (float)((u >> 16) & 255u), f3 = (float)(u >> 24);
#pragma unroll
          for (int kx = 0; kx < KK; kx++) {
            if ((j - kx) >= 0 && (j - kx) % S == 0 && (j - kx) / S < 4) {
              const int o = (j - kx) / S;
              dacc[o][0] = __builtin_fmaf(f0, wr[kx].x, dacc[o][0]);
              dacc[o][1] = __builtin_fmaf(f1, wr[kx].y, dacc[o][1]);
              dacc[o][2] = __builtin_fmaf(f2, wr[kx].z, dacc[o][2]);
              dacc[o][3] = __builtin_fmaf(f3, wr[kx].w, dacc[o][3]);
            }
          }
        }
      }
      const int4 bq = *(const int4*)(a.bd + c * 64 + 4 * cq);
      const float4 mu = *(const float4*)(a.md + c * 64 + 4 * cq);
#pragma unroll
      for (int o = 0; o < 4; o++) {
        v4i ai = {(int)dacc[o][0], (int)dacc[o][1], (int)dacc[o][2], (int)dacc[o][3]};
        unsigned dd = rq_pack_i(ai, bq, mu, a.rqd);
        *(unsigned*)(D + (sy * TXp + sx + o) * FB_DST + 4 * cq) = dd;
      }
    }
    __syncthreads();
    // ---- stage P: project accumulation, wave w <- pixel slots 16*PPW*w .. +16*PPW-1, K = this chunk's 64 channels: ONE 16x16x64 MFMA per
    // (slot group, 16-channel tile) and one 16-byte operand read per slot group (two 16x16x32 and two 8-byte reads before) ----
    {
      v4i bv[PPW];
#pragma unroll
      for (int pp = 0; pp < PPW; pp++) bv[pp] = *(const v4i*)(D + ((wave * PPW + pp) * 16 + r) * FB_DST + 16 * g);
#pragma unroll
      for (int nb = 0; nb < NBP; nb++) {
        const v4i* w = stage_p ? (const v4i*)WPS + (nb * 4) * 64 + lane
                               : (NT == 3 ? a.wp3 + ((long)(nb * a.KSp3 + c) * 4) * 64 : a.wp + ((long)(nb * a.KSp + c) * 4) * 64) + lane;
#pragma unroll
        for (int t = 0; t < 4; t++) {
          const v4i wv = w[t * 64];
#pragma unroll
          for (int pp = 0; pp < PPW; pp++) acc[pp][nb][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wv, bv[pp], acc[pp][nb][t], 0, 0, 0);
        }
      }
    }
    if (!EXPAND && c + 1 < nch) __syncthreads();  // D is rewritten by the next chunk's depthwise
  }

  // ---- epilogue: requantise, optional residual ADD with the block input (centre of T0), store ----
#pragma unroll
  for (int pp = 0; pp < PPW; pp++) {
  const int slot = (wave * PPW + pp) * 16 + r;
  const int py = fdiv_small(slot, rcp_txp), px = slot - py * TXp;
  const int oy = oy0 + py, ox = ox0 + px;
  if (py < a.TY && px < a.TX && oy < a.OH && ox < a.OW) {
    const unsigned char* skip = T0 + ((py + a.pad_t) * HWx + (px + a.pad_l)) * a.T0S;  // S == 1 when has_res
    int8_t* orow = a.out + ((b * a.OH + oy) * (long)a.OW + ox) * a.Cout;
#pragma unroll
    for (int nb = 0; nb < NBP; nb++) {
      const int c0 = nb * 64 + 16 * g;
      if (c0 >= a.Cout) continue;
      unsigned d[4];
#pragma unroll
      for (int t = 0; t < 4; t++) {
        int4 bb;
        float4 mu;
        if (stage_p) {
          bb = *(const int4*)(BPS + 4 * (c0 + 4 * t));
          mu = *(const float4*)(BPS + NBP * 256 + 4 * (c0 + 4 * t));
        } else {
          bb = *(const int4*)(a.bp + c0 + 4 * t);
          mu = *(const float4*)(a.mp + c0 + 4 * t);
        }
        unsigned dq = rq_pack_i(acc[pp][nb][t], bb, mu, a.rqp);
        if (a.has_res) {   // Cin == Cout, a multiple of 8: the skip dword is in range whenever c0 + 4t < Cout
          const unsigned xs = *(const unsigned*)(skip + min(c0 + 4 * t, a.Cin - 4));
          dq = addq4(dq, xs, a.resq);
        }
        d[t] = dq;
      }
      int8_t* o = orow + c0;
      if ((a.Cout & 15) == 0) {
        *(uint4*)o = make_uint4(d[0], d[1], d[2], d[3]);
      } else if ((a.Cout & 3) == 0) {
#pragma unroll
        for (int t = 0; t < 4; t++)
          if (c0 + 4 * t < a.Cout) *(unsigned*)(o + 4 * t) = d[t];
      } else {
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
          for (int j = 0; j < 4; j++)
            if (c0 + 4 * t + j < a.Cout) o[4 * t + j] = (int8_t)(d[t] >> (8 * j));
      }
    }
  }
  }
}


#ifndef FB_MINW
#define FB_MINW 1
#endif
template <int KK, int S, int NBP, bool EXPAND, bool MDW, int KSE = 0, int NT = 4, int PPW = 1, bool DW64 = false>
__global__ __launch_bounds__(256, FB_MINW) void fused_block_kernel(FusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fb_smem_dyn[];
  fused_block_body<KK, S, NBP, EXPAND, MDW, KSE, NT, PPW, DW64>(a, blockIdx.x, fb_smem_dyn);
}

// Several independent problems (e.g. the same head layer on all 5 pyramid levels of both heads) in ONE grid:
// a workgroup finds its problem from the cumulative tile counts and runs the same body.
struct MultiTiles {
  int n;
  int start[13];  // start[p] = first workgroup of problem p, start[n] = grid size
};
template <int KK, int S, int NBP, bool EXPAND, bool MDW>
__global__ __launch_bounds__(256, FB_MINW) void fused_block_multi_kernel(const FusedArgs* __restrict__ args, MultiTiles mt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fb_smem_dyn[];
  int p = 0;
#pragma unroll
  for (int i = 1; i < 12; i++)
    if (i < mt.n && (int)blockIdx.x >= mt.start[i]) p = i;
  fused_block_body<KK, S, NBP, EXPAND, MDW>(args[p], (int)blockIdx.x - mt.start[p], fb_smem_dyn);
}

// ------------------------------------------------------------------------------------------
// LDS-tiled depthwise conv (stand-alone): the depthwise stage above with a chunk-parallel grid.
// grid.x = image tiles, grid.y = 64-channel chunks; the halo tile of the chunk is staged in LDS with
// 8-byte coalesced loads, every lane then computes 4 channels x 4 output columns from LDS and stores
// its dwords straight to HBM.  Used where the column walker is a serial latency chain (small maps,
// many channels).
// ------------------------------------------------------------------------------------------
struct DwTileArgs {
  const int8_t* x;
  int8_t* out;
  const float* wf;   // [k*k][C]
  const int* bias;   // folded
  const float* mult;
  const long* wdm;   // matrix-pipe form [chunk][cg][m][lane] (MDW)
  const int* bdm;    // bias folded for raw int8 inputs, padded to 64-channel chunks
  const float* mdm;  // multipliers padded to 64-channel chunks
  int H, W, C, OH, OW, pad_t, pad_l, TX, TY, tiles_x, tiles_y, zx;
  Rq rq;
};

template <int KK, int S, bool MDW>
__global__ __launch_bounds__(256) void dw_tile_kernel(DwTileArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dt_smem[];
  constexpr int TS = 80;  // bytes per halo pixel (64 channels + pad)
  const int tid = threadIdx.x;
  int tile = blockIdx.x;
  const int tx = tile % a.tiles_x;
  tile /= a.tiles_x;
  const int ty = tile % a.tiles_y;
  const long b = tile / a.tiles_y;
  const int c0 = blockIdx.y * 64;
  const int cvalid = min(64, a.C - c0);  // channels of this chunk (multiple of 4)
  const int oy0 = ty * a.TY, ox0 = tx * a.TX;
  const int TXp = (a.TX + 3) & ~3;
  const int HWx = (TXp - 1) * S + KK, HWy = (a.TY - 1) * S + KK, NPh = HWx * HWy;
  const int iy0 = oy0 * S - a.pad_t, ix0 = ox0 * S - a.pad_l;
  {
    const int ng = (cvalid + 7) >> 3;  // 8-byte granules per pixel (C % 8 == 0 is required by the planner)
    const unsigned long long zb = (unsigned long long)(a.zx & 255) * 0x0101010101010101ull;
    const int8_t* xb = a.x + b * (long)a.H * a.W * a.C + c0;
    const int pstep = 256 / ng;
    int p = tid / ng;
    const int sg = tid - p * ng;
    int hy = p / HWx, hx = p - hy * HWx;
    for (; tid < pstep * ng && p < NPh; p += pstep) {
      const int iy = iy0 + hy, ix = ix0 + hx;
      unsigned long long v = zb;
      if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) v = *(const unsigned long long*)(xb + ((long)iy * a.W + ix) * a.C + 8 * sg);
      *(unsigned long long*)(dt_smem + p * TS + 8 * sg) = v;
      hx += pstep;
      while (hx >= HWx) { hx -= HWx; hy++; }
    }
  }
  if constexpr (MDW) {
    // depthwise on the matrix pipe (see fused_block_body): wave = 16-channel group, 4 slot groups of 16 pixels
    constexpr int KT = (KK * KK + 1) / 2;
    const int wave = tid >> 6, lane = tid & 63, r = lane & 15, g = lane >> 4;
    const long* wm = a.wdm + ((long)(blockIdx.y * 4 + wave) * KT) * 64 + lane;
    long wreg[KT];
#pragma unroll
    for (int mi = 0; mi < KT; mi++) wreg[mi] = wm[mi * 64];
    const int cbase = c0 + 16 * wave + 4 * g;
    const int4 bqm = *(const int4*)(a.bdm + cbase);
    const float4 mum = *(const float4*)(a.mdm + cbase);
    const float rcp_txp = 1.0f / (float)TXp;
    __syncthreads();
    if (16 * wave >= cvalid) return;
    const unsigned char* lane_base = dt_smem + 16 * wave + 8 * (g & 1);
    const int hi_half = g >> 1;
#pragma unroll
    for (int pg = 0; pg < 4; pg++) {
      const int slot = pg * 16 + r;
      const int sq = fdiv_small(slot, rcp_txp);
      const int py = min(sq, a.TY - 1), px = slot - sq * TXp;
      const unsigned char* pb = lane_base + ((py * S) * HWx + px * S) * TS;
      v4i dq = v4i_from(bqm);
#pragma unroll
      for (int mi = 0; mi < KT; mi++) {
        const int ta = 2 * mi, tb = (2 * mi + 1 < KK * KK) ? 2 * mi + 1 : 2 * mi;
        const int offa = ((ta / KK) * HWx + (ta % KK)) * TS, offb = ((tb / KK) * HWx + (tb % KK)) * TS;
        long bv = *(const long*)(pb + (hi_half ? offb : offa));
        dq = __builtin_amdgcn_mfma_i32_16x16x32_i8(wreg[mi], bv, dq, 0, 0, 0);
      }
      const int oy = oy0 + sq, ox = ox0 + px;
      if (sq < a.TY && px < a.TX && oy < a.OH && ox < a.OW && cbase < a.C)
        *(unsigned*)(a.out + ((b * a.OH + oy) * (long)a.OW + ox) * a.C + cbase) = rq_pack_b(dq, mum, a.rq);
    }
    return;
  }
  const int cq = tid & 15, strip = tid >> 4;
  const int nsx = TXp >> 2;
  const int sy = strip / nsx, sx = (strip - sy * nsx) * 4;
  const bool active = sy < a.TY && 4 * cq < cvalid;
  // parameters are fetched while the tile loads are in flight
  float4 wr[KK][KK];
  int4 bq = make_int4(0, 0, 0, 0);
  float4 mu = make_float4(0.f, 0.f, 0.f, 0.f);
  if (active) {
#pragma unroll
    for (int ky = 0; ky < KK; ky++)
#pragma unroll
      for (int kx = 0; kx < KK; kx++) wr[ky][kx] = *(const float4*)(a.wf + (long)(ky * KK + kx) * a.C + c0 + 4 * cq);
    bq = *(const int4*)(a.bias + c0 + 4 * cq);
    mu = *(const float4*)(a.mult + c0 + 4 * cq);
  }
  __syncthreads();
  if (!active) return;
  constexpr int IW = 3 * S + KK;
  float dacc[4][4];
#pragma unroll
  for (int o = 0; o < 4; o++)
#pragma unroll
    for (int j = 0; j < 4; j++) dacc[o][j] = 0.0f;
#pragma unroll
  for (int ky = 0; ky < KK; ky++) {
    const unsigned char* rowp = dt_smem + ((sy * S + ky) * HWx + sx * S) * TS + 4 * cq;
#pragma unroll
    for (int j = 0; j < IW; j++) {
      unsigned u = *(const unsigned*)(rowp + j * TS) ^ 0x80808080u;
      float f0 = (float)(u & 255u), f1 = (float)((u >> 8) & 255u), f2 = (float)((u >> 16) & 255u), f3 = (float)(u >> 24);
#pragma unroll
      for (int kx = 0; kx < KK; kx++) {
        if ((j - kx) >= 0 && (j - kx) % S == 0 && (j - kx) / S < 4) {
          const int o = (j - kx) / S;
          dacc[o][0] = __builtin_fmaf(f0, wr[ky][kx].x, dacc[o][0]);
          dacc[o][1] = __builtin_fmaf(f1, wr[ky][kx].y, dacc[o][1]);
          dacc[o][2] = __builtin_fmaf(f2, wr[ky][kx].z, dacc[o][2]);
          dacc[o][3] = __builtin_fmaf(f3, wr[ky][kx].w, dacc[o][3]);
        }
      }
    }
  }
  const int oy = oy0 + sy;
  if (oy < a.OH) {
    int8_t* orow = a.out + ((b * a.OH + oy) * (long)a.OW) * a.C + c0 + 4 * cq;
#pragma unroll
    for (int o = 0; o < 4; o++) {
      const int ox = ox0 + sx + o;
      if (sx + o < a.TX && ox < a.OW) {
        const v4i ai = {(int)dacc[o][0], (int)dacc[o][1], (int)dacc[o][2], (int)dacc[o][3]};
        *(unsigned*)(orow + (long)ox * a.C) = rq_pack_i(ai, bq, mu, a.rq);
      }
    }
  }
}
