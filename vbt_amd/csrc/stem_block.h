// Fused network entry: stem 3x3/2 conv on the uint8 frame -> depthwise 3x3/1 -> project 1x1, one kernel.
//
// The two largest activation tensors of the network (stem output 160x160x32 and the first depthwise output) never
// reach HBM: a workgroup reads a 37x37 RGB patch and writes a 16x16x16 output tile.  Bit-identical to the three
// graph ops run one by one (same int32 accumulations, same float requantisation).
//
//   R  [37][120]    raw uint8 rows of the patch (dword-aligned start, out-of-image dwords = input zero point)
//   P  [336][32]    per stem pixel of the 18x18 halo: kernel row 0|1|2 as 8 bytes each (px0 RGB px1 RGB px2 RG)
//                   + the three px2-B bytes: exactly the K=32 operand of one 16x16x32 int8 MFMA
//   S  [324][48]    stem output on the halo, int8 x 32 channels (out-of-map pixels = its zero point: the
//                   depthwise pads ITS input); 48-byte rows: 16-byte aligned channel groups, conflict-free b128 reads
//   D  [256][40]    depthwise output, int8 x 32 channels
// stem / project: weights are the MFMA A operand so a lane ends with consecutive output channels of one pixel;
// depthwise: diagonal-embedded weights on the 16x16x64 MFMA, tap (row m, column g) per instruction m and lane group g
// (three instructions for 3x3, every LDS address = lane base + immediate; see DW64 in fused_block.h).
#pragma once

struct StemBlockArgs {
  const uint8_t* frames;
  int8_t* out;
  int H, W, SH, SW, Cout;
  int spad_t, spad_l;   // SAME padding of the stem
  int tiles_x, tiles_y;
  unsigned in_pad4;     // out-of-image input byte (uint8 domain, zero point + 128) x 4
  const long* ws;       // stem weights [t(2)][lane] x 8 B; row i of tile t = cout 8(i>>2) + 4t + (i&3)
  const int* bs;        // [32] folded bias
  const float* ms;      // [32]
  Rq rqs;
  unsigned zs4;         // stem output zero point x 4
  const v4i* wd64;      // depthwise [cg(2)][3][lane] x 16 B (FusedArgs::wd64 layout)
  const int* bdm;
  const float* mdm;
  Rq rqd;
  const long* wp;       // project [lane] x 8 B (row i = cout i, K = 32)
  const int* bp;        // [16] folded
  const float* mp;      // [16]
  Rq rqp;
};

constexpr int SB_HW = 18, SB_NPH = SB_HW * SB_HW, SB_NPG = (SB_NPH + 15) / 16;
constexpr int SB_RROWS = 37, SB_RDW = 30, SB_RST = SB_RDW * 4, SB_PST = 32, SB_SST = 48, SB_DST = 40;

// MODE 1: all three requantisations clamp at the int8 limits (the saturating flavour, no per-element branch); MODE 3: and all
// three convs' accumulators are proven to stay inside +-2^22 (Rq::kb: accumulators start at bias + RQ_KBIAS); MODE 0: read at run time
template <int MODE>
__global__ __launch_bounds__(256) void stem_block_kernel(StemBlockArgs a) {
  constexpr int FK = MODE == 0 ? -1 : MODE;
  constexpr int KB = MODE >= 2 ? RQ_KBIAS : 0;
  // R is dead once P is built and P once S is built: R shares S's storage, D shares P's (23.8 KB -> 6 workgroups / CU)
  __shared__ __attribute__((aligned(16))) unsigned char SR[SB_NPH * SB_SST + 64];
  __shared__ __attribute__((aligned(16))) unsigned char PD[SB_NPG * 16 * SB_PST];
  static_assert(SB_RROWS * SB_RST <= SB_NPH * SB_SST && 256 * SB_DST <= SB_NPG * 16 * SB_PST, "aliased LDS regions");
  unsigned char* const R = SR;
  unsigned char* const S = SR;
  unsigned char* const P = PD;
  unsigned char* const D = PD;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 15, g = lane >> 4;
  const int tile = blockIdx.x;
  const int trow = fdiv_small(tile, frcp(a.tiles_x));
  const int tx = tile - trow * a.tiles_x;
  const int bimg = fdiv_small(trow, frcp(a.tiles_y));
  const int ty = trow - bimg * a.tiles_y;
  const long b = bimg;
  const int oy0 = ty * 16, ox0 = tx * 16;
  const int sy0 = oy0 - 1, sx0 = ox0 - 1;                      // depthwise 3x3/1 SAME: one halo pixel
  const int iy0 = 2 * sy0 - a.spad_t, ix0 = 2 * sx0 - a.spad_l;
  const int rb0 = ix0 * 3;
  const int al = rb0 & 3;                                        // two's complement: also right for negative rb0
  const int rstart = rb0 - al;                                   // multiple of 4
  const int rowbytes = a.W * 3;                                  // multiple of 4 (planner)

  // ---- 1a: raw patch rows -> R (aligned dwords) ----
  {
    const uint8_t* f = a.frames + b * (long)a.H * rowbytes;
    for (int idx = tid; idx < SB_RROWS * SB_RDW; idx += 256) {
      const int row = idx / SB_RDW, d = idx - row * SB_RDW;
      const int iy = iy0 + row, off = rstart + 4 * d;
      unsigned v = a.in_pad4;
      if (iy >= 0 && iy < a.H && off >= 0 && off < rowbytes) v = *(const unsigned*)(f + (long)iy * rowbytes + off);
      *(unsigned*)(R + row * SB_RST + 4 * d) = v;
    }
  }
  __syncthreads();
  // ---- 1b: MFMA operand slots: (stem pixel, kernel row) -> 9 bytes, u8 -> s8 by XOR 0x80 ----
  for (int slot = tid; slot < SB_NPH * 3; slot += 256) {
    const int p = slot / 3, ky = slot - 3 * p;
    const int hy = p / SB_HW, hx = p - hy * SB_HW;
    const int o = al + 6 * hx;
    const unsigned char* src = R + (2 * hy + ky) * SB_RST + (o & ~3);
    const unsigned d0 = *(const unsigned*)src, d1 = *(const unsigned*)(src + 4), d2 = *(const unsigned*)(src + 8);
    const unsigned sh = (unsigned)(o & 3);
    const unsigned w0 = __builtin_amdgcn_alignbyte(d1, d0, sh) ^ 0x80808080u;
    const unsigned w1 = __builtin_amdgcn_alignbyte(d2, d1, sh) ^ 0x80808080u;
    const unsigned b8 = ((d2 >> (8 * sh)) & 255u) ^ 0x80u;
    *(uint2*)(P + p * SB_PST + 8 * ky) = make_uint2(w0, w1);
    P[p * SB_PST + 24 + ky] = (unsigned char)b8;
  }
  __syncthreads();
  // ---- 2: stem conv on the 18x18 halo, 2 MFMAs per 16 pixels (32 output channels) ----
  {
    const long wa0 = a.ws[lane], wa1 = a.ws[64 + lane];
    const int4 b0 = int4_plus(*(const int4*)(a.bs + 8 * g), KB), b1 = int4_plus(*(const int4*)(a.bs + 8 * g + 4), KB);
    const float4 m0 = *(const float4*)(a.ms + 8 * g), m1 = *(const float4*)(a.ms + 8 * g + 4);
    // halo pixels outside the stem's output map (tiles on the map border only): bit i <-> pixel group wave + 4i
    unsigned oob_mask = 0;
    if (sy0 < 0 || sx0 < 0 || sy0 + SB_HW > a.SH || sx0 + SB_HW > a.SW) {
      for (int i = 0, pg = wave; pg < SB_NPG; pg += 4, i++) {
        const int pc = min(pg * 16 + r, SB_NPH - 1);
        const int hy = pc / SB_HW, hx = pc - hy * SB_HW;
        const int sy = sy0 + hy, sx = sx0 + hx;
        if (!(sy >= 0 && sy < a.SH && sx >= 0 && sx < a.SW)) oob_mask |= 1u << i;
      }
    }
    for (int i = 0, pg = wave; pg < SB_NPG; pg += 4, i++) {
      const int p = pg * 16 + r;
      const int pc = min(p, SB_NPH - 1);
      const long bv = *(const long*)(P + pc * SB_PST + 8 * g);
      v4i a0 = v4i_from(b0), a1 = v4i_from(b1);
      a0 = __builtin_amdgcn_mfma_i32_16x16x32_i8(wa0, bv, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_i32_16x16x32_i8(wa1, bv, a1, 0, 0, 0);
      unsigned q0 = rq_pack_b<FK>(a0, m0, a.rqs), q1 = rq_pack_b<FK>(a1, m1, a.rqs);
      if ((oob_mask >> i) & 1u) { q0 = a.zs4; q1 = a.zs4; }
      if (p < SB_NPH) *(uint2*)(S + p * SB_SST + 8 * g) = make_uint2(q0, q1);
    }
  }
  __syncthreads();
  // ---- 3: depthwise 3x3/1 on the 16x16x64 MFMA: wave -> channel group (wave & 1), output rows 8(wave>>1).. ----
  {
    const int cg = wave & 1;
    v4i wreg[3];
#pragma unroll
    for (int mi = 0; mi < 3; mi++) wreg[mi] = a.wd64[(cg * 3 + mi) * 64 + lane];
    const int4 bqm = int4_plus(*(const int4*)(a.bdm + 16 * cg + 4 * g), KB);
    const float4 mum = *(const float4*)(a.mdm + 16 * cg + 4 * g);
    // lane (r = output column, g = tap column): B operand of instruction m for output row py = 16 channels of halo pixel
    // (py + m, r + g); g = 3 carries zero weights (it reads one pixel past the window: still inside S)
    const unsigned char* lane_base = S + (r + g) * SB_SST + 16 * cg + (wave >> 1) * 8 * SB_HW * SB_SST;
#pragma unroll
    for (int i = 0; i < 8; i += 2) {   // two output rows at a time: independent accumulate chains
      v4i dqa = v4i_from(bqm), dqb = v4i_from(bqm);
#pragma unroll
      for (int mi = 0; mi < 3; mi++) {
        dqa = __builtin_amdgcn_mfma_i32_16x16x64_i8(wreg[mi], *(const v4i*)(lane_base + (i + mi) * SB_HW * SB_SST), dqa, 0, 0, 0);
        dqb = __builtin_amdgcn_mfma_i32_16x16x64_i8(wreg[mi], *(const v4i*)(lane_base + (i + 1 + mi) * SB_HW * SB_SST), dqb, 0, 0, 0);
      }
      const int py = (wave >> 1) * 8 + i;
      *(unsigned*)(D + (py * 16 + r) * SB_DST + 16 * cg + 4 * g) = rq_pack_b<FK>(dqa, mum, a.rqd);
      *(unsigned*)(D + ((py + 1) * 16 + r) * SB_DST + 16 * cg + 4 * g) = rq_pack_b<FK>(dqb, mum, a.rqd);
    }
  }
  __syncthreads();
  // ---- 4: project 32 -> Cout (<= 16): one MFMA per 16 pixels, lane -> 4 consecutive channels of pixel r ----
  {
    const long wa = a.wp[lane];
    const int4 bb = int4_plus(*(const int4*)(a.bp + 4 * g), KB);
    const float4 mm = *(const float4*)(a.mp + 4 * g);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int py = wave * 4 + i;
      const long bv = *(const long*)(D + (py * 16 + r) * SB_DST + 8 * g);
      v4i acc = v4i_from(bb);
      acc = __builtin_amdgcn_mfma_i32_16x16x32_i8(wa, bv, acc, 0, 0, 0);
      const unsigned q = rq_pack_b<FK>(acc, mm, a.rqp);
      const int oy = oy0 + py, ox = ox0 + r;
      if (oy < a.SH && ox < a.SW && 4 * g < a.Cout) *(unsigned*)(a.out + ((b * a.SH + oy) * (long)a.SW + ox) * a.Cout + 4 * g) = q;
    }
  }
}
