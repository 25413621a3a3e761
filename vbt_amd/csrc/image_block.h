// Whole-image MBConv: expand 1x1 + ReLU6 -> depthwise kxk (matrix pipe) -> project 1x1 [-> residual ADD] with ONE
// workgroup per image, for the low-resolution backbone blocks (20x20 and 10x10 maps).
//
// The tile kernel (fused_block.h) recomputes the expand on every tile's halo; on a 10x10 map with a 5x5 depthwise that
// is 2.9x the useful work, most of it on padding pixels.  Here the tile IS the image: the expand runs on the real
// pixels only and lands in a zero-point-bordered copy of the image in LDS, so SAME padding costs nothing.
//   T0 [H*W][T0S]      block input (also the residual source)
//   E  [PH*PW][80]     one 64-channel chunk of the expanded tensor inside a border of its zero point
//   D  [NPGo*16][72]   one 64-channel chunk of the depthwise output
//   WB [bundle]        every weight / bias / multiplier the current chunk needs ("bundle", built on the host as one
//                      contiguous record per chunk).  One workgroup per CU cannot hide L2 latency behind other
//                      workgroups, so the next chunk's bundle is fetched into registers while this chunk computes and
//                      dropped into WB at the chunk boundary; every MFMA operand then comes from LDS.
// 16 wavefronts; work units are dealt round-robin: expand (pixel group, 16-channel tile), depthwise (output pixel
// group, 16-channel group = wave & 3), project (output pixel group, 64-channel block) whose int32 accumulators stay in
// registers across chunks.  Depthwise weights travel compact ([tap][64] bytes); the diagonal MFMA operand is rebuilt in
// registers once per chunk.  Same integer / float arithmetic as the tile kernel: identical results.
#pragma once

constexpr int IB_WAVES = 16, IB_THREADS = 64 * IB_WAVES, IB_NPF = 3;  // NPF uint4 prefetch registers per lane (48 KB bundles)

struct ImageBundle {       // byte offsets inside one chunk's record
  const unsigned char* data;
  int bytes;               // record size, multiple of 16
  int o_be, o_me, o_dw, o_bd, o_md, o_wp;
};

template <int KK, int S, int MAXU>
__global__ __launch_bounds__(IB_THREADS) void mbconv_image_kernel(FusedArgs a, ImageBundle wb, int PW, int PH, int NB) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ib_smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 15, g = lane >> 4;
  const long b = blockIdx.x;
  const int HW = a.H * a.W, OHW = a.OH * a.OW;
  const int NPGi = (HW + 15) >> 4, NPGo = (OHW + 15) >> 4;
  unsigned char* T0 = ib_smem;
  unsigned char* E = T0 + ((HW * a.T0S + 15) & ~15);
  unsigned char* D = E + PH * PW * FB_EST;
  unsigned char* WB = D + NPGo * 16 * FB_DST;
  const float rcp_w = 1.0f / (float)a.W, rcp_ow = 1.0f / (float)a.OW;
  const int nv = wb.bytes >> 4;  // uint4 per bundle (<= IB_NPF * IB_THREADS)

  // ---- input image -> T0 (8-byte granules), E <- zero point everywhere (the border keeps it), bundle 0 -> WB ----
  {
    const int ng = a.Cin >> 3;
    const int8_t* xb = a.x + b * (long)HW * a.Cin;
    for (int i = tid; i < HW * ng; i += IB_THREADS) {
      const int p = i / ng, sg = i - p * ng;
      *(unsigned long long*)(T0 + p * a.T0S + 8 * sg) = *(const unsigned long long*)(xb + (long)p * a.Cin + 8 * sg);
    }
    const unsigned zeb = (unsigned)(a.ze & 255) * 0x01010101u;
    const uint4 z4 = make_uint4(zeb, zeb, zeb, zeb);
    for (int i = tid; i < PH * PW * (FB_EST / 16); i += IB_THREADS) *(uint4*)(E + 16 * i) = z4;
    for (int i = tid; i < nv; i += IB_THREADS) *(uint4*)(WB + 16 * i) = *(const uint4*)(wb.data + 16 * (long)i);
  }
  v4i acc[MAXU][4];
#pragma unroll
  for (int i = 0; i < MAXU; i++)
#pragma unroll
    for (int t = 0; t < 4; t++) acc[i][t] = (v4i){0, 0, 0, 0};
  const int NUP = NPGo * NB;
  __syncthreads();

  constexpr int KT = (KK * KK + 1) / 2;
  for (int c = 0; c < a.nchunks; c++) {
    // next chunk's bundle: global -> registers now, registers -> WB after this chunk's last read of WB
    uint4 pf[IB_NPF];
    const bool more = c + 1 < a.nchunks;
    if (more) {
      const unsigned char* src = wb.data + (long)(c + 1) * wb.bytes;
#pragma unroll
      for (int j = 0; j < IB_NPF; j++) {
        const int i = tid + j * IB_THREADS;
        if (i < nv) pf[j] = *(const uint4*)(src + 16 * (long)i);
      }
    }
    // ---- expand chunk c: unit = (pixel group, 16-row weight tile t) -> 4 channels x 16 pixels per lane ----
    for (int u = wave; u < NPGi * 4; u += IB_WAVES) {
      const int pg = u >> 2, t = u & 3;
      const int p = pg * 16 + r, pc = min(p, HW - 1);
      v4i ea = v4i_from(*(const int4*)(WB + wb.o_be + 4 * (16 * g + 4 * t)));
      const float4 em = *(const float4*)(WB + wb.o_me + 4 * (16 * g + 4 * t));
      const unsigned char* brow = T0 + pc * a.T0S + 8 * g;
      const unsigned char* w = WB + (t * 64 + lane) * 8;
#pragma unroll 2
      for (int ks = 0; ks < a.KSe; ks++) ea = __builtin_amdgcn_mfma_i32_16x16x32_i8(*(const long*)(w + ks * 2048), *(const long*)(brow + 32 * ks), ea, 0, 0, 0);
      const unsigned d = rq_pack_b(ea, em, a.rqe);
      if (p < HW) {
        const int py = fdiv_small(p, rcp_w), px = p - py * a.W;
        *(unsigned*)(E + ((py + a.pad_t) * PW + px + a.pad_l) * FB_EST + 16 * g + 4 * t) = d;
      }
    }
    __syncthreads();
    // ---- depthwise chunk c on the matrix pipe: unit = (output pixel group, channel group wave & 3) ----
    {
      const int cg = wave & 3;
      long wreg[KT];
      const bool diag = (r >> 3) == (g & 1);
      const int sh = 8 * (r & 7);
#pragma unroll
      for (int mi = 0; mi < KT; mi++) {
        const int tap = 2 * mi + (g >> 1);
        const unsigned v = (tap < KK * KK && diag) ? (unsigned)WB[wb.o_dw + tap * 64 + 16 * cg + r] : 0u;
        wreg[mi] = (long)((unsigned long long)v << sh);
      }
      const int4 bqm = *(const int4*)(WB + wb.o_bd + 4 * (16 * cg + 4 * g));
      const float4 mum = *(const float4*)(WB + wb.o_md + 4 * (16 * cg + 4 * g));
      const int hi_half = g >> 1;
      for (int pgo = wave >> 2; pgo < NPGo; pgo += IB_WAVES / 4) {
        const int slot = pgo * 16 + r, sc = min(slot, OHW - 1);
        const int oy = fdiv_small(sc, rcp_ow), ox = sc - oy * a.OW;
        const unsigned char* pb = E + ((oy * S) * PW + ox * S) * FB_EST + 16 * cg + 8 * (g & 1);
        v4i dq = v4i_from(bqm);
#pragma unroll
        for (int mi = 0; mi < KT; mi++) {
          const int ta = 2 * mi, tb = (2 * mi + 1 < KK * KK) ? 2 * mi + 1 : 2 * mi;
          const int offa = ((ta / KK) * PW + (ta % KK)) * FB_EST, offb = ((tb / KK) * PW + (tb % KK)) * FB_EST;
          dq = __builtin_amdgcn_mfma_i32_16x16x32_i8(wreg[mi], *(const long*)(pb + (hi_half ? offb : offa)), dq, 0, 0, 0);
        }
        *(unsigned*)(D + slot * FB_DST + 16 * cg + 4 * g) = rq_pack_b(dq, mum, a.rqd);
      }
    }
    __syncthreads();
    // ---- project: K = this chunk's 64 channels ----
#pragma unroll
    for (int i = 0; i < MAXU; i++) {
      const int u = wave + IB_WAVES * i;
      if (u < NUP) {
        const int pgo = u / NB, nb = u - pgo * NB;
#pragma unroll
        for (int k2 = 0; k2 < 2; k2++) {
          const long bv = *(const long*)(D + (pgo * 16 + r) * FB_DST + 32 * k2 + 8 * g);
          const unsigned char* w = WB + wb.o_wp + (((nb * 2 + k2) * 4) * 64 + lane) * 8;
#pragma unroll
          for (int t = 0; t < 4; t++) acc[i][t] = __builtin_amdgcn_mfma_i32_16x16x32_i8(*(const long*)(w + t * 512), bv, acc[i][t], 0, 0, 0);
        }
      }
    }
    if (more) {
      __syncthreads();  // every wave is done with WB (and D)
#pragma unroll
      for (int j = 0; j < IB_NPF; j++) {
        const int i = tid + j * IB_THREADS;
        if (i < nv) *(uint4*)(WB + 16 * i) = pf[j];
      }
      __syncthreads();
    }
  }

  // ---- epilogue: requantise (+ residual ADD with the block input), 16 channels per lane ----
#pragma unroll
  for (int i = 0; i < MAXU; i++) {
    const int u = wave + IB_WAVES * i;
    if (u >= NUP) continue;
    const int pgo = u / NB, nb = u - pgo * NB;
    const int slot = pgo * 16 + r;
    const int c0 = nb * 64 + 16 * g;
    if (slot >= OHW || c0 >= a.Cout) continue;
    const unsigned char* skip = T0 + slot * a.T0S;  // S == 1 when has_res: same pixel
    unsigned d[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
      const int4 bb = *(const int4*)(a.bp + c0 + 4 * t);
      const float4 mu = *(const float4*)(a.mp + c0 + 4 * t);
      const unsigned dq = rq_pack_i(acc[i][t], bb, mu, a.rqp);
      d[t] = a.has_res ? addq4(dq, *(const unsigned*)(skip + min(c0 + 4 * t, a.Cin - 4)), a.resq) : dq;
    }
    int8_t* o = a.out + (b * OHW + slot) * (long)a.Cout + c0;
    if ((a.Cout & 15) == 0) {
      *(uint4*)o = make_uint4(d[0], d[1], d[2], d[3]);
    } else {
#pragma unroll
      for (int t = 0; t < 4; t++)
        if (c0 + 4 * t < a.Cout) *(unsigned*)(o + 4 * t) = d[t];
    }
  }
}
