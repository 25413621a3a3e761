// Low-resolution MBConv, first half (expand 1x1 + ReLU6 -> depthwise kxk, stride 1 or 2), second form of expdw_block.h.  Same split
// of the work - workgroup (image b, row band, channel group) produces `cpw` 64-channel chunks of the depthwise output, the
// projection runs afterwards as a pointwise GEMM - and the same arithmetic (int32 accumulation, one float requantisation per
// element), but organised around what the stage stamps of tools/probes/xd_probe.hip showed for the first form: its stages did not
// overlap at all (LDS operand reads 40 %, MFMA 20 %, requantisation 15 %, and a third of a chunk's time in per-unit address
// arithmetic, exposed parameter loads and three barriers), and the depthwise read every expanded element k*k times from LDS.
//
//   * The block input never goes through LDS.  A wave owns fixed pixel groups; their expand B operands (16 pixels x 64 input
//     channels per MFMA) are loaded from global memory into registers once and serve every chunk.  No T0 tile (45 KB at
//     20x20x112), no LDS reads and no address arithmetic in the expand stage.
//   * Depthwise as a band-Toeplitz matrix product instead of a diagonal one.  The expanded chunk lives in LDS QUAD-PLANAR:
//     E[row][channel quad][x][4 channels].  One MFMA multiplies A[(dx, c)][(ty, x', c')] = w[ty][x' - dx][c] delta(c, c')
//     (4 output columns dx x 4 channels c; K = 2 kernel rows x 8 input columns x 4 channels) with B = 16 positions (output row,
//     block of 4 output columns), one aligned 16-byte LDS read per lane: 10 taps per output and instruction instead of 4, i.e.
//     2 MFMAs + 2 reads per 256 outputs for 3x3 (was 3) and 3 for 5x5 (was 7).  A lane ends with the 4 channels of one output
//     pixel: one dword store.
//   * The depthwise output chunk is staged quad-planar too (D[quad][pixel][4]): conflict-free dword stores, and the copy to
//     the tensor gathers a pixel's 16 channels with four dword reads.
//   * Two barriers per chunk instead of three: the copy-out of chunk c runs in the same interval as the expand of chunk c + 1.
//   * A chunk's parameters (expand weights, depthwise weights, biases, multipliers: 21-25 KB) are fetched ONCE per workgroup, three
//     16-byte loads per thread issued a stage ahead, and handed to the waves through LDS.  Per-wave loads of the same data cost 22
//     vector-memory instructions per wave and chunk - 8 waves x 22 x 16 address cycles = 2800 cycles of the CU's one address path,
//     the fixed cost the stage stamps showed in every interval.
//   * LDS: E 39 KB + D 26 KB + parameters 21-25 KB at 20x20 (the first form: 120 KB).  One workgroup per CU either way; the 16-wave shape
//     (below) puts four waves on every SIMD, so that one wave's LDS / MFMA waits are covered by three others.
//   * Stride 2: the operand covers 2 x 2 output pixels x 4 channels (rows 4 py .. 4 py + k + 1 of the expanded image, 8 columns from 4 px:
//     3 MFMAs per 256 outputs for 3x3, 4 for 5x5 - the diagonal form takes 3 / 7), positions are (output row pair, output column pair).
// Inputs whose channel count is not a multiple of 8 stay on expdw_block.h.
#pragma once
#include <type_traits>

#ifdef VBT_XD_PROF
#define XD2_STAMP(k) do { if (tid == 0) a.prof[(long)blockIdx.x * 32 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define XD2_STAMP(k) do { } while (0)
#endif


struct ExpDw2Args {
  const int8_t* x;   // [B][H][W][Cin]
  int8_t* out;       // [B][OH][OW][Ce]: the graph's depthwise output tensor
  int H, W, Cin, OH, OW, Ce;
  int pad_t, pad_l;
  int nchunks, cpw;  // 64-channel chunks in all / per workgroup
  int nbands, brows; // row bands per image / output rows per band
  int XB;            // position columns per row: blocks of 4 output columns (stride 1) / of 2 (stride 2)
  int EQS, EYS;      // E: bytes per (row, quad) = 4 * padded width rounded to 16; bytes per row = 16 * EQS + bank-spreading pad
  int e_bytes;       // LDS bytes of E (tallest band, + slack for the reads past the last row)
  int PS;            // D: bytes per quad plane (4 * pixels of the tallest band, rounded so that PS / 4 = 2 mod 32)
  int pe_off;        // LDS byte offsets of the parameter areas Pe / Pd (behind D)
  int pd_off;
  // per chunk, 16-byte pieces, copied to LDS as they are:
  const v4i* pe;     // expand: weights [ks][t][lane] x 16 B (row i of tile t = channel 64 c + 16 t + i, k = 64 ks + 16 g + j) | bias (input zero
                     // point folded) x 64 | multipliers x 64                                          = KS64 * 256 + 32 pieces
  const v4i* pd;     // depthwise: [quad][mi][lane] dwords, byte j = w[2 mi + (g >> 1) - S dy][4 (g & 1) + j - S dx][64 c + 4 quad + cc] for A row
                     // (lane & 15) = 4 q + cc with q = dx (stride 1: dy = 0) or 2 dy + dx (stride 2); 0 outside the kernel; the MFMA operand
                     // (dword j = that byte at byte position cc) is rebuilt in registers | bias (expanded zero point folded) x 64 |
                     // multipliers x 64                                                                = KT2 * 256 + 32 pieces
  Rq rqe;
  unsigned zeb;      // zero point of the expanded tensor x4
  Rq rqd;
#ifdef VBT_XD_PROF
  unsigned long long* prof;
#endif
};

// A operand of the band-Toeplitz depthwise MFMA from its four bytes: dword j holds byte j at byte position c = row & 3
__device__ __forceinline__ v4i toeplitz_operand(unsigned w4, int c) {
  const int sh = 8 * c;
  return (v4i){(int)((w4 & 0xffu) << sh), (int)(((w4 >> 8) & 0xffu) << sh), (int)(((w4 >> 16) & 0xffu) << sh), (int)((w4 >> 24) << sh)};
}

constexpr int XD2_NPG = 8;   // position groups a band may have (128 positions = 512 output pixels)
template <int PG, class F, class FC>
__device__ __forceinline__ void d_pair(F& d_units, FC full_c, int NPGo) {
  if constexpr (PG < XD2_NPG) {
    if (PG + 1 < NPGo) d_units(full_c, std::integral_constant<int, 2>{}, std::integral_constant<int, PG>{});
    else if (PG < NPGo) d_units(full_c, std::integral_constant<int, 1>{}, std::integral_constant<int, PG>{});
    if (PG + 2 < NPGo) d_pair<PG + 2>(d_units, full_c, NPGo);
  }
}

// KK / S: depthwise kernel size / stride; KS64: 64-channel K steps of the expand; NW: waves per workgroup; GPW: input pixel groups a wave
// owns (>= ceil(pixel groups / (NW / 2))).
// NW = 8: a wave expands a pair of 16-channel tiles on up to 7 pixel groups and runs the depthwise of two channel quads - 180-240
// registers, two waves per SIMD.  NW = 16: a tile pair on up to 4 pixel groups, one quad - the compiler has to stay within 128
// registers, four waves per SIMD: the vector ALU issues a wave-instruction every 2.8 cycles instead of 3.5
// (tools/probes/valu_rate.hip) and a wave's LDS / MFMA waits are covered by three others instead of one.
template <int KK, int S, int KS64, int NW, int GPW>
__global__ __launch_bounds__(64 * NW) void expdw2_kernel(ExpDw2Args a) {
  constexpr int DY = S, DX = 4 / S;   // output pixels of a depthwise position: 1 x 4 (stride 1), 2 x 2 (stride 2)
  constexpr int XD2_THREADS = 64 * NW, HW2 = NW / 2, QW = 16 / NW;   // threads; waves per tile pair; channel quads per wave in the depthwise
  extern __shared__ __attribute__((aligned(16))) unsigned char xd2_smem[];
  constexpr int KT2 = (S * (DY - 1) + KK + 1) / 2;   // depthwise MFMAs per unit: two rows of the expanded image each
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  XD2_STAMP(0);
  const int ngroups = fdiv_small(a.nchunks + a.cpw - 1, frcp(a.cpw));
  const int per_image = ngroups * a.nbands;
  const long b = fdiv_small((int)blockIdx.x, frcp(per_image));
  const int rem = blockIdx.x - (int)b * per_image;
  const int band = fdiv_small(rem, frcp(ngroups));
  const int grp = rem - band * ngroups;
  const int oy0 = band * a.brows, oy1 = min(oy0 + a.brows, a.OH), OHb = oy1 - oy0;
  const int iy_lo = max(oy0 * S - a.pad_t, 0), iy_hi = min(oy0 * S - a.pad_t + (OHb - 1) * S + KK, a.H);
  const int erow0 = iy_lo + a.pad_t - oy0 * S;                   // E row of input row iy_lo
  const int HW = (iy_hi - iy_lo) * a.W, OHW = OHb * a.OW;        // pixels of the band: input / output
  const int NPOS = ((OHb + DY - 1) / DY) * a.XB, NPGo = (NPOS + 15) >> 4;   // depthwise positions (DY rows x DX columns of outputs) and their groups of 16
  unsigned char* E = xd2_smem;
  unsigned char* D = E + a.e_bytes;
  const int c_first = grp * a.cpw, c_last = min(c_first + a.cpw, a.nchunks);

  // ---- expand stage, per-lane constants: wave = (tile pair tp, pixel group phase wq); lane (r, g) = pixel r of a group, K slice g ----
  const int tp = wave / HW2, wq = wave % HW2;
  v4i xin[GPW][KS64];
  int eoff[GPW];   // E byte offset of this lane's dword for tile 2 tp (channel quad 8 tp + g), -1: no such pixel
  {
    const int8_t* xb = a.x + (b * (long)a.H + iy_lo) * a.W * a.Cin;
    const float rcp_w = frcp(a.W);
#pragma unroll
    for (int i = 0; i < GPW; i++) {
      const int p = (wq + HW2 * i) * 16 + r, pc = min(p, HW - 1);
#pragma unroll
      for (int ks = 0; ks < KS64; ks++) {
        // a K slice lies inside the pixel's channels, straddles their end (Cin % 16 == 8: the bytes past it are the next pixel's - or
        // the arena's slack - and meet zero weights) or lies wholly in the zero-weight padding (not loaded); 8-byte aligned when Cin % 16 == 8
        const int k0 = 64 * ks + 16 * g;
        v4i v = {0, 0, 0, 0};
        if (k0 < a.Cin) __builtin_memcpy(&v, xb + (long)pc * a.Cin + k0, 16);
        xin[i][ks] = v;
      }
      const int py = fdiv_small(pc, rcp_w), px = pc - py * a.W;
      eoff[i] = p < HW ? (py + erow0) * a.EYS + (px + a.pad_l) * 4 + (8 * tp + g) * a.EQS : -1;
    }
  }
  // ---- parameters of a chunk: global -> registers (a stage ahead) -> LDS -> the waves that need them ----
  constexpr int NE = KS64 * 256 + 32, ND = KT2 * 256 + 32;                     // 16-byte pieces per chunk
  constexpr int NPE = (NE + XD2_THREADS - 1) / XD2_THREADS, NPD = (ND + XD2_THREADS - 1) / XD2_THREADS;
  unsigned char* Pe = xd2_smem + a.pe_off;
  unsigned char* Pd = xd2_smem + a.pd_off;
  v4i pe_r[NPE], pd_r[NPD];
  // No load is issued under a condition (indices are clamped instead) and none stays in flight across the copy-out's stores: gfx9
  // counts loads and stores in one vmcnt, and a conditional prefetch or a store loop between a load and its use makes the compiler's
  // waits conservative - the stage then waits for loads it has only just issued.
  auto fetch_params = [&](int c) {
#pragma unroll
    for (int k = 0; k < NPD; k++) pd_r[k] = a.pd[(long)c * ND + min(tid + k * XD2_THREADS, ND - 1)];
#pragma unroll
    for (int k = 0; k < NPE; k++) pe_r[k] = a.pe[(long)c * NE + min(tid + k * XD2_THREADS, NE - 1)];
  };
  auto put_pe = [&]() {
#pragma unroll
    for (int k = 0; k < NPE; k++) *(v4i*)(Pe + 16 * min(tid + k * XD2_THREADS, NE - 1)) = pe_r[k];
  };
  auto put_pd = [&]() {
#pragma unroll
    for (int k = 0; k < NPD; k++) *(v4i*)(Pd + 16 * min(tid + k * XD2_THREADS, ND - 1)) = pd_r[k];
  };
  fetch_params(c_first);
  asm volatile("" ::: "memory");   // the compiler otherwise sinks loads to their first use and waits for them there
  const int cq0 = QW * wave;       // depthwise stage: this wave's channel quads are cq0 .. cq0 + QW - 1
  // E <- zero point: border, rows outside the image and the padding columns keep it (the expand writes real pixels only)
  {
    const uint4 z4 = make_uint4(a.zeb, a.zeb, a.zeb, a.zeb);
    for (int i = tid; i < (a.e_bytes >> 4); i += XD2_THREADS) *(uint4*)(E + 16 * i) = z4;
  }
  // ---- depthwise stage, per-lane constants: wave owns channel quads 2 wave, 2 wave + 1; lane (r, g) = position r of a group, K slice g ----
  // E byte offset of this lane's operand (kernel rows 0 / 1, column half g & 1, quad cq0) and D byte offset of its output dword for
  // every position group, once per kernel: the stage itself then has no address arithmetic (it was a quarter of its vector instructions)
  unsigned dofs[XD2_NPG];   // low half: E offset (< 64 KB); high half: D offset (< 64 KB), 0xffff: no such output pixel
  {
    const int gofs = (g >> 1) * a.EYS + 16 * (g & 1) + cq0 * a.EQS;
    const float rcp_xb = frcp(a.XB);
#pragma unroll
    for (int pg = 0; pg < XD2_NPG; pg++) {
      dofs[pg] = 0xffff0000u;
      if (pg >= NPGo) continue;   // (uniform: a 10x10 map has two position groups, not eight)
      const int n = pg * 16 + r, nc = min(n, NPOS - 1);
      const int y = fdiv_small(nc, rcp_xb), xk = nc - y * a.XB;   // position (row, column); its first input row / column: S DY y, S DX xk = 4 xk
      const int oy = DY * y + (S == 2 ? g >> 1 : 0), ox = DX * xk + (S == 2 ? g & 1 : g);
      dofs[pg] = (unsigned)(S * DY * y * a.EYS + xk * 16 + gofs) |
                 ((n < NPOS && oy < OHb && ox < a.OW) ? (unsigned)((oy * a.OW + ox) * 4 + cq0 * a.PS) << 16 : 0xffff0000u);
    }
  }
  // ---- copy-out: D (quad-planar) -> the depthwise output tensor, 16 bytes per lane ----
  auto copy_out = [&](int c) {
    const int nv = min(64, a.Ce - 64 * c) >> 4;   // 16-byte parts of this chunk (Ce % 16 == 0)
    const float rcp_nv = frcp(nv);
    int8_t* ob = a.out + (b * (long)a.OH + oy0) * a.OW * a.Ce + 64 * c;
    for (int i = tid; i < OHW * nv; i += XD2_THREADS) {
      const int slot = fdiv_small(i, rcp_nv), part = i - slot * nv;
      const unsigned char* s = D + 4 * part * a.PS + 4 * slot;
      *(uint4*)(ob + (long)slot * a.Ce + 16 * part) =
          make_uint4(*(const unsigned*)s, *(const unsigned*)(s + a.PS), *(const unsigned*)(s + 2 * a.PS), *(const unsigned*)(s + 3 * a.PS));
    }
  };
  put_pe();
  __syncthreads();
  XD2_STAMP(1);
  for (int c = c_first; c < c_last; c++) {
    // Interval A: depthwise parameters of chunk c -> LDS (every wave took chunk c - 1's out of Pd at the start of its depthwise stage,
    // before the barrier behind us); the next chunk's parameters requested; the previous chunk's output leaves; expand of chunk c.
    put_pd();
    fetch_params(min(c + 1, c_last - 1));   // two intervals ahead of put_pe: a depthwise stage can be shorter than a round trip to memory
    asm volatile("" ::: "memory");
    if (c > c_first) copy_out(c - 1);
    // ---- stage E: expand chunk c; unit = (pixel group i, tile tt of the pair); every B operand is in registers ----
    {
      auto e_stage = [&](auto full_c) {
        constexpr int FULL = decltype(full_c)::value;
#pragma unroll
        for (int tt = 0; tt < 2; tt++) {
          // this tile's weights / bias / multipliers out of Pe (one tile at a time: registers - the 16-wave form lives within 128)
          v4i ew[KS64];
#pragma unroll
          for (int ks = 0; ks < KS64; ks++) ew[ks] = *(const v4i*)(Pe + 16 * ((ks * 4 + 2 * tp + tt) * 64 + lane));
          const int4 eb = int4_plus(*(const int4*)(Pe + 16 * (KS64 * 256) + 4 * (16 * (2 * tp + tt) + 4 * g)), FULL >= 2 ? RQ_KBIAS : 0);
          const float4 em = *(const float4*)(Pe + 16 * (KS64 * 256 + 16) + 4 * (16 * (2 * tp + tt) + 4 * g));
          v4i acc[GPW];
#pragma unroll
          for (int i = 0; i < GPW; i++) acc[i] = v4i_from(eb);
#pragma unroll
          for (int ks = 0; ks < KS64; ks++)
#pragma unroll
            for (int i = 0; i < GPW; i++) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ew[ks], xin[i][ks], acc[i], 0, 0, 0);
#pragma unroll
          for (int i = 0; i < GPW; i++) {
            const unsigned val = rq_pack_b<FULL>(acc[i], em, a.rqe);
            if (eoff[i] >= 0) *(unsigned*)(E + eoff[i] + tt * 4 * a.EQS) = val;
          }
        }
      };
      rq_dispatch(a.rqe, e_stage);
    }
    __syncthreads();   // E complete, D free
    XD2_STAMP(2 + 2 * (c - c_first));
    // ---- stage D: depthwise on chunk c; unit = (position group, quad q of the pair) ----
    {
      // Interval B: this wave's depthwise parameters out of Pd, depthwise of chunk c, then the next chunk's expand parameters -> Pe
      // (requested in interval A; the expand is done with Pe).
      v4i dwa[QW][KT2];
      int4 bq[QW];
      float4 mq[QW];
#pragma unroll
      for (int q = 0; q < QW; q++) {
#pragma unroll
        for (int mi = 0; mi < KT2; mi++) dwa[q][mi] = toeplitz_operand(*(const unsigned*)(Pd + 4 * (((cq0 + q) * KT2 + mi) * 64 + lane)), r & 3);
        bq[q] = *(const int4*)(Pd + 16 * (KT2 * 256) + 16 * (cq0 + q));
        mq[q] = *(const float4*)(Pd + 16 * (KT2 * 256 + 16) + 16 * (cq0 + q));
      }
      auto d_units = [&](auto full_c, auto u_c, auto pg_c) {
        constexpr int U = decltype(u_c)::value, pg0 = decltype(pg_c)::value;
        constexpr int FULL = decltype(full_c)::value;
        const unsigned char* base[U];
        int doff[U];
#pragma unroll
        for (int u = 0; u < U; u++) { base[u] = E + (dofs[pg0 + u] & 0xffffu); doff[u] = (int)(dofs[pg0 + u] >> 16); }
        v4i bv[U][QW][KT2];
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
          for (int q = 0; q < QW; q++)
#pragma unroll
            for (int mi = 0; mi < KT2; mi++) bv[u][q][mi] = *(const v4i*)(base[u] + q * a.EQS + mi * 2 * a.EYS);
        v4i acc[U][QW];
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
          for (int q = 0; q < QW; q++) acc[u][q] = v4i_from(int4_plus(bq[q], FULL >= 2 ? RQ_KBIAS : 0));
#pragma unroll
        for (int mi = 0; mi < KT2; mi++)
#pragma unroll
          for (int u = 0; u < U; u++)
#pragma unroll
            for (int q = 0; q < QW; q++) acc[u][q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(dwa[q][mi], bv[u][q][mi], acc[u][q], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < U; u++) {
          unsigned v[QW];
#pragma unroll
          for (int q = 0; q < QW; q++) v[q] = rq_pack_b<FULL>(acc[u][q], mq[q], a.rqd);
          if (doff[u] != 0xffff) {
#pragma unroll
            for (int q = 0; q < QW; q++) *(unsigned*)(D + doff[u] + q * a.PS) = v[q];
          }
        }
      };
      auto d_walk = [&](auto full_c) {   // position groups in pairs (two groups x QW quads: independent MFMA chains), unrolled: register-indexed offsets
        d_pair<0>(d_units, full_c, NPGo);
      };
      rq_dispatch(a.rqd, d_walk);
    }
    put_pe();
    __syncthreads();   // D complete, E free for the next chunk's expand, its expand parameters in Pe
    XD2_STAMP(3 + 2 * (c - c_first));
  }
  copy_out(c_last - 1);
  XD2_STAMP(2 + 2 * (c_last - c_first));
}
