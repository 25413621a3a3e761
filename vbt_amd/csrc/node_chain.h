// BiFPN node chain: several consecutive fused BiFPN nodes ({resample}* -> n-ary ADD + ReLU6 -> depthwise 3x3 -> pointwise)
// on small maps (<= 20x20) executed by ONE workgroup per image, node after node.
//
// The nodes of a BiFPN cell form a strict dependency chain on maps of 3x3 ... 20x20 pixels; as separate launches each
// costs a launch boundary (5-10 us) for ~1 us of work.  Here a 16-wave workgroup owns one image and runs the chain:
// per node, the summed input lands in a zero-point-bordered LDS image (SAME padding for free, no halo tiles), the
// depthwise runs on the matrix pipe from that image, the pointwise accumulates over 64-channel chunks and the result
// is written to the node's HBM tensor; later nodes of the chain read it back through L2 after a fence + barrier.
// Arithmetic and parameters are those of fused_block.h's node path (same FusedArgs), so results are identical.
#pragma once

constexpr int NC_WAVES = 16, NC_THREADS = 64 * NC_WAVES;
constexpr int NC_MAXU = 4;  // project units (16 pixels x 64 channels) per wave: 25 pixel groups x 2 blocks / 16 waves

#ifdef VBT_DEFINE_CHAIN_KERNELS   // not a template: defined by k_image.hip only
__global__ __launch_bounds__(NC_THREADS) void node_chain_kernel(const FusedArgs* __restrict__ nodes, int n_nodes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char nc_smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 15, g = lane >> 4;
  const long b = blockIdx.x;
  for (int ni = 0; ni < n_nodes; ni++) {
    const FusedArgs& a = nodes[ni];
    const int H = a.H, W = a.W, HW = H * W, C = a.Cin, Cp = a.nchunks * 64;
    const int PW = W + 2, PH = H + 2, TS = Cp + 16;       // padded image, bytes per pixel
    const int NPG = (HW + 15) >> 4, NB = (a.Cout + 63) >> 6;
    unsigned char* T = nc_smem;
    unsigned char* D = T + PH * PW * TS;
    const float rcp_w = frcp(W), rcp_pw = frcp(PW), rcp_nb = frcp(NB);
    // ---- sum stage: T interior <- binary integer ADD(s) of the resampled sources (node_sum4), border <- zx; 4 channels per lane-iteration ----
    {
      const int nd = C >> 2, ndp = TS >> 2;                // dwords per pixel: real channels / whole LDS row
      const unsigned zb4 = (unsigned)(a.zx & 255) * 0x01010101u;
      const bool up2[3] = {H == 2 * a.sh[0] && W == 2 * a.sw[0], H == 2 * a.sh[1] && W == 2 * a.sw[1], H == 2 * a.sh[2] && W == 2 * a.sw[2]};
      const float rcp_ndp = frcp(ndp);
      for (int i = tid; i < PH * PW * ndp; i += NC_THREADS) {
        const int p = fdiv_small(i, rcp_ndp), cd = i - p * ndp;
        const int py = fdiv_small(p, rcp_pw), px = p - py * PW;
        const int iy = py - 1, ix = px - 1;
        unsigned v = zb4;
        if (cd < nd && iy >= 0 && iy < H && ix >= 0 && ix < W) {
          unsigned us[3] = {0u, 0u, 0u};
#pragma unroll
          for (int j = 0; j < 3; j++) {
            if (j < a.n_src) {
              const int8_t* sb = a.src[j] + b * (long)a.sh[j] * a.sw[j] * C + 4 * cd;
              unsigned u;
              if (a.smode[j] == 0) {
                u = *(const unsigned*)(sb + ((long)iy * a.sw[j] + ix) * C);
              } else if (a.smode[j] == 1) {
                int yy, xx;
                if (up2[j]) { yy = iy >> 1; xx = ix >> 1; }
                else { yy = (iy * a.sh[j]) / H; xx = (ix * a.sw[j]) / W; }
                u = *(const unsigned*)(sb + ((long)yy * a.sw[j] + xx) * C);
              } else {
                int m0 = -128, m1 = -128, m2 = -128, m3 = -128;
                for (int ky = 0; ky < 3; ky++) {
                  const int yy = iy * 2 + ky - a.spt[j];
                  if (yy < 0 || yy >= a.sh[j]) continue;
                  for (int kx = 0; kx < 3; kx++) {
                    const int xx = ix * 2 + kx - a.spl[j];
                    if (xx < 0 || xx >= a.sw[j]) continue;
                    const unsigned t = *(const unsigned*)(sb + ((long)yy * a.sw[j] + xx) * C);
                    m0 = max(m0, (int)(int8_t)(t & 255u)); m1 = max(m1, (int)(int8_t)((t >> 8) & 255u));
                    m2 = max(m2, (int)(int8_t)((t >> 16) & 255u)); m3 = max(m3, (int)(int8_t)(t >> 24));
                  }
                }
                u = pack4(m0, m1, m2, m3);
              }
              us[j] = u;
            }
          }
          v = node_sum4(us, a);
        }
        *(unsigned*)(T + p * TS + 4 * cd) = v;
      }
    }
    v4i acc[NC_MAXU][4];
#pragma unroll
    for (int i = 0; i < NC_MAXU; i++)
#pragma unroll
      for (int t = 0; t < 4; t++) acc[i][t] = (v4i){0, 0, 0, 0};
    const int NUP = NPG * NB;
    __syncthreads();
    constexpr int KT = 5;
    for (int c = 0; c < a.nchunks; c++) {
      // ---- depthwise 3x3/1 chunk c on the matrix pipe: unit = (pixel group, channel group wave & 3) ----
      {
        const int cg = wave & 3;
        const long* wm = a.wdm + ((long)(c * 4 + cg) * KT) * 64 + lane;
        long wreg[KT];
#pragma unroll
        for (int mi = 0; mi < KT; mi++) wreg[mi] = wm[mi * 64];
        const int4 bqm = *(const int4*)(a.bdm + c * 64 + 16 * cg + 4 * g);
        const float4 mum = *(const float4*)(a.md + c * 64 + 16 * cg + 4 * g);
        const int hi_half = g >> 1;
        for (int pg = wave >> 2; pg < NPG; pg += NC_WAVES / 4) {
          const int slot = pg * 16 + r, sc = min(slot, HW - 1);
          const int oy = fdiv_small(sc, rcp_w), ox = sc - oy * W;
          const unsigned char* pb = T + (oy * PW + ox) * TS + 64 * c + 16 * cg + 8 * (g & 1);
          v4i dq = v4i_from(bqm);
#pragma unroll
          for (int mi = 0; mi < KT; mi++) {
            const int ta = 2 * mi, tb = (2 * mi + 1 < 9) ? 2 * mi + 1 : 2 * mi;
            const int offa = ((ta / 3) * PW + (ta % 3)) * TS, offb = ((tb / 3) * PW + (tb % 3)) * TS;
            dq = __builtin_amdgcn_mfma_i32_16x16x32_i8(wreg[mi], *(const long*)(pb + (hi_half ? offb : offa)), dq, 0, 0, 0);
          }
          *(unsigned*)(D + slot * FB_DST + 16 * cg + 4 * g) = rq_pack_b(dq, mum, a.rqd);
        }
      }
      __syncthreads();
      // ---- pointwise: K = this chunk's 64 channels, unit = (pixel group, 64-channel output block) ----
#pragma unroll
      for (int i = 0; i < NC_MAXU; i++) {
        const int u = wave + NC_WAVES * i;
        if (u < NUP) {
          const int pg = fdiv_small(u, rcp_nb), nb = u - pg * NB;
#pragma unroll
          for (int k2 = 0; k2 < 2; k2++) {
            const long bv = *(const long*)(D + (pg * 16 + r) * FB_DST + 32 * k2 + 8 * g);
            const long* w = a.wp + ((long)(nb * a.KSp + 2 * c + k2) * 4) * 64 + lane;
#pragma unroll
            for (int t = 0; t < 4; t++) acc[i][t] = __builtin_amdgcn_mfma_i32_16x16x32_i8(w[t * 64], bv, acc[i][t], 0, 0, 0);
          }
        }
      }
      __syncthreads();  // D is rewritten by the next chunk / T by the next node
    }
    // ---- epilogue: 16 channels per lane -> the node's HBM tensor ----
#pragma unroll
    for (int i = 0; i < NC_MAXU; i++) {
      const int u = wave + NC_WAVES * i;
      if (u >= NUP) continue;
      const int pg = fdiv_small(u, rcp_nb), nb = u - pg * NB;
      const int slot = pg * 16 + r;
      const int c0 = nb * 64 + 16 * g;
      if (slot >= HW || c0 >= a.Cout) continue;
      unsigned d[4];
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const int4 bb = *(const int4*)(a.bp + c0 + 4 * t);
        const float4 mu = *(const float4*)(a.mp + c0 + 4 * t);
        d[t] = rq_pack_i(acc[i][t], bb, mu, a.rqp);
      }
      int8_t* o = a.out + (b * HW + slot) * (long)a.Cout + c0;
      if ((a.Cout & 15) == 0) {
        *(uint4*)o = make_uint4(d[0], d[1], d[2], d[3]);
      } else {
#pragma unroll
        for (int t = 0; t < 4; t++)
          if (c0 + 4 * t < a.Cout) *(unsigned*)(o + 4 * t) = d[t];
      }
    }
    __threadfence();   // the next node of this image reads this tensor back through L2
    __syncthreads();
  }
}
#endif  // VBT_DEFINE_CHAIN_KERNELS
