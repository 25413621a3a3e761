// Single-op kernels of the int8 detector (one graph op per launch): pointwise convs on the 16x16x64 int8 MFMA (four forms), stem conv,
// depthwise convs (row / column walkers), integer ADD, max pool, nearest-neighbour resize, decode + NMS, bilinear frame resize.
// Included by detector.hip inside namespace vbt, after dev_common.h; the planner, the autotuner and the C ABI stay in detector.hip, the
// fused kernel families in their own headers / translation units (launchers.h).
#pragma once

// ------------------------------------------------------------------------------------------
// pointwise conv on the gfx950 double-rate int8 MFMA (v_mfma_i32_16x16x64_i8: same 16 issue cycles as the legacy
// 16x16x32 form for twice the K; measured in tools/probes).  A operand = packed weights, B operand = 16 input channels
// of one pixel per lane (one 16-byte load).  K is padded to a multiple of 64 with zero weights: the activation bytes
// read beyond a pixel's K channels (the next pixel, or the arena slack) multiply zeros.
// wp = packed weights [nb][ks][t][lane] x 16 bytes (pack_weights64).
// Epilogue: requantisation, optionally followed by the block's residual ADD (integer, XNNPACK qs8-vadd) with `res`.
// ------------------------------------------------------------------------------------------
struct ResArgs {
  const int8_t* res;  // nullptr: no residual; else the second ADD input, same [M][N] layout as the output
  AddQ q;
};
__device__ __forceinline__ void store_tile_r(const v4i acc[4], const Epi& e, const ResArgs& ra, int8_t* __restrict__ out, long m, int N,
                                             int nb, int g) {
  int c0 = nb * 64 + 16 * g;
  if (c0 >= N) return;
  unsigned d[4];
#pragma unroll
  for (int t = 0; t < 4; t++) {
    int4 b = *(const int4*)(e.bias + c0 + 4 * t);
    float4 mu = *(const float4*)(e.mult + c0 + 4 * t);
    d[t] = rq_pack_i(acc[t], b, mu, e.rq);
  }
  if (ra.res) {   // N % 8 == 0 for every residual block
    const int8_t* r = ra.res + m * N + c0;
#pragma unroll
    for (int t = 0; t < 4; t++)
      if (c0 + 4 * t < N) d[t] = addq4(d[t], *(const unsigned*)(r + 4 * t), ra.q);
  }
  int8_t* o = out + m * N + c0;
  if ((N & 15) == 0) {
    *(uint4*)o = make_uint4(d[0], d[1], d[2], d[3]);
  } else if ((N & 3) == 0) {
#pragma unroll
    for (int t = 0; t < 4; t++)
      if (c0 + 4 * t < N) *(unsigned*)(o + 4 * t) = d[t];
  } else {
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (c0 + 4 * t + j < N) o[4 * t + j] = (int8_t)(d[t] >> (8 * j));
  }
}
// epilogue operands of one lane's 16 output channels (bias, multipliers, residual bytes): requested BEFORE the K loop so
// that their latency overlaps the weight / activation streams instead of following the last MFMA
struct EpiRegs {
  int4 b[4];
  float4 mu[4];
  unsigned res[4];
};
__device__ __forceinline__ void load_epi(EpiRegs& er, const Epi& e, const ResArgs& ra, long m, int N, int nb, int g) {
  const int c0 = min(nb * 64 + 16 * g, ((N + 15) & ~15) - 16);   // (bias / mult arrays are padded to 64-channel blocks)
#pragma unroll
  for (int t = 0; t < 4; t++) {
    er.b[t] = *(const int4*)(e.bias + c0 + 4 * t);
    er.mu[t] = *(const float4*)(e.mult + c0 + 4 * t);
    er.res[t] = 0u;
  }
  if (ra.res) {
    const int8_t* r = ra.res + m * N + nb * 64 + 16 * g;
#pragma unroll
    for (int t = 0; t < 4; t++)
      if (nb * 64 + 16 * g + 4 * t < N) er.res[t] = *(const unsigned*)(r + 4 * t);
  }
}
__device__ __forceinline__ void store_tile_e(const v4i acc[4], const EpiRegs& er, const Epi& e, const ResArgs& ra, int8_t* __restrict__ out,
                                             long m, int N, int nb, int g) {
  int c0 = nb * 64 + 16 * g;
  if (c0 >= N) return;
  unsigned d[4];
#pragma unroll
  for (int t = 0; t < 4; t++) {
    d[t] = rq_pack_i(acc[t], er.b[t], er.mu[t], e.rq);
    if (ra.res) d[t] = addq4(d[t], er.res[t], ra.q);
  }
  int8_t* o = out + m * N + c0;
  if ((N & 15) == 0) {
    *(uint4*)o = make_uint4(d[0], d[1], d[2], d[3]);
  } else if ((N & 3) == 0) {
#pragma unroll
    for (int t = 0; t < 4; t++)
      if (c0 + 4 * t < N) *(unsigned*)(o + 4 * t) = d[t];
  } else {
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (c0 + 4 * t + j < N) o[4 * t + j] = (int8_t)(d[t] >> (8 * j));
  }
}
__device__ __forceinline__ v4i ld16(const int8_t* p) {  // 16 bytes, any 4-byte alignment
  v4i v;
  __builtin_memcpy(&v, p, 16);
  return v;
}

// variant A: K <= 256: the activations of 16*MS pixels stay in registers while the wave walks over the channel blocks
template <int KS, int MS>
__global__ __launch_bounds__(256) void pw_a_kernel(const int8_t* __restrict__ x, const v4i* __restrict__ wp, Epi e, ResArgs ra,
                                                   int8_t* __restrict__ out, long M, int K, int N, int NB,
                                                   int nb_per_y) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = lane & 15, g = lane >> 4;
  const long m0 = ((long)blockIdx.x * 4 + wave) * (16 * MS);
  if (m0 >= M) return;
  v4i a[MS][KS];
#pragma unroll
  for (int ms = 0; ms < MS; ms++) {
    long m = min(m0 + 16 * ms + r, M - 1);
    const int8_t* p = x + m * K + 16 * g;
#pragma unroll
    for (int ks = 0; ks < KS; ks++) a[ms][ks] = ld16(p + 64 * ks);
  }
  const int nb0 = blockIdx.y * nb_per_y, nb1 = min(nb0 + nb_per_y, NB);
  for (int nb = nb0; nb < nb1; nb++) {
    v4i acc[MS][4];
#pragma unroll
    for (int ms = 0; ms < MS; ms++)
#pragma unroll
      for (int t = 0; t < 4; t++) acc[ms][t] = (v4i){0, 0, 0, 0};
    const v4i* w = wp + (long)nb * KS * 4 * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < KS; ks++)
#pragma unroll
      for (int t = 0; t < 4; t++) {
        v4i wv = w[(ks * 4 + t) * 64];
#pragma unroll
        for (int ms = 0; ms < MS; ms++)
          acc[ms][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wv, a[ms][ks], acc[ms][t], 0, 0, 0);
      }
#pragma unroll
    for (int ms = 0; ms < MS; ms++) {
      long m = m0 + 16 * ms + r;
      if (m < M) store_tile_r(acc[ms], e, ra, out, m, N, nb, g);
    }
  }
}

// variant B: large K, few output channels: accumulators for NBT channel blocks stay in registers
// while the wave streams the K dimension of its 16 pixels.
template <int NBT>
__global__ __launch_bounds__(256) void pw_b_kernel(const int8_t* __restrict__ x, const v4i* __restrict__ wp, Epi e, ResArgs ra,
                                                   int8_t* __restrict__ out, long M, int K, int KS, int N, int NB) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = lane & 15, g = lane >> 4;
  const long m0 = ((long)blockIdx.x * 4 + wave) * 16;
  if (m0 >= M) return;
  const int nb0 = blockIdx.y * NBT;
  v4i acc[NBT][4];
#pragma unroll
  for (int i = 0; i < NBT; i++)
#pragma unroll
    for (int t = 0; t < 4; t++) acc[i][t] = (v4i){0, 0, 0, 0};
  const int8_t* p = x + min(m0 + r, M - 1) * K + 16 * g;
  EpiRegs er[NBT];
#pragma unroll
  for (int i = 0; i < NBT; i++) load_epi(er[i], e, ra, min(m0 + r, M - 1), N, min(nb0 + i, NB - 1), g);
#pragma unroll 4
  for (int ks = 0; ks < KS; ks++) {  // unrolled so that the loads of several k-steps are in flight together
    v4i av = ld16(p + 64 * ks);
#pragma unroll
    for (int i = 0; i < NBT; i++) {
      int nb = min(nb0 + i, NB - 1);
      const v4i* w = wp + ((long)(nb * KS + ks) * 4) * 64 + lane;
#pragma unroll
      for (int t = 0; t < 4; t++) acc[i][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w[t * 64], av, acc[i][t], 0, 0, 0);
    }
  }
  long m = m0 + r;
  if (m < M) {
#pragma unroll
    for (int i = 0; i < NBT; i++)
      if (nb0 + i < NB) store_tile_e(acc[i], er[i], e, ra, out, m, N, nb0 + i, g);
  }
}

// variant C: large K on FEW pixels (low-resolution project convs): the 4 waves of a workgroup share the same
// 16 pixels and split K in four; partial accumulators meet in LDS, then each wave requantises one 16-channel
// tile of every 64-channel block.  Quarter-length serial K loop, 4x the workgroups of variant B.
template <int NBT, int MS>
__global__ __launch_bounds__(256) void pw_c_kernel(const int8_t* __restrict__ x, const v4i* __restrict__ wp, Epi e, ResArgs ra,
                                                   int8_t* __restrict__ out, long M, int K, int KS, int N, int NB) {
  // MS pixel groups per workgroup share every weight operand a wave loads: the weights are 4/5 of the bytes this kernel pulls
  // through L1 (4 KB of weights against 1 KB of activations per K-step and pixel group), and L1 is what it saturates
  __shared__ v4i red[4][MS * NBT * 4][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = lane & 15, g = lane >> 4;
  const long m0 = (long)blockIdx.x * (16 * MS);
  const int nb0 = blockIdx.y * NBT;
  v4i acc[MS][NBT][4];
#pragma unroll
  for (int s = 0; s < MS; s++)
#pragma unroll
    for (int i = 0; i < NBT; i++)
#pragma unroll
      for (int t = 0; t < 4; t++) acc[s][i][t] = (v4i){0, 0, 0, 0};
  const int8_t* p[MS];
#pragma unroll
  for (int s = 0; s < MS; s++) p[s] = x + min(m0 + 16 * s + r, M - 1) * K + 16 * g;
  // this wave's epilogue operands (tile t = wave of every block), requested before the K loop
  int4 eb[NBT];
  float4 em[NBT];
  unsigned eres[MS][NBT];
#pragma unroll
  for (int i = 0; i < NBT; i++) {
    const int c0 = min((nb0 + i) * 64 + 16 * g + 4 * wave, NB * 64 - 4);
    eb[i] = *(const int4*)(e.bias + c0);
    em[i] = *(const float4*)(e.mult + c0);
#pragma unroll
    for (int s = 0; s < MS; s++)
      eres[s][i] = (ra.res && (N & 3) == 0 && c0 < N) ? *(const unsigned*)(ra.res + min(m0 + 16 * s + r, M - 1) * N + c0) : 0u;
  }
  const int per = (KS + 3) >> 2;
  const int k0 = wave * per, k1 = min(k0 + per, KS);
#pragma unroll 3
  for (int ks = k0; ks < k1; ks++) {
    v4i av[MS];
#pragma unroll
    for (int s = 0; s < MS; s++) av[s] = ld16(p[s] + 64 * ks);
#pragma unroll
    for (int i = 0; i < NBT; i++) {
      int nb = min(nb0 + i, NB - 1);
      const v4i* w = wp + ((long)(nb * KS + ks) * 4) * 64 + lane;
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const v4i wv = w[t * 64];
#pragma unroll
        for (int s = 0; s < MS; s++) acc[s][i][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wv, av[s], acc[s][i][t], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int s = 0; s < MS; s++)
#pragma unroll
    for (int i = 0; i < NBT; i++)
#pragma unroll
      for (int t = 0; t < 4; t++) red[wave][(s * NBT + i) * 4 + t][lane] = acc[s][i][t];
  __syncthreads();
  // wave w finishes tile t = w of every block: lane -> 4 channels (64 nb + 16 g + 4 t + j) of pixel r
#pragma unroll
  for (int s = 0; s < MS; s++) {
    const long m = m0 + 16 * s + r;
#pragma unroll
    for (int i = 0; i < NBT; i++) {
      const int nb = nb0 + i, t = wave;
      const int c0 = nb * 64 + 16 * g + 4 * t;
      if (nb < NB && c0 < N && m < M) {
        v4i sum = red[0][(s * NBT + i) * 4 + t][lane];
#pragma unroll
        for (int w2 = 1; w2 < 4; w2++) {
          v4i o = red[w2][(s * NBT + i) * 4 + t][lane];
          sum[0] += o[0]; sum[1] += o[1]; sum[2] += o[2]; sum[3] += o[3];
        }
        unsigned d = rq_pack_i(sum, eb[i], em[i], e.rq);
        if (ra.res && (N & 3) == 0) d = addq4(d, eres[s][i], ra.q);
        int8_t* o = out + m * N + c0;
        if ((N & 3) == 0) *(unsigned*)o = d;
        else
          for (int j = 0; j < 4; j++)
            if (c0 + j < N) o[j] = (int8_t)(d >> (8 * j));
      }
    }
  }
}

// variant D: large K, weights shared through LDS.  A workgroup owns 64 * MS pixels x one 64-channel block over the whole K; per
// K-step each wave fetches ONE 16-byte weight operand per lane (a quarter of the block's 4 KB) into a double-buffered LDS copy and
// its own MS activation operands into registers, a step ahead, then reads the four weight tiles back from LDS (256 B/clk/CU
// against the 64 B/clk of the vector-memory path) for 4 * MS MFMAs.  Variants B and C pull every weight operand through the
// vector-memory path once per wave - 4 loads of 1 KB per K-step and wave, 16 address cycles each on the CU's one address unit -
// which is what they saturate (tools/probes: MFMA 5 % busy, three quarters of the wave-cycles waiting).  One barrier per K-step;
// no load is issued under a condition (the prefetch index is clamped) so that the compiler's vmcnt waits stay exact.
template <int MS>
__global__ __launch_bounds__(256) void pw_d_kernel(const int8_t* __restrict__ x, const v4i* __restrict__ wp, Epi e, ResArgs ra,
                                                   int8_t* __restrict__ out, long M, int K, int KS, int N, int NB) {
  __shared__ v4i wbuf[2][4][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = lane & 15, g = lane >> 4;
  const long m0 = ((long)blockIdx.x * 4 + wave) * (16 * MS);
  const int nb = blockIdx.y;
  const int8_t* p[MS];
  EpiRegs er[MS];
#pragma unroll
  for (int s = 0; s < MS; s++) {
    const long mc = min(m0 + 16 * s + r, M - 1);
    p[s] = x + mc * K + 16 * g;
    load_epi(er[s], e, ra, mc, N, nb, g);
  }
  v4i acc[MS][4];
#pragma unroll
  for (int s = 0; s < MS; s++)
#pragma unroll
    for (int t = 0; t < 4; t++) acc[s][t] = (v4i){0, 0, 0, 0};
  const v4i* w = wp + ((long)nb * KS * 4 + wave) * 64 + lane;   // this wave's tile of K-step 0; a K-step is 4 * 64 operands further
  v4i wreg = w[0];
  v4i a_cur[MS], a_nxt[MS];
#pragma unroll
  for (int s = 0; s < MS; s++) a_cur[s] = ld16(p[s]);
  wbuf[0][wave][lane] = wreg;
  __syncthreads();
  for (int ks = 0; ks < KS; ks++) {
    const int kn = min(ks + 1, KS - 1);
    wreg = w[(long)kn * 4 * 64];
#pragma unroll
    for (int s = 0; s < MS; s++) a_nxt[s] = ld16(p[s] + 64 * kn);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int t = 0; t < 4; t++) {
      const v4i wv = wbuf[ks & 1][t][lane];
#pragma unroll
      for (int s = 0; s < MS; s++) acc[s][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wv, a_cur[s], acc[s][t], 0, 0, 0);
    }
    wbuf[(ks + 1) & 1][wave][lane] = wreg;
#pragma unroll
    for (int s = 0; s < MS; s++) a_cur[s] = a_nxt[s];
    __syncthreads();
  }
#pragma unroll
  for (int s = 0; s < MS; s++) {
    const long m = m0 + 16 * s + r;
    if (m < M) store_tile_e(acc[s], er[s], e, ra, out, m, N, nb, g);
  }
}

// variant E: the whole weight panel of the workgroup's 64-channel block - KS x 4 KB, 44 KB for K = 672, 72 KB for K = 1152 - goes to LDS ONCE, with
// every load of a thread in flight together, then ONE barrier; the K loop has no barrier at all: per step a wave reads its four weight tiles from
// LDS (4 x 1 KB at 128 B/clk) and issues 4 x MS MFMAs, with the activation operands PD steps ahead in registers.  Variant D meets at a
// workgroup barrier once per K-step - 11 to 18 times per launch, four waves each time - and with one to three workgroups per CU nothing fills
// those waits (rocprofv3: 9-10 us per launch for 2-4 GOP, the matrix pipe 5 % busy).  Measured (profiles/r05_ab.md 6): per launch the 20 x 20
// projections gain 0-1.5 us and the 10 x 10 ones lose 2 us when timed alone, yet with three forwards in flight the plan that runs b6-b15 on this
// variant is +0.6 % end to end in every round of the A/B - so the plan uses it, and the launches' 10 us are not their K loop (a variant that kept
// the barrier but requested its loads four steps ahead measured the same).
template <int MS, int PD>
__global__ __launch_bounds__(256) void pw_e_kernel(const int8_t* __restrict__ x, const v4i* __restrict__ wp, Epi e, ResArgs ra,
                                                   int8_t* __restrict__ out, long M, int K, int KS, int N, int NB) {
  extern __shared__ __attribute__((aligned(16))) unsigned char pwe_smem[];
  v4i* wpanel = (v4i*)pwe_smem;   // [KS][4][64]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 15, g = lane >> 4;
  const long m0 = ((long)blockIdx.x * 4 + wave) * (16 * MS);
  const int nb = blockIdx.y;
  // 1. the panel: KS x 256 operands of 16 bytes, thread t takes t, t + 256, ... - six at a time in flight
  const v4i* wsrc = wp + (long)nb * KS * 4 * 64;
  const int total = KS * 256;
  for (int i0 = tid; i0 < total; i0 += 6 * 256) {
    v4i v[6];
#pragma unroll
    for (int k = 0; k < 6; k++) v[k] = wsrc[min(i0 + k * 256, total - 1)];
#pragma unroll
    for (int k = 0; k < 6; k++)
      if (i0 + k * 256 < total) wpanel[i0 + k * 256] = v[k];
  }
  // 2. this wave's epilogue operands and the first PD activation operands, requested before the barrier
  const int8_t* p[MS];
  EpiRegs er[MS];
#pragma unroll
  for (int s = 0; s < MS; s++) {
    const long mc = min(m0 + 16 * s + r, M - 1);
    p[s] = x + mc * K + 16 * g;
    load_epi(er[s], e, ra, mc, N, nb, g);
  }
  v4i aq[PD][MS];
#pragma unroll
  for (int i = 0; i < PD; i++) {
    const int kn = min(i, KS - 1);
#pragma unroll
    for (int s = 0; s < MS; s++) aq[i][s] = ld16(p[s] + 64 * kn);
  }
  v4i acc[MS][4];
#pragma unroll
  for (int s = 0; s < MS; s++)
#pragma unroll
    for (int t = 0; t < 4; t++) acc[s][t] = (v4i){0, 0, 0, 0};
  __syncthreads();
  // 3. the K loop: no barrier
  const v4i* wl = wpanel + lane;
  for (int ks0 = 0; ks0 < KS; ks0 += PD) {
#pragma unroll
    for (int j = 0; j < PD; j++) {
      const int ks = ks0 + j;
      if (ks < KS) {
#pragma unroll
        for (int t = 0; t < 4; t++) {
          const v4i wv = wl[(ks * 4 + t) * 64];
#pragma unroll
          for (int s = 0; s < MS; s++) acc[s][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wv, aq[j][s], acc[s][t], 0, 0, 0);
        }
      }
      const int kn = min(ks + PD, KS - 1);
#pragma unroll
      for (int s = 0; s < MS; s++) aq[j][s] = ld16(p[s] + 64 * kn);
    }
  }
#pragma unroll
  for (int s = 0; s < MS; s++) {
    const long m = m0 + 16 * s + r;
    if (m < M) store_tile_e(acc[s], er[s], e, ra, out, m, N, nb, g);
  }
}

__device__ __forceinline__ unsigned max4_s8(unsigned a, unsigned b) {
  unsigned r = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    int x = (int)(int8_t)(a >> (8 * j)), y = (int)(int8_t)(b >> (8 * j));
    r |= (unsigned)(max(x, y) & 255) << (8 * j);
  }
  return r;
}

// ------------------------------------------------------------------------------------------
// Several independent pointwise convs in ONE launch: the 1x1 "lateral" convs the first BiFPN cell applies to the backbone outputs
// (P3 x 1, P4 x 2, P5 x 2) and the P6 conv read nothing but backbone tensors, so they need not wait for each other or for the
// nodes before them - as six launches of 5-11 us they were head and tail of a 3 us kernel each (and six of the 65 launches of
// a batch-1 forward).  A workgroup = 64 pixels x one 64-channel block of one problem, K streamed as in variant B.
// The P6 problem can carry the two 3x3/2 max pools that follow it (P6 = pool(conv(P5)), P7 = pool(P6)): then one workgroup owns one
// image, keeps the conv output in LDS and writes all three tensors.
// ------------------------------------------------------------------------------------------
constexpr int PWM_MAX = 8;
struct PwProb {
  const int8_t* x;
  const v4i* wp;
  const int* bias;
  const float* mult;
  int8_t* out;
  Rq rq;
  int M, K, KS, N, NB;     // M = pixels of the whole batch (chain: of one image)
  // chain (pool1 != nullptr): conv output H x W per image, pooled to H1 x W1 (pad pt1 / pl1) and again to H2 x W2
  int8_t* pool1;
  int8_t* pool2;
  int H, W, H1, W1, pt1, pl1, H2, W2, pt2, pl2;
};
struct PwMulti {
  int n;
  int start[PWM_MAX + 1];   // first workgroup of every problem
  PwProb p[PWM_MAX];
};

// 3x3 stride-2 max pool of an int8 [H][W][N] map held in LDS (N % 4 == 0), out-of-map taps skipped (SAME): -> global + optional LDS copy
__device__ __forceinline__ void pwm_pool(const unsigned char* src, int H, int W, int N, int OH, int OW, int pt, int pl, int8_t* gout,
                                         unsigned char* lout, int tid) {
  const int N4 = N >> 2, total = OH * OW * N4;
  const float rcp_n4 = frcp(N4), rcp_ow = frcp(OW);
  for (int i = tid; i < total; i += 256) {
    const int pix = fdiv_small(i, rcp_n4), c4 = i - pix * N4;
    const int oy = fdiv_small(pix, rcp_ow), ox = pix - oy * OW;
    unsigned best = 0x80808080u;
#pragma unroll
    for (int ky = 0; ky < 3; ky++) {
      const int iy = oy * 2 + ky - pt;
#pragma unroll
      for (int kx = 0; kx < 3; kx++) {
        const int ix = ox * 2 + kx - pl;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) best = max4_s8(best, *(const unsigned*)(src + (iy * W + ix) * N + 4 * c4));
      }
    }
    *(unsigned*)(gout + (long)pix * N + 4 * c4) = best;
    if (lout) *(unsigned*)(lout + pix * N + 4 * c4) = best;
  }
}

__global__ __launch_bounds__(256) void pw_multi_kernel(PwMulti pm) {
  extern __shared__ __attribute__((aligned(16))) unsigned char pwm_smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 15, g = lane >> 4;
  int pi = 0;
#pragma unroll
  for (int i = 1; i < PWM_MAX; i++) pi += (i < pm.n && (int)blockIdx.x >= pm.start[i]) ? 1 : 0;
  const PwProb& q = pm.p[pi];
  const int local = blockIdx.x - pm.start[pi];
  const Epi e{q.bias, q.mult, 0, 0, 0, q.rq};
  const ResArgs ra{nullptr, AddQ{0, 0, 0, 0, 0, 0, 0}};
  if (q.pool1 == nullptr) {
    const int nb = local % q.NB;
    const long m0 = ((long)(local / q.NB) * 4 + wave) * 16;
    if (m0 >= q.M) return;
    const long m = min(m0 + r, (long)q.M - 1);
    const int8_t* px = q.x + m * q.K + 16 * g;
    EpiRegs er;
    load_epi(er, e, ra, m, q.N, nb, g);
    v4i acc[4];
#pragma unroll
    for (int t = 0; t < 4; t++) acc[t] = (v4i){0, 0, 0, 0};
    const v4i* w = q.wp + (long)nb * q.KS * 4 * 64 + lane;
#pragma unroll 4
    for (int ks = 0; ks < q.KS; ks++) {
      const v4i av = ld16(px + 64 * ks);
#pragma unroll
      for (int t = 0; t < 4; t++) acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w[(ks * 4 + t) * 64], av, acc[t], 0, 0, 0);
    }
    if (m0 + r < q.M) store_tile_e(acc, er, e, ra, q.out, m0 + r, q.N, nb, g);
    return;
  }
  // chain: image `local`; conv output kept in LDS for the two pools
  const int HW = q.H * q.W, N = q.N;
  unsigned char* L0 = pwm_smem;                               // [HW][N]
  unsigned char* L1 = L0 + ((HW * N + 15) & ~15);             // [H1 * W1][N]
  const int8_t* xb = q.x + (long)local * HW * q.K;
  int8_t* ob = q.out + (long)local * HW * N;
  const int npg = (HW + 15) >> 4;
  for (int u = wave; u < npg * q.NB; u += 4) {
    const int pg = u / q.NB, nb = u - pg * q.NB;
    const int m = min(16 * pg + r, HW - 1);
    const int8_t* px = xb + (long)m * q.K + 16 * g;
    EpiRegs er;
    load_epi(er, e, ra, m, N, nb, g);
    v4i acc[4];
#pragma unroll
    for (int t = 0; t < 4; t++) acc[t] = (v4i){0, 0, 0, 0};
    const v4i* w = q.wp + (long)nb * q.KS * 4 * 64 + lane;
#pragma unroll 4
    for (int ks = 0; ks < q.KS; ks++) {
      const v4i av = ld16(px + 64 * ks);
#pragma unroll
      for (int t = 0; t < 4; t++) acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w[(ks * 4 + t) * 64], av, acc[t], 0, 0, 0);
    }
    const int c0 = nb * 64 + 16 * g;
    if (16 * pg + r < HW && c0 < N) {     // N % 16 == 0 (checked by the planner): a lane's 16 channels are all real
      const uint4 d = make_uint4(rq_pack_i(acc[0], er.b[0], er.mu[0], e.rq), rq_pack_i(acc[1], er.b[1], er.mu[1], e.rq),
                                 rq_pack_i(acc[2], er.b[2], er.mu[2], e.rq), rq_pack_i(acc[3], er.b[3], er.mu[3], e.rq));
      *(uint4*)(ob + (long)m * N + c0) = d;
      *(uint4*)(L0 + m * N + c0) = d;
    }
  }
  __syncthreads();
  pwm_pool(L0, q.H, q.W, N, q.H1, q.W1, q.pt1, q.pl1, q.pool1 + (long)local * q.H1 * q.W1 * N, L1, tid);
  __syncthreads();
  pwm_pool(L1, q.H1, q.W1, N, q.H2, q.W2, q.pt2, q.pl2, q.pool2 + (long)local * q.H2 * q.W2 * N, nullptr, tid);
}

// ------------------------------------------------------------------------------------------
// stem: 3x3 stride-2 conv on the uint8 frame as one 16x16x32 MFMA K-step.  The 27 taps are
// laid out per lane group g: g<3 -> the first 8 bytes (px0 RGB, px1 RGB, px2 RG) of kernel row g,
// g==3 -> the B byte of px2 of rows 0..2 (+5 zero weights).  QUANTIZE u8 -> s8 is the XOR 0x80.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stem_kernel(const uint8_t* __restrict__ frames, const long* __restrict__ wp, Epi e,
                                                   int8_t* __restrict__ out, long M, int H, int W, int OH, int OW,
                                                   int N, int pad_t, int pad_l, int zx) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = lane & 15, g = lane >> 4;
  const long m0 = ((long)blockIdx.x * 4 + wave) * 16;
  if (m0 >= M) return;
  long m = min(m0 + r, M - 1);
  int ox = (int)(m % OW);
  long t = m / OW;
  int oy = (int)(t % OH);
  long b = t / OH;
  const uint8_t* f = frames + b * (long)H * W * 3;
  const int ix0 = 2 * ox - pad_l, iy0 = 2 * oy - pad_t;
  const unsigned padb = (unsigned)(zx & 255);
  unsigned char by[8];
  if (g < 3) {
    int iy = iy0 + g;
    bool rowok = iy >= 0 && iy < H;
    if (rowok && ix0 >= 0 && ix0 + 2 < W) {
      unsigned long long v;
      __builtin_memcpy(&v, f + ((long)iy * W + ix0) * 3, 8);
      v ^= 0x8080808080808080ull;
      __builtin_memcpy(by, &v, 8);
    } else {
#pragma unroll
      for (int j = 0; j < 8; j++) {
        int ix = ix0 + j / 3;
        bool ok = rowok && ix >= 0 && ix < W;
        by[j] = ok ? (unsigned char)(f[((long)iy * W + ix) * 3 + j % 3] ^ 0x80) : (unsigned char)padb;
      }
    }
  } else {
    int ix = ix0 + 2;
#pragma unroll
    for (int j = 0; j < 3; j++) {
      int iy = iy0 + j;
      bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
      by[j] = ok ? (unsigned char)(f[((long)iy * W + ix) * 3 + 2] ^ 0x80) : (unsigned char)padb;
    }
#pragma unroll
    for (int j = 3; j < 8; j++) by[j] = 0;
  }
  long av;
  __builtin_memcpy(&av, by, 8);
  v4i acc[4];
#pragma unroll
  for (int t4 = 0; t4 < 4; t4++) {
    acc[t4] = (v4i){0, 0, 0, 0};
    acc[t4] = __builtin_amdgcn_mfma_i32_16x16x32_i8(wp[t4 * 64 + lane], av, acc[t4], 0, 0, 0);
  }
  if (m0 + r < M) store_tile(acc, e, out, m, N, 0, g);
}

// ------------------------------------------------------------------------------------------
// depthwise conv: lane = 4 channels x R=4 consecutive output columns.
// acc = sum u*w with u = x_q + 128 (cvt_f32_ubyte), exact in fp32; folded bias restores (x_q - z_x).
// ------------------------------------------------------------------------------------------
template <int KK, int S>
__global__ __launch_bounds__(256) void dw_kernel(const int8_t* __restrict__ x, const float* __restrict__ wf, Epi e,
                                                 int8_t* __restrict__ out, long total, int H, int W, int C, int OH,
                                                 int OW, int pad_t, int pad_l, unsigned pad4) {
  constexpr int R = 4;
  constexpr int IW = S * (R - 1) + KK;
  long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int C4 = C >> 2;
  const int XR = (OW + R - 1) / R;
  int c4 = (int)(idx % C4);
  long t = idx / C4;
  int xr = (int)(t % XR);
  t /= XR;
  int oy = (int)(t % OH);
  long b = t / OH;
  const int ox0 = xr * R;
  float acc[R][4];
#pragma unroll
  for (int o = 0; o < R; o++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[o][j] = 0.0f;
  const int8_t* xb = x + b * (long)H * W * C + 4 * c4;
#pragma unroll
  for (int ky = 0; ky < KK; ky++) {
    int iy = oy * S + ky - pad_t;
    bool rowok = iy >= 0 && iy < H;
    float4 wr[KK];
#pragma unroll
    for (int kx = 0; kx < KK; kx++) wr[kx] = *(const float4*)(wf + (long)(ky * KK + kx) * C + 4 * c4);
#pragma unroll
    for (int j = 0; j < IW; j++) {
      int ix = ox0 * S + j - pad_l;
      bool ok = rowok && ix >= 0 && ix < W;
      unsigned u = pad4;
      if (ok) u = *(const unsigned*)(xb + ((long)iy * W + ix) * C) ^ 0x80808080u;
      float f0 = (float)(u & 255u), f1 = (float)((u >> 8) & 255u), f2 = (float)((u >> 16) & 255u), f3 = (float)(u >> 24);
#pragma unroll
      for (int kx = 0; kx < KK; kx++) {
        if ((j - kx) >= 0 && (j - kx) % S == 0 && (j - kx) / S < R) {
          const int o = (j - kx) / S;
          acc[o][0] = __builtin_fmaf(f0, wr[kx].x, acc[o][0]);
          acc[o][1] = __builtin_fmaf(f1, wr[kx].y, acc[o][1]);
          acc[o][2] = __builtin_fmaf(f2, wr[kx].z, acc[o][2]);
          acc[o][3] = __builtin_fmaf(f3, wr[kx].w, acc[o][3]);
        }
      }
    }
  }
  int4 bq = *(const int4*)(e.bias + 4 * c4);
  float4 mu = *(const float4*)(e.mult + 4 * c4);
  int8_t* ob = out + ((b * OH + oy) * (long)OW) * C + 4 * c4;
#pragma unroll
  for (int o = 0; o < R; o++) {
    int ox = ox0 + o;
    if (ox < OW) {
      v4i ai = {(int)acc[o][0], (int)acc[o][1], (int)acc[o][2], (int)acc[o][3]};
      unsigned d = rq_pack_i(ai, bq, mu, e.rq);
      *(unsigned*)(ob + (long)ox * C) = d;
    }
  }
}

// ------------------------------------------------------------------------------------------
// depthwise conv, column walker: a lane owns 4 channels x 4 output columns and walks DOWN a segment of
// output rows.  The k*k*4 weights stay in registers for the whole walk; every input row is loaded and
// converted once and scattered into the (at most ceil(k/s)) output rows still in flight, which live in a
// statically indexed accumulator ring (the row loop is unrolled by the ring period).
// ------------------------------------------------------------------------------------------
constexpr int cmod(int a, int n) { return ((a % n) + n) % n; }
constexpr int cfloordiv(int a, int n) { return (a - cmod(a, n)) / n; }

template <int KK, int S>
__global__ __launch_bounds__(256) void dw_col_kernel(const int8_t* __restrict__ x, const float* __restrict__ wf, Epi e,
                                                     int8_t* __restrict__ out, long total, int H, int W, int C, int OH, int OW,
                                                     int pad_t, int pad_l, unsigned pad4, int rows, int nseg) {
  constexpr int IW = 3 * S + KK;
  constexpr int NS = (KK + S - 1) / S;
  constexpr int P = NS * S;
  long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int C4 = C >> 2, XR = (OW + 3) >> 2;
  const int c4 = (int)(idx % C4);
  long t = idx / C4;
  const int xr = (int)(t % XR);
  t /= XR;
  const int seg = (int)(t % nseg);
  const long b = t / nseg;
  const int oy_b = seg * rows;
  const int nrows = min(rows, OH - oy_b);
  if (nrows <= 0) return;
  const int ox0 = xr * 4;
  const int iy_b = oy_b * S - pad_t, ix_b = ox0 * S - pad_l;
  const int n_in = (nrows - 1) * S + KK;
  float4 w[KK][KK];
#pragma unroll
  for (int ky = 0; ky < KK; ky++)
#pragma unroll
    for (int kx = 0; kx < KK; kx++) w[ky][kx] = *(const float4*)(wf + (long)(ky * KK + kx) * C + 4 * c4);
  const int4 bq = *(const int4*)(e.bias + 4 * c4);
  const float4 mu = *(const float4*)(e.mult + 4 * c4);
  float acc[NS][4][4];
#pragma unroll
  for (int sl = 0; sl < NS; sl++)
#pragma unroll
    for (int o = 0; o < 4; o++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[sl][o][j] = 0.0f;
  unsigned colmask = 0;
#pragma unroll
  for (int j = 0; j < IW; j++)
    if (ix_b + j >= 0 && ix_b + j < W) colmask |= 1u << j;
  const int8_t* xb = x + b * (long)H * W * C + 4 * c4;
  int8_t* ob = out + b * (long)OH * OW * C + 4 * c4;
  for (int i0 = 0; i0 < n_in; i0 += P) {
#pragma unroll
    for (int r = 0; r < P; r++) {
      const int i = i0 + r;
      if (i >= n_in) break;
      const int iy = iy_b + i;
      const bool rowok = iy >= 0 && iy < H;
      const int8_t* rp = xb + ((long)iy * W + ix_b) * C;
      bool kyok[KK];
#pragma unroll
      for (int ky = 0; ky < KK; ky++) kyok[ky] = (i - ky) >= 0 && (i - ky) / S < nrows;
#pragma unroll
      for (int j = 0; j < IW; j++) {
        unsigned u = pad4;
        if (rowok && ((colmask >> j) & 1u)) u = *(const unsigned*)(rp + (long)j * C) ^ 0x80808080u;
        const float f0 = (float)(u & 255u), f1 = (float)((u >> 8) & 255u), f2 = (float)((u >> 16) & 255u), f3 = (float)(u >> 24);
#pragma unroll
        for (int ky = 0; ky < KK; ky++) {
          if (cmod(r - ky, S) == 0) {
            constexpr int dummy = 0;
            (void)dummy;
            const int sl = cmod(cfloordiv(r - ky, S), NS);
            if (kyok[ky]) {
#pragma unroll
              for (int kx = 0; kx < KK; kx++) {
                if ((j - kx) >= 0 && (j - kx) % S == 0 && (j - kx) / S < 4) {
                  const int o = (j - kx) / S;
                  acc[sl][o][0] = __builtin_fmaf(f0, w[ky][kx].x, acc[sl][o][0]);
                  acc[sl][o][1] = __builtin_fmaf(f1, w[ky][kx].y, acc[sl][o][1]);
                  acc[sl][o][2] = __builtin_fmaf(f2, w[ky][kx].z, acc[sl][o][2]);
                  acc[sl][o][3] = __builtin_fmaf(f3, w[ky][kx].w, acc[sl][o][3]);
                }
              }
            }
          }
        }
      }
      // the output row whose last input row this was
      if (cmod(r - (KK - 1), S) == 0) {
        const int sl = cmod(cfloordiv(r - (KK - 1), S), NS);
        const int od = (i - (KK - 1)) / S;
        if (i - (KK - 1) >= 0 && od < nrows) {
          int8_t* orow = ob + ((long)(oy_b + od) * OW) * C;
#pragma unroll
          for (int o = 0; o < 4; o++) {
            const v4i ai = {(int)acc[sl][o][0], (int)acc[sl][o][1], (int)acc[sl][o][2], (int)acc[sl][o][3]};
            if (ox0 + o < OW) *(unsigned*)(orow + (long)(ox0 + o) * C) = rq_pack_i(ai, bq, mu, e.rq);
          }
        }
#pragma unroll
        for (int o = 0; o < 4; o++)
#pragma unroll
          for (int j = 0; j < 4; j++) acc[sl][o][j] = 0.0f;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// elementwise binary int8 ADD: 16 bytes per lane
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void add_kernel(const int8_t* __restrict__ xa, const int8_t* __restrict__ xb, AddQ q,
                                                  int8_t* __restrict__ out, long n4) {
  long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  const bool al16 = ((((unsigned long)xa) | ((unsigned long)xb) | ((unsigned long)out)) & 15ul) == 0;   // (sub-batch offsets may break it)
  if (al16 && i + 4 <= n4) {
    const uint4 va = *(const uint4*)(xa + 4 * i), vb = *(const uint4*)(xb + 4 * i);
    *(uint4*)(out + 4 * i) = make_uint4(addq4(va.x, vb.x, q), addq4(va.y, vb.y, q), addq4(va.z, vb.z, q), addq4(va.w, vb.w, q));
  } else {
    for (long e = i + 4; i < n4 && i < e; i++) ((unsigned*)out)[i] = addq4(((const unsigned*)xa)[i], ((const unsigned*)xb)[i], q);
  }
}

__global__ __launch_bounds__(256) void maxpool_kernel(const int8_t* __restrict__ x, int8_t* __restrict__ out, long total,
                                                      int H, int W, int C, int OH, int OW, int pad_t, int pad_l) {
  long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int C4 = C >> 2;
  int c4 = (int)(idx % C4);
  long t = idx / C4;
  int ox = (int)(t % OW);
  t /= OW;
  int oy = (int)(t % OH);
  long b = t / OH;
  unsigned best = 0x80808080u;  // -128 x4
  for (int ky = 0; ky < 3; ky++) {
    int iy = oy * 2 + ky - pad_t;
    if (iy < 0 || iy >= H) continue;
    for (int kx = 0; kx < 3; kx++) {
      int ix = ox * 2 + kx - pad_l;
      if (ix < 0 || ix >= W) continue;
      unsigned v = *(const unsigned*)(x + ((b * H + iy) * (long)W + ix) * C + 4 * c4);
      best = max4_s8(best, v);
    }
  }
  *(unsigned*)(out + ((b * OH + oy) * (long)OW + ox) * C + 4 * c4) = best;
}

__global__ __launch_bounds__(256) void resize_kernel(const int8_t* __restrict__ x, int8_t* __restrict__ out, long total,
                                                     int H, int W, int C, int OH, int OW) {
  long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int C4 = C >> 2;
  int c4 = (int)(idx % C4);
  long t = idx / C4;
  int ox = (int)(t % OW);
  t /= OW;
  int oy = (int)(t % OH);
  long b = t / OH;
  int iy = (oy * H) / OH, ix = (ox * W) / OW;
  *(unsigned*)(out + ((b * OH + oy) * (long)OW + ox) * C + 4 * c4) =
      *(const unsigned*)(x + ((b * H + iy) * (long)W + ix) * C + 4 * c4);
}

// ------------------------------------------------------------------------------------------
// TFLite_Detection_PostProcess (fast path, max_classes_per_detection = 1: every anchor scores with the best of its C class columns, then
// single-class NMS): one workgroup per frame.
// ------------------------------------------------------------------------------------------
struct PostArgs {
  const int8_t* cls[5];
  const int8_t* box[5];
  int base[6];         // first anchor index of each level, base[5] = A
  const float* anchors;  // [A][4] ycenter, xcenter, h, w
  // device tables built at model load from the container's (vbt_amd/quant.py): score f32[256] indexed by RANK byte + 128 |
  // dq f64[256] | ex f64[256] indexed by box byte + 128 | rank int8[256] indexed by class byte + 128.
  // rank byte: class bytes with EQUAL scores (plateaus of the LOGISTIC table) share one, higher score = higher rank byte,
  // so that sorting by (rank desc, anchor asc) is the reference's stable sort on the float scores.
  const unsigned char* tables;
  int A, max_det, qmin;  // qmin: lowest RANK byte whose score >= nms_score_threshold (128 = none)
  float iou_thr;
  int C;                 // class columns per anchor: the class tensors are [h][w][anchors_per_location * C], anchor a's bytes are contiguous
};
constexpr int POST_CAP = 2048;
// Phase timers (developer builds only: VBT_EXTRA_CXXFLAGS=-DVBT_POST_PROF): s_memtime deltas of workgroup 0, thread 0, summed per phase;
// read with vbt_post_prof_read (tools/post_prof.py).  The product build holds no stamp.
#ifdef VBT_POST_PROF
__device__ unsigned long long g_post_prof[16];
#define POST_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); atomicAdd(&g_post_prof[i], t_ - t_last); t_last = t_; } } while (0)
#else
#define POST_STAMP(i) do { } while (0)
#endif

// LDS carve-up (dynamic, every offset a multiple of 16): class bytes of the frame in anchor order | 32 KB: the 32 interleaved
// histograms of pass 1, afterwards candidate keys, their sorted copy and the decoded boxes / scores | small tables.
constexpr int POST_THREADS = 1024;  // one workgroup per frame: sixteen waves shorten every parallel phase of a latency-bound kernel
constexpr int POST_FAST = 1024;     // rounds of at most this many candidates: counting sort + all boxes decoded at once, one per thread
constexpr int POST_LDS_FIXED = 32768 + 1024 + 1024 + 2048 + 2048 + 256 + 32 * 16 + 128 + 128;   // (+ scalars, + areas of the selected boxes)
static_assert(POST_CAP * 4 + POST_FAST * 4 + POST_FAST * 16 + POST_FAST * 4 <= 32768, "keys | sorted keys | boxes | scores share the histogram's 32 KB");
__host__ __device__ inline int post_lds_bytes(int A) { return ((A + 15) & ~15) + POST_LDS_FIXED; }

// inclusive prefix sum over the 64 lanes of a wavefront on the DPP path (row shifts inside rows of 16 lanes, then the row
// broadcasts of GFX9): six dependent VALU instructions instead of six LDS-crossbar shuffles
__device__ __forceinline__ int wave_incl_scan_i(int x) {
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);   // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);   // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);   // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);   // row_shr:8
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
  return x;
}
// first lane whose flag is set (64: none), and a value of that lane (wave-uniform results, no shuffles)
__device__ __forceinline__ int wave_first_lane(bool flag) {
  const unsigned long long m = __ballot(flag);
  return m ? __ffsll((long long)m) - 1 : 64;
}

// One workgroup per frame.  Latency is what matters here (one frame = one workgroup, nothing to overlap it with inside a
// forward), so every phase is laid out for few dependent round trips and short dependent chains:
//   1. the frame's class bytes (A of them, five level tensors) are copied into LDS with ALL loads of a thread in flight at once;
//   2. histogram of the class bytes in 32 interleaved copies (lane l adds to copy l & 31: every lane of a half-wave owns a bank,
//      so the bulk of the anchors, which shares a handful of bins, does not serialise), folded into a histogram of RANK bytes
//      (equal scores share a rank);
//   3. wavefront 0 picks the next range of ranks, from the top: a prefix scan over the 256 bins (bins are added while fewer than
//      256 candidates are held and the next one still fits POST_CAP; one oversized bin is walked in slices of anchor indices);
//   4. candidates of that range: counted per thread, block-wide exclusive scan, written as keys (127 - rank) << 16 | anchor;
//      ascending key = (score desc, anchor asc), the reference's stable sort; <= POST_FAST keys are sorted by counting (no
//      barriers inside), more by a bitonic network;
//   5. <= POST_FAST candidates are decoded by all threads at once (one round trip for box bytes + anchors) into LDS;
//   6. greedy suppression by wavefront 0; next round only if fewer than max_det boxes survived.
__global__ __launch_bounds__(POST_THREADS) void postprocess_kernel(PostArgs p, float* __restrict__ boxes, float* __restrict__ scores,
                                                                   float* __restrict__ classes, int* __restrict__ counts) {
  extern __shared__ __attribute__((aligned(16))) unsigned char post_smem[];
  constexpr int NT = POST_THREADS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long b = blockIdx.x;
#ifdef VBT_POST_PROF
  unsigned long long t_last = __builtin_amdgcn_s_memtime();
#endif
  const int A = p.A;
  // fixed-size regions first (compile-time LDS addresses), the class bytes - the one region whose size depends on the model - last
  int* hist32 = (int*)post_smem;                                   // [256][32] during pass 1; afterwards:
  unsigned* keys = (unsigned*)hist32;                              // [POST_CAP]
  unsigned* skeys = keys + POST_CAP;                               // [POST_FAST]
  float4* lbox = (float4*)(skeys + POST_FAST);                     // [POST_FAST]
  float* lsc = (float*)(lbox + POST_FAST);                         // [POST_FAST]
  int* rhist = (int*)(post_smem + 32768);                          // [256] candidates per rank byte
  float* s_score = (float*)(rhist + 256);                          // dequantised LOGISTIC output per rank byte
  double* s_dq = (double*)(s_score + 256);                         // (double)box / y_scale per box byte
  double* s_ex = s_dq + 256;                                       // exp((double)box / h_scale) per box byte
  signed char* s_rank = (signed char*)(s_ex + 256);                // rank byte per class byte
  float4* selbox = (float4*)(s_rank + 256);                        // [32] boxes selected so far
  int* sv = (int*)(selbox + 32);                                   // scalars + per-wave totals
  unsigned char* cb = post_smem + POST_LDS_FIXED;                  // [A16] class bytes in anchor order
  int& s_n = sv[0]; int& s_qlo = sv[1]; int& s_qhi = sv[2]; int& s_i0 = sv[3]; int& s_i1 = sv[4]; int& s_nsel = sv[5]; int& s_done = sv[6];
  int& s_blo = sv[7]; int& s_bhi = sv[8];
  int* s_wtot = sv + 12;                                           // [16]
  // ---- 1. class bytes -> registers (+ LDS).  The frame's A class bytes, five level tensors, are taken as ONE array in anchor order;
  // thread t owns the dwords i = t + 1024 k (anchors 4 i .. 4 i + 3).  A dword that lies inside one level at a dword-aligned
  // address is one load (all PS loads of a thread are in flight together); the few that straddle two levels or sit in a
  // level whose frame offset is odd (9 H W bytes per frame) are patched byte by byte afterwards.
  constexpr int PS = 5;                        // dword slots per thread (x 1024 threads x 4 anchors: Lite0); beyond that: loops over LDS
  const int n4 = (A + 3) >> 2;                 // (bytes past A inside the last dword are excluded by their anchor index)
  const int b1 = p.base[1], b2 = p.base[2], b3 = p.base[3], b4 = p.base[4];
  auto level_of = [&](int a) -> int { return (a >= b1) + (a >= b2) + (a >= b3) + (a >= b4); };
  // Anchor a's class byte sits at pc0 + a + (sum of the steps of the level boundaries at or below a): every level pointer is
  // biased by its first anchor, and the choice of the level is four conditional 64-bit adds (a select chain over five pointers
  // is turned into a table in scratch memory by the compiler: a dependent load per address).
  const unsigned char *pc0, *pb0;
  long ce1, ce2, ce3, ce4, be1, be2, be3, be4;
  {
    auto cbias = [&](int l) { return (long)p.cls[l] + (b * (long)(p.base[l + 1] - p.base[l]) - p.base[l]) * (long)p.C; };
    auto bbias = [&](int l) { return (long)p.box[l] + (b * (long)(p.base[l + 1] - p.base[l]) - p.base[l]) * 4; };
    pc0 = (const unsigned char*)cbias(0); pb0 = (const unsigned char*)bbias(0);
    ce1 = cbias(1) - cbias(0); ce2 = cbias(2) - cbias(1); ce3 = cbias(3) - cbias(2); ce4 = cbias(4) - cbias(3);
    be1 = bbias(1) - bbias(0); be2 = bbias(2) - bbias(1); be3 = bbias(3) - bbias(2); be4 = bbias(4) - bbias(3);
  }
  auto cls_ptr = [&](int a) -> const unsigned char* {      // address of anchor a's class byte(s)
    long o = (long)a * p.C;
    o += a >= b1 ? ce1 : 0l; o += a >= b2 ? ce2 : 0l; o += a >= b3 ? ce3 : 0l; o += a >= b4 ? ce4 : 0l;
    return pc0 + o;
  };
  auto box_ptr = [&](int a) -> const unsigned char* {      // address of anchor a's four box bytes
    long o = 4l * a;
    o += a >= b1 ? be1 : 0l; o += a >= b2 ? be2 : 0l; o += a >= b3 ? be3 : 0l; o += a >= b4 ? be4 : 0l;
    return pb0 + o;
  };
  auto whole = [&](int i, const unsigned char* src) -> bool {   // dword i: inside one level, dword-aligned, entirely below A
    const int a = 4 * i;
    return level_of(a) == level_of(a + 3) && a + 3 < A && (((unsigned long)src) & 3ul) == 0;
  };
  auto patch = [&](int i) -> unsigned {                      // the same dword assembled from single bytes
    unsigned w = 0;
    for (int e = 0; e < 4; e++)
      if (4 * i + e < A) w |= (unsigned)*cls_ptr(4 * i + e) << (8 * e);
    return w;
  };
  if (p.C == 1) {
    unsigned w[PS];
#pragma unroll
    for (int k = 0; k < PS; k++)      // (a dword that will be patched reads the aligned dword around its first byte: inside the arena)
      w[k] = *(const unsigned*)((unsigned long)cls_ptr(4 * min(tid + NT * k, n4 - 1)) & ~3ul);
#pragma unroll
    for (int k = 0; k < PS; k++)
      if (tid + NT * k < n4) ((unsigned*)cb)[tid + NT * k] = w[k];
#pragma unroll 1
    for (int i = tid + NT * PS; i < n4; i += NT) ((unsigned*)cb)[i] = *(const unsigned*)((unsigned long)cls_ptr(4 * i) & ~3ul);
#pragma unroll 1
    for (int i = tid; i < n4; i += NT)
      if (!whole(i, cls_ptr(4 * i))) ((unsigned*)cb)[i] = patch(i);
  } else {
    // C class columns per anchor: the anchor scores with its best column (NonMaxSuppressionMultiClassFastHelper, one class per
    // detection).  The LOGISTIC table is monotone, so the best score is the score of the largest byte; the byte array in LDS then is
    // exactly the one-column case.
    auto best4 = [&](int i) -> unsigned {       // dword i of the array, column by column (any C; the few dwords the fast path leaves)
      unsigned w = 0;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int a = min(4 * i + e, A - 1);
        const signed char* src = (const signed char*)cls_ptr(a);
        int best = src[0];
        for (int c = 1; c < p.C; c++) best = max(best, (int)src[c]);
        w |= ((unsigned)best & 255u) << (8 * e);
      }
      return w;
    };
    if (p.C == 2) {
      // two columns (the reference's models): the 8 class bytes of 4 anchors are one 8-byte load where they lie inside one level at an
      // 8-byte-aligned address (all PS loads of a thread in flight together); signed byte maxima of the pairs as packed-u16 maxima on the
      // u8 image of the bytes.  Dwords that straddle two levels or sit at an odd offset (levels of 450 / 162 bytes per frame) are patched.
      constexpr unsigned HB = 0x80808080u;
      auto pair_max = [&](uint2 v) -> unsigned {
        const unsigned w0 = v.x ^ HB, w1 = v.y ^ HB;
        const unsigned m0 = pk_max_u16(w0 & 0x00FF00FFu, (w0 >> 8) & 0x00FF00FFu);   // bits 0-7: max(b0, b1), bits 16-23: max(b2, b3)
        const unsigned m1 = pk_max_u16(w1 & 0x00FF00FFu, (w1 >> 8) & 0x00FF00FFu);
        return ((m0 & 0xFFu) | ((m0 >> 8) & 0xFF00u) | ((m1 & 0xFFu) << 16) | ((m1 << 8) & 0xFF000000u)) ^ HB;
      };
      auto whole2 = [&](int i, const unsigned char* src) -> bool {
        const int a = 4 * i;
        return level_of(a) == level_of(a + 3) && a + 3 < A && (((unsigned long)src) & 7ul) == 0;
      };
      uint2 w[PS];
#pragma unroll
      for (int k = 0; k < PS; k++)
        w[k] = *(const uint2*)((unsigned long)cls_ptr(4 * min(tid + NT * k, n4 - 1)) & ~7ul);
#pragma unroll
      for (int k = 0; k < PS; k++)
        if (tid + NT * k < n4) ((unsigned*)cb)[tid + NT * k] = pair_max(w[k]);
#pragma unroll 1
      for (int i = tid + NT * PS; i < n4; i += NT) ((unsigned*)cb)[i] = pair_max(*(const uint2*)((unsigned long)cls_ptr(4 * i) & ~7ul));
#pragma unroll 1
      for (int i = tid; i < n4; i += NT)
        if (!whole2(i, cls_ptr(4 * i))) ((unsigned*)cb)[i] = best4(i);
    } else {
#pragma unroll 1
      for (int i = tid; i < n4; i += NT) ((unsigned*)cb)[i] = best4(i);
    }
  }
#pragma unroll
  for (int k = 0; k < 8; k++) hist32[tid + NT * k] = 0;
  if (tid < 256) {
    rhist[tid] = 0;
    s_score[tid] = ((const float*)p.tables)[tid];
    s_dq[tid] = ((const double*)(p.tables + 1024))[tid];
    s_ex[tid] = ((const double*)(p.tables + 3072))[tid];
    s_rank[tid] = ((const signed char*)(p.tables + 5120))[tid];
  }
  if (tid == 0) { s_nsel = 0; s_done = 0; }
  __syncthreads();
  POST_STAMP(0);
  // ---- 2. histogram of the class bytes (bin = byte + 128); a thread keeps its first PS dwords in registers for the candidate passes
  unsigned u[PS];
#pragma unroll
  for (int k = 0; k < PS; k++) u[k] = ((const unsigned*)cb)[min(tid + NT * k, n4 - 1)] ^ 0x80808080u;
  {
    const int copy = tid & 31;
#pragma unroll
    for (int k = 0; k < PS; k++) {
      const int i = tid + NT * k;
#pragma unroll
      for (int e = 0; e < 4; e++)
        if (4 * i + e < A) atomicAdd(&hist32[(((u[k] >> (8 * e)) & 255u) << 5) + copy], 1);
    }
    for (int i = tid + NT * PS; i < n4; i += NT) {
      const unsigned w = ((const unsigned*)cb)[i] ^ 0x80808080u;
#pragma unroll
      for (int e = 0; e < 4; e++)
        if (4 * i + e < A) atomicAdd(&hist32[(((w >> (8 * e)) & 255u) << 5) + copy], 1);
    }
  }
  __syncthreads();
  {  // bin tid >> 2, copies 8 (tid & 3) .. + 7 (rotated by the bin: consecutive bins start in different banks)
    const int bin = tid >> 2, part = tid & 3;
    int c = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) c += hist32[(bin << 5) + 8 * part + ((k + bin) & 7)];
    if (c) atomicAdd(&rhist[(int)s_rank[bin] + 128], c);
  }
  __syncthreads();      // (hist32 is dead from here on: its bytes hold keys, sorted keys, boxes and scores)
  POST_STAMP(1);
  int qcur = 127;   // highest rank byte not yet consumed (uniform across the block)
  int seg0 = 0;     // for an oversized bin: next anchor index to scan
  while (true) {
    // ---- 3. next range of ranks: lane L of wavefront 0 owns the ranks 127 - 4L .. 124 - 4L (descending positions t = 4L + j)
    if (wave == 0) {
      int v[4], before[4];
      int sum = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int r = 127 - (4 * lane + j);
        v[j] = (r <= qcur && r >= p.qmin) ? rhist[r + 128] : 0;
        before[j] = sum;
        sum += v[j];
      }
      const int excl = wave_incl_scan_i(sum) - sum;
      int ffail = 4, ftop = 4;              // first failing / first non-empty position of this lane (4: none)
#pragma unroll
      for (int j = 3; j >= 0; j--) {
        const int r = 127 - (4 * lane + j);
        const int bf = excl + before[j];
        const bool ok = r > qcur || (r >= p.qmin && bf < 256 && bf + v[j] <= POST_CAP);
        if (!ok) ffail = j;
        if (v[j] > 0) ftop = j;
      }
      const int lf = wave_first_lane(ffail < 4), lt = wave_first_lane(ftop < 4);
      const int tfail = lf < 64 ? 4 * lf + __builtin_amdgcn_readlane(ffail, lf) : 256;
      const int ttop = lt < 64 ? 4 * lt + __builtin_amdgcn_readlane(ftop, lt) : 256;
      if (lane == 0) {
        if (ttop >= 256) {
          s_done = 1;                                   // nothing left at or above the score threshold
        } else if (tfail <= ttop) {                     // the first non-empty bin alone exceeds POST_CAP: a slice of anchor indices
          s_qhi = 127 - ttop; s_qlo = 127 - ttop; s_i0 = seg0; s_i1 = min(seg0 + POST_CAP, A);
        } else {
          s_qhi = 127 - ttop; s_qlo = 128 - tfail; s_i0 = 0; s_i1 = A;
        }
        s_blo = 255; s_bhi = 0;
      }
    }
    __syncthreads();
    POST_STAMP(2);
    if (s_done) break;
    const int qhi = s_qhi, qlo = s_qlo, i0 = s_i0, i1 = s_i1;
    if (tid < 256) {  // class-byte bins whose rank lies in [qlo, qhi]: a contiguous range (the LOGISTIC table is monotone)
      const int r = (int)s_rank[tid];
      const unsigned long long m = __ballot(r >= qlo && r <= qhi);
      if (lane == 0 && m) {
        atomicMin(&s_blo, 64 * wave + __ffsll((long long)m) - 1);
        atomicMax(&s_bhi, 64 * wave + 63 - __clzll((long long)m));
      }
    }
    __syncthreads();
    POST_STAMP(11);
    // ---- 4. candidates: count, block-wide exclusive scan, write.  The range test runs on four class bytes at a time (SWAR):
    // d = (x - blo) mod 256 per byte, hit = d <= span per byte, as bit 7 of each byte of `hm`; candidates are about 1 % of the anchors,
    // so the per-candidate work (below) is rare and the per-dword work is what counts.
    {
      constexpr unsigned H = 0x80808080u;
      const unsigned LO7 = ((unsigned)s_blo * 0x01010101u) & ~H, NLO = ~((unsigned)s_blo * 0x01010101u);
      const unsigned SP = (unsigned)(s_bhi - s_blo) * 0x01010101u;
      const bool fullrange = i0 == 0 && i1 == A;
      auto hits = [&](unsigned x, int i) -> unsigned {
        const unsigned d = ((x | H) - LO7) ^ ((x ^ NLO) & H);             // bytewise x - blo
        const unsigned t = (SP | H) - (d & ~H);                           // bit 7: low seven bits of span >= those of d
        unsigned m = ((SP & ~d) | (~(SP ^ d) & t)) & H;                   // bytewise d <= span
        if (fullrange) {
          if (i == n4 - 1 && (A & 3)) m &= (1u << (8 * (A & 3))) - 1u;    // bytes past the last anchor
        } else {                                                          // a slice [i0, i1) of anchor indices (oversized bin)
          const int lo = min(max(i0 - 4 * i, 0), 4), hi = min(max(i1 - 4 * i, 0), 4);
          m &= (unsigned)(((1ull << (8 * hi)) - 1ull) & ~((1ull << (8 * lo)) - 1ull));
        }
        return m;
      };
      unsigned hm[PS];
      int cnt = 0;
#pragma unroll
      for (int k = 0; k < PS; k++) {
        hm[k] = tid + NT * k < n4 ? hits(u[k], tid + NT * k) : 0u;
        cnt += __popc(hm[k]);
      }
#pragma unroll 1
      for (int i = tid + NT * PS; i < n4; i += NT) cnt += __popc(hits(((const unsigned*)cb)[i] ^ H, i));
      const int incl = wave_incl_scan_i(cnt);
      if (lane == 63) s_wtot[wave] = incl;
      __syncthreads();
      POST_STAMP(12);
      int pos = incl - cnt;
#pragma unroll
      for (int w = 0; w < NT / 64; w++) pos += (w < wave) ? s_wtot[w] : 0;
      if (tid == NT - 1) s_n = pos + cnt;
      // keys = (127 - rank) << 16 | anchor  (ascending key = score desc, anchor asc)
      auto emit = [&](unsigned x, unsigned m, int i) {
#pragma unroll 1
        while (m) {
          const int e = (__ffs((int)m) - 1) >> 3;
          m &= m - 1u;
          keys[pos++] = ((unsigned)(127 - (int)s_rank[(x >> (8 * e)) & 255u]) << 16) | (unsigned)(4 * i + e);
        }
      };
#pragma unroll
      for (int k = 0; k < PS; k++)
        if (hm[k]) emit(u[k], hm[k], tid + NT * k);
#pragma unroll 1
      for (int i = tid + NT * PS; i < n4; i += NT) {
        const unsigned x = ((const unsigned*)cb)[i] ^ H;
        const unsigned m = hits(x, i);
        if (m) emit(x, m, i);
      }
    }
    __syncthreads();
    const int n = s_n;
    const bool fast = n <= POST_FAST;
    const unsigned* sk = fast ? skeys : keys;
    if (fast) {
      const int nq = (n + 3) >> 2;
      if (tid < 4 * nq - n) keys[n + tid] = 0xFFFFFFFFu;
    } else {
      int n2 = 2048;
      for (int i = n + tid; i < n2; i += NT) keys[i] = 0xFFFFFFFFu;
    }
    __syncthreads();
    POST_STAMP(3);
    if (fast) {
      // counting sort: position of a key = number of smaller keys (keys are unique: the anchor index is part of them); the keys to
      // compare with are dealt to `parts` threads per key, their partial counts meet in LDS (the box region, not yet in use)
      const int nq = (n + 3) >> 2;
      const int parts = n > 0 ? min(NT / n, 4) : 1;
      int* partial = (int*)lbox;
      if (tid < parts * n) {
        const int m = tid % n, part = tid / n;
        const unsigned mine = keys[m];
        int r = 0;
#pragma unroll 4
        for (int j = part * nq / parts; j < (part + 1) * nq / parts; j++) {
          const uint4 k4 = ((const uint4*)keys)[j];
          r += (k4.x < mine) + (k4.y < mine) + (k4.z < mine) + (k4.w < mine);
        }
        partial[part * POST_FAST + m] = r;
      }
      __syncthreads();
      if (tid < n) {
        int r = partial[tid];
        for (int q = 1; q < parts; q++) r += partial[q * POST_FAST + tid];
        skeys[r] = keys[tid];
      }
      __syncthreads();
    } else {
      const int n2 = 2048;
      for (int k = 2; k <= n2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
          for (int i = tid; i < n2; i += NT) {
            int ixj = i ^ j;
            if (ixj > i) {
              unsigned a = keys[i], c = keys[ixj];
              bool up = (i & k) == 0;
              if ((a > c) == up) { keys[i] = c; keys[ixj] = a; }
            }
          }
          __syncthreads();
        }
    }
    POST_STAMP(4);
#ifdef VBT_POST_PROF
    if (tid == 0 && blockIdx.x == 0) { atomicAdd(&g_post_prof[8], 1ull); atomicAdd(&g_post_prof[9], (unsigned long long)n); }
#endif
    // DecodeCenterSizeBoxes (detection_postprocess.cc): double intermediates (one mul, one add: no contraction),
    // one rounding to float per quantity, then float corner arithmetic
    auto decode = [&](unsigned key, float& sc) -> float4 {
      const int a = (int)(key & 0xFFFFu);
      const int q = 127 - (int)(key >> 16);
      const unsigned bq = *(const unsigned*)box_ptr(a);
      const float4 an = *(const float4*)(p.anchors + (long)a * 4);
      const float yc = (float)(s_dq[(int)((bq & 255u) ^ 128u)] * (double)an.z + (double)an.x);
      const float xc = (float)(s_dq[(int)(((bq >> 8) & 255u) ^ 128u)] * (double)an.w + (double)an.y);
      const float hh = (float)(0.5 * s_ex[(int)(((bq >> 16) & 255u) ^ 128u)] * (double)an.z);
      const float hw = (float)(0.5 * s_ex[(int)((bq >> 24) ^ 128u)] * (double)an.w);
      sc = s_score[q + 128];
      return make_float4(yc - hh, xc - hw, yc + hh, xc + hw);
    };
    // ---- 5. fast path: every candidate of the round decoded at once
    if (fast) {
      if (tid < n) {
        float sc;
        lbox[tid] = decode(skeys[tid], sc);
        lsc[tid] = sc;
      }
      __syncthreads();
    }
    POST_STAMP(13);
    // ---- 6. greedy suppression by wavefront 0 (IoU as in detection_postprocess.cc: 0 when either area is not positive; the
    // areas of the candidates and of the selected boxes are computed once)
    if (wave == 0) {
      float* selarea = (float*)(sv + 32);
      auto sup = [&](float4 sb, float sa, float4 cx, float ca) -> bool {
        const float iy0 = fmaxf(sb.x, cx.x), ix0 = fmaxf(sb.y, cx.y);
        const float iy1 = fminf(sb.z, cx.z), ix1 = fminf(sb.w, cx.w);
        const float inter = fmaxf(iy1 - iy0, 0.0f) * fmaxf(ix1 - ix0, 0.0f);
        return sa > 0.0f && ca > 0.0f && inter / (sa + ca - inter) > p.iou_thr;
      };
      int nsel = s_nsel;
      for (int base = 0; base < n && nsel < p.max_det; base += 64) {
        int ci = base + lane;
        bool alive = ci < n;
        float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
        float sc = 0.f;
        int anchor = 0;
        if (alive) {
          anchor = (int)(sk[ci] & 0xFFFFu);
          if (fast) { bx = lbox[ci]; sc = lsc[ci]; }
          else bx = decode(sk[ci], sc);
        }
        const float area = (bx.z - bx.x) * (bx.w - bx.y);
        if (alive)
          for (int s = 0; s < nsel; s++)
            if (sup(selbox[s], selarea[s], bx, area)) { alive = false; break; }
        while (nsel < p.max_det) {
          const unsigned long long mask = __ballot(alive);
          if (mask == 0ull) break;
          const int j = __ffsll((long long)mask) - 1;     // wave-uniform: the lane's values come over the scalar path
          float4 sb;
          sb.x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bx.x), j));
          sb.y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bx.y), j));
          sb.z = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bx.z), j));
          sb.w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bx.w), j));
          const float ss = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sc), j));
          const float sa = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(area), j));
          const int sel_anchor = __builtin_amdgcn_readlane(anchor, j);
          if (lane == 0) {
            selbox[nsel] = sb;
            selarea[nsel] = sa;
            float* bo = boxes + (b * p.max_det + nsel) * 4;
            bo[0] = sb.x; bo[1] = sb.y; bo[2] = sb.z; bo[3] = sb.w;
            scores[b * p.max_det + nsel] = ss;
            // detection class = ArgMaxVector over the float scores of the anchor's columns: the FIRST column on the best score's rank
            int cls_id = 0;
            if (p.C > 1) {
              const unsigned char* src = cls_ptr(sel_anchor);
              int best = -128;
              for (int c = 0; c < p.C; c++) {
                const int r = (int)s_rank[(int)(src[c] ^ 128u)];
                if (r > best) { best = r; cls_id = c; }
              }
            }
            classes[b * p.max_det + nsel] = (float)cls_id;
          }
          __threadfence_block();  // selbox[] is read by the other lanes of this wavefront
          nsel++;
          if (lane == j) alive = false;
          else if (alive && sup(sb, sa, bx, area)) alive = false;
        }
      }
      if (lane == 0) s_nsel = nsel;
    }
    __syncthreads();
    POST_STAMP(5);
    if (s_nsel >= p.max_det) break;
    if (qhi == qlo && rhist[qhi + 128] > POST_CAP && i1 < A) { seg0 = i1; qcur = qhi; }
    else { seg0 = 0; qcur = qlo - 1; }
    __syncthreads();
  }
  const int nsel = s_nsel;
  for (int s = nsel + tid; s < p.max_det; s += NT) {
    float* bo = boxes + (b * p.max_det + s) * 4;
    bo[0] = bo[1] = bo[2] = bo[3] = 0.0f;
    scores[b * p.max_det + s] = 0.0f;
    classes[b * p.max_det + s] = 0.0f;
  }
  if (tid == 0) counts[b] = nsel;
  POST_STAMP(6);
#ifdef VBT_POST_PROF
  if (tid == 0 && blockIdx.x == 0) atomicAdd(&g_post_prof[10], 1ull);
#endif
}

// ------------------------------------------------------------------------------------------
// preprocess_image (reference odt.py:10-19): tf.image.resize bilinear with half-pixel centres
// [EXTERNAL TF2 ResizeBilinear: in = (out+0.5)*scale-0.5, lower = max(floor(in),0),
// upper = min(ceil(in), size-1), lerp = in - floor(in); top + (bottom-top)*ly], float32,
// then tf.cast(..., uint8) = truncation.  Optional BGR->RGB swap (reference track.py:171).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                              long total, int H, int W, int h, int w, float sy, float sx,
                                                              int swap_rb, int compact) {
  long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  int ox = (int)(idx % w);
  long t = idx / w;
  int oy = (int)(t % h);
  long b = t / h;
  float iy = ((float)oy + 0.5f) * sy - 0.5f, ix = ((float)ox + 0.5f) * sx - 0.5f;
  float fy = floorf(iy), fx = floorf(ix);
  int y0 = max((int)fy, 0), y1 = min((int)ceilf(iy), H - 1);
  int x0 = max((int)fx, 0), x1 = min((int)ceilf(ix), W - 1);
  float ly = iy - fy, lx = ix - fx;
  // compact: the source holds only the row pairs the resize reads, [B][2h][W][3]: pair oy = source rows p, p + 1 with p = min(y0, H - 2)
  // (the host-fed pipeline uploads nothing else, pipeline.hip); y0 and y1 then become rows 2 oy + (y - p)
  const uint8_t* s = src + b * (long)(compact ? 2 * h : H) * W * 3;
  if (compact) {
    const int p = min(y0, H - 2);
    y0 = 2 * oy + (y0 - p);
    y1 = 2 * oy + (y1 - p);
  }
  uint8_t* d = dst + ((b * h + oy) * (long)w + ox) * 3;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    float tl = (float)s[((long)y0 * W + x0) * 3 + c], tr = (float)s[((long)y0 * W + x1) * 3 + c];
    float bl = (float)s[((long)y1 * W + x0) * 3 + c], br = (float)s[((long)y1 * W + x1) * 3 + c];
    float top = tl + (tr - tl) * lx;
    float bot = bl + (br - bl) * lx;
    float v = top + (bot - top) * ly;
    d[swap_rb ? 2 - c : c] = (uint8_t)(int)v;
  }
}
