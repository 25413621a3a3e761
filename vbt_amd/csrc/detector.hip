// EfficientDet-Lite int8 detector on gfx950 (MI355X): kernels + execution plan + C ABI.
//
// Replaces the TFLite interpreter invoke at reference odt.py:58-66 (signature_fn(images=...)).
// Arithmetic contract (bit-exact with oracle/detector.c, which restates TFLite/XNNPACK int8):
//   conv:  acc(int32) = sum (x_q - z_x) * w_q + bias_q ;  q = clamp(rne(float(acc) * M[c]) + z_y)
//   add :  r = (a-z_a)*k_a ; r = fma(b-z_b, k_b, r) [; r = fma(c-z_c, k_c, r)] ; q = clamp(rne(r)+z_y)
//   post:  table look-ups + single IEEE ops, greedy NMS in (score desc, anchor asc) order.
// Design (MI355X-first):
//   * activations int8 NHWC, batch-major [B][H][W][C]; pointwise convs run on the int8 MFMA
//     (v_mfma_i32_16x16x32_i8) with the WEIGHTS as the A operand so every lane ends up holding 16
//     consecutive output channels of one pixel -> one 16-byte coalesced store per lane;
//   * depthwise convs convert bytes with v_cvt_f32_ubyteN and accumulate with v_fma_f32 (exact:
//     |sum| < 2^24), 4 channels per lane so a wave reads contiguous NHWC channel vectors;
//   * zero points are folded into the bias on the host at load time; padding uses the zero point;
//   * decode + NMS: one workgroup per frame, 256-bin score histogram -> bitonic sort of the top
//     candidates in LDS -> greedy suppression by one wavefront with ballot/shuffle.
#include <algorithm>
#include <cmath>
#include <map>

#include "common.h"

namespace vbt {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

typedef int v4i __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int requant(int acc, float mult, int zp, int lo, int hi) {
  float t = (float)acc * mult;
  t = fminf(fmaxf(t, -65536.0f), 65536.0f);
  int q = (int)__builtin_rintf(t) + zp;  // v_rndne_f32: round-to-nearest-even
  return min(max(q, lo), hi);
}
__device__ __forceinline__ unsigned pack4(int a, int b, int c, int d) {
  return (unsigned)(a & 255) | ((unsigned)(b & 255) << 8) | ((unsigned)(c & 255) << 16) | ((unsigned)(d & 255) << 24);
}

struct Epi {  // requantisation parameters of one conv
  const int* bias;    // folded bias, padded to NB*64
  const float* mult;  // padded to NB*64
  int zp, lo, hi;
};

// Lane (r = lane&15 pixel, g = lane>>4) holds acc[t][j] = channel nb*64 + 16g + 4t + j of pixel r.
__device__ __forceinline__ void store_tile(const v4i acc[4], const Epi& e, int8_t* __restrict__ out, long m, int N,
                                           int nb, int g) {
  int c0 = nb * 64 + 16 * g;
  if (c0 >= N) return;
  unsigned d[4];
#pragma unroll
  for (int t = 0; t < 4; t++) {
    int4 b = *(const int4*)(e.bias + c0 + 4 * t);
    float4 mu = *(const float4*)(e.mult + c0 + 4 * t);
    d[t] = pack4(requant(acc[t][0] + b.x, mu.x, e.zp, e.lo, e.hi), requant(acc[t][1] + b.y, mu.y, e.zp, e.lo, e.hi),
                 requant(acc[t][2] + b.z, mu.z, e.zp, e.lo, e.hi), requant(acc[t][3] + b.w, mu.w, e.zp, e.lo, e.hi));
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

// ------------------------------------------------------------------------------------------
// pointwise conv, variant A: K <= 256, activations of 16*MS pixels stay in registers while the
// wave walks over the output-channel blocks.  wp = packed weights [nb][ks][t][lane] x 8 bytes.
// ------------------------------------------------------------------------------------------
template <int KS, int MS>
__global__ __launch_bounds__(256) void pw_a_kernel(const int8_t* __restrict__ x, const long* __restrict__ wp, Epi e,
                                                   int8_t* __restrict__ out, long M, int K, int N, int NB,
                                                   int nb_per_y) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = lane & 15, g = lane >> 4;
  const long m0 = ((long)blockIdx.x * 4 + wave) * (16 * MS);
  if (m0 >= M) return;
  long a[MS][KS];
#pragma unroll
  for (int ms = 0; ms < MS; ms++) {
    long m = min(m0 + 16 * ms + r, M - 1);
    const int8_t* p = x + m * K + 8 * g;
#pragma unroll
    for (int ks = 0; ks < KS; ks++) a[ms][ks] = *(const long*)(p + 32 * ks);
  }
  const int nb0 = blockIdx.y * nb_per_y, nb1 = min(nb0 + nb_per_y, NB);
  for (int nb = nb0; nb < nb1; nb++) {
    v4i acc[MS][4];
#pragma unroll
    for (int ms = 0; ms < MS; ms++)
#pragma unroll
      for (int t = 0; t < 4; t++) acc[ms][t] = (v4i){0, 0, 0, 0};
    const long* w = wp + (long)nb * KS * 4 * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < KS; ks++)
#pragma unroll
      for (int t = 0; t < 4; t++) {
        long wv = w[(ks * 4 + t) * 64];
#pragma unroll
        for (int ms = 0; ms < MS; ms++)
          acc[ms][t] = __builtin_amdgcn_mfma_i32_16x16x32_i8(wv, a[ms][ks], acc[ms][t], 0, 0, 0);
      }
#pragma unroll
    for (int ms = 0; ms < MS; ms++) {
      long m = m0 + 16 * ms + r;
      if (m < M) store_tile(acc[ms], e, out, m, N, nb, g);
    }
  }
}

// variant B: large K, few output channels: accumulators for NBT channel blocks stay in registers
// while the wave streams the K dimension of its 16 pixels.
template <int NBT>
__global__ __launch_bounds__(256) void pw_b_kernel(const int8_t* __restrict__ x, const long* __restrict__ wp, Epi e,
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
  const int8_t* p = x + min(m0 + r, M - 1) * K + 8 * g;
  for (int ks = 0; ks < KS; ks++) {
    long av = *(const long*)(p + 32 * ks);
#pragma unroll
    for (int i = 0; i < NBT; i++) {
      int nb = min(nb0 + i, NB - 1);
      const long* w = wp + ((long)(nb * KS + ks) * 4) * 64 + lane;
#pragma unroll
      for (int t = 0; t < 4; t++) acc[i][t] = __builtin_amdgcn_mfma_i32_16x16x32_i8(w[t * 64], av, acc[i][t], 0, 0, 0);
    }
  }
  long m = m0 + r;
  if (m < M) {
#pragma unroll
    for (int i = 0; i < NBT; i++)
      if (nb0 + i < NB) store_tile(acc[i], e, out, m, N, nb0 + i, g);
  }
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
      unsigned d = pack4(requant((int)acc[o][0] + bq.x, mu.x, e.zp, e.lo, e.hi), requant((int)acc[o][1] + bq.y, mu.y, e.zp, e.lo, e.hi),
                         requant((int)acc[o][2] + bq.z, mu.z, e.zp, e.lo, e.hi), requant((int)acc[o][3] + bq.w, mu.w, e.zp, e.lo, e.hi));
      *(unsigned*)(ob + (long)ox * C) = d;
    }
  }
}

// ------------------------------------------------------------------------------------------
// elementwise n-ary add with requantisation: 4 bytes per lane
// ------------------------------------------------------------------------------------------
struct AddArgs {
  const int8_t* in[3];
  int z[3];
  float k[3];
  int n_in, zo, lo, hi;
};
__global__ __launch_bounds__(256) void add_kernel(AddArgs a, int8_t* __restrict__ out, long n4) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  unsigned va = ((const unsigned*)a.in[0])[i], vb = ((const unsigned*)a.in[1])[i];
  unsigned vc = a.n_in > 2 ? ((const unsigned*)a.in[2])[i] : 0u;
  int q[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    int xa = (int)(int8_t)(va >> (8 * j)), xb = (int)(int8_t)(vb >> (8 * j));
    float r = (float)(xa - a.z[0]) * a.k[0];
    r = __builtin_fmaf((float)(xb - a.z[1]), a.k[1], r);
    if (a.n_in > 2) {
      int xc = (int)(int8_t)(vc >> (8 * j));
      r = __builtin_fmaf((float)(xc - a.z[2]), a.k[2], r);
    }
    int v = (int)__builtin_rintf(r) + a.zo;
    q[j] = min(max(v, a.lo), a.hi);
  }
  ((unsigned*)out)[i] = pack4(q[0], q[1], q[2], q[3]);
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
// TFLite_Detection_PostProcess (fast single-class path): one workgroup per frame.
// ------------------------------------------------------------------------------------------
struct PostArgs {
  const int8_t* cls[5];
  const int8_t* box[5];
  int base[6];         // first anchor index of each level, base[5] = A
  const float* anchors;  // [A][4] ycenter, xcenter, h, w
  const float* luts;     // score[256], delta[256], exp[256]
  int A, max_det, qmin;  // qmin: lowest int8 class value whose score >= nms_score_threshold (128 = none)
  float iou_thr;
};
constexpr int POST_CAP = 2048;

__device__ __forceinline__ float iou_box(float4 a, float4 b) {  // (ymin, xmin, ymax, xmax)
  float area_a = (a.z - a.x) * (a.w - a.y);
  float area_b = (b.z - b.x) * (b.w - b.y);
  if (area_a <= 0.0f || area_b <= 0.0f) return 0.0f;
  float iy0 = fmaxf(a.x, b.x), ix0 = fmaxf(a.y, b.y);
  float iy1 = fminf(a.z, b.z), ix1 = fminf(a.w, b.w);
  float inter = fmaxf(iy1 - iy0, 0.0f) * fmaxf(ix1 - ix0, 0.0f);
  return inter / (area_a + area_b - inter);
}

__global__ __launch_bounds__(256) void postprocess_kernel(PostArgs p, float* __restrict__ boxes, float* __restrict__ scores,
                                                          float* __restrict__ classes, int* __restrict__ counts) {
  __shared__ int hist[256];
  __shared__ unsigned keys[POST_CAP];
  __shared__ float lut[768];
  __shared__ float4 selbox[VBT_MAX_DETECTIONS + 7];
  __shared__ int s_n, s_qlo, s_qhi, s_i0, s_i1, s_nsel, s_done;
  const int tid = threadIdx.x;
  const long b = blockIdx.x;
  const int8_t* cls[5];
  const int8_t* box[5];
#pragma unroll
  for (int l = 0; l < 5; l++) {
    int nl = p.base[l + 1] - p.base[l];
    cls[l] = p.cls[l] + b * nl;
    box[l] = p.box[l] + b * (long)nl * 4;
  }
  hist[tid] = 0;
  for (int i = tid; i < 768; i += 256) lut[i] = p.luts[i];
  if (tid == 0) { s_nsel = 0; s_done = 0; }
  __syncthreads();
  // pass 1: histogram of the class bytes
  for (int l = 0; l < 5; l++) {
    int nl = p.base[l + 1] - p.base[l];
    for (int i = tid; i < nl; i += 256) atomicAdd(&hist[(int)cls[l][i] + 128], 1);
  }
  __syncthreads();
  int qcur = 127;   // highest class byte not yet consumed (uniform across the block)
  int seg0 = 0;     // for an oversized bin: next anchor index to scan
  while (true) {
    if (tid == 0) {
      // pick the next range of score bins (and, for one oversized bin, a slice of anchor indices)
      int q = qcur;
      while (q >= p.qmin && hist[q + 128] == 0) q--;
      if (q < p.qmin) {
        s_done = 1;
      } else if (hist[q + 128] > POST_CAP) {
        s_qhi = q; s_qlo = q; s_i0 = seg0; s_i1 = min(seg0 + POST_CAP, p.A);
      } else {
        int tot = 0, qlo = q;
        while (qlo >= p.qmin && tot + hist[qlo + 128] <= POST_CAP && tot < 256) { tot += hist[qlo + 128]; qlo--; }
        s_qhi = q; s_qlo = qlo + 1; s_i0 = 0; s_i1 = p.A;
      }
      s_n = 0;
    }
    __syncthreads();
    if (s_done) break;
    const int qhi = s_qhi, qlo = s_qlo, i0 = s_i0, i1 = s_i1;
    // pass 2: collect candidate keys = (127 - q) << 16 | anchor  (ascending key = score desc, anchor asc)
    for (int l = 0; l < 5; l++) {
      int lo = max(i0, p.base[l]), hi = min(i1, p.base[l + 1]);
      for (int i = lo + tid; i < hi; i += 256) {
        int q = cls[l][i - p.base[l]];
        if (q >= qlo && q <= qhi) {
          int pos = atomicAdd(&s_n, 1);
          keys[pos] = ((unsigned)(127 - q) << 16) | (unsigned)i;
        }
      }
    }
    __syncthreads();
    const int n = s_n;
    int n2 = 64;
    while (n2 < n) n2 <<= 1;
    for (int i = n + tid; i < n2; i += 256) keys[i] = 0xFFFFFFFFu;
    __syncthreads();
    for (int k = 2; k <= n2; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < n2; i += 256) {
          int ixj = i ^ j;
          if (ixj > i) {
            unsigned a = keys[i], c = keys[ixj];
            bool up = (i & k) == 0;
            if ((a > c) == up) { keys[i] = c; keys[ixj] = a; }
          }
        }
        __syncthreads();
      }
    // greedy suppression by wavefront 0
    if (tid < 64) {
      int nsel = s_nsel;
      for (int base = 0; base < n && nsel < p.max_det; base += 64) {
        int ci = base + tid;
        bool alive = ci < n;
        float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
        float sc = 0.f;
        if (alive) {
          unsigned key = keys[ci];
          int a = (int)(key & 0xFFFFu);
          int q = 127 - (int)(key >> 16);
          int l = 0;
#pragma unroll
          for (int t = 1; t < 5; t++) l += (a >= p.base[t]) ? 1 : 0;
          unsigned bq = *(const unsigned*)(box[l] + (long)(a - p.base[l]) * 4);
          float4 an = *(const float4*)(p.anchors + (long)a * 4);
          float yc = __builtin_fmaf(lut[256 + (int)((bq & 255u) ^ 128u)], an.z, an.x);
          float xc = __builtin_fmaf(lut[256 + (int)(((bq >> 8) & 255u) ^ 128u)], an.w, an.y);
          float hh = (0.5f * lut[512 + (int)(((bq >> 16) & 255u) ^ 128u)]) * an.z;
          float hw = (0.5f * lut[512 + (int)((bq >> 24) ^ 128u)]) * an.w;
          bx = make_float4(yc - hh, xc - hw, yc + hh, xc + hw);
          sc = lut[q + 128];
          for (int s = 0; s < nsel; s++)
            if (iou_box(selbox[s], bx) > p.iou_thr) { alive = false; break; }
        }
        while (nsel < p.max_det) {
          unsigned long long mask = __ballot(alive);
          if (mask == 0ull) break;
          int j = __ffsll((long long)mask) - 1;
          float4 sb;
          sb.x = __shfl(bx.x, j); sb.y = __shfl(bx.y, j); sb.z = __shfl(bx.z, j); sb.w = __shfl(bx.w, j);
          float ss = __shfl(sc, j);
          if (tid == 0) {
            selbox[nsel] = sb;
            float* bo = boxes + (b * p.max_det + nsel) * 4;
            bo[0] = sb.x; bo[1] = sb.y; bo[2] = sb.z; bo[3] = sb.w;
            scores[b * p.max_det + nsel] = ss;
            classes[b * p.max_det + nsel] = 0.0f;
          }
          __threadfence_block();  // selbox[] is read by the other lanes of this wavefront
          nsel++;
          if (tid == j) alive = false;
          else if (alive && iou_box(sb, bx) > p.iou_thr) alive = false;
        }
      }
      if (tid == 0) s_nsel = nsel;
    }
    __syncthreads();
    if (s_nsel >= p.max_det) break;
    if (qhi == qlo && hist[qhi + 128] > POST_CAP && i1 < p.A) { seg0 = i1; qcur = qhi; }
    else { seg0 = 0; qcur = qlo - 1; }
    __syncthreads();
  }
  const int nsel = s_nsel;
  for (int s = nsel + tid; s < p.max_det; s += 256) {
    float* bo = boxes + (b * p.max_det + s) * 4;
    bo[0] = bo[1] = bo[2] = bo[3] = 0.0f;
    scores[b * p.max_det + s] = 0.0f;
    classes[b * p.max_det + s] = 0.0f;
  }
  if (tid == 0) counts[b] = nsel;
}

// ------------------------------------------------------------------------------------------
// preprocess_image (reference odt.py:10-19): tf.image.resize bilinear with half-pixel centres
// [EXTERNAL TF2 ResizeBilinear: in = (out+0.5)*scale-0.5, lower = max(floor(in),0),
// upper = min(ceil(in), size-1), lerp = in - floor(in); top + (bottom-top)*ly], float32,
// then tf.cast(..., uint8) = truncation.  Optional BGR->RGB swap (reference track.py:171).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                              long total, int H, int W, int h, int w, float sy, float sx,
                                                              int swap_rb) {
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
  const uint8_t* s = src + b * (long)H * W * 3;
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

// ------------------------------------------------------------------------------------------
// host: model, plan, launches
// ------------------------------------------------------------------------------------------
enum Family { F_STEM = 0, F_PW, F_DW, F_ADD, F_MAXPOOL, F_RESIZE, F_POST, F_COUNT };
static const char* kFamilyName[F_COUNT] = {"stem_conv_mfma_i8", "pw_conv_mfma_i8", "dw_conv_f32acc", "add_requant",
                                           "maxpool3x3s2", "resize_nn", "decode_nms"};

struct Step {
  int op;       // index into ops
  int family;
  // conv
  long* wp = nullptr;      // packed MFMA weights (device)
  float* wf = nullptr;     // depthwise weights as float [k*k][C] (device)
  int* bias = nullptr;     // folded bias (device, padded)
  float* mult = nullptr;   // multipliers (device, padded)
  int KS = 0, NB = 0;
  double alg_bytes_per_frame = 0, weight_bytes = 0, macs_per_frame = 0;
};

}  // namespace vbt

using namespace vbt;

struct vbt_model {
  Header hdr;
  std::vector<TensorRec> tensors;
  std::vector<OpRec> ops;
  std::vector<uint8_t> blob;
  int device = 0, max_batch = 0;
  std::vector<int8_t*> tptr;   // device pointer of each tensor ([max_batch][h][w][c])
  std::vector<size_t> telems;  // per-frame elements
  int8_t* arena = nullptr;
  uint8_t* frames_stage = nullptr;  // device staging for host frames
  float* out_boxes = nullptr;       // device staging for host outputs
  float* out_scores = nullptr;
  float* out_classes = nullptr;
  int* out_counts = nullptr;
  float* d_anchors = nullptr;
  float* d_luts = nullptr;
  std::vector<Step> steps;
  std::vector<void*> owned;  // device allocations to free
  int last_B = 0;
};

namespace vbt {

template <typename T>
static int upload(vbt_model* m, const std::vector<T>& h, T** d) {
  size_t bytes = std::max<size_t>(h.size() * sizeof(T), 16);
  VBT_HIP_CHECK(hipMalloc((void**)d, bytes + 64));
  m->owned.push_back(*d);
  if (!h.empty()) VBT_HIP_CHECK(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return VBT_OK;
}

// Packed MFMA A-operand layout: [(nb*KS + ks)*4 + t][lane][8 bytes]; lane (i = lane&15, g = lane>>4) holds
// W[cout = 64nb + 16(i>>2) + 4t + (i&3)][k = 32ks + 8g + j].  kmap translates packed k -> source k (or -1).
static void pack_weights(const int8_t* w, int N, int K, int KS, int NB, const std::vector<int>* kmap, std::vector<long>& out) {
  out.assign((size_t)NB * KS * 4 * 64, 0);
  int8_t* o = (int8_t*)out.data();
  for (int nb = 0; nb < NB; nb++)
    for (int ks = 0; ks < KS; ks++)
      for (int t = 0; t < 4; t++)
        for (int lane = 0; lane < 64; lane++) {
          int i = lane & 15, g = lane >> 4;
          int co = 64 * nb + 16 * (i >> 2) + 4 * t + (i & 3);
          for (int j = 0; j < 8; j++) {
            int kp = 32 * ks + 8 * g + j;
            int k = kmap ? (kp < (int)kmap->size() ? (*kmap)[kp] : -1) : (kp < K ? kp : -1);
            int8_t v = (co < N && k >= 0) ? w[(size_t)co * K + k] : 0;
            o[((((size_t)(nb * KS + ks) * 4 + t) * 64 + lane) * 8) + j] = v;
          }
        }
}

static int build_plan(vbt_model* m) {
  const int no = (int)m->ops.size();
  for (int oi = 0; oi < no; oi++) {
    const OpRec& op = m->ops[oi];
    const TensorRec& to = m->tensors[op.output];
    Step s;
    s.op = oi;
    double in_el = 0;
    for (int i = 0; i < op.n_inputs; i++) {
      const TensorRec& ti = m->tensors[op.inputs[i]];
      in_el += (double)ti.h * ti.w * ti.c;
    }
    double out_el = (double)to.h * to.w * to.c;
    s.alg_bytes_per_frame = in_el + out_el;
    if (op.type == OP_STEM || op.type == OP_PW || op.type == OP_DW) {
      const TensorRec& ti = m->tensors[op.inputs[0]];
      const int8_t* w = (const int8_t*)(m->blob.data() + op.w_off);
      const int32_t* bq = (const int32_t*)(m->blob.data() + op.b_off);
      const float* mu = (const float*)(m->blob.data() + op.m_off);
      const int N = to.c;
      const int zx = ti.zero_point;
      if (op.type == OP_DW) {
        s.family = F_DW;
        const int C = N, kk = op.k * op.k;
        if (C % 4 != 0 || !((op.k == 3 || op.k == 5) && (op.stride == 1 || op.stride == 2))) {
          set_error("unsupported depthwise conv: C=%d k=%d s=%d", C, op.k, op.stride);
          return VBT_ERR_ARG;
        }
        std::vector<float> wf((size_t)kk * C);
        std::vector<int> bias(C);
        std::vector<float> mult(mu, mu + C);
        for (int c = 0; c < C; c++) {
          long sw = 0;
          for (int t = 0; t < kk; t++) { wf[(size_t)t * C + c] = (float)w[(size_t)t * C + c]; sw += w[(size_t)t * C + c]; }
          bias[c] = (int)((long)bq[c] - (long)(128 + zx) * sw);  // acc uses u = x_q + 128, pad u = 128 + z_x
        }
        int rc;
        if ((rc = upload(m, wf, &s.wf)) || (rc = upload(m, bias, &s.bias)) || (rc = upload(m, mult, &s.mult))) return rc;
        s.weight_bytes = (double)kk * C;
        s.macs_per_frame = out_el * kk;
      } else {
        const bool stem = op.type == OP_STEM;
        s.family = stem ? F_STEM : F_PW;
        const int K = stem ? op.k * op.k * ti.c : ti.c;
        if (stem && (op.k != 3 || ti.c != 3 || op.stride != 2)) { set_error("unsupported stem conv"); return VBT_ERR_ARG; }
        if (!stem && (K % 8) != 0) { set_error("pointwise conv needs Cin %% 8 == 0 (got %d)", K); return VBT_ERR_ARG; }
        s.KS = (K + 31) / 32;
        s.NB = (N + 63) / 64;
        std::vector<int> kmap;
        if (stem) {
          kmap.assign(32, -1);
          for (int g = 0; g < 3; g++)
            for (int j = 0; j < 8; j++) kmap[8 * g + j] = (g * 3 + j / 3) * 3 + j % 3;  // (ky=g, kx=j/3, c=j%3)
          for (int j = 0; j < 3; j++) kmap[24 + j] = (j * 3 + 2) * 3 + 2;                // (ky=j, kx=2, c=2)
        }
        std::vector<long> wp;
        pack_weights(w, N, K, s.KS, s.NB, stem ? &kmap : nullptr, wp);
        std::vector<int> bias((size_t)s.NB * 64, 0);
        std::vector<float> mult((size_t)s.NB * 64, 0.0f);
        for (int c = 0; c < N; c++) {
          long sw = 0;
          for (int k = 0; k < K; k++) sw += w[(size_t)c * K + k];
          bias[c] = (int)((long)bq[c] - (long)zx * sw);  // acc = sum x_q*w ; (x_q - z_x) folded here
          mult[c] = mu[c];
        }
        int rc;
        if ((rc = upload(m, wp, &s.wp)) || (rc = upload(m, bias, &s.bias)) || (rc = upload(m, mult, &s.mult))) return rc;
        s.weight_bytes = (double)N * K;
        s.macs_per_frame = out_el * K;
      }
    } else if (op.type == OP_ADD) {
      s.family = F_ADD;
      if (((long)to.h * to.w * to.c) % 4 != 0 || op.n_inputs < 2 || op.n_inputs > 3) { set_error("unsupported add"); return VBT_ERR_ARG; }
    } else if (op.type == OP_MAXPOOL) {
      s.family = F_MAXPOOL;
      if (to.c % 4 != 0 || op.k != 3 || op.stride != 2) { set_error("unsupported maxpool"); return VBT_ERR_ARG; }
    } else if (op.type == OP_RESIZE_NN) {
      s.family = F_RESIZE;
      if (to.c % 4 != 0) { set_error("unsupported resize"); return VBT_ERR_ARG; }
    } else if (op.type == OP_POSTPROCESS) {
      s.family = F_POST;
      if (op.n_inputs != 10 || m->hdr.max_detections != VBT_MAX_DETECTIONS || m->hdr.num_anchors > 65535) {
        set_error("unsupported postprocess configuration");
        return VBT_ERR_ARG;
      }
      s.alg_bytes_per_frame = in_el + m->hdr.max_detections * 24.0;
      s.weight_bytes = (double)m->hdr.num_anchors * 16;
    } else {
      set_error("unknown op type %d", op.type);
      return VBT_ERR_ARG;
    }
    m->steps.push_back(s);
  }
  return VBT_OK;
}

template <int KS>
static void launch_pw_a(int MS, dim3 grid, hipStream_t st, const int8_t* x, const long* wp, Epi e, int8_t* out, long M, int K,
                        int N, int NB, int nb_per_y) {
  if (MS == 2) pw_a_kernel<KS, 2><<<grid, 256, 0, st>>>(x, wp, e, out, M, K, N, NB, nb_per_y);
  else pw_a_kernel<KS, 1><<<grid, 256, 0, st>>>(x, wp, e, out, M, K, N, NB, nb_per_y);
}

static int launch_step(vbt_model* m, const Step& s, int B, hipStream_t st, const uint8_t* frames, float* boxes, float* scores,
                       float* classes, int* counts) {
  const OpRec& op = m->ops[s.op];
  const TensorRec& to = m->tensors[op.output];
  int8_t* out = m->tptr[op.output];
  Epi e{s.bias, s.mult, to.zero_point, op.act_min, op.act_max};
  switch (s.family) {
    case F_STEM: {
      const TensorRec& ti = m->tensors[op.inputs[0]];
      long M = (long)B * to.h * to.w;
      dim3 grid((unsigned)((M + 63) / 64));
      stem_kernel<<<grid, 256, 0, st>>>(frames, s.wp, e, out, M, ti.h, ti.w, to.h, to.w, to.c, op.pad_t, op.pad_l, ti.zero_point);
      break;
    }
    case F_PW: {
      const TensorRec& ti = m->tensors[op.inputs[0]];
      const int8_t* x = m->tptr[op.inputs[0]];
      long M = (long)B * to.h * to.w;
      int K = ti.c, N = to.c;
      if (s.KS <= 8) {
        int MS = M >= 32768 ? 2 : 1;
        long waves = (M + 16 * MS - 1) / (16 * MS);
        unsigned gx = (unsigned)((waves + 3) / 4);
        int ysplit = 1;
        while (gx * ysplit < 1024 && ysplit < s.NB) ysplit++;
        int nb_per_y = (s.NB + ysplit - 1) / ysplit;
        ysplit = (s.NB + nb_per_y - 1) / nb_per_y;
        dim3 grid(gx, ysplit);
        switch (s.KS) {
          case 1: launch_pw_a<1>(MS, grid, st, x, s.wp, e, out, M, K, N, s.NB, nb_per_y); break;
          case 2: launch_pw_a<2>(MS, grid, st, x, s.wp, e, out, M, K, N, s.NB, nb_per_y); break;
          case 3: launch_pw_a<3>(MS, grid, st, x, s.wp, e, out, M, K, N, s.NB, nb_per_y); break;
          case 4: launch_pw_a<4>(MS, grid, st, x, s.wp, e, out, M, K, N, s.NB, nb_per_y); break;
          case 5: launch_pw_a<5>(MS, grid, st, x, s.wp, e, out, M, K, N, s.NB, nb_per_y); break;
          case 6: launch_pw_a<6>(MS, grid, st, x, s.wp, e, out, M, K, N, s.NB, nb_per_y); break;
          case 7: launch_pw_a<7>(MS, grid, st, x, s.wp, e, out, M, K, N, s.NB, nb_per_y); break;
          default: launch_pw_a<8>(MS, grid, st, x, s.wp, e, out, M, K, N, s.NB, nb_per_y); break;
        }
      } else {
        long waves = (M + 15) / 16;
        unsigned gx = (unsigned)((waves + 3) / 4);
        int nbt = std::min(s.NB, 4);
        if (gx < 512 && nbt > 2) nbt = 2;  // more workgroups for the low-resolution layers
        if (gx < 128) nbt = 1;
        dim3 grid(gx, (s.NB + nbt - 1) / nbt);
        switch (nbt) {
          case 1: pw_b_kernel<1><<<grid, 256, 0, st>>>(x, s.wp, e, out, M, K, s.KS, N, s.NB); break;
          case 2: pw_b_kernel<2><<<grid, 256, 0, st>>>(x, s.wp, e, out, M, K, s.KS, N, s.NB); break;
          case 3: pw_b_kernel<3><<<grid, 256, 0, st>>>(x, s.wp, e, out, M, K, s.KS, N, s.NB); break;
          default: pw_b_kernel<4><<<grid, 256, 0, st>>>(x, s.wp, e, out, M, K, s.KS, N, s.NB); break;
        }
      }
      break;
    }
    case F_DW: {
      const TensorRec& ti = m->tensors[op.inputs[0]];
      const int8_t* x = m->tptr[op.inputs[0]];
      int C = to.c;
      long total = (long)B * to.h * ((to.w + 3) / 4) * (C / 4);
      dim3 grid((unsigned)((total + 255) / 256));
      unsigned pb = (unsigned)((128 + ti.zero_point) & 255);
      unsigned pad4 = pb | (pb << 8) | (pb << 16) | (pb << 24);
#define DW_LAUNCH(KK, S) dw_kernel<KK, S><<<grid, 256, 0, st>>>(x, s.wf, e, out, total, ti.h, ti.w, C, to.h, to.w, op.pad_t, op.pad_l, pad4)
      if (op.k == 3 && op.stride == 1) DW_LAUNCH(3, 1);
      else if (op.k == 3 && op.stride == 2) DW_LAUNCH(3, 2);
      else if (op.k == 5 && op.stride == 1) DW_LAUNCH(5, 1);
      else DW_LAUNCH(5, 2);
#undef DW_LAUNCH
      break;
    }
    case F_ADD: {
      AddArgs a;
      a.n_in = op.n_inputs;
      for (int i = 0; i < 3; i++) {
        int ii = i < op.n_inputs ? i : 0;
        a.in[i] = m->tptr[op.inputs[ii]];
        a.z[i] = m->tensors[op.inputs[ii]].zero_point;
        a.k[i] = op.in_mult[ii];
      }
      a.zo = to.zero_point; a.lo = op.act_min; a.hi = op.act_max;
      long n4 = (long)B * to.h * to.w * to.c / 4;
      add_kernel<<<dim3((unsigned)((n4 + 255) / 256)), 256, 0, st>>>(a, out, n4);
      break;
    }
    case F_MAXPOOL: {
      const TensorRec& ti = m->tensors[op.inputs[0]];
      long total = (long)B * to.h * to.w * (to.c / 4);
      maxpool_kernel<<<dim3((unsigned)((total + 255) / 256)), 256, 0, st>>>(m->tptr[op.inputs[0]], out, total, ti.h, ti.w, ti.c,
                                                                             to.h, to.w, op.pad_t, op.pad_l);
      break;
    }
    case F_RESIZE: {
      const TensorRec& ti = m->tensors[op.inputs[0]];
      long total = (long)B * to.h * to.w * (to.c / 4);
      resize_kernel<<<dim3((unsigned)((total + 255) / 256)), 256, 0, st>>>(m->tptr[op.inputs[0]], out, total, ti.h, ti.w, ti.c,
                                                                            to.h, to.w);
      break;
    }
    case F_POST: {
      PostArgs p;
      int base = 0;
      for (int l = 0; l < 5; l++) {
        const TensorRec& tc = m->tensors[op.inputs[l]];
        p.cls[l] = m->tptr[op.inputs[l]];
        p.box[l] = m->tptr[op.inputs[5 + l]];
        p.base[l] = base;
        base += tc.h * tc.w * tc.c;
      }
      p.base[5] = base;
      p.anchors = m->d_anchors;
      p.luts = m->d_luts;
      p.A = m->hdr.num_anchors;
      p.max_det = m->hdr.max_detections;
      p.iou_thr = m->hdr.nms_iou_threshold;
      const float* lut = (const float*)(m->blob.data() + op.aux2_off);
      int qmin = 128;
      for (int q = 127; q >= -128; q--)
        if (lut[q + 128] >= m->hdr.nms_score_threshold) qmin = q; else break;
      p.qmin = qmin;
      postprocess_kernel<<<dim3((unsigned)B), 256, 0, st>>>(p, boxes, scores, classes, counts);
      break;
    }
  }
  return VBT_OK;
}

static int enqueue_forward(vbt_model* m, const uint8_t* frames_dev, int B, hipStream_t st, float* boxes, float* scores,
                           float* classes, int* counts, hipEvent_t* evs) {
  int i = 0;
  for (const Step& s : m->steps) {
    if (evs) (void)hipEventRecord(evs[i], st);
    int rc = launch_step(m, s, B, st, frames_dev, boxes, scores, classes, counts);
    if (rc) return rc;
    i++;
  }
  if (evs) (void)hipEventRecord(evs[i], st);
  VBT_HIP_CHECK(hipGetLastError());
  m->last_B = B;
  return VBT_OK;
}

}  // namespace vbt

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" {

const char* vbt_last_error(void) { return vbt::g_err; }

int vbt_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int vbt_model_create(const char* path, int device, int max_batch, vbt_model** out) {
  if (!path || !out || max_batch < 1) { set_error("vbt_model_create: bad argument"); return VBT_ERR_ARG; }
  *out = nullptr;
  FILE* f = fopen(path, "rb");
  if (!f) { set_error("cannot open model container '%s'", path); return VBT_ERR_IO; }
  vbt_model* m = new vbt_model();
  bool ok = fread(&m->hdr, sizeof(Header), 1, f) == 1 && memcmp(m->hdr.magic, "VBTM0001", 8) == 0;
  if (ok) {
    m->tensors.resize(m->hdr.num_tensors);
    m->ops.resize(m->hdr.num_ops);
    m->blob.resize(m->hdr.blob_bytes);
    ok = fread(m->tensors.data(), sizeof(TensorRec), m->tensors.size(), f) == m->tensors.size() &&
         fread(m->ops.data(), sizeof(OpRec), m->ops.size(), f) == m->ops.size() &&
         fseek(f, (long)m->hdr.blob_offset, SEEK_SET) == 0 &&
         fread(m->blob.data(), 1, m->blob.size(), f) == m->blob.size();
  }
  fclose(f);
  if (!ok) { delete m; set_error("'%s' is not a valid VBTM container", path); return VBT_ERR_IO; }
  m->device = device;
  m->max_batch = max_batch;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    delete m;
    set_error("vbt_model_create: HIP device %d not available (%d visible) - the HIP path has no CPU fallback", device, ndev);
    return VBT_ERR_HIP;
  }
  int rc = VBT_OK;
  auto fail = [&](int code) { vbt_model_destroy(m); return code; };
  if (hipSetDevice(device) != hipSuccess) { set_error("hipSetDevice(%d) failed", device); return fail(VBT_ERR_HIP); }
  // activation arena: every graph tensor keeps its own [max_batch][h][w][c] int8 buffer
  size_t total = 0;
  m->telems.resize(m->tensors.size());
  std::vector<size_t> off(m->tensors.size());
  for (size_t i = 0; i < m->tensors.size(); i++) {
    const TensorRec& t = m->tensors[i];
    m->telems[i] = (size_t)t.h * t.w * t.c;
    off[i] = total;
    size_t bytes = (int)i == m->hdr.input_tensor ? 0 : m->telems[i] * max_batch;
    total += (bytes + 255) / 256 * 256 + 256;
  }
  if (hipMalloc((void**)&m->arena, total + 4096) != hipSuccess) { set_error("hipMalloc(%zu) for activations failed", total); return fail(VBT_ERR_HIP); }
  (void)hipMemset(m->arena, 0, total + 4096);
  m->tptr.resize(m->tensors.size());
  for (size_t i = 0; i < m->tensors.size(); i++) m->tptr[i] = m->arena + off[i];
  size_t fbytes = (size_t)max_batch * m->hdr.image_size * m->hdr.image_size * 3;
  const int md = m->hdr.max_detections;
  if (hipMalloc((void**)&m->frames_stage, fbytes + 64) != hipSuccess || hipMalloc((void**)&m->out_boxes, (size_t)max_batch * md * 16) != hipSuccess ||
      hipMalloc((void**)&m->out_scores, (size_t)max_batch * md * 4) != hipSuccess || hipMalloc((void**)&m->out_classes, (size_t)max_batch * md * 4) != hipSuccess ||
      hipMalloc((void**)&m->out_counts, (size_t)max_batch * 4) != hipSuccess) {
    set_error("hipMalloc for staging buffers failed");
    return fail(VBT_ERR_HIP);
  }
  if ((rc = build_plan(m)) != VBT_OK) return fail(rc);
  for (const OpRec& op : m->ops)
    if (op.type == OP_POSTPROCESS) {
      std::vector<float> an((const float*)(m->blob.data() + op.aux_off), (const float*)(m->blob.data() + op.aux_off) + (size_t)m->hdr.num_anchors * 4);
      std::vector<float> lut((const float*)(m->blob.data() + op.aux2_off), (const float*)(m->blob.data() + op.aux2_off) + 768);
      if ((rc = upload(m, an, &m->d_anchors)) || (rc = upload(m, lut, &m->d_luts))) return fail(rc);
    }
  *out = m;
  return VBT_OK;
}

void vbt_model_destroy(vbt_model* m) {
  if (!m) return;
  for (void* p : m->owned) (void)hipFree(p);
  (void)hipFree(m->arena); (void)hipFree(m->frames_stage); (void)hipFree(m->out_boxes);
  (void)hipFree(m->out_scores); (void)hipFree(m->out_classes); (void)hipFree(m->out_counts);
  delete m;
}

int vbt_model_input_shape(const vbt_model* m, int shape[4]) {
  if (!m || !shape) { set_error("bad argument"); return VBT_ERR_ARG; }
  shape[0] = m->max_batch; shape[1] = m->hdr.image_size; shape[2] = m->hdr.image_size; shape[3] = 3;
  return VBT_OK;
}
int vbt_model_num_tensors(const vbt_model* m) { return m ? (int)m->tensors.size() : VBT_ERR_ARG; }
int vbt_model_num_ops(const vbt_model* m) { return m ? (int)m->ops.size() : VBT_ERR_ARG; }
int vbt_model_tensor_shape(const vbt_model* m, int id, int shape[3]) {
  if (!m || !shape || id < 0 || id >= (int)m->tensors.size()) { set_error("bad tensor id"); return VBT_ERR_ARG; }
  shape[0] = m->tensors[id].h; shape[1] = m->tensors[id].w; shape[2] = m->tensors[id].c;
  return VBT_OK;
}

int vbt_detect_async(vbt_model* m, const uint8_t* frames_dev, int B, void* stream, float* boxes, float* scores, float* classes,
                     int32_t* counts) {
  if (!m || !frames_dev || !boxes || !scores || !classes || !counts) { set_error("vbt_detect_async: NULL argument"); return VBT_ERR_ARG; }
  if (B < 1 || B > m->max_batch) { set_error("vbt_detect: batch %d outside 1..%d", B, m->max_batch); return VBT_ERR_CAPACITY; }
  return enqueue_forward(m, frames_dev, B, (hipStream_t)stream, boxes, scores, classes, counts, nullptr);
}

int vbt_detect(vbt_model* m, const uint8_t* frames, int B, int frames_on_device, void* stream, float* boxes, float* scores,
               float* classes, int32_t* counts, int outputs_on_device) {
  if (!m || !frames || !boxes || !scores || !classes || !counts) { set_error("vbt_detect: NULL argument"); return VBT_ERR_ARG; }
  if (B < 1 || B > m->max_batch) { set_error("vbt_detect: batch %d outside 1..%d", B, m->max_batch); return VBT_ERR_CAPACITY; }
  hipStream_t st = (hipStream_t)stream;
  VBT_HIP_CHECK(hipSetDevice(m->device));
  const uint8_t* fd = frames;
  size_t fbytes = (size_t)B * m->hdr.image_size * m->hdr.image_size * 3;
  if (!frames_on_device) {
    VBT_HIP_CHECK(hipMemcpyAsync(m->frames_stage, frames, fbytes, hipMemcpyHostToDevice, st));
    fd = m->frames_stage;
  }
  float *db = boxes, *ds = scores, *dc = classes;
  int* dn = counts;
  if (!outputs_on_device) { db = m->out_boxes; ds = m->out_scores; dc = m->out_classes; dn = m->out_counts; }
  int rc = enqueue_forward(m, fd, B, st, db, ds, dc, dn, nullptr);
  if (rc) return rc;
  if (!outputs_on_device) {
    const int md = m->hdr.max_detections;
    VBT_HIP_CHECK(hipMemcpyAsync(boxes, db, (size_t)B * md * 16, hipMemcpyDeviceToHost, st));
    VBT_HIP_CHECK(hipMemcpyAsync(scores, ds, (size_t)B * md * 4, hipMemcpyDeviceToHost, st));
    VBT_HIP_CHECK(hipMemcpyAsync(classes, dc, (size_t)B * md * 4, hipMemcpyDeviceToHost, st));
    VBT_HIP_CHECK(hipMemcpyAsync(counts, dn, (size_t)B * 4, hipMemcpyDeviceToHost, st));
    VBT_HIP_CHECK(hipStreamSynchronize(st));
  }
  return VBT_OK;
}

int vbt_model_read_tensor(vbt_model* m, int id, int B, int8_t* host_out) {
  if (!m || !host_out || id < 0 || id >= (int)m->tensors.size() || id == m->hdr.input_tensor) { set_error("bad tensor id"); return VBT_ERR_ARG; }
  if (B < 1 || B > m->max_batch) { set_error("bad batch"); return VBT_ERR_CAPACITY; }
  VBT_HIP_CHECK(hipDeviceSynchronize());
  VBT_HIP_CHECK(hipMemcpy(host_out, m->tptr[id], m->telems[id] * B, hipMemcpyDeviceToHost));
  return VBT_OK;
}

int vbt_resize_frames(const uint8_t* src, int B, int H, int W, int src_on_device, uint8_t* dst, int h, int w, int dst_on_device,
                      int swap_rb, int device, void* stream) {
  if (!src || !dst || B < 1 || H < 1 || W < 1 || h < 1 || w < 1) { set_error("vbt_resize_frames: bad argument"); return VBT_ERR_ARG; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    set_error("vbt_resize_frames: HIP device %d not available (%d visible) - no CPU fallback", device, ndev);
    return VBT_ERR_HIP;
  }
  VBT_HIP_CHECK(hipSetDevice(device));
  hipStream_t st = (hipStream_t)stream;
  size_t sb = (size_t)B * H * W * 3, db = (size_t)B * h * w * 3;
  uint8_t *ds = nullptr, *dd = nullptr;
  const uint8_t* sp = src;
  uint8_t* dp = dst;
  if (!src_on_device) {
    VBT_HIP_CHECK(hipMalloc((void**)&ds, sb));
    VBT_HIP_CHECK(hipMemcpyAsync(ds, src, sb, hipMemcpyHostToDevice, st));
    sp = ds;
  }
  if (!dst_on_device) {
    VBT_HIP_CHECK(hipMalloc((void**)&dd, db));
    dp = dd;
  }
  long total = (long)B * h * w;
  float sy = (float)H / (float)h, sx = (float)W / (float)w;
  resize_bilinear_kernel<<<dim3((unsigned)((total + 255) / 256)), 256, 0, st>>>(sp, dp, total, H, W, h, w, sy, sx, swap_rb);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && !dst_on_device) e = hipMemcpyAsync(dst, dd, db, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess && (ds || dd)) e = hipStreamSynchronize(st);
  if (ds) (void)hipFree(ds);
  if (dd) (void)hipFree(dd);
  if (e != hipSuccess) { set_error("vbt_resize_frames failed: %s", hipGetErrorString(e)); return VBT_ERR_HIP; }
  return VBT_OK;
}

int vbt_model_kernel_stats(const vbt_model* m, int B, vbt_kernel_stat* out, int cap, int* n) {
  if (!m || !out || !n || cap < F_COUNT) { set_error("bad argument"); return VBT_ERR_ARG; }
  for (int i = 0; i < F_COUNT; i++) {
    memset(&out[i], 0, sizeof(out[i]));
    snprintf(out[i].name, sizeof(out[i].name), "%s", kFamilyName[i]);
  }
  for (const Step& s : m->steps) {
    out[s.family].launches++;
    out[s.family].algorithmic_bytes += s.alg_bytes_per_frame * B + s.weight_bytes;
    out[s.family].macs += s.macs_per_frame * B;
  }
  *n = F_COUNT;
  return VBT_OK;
}

int vbt_model_profile(vbt_model* m, const uint8_t* frames_dev, int B, int reps, void* stream, double* ms_out, int cap) {
  if (!m || !frames_dev || !ms_out || cap < F_COUNT || reps < 1) { set_error("bad argument"); return VBT_ERR_ARG; }
  if (B < 1 || B > m->max_batch) { set_error("bad batch"); return VBT_ERR_CAPACITY; }
  hipStream_t st = (hipStream_t)stream;
  const int ns = (int)m->steps.size();
  std::vector<hipEvent_t> evs(ns + 1);
  for (auto& e : evs) VBT_HIP_CHECK(hipEventCreate(&e));
  for (int i = 0; i < F_COUNT; i++) ms_out[i] = 0.0;
  int rc = VBT_OK;
  for (int r = 0; r < reps && rc == VBT_OK; r++) {
    rc = enqueue_forward(m, frames_dev, B, st, m->out_boxes, m->out_scores, m->out_classes, m->out_counts, evs.data());
    if (rc) break;
    if (hipStreamSynchronize(st) != hipSuccess) { set_error("stream sync failed"); rc = VBT_ERR_HIP; break; }
    for (int i = 0; i < ns; i++) {
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, evs[i], evs[i + 1]);
      ms_out[m->steps[i].family] += ms;
    }
  }
  for (auto& e : evs) (void)hipEventDestroy(e);
  for (int i = 0; i < F_COUNT; i++) ms_out[i] /= reps;
  return rc;
}

}  // extern "C"
