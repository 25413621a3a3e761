// Device-side helpers shared by every kernel translation unit of libvbt_hip.so: the exact requantisation of the int8
// convolutions (XNNPACK qs8-qc8w, fp32 scale), the integer ADD (qs8-vadd-minmax) and the MFMA tile store.
#pragma once
#include "common.h"

namespace vbt {

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

// The same requantisation in 5 VALU ops + 1 pack op per element: clamp(rne(t) + zp, lo, hi) ==
// clamp(rne(t), lo - zp, hi - zp) + zp because every quantity after rne() is an exact small integer in
// fp32; the result is produced in the unsigned domain (q + 128 in [0,255]) so v_cvt_pk_u8_f32 can pack
// it, and one XOR 0x80808080 per dword turns the four bytes back into int8.
struct Rq {
  float lo_f, hi_f, off;  // lo - zp, hi - zp, zp + 128
  int full;               // lo == -128 && hi == 127: the clamp is the [0,255] saturation of v_cvt_pk_u8_f32 itself
  int kb;                 // every accumulator of this conv (bias included) lies in (-2^22, 2^22) - proven on the host from the weights
                          // (conv_acc_bound): kernels may start their accumulators at bias + RQ_KBIAS and read them back as floats
};
// int -> float without v_cvt_f32_i32: for -2^22 <= acc < 2^22 the bit pattern 0x4B400000 + acc IS the float 1.5 * 2^23 + acc
// (exponent of 2^23, mantissa 0x400000 + acc), and subtracting 1.5 * 2^23 is exact.  The MFMA accumulates on top of the initial
// value, so a conv that starts at bias + RQ_KBIAS ends with that pattern for free; two v_pk_add_f32 then replace four
// v_cvt_f32_i32 per dword of outputs (15 -> 13 vector instructions per four requantised elements, none of them VOP1).
constexpr int RQ_KBIAS = 0x4B400000;
__host__ __device__ inline Rq make_rq(int zp, int lo, int hi, int kb = 0) {
  return Rq{(float)(lo - zp), (float)(hi - zp), (float)(zp + 128), (lo <= -128 && hi >= 127) ? 1 : 0, kb};
}
__device__ __forceinline__ float rq_u8(float accf, float mult, const Rq& q) {
  float r = __builtin_rintf(accf * mult);
  return __builtin_amdgcn_fmed3f(r, q.lo_f, q.hi_f) + q.off;
}
__device__ __forceinline__ unsigned pack4_u8f(float a, float b, float c, float d) {  // -> int8 x4
  unsigned v = __builtin_amdgcn_cvt_pk_u8_f32(a, 0, 0);
  v = __builtin_amdgcn_cvt_pk_u8_f32(b, 1, v);
  v = __builtin_amdgcn_cvt_pk_u8_f32(c, 2, v);
  v = __builtin_amdgcn_cvt_pk_u8_f32(d, 3, v);
  return v ^ 0x80808080u;
}
// acc already contains the bias (accumulators are initialised with it).  The multiply and the offset add use the
// packed fp32 VALU forms (v_pk_mul_f32 / v_pk_add_f32: two IEEE single ops per instruction, same results).
typedef float v2f __attribute__((ext_vector_type(2)));
// FULLK: -1 = the clamp flavour is read from q.full at run time; otherwise bit 0 = saturating flavour (q.full), bit 1 = the
// accumulators were started at bias + RQ_KBIAS (q.kb; the caller adds it where it loads the bias)
template <int FULLK = -1>
__device__ __forceinline__ unsigned rq_pack_b(const v4i& acc, const float4& mu, const Rq& q) {
  v2f f0, f1;
  if (FULLK >= 2) {
    const v2f kf = {12582912.0f, 12582912.0f};
    f0 = (v2f){__int_as_float(acc[0]), __int_as_float(acc[1])} - kf;
    f1 = (v2f){__int_as_float(acc[2]), __int_as_float(acc[3])} - kf;
  } else {
    f0 = (v2f){(float)acc[0], (float)acc[1]};
    f1 = (v2f){(float)acc[2], (float)acc[3]};
  }
  v2f t0 = f0 * (v2f){mu.x, mu.y};
  v2f t1 = f1 * (v2f){mu.z, mu.w};
  const v2f off = {q.off, q.off};
  if ((FULLK >= 0 && (FULLK & 1)) || (FULLK < 0 && q.full)) {
    // rne(t) by the float adder: t + 1.5*2^23 has ulp 1 and an even base, so the sum is exactly 1.5*2^23 + rne(t) for
    // |t| < 2^22; adding (zp + 128 - 1.5*2^23) is exact again and the u8 conversion saturates to [0, 255], which IS the
    // clamp to int8 (+128).  |t| >= 2^22 stays far outside [0, 255] on the same side, i.e. saturates like the clamp.
    // Two packed adds replace four v_rndne + one packed add.
    const v2f magic = {12582912.0f, 12582912.0f};
    const v2f back = {q.off - 12582912.0f, q.off - 12582912.0f};
    v2f r0 = (t0 + magic) + back;
    v2f r1 = (t1 + magic) + back;
    return pack4_u8f(r0.x, r0.y, r1.x, r1.y);
  }
  v2f r0 = {__builtin_amdgcn_fmed3f(__builtin_rintf(t0.x), q.lo_f, q.hi_f), __builtin_amdgcn_fmed3f(__builtin_rintf(t0.y), q.lo_f, q.hi_f)};
  v2f r1 = {__builtin_amdgcn_fmed3f(__builtin_rintf(t1.x), q.lo_f, q.hi_f), __builtin_amdgcn_fmed3f(__builtin_rintf(t1.y), q.lo_f, q.hi_f)};
  r0 = r0 + off;
  r1 = r1 + off;
  return pack4_u8f(r0.x, r0.y, r1.x, r1.y);
}
// A operand of the depthwise-as-matrix-product MFMAs (out[c][p] = sum W'[c][(t,c')] X[(t,c')][p], W' = w[t][c] delta(c,c')): the
// 16 bytes of lane (i = lane & 15, g) hold ONE non-zero byte, byte i = the weight of the lane's tap for channel i.  Built in
// registers from that byte (6 VALU) instead of loaded as 16 bytes per lane: the diagonal operands were ~1 KB per wave and
// instruction out of L1, most of the weight stream of the fused MBConv kernels (L1 accesses: 18 TB/s in fused_mbconv).
__device__ __forceinline__ v4i diag_operand(unsigned wb, int i) {
  const unsigned sh = wb << (8 * (i & 3));
  const int d = i >> 2;
  return (v4i){d == 0 ? (int)sh : 0, d == 1 ? (int)sh : 0, d == 2 ? (int)sh : 0, d == 3 ? (int)sh : 0};
}
// exact n / d for 0 <= n < 2^20, 1 <= d <= 4096 without the ~40-instruction integer division
__device__ __forceinline__ int fdiv_small(int n, float rcp_d) { return (int)(((float)n + 0.5f) * rcp_d); }
// reciprocal for fdiv_small: one v_rcp_f32 (1 ulp) instead of the IEEE division sequence.  (n + 0.5) / d lies at least
// 0.5 / d away from an integer and the product carries < 2^-22 relative error, so the floor is exact for n < 2^20.
__device__ __forceinline__ float frcp(int d) { return __builtin_amdgcn_rcpf((float)d); }
__device__ __forceinline__ v4i v4i_from(const int4& b) { return (v4i){b.x, b.y, b.z, b.w}; }
__device__ __forceinline__ int4 int4_plus(const int4& b, int k) { return make_int4(b.x + k, b.y + k, b.z + k, b.w + k); }
// run f(std::integral_constant<int, mode>) with mode = q.full | q.kb << 1 (uniform: one of four instantiations of a stage)
template <class F>
__device__ __forceinline__ void rq_dispatch(const Rq& q, F&& f) {
  if (q.kb) { if (q.full) f(std::integral_constant<int, 3>{}); else f(std::integral_constant<int, 2>{}); }
  else { if (q.full) f(std::integral_constant<int, 1>{}); else f(std::integral_constant<int, 0>{}); }
}

__device__ __forceinline__ unsigned rq_pack_i(const v4i& acc, const int4& b, const float4& mu, const Rq& q) {
  const v4i a2 = {acc[0] + b.x, acc[1] + b.y, acc[2] + b.z, acc[3] + b.w};
  return rq_pack_b(a2, mu, q);
}

// ---- int8 ADD, XNNPACK qs8-vadd-minmax: q = clamp(((bias + a*am + b*bm) >> shift) + z_out, lo, hi).  The kernel's
// int16 / int8 saturating packs are monotone and the activation range lies inside int8, so the chain of saturations equals
// one clamp; it is applied before the zero point is added (lo - z_out, hi - z_out), and `off` = z_out + 128 moves the
// result to its u8 image so that four of them pack with shifts (no masks) and one XOR restores int8.
struct AddQ { int bias, am, bm, shift, lo, hi, off; };
static inline AddQ make_addq(const AddParams& p, int z_out, int lo, int hi) {
  return AddQ{p.bias, p.am, p.bm, p.shift, lo - z_out, hi - z_out, z_out + 128};
}
__device__ __forceinline__ int addq_u8(int a, int b, const AddQ& q) {  // -> q + 128 in [0, 255]
  const int t = (q.bias + __mul24(a, q.am) + __mul24(b, q.bm)) >> q.shift;   // |am|, |bm| < 2^22, a, b int8: 24-bit products
  return min(max(t, q.lo), q.hi) + q.off;
}
__device__ __forceinline__ unsigned addq4(unsigned ua, unsigned ub, const AddQ& q) {  // four int8 lanes per dword
  unsigned r = 0;
#pragma unroll
  for (int e = 0; e < 4; e++)
    r |= (unsigned)addq_u8((int)(int8_t)(ua >> (8 * e)), (int)(int8_t)(ub >> (8 * e)), q) << (8 * e);
  return r ^ 0x80808080u;
}

struct Epi {  // requantisation parameters of one conv
  const int* bias;    // folded bias, padded to NB*64
  const float* mult;  // padded to NB*64
  int zp, lo, hi;
  Rq rq;
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
    d[t] = rq_pack_i(acc[t], b, mu, e.rq);
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

}  // namespace vbt
