// Probe: discover the A/B/C lane maps of the gfx950 int8 MFMA shapes with exact integer data.
// A[i][k] = 1 only at one (i,k); B[k][j] = j+1 (asymmetric)  -> C[i][j] = j+1 for that i.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// generic: each lane gets raw A bytes / B bytes from memory laid out [lane][nbytes]
template <int NB>
__global__ void k16x16(const int8_t* a, const int8_t* b, int* c) {
  int l = threadIdx.x;
  v4i acc = {0, 0, 0, 0};
  if constexpr (NB == 8) {
    long av = *(const long*)(a + l * 8), bv = *(const long*)(b + l * 8);
    acc = __builtin_amdgcn_mfma_i32_16x16x32_i8(av, bv, acc, 0, 0, 0);
  } else {
    v4i av = *(const v4i*)(a + l * 16), bv = *(const v4i*)(b + l * 16);
    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, bv, acc, 0, 0, 0);
  }
  for (int r = 0; r < 4; r++) c[l * 4 + r] = acc[r];
}
template <int NB>
__global__ void k32x32(const int8_t* a, const int8_t* b, int* c) {
  int l = threadIdx.x;
  v16i acc;
  for (int r = 0; r < 16; r++) acc[r] = 0;
  if constexpr (NB == 8) {
    long av = *(const long*)(a + l * 8), bv = *(const long*)(b + l * 8);
    acc = __builtin_amdgcn_mfma_i32_32x32x16_i8(av, bv, acc, 0, 0, 0);
  } else {
    v4i av = *(const v4i*)(a + l * 16), bv = *(const v4i*)(b + l * 16);
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bv, acc, 0, 0, 0);
  }
  for (int r = 0; r < 16; r++) c[l * 16 + r] = acc[r];
}

// Hypothesis maps. 16x16: lane l row/col = l&15, k = NB*(l>>4)+j.  32x32: row/col = l&31, k = NB*(l>>5)+j
int main() {
  for (int shape = 0; shape < 4; shape++) {
    int MN = shape < 2 ? 16 : 32;
    int NB = (shape & 1) ? 16 : 8;
    int K = shape < 2 ? NB * 4 : NB * 2;
    int nacc = shape < 2 ? 4 : 16;
    std::vector<int8_t> A(MN * K), B(K * MN);
    for (int i = 0; i < MN; i++) for (int k = 0; k < K; k++) A[i * K + k] = (int8_t)((i * 7 + k * 3) % 11 - 5);
    for (int k = 0; k < K; k++) for (int j = 0; j < MN; j++) B[k * MN + j] = (int8_t)((k * 5 + j * 13) % 17 - 8);
    std::vector<int> Cref(MN * MN, 0);
    for (int i = 0; i < MN; i++) for (int j = 0; j < MN; j++) { int s = 0; for (int k = 0; k < K; k++) s += A[i * K + k] * B[k * MN + j]; Cref[i * MN + j] = s; }
    std::vector<int8_t> la(64 * NB), lb(64 * NB);
    for (int l = 0; l < 64; l++) for (int j = 0; j < NB; j++) {
      int rc = MN == 16 ? (l & 15) : (l & 31);
      int k = MN == 16 ? NB * (l >> 4) + j : NB * (l >> 5) + j;
      la[l * NB + j] = A[rc * K + k];
      lb[l * NB + j] = B[k * MN + rc];
    }
    int8_t *da, *db; int* dc;
    hipMalloc(&da, 64 * NB); hipMalloc(&db, 64 * NB); hipMalloc(&dc, 64 * nacc * 4);
    hipMemcpy(da, la.data(), 64 * NB, hipMemcpyHostToDevice);
    hipMemcpy(db, lb.data(), 64 * NB, hipMemcpyHostToDevice);
    if (shape == 0) k16x16<8><<<1, 64>>>(da, db, dc);
    if (shape == 1) k16x16<16><<<1, 64>>>(da, db, dc);
    if (shape == 2) k32x32<8><<<1, 64>>>(da, db, dc);
    if (shape == 3) k32x32<16><<<1, 64>>>(da, db, dc);
    std::vector<int> C(64 * nacc);
    hipMemcpy(C.data(), dc, 64 * nacc * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++) for (int r = 0; r < nacc; r++) {
      int row, col;
      if (MN == 16) { col = l & 15; row = (l >> 4) * 4 + r; }
      else { col = l & 31; row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5); }
      if (C[l * nacc + r] != Cref[row * MN + col]) bad++;
    }
    printf("shape %dx%dx%d (NB=%d): hypothesis %s (bad=%d)\n", MN, MN, K, NB, bad ? "WRONG" : "OK", bad);
  }
  return 0;
}
