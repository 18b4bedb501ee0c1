// fp32 dense transforms on the bf16 matrix pipe: every fp32 operand is split EXACTLY into three bf16
// pieces (x = hi + mid + lo, round-to-nearest at each step: 8 + 8 + 8 significand bits) and a product
// x*y is formed from the six piece products whose weight is >= 2^-16 of it (hi*hi, hi*mid, mid*hi,
// hi*lo, lo*hi, mid*mid), accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  Every piece product is exact
// in fp32 (8 x 8 bits), so the only errors are the fp32 accumulation (as in any fp32 GEMM) and the three
// dropped piece products (<= 2^-25 relative): the result is as close to the exact product as the
// fp32-operand instruction's (measured: closer, tests/test_hip_ops.py::test_x3_*).  Cost: 6 MFMAs of
// 32 cycles per 32x32x16 block against 8 of 64 cycles with v_mfma_f32_32x32x2_f32 - 0.375x the matrix
// time - and, unlike the fp32 instruction, the bf16 one leaves the SIMD's issue port free for the
// wave's own vector instructions while it runs (tools/probes/mfma_valu_probe.hip).
#pragma once
#include "common.h"

namespace gcl {

// y = act(x) W^T + b for K <= 64, N <= 64 (N % 4 == 0), 16-B aligned rows.  `applicable` is the host-side
// shape test; the launch returns GCL_OK / GCL_EHIP.
bool x3_linear_fwd_applicable(const float* x, int64_t ldx, const float* y, int64_t ldy, int K, int N, int akind);
int x3_linear_fwd(const float* x, int64_t ldx, int akind, const float* slope, const float* W, const float* bias,
                  float* y, int64_t ldy, int64_t rows, int K, int N, hipStream_t st);

// Fused backward (dX, dW, db, colsum(dX), d_slope partials) for Fin in 33..64 (Fin % 4 == 0), Fout <= 64: same
// per-block partial records as linear_bwd_fused64_kernel (linear.hip), x3_linear_bwd_blocks(rows) of them.
bool x3_linear_bwd_applicable(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* dx, int64_t lddx,
                              int Fin, int Fout);
int x3_linear_bwd_blocks(int64_t rows);
int x3_linear_bwd(const float* dy, int64_t lddy, const float* W, const float* x, int64_t ldx, const float* in_slope,
                  float* dx, int64_t lddx, int64_t rows, int Fin, int Fout, float* part_dw, float* part_db,
                  float* part_cs, double* part_slope, hipStream_t st);

}  // namespace gcl

#ifdef __HIPCC__
// ---- device side: the split and the MFMA passes shared by linear_x3.hip, gcn_layer.hip and gemm_tile.h ----
namespace gcl {
namespace x3 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

// two fp32 -> two packed bf16, round to nearest even (low half = a): v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}
struct Pk3 {
  unsigned h, m, l;  // packed (a, b) pieces: a = hi + mid + lo exactly, likewise b
};
__device__ __forceinline__ Pk3 split2(float a, float b) {
  Pk3 p;
  p.h = pk_bf16(a, b);
  float ra = a - __uint_as_float(p.h << 16), rb = b - __uint_as_float(p.h & 0xffff0000u);  // exact: <= 16 bits left
  p.m = pk_bf16(ra, rb);
  ra -= __uint_as_float(p.m << 16);  // exact: <= 8 bits left, so the last rounding is exact too
  rb -= __uint_as_float(p.m & 0xffff0000u);
  p.l = pk_bf16(ra, rb);
  return p;
}

// acc += (ah + am + al) x (bh + bm + bl) without the three smallest piece products, as three groups that callers
// run smallest first over all their k-steps (or keep in separate accumulators): the 2^-16-level products, the
// 2^-8-level ones, hi x hi.  Adding corrections to a full-size accumulator costs ~10 ulp: the matrix pipe does not
// round each of its internal additions to nearest.
__device__ __forceinline__ f32x16 mfma_lo(f32x16 acc, bf16x8 ah, bf16x8 am, bf16x8 al, bf16x8 bh, bf16x8 bm, bf16x8 bl) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
  return acc;
}
__device__ __forceinline__ f32x16 mfma_mid(f32x16 acc, bf16x8 ah, bf16x8 am, bf16x8 bh, bf16x8 bm) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
  return acc;
}
__device__ __forceinline__ f32x16 mfma_hi(f32x16 acc, bf16x8 ah, bf16x8 bh) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
}

}  // namespace x3
}  // namespace gcl
#endif
