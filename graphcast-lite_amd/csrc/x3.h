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
