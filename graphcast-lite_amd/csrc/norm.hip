// PyG LayerNorm (mode="node" / mode="graph") forward + backward, and column sums.
// Reference: src/models.py:102-104,368-374 (construction), :108,:421 (application); arithmetic per
// SURVEY.md Appendix A.4.  Row-wise kernels use the same row->lane-group mapping as the
// aggregation kernel (LPR lanes x 4 channels, 16-B accesses, shuffle reductions inside the group).
#include "common.h"

namespace {

template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int off = LPR / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ void load4(const float* p, int c0, int F, bool vec, float& a, float& b, float& c, float& d) {
  if (vec) {
    // the rows of these kernels are read exactly once: non-temporal loads keep them from displacing what the NEXT
    // kernel gathers out of the L2 (the aggregation that follows the LayerNorm backward reads 2 % faster; end to end
    // unchanged).  -DGCL_NORM_LD_PLAIN: plain loads.
#ifndef GCL_NORM_LD_PLAIN
    typedef float lv4f __attribute__((ext_vector_type(4)));
    const lv4f v = __builtin_nontemporal_load(reinterpret_cast<const lv4f*>(p));
#else
    const float4 v = *reinterpret_cast<const float4*>(p);
#endif
    a = v.x; b = v.y; c = v.z; d = v.w;
  } else {
    a = (c0 < F) ? p[0] : 0.f;
    b = (c0 + 1 < F) ? p[1] : 0.f;
    c = (c0 + 2 < F) ? p[2] : 0.f;
    d = (c0 + 3 < F) ? p[3] : 0.f;
  }
}
__device__ __forceinline__ void store4(float* p, int c0, int F, bool vec, float a, float b, float c, float d) {
  if (vec) {
    *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
  } else {
    if (c0 < F) p[0] = a;
    if (c0 + 1 < F) p[1] = b;
    if (c0 + 2 < F) p[2] = c;
    if (c0 + 3 < F) p[3] = d;
  }
}

// y = (x - mean) * rstd * gamma + beta ; stats[row] = (mean, rstd)
// V: every row access is a clean 16-byte one (compile-time, so the compiler emits dwordx4 instead of
// merging the vector and the scalar path into dwordx3 + dword accesses)
// MAP: the output goes through a row map - row (b, i) of the [B][n_per] row space is written to
// Y[b * bsy + pos[i] * ldy] when pos[i] >= 0 and not at all otherwise (a LayerNorm whose output is only consumed through a
// row gather writes the gathered rows straight into the consumer's input: src/models.py:860-862 for the processor).
template <int LPR, bool V, bool MAP = false>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ X, int64_t ldx,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float eps, float* __restrict__ Y, int64_t ldy,
                                                     float* __restrict__ stats, int64_t rows, int32_t F, int32_t vx,
                                                     int32_t vy, const int32_t* __restrict__ pos = nullptr,
                                                     int64_t bsy = 0, int32_t n_per = 1) {
  if (V) vx = vy = 1;
  constexpr int RPB = (64 / LPR) * 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / LPR, l = lane % LPR, c0 = l * 4;
  const float invF = 1.f / (float)F;
  float g0 = 0, g1 = 0, g2 = 0, g3 = 0, b0 = 0, b1 = 0, b2 = 0, b3 = 0;
  if (c0 < F) {
    load4(gamma + c0, c0, F, false, g0, g1, g2, g3);
    load4(beta + c0, c0, F, false, b0, b1, b2, b3);
  }
  for (int64_t row = (int64_t)blockIdx.x * RPB + wave * (64 / LPR) + sub; row < rows; row += (int64_t)gridDim.x * RPB) {
    float x0 = 0, x1 = 0, x2 = 0, x3 = 0;
    if (c0 < F) load4(X + row * ldx + c0, c0, F, vx, x0, x1, x2, x3);
    const float mean = group_sum<LPR>(x0 + x1 + x2 + x3) * invF;
    const float d0 = (c0 < F) ? x0 - mean : 0.f, d1 = (c0 + 1 < F) ? x1 - mean : 0.f;
    const float d2 = (c0 + 2 < F) ? x2 - mean : 0.f, d3 = (c0 + 3 < F) ? x3 - mean : 0.f;
    float sq;
    {
      // products and sums rounded separately: left to the compiler, the plain and the mapped instantiation contract
      // this expression differently (fused multiply-adds in one, packed multiplies + adds in the other) and their
      // statistics differ in the last bit
#pragma clang fp contract(off)
      const float q0 = d0 * d0, q1 = d1 * d1, q2 = d2 * d2, q3 = d3 * d3;
      sq = (q0 + q1) + (q2 + q3);
    }
    const float var = group_sum<LPR>(sq) * invF;
    const float rstd = 1.0f / sqrtf(var + eps);
    if (MAP) {
      const int64_t bq = row / n_per;
      const int pj = pos[(int)(row - bq * n_per)];
      if (c0 < F && pj >= 0)
        store4(Y + bq * bsy + (int64_t)pj * ldy + c0, c0, F, vy, d0 * rstd * g0 + b0, d1 * rstd * g1 + b1,
               d2 * rstd * g2 + b2, d3 * rstd * g3 + b3);
    } else if (c0 < F)
      store4(Y + row * ldy + c0, c0, F, vy, d0 * rstd * g0 + b0, d1 * rstd * g1 + b1, d2 * rstd * g2 + b2,
             d3 * rstd * g3 + b3);
    if (l == 0 && stats) {
      stats[2 * row] = mean;
      stats[2 * row + 1] = rstd;
    }
  }
}

// dx = rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dy*gamma;  partial dgamma/dbeta per block
// CS: also the column sums of dX (the bias gradient of the layer below: dX is that layer's dY) as a third vector
// MAP: dY is given through a row map instead of densely - row (b, i) of the [B][n] row space reads
// dY[b * bsdy + pos[i] * lddy] when pos[i] >= 0 and is zero otherwise (the gradient of a layer whose output was only
// consumed through a row gather: no zero-filled dense gradient has to exist).
template <int LPR, bool V, bool CS, bool MAP>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dY, int64_t lddy,
                                                     const float* __restrict__ X, int64_t ldx,
                                                     const float* __restrict__ gamma, const float* __restrict__ stats,
                                                     float* __restrict__ dX, int64_t lddx, float* __restrict__ part,
                                                     int64_t rows, int32_t F, int32_t FP, int32_t vdy, int32_t vx,
                                                     int32_t vdx, const int32_t* __restrict__ pos, int64_t bsdy,
                                                     int32_t n_per) {
  if (V) vdy = vx = vdx = 1;
  constexpr int RPW = 64 / LPR;
  constexpr int RPB = RPW * 4;
  constexpr int NV = CS ? 3 : 2;
  __shared__ float red[RPB][LPR * 4 * NV + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / LPR, l = lane % LPR, c0 = l * 4;
  const float invF = 1.f / (float)F;
  float g0 = 0, g1 = 0, g2 = 0, g3 = 0;
  if (c0 < F) load4(gamma + c0, c0, F, false, g0, g1, g2, g3);
  float dg0 = 0, dg1 = 0, dg2 = 0, dg3 = 0, db0 = 0, db1 = 0, db2 = 0, db3 = 0;
  float cs0 = 0, cs1 = 0, cs2 = 0, cs3 = 0;
  // MAP: (sample, row in sample) of this lane group's row, advanced with the grid stride (no division per row)
  const int64_t row_first = (int64_t)blockIdx.x * RPB + wave * RPW + sub, row_step = (int64_t)gridDim.x * RPB;
  int64_t mb = 0, msb = 0;
  int mi = 0, msi = 0;
  if (MAP) {
    mb = row_first / n_per;
    mi = (int)(row_first - mb * n_per);
    msb = row_step / n_per;
    msi = (int)(row_step - msb * n_per);
  }
  for (int64_t row = row_first; row < rows; row += row_step) {
    float x0 = 0, x1 = 0, x2 = 0, x3 = 0, y0 = 0, y1 = 0, y2 = 0, y3 = 0;
    if (c0 < F) {
      load4(X + row * ldx + c0, c0, F, vx, x0, x1, x2, x3);
      if (MAP) {
        // unmapped rows issue no load at all: this kernel is bound by its load INSTRUCTIONS (a wave-instruction serves
        // only four rows), so a dummy load for them costs more than the divergent branch around the real one
        const int pj = pos[mi];
        if (pj >= 0) load4(dY + mb * bsdy + (int64_t)pj * lddy + c0, c0, F, vdy, y0, y1, y2, y3);
      } else {
        load4(dY + row * lddy + c0, c0, F, vdy, y0, y1, y2, y3);
      }
    }
    if (MAP) {
      mi += msi;
      mb += msb;
      if (mi >= n_per) { mi -= n_per; ++mb; }
    }
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    const float h0 = (c0 < F) ? (x0 - mean) * rstd : 0.f, h1 = (c0 + 1 < F) ? (x1 - mean) * rstd : 0.f;
    const float h2 = (c0 + 2 < F) ? (x2 - mean) * rstd : 0.f, h3 = (c0 + 3 < F) ? (x3 - mean) * rstd : 0.f;
    const float q0 = y0 * g0, q1 = y1 * g1, q2 = y2 * g2, q3 = y3 * g3;
    const float m1 = group_sum<LPR>(q0 + q1 + q2 + q3) * invF;
    const float m2 = group_sum<LPR>(q0 * h0 + q1 * h1 + q2 * h2 + q3 * h3) * invF;
    const float o0 = rstd * (q0 - m1 - h0 * m2), o1 = rstd * (q1 - m1 - h1 * m2), o2 = rstd * (q2 - m1 - h2 * m2),
                o3 = rstd * (q3 - m1 - h3 * m2);
    if (c0 < F) store4(dX + row * lddx + c0, c0, F, vdx, o0, o1, o2, o3);
    if (CS) {
      cs0 += (c0 < F) ? o0 : 0.f; cs1 += (c0 + 1 < F) ? o1 : 0.f; cs2 += (c0 + 2 < F) ? o2 : 0.f; cs3 += (c0 + 3 < F) ? o3 : 0.f;
    }
    dg0 += y0 * h0; dg1 += y1 * h1; dg2 += y2 * h2; dg3 += y3 * h3;
    db0 += y0; db1 += y1; db2 += y2; db3 += y3;
  }
  // reduce the RPB row groups of this block -> part[block][2*FP]  (dgamma | dbeta)
  const int g = wave * RPW + sub;
  float* r = red[g];
  r[c0] = dg0; r[c0 + 1] = dg1; r[c0 + 2] = dg2; r[c0 + 3] = dg3;
  r[LPR * 4 + c0] = db0; r[LPR * 4 + c0 + 1] = db1; r[LPR * 4 + c0 + 2] = db2; r[LPR * 4 + c0 + 3] = db3;
  if (CS) {
    r[2 * LPR * 4 + c0] = cs0; r[2 * LPR * 4 + c0 + 1] = cs1; r[2 * LPR * 4 + c0 + 2] = cs2; r[2 * LPR * 4 + c0 + 3] = cs3;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < NV * LPR * 4; idx += 256) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < RPB; ++q) s += red[q][idx];
    const int which = idx / (LPR * 4), c = idx % (LPR * 4);
    if (c < F) part[(size_t)blockIdx.x * NV * FP + which * FP + c] = s;  // record: dgamma | dbeta (| colsum dX)
  }
}

// column sums: part[block][FP]
template <int LPR, bool V>
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, int64_t ldx,
                                                     float* __restrict__ part, int64_t rows, int32_t F, int32_t FP,
                                                     int32_t vx) {
  if (V) vx = 1;
  constexpr int RPW = 64 / LPR;
  constexpr int RPB = RPW * 4;
  __shared__ float red[RPB][LPR * 4 + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / LPR, l = lane % LPR, c0 = l * 4;
  float s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  if (c0 < F) {
    // four rows in flight per thread (one load per trip left a wave with a single 16-byte load outstanding: 2.7 TB/s at
    // 128 columns); the sums are taken in the same row order as before
    const int64_t step = (int64_t)gridDim.x * RPB;
    int64_t row = (int64_t)blockIdx.x * RPB + wave * RPW + sub;
    for (; row + 3 * step < rows; row += 4 * step) {
      float a0, a1, a2, a3, b0, b1, b2, b3, c1_, c2_, c3_, c4_, d0, d1, d2, d3;
      load4(X + row * ldx + c0, c0, F, vx, a0, a1, a2, a3);
      load4(X + (row + step) * ldx + c0, c0, F, vx, b0, b1, b2, b3);
      load4(X + (row + 2 * step) * ldx + c0, c0, F, vx, c1_, c2_, c3_, c4_);
      load4(X + (row + 3 * step) * ldx + c0, c0, F, vx, d0, d1, d2, d3);
      s0 += a0; s1 += a1; s2 += a2; s3 += a3;
      s0 += b0; s1 += b1; s2 += b2; s3 += b3;
      s0 += c1_; s1 += c2_; s2 += c3_; s3 += c4_;
      s0 += d0; s1 += d1; s2 += d2; s3 += d3;
    }
    for (; row < rows; row += step) {
      float x0, x1, x2, x3;
      load4(X + row * ldx + c0, c0, F, vx, x0, x1, x2, x3);
      s0 += x0; s1 += x1; s2 += x2; s3 += x3;
    }
  }
  float* r = red[wave * RPW + sub];
  r[c0] = s0; r[c0 + 1] = s1; r[c0 + 2] = s2; r[c0 + 3] = s3;
  __syncthreads();
  for (int c = threadIdx.x; c < F; c += 256) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < RPB; ++q) s += red[q][c];
    part[(size_t)blockIdx.x * FP + c] = s;
  }
}

constexpr int kNormBlocks = 1024;  // 4 blocks per CU (measured: 512 -> 1024 +1.8 % end to end, 2048 the same, 4096 less)

inline int lpr_for(int F) {
  const int lanes = (F + 3) / 4;
  return lanes <= 4 ? 4 : lanes <= 8 ? 8 : lanes <= 16 ? 16 : lanes <= 32 ? 32 : 64;
}
inline bool vec_ok(const float* p, int64_t ld, int F) { return (ld % 4 == 0) && gcl::aligned16(p) && ld >= ((F + 3) / 4) * 4; }
inline bool vec_store_ok(const float* p, int64_t ld, int F) { return (ld % 4 == 0) && gcl::aligned16(p) && (F % 4 == 0); }

}  // namespace

#define GCL_DISPATCH_LPR(lpr, CALL) \
  switch (lpr) {                    \
    case 4: CALL(4); break;         \
    case 8: CALL(8); break;         \
    case 16: CALL(16); break;       \
    case 32: CALL(32); break;       \
    default: CALL(64); break;       \
  }

extern "C" int gcl_layernorm_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps,
                                 float* y, int64_t ldy, float* stats, int64_t rows, int32_t F, gcl_stream_t stream) {
  GCL_CHECK_ARG(x && gamma && beta && y, "layernorm_fwd: null argument");
  GCL_CHECK_ARG(F >= 1 && F <= 256 && ldx >= F && ldy >= F, "layernorm_fwd: bad shape F=%d", F);
  if (rows == 0) return GCL_OK;
  const int lpr = lpr_for(F);
  const int rpb = (64 / lpr) * 4;
  int64_t nb = gcl::cdiv(rows, rpb);
  if (nb > 8192) nb = 8192;
  const int vx = vec_ok(x, ldx, F), vy = vec_store_ok(y, ldy, F);
#define CALL(L)                                                                                                  \
  if (vx && vy)                                                                                                  \
    hipLaunchKernelGGL((ln_fwd_kernel<L, true>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, ldx,  \
                       gamma, beta, eps, y, ldy, stats, rows, F, vx, vy, (const int32_t*)nullptr, (int64_t)0, 1); \
  else                                                                                                           \
    hipLaunchKernelGGL((ln_fwd_kernel<L, false>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, ldx, \
                       gamma, beta, eps, y, ldy, stats, rows, F, vx, vy, (const int32_t*)nullptr, (int64_t)0, 1)
  GCL_DISPATCH_LPR(lpr, CALL)
#undef CALL
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

extern "C" int gcl_layernorm_fwd_map(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps,
                                     float* y, int64_t ldy, int64_t bsy, const int32_t* pos, int32_t n_per, float* stats,
                                     int64_t rows, int32_t F, gcl_stream_t stream) {
  GCL_CHECK_ARG(x && gamma && beta && y && pos && stats, "layernorm_fwd_map: null argument");
  GCL_CHECK_ARG(F >= 1 && F <= 256 && ldx >= F && ldy >= F, "layernorm_fwd_map: bad shape F=%d", F);
  GCL_CHECK_ARG(n_per > 0 && rows % n_per == 0, "layernorm_fwd_map: rows must be B * n_per");
  if (rows == 0) return GCL_OK;
  const int lpr = lpr_for(F);
  const int rpb = (64 / lpr) * 4;
  int64_t nb = gcl::cdiv(rows, rpb);
  if (nb > 8192) nb = 8192;
  const int vx = vec_ok(x, ldx, F), vy = vec_store_ok(y, ldy, F) && (bsy % 4 == 0);
#define CALL(L)                                                                                                        \
  if (vx && vy)                                                                                                        \
    hipLaunchKernelGGL((ln_fwd_kernel<L, true, true>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, ldx,  \
                       gamma, beta, eps, y, ldy, stats, rows, F, vx, vy, pos, bsy, n_per);                             \
  else                                                                                                                 \
    hipLaunchKernelGGL((ln_fwd_kernel<L, false, true>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, ldx, \
                       gamma, beta, eps, y, ldy, stats, rows, F, vx, vy, pos, bsy, n_per)
  GCL_DISPATCH_LPR(lpr, CALL)
#undef CALL
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

extern "C" size_t gcl_layernorm_bwd_ws_bytes(int64_t rows, int32_t F) {
  (void)rows;
  const size_t FP = (size_t)((F + 3) / 4) * 4;
  return (size_t)kNormBlocks * 3 * FP * sizeof(float);
}

extern "C" int gcl_layernorm_bwd_cs(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma,
                                    const float* stats, float* dx, int64_t lddx, float* dgamma, float* dbeta,
                                    float* colsum_dx, int32_t accumulate, int64_t rows, int32_t F, void* ws,
                                    size_t ws_bytes, gcl_stream_t stream) {
  return gcl_layernorm_bwd_map(dy, lddy, 0, nullptr, 0, x, ldx, gamma, stats, dx, lddx, dgamma, dbeta, colsum_dx, accumulate,
                               rows, F, ws, ws_bytes, stream);
}

extern "C" int gcl_layernorm_bwd_map(const float* dy, int64_t lddy, int64_t bsdy, const int32_t* pos, int32_t n_per,
                                     const float* x, int64_t ldx, const float* gamma, const float* stats, float* dx,
                                     int64_t lddx, float* dgamma, float* dbeta, float* colsum_dx, int32_t accumulate,
                                     int64_t rows, int32_t F, void* ws, size_t ws_bytes, gcl_stream_t stream) {
  GCL_CHECK_ARG(dy && x && gamma && stats && dx && dgamma && dbeta, "layernorm_bwd: null argument");
  GCL_CHECK_ARG(!pos || (n_per > 0 && rows % n_per == 0), "layernorm_bwd: mapped dy needs rows = B * n_per");
  GCL_CHECK_ARG(F >= 1 && F <= 256 && ldx >= F && lddy >= F && lddx >= F, "layernorm_bwd: bad shape F=%d", F);
  GCL_CHECK_ARG(ws && ws_bytes >= gcl_layernorm_bwd_ws_bytes(rows, F), "layernorm_bwd: workspace too small");
  if (rows == 0) return GCL_OK;
  hipStream_t st = (hipStream_t)stream;
  const int lpr = lpr_for(F);
  const int rpb = (64 / lpr) * 4;
  const int FP = ((F + 3) / 4) * 4;
  int64_t nb = gcl::cdiv(rows, rpb);
  if (nb > kNormBlocks) nb = kNormBlocks;
  float* part = (float*)ws;
  const int vdy = vec_ok(dy, lddy, F) && (bsdy % 4 == 0), vx = vec_ok(x, ldx, F), vdx = vec_store_ok(dx, lddx, F);
  // dgamma / dbeta share GCL_ACC_DW, the column sums have their own bit (they belong to another parameter)
  const int acc_p = (accumulate & GCL_ACC_DW) ? 1 : 0, acc_cs = (accumulate & GCL_ACC_COLSUM) ? 1 : 0;
#define CALL4(L, V_, CS_, MAP_)                                                                                       \
  hipLaunchKernelGGL((ln_bwd_kernel<L, V_, CS_, MAP_>), dim3((unsigned)nb), dim3(256), 0, st, dy, lddy, x, ldx, gamma, \
                     stats, dx, lddx, part, rows, F, FP, vdy, vx, vdx, pos, bsdy, n_per)
#define CALL3(L, V_, CS_)            \
  do {                               \
    if (pos) CALL4(L, V_, CS_, true); \
    else CALL4(L, V_, CS_, false);   \
  } while (0)
#define CALL(L)                                     \
  if (vdy && vx && vdx) {                           \
    if (colsum_dx) CALL3(L, true, true);            \
    else CALL3(L, true, false);                     \
  } else {                                          \
    if (colsum_dx) CALL3(L, false, true);           \
    else CALL3(L, false, false);                    \
  }
  GCL_DISPATCH_LPR(lpr, CALL)
#undef CALL
#undef CALL3
#undef CALL4
  GCL_CHECK_LAUNCH();
  if (!colsum_dx) return gcl::launch_reduce_parts2(part, (int)nb, 2 * FP, FP, FP, dgamma, dbeta, F, acc_p, st);
  return gcl::launch_reduce_parts3(part, (int)nb, 3 * FP, FP, dgamma, acc_p, dbeta, acc_p, colsum_dx, acc_cs, F, st);
}

extern "C" int gcl_layernorm_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma,
                                 const float* stats, float* dx, int64_t lddx, float* dgamma, float* dbeta,
                                 int32_t accumulate, int64_t rows, int32_t F, void* ws, size_t ws_bytes,
                                 gcl_stream_t stream) {
  return gcl_layernorm_bwd_cs(dy, lddy, x, ldx, gamma, stats, dx, lddx, dgamma, dbeta, nullptr,
                              accumulate ? GCL_ACC_DW : 0, rows, F, ws, ws_bytes, stream);
}

extern "C" size_t gcl_colsum_ws_bytes(int64_t rows, int32_t F) {
  (void)rows;
  return (size_t)kNormBlocks * (size_t)(((F + 3) / 4) * 4) * sizeof(float);
}

extern "C" int gcl_colsum(const float* x, int64_t ldx, int64_t rows, int32_t F, float* out, int32_t accumulate,
                          void* ws, size_t ws_bytes, gcl_stream_t stream) {
  GCL_CHECK_ARG(x && out, "colsum: null argument");
  GCL_CHECK_ARG(F >= 1 && F <= 256 && ldx >= F, "colsum: bad shape F=%d", F);
  GCL_CHECK_ARG(ws && ws_bytes >= gcl_colsum_ws_bytes(rows, F), "colsum: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int lpr = lpr_for(F);
  const int rpb = (64 / lpr) * 4;
  const int FP = ((F + 3) / 4) * 4;
  int64_t nb = gcl::cdiv(rows > 0 ? rows : 1, rpb);
  if (nb > kNormBlocks) nb = kNormBlocks;
  float* part = (float*)ws;
  const int vx = vec_ok(x, ldx, F);
#define CALL(L)                                                                                                     \
  if (vx) hipLaunchKernelGGL((colsum_kernel<L, true>), dim3((unsigned)nb), dim3(256), 0, st, x, ldx, part, rows, F, FP, vx); \
  else hipLaunchKernelGGL((colsum_kernel<L, false>), dim3((unsigned)nb), dim3(256), 0, st, x, ldx, part, rows, F, FP, vx)
  GCL_DISPATCH_LPR(lpr, CALL)
#undef CALL
  GCL_CHECK_LAUNCH();
  return gcl::launch_reduce_parts(part, (int)nb, FP, FP, out, F, 1, F, accumulate, st);
}
