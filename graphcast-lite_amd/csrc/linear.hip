// Dense per-node transforms  y = act(x) W^T + b  and their gradients.
//
// Replaces nn.Linear (+ preceding nn.PReLU) of MLP.forward (src/models.py:106-109) and the `lin`
// GEMM inside GCNConv / GATConv (src/models.py:419,425).  These are tall-skinny contractions
// (rows = B*n up to ~1.4M, K,N <= 128): at fp32 they sit at 16-32 flop/B, i.e. at or above the
// HBM ridge, so they run on the matrix cores with the EXACT-fp32 instruction
// v_mfma_f32_32x32x2_f32 (an fp32 FMA chain in k order; same peak as the fp32 VALU but one operand
// register per lane and no VALU pressure).  A plain VALU implementation of the same contract is
// kept (GCL_LINEAR_IMPL=valu) as an in-library cross-check for the MFMA operand layouts.
//
// Operand layouts of v_mfma_f32_32x32x2_f32 (cdna_hip_programming.md §3):
//   A: lane l holds A[i = l&31][k = l>>5]      B: lane l holds B[k = l>>5][j = l&31]
//   D: lane l, reg r holds D[i = (r&3) + 8*(r>>2) + 4*(l>>5)][j = l&31]
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int d_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

enum { EPI_BIAS = 0, EPI_DX = 1 };

// ---------------------------------------------------------------------------------------------
// Y[r, j] = sum_k act(X[r,k]) * Wl[j,k]   with Wl[j,k] = TRANS ? W[k*ldw + j] : W[j*ldw + k]
//   EPI_BIAS: + bias[j]
//   EPI_DX  : * PReLU'(Z[r,j]) and the per-block sum of value*min(0,Z) goes to slope_part[block]
// Block = 4 waves, each wave owns 32 rows of a 128-row tile; persistent over tiles.  The weight
// panel is staged once per block into LDS with an odd row stride (conflict-free ds_read_b32 for the
// 32-lanes-same-k fragment reads); every tile's rows are staged row-major the same way.
// ---------------------------------------------------------------------------------------------
template <int NS, int EPI, int WAVES>
__global__ __launch_bounds__(256) void linear_mfma_kernel(const float* __restrict__ X, int64_t ldx,
                                                          const float* __restrict__ in_slope,
                                                          const float* __restrict__ W, int32_t ldw, int32_t trans,
                                                          const float* __restrict__ bias, float* __restrict__ Y,
                                                          int64_t ldy, int64_t rows, int32_t K, int32_t N,
                                                          const float* __restrict__ Z, int64_t ldz,
                                                          const float* __restrict__ z_slope,
                                                          double* __restrict__ slope_part, int32_t vec_x) {
  extern __shared__ __align__(16) float smem[];
  const int KP = ((K + 1) & ~1) | 1;  // even K rounded up, then odd stride
  float* Wl = smem;                   // [NS*32][KP]
  float* Xl = smem + (size_t)NS * 32 * KP;  // [WAVES][32][KP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int KE = (K + 1) & ~1;

  // stage weights (zero-padded to NS*32 rows and KE columns)
  constexpr int NT = WAVES * 64;
  constexpr int TM = WAVES * 32;  // rows per tile
  for (int idx = tid; idx < NS * 32 * KE; idx += NT) {
    const int j = idx / KE, k = idx - j * KE;
    float v = 0.f;
    if (j < N && k < K) v = trans ? W[(int64_t)k * ldw + j] : W[(int64_t)j * ldw + k];
    Wl[j * KP + k] = v;
  }
  const float slope = in_slope ? *in_slope : 1.f;
  const bool act = in_slope != nullptr;
  float* Xw = Xl + (size_t)wave * 32 * KP;
  const int64_t ntiles = (rows + TM - 1) / TM;
  double slope_acc = 0.0;  // fp64: the slope gradient is a long signed sum with heavy cancellation
  const float zs = (EPI == EPI_DX && z_slope) ? *z_slope : 1.f;

  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int64_t row0 = t * TM + wave * 32;
    __syncthreads();  // previous tile's fragment reads are done (also orders the Wl staging)
    if (vec_x) {
      const int nv = K >> 2;  // K % 4 == 0 on this path
      for (int idx = lane; idx < 32 * nv; idx += 64) {
        const int r = idx / nv, c4 = (idx - r * nv) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row0 + r < rows) v = *reinterpret_cast<const float4*>(X + (row0 + r) * ldx + c4);
        if (act) {
          v.x = gcl::prelu_f(v.x, slope); v.y = gcl::prelu_f(v.y, slope);
          v.z = gcl::prelu_f(v.z, slope); v.w = gcl::prelu_f(v.w, slope);
        }
        float* d = Xw + r * KP + c4;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
      }
    } else {
      for (int idx = lane; idx < 32 * KE; idx += 64) {
        const int r = idx / KE, k = idx - r * KE;
        float v = 0.f;
        if (row0 + r < rows && k < K) v = X[(row0 + r) * ldx + k];
        if (act) v = gcl::prelu_f(v, slope);
        Xw[r * KP + k] = v;
      }
    }
    __syncthreads();

    f32x16 acc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;

    const float* ap = Xw + (lane & 31) * KP + (lane >> 5);
    const float* bp = Wl + (lane & 31) * KP + (lane >> 5);
    for (int k0 = 0; k0 < KE; k0 += 2) {
      const float a = ap[k0];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const float bv = bp[s * 32 * KP + k0];
        acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[s], 0, 0, 0);
      }
    }

    // epilogue: lane owns column j = s*32 + (lane&31), 16 rows
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int j = s * 32 + (lane & 31);
      if (j < N) {
        const float bj = (EPI == EPI_BIAS && bias) ? bias[j] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = row0 + d_row(r, lane);
          if (row < rows) {
            float v = acc[s][r];
            if (EPI == EPI_BIAS) {
              v += bj;
            } else if (Z) {
              const float z = Z[row * ldz + j];
              if (z <= 0.f) {
                slope_acc += (double)(v * z);
                v *= zs;
              }
            }
            Y[row * ldy + j] = v;
          }
        }
      }
    }
  }

  if (EPI == EPI_DX && slope_part) {
    // block reduction of the slope gradient: wave shuffle, then LDS across the waves
    for (int off = 32; off > 0; off >>= 1) slope_acc += __shfl_down(slope_acc, off, 64);
    __syncthreads();
    double* dred = reinterpret_cast<double*>(smem);
    if (lane == 0) dred[wave] = slope_acc;
    __syncthreads();
    if (tid == 0) {
      double tot = 0.0;
      for (int wv = 0; wv < WAVES; ++wv) tot += dred[wv];
      slope_part[blockIdx.x] = tot;
    }
  }
}

// Simple VALU implementation of the same contract (cross-check; one thread per output element).
template <int EPI>
__global__ __launch_bounds__(256) void linear_valu_kernel(const float* __restrict__ X, int64_t ldx,
                                                          const float* __restrict__ in_slope,
                                                          const float* __restrict__ W, int32_t ldw, int32_t trans,
                                                          const float* __restrict__ bias, float* __restrict__ Y,
                                                          int64_t ldy, int64_t rows, int32_t K, int32_t N,
                                                          const float* __restrict__ Z, int64_t ldz,
                                                          const float* __restrict__ z_slope,
                                                          double* __restrict__ slope_part) {
  __shared__ double red[4];
  const float slope = in_slope ? *in_slope : 1.f;
  const bool act = in_slope != nullptr;
  const float zs = (EPI == EPI_DX && z_slope) ? *z_slope : 1.f;
  double slope_acc = 0.0;
  const int64_t total = rows * N;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t r = idx / N;
    const int j = (int)(idx - r * N);
    float acc = 0.f;
    for (int k = 0; k < K; ++k) {
      float xv = X[r * ldx + k];
      if (act) xv = gcl::prelu_f(xv, slope);
      const float wv = trans ? W[(int64_t)k * ldw + j] : W[(int64_t)j * ldw + k];
      acc = fmaf(xv, wv, acc);
    }
    if (EPI == EPI_BIAS) {
      if (bias) acc += bias[j];
    } else if (Z) {
      const float z = Z[r * ldz + j];
      if (z <= 0.f) {
        slope_acc += (double)(acc * z);
        acc *= zs;
      }
    }
    Y[r * ldy + j] = acc;
  }
  if (EPI == EPI_DX && slope_part) {
    for (int off = 32; off > 0; off >>= 1) slope_acc += __shfl_down(slope_acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = slope_acc;
    __syncthreads();
    if (threadIdx.x == 0) slope_part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
  }
}

// ---------------------------------------------------------------------------------------------
// dW partials:  P[chunk][o][c] = sum_{r in chunk} dY[r,o] * act(X[r,c])   (+ db partials)
// A = dY^T and B = act(X) fragments come straight from global memory (lanes 0-31 read 128
// contiguous bytes of row r, lanes 32-63 of row r+1), no LDS staging.  blockIdx.y = 32-wide slab of
// output channels; each wave accumulates NC slabs of input channels over its rows, the 4 waves are
// summed through LDS, one partial tile per block.
// ---------------------------------------------------------------------------------------------
template <int NC>
__global__ __launch_bounds__(256) void dw_mfma_kernel(const float* __restrict__ dY, int64_t lddy,
                                                      const float* __restrict__ X, int64_t ldx,
                                                      const float* __restrict__ in_slope, float* __restrict__ part,
                                                      float* __restrict__ dbpart, int64_t rows, int32_t Fin,
                                                      int32_t Fout, int64_t rows_per_block, int32_t FinP,
                                                      int32_t FoutP) {
  extern __shared__ __align__(16) float smem[];  // [4][NC][32*32] + [4][32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int o0 = blockIdx.y * 32;
  const int li = lane & 31, lk = lane >> 5;
  const float slope = in_slope ? *in_slope : 1.f;
  const bool act = in_slope != nullptr;
  const int64_t rb = (int64_t)blockIdx.x * rows_per_block;
  const int64_t re = min(rows, rb + rows_per_block);

  f32x16 acc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  float dbacc = 0.f;
  const bool oact = (o0 + li) < Fout;

  // waves interleave row pairs: wave w takes pairs w, w+4, ...
  for (int64_t r = rb + 2 * wave; r < re; r += 8) {
    const int64_t rr = r + lk;
    const bool v = rr < re;
    const float a = (v && oact) ? dY[rr * lddy + o0 + li] : 0.f;
    dbacc += a;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int ch = c * 32 + li;
      float bv = (v && ch < Fin) ? X[rr * ldx + ch] : 0.f;
      if (act) bv = gcl::prelu_f(bv, slope);
      acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[c], 0, 0, 0);
    }
  }

  // cross-wave sum through LDS: tile element (i = out channel, j = in channel)
  float* tile = smem + (size_t)wave * NC * 1024;
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) tile[c * 1024 + d_row(r, lane) * 32 + li] = acc[c][r];
  dbacc += __shfl_xor(dbacc, 32, 64);
  float* dbl = smem + (size_t)4 * NC * 1024;
  if (lane < 32) dbl[wave * 32 + lane] = dbacc;
  __syncthreads();
  float* out = part + ((size_t)blockIdx.x * FoutP + o0) * FinP;
  for (int idx = tid; idx < NC * 1024; idx += 256) {
    const int c = idx >> 10, i = (idx >> 5) & 31, j = idx & 31;
    const float s = smem[idx] + smem[NC * 1024 + idx] + smem[2 * NC * 1024 + idx] + smem[3 * NC * 1024 + idx];
    out[(size_t)i * FinP + c * 32 + j] = s;
  }
  if (dbpart && tid < 32)
    dbpart[(size_t)blockIdx.x * FoutP + o0 + tid] = dbl[tid] + dbl[32 + tid] + dbl[64 + tid] + dbl[96 + tid];
}

// out[i*ldo + j] (+)= sum_p part[p*pstride + i*FinP + j]   for i < R, j < C
__global__ __launch_bounds__(256) void reduce_tiles_kernel(const float* __restrict__ part, int32_t nparts,
                                                           int64_t pstride, int32_t pld, float* __restrict__ out,
                                                           int32_t ldo, int32_t R, int32_t C, int32_t accumulate) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= R * C) return;
  const int i = idx / C, j = idx - i * C;
  float s = 0.f;
  for (int p = 0; p < nparts; ++p) s += part[(size_t)p * pstride + (size_t)i * pld + j];
  float* o = out + (size_t)i * ldo + j;
  *o = accumulate ? *o + s : s;
}

__global__ void reduce_scalar_kernel(const double* __restrict__ part, int32_t nparts, float* __restrict__ out) {
  // single wave; fixed order => deterministic
  double s = 0.0;
  for (int p = threadIdx.x; p < nparts; p += 64) s += part[p];
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (threadIdx.x == 0) *out += (float)s;
}

bool use_valu() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("GCL_LINEAR_IMPL");
    v = (e && strcmp(e, "valu") == 0) ? 1 : 0;
  }
  return v == 1;
}

constexpr int kMaxPersistentBlocks = 1024;
constexpr int kDwBlocks = 256;

struct LinGeom {
  int NS;
  int waves;
  size_t lds;
  int grid;
};

int lin_geom(int64_t rows, int K, int N, LinGeom* g) {
  GCL_CHECK_ARG(K >= 1 && K <= 256 && N >= 1 && N <= 256, "linear: unsupported K=%d N=%d (K<=256, N<=256)", K, N);
  g->NS = (N + 31) / 32;
  const int KP = ((K + 1) & ~1) | 1;
  g->waves = 4;
  g->lds = ((size_t)g->NS * 32 + 128) * KP * sizeof(float);
  if (g->lds > 160 * 1024) {  // wide panels: 64-row tiles (2 waves) keep the weight panel resident
    g->waves = 2;
    g->lds = ((size_t)g->NS * 32 + 64) * KP * sizeof(float);
  }
  GCL_CHECK_ARG(g->lds <= 160 * 1024, "linear: K=%d N=%d needs %zu B of LDS (>160 KiB)", K, N, g->lds);
  const int64_t ntiles = gcl::cdiv(rows, 32 * g->waves);
  g->grid = (int)(ntiles < kMaxPersistentBlocks ? ntiles : kMaxPersistentBlocks);
  return GCL_OK;
}

template <int EPI>
int launch_linear(const float* X, int64_t ldx, const float* in_slope, const float* W, int ldw, int trans,
                  const float* bias, float* Y, int64_t ldy, int64_t rows, int K, int N, const float* Z, int64_t ldz,
                  const float* z_slope, double* slope_part, int* nparts, hipStream_t st) {
  if (rows == 0) {
    if (nparts) *nparts = 0;
    return GCL_OK;
  }
  if (use_valu()) {
    const int64_t total = rows * N;
    int grid = (int)(gcl::cdiv(total, 256) < 4096 ? gcl::cdiv(total, 256) : 4096);
    hipLaunchKernelGGL((linear_valu_kernel<EPI>), dim3(grid), dim3(256), 0, st, X, ldx, in_slope, W, ldw, trans,
                       bias, Y, ldy, rows, K, N, Z, ldz, z_slope, slope_part);
    GCL_CHECK_LAUNCH();
    if (nparts) *nparts = grid;
    return GCL_OK;
  }
  LinGeom g;
  int rc = lin_geom(rows, K, N, &g);
  if (rc) return rc;
  const int vec_x = (K % 4 == 0) && (ldx % 4 == 0) && gcl::aligned16(X);
#define GCL_LIN2(NS_, W_)                                                                                         \
  do {                                                                                                            \
    auto kern = linear_mfma_kernel<NS_, EPI, W_>;                                                                 \
    if (g.lds > 64 * 1024)                                                                                        \
      GCL_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds)); \
    hipLaunchKernelGGL(kern, dim3(g.grid), dim3(W_ * 64), g.lds, st, X, ldx, in_slope, W, ldw, trans, bias, Y,    \
                       ldy, rows, K, N, Z, ldz, z_slope, slope_part, vec_x);                                      \
  } while (0)
#define GCL_LIN(NS_)                 \
  do {                               \
    if (g.waves == 4) GCL_LIN2(NS_, 4); \
    else GCL_LIN2(NS_, 2);           \
  } while (0)
  switch (g.NS) {
    case 1: GCL_LIN(1); break;
    case 2: GCL_LIN(2); break;
    case 3: GCL_LIN(3); break;
    case 4: GCL_LIN(4); break;
    case 5: GCL_LIN(5); break;
    case 6: GCL_LIN(6); break;
    case 7: GCL_LIN(7); break;
    default: GCL_LIN(8); break;
  }
#undef GCL_LIN2
#undef GCL_LIN
  GCL_CHECK_LAUNCH();
  if (nparts) *nparts = g.grid;
  return GCL_OK;
}

}  // namespace

extern "C" int gcl_linear_fwd(const float* x, int64_t ldx, const float* in_slope, const float* W, const float* bias,
                              float* y, int64_t ldy, int64_t rows, int32_t Fin, int32_t Fout, gcl_stream_t stream) {
  GCL_CHECK_ARG(x && W && y, "linear_fwd: null argument");
  GCL_CHECK_ARG(rows >= 0 && ldx >= Fin && ldy >= Fout, "linear_fwd: bad shape rows=%lld ldx=%lld ldy=%lld",
                (long long)rows, (long long)ldx, (long long)ldy);
  return launch_linear<EPI_BIAS>(x, ldx, in_slope, W, Fin, 0, bias, y, ldy, rows, Fin, Fout, nullptr, 0, nullptr,
                                 nullptr, nullptr, (hipStream_t)stream);
}

extern "C" size_t gcl_linear_bwd_ws_bytes(int64_t rows, int32_t Fin, int32_t Fout) {
  const size_t FinP = (size_t)((Fin + 31) / 32) * 32, FoutP = (size_t)((Fout + 31) / 32) * 32;
  const size_t dw = (size_t)kDwBlocks * FoutP * (FinP + 1) * sizeof(float);
  const size_t sl = (size_t)4096 * sizeof(double);
  (void)rows;
  return dw + sl;
}

extern "C" int gcl_linear_bwd_dx(const float* dy, int64_t lddy, const float* W, const float* x, int64_t ldx,
                                 const float* in_slope, float* d_in_slope, float* dx, int64_t lddx, int64_t rows,
                                 int32_t Fin, int32_t Fout, void* ws, size_t ws_bytes, gcl_stream_t stream) {
  GCL_CHECK_ARG(dy && W && dx, "linear_bwd_dx: null argument");
  GCL_CHECK_ARG(lddy >= Fout && lddx >= Fin, "linear_bwd_dx: leading dimension too small");
  GCL_CHECK_ARG(!in_slope || (x && ldx >= Fin), "linear_bwd_dx: in_slope given without the forward input x");
  hipStream_t st = (hipStream_t)stream;
  double* slope_part = nullptr;
  if (in_slope && d_in_slope) {
    GCL_CHECK_ARG(ws && ws_bytes >= 4096 * sizeof(double) && gcl::aligned16(ws), "linear_bwd_dx: workspace too small");
    slope_part = (double*)ws;
  }
  int nparts = 0;
  // contraction over Fout: "weights" are W^T, i.e. Wl[j=c][k=o] = W[o*Fin + c]
  int rc = launch_linear<EPI_DX>(dy, lddy, nullptr, W, Fin, 1, nullptr, dx, lddx, rows, Fout, Fin,
                                 in_slope ? x : nullptr, ldx, in_slope, slope_part, &nparts, st);
  if (rc) return rc;
  if (slope_part && nparts > 0) {
    hipLaunchKernelGGL(reduce_scalar_kernel, dim3(1), dim3(64), 0, st, slope_part, nparts, d_in_slope);
    GCL_CHECK_LAUNCH();
  }
  return GCL_OK;
}

extern "C" int gcl_linear_bwd_dw(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* in_slope,
                                 float* dW, float* db, int64_t rows, int32_t Fin, int32_t Fout, int32_t accumulate,
                                 void* ws, size_t ws_bytes, gcl_stream_t stream) {
  GCL_CHECK_ARG(dy && x && dW, "linear_bwd_dw: null argument");
  GCL_CHECK_ARG(lddy >= Fout && ldx >= Fin, "linear_bwd_dw: leading dimension too small");
  GCL_CHECK_ARG(Fin >= 1 && Fin <= 128 && Fout >= 1 && Fout <= 256, "linear_bwd_dw: unsupported Fin=%d Fout=%d (Fin<=128, Fout<=256)", Fin, Fout);
  GCL_CHECK_ARG(ws && ws_bytes >= gcl_linear_bwd_ws_bytes(rows, Fin, Fout), "linear_bwd_dw: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int NC = (Fin + 31) / 32, NO = (Fout + 31) / 32;
  const int FinP = NC * 32, FoutP = NO * 32;
  int64_t nblk = gcl::cdiv(rows, 64);  // at least 64 rows per block
  if (nblk > kDwBlocks) nblk = kDwBlocks;
  if (nblk < 1) nblk = 1;
  int64_t rpb = gcl::cdiv(rows, nblk);
  rpb = (rpb + 7) & ~(int64_t)7;  // whole 8-row wave rounds
  nblk = rows > 0 ? gcl::cdiv(rows, rpb) : 1;
  float* part = (float*)ws;
  float* dbpart = part + (size_t)kDwBlocks * FoutP * FinP;
  const size_t lds = ((size_t)4 * NC * 1024 + 128) * sizeof(float);
#define GCL_DW(NC_)                                                                                               \
  do {                                                                                                            \
    auto kern = dw_mfma_kernel<NC_>;                                                                              \
    if (lds > 64 * 1024)                                                                                          \
      GCL_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk, NO), dim3(256), lds, st, dy, lddy, x, ldx, in_slope, part,      \
                       db ? dbpart : nullptr, rows, Fin, Fout, rpb, FinP, FoutP);                                 \
  } while (0)
  switch (NC) {
    case 1: GCL_DW(1); break;
    case 2: GCL_DW(2); break;
    case 3: GCL_DW(3); break;
    default: GCL_DW(4); break;
  }
#undef GCL_DW
  GCL_CHECK_LAUNCH();
  hipLaunchKernelGGL(reduce_tiles_kernel, dim3((unsigned)gcl::cdiv((int64_t)Fout * Fin, 256)), dim3(256), 0, st, part,
                     (int)nblk, (int64_t)FoutP * FinP, FinP, dW, Fin, Fout, Fin, accumulate);
  GCL_CHECK_LAUNCH();
  if (db) {
    hipLaunchKernelGGL(reduce_tiles_kernel, dim3((unsigned)gcl::cdiv(Fout, 256)), dim3(256), 0, st, dbpart, (int)nblk,
                       (int64_t)FoutP, FoutP, db, Fout, 1, Fout, accumulate);
    GCL_CHECK_LAUNCH();
  }
  return GCL_OK;
}
