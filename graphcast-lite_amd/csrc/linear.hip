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
#include "x3.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int d_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

enum { EPI_BIAS = 0, EPI_DX = 1 };

// Raw buffer access (hardware range check): loads beyond num_records return 0, stores are dropped.
// Keeping every global access UNCONDITIONAL keeps the kernels free of divergent branches, which is
// what lets hipcc emit counted s_waitcnt vmcnt(N) (CDNA counts loads and stores in one in-order
// counter; a branch around a store forces vmcnt(0) and serialises the whole store stream).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOOB = 0x80000000u;  // offset that is out of range for every descriptor we build
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, int64_t nbytes) {
  const int64_t cap = 0x7FFFFF00;
  const int n = (int)(nbytes < 0 ? 0 : (nbytes > cap ? cap : nbytes));
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, n, 0x00020000);
}
__device__ __forceinline__ float buf_ld1(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
__device__ __forceinline__ float4 buf_ld4(__amdgpu_buffer_rsrc_t r, unsigned off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
  return make_float4(__builtin_bit_cast(float, v.x), __builtin_bit_cast(float, v.y), __builtin_bit_cast(float, v.z),
                     __builtin_bit_cast(float, v.w));
}
#ifndef GCL_ST_AUX
#define GCL_ST_AUX 0  // cache-policy bits of the streamed-out stores (2 = nt)
#endif
__device__ __forceinline__ void buf_st1(__amdgpu_buffer_rsrc_t r, unsigned off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, off, 0, GCL_ST_AUX);
}
// Zero page for masked loads: an out-of-range lane reads from here instead of branching around
// its load (pointer select + unconditional plain load keeps global_load_dwordx4 and no branch).
__device__ float4 gcl_zero4[1];

// bytes of a [nrows x F] window with row stride ld (last row counted only up to F)
__device__ __forceinline__ int64_t win_bytes(int64_t nrows, int64_t ld, int F) {
  return nrows > 0 ? ((nrows - 1) * ld + F) * 4 : 0;
}

#include "gemm_tile.h"

// A/B switch while tuning (kernel argument `abl` bit 8 = keep the per-tile block barriers of round 1)
#define gcl_lin_sync() ((abl & 8) != 0)

// ---------------------------------------------------------------------------------------------
// Y[r, j] = sum_k act(X[r,k]) * Wl[j,k]   with Wl[j,k] = TRANS ? W[k*ldw + j] : W[j*ldw + k]
//   EPI_BIAS: + bias[j]
//   EPI_DX  : * PReLU'(Z[r,j]) and the per-block sum of value*min(0,Z) goes to slope_part[block]
// Block = blockDim.x/64 waves, each wave owns 32 rows of a tile; persistent over tiles.  The weight
// panel is staged once per block into LDS with an odd row stride (conflict-free ds_read_b32 for the
// 32-lanes-same-k fragment reads).  Row tiles are software-pipelined through registers: the global
// loads of tile t+1 are issued before the MFMA loop of tile t and committed to LDS (with the
// activation applied) after it, so HBM latency hides under the matrix work instead of being paid
// once per tile.  KT = compile-time bound on K (size of the prefetch register array).
// ---------------------------------------------------------------------------------------------
// waves per SIMD the register allocator must leave room for (LDS admits ~3 blocks of 4 waves at
// K,N <= 64; wide panels are LDS-limited to 1-2 blocks anyway)
constexpr int lin_min_waves(int NS, int KT) { return (NS <= 2 && KT <= 64) ? 3 : (NS <= 2 && KT <= 128) ? 2 : 1; }
// wide panels (NS >= 4) fill the LDS with one block per CU, i.e. one wave per SIMD: give those
// instantiations the whole register file and fetch Z before the MFMA loop instead of after it
constexpr bool lin_zpre(int NS, int EPI) { return EPI == 1 && NS >= 4; }

template <int NS, int EPI, int KT, bool VEC>
__global__ __launch_bounds__(256, lin_min_waves(NS, KT)) void linear_mfma_kernel(const float* __restrict__ X, int64_t ldx,
                                                          const float* __restrict__ in_slope,
                                                          const float* __restrict__ W, int32_t ldw, int32_t trans,
                                                          const float* __restrict__ bias, float* __restrict__ Y,
                                                          int64_t ldy, int64_t rows, int32_t K, int32_t N,
                                                          const float* __restrict__ Z, int64_t ldz,
                                                          const float* __restrict__ z_slope,
                                                          double* __restrict__ slope_part, int32_t akind,
                                                          int32_t abl) {
  // abl: timing-only ablation bits of tools/ablate.sh (0 in the product): 1 = drop the stores (empty
  // window), 2 = skip the MFMA loop, 4 = loads read the zero page.  Runtime values: codegen is unchanged.
  extern __shared__ __align__(16) float smem[];
  const int KE = (K + 3) & ~3;        // K padded to a multiple of 4 (two MFMA k-steps per 8-B read)
  const int KP = KE + 2;              // even stride with KP/2 odd: conflict-free ds_read_b64 fragments
  float* Wl = smem;                   // [NS*32][KP]
  float* Xl = smem + (size_t)NS * 32 * KP;  // [waves][32][KP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int NT = blockDim.x;
  const int TM = (NT >> 6) * 32;  // rows per tile

  // weight panel: 8 independent loads in flight per thread before the first LDS write (a load -> wait ->
  // write loop costs one L2 round trip per element and that latency is paid by every block's first tile)
  for (int base = 0; base < NS * 32 * KE; base += NT * 8) {
    float wv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + u * NT + tid;
      const int j = idx / KE, k = idx - j * KE;
      const bool ok = (idx < NS * 32 * KE) && (j < N) && (k < K);
      const float* src = ok ? (trans ? W + (int64_t)k * ldw + j : W + (int64_t)j * ldw + k)
                            : reinterpret_cast<const float*>(gcl_zero4);
      wv[u] = *src;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + u * NT + tid;
      const int j = idx / KE, k = idx - j * KE;
      if (idx < NS * 32 * KE) Wl[j * KP + k] = wv[u];
    }
  }
  // akind: activation of X (EPI_BIAS) or of Z (EPI_DX); PReLU reads its slope from in_slope / z_slope
  const float slope = in_slope ? *in_slope : 1.f;
  const bool act = EPI == EPI_BIAS && akind != gcl::kActNone;
  const bool silu = akind == gcl::kActSilu;
  float* Xw = Xl + (size_t)wave * 32 * KP;
  const int64_t ntiles = (rows + TM - 1) / TM;
  double slope_acc = 0.0;  // fp64: the slope gradient is a long signed sum with heavy cancellation
  const float zs = (EPI == EPI_DX && z_slope) ? *z_slope : 1.f;

  // Staging map of a wave's 32 x K tile (no divisions).  VEC: a row is KT/4 float4 columns, a
  // wave-instruction covers 64/(KT/4) rows; scalar: a row is KT columns, KT/64 instructions per row.
  constexpr int CPR = VEC ? KT / 4 : KT;                  // columns (float4 or float) per row slot
  constexpr int RPI = CPR >= 64 ? 1 : 64 / CPR;           // rows per wave-instruction
  constexpr int IPR = CPR >= 64 ? CPR / 64 : 1;           // instructions per row
  constexpr int NIT = (32 / RPI) * IPR;                   // wave-instructions per tile
  const int rsub = (CPR >= 64) ? 0 : lane / CPR;
  const int csub = (CPR >= 64) ? lane : lane % CPR;
  float pre[VEC ? NIT * 4 : NIT];
  const int64_t rows_ld = (abl & 4) ? 0 : rows;
  auto issue = [&](int64_t tile) {  // rows past the end (and tiles past the last) read the zero page
    const int64_t r0 = tile * TM + wave * 32;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int64_t row = r0 + (it / IPR) * RPI + rsub;
      const int c = csub + 64 * (it % IPR);
      if (VEC) {
        const bool ok = (c * 4 < K) && (row < rows_ld);
        const float4* p = ok ? reinterpret_cast<const float4*>(X + row * ldx + c * 4) : gcl_zero4;
        const float4 v = *p;
        pre[4 * it] = v.x; pre[4 * it + 1] = v.y; pre[4 * it + 2] = v.z; pre[4 * it + 3] = v.w;
      } else {
        const bool ok = (c < K) && (row < rows_ld);
        const float* p = ok ? X + row * ldx + c : reinterpret_cast<const float*>(gcl_zero4);
        pre[it] = *p;
      }
    }
  };
  // Activation while committing: PReLU and "none" share one branch-free form (x > 0 ? x : x * a with a = 1 for
  // none: exact); SiLU is a block-uniform variant of the whole commit, not a branch per element.
  const float pa = (act && !silu) ? slope : 1.f;
  auto commit = [&]() {
    if (act && silu) {
#pragma unroll
      for (int q = 0; q < (VEC ? NIT * 4 : NIT); ++q) pre[q] = gcl::silu_f(pre[q]);
    } else {
#pragma unroll
      for (int q = 0; q < (VEC ? NIT * 4 : NIT); ++q) pre[q] = pre[q] > 0.f ? pre[q] : pre[q] * pa;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int r = (it / IPR) * RPI + rsub;
      const int c = csub + 64 * (it % IPR);
      if (VEC) {
        if (c * 4 < KE) {
          float2* d = reinterpret_cast<float2*>(Xw + r * KP + c * 4);
          d[0] = make_float2(pre[4 * it], pre[4 * it + 1]);
          d[1] = make_float2(pre[4 * it + 2], pre[4 * it + 3]);
        }
      } else {
        if (c < KE) Xw[r * KP + c] = pre[it];  // zero beyond K
      }
    }
  };

  // bias of this lane's columns (loaded once, outside the tile loop)
  float bj[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int j = s * 32 + (lane & 31);
    bj[s] = (EPI == EPI_BIAS && bias && j < N) ? bias[j] : 0.f;
  }
  const bool has_z = (EPI == EPI_DX) && (Z != nullptr);

  issue(blockIdx.x);
  __syncthreads();  // the weight panel is staged; from here on every wave only touches ITS rows of Xl
  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    // No block barrier inside the loop: a wave's staging rows are private to it and LDS operations of one
    // wave complete in order, so the waves of a block (and of the other blocks on the CU) drift apart and one
    // wave's MFMA chain overlaps the others' loads, LDS commits and stores.
    if (!gcl_lin_sync()) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    else __syncthreads();
    commit();
    if (gcl_lin_sync()) __syncthreads();
    issue(t + gridDim.x);  // unconditional: an empty window past the last tile returns zeros

    const int64_t r0 = t * TM;
    const int64_t nr = (rows - r0) < TM ? (rows - r0) : TM;
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(Y + r0 * ldy, (abl & 1) ? 0 : win_bytes(nr, ldy, N));
    const __amdgpu_buffer_rsrc_t rz = make_rsrc(has_z ? Z + r0 * ldz : Y, has_z ? win_bytes(nr, ldz, N) : 0);

    f32x16 acc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;

    constexpr bool ZPRE = lin_zpre(NS, EPI);
    float zpre[ZPRE ? NS : 1][16];
    if (ZPRE) {  // Z of this tile: in flight during the MFMA loop
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int j = s * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rr = wave * 32 + d_row(r, lane);
          zpre[ZPRE ? s : 0][r] = buf_ld1(rz, (j < N) ? (unsigned)((rr * ldz + j) * 4) : kOOB);
        }
      }
    }

    // k-step pair q covers k = 4q..4q+3: lane half h reads the 8 bytes k = 4q+2h, 4q+2h+1 and
    // feeds .x to the first MFMA (k = 4q | 4q+2) and .y to the second (k = 4q+1 | 4q+3); A and B
    // use the same assignment.  Fragments of pair q+1 are read before the MFMAs of pair q issue.
    const float* ap = Xw + (lane & 31) * KP + 2 * (lane >> 5);
    const float* bp = Wl + (lane & 31) * KP + 2 * (lane >> 5);
    const int nq = (abl & 2) ? 0 : (KE >> 2);
    float2 a_c = *reinterpret_cast<const float2*>(ap);
    float2 b_c[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) b_c[s] = *reinterpret_cast<const float2*>(bp + s * 32 * KP);
    for (int q = 0; q < nq; ++q) {
      // reads one pair ahead; the last iteration re-reads pair nq-1 (in range, unused)
      const int qn = (q + 1 < nq) ? q + 1 : q;
      const float2 a_n = *reinterpret_cast<const float2*>(ap + qn * 4);
      float2 b_n[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) b_n[s] = *reinterpret_cast<const float2*>(bp + s * 32 * KP + qn * 4);
#pragma unroll
      for (int s = 0; s < NS; ++s) acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c.x, b_c[s].x, acc[s], 0, 0, 0);
#pragma unroll
      for (int s = 0; s < NS; ++s) acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c.y, b_c[s].y, acc[s], 0, 0, 0);
      a_c = a_n;
#pragma unroll
      for (int s = 0; s < NS; ++s) b_c[s] = b_n[s];
    }

    // epilogue: lane owns column j = s*32 + (lane&31) and 16 rows; all accesses unconditional
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int j = s * 32 + (lane & 31);
      const bool jok = j < N;
      if (EPI == EPI_DX) {
        float zv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rr = wave * 32 + d_row(r, lane);
          if (ZPRE) zv[r] = zpre[ZPRE ? s : 0][r];
          else zv[r] = buf_ld1(rz, jok ? (unsigned)((rr * ldz + j) * 4) : kOOB);  // 0 when absent / out of range
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rr = wave * 32 + d_row(r, lane);
          float v = acc[s][r];
          if (silu) {  // block-uniform
            v *= has_z ? gcl::dsilu_f(zv[r]) : 1.f;
          } else {
            const bool neg = has_z && (zv[r] <= 0.f);
            slope_acc += neg ? (double)(v * zv[r]) : 0.0;
            v = neg ? v * zs : v;
          }
          buf_st1(ry, jok ? (unsigned)((rr * ldy + j) * 4) : kOOB, v);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rr = wave * 32 + d_row(r, lane);
          buf_st1(ry, jok ? (unsigned)((rr * ldy + j) * 4) : kOOB, acc[s][r] + bj[s]);
        }
      }
    }
  }

  if (EPI == EPI_DX && slope_part) {
    for (int off = 32; off > 0; off >>= 1) slope_acc += __shfl_down(slope_acc, off, 64);
    __syncthreads();
    double* dred = reinterpret_cast<double*>(smem);
    if (lane == 0) dred[wave] = slope_acc;
    __syncthreads();
    if (tid == 0) {
      double tot = 0.0;
      for (int wv = 0; wv < (NT >> 6); ++wv) tot += dred[wv];
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
                                                          double* __restrict__ slope_part, int32_t akind) {
  __shared__ double red[4];
  const float slope = in_slope ? *in_slope : 1.f;
  const bool act = EPI == EPI_BIAS && akind != gcl::kActNone;
  const float zs = (EPI == EPI_DX && z_slope) ? *z_slope : 1.f;
  double slope_acc = 0.0;
  const int64_t total = rows * N;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t r = idx / N;
    const int j = (int)(idx - r * N);
    float acc = 0.f;
    for (int k = 0; k < K; ++k) {
      float xv = X[r * ldx + k];
      if (act) xv = gcl::act_f(xv, slope, akind);
      const float wv = trans ? W[(int64_t)k * ldw + j] : W[(int64_t)j * ldw + k];
      acc = fmaf(xv, wv, acc);
    }
    if (EPI == EPI_BIAS) {
      if (bias) acc += bias[j];
    } else if (Z) {
      const float z = Z[r * ldz + j];
      if (akind == gcl::kActSilu) {
        acc *= gcl::dsilu_f(z);
      } else if (z <= 0.f) {
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
// dW partials:  P[block][o][c] = sum_{r in block's rows} dY[r,o] * act(X[r,c])   (+ db partials)
// A tall-skinny contraction over rows.  Each block walks its row range in 64-row steps: the dY and
// act(X) rows of a step are fetched with 16-B loads into registers one step ahead (software
// pipeline), committed to LDS, and the 4 waves run v_mfma_f32_32x32x2_f32 with A = dY^T and B = X
// fragments read from LDS (lanes 0-31 read 32 consecutive channels of row k, lanes 32-63 of row
// k+1: conflict-free).  The 32x32 output tiles are split over the waves as a WO x WC grid; wave
// (wo,wc) owns o-slabs {wo, wo+WO, ..} x c-slabs {wc, wc+WC, ..} (TO x TC accumulators), so no
// cross-wave reduction is needed.  One partial tile set per block, reduced by reduce_parts_kernel.
// ---------------------------------------------------------------------------------------------
constexpr int kDwRT = 64;  // rows per step

template <int NO, int NC>
struct DwCfg {
  static constexpr int WO = NO >= 4 ? 4 : NO;         // waves along the output-channel slabs
  static constexpr int WC = 4 / WO;                   // waves along the input-channel slabs
  static constexpr int TO = NO / WO;                  // o-slabs per wave
  static constexpr int TC = (NC + WC - 1) / WC;       // c-slabs per wave
  static constexpr int PRE = 2 * (NO + NC);           // float4 staging registers per thread
  static constexpr int FoutP = NO * 32, FinP = NC * 32;
  static constexpr int min_waves = (PRE <= 8 && TO * TC <= 2) ? 3 : (PRE <= 16 && TO * TC <= 4) ? 2 : 1;
};

template <int NO, int NC, bool VEC>
__global__ __launch_bounds__(256, (DwCfg<NO, NC>::min_waves)) void dw_mfma_kernel(
    const float* __restrict__ dY, int64_t lddy, const float* __restrict__ X, int64_t ldx,
    const float* __restrict__ in_slope, float* __restrict__ part, float* __restrict__ dbpart, int64_t rows,
    int32_t Fin_all, int32_t Fout, int64_t rows_per_block, int32_t akind, int64_t tile_stride) {
  using Cfg = DwCfg<NO, NC>;
  // blockIdx.y = 128-column chunk of the input features (one launch covers a whole wide dW)
  const int ct = blockIdx.y;
  X += (int64_t)ct * 128;
  part += (int64_t)ct * tile_stride;
  if (ct != 0) dbpart = nullptr;
  const int Fin = (Fin_all - ct * 128) < 128 ? (Fin_all - ct * 128) : 128;
  constexpr int WO = Cfg::WO, WC = Cfg::WC, TO = Cfg::TO, TC = Cfg::TC, PRE = Cfg::PRE;
  constexpr int FoutP = Cfg::FoutP, FinP = Cfg::FinP;
  extern __shared__ __align__(16) float smem[];
  float* Yl = smem;                          // [kDwRT][FoutP]
  float* Xl = smem + (size_t)kDwRT * FoutP;  // [kDwRT][FinP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lk = lane >> 5;
  const int wo = wave % WO, wc = wave / WO;
  const float slope = in_slope ? *in_slope : 1.f;
  const bool act = akind != gcl::kActNone;
  const int64_t rb = (int64_t)blockIdx.x * rows_per_block;
  const int64_t re = min(rows, rb + rows_per_block);
  const int nrows = (int)(re - rb);

  f32x16 acc[TO][TC];
#pragma unroll
  for (int a = 0; a < TO; ++a)
#pragma unroll
    for (int c = 0; c < TC; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
  float dbacc[TO];
#pragma unroll
  for (int a = 0; a < TO; ++a) dbacc[a] = 0.f;

  // Staging map of a 64-row step (no divisions): thread = (row slot rs = tid/8, column group
  // cg = tid%8); it owns rows rs and rs+32 and the float4 columns cg + 8*j of dY (j < NO) and of
  // X (j < NC).  8 consecutive lanes read 128 contiguous bytes.  All loads are unconditional:
  // rows past the block's range and columns past F read the zero page.
  const int rs = tid >> 3, cg = tid & 7;
  float4 pre[PRE];
  const float* ybase = dY + rb * lddy;
  const float* xbase = X + rb * ldx;
  auto ld_slot = [&](const float* base, int rr, int c4, int64_t ld, int F) -> float4 {
    const bool ok = rr < nrows;
    const float* z = reinterpret_cast<const float*>(gcl_zero4);
    const float* src = base + (int64_t)rr * ld + c4;
    if (VEC) return *reinterpret_cast<const float4*>((ok && c4 < F) ? src : z);
    float4 v;
    v.x = *((ok && c4 < F) ? src : z);
    v.y = *((ok && c4 + 1 < F) ? src + 1 : z);
    v.z = *((ok && c4 + 2 < F) ? src + 2 : z);
    v.w = *((ok && c4 + 3 < F) ? src + 3 : z);
    return v;
  };
  auto issue = [&](int rel0) {
#pragma unroll
    for (int it = 0; it < PRE; ++it) {
      const int j = it >> 1;
      const int rr = rel0 + rs + 32 * (it & 1);
      if (j < NO) pre[it] = ld_slot(ybase, rr, (cg + 8 * j) * 4, lddy, Fout);
      else pre[it] = ld_slot(xbase, rr, (cg + 8 * (j - NO)) * 4, ldx, Fin);
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int it = 0; it < PRE; ++it) {
      const int j = it >> 1;
      const int r = rs + 32 * (it & 1);
      float4 v = pre[it];
      if (j < NO) {
        *reinterpret_cast<float4*>(Yl + r * FoutP + (cg + 8 * j) * 4) = v;
      } else {
        if (act) {
          v.x = gcl::act_f(v.x, slope, akind); v.y = gcl::act_f(v.y, slope, akind);
          v.z = gcl::act_f(v.z, slope, akind); v.w = gcl::act_f(v.w, slope, akind);
        }
        *reinterpret_cast<float4*>(Xl + r * FinP + (cg + 8 * (j - NO)) * 4) = v;
      }
    }
  };

  issue(0);
  for (int rel0 = 0; rel0 < nrows; rel0 += kDwRT) {
    __syncthreads();
    commit();
    __syncthreads();
    issue(rel0 + kDwRT);  // unconditional (zeros past the end)
    const float* yp = Yl + lk * FoutP + li;
    const float* xp = Xl + lk * FinP + li;
    float av[TO], bv[TC];
#pragma unroll
    for (int a = 0; a < TO; ++a) av[a] = yp[(wo + a * WO) * 32];
#pragma unroll
    for (int c = 0; c < TC; ++c) {
      const int sc = wc + c * WC;
      bv[c] = (sc < NC) ? xp[sc * 32] : 0.f;
    }
    for (int k = 0; k < kDwRT; k += 2) {
      // fragments of the next row pair, read before this pair's MFMAs issue (last: re-read)
      const int kn = (k + 2 < kDwRT) ? k + 2 : k;
      float an[TO], bn[TC];
#pragma unroll
      for (int a = 0; a < TO; ++a) an[a] = yp[kn * FoutP + (wo + a * WO) * 32];
#pragma unroll
      for (int c = 0; c < TC; ++c) {
        const int sc = wc + c * WC;
        bn[c] = (sc < NC) ? xp[kn * FinP + sc * 32] : 0.f;
      }
#pragma unroll
      for (int a = 0; a < TO; ++a) {
        dbacc[a] += av[a];
#pragma unroll
        for (int c = 0; c < TC; ++c)
          acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[c], acc[a][c], 0, 0, 0);
      }
#pragma unroll
      for (int a = 0; a < TO; ++a) av[a] = an[a];
#pragma unroll
      for (int c = 0; c < TC; ++c) bv[c] = bn[c];
    }
  }

  float* out = part + (size_t)blockIdx.x * FoutP * FinP;
#pragma unroll
  for (int a = 0; a < TO; ++a) {
    const int so = wo + a * WO;
#pragma unroll
    for (int c = 0; c < TC; ++c) {
      const int sc = wc + c * WC;
      if (sc < NC) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          out[(size_t)(so * 32 + d_row(r, lane)) * FinP + sc * 32 + li] = acc[a][c][r];
      }
    }
    if (dbpart && wc == 0) {
      const float d = dbacc[a] + __shfl_xor(dbacc[a], 32, 64);
      if (lane < 32) dbpart[(size_t)blockIdx.x * FoutP + so * 32 + lane] = d;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Fused backward of  y = act(x) W^T (+ b):  one pass over dY and the pre-activation input P.
//   dX[r,c]  = (sum_o dY[r,o] W[o,c]) * PReLU'(P[r,c])          -> global (raw buffer stores)
//   dW[o,c] += sum_r dY[r,o] act(P[r,c])                         -> per-block partial tiles
//   db[o]   += sum_r dY[r,o],  cs[c] += sum_r dX[r,c],  dslope += sum dX_preact * min(0,P)
// Traffic per call: read dY + read P + write dX (600 MB at 786k x 64) instead of the 1600 MB the
// separate dx / dW / colsum launches move, because P doubles as Z (PReLU') and as X (dW), and dY
// is read once.  One 4-wave block per CU (LDS ~84 KB at 64x64), one wave per SIMD, explicit
// register prefetch of the next tile during the ~8k MFMA cycles of the current one.
//   LDS: Wt[FiP][KPo] (W transposed, for dX), dYl[128][KPo] (row-major: A of dX as 8-B fragments,
//   A' of dW as channel-contiguous 4-B fragments), Pl[128][FiP] (pre-activation; PReLU is applied
//   when the B' fragment is formed, PReLU' in the dX epilogue).
// Constraints (else the caller falls back to the separate kernels): Fout <= 64, Fin <= 96,
// Fout % 4 == Fin % 4 == 0, 16-B aligned rows.
// ---------------------------------------------------------------------------------------------
template <int NO, int NC>
__global__ __launch_bounds__(256, 1) void linear_bwd_fused_kernel(
    const float* __restrict__ dY, int64_t lddy, const float* __restrict__ W, const float* __restrict__ P,
    int64_t ldp, const float* __restrict__ in_slope, float* __restrict__ dX, int64_t lddx, int64_t rows, int32_t Fin,
    int32_t Fout, float* __restrict__ part_dw, float* __restrict__ part_db, float* __restrict__ part_cs,
    double* __restrict__ part_slope) {
  constexpr int FoP = NO * 32, FiP = NC * 32;
  constexpr int KPo = FoP + 2;             // even stride, KPo/2 odd
  constexpr int TM = 128;
  constexpr int NT = NO * NC;              // dW tiles
  constexpr int TPW = (NT + 3) / 4;        // dW tiles per wave
  constexpr int CY = FoP / 4;              // float4 columns of a dY row (8 or 16)
  constexpr int CP = NC == 3 ? 32 : FiP / 4;  // float4 column slots of a P row (power of two)
  constexpr int RY = 64 / CY, RP = 64 / CP;   // rows per wave-instruction
  constexpr int NY = 32 / RY, NP = 32 / RP;   // wave-instructions per 32-row slice
  extern __shared__ __align__(16) float smem[];
  float* Wt = smem;                         // [FiP][KPo]
  float* dYl = Wt + FiP * KPo;              // [TM][KPo]
  float* Pl = dYl + TM * KPo;               // [TM][FiP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lk = lane >> 5;

  for (int idx = tid; idx < FiP * FoP; idx += 256) {  // Wt[c][o] = W[o][c], zero padded
    const int c = idx / FoP, o = idx - c * FoP;
    Wt[c * KPo + o] = (c < Fin && o < Fout) ? W[(int64_t)o * Fin + c] : 0.f;
  }
  const bool act = in_slope != nullptr;
  const float slope = act ? *in_slope : 1.f;
  const int64_t ntiles = (rows + TM - 1) / TM;

  f32x16 dw[TPW];
#pragma unroll
  for (int q = 0; q < TPW; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) dw[q][r] = 0.f;
  float dbacc[TPW], cs[NC];
#pragma unroll
  for (int q = 0; q < TPW; ++q) dbacc[q] = 0.f;
#pragma unroll
  for (int s2 = 0; s2 < NC; ++s2) cs[s2] = 0.f;
  double slope_acc = 0.0;

  const int ysub = lane / CY, ycol = lane % CY, psub = lane / CP, pcol = lane % CP;
  float4 pre[NY + NP];
  const float4* zero = gcl_zero4;
  auto issue = [&](int64_t tile) {
    const int64_t r0 = tile * TM + wave * 32;
#pragma unroll
    for (int it = 0; it < NY; ++it) {
      const int64_t row = r0 + it * RY + ysub;
      const bool ok = (ycol * 4 < Fout) && (row < rows);
      pre[it] = *(ok ? reinterpret_cast<const float4*>(dY + row * lddy + ycol * 4) : zero);
    }
#pragma unroll
    for (int it = 0; it < NP; ++it) {
      const int64_t row = r0 + it * RP + psub;
      const bool ok = (pcol * 4 < Fin) && (row < rows);
      pre[NY + it] = *(ok ? reinterpret_cast<const float4*>(P + row * ldp + pcol * 4) : zero);
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int it = 0; it < NY; ++it) {
      const int r = wave * 32 + it * RY + ysub;
      float2* d = reinterpret_cast<float2*>(dYl + r * KPo + ycol * 4);
      const float4 v = pre[it];
      d[0] = make_float2(v.x, v.y);
      d[1] = make_float2(v.z, v.w);
    }
#pragma unroll
    for (int it = 0; it < NP; ++it) {
      const int r = wave * 32 + it * RP + psub;
      const float4 v = pre[NY + it];
      if (pcol * 4 < FiP) *reinterpret_cast<float4*>(Pl + r * FiP + pcol * 4) = v;
    }
  };

  issue(blockIdx.x);
  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    __syncthreads();
    commit();
    __syncthreads();
    issue(t + gridDim.x);

    const int64_t r0 = t * TM;
    const int64_t nr = (rows - r0) < TM ? (rows - r0) : TM;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(dX + r0 * lddx, win_bytes(nr, lddx, Fin));

    // ---- dX slice of this wave: [32 x Fin] = dYl[32 rows] (K = Fout) x Wt ----
    f32x16 acc[NC];
#pragma unroll
    for (int s2 = 0; s2 < NC; ++s2)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[s2][r] = 0.f;
    {
      const float* ap = dYl + (wave * 32 + li) * KPo + 2 * lk;
      const float* bp = Wt + li * KPo + 2 * lk;
      constexpr int nq = FoP / 4;
      float2 a_c = *reinterpret_cast<const float2*>(ap);
      float2 b_c[NC];
#pragma unroll
      for (int s2 = 0; s2 < NC; ++s2) b_c[s2] = *reinterpret_cast<const float2*>(bp + s2 * 32 * KPo);
#pragma unroll 2
      for (int q = 0; q < nq; ++q) {
        const int qn = (q + 1 < nq) ? q + 1 : q;
        const float2 a_n = *reinterpret_cast<const float2*>(ap + qn * 4);
        float2 b_n[NC];
#pragma unroll
        for (int s2 = 0; s2 < NC; ++s2) b_n[s2] = *reinterpret_cast<const float2*>(bp + s2 * 32 * KPo + qn * 4);
#pragma unroll
        for (int s2 = 0; s2 < NC; ++s2)
          acc[s2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c.x, b_c[s2].x, acc[s2], 0, 0, 0);
#pragma unroll
        for (int s2 = 0; s2 < NC; ++s2)
          acc[s2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c.y, b_c[s2].y, acc[s2], 0, 0, 0);
        a_c = a_n;
#pragma unroll
        for (int s2 = 0; s2 < NC; ++s2) b_c[s2] = b_n[s2];
      }
    }
    // epilogue: PReLU' from the LDS copy of P, slope / column-sum accumulation; branch-free (all
    // 16 Z reads of a slab are issued together, masks are selects, stores are unconditional)
#pragma unroll
    for (int s2 = 0; s2 < NC; ++s2) {
      const int j = s2 * 32 + li;
      const bool jok = j < Fin;
      float zv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) zv[r] = Pl[(wave * 32 + d_row(r, lane)) * FiP + j];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = wave * 32 + d_row(r, lane);
        float v = acc[s2][r];
        const bool neg = act && (zv[r] <= 0.f);
        slope_acc += neg ? (double)(v * zv[r]) : 0.0;
        v = neg ? v * slope : v;
        cs[s2] += v;
        buf_st1(rx, jok ? (unsigned)((rr * lddx + j) * 4) : kOOB, v);
      }
    }

    // ---- dW tiles of this wave over all 128 rows of the tile ----
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
      const int tix = wave + 4 * q;
      if (tix < NT) {  // wave-uniform
        const int so = tix % NO, sc = tix / NO;
        const float* yp = dYl + lk * KPo + so * 32 + li;
        const float* xp = Pl + lk * FiP + sc * 32 + li;
        constexpr int DP = 4;  // LDS fragments are read 4 row pairs ahead of their MFMAs
        float av[DP], bv[DP];
#pragma unroll
        for (int d = 0; d < DP; ++d) {
          av[d] = yp[(2 * d) * KPo];
          bv[d] = xp[(2 * d) * FiP];
        }
        for (int k = 0; k < TM; k += 2 * DP) {
          float an[DP], bn[DP];
#pragma unroll
          for (int d = 0; d < DP; ++d) {
            const int kn = (k + 2 * DP + 2 * d < TM) ? k + 2 * DP + 2 * d : 2 * d;
            an[d] = yp[kn * KPo];
            bn[d] = xp[kn * FiP];
          }
#pragma unroll
          for (int d = 0; d < DP; ++d) {
            const float bx = act ? gcl::prelu_f(bv[d], slope) : bv[d];
            if (sc == 0) dbacc[q] += av[d];
            dw[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[d], bx, dw[q], 0, 0, 0);
          }
#pragma unroll
          for (int d = 0; d < DP; ++d) {
            av[d] = an[d];
            bv[d] = bn[d];
          }
        }
      }
    }
  }

  // ---- per-block partials ----
  constexpr size_t REC = (size_t)FoP * FiP + FoP + FiP;  // per-block record: [dW | db | colsum]
  float* out = part_dw + (size_t)blockIdx.x * REC;
#pragma unroll
  for (int q = 0; q < TPW; ++q) {
    const int tix = wave + 4 * q;
    if (tix < NT) {
      const int so = tix % NO, sc = tix / NO;
#pragma unroll
      for (int r = 0; r < 16; ++r) out[(size_t)(so * 32 + d_row(r, lane)) * FiP + sc * 32 + li] = dw[q][r];
      if (part_db && sc == 0) {
        const float d = dbacc[q] + __shfl_xor(dbacc[q], 32, 64);
        if (lane < 32) part_db[(size_t)blockIdx.x * REC + so * 32 + lane] = d;
      }
    }
  }
  __syncthreads();  // LDS is free: reuse it for the cross-wave column sums and the slope partial
  float* red = smem;  // [4][FiP]
#pragma unroll
  for (int s2 = 0; s2 < NC; ++s2) {
    const float v = cs[s2] + __shfl_xor(cs[s2], 32, 64);
    if (lane < 32) red[wave * FiP + s2 * 32 + lane] = v;
  }
  for (int off = 32; off > 0; off >>= 1) slope_acc += __shfl_down(slope_acc, off, 64);
  double* dred = reinterpret_cast<double*>(smem + 4 * FiP);
  if (lane == 0) dred[wave] = slope_acc;
  __syncthreads();
  if (part_cs)
    for (int c = tid; c < FiP; c += 256)
      part_cs[(size_t)blockIdx.x * REC + c] = (red[c] + red[FiP + c]) + (red[2 * FiP + c] + red[3 * FiP + c]);
  if (part_slope && tid == 0) part_slope[blockIdx.x] = (dred[0] + dred[1]) + (dred[2] + dred[3]);
}

// ---------------------------------------------------------------------------------------------
// Fused backward, 64-wide input panel (Fin in 33..64), 64-row tiles: 50 KB of LDS per block, so
// THREE blocks (12 waves) share a CU instead of one - the loads of one block hide behind the MFMAs
// of the others.  Waves are (row group rg = w&1, column slab sg = w>>1): wave (rg,sg) computes the
// dX slice rows 32rg.. x columns 32sg.., and dW tile w (of NO x 2) over the tile's 64 rows.
// Same partial-record layout and outputs as linear_bwd_fused_kernel.
// ---------------------------------------------------------------------------------------------
template <int NO>
__global__ __launch_bounds__(256, 3) void linear_bwd_fused64_kernel(
    const float* __restrict__ dY, int64_t lddy, const float* __restrict__ W, const float* __restrict__ P,
    int64_t ldp, const float* __restrict__ in_slope, float* __restrict__ dX, int64_t lddx, int64_t rows, int32_t Fin,
    int32_t Fout, float* __restrict__ part_dw, float* __restrict__ part_db, float* __restrict__ part_cs,
    double* __restrict__ part_slope) {
  constexpr int NC = 2;
  constexpr int FoP = NO * 32, FiP = NC * 32;
  constexpr int KPo = FoP + 2;
  constexpr int TM = 64;
  constexpr int NT = NO * NC;
  constexpr int CY = FoP / 4, CP = FiP / 4;     // float4 columns per row (8|16, 16)
  constexpr int RY = 64 / CY, RP = 64 / CP;     // rows per wave-instruction
  constexpr int NY = 16 / RY, NP = 16 / RP;     // wave-instructions per 16-row slice
  extern __shared__ __align__(16) float smem[];
  float* Wt = smem;               // [FiP][KPo]
  float* dYl = Wt + FiP * KPo;    // [TM][KPo]
  float* Pl = dYl + TM * KPo;     // [TM][FiP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lk = lane >> 5;
  const int rg = wave & 1, sg = wave >> 1;

  for (int idx = tid; idx < FiP * FoP; idx += 256) {
    const int c = idx / FoP, o = idx - c * FoP;
    Wt[c * KPo + o] = (c < Fin && o < Fout) ? W[(int64_t)o * Fin + c] : 0.f;
  }
  const bool act = in_slope != nullptr;
  const float slope = act ? *in_slope : 1.f;
  const int64_t ntiles = (rows + TM - 1) / TM;

  f32x16 dw;
#pragma unroll
  for (int r = 0; r < 16; ++r) dw[r] = 0.f;
  float dbacc = 0.f, cs = 0.f;
  double slope_acc = 0.0;
  const bool has_tile = wave < NT;
  const int so = wave % NO, sc = wave / NO;

  const int ysub = lane / CY, ycol = lane % CY, psub = lane / CP, pcol = lane % CP;
  float4 pre[NY + NP];
  const float4* zero = gcl_zero4;
  auto issue = [&](int64_t tile) {
    const int64_t r0 = tile * TM + wave * 16;
#pragma unroll
    for (int it = 0; it < NY; ++it) {
      const int64_t row = r0 + it * RY + ysub;
      const bool ok = (ycol * 4 < Fout) && (row < rows);
      pre[it] = *(ok ? reinterpret_cast<const float4*>(dY + row * lddy + ycol * 4) : zero);
    }
#pragma unroll
    for (int it = 0; it < NP; ++it) {
      const int64_t row = r0 + it * RP + psub;
      const bool ok = (pcol * 4 < Fin) && (row < rows);
      pre[NY + it] = *(ok ? reinterpret_cast<const float4*>(P + row * ldp + pcol * 4) : zero);
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int it = 0; it < NY; ++it) {
      const int r = wave * 16 + it * RY + ysub;
      float2* d = reinterpret_cast<float2*>(dYl + r * KPo + ycol * 4);
      const float4 v = pre[it];
      d[0] = make_float2(v.x, v.y);
      d[1] = make_float2(v.z, v.w);
    }
#pragma unroll
    for (int it = 0; it < NP; ++it) {
      const int r = wave * 16 + it * RP + psub;
      *reinterpret_cast<float4*>(Pl + r * FiP + pcol * 4) = pre[NY + it];
    }
  };

  issue(blockIdx.x);
  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    __syncthreads();
    commit();
    __syncthreads();
    issue(t + gridDim.x);

    const int64_t r0 = t * TM;
    const int64_t nr = (rows - r0) < TM ? (rows - r0) : TM;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(dX + r0 * lddx, win_bytes(nr, lddx, Fin));

    // ---- dX slice: rows 32rg.., columns 32sg.. ----
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    {
      const float* ap = dYl + (rg * 32 + li) * KPo + 2 * lk;
      const float* bp = Wt + (sg * 32 + li) * KPo + 2 * lk;
      constexpr int nq = FoP / 4;
      float2 a_c = *reinterpret_cast<const float2*>(ap);
      float2 b_c = *reinterpret_cast<const float2*>(bp);
#pragma unroll 4
      for (int q = 0; q < nq; ++q) {
        const int qn = (q + 1 < nq) ? q + 1 : q;
        const float2 a_n = *reinterpret_cast<const float2*>(ap + qn * 4);
        const float2 b_n = *reinterpret_cast<const float2*>(bp + qn * 4);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c.x, b_c.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c.y, b_c.y, acc, 0, 0, 0);
        a_c = a_n;
        b_c = b_n;
      }
    }
    {
      const int j = sg * 32 + li;
      const bool jok = j < Fin;
      float zv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) zv[r] = Pl[(rg * 32 + d_row(r, lane)) * FiP + j];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = rg * 32 + d_row(r, lane);
        float v = acc[r];
        const bool neg = act && (zv[r] <= 0.f);
        slope_acc += neg ? (double)(v * zv[r]) : 0.0;
        v = neg ? v * slope : v;
        cs += v;
        buf_st1(rx, jok ? (unsigned)((rr * lddx + j) * 4) : kOOB, v);
      }
    }

    // ---- dW tile of this wave over the 64 rows of the tile ----
    if (has_tile) {  // wave-uniform
      const float* yp = dYl + lk * KPo + so * 32 + li;
      const float* xp = Pl + lk * FiP + sc * 32 + li;
      float av = yp[0], bv = xp[0];
#pragma unroll 4
      for (int k = 0; k < TM; k += 2) {
        const int kn = (k + 2 < TM) ? k + 2 : k;
        const float an = yp[kn * KPo];
        const float bn = xp[kn * FiP];
        const float bx = act ? gcl::prelu_f(bv, slope) : bv;
        dbacc += av;
        dw = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bx, dw, 0, 0, 0);
        av = an;
        bv = bn;
      }
    }
  }

  constexpr size_t REC = (size_t)FoP * FiP + FoP + FiP;
  float* out = part_dw + (size_t)blockIdx.x * REC;
  if (has_tile) {
#pragma unroll
    for (int r = 0; r < 16; ++r) out[(size_t)(so * 32 + d_row(r, lane)) * FiP + sc * 32 + li] = dw[r];
    if (part_db && sc == 0) {
      const float d = dbacc + __shfl_xor(dbacc, 32, 64);
      if (lane < 32) part_db[(size_t)blockIdx.x * REC + so * 32 + lane] = d;
    }
  }
  __syncthreads();
  float* red = smem;  // [4][32]
  {
    const float v = cs + __shfl_xor(cs, 32, 64);
    if (lane < 32) red[wave * 32 + lane] = v;
  }
  for (int off = 32; off > 0; off >>= 1) slope_acc += __shfl_down(slope_acc, off, 64);
  double* dred = reinterpret_cast<double*>(smem + 4 * 32);
  if (lane == 0) dred[wave] = slope_acc;
  __syncthreads();
  if (part_cs && tid < FiP) {
    const int s2 = tid >> 5, c = tid & 31;  // column slab s2 is held by waves 2*s2 (rg 0) and 2*s2+1 (rg 1)
    part_cs[(size_t)blockIdx.x * REC + tid] = red[(2 * s2) * 32 + c] + red[(2 * s2 + 1) * 32 + c];
  }
  if (part_slope && tid == 0) part_slope[blockIdx.x] = (dred[0] + dred[1]) + (dred[2] + dred[3]);
}

// Wt[c][o] = W[o*ldw + c]  (o < R, c < Cn): lets the dX contraction read its weights k-contiguously
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ W, int64_t ldw, float* __restrict__ Wt,
                                                        int32_t R, int32_t Cn) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int o0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  for (int i = ty; i < 32; i += 8) {
    const int o = o0 + i, c = c0 + tx;
    tile[i][tx] = (o < R && c < Cn) ? W[(int64_t)o * ldw + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, o = o0 + tx;
    if (c < Cn && o < R) Wt[(int64_t)c * R + o] = tile[tx][i];
  }
}

__global__ void reduce_scalar_kernel(const double* __restrict__ part, int32_t nparts, float* __restrict__ out) {
  // single wave; fixed order => deterministic
  double s = 0.0;
  for (int p = threadIdx.x; p < nparts; p += 64) s += part[p];
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (threadIdx.x == 0) *out += (float)s;
}

// Up to 3 output segments reduced in ONE launch: segment g covers `count[g]` consecutive floats of
// every partial record starting at offset `poff[g]`; element e of it goes to out[g][(e / pld) * ldo
// + e % pld] when (e % pld) < cols (partial tiles are padded to pld columns).  16 threads split
// the partial records of one element, combined through LDS in a fixed order.
struct RedSeg {
  float* out;
  int32_t poff, count, pld, cols, ldo;
  int32_t acc;  // add into out (1) or overwrite it (0): every destination has its own flag
};
__global__ __launch_bounds__(256) void reduce_multi_kernel(const float* __restrict__ part, int32_t nparts,
                                                           int64_t pstride, RedSeg s0, RedSeg s1, RedSeg s2,
                                                           const double* __restrict__ spart, int32_t ns,
                                                           float* __restrict__ sout) {
  if (spart && blockIdx.x == gridDim.x - 1) {  // one extra block: the scalar (PReLU slope) partials, fixed order
    if (threadIdx.x < 64) {
      double t = 0.0;
      for (int p = threadIdx.x; p < ns; p += 64) t += spart[p];
      for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
      if (threadIdx.x == 0) *sout += (float)t;
    }
    return;
  }
  __shared__ float red[16][17];
  const int e16 = threadIdx.x & 15, g = threadIdx.x >> 4;
  int idx = blockIdx.x * 16 + e16;
  RedSeg sg = s0;
  if (idx >= s0.count) {
    idx -= s0.count;
    sg = s1;
    if (idx >= s1.count) {
      idx -= s1.count;
      sg = s2;
    }
  }
  const bool ok = idx < sg.count;
  float s = 0.f;
  if (ok)
    for (int p = g; p < nparts; p += 16) s += part[(size_t)p * pstride + sg.poff + idx];
  red[g][e16] = s;
  __syncthreads();
  if (g == 0 && ok) {
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) tot += red[q][e16];
    const int i = idx / sg.pld, j = idx - i * sg.pld;
    if (j < sg.cols) {
      float* o = sg.out + (size_t)i * sg.ldo + j;
      *o = sg.acc ? *o + tot : tot;
    }
  }
}

// Same contract, 16-byte loads: a block reduces 32 consecutive floats of the record (8 lanes x float4 =
// one 128-B line per partial record) with 32 groups of lanes striding over the records, then a
// fixed-order combine through LDS.  Needs pstride, every poff and every count to be multiples of 4.
__device__ __forceinline__ void reduce4_block(const float* __restrict__ part, int32_t nparts, int64_t pstride,
                                              const RedSeg& s0, const RedSeg& s1, const RedSeg& s2, int blk);

__global__ __launch_bounds__(256) void reduce_multi4_kernel(const float* __restrict__ part, int32_t nparts,
                                                            int64_t pstride, RedSeg s0, RedSeg s1, RedSeg s2,
                                                            const double* __restrict__ spart, int32_t ns,
                                                            float* __restrict__ sout) {
  if (spart && blockIdx.x == gridDim.x - 1) {  // one extra block: the scalar (PReLU slope) partials, fixed order
    if (threadIdx.x < 64) {
      double t = 0.0;
      for (int p = threadIdx.x; p < ns; p += 64) t += spart[p];
      for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
      if (threadIdx.x == 0) *sout += (float)t;
    }
    return;
  }
  reduce4_block(part, nparts, pstride, s0, s1, s2, blockIdx.x);
}

__device__ __forceinline__ void reduce4_block(const float* __restrict__ part, int32_t nparts, int64_t pstride,
                                              const RedSeg& s0, const RedSeg& s1, const RedSeg& s2, int blk) {
  __shared__ float red[32][33];
  const int l8 = threadIdx.x & 7, g = threadIdx.x >> 3;
  int idx = blk * 32 + l8 * 4;  // first of this lane's 4 elements, in the concatenated segments
  RedSeg sg = s0;
  if (idx >= s0.count) {
    idx -= s0.count;
    sg = s1;
    if (idx >= s1.count) {
      idx -= s1.count;
      sg = s2;
    }
  }
  const bool ok = idx < sg.count;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ok) {
    const float* base = part + sg.poff + idx;
    int p = g;
    for (; p + 96 < nparts; p += 128) {  // four records in flight per lane
      const float4 a = *reinterpret_cast<const float4*>(base + (size_t)p * pstride);
      const float4 b = *reinterpret_cast<const float4*>(base + (size_t)(p + 32) * pstride);
      const float4 c = *reinterpret_cast<const float4*>(base + (size_t)(p + 64) * pstride);
      const float4 d = *reinterpret_cast<const float4*>(base + (size_t)(p + 96) * pstride);
      s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
      s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
      s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w;
      s.x += d.x; s.y += d.y; s.z += d.z; s.w += d.w;
    }
    for (; p < nparts; p += 32) {
      const float4 a = *reinterpret_cast<const float4*>(base + (size_t)p * pstride);
      s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    }
  }
  red[g][l8 * 4] = s.x; red[g][l8 * 4 + 1] = s.y; red[g][l8 * 4 + 2] = s.z; red[g][l8 * 4 + 3] = s.w;
  __syncthreads();
  if (threadIdx.x < 32) {
    // element threadIdx.x of the block: recompute its segment position (its lane group's idx + offset)
    int e = blk * 32 + threadIdx.x;
    RedSeg so = s0;
    if (e >= s0.count) {
      e -= s0.count;
      so = s1;
      if (e >= s1.count) {
        e -= s1.count;
        so = s2;
      }
    }
    if (e < so.count) {
      float tot = 0.f;
#pragma unroll
      for (int q = 0; q < 32; ++q) tot += red[q][threadIdx.x];
      const int i = e / so.pld, j = e - i * so.pld;
      if (j < so.cols) {
        float* o = so.out + (size_t)i * so.ldo + j;
        *o = so.acc ? *o + tot : tot;
      }
    }
  }
}

// picks the 16-byte variant when the record layout allows it
inline void launch_reduce_multi(const float* part, int nparts, int64_t pstride, const RedSeg& s0, const RedSeg& s1,
                                const RedSeg& s2, hipStream_t st, const double* spart = nullptr,
                                int ns = 0, float* sout = nullptr) {
  const int total = s0.count + s1.count + s2.count;
  const bool vec = (pstride % 4 == 0) && ((s0.poff | s1.poff | s2.poff | s0.count | s1.count | s2.count) % 4 == 0) &&
                   gcl::aligned16(part) && nparts >= 64;
  const unsigned extra = (spart && sout && ns > 0) ? 1u : 0u;  // one more block for the scalar partials
  const double* sp = extra ? spart : nullptr;
  if (vec)
    hipLaunchKernelGGL(reduce_multi4_kernel, dim3((unsigned)gcl::cdiv(total, 32) + extra), dim3(256), 0, st, part, nparts,
                       pstride, s0, s1, s2, sp, ns, sout);
  else
    hipLaunchKernelGGL(reduce_multi_kernel, dim3((unsigned)gcl::cdiv(total, 16) + extra), dim3(256), 0, st, part, nparts,
                       pstride, s0, s1, s2, sp, ns, sout);
}

// Final pass of up to 16 fused backward calls in ONE launch (gcl_reduce_jobs): block -> (job, block of the job).
// The scalar (PReLU slope) partials of ALL jobs are summed by the last block, job after job, so jobs that share a
// slope parameter add to it in a fixed order.
constexpr int kJobsPerLaunch = 16;
struct RedJobK {
  const float* part;
  int64_t pstride;
  RedSeg s0, s1, s2;
  const double* spart;
  float* sout;
  int32_t nparts, ns, blk0;  // blk0: first block of this job in the launch
};
struct RedJobsK {
  RedJobK j[kJobsPerLaunch];
  int32_t n, nblk;
};
__global__ __launch_bounds__(256) void reduce_jobs_kernel(RedJobsK J) {
  const int blk = blockIdx.x;
  if (blk == J.nblk) {  // the extra block: scalar partials, in job order
    if (threadIdx.x < 64) {
      for (int q = 0; q < J.n; ++q) {
        if (!J.j[q].spart) continue;
        double t = 0.0;
        for (int p = threadIdx.x; p < J.j[q].ns; p += 64) t += J.j[q].spart[p];
        for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
        if (threadIdx.x == 0) *J.j[q].sout += (float)t;
      }
    }
    return;
  }
  int q = 0;
  while (q + 1 < J.n && blk >= J.j[q + 1].blk0) ++q;  // uniform
  const RedJobK& jb = J.j[q];
  reduce4_block(jb.part, jb.nparts, jb.pstride, jb.s0, jb.s1, jb.s2, blk - jb.blk0);
}

}  // namespace

namespace gcl {
int launch_reduce_parts(const float* part, int nparts, int64_t pstride, int pld, float* out, int ldo, int R, int C,
                        int accumulate, hipStream_t st) {
  // one segment of the 16-way reducer: R rows of pld (padded) columns, C of them valid
  RedSeg s0{out, 0, R * pld, pld, C, ldo, accumulate};
  RedSeg none{nullptr, 0, 0, 1, 0, 0, 0};
  launch_reduce_multi(part, nparts, pstride, s0, none, none, st);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}
int launch_reduce_parts2(const float* part, int nparts, int64_t pstride, int pld, int off1, float* out0, float* out1,
                         int C, int accumulate, hipStream_t st) {
  RedSeg s0{out0, 0, pld, pld, C, C, accumulate};
  RedSeg s1{out1, off1, pld, pld, C, C, accumulate};
  RedSeg none{nullptr, 0, 0, 1, 0, 0, 0};
  launch_reduce_multi(part, nparts, pstride, s0, s1, none, st);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}
// three vectors of one partial record (each padded to pld entries, back to back), each with its own accumulate flag
int launch_reduce_parts3(const float* part, int nparts, int64_t pstride, int pld, float* out0, int acc0, float* out1,
                         int acc1, float* out2, int acc2, int C, hipStream_t st) {
  RedSeg s0{out0, 0, pld, pld, C, C, acc0};
  RedSeg s1{out1, pld, pld, pld, C, C, acc1};
  RedSeg s2{out2, 2 * pld, pld, pld, C, C, acc2};
  launch_reduce_multi(part, nparts, pstride, s0, s1, s2, st);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}
}  // namespace gcl

namespace {

int gcl_ablate() {  // GCL_ABLATE: timing-only experiments (tools/ablate.sh) - honoured by the diagnostic build only
#ifndef GCL_STAMPS
  return 0;  // the product library never turns stores / MFMA / loads off, whatever the environment says
#endif
  static const int v = [] { const char* e = getenv("GCL_ABLATE"); return e ? atoi(e) : 0; }();
  return v;
}

bool use_valu() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("GCL_LINEAR_IMPL");
    v = (e && strcmp(e, "valu") == 0) ? 1 : 0;
  }
  return v == 1;
}

constexpr int kDwBlocks = 512;

struct LinGeom {
  int NS;
  int waves;
  int KT;
  size_t lds;
  int grid;
};

int lin_geom(int64_t rows, int K, int N, bool vec_x, LinGeom* g) {
  GCL_CHECK_ARG(K >= 1 && K <= 256 && N >= 1 && N <= 256, "linear: unsupported K=%d N=%d (K<=256, N<=256)", K, N);
  GCL_CHECK_ARG(vec_x || K <= 128, "linear: K=%d > 128 needs 16-B aligned rows (ld %% 4 == 0)", K);
  g->NS = (N + 31) / 32;
  if (g->NS == 3) g->NS = 4;
  if (g->NS > 4) g->NS = 8;  // instantiated: 1,2,4,8
  g->KT = K <= 64 ? 64 : K <= 128 ? 128 : 256;
  const int KP = ((K + 3) & ~3) + 2;
  g->waves = 4;
  g->lds = ((size_t)g->NS * 32 + 128) * KP * sizeof(float) + 64;
  if (g->lds > 160 * 1024) {  // wide panels: 64-row tiles (2 waves) keep the weight panel resident
    g->waves = 2;
    g->lds = ((size_t)g->NS * 32 + 64) * KP * sizeof(float) + 64;
  }
  GCL_CHECK_ARG(g->lds <= 160 * 1024, "linear: K=%d N=%d needs %zu B of LDS (>160 KiB)", K, N, g->lds);
  // persistent grid = what is resident at once (LDS- and register-limited), so no partial last wave
  int bpc = (int)((160 * 1024) / g->lds);
  const int by_regs = lin_min_waves(g->NS, g->KT) * 4 / g->waves;
  if (bpc > by_regs) bpc = by_regs;
  if (bpc < 1) bpc = 1;
  const int64_t ntiles = gcl::cdiv(rows, 32 * g->waves);
  const int64_t cap = (int64_t)gcl::kNumCU * bpc;
  g->grid = (int)(ntiles < cap ? ntiles : cap);
  return GCL_OK;
}

template <int EPI>
int launch_linear(const float* X, int64_t ldx, const float* in_slope, const float* W, int ldw, int trans,
                  const float* bias, float* Y, int64_t ldy, int64_t rows, int K, int N, const float* Z, int64_t ldz,
                  const float* z_slope, double* slope_part, int* nparts, hipStream_t st, int akind) {
  if (rows == 0) {
    if (nparts) *nparts = 0;
    return GCL_OK;
  }
  if (use_valu()) {
    const int64_t total = rows * N;
    int grid = (int)(gcl::cdiv(total, 256) < 4096 ? gcl::cdiv(total, 256) : 4096);
    hipLaunchKernelGGL((linear_valu_kernel<EPI>), dim3(grid), dim3(256), 0, st, X, ldx, in_slope, W, ldw, trans,
                       bias, Y, ldy, rows, K, N, Z, ldz, z_slope, slope_part, akind);
    GCL_CHECK_LAUNCH();
    if (nparts) *nparts = grid;
    return GCL_OK;
  }
  const int vec_x = (K % 4 == 0) && (ldx % 4 == 0) && gcl::aligned16(X);
  LinGeom g;
  int rc = lin_geom(rows, K, N, vec_x != 0, &g);
  if (rc) return rc;
#define GCL_LIN3(NS_, KT_, V_)                                                                                    \
  do {                                                                                                            \
    auto kern = linear_mfma_kernel<NS_, EPI, KT_, V_>;                                                            \
    { const int lrc_ = gcl::ensure_dyn_lds((const void*)kern, 160 * 1024); if (lrc_) return lrc_; } \
    hipLaunchKernelGGL(kern, dim3(g.grid), dim3(g.waves * 64), g.lds, st, X, ldx, in_slope, W, ldw, trans, bias,  \
                       Y, ldy, rows, K, N, Z, ldz, z_slope, slope_part, akind, gcl_ablate());                     \
  } while (0)
#define GCL_LIN2(NS_, KT_)                 \
  do {                                     \
    if (vec_x) GCL_LIN3(NS_, KT_, true);   \
    else GCL_LIN3(NS_, KT_, false);        \
  } while (0)
#define GCL_LIN(NS_)                                \
  do {                                              \
    if (g.KT == 64) GCL_LIN2(NS_, 64);              \
    else if (g.KT == 128) GCL_LIN2(NS_, 128);       \
    else GCL_LIN3(NS_, 256, true);                  \
  } while (0)
  switch (g.NS) {
    case 1: GCL_LIN(1); break;
    case 2: GCL_LIN(2); break;
    case 4: GCL_LIN(4); break;
    default: GCL_LIN(8); break;
  }
#undef GCL_LIN3
#undef GCL_LIN
#undef GCL_LIN2
  GCL_CHECK_LAUNCH();
  if (nparts) *nparts = g.grid;
  return GCL_OK;
}

// does the resident-panel kernel take this shape?
// K: contraction length, N: output width.  Measured on MI355X (tools/gemm_bench.py): the 128x128
// tile kernel wins once the contraction is long or the output fills its 128 columns.
bool panel_fits(int K, int N, bool vec_x, bool trans, bool w_ok) {
  static const int impl = [] {
    const char* e = getenv("GCL_DENSE_IMPL");
    return !e ? 0 : strcmp(e, "tile") == 0 ? 1 : strcmp(e, "panel") == 0 ? 2 : 0;
  }();
  const bool tile_ok = vec_x && w_ok && K % 4 == 0 && N % 4 == 0;
  if (impl == 1 && tile_ok) return false;
  if (impl == 0 && tile_ok && (K > 128 || (N >= 128 && (K >= 128 || trans)))) return false;
  if (K < 1 || K > 256 || N < 1 || N > 256) return false;
  if (!vec_x && K > 128) return false;
  int NS = (N + 31) / 32;
  if (NS == 3) NS = 4;
  if (NS > 4) NS = 8;
  const int KP = ((K + 3) & ~3) + 2;
  return ((size_t)NS * 32 + 64) * KP * sizeof(float) + 64 <= 160 * 1024;
}

template <int EPI>
int launch_gemm(const float* X, int64_t ldx, int akind, const float* slope, const float* W, int64_t ldw, int trans,
                const float* bias, float* Y, int64_t ldy, int64_t rows, int K, int N, const float* Z, int64_t ldz,
                const float* add, int64_t ldadd, double* slope_part, int* nparts, hipStream_t st) {
  if (nparts) *nparts = 0;
  if (rows == 0) return GCL_OK;
  GCL_CHECK_ARG((K % 4 == 0) && (ldx % 4 == 0) && gcl::aligned16(X), "dense: wide shapes need K %% 4 == 0 and 16-B aligned rows (K=%d ldx=%lld)", K, (long long)ldx);
  GCL_CHECK_ARG((ldw % 4 == 0) && gcl::aligned16(W) && (!trans || N % 4 == 0), "dense: wide shapes need a 16-B aligned weight block (ldw=%lld N=%d)", (long long)ldw, N);
  GtGeom g = gt_geom(rows, N);
  // split-operand bf16 variant (gemm_tile_x3_kernel): non-transposed weights, 128-row tiles
  static const int x3_tile = [] { const char* e = getenv("GCL_X3"); const char* f = getenv("GCL_X3_TILE");
                                  return ((e && atoi(e) == 0) || (f && atoi(f) == 0)) ? 0 : 1; }();
  if (x3_tile && !trans && g.mi == 2) {
    auto kern = gemm_tile_x3_kernel<EPI>;
    { const int lrc_ = gcl::ensure_dyn_lds((const void*)kern, 160 * 1024); if (lrc_) return lrc_; }
    hipLaunchKernelGGL(kern, dim3(g.grid), dim3(256), gtx3_lds(), st, X, ldx, akind, slope, W, ldw, bias, Y, ldy, rows, K, N,
                       Z, ldz, add, ldadd, slope_part, g.nt, g.total, g.per_xcd);
    GCL_CHECK_LAUNCH();
    if (nparts) *nparts = (int)g.grid;
    return GCL_OK;
  }
#define GCL_GT(T_, MI_)                                                                                           \
  do {                                                                                                            \
    auto kern = gemm_tile_kernel<EPI, T_, MI_>;                                                                   \
    { const int lrc_ = gcl::ensure_dyn_lds((const void*)kern, 160 * 1024); if (lrc_) return lrc_; } \
    hipLaunchKernelGGL(kern, dim3(g.grid), dim3(256), gt_lds(MI_), st, X, ldx, akind, slope, W, ldw, bias, Y, ldy, \
                       rows, K, N, Z, ldz, add, ldadd, slope_part, g.nt, g.total, g.per_xcd);                     \
  } while (0)
  if (trans) {
    if (g.mi == 1) GCL_GT(true, 1);
    else GCL_GT(true, 2);
  } else {
    if (g.mi == 1) GCL_GT(false, 1);
    else GCL_GT(false, 2);
  }
#undef GCL_GT
  GCL_CHECK_LAUNCH();
  if (nparts) *nparts = (int)g.grid;
  return GCL_OK;
}

}  // namespace

extern "C" size_t gcl_colsum_ws_bytes(int64_t, int32_t);
extern "C" int gcl_colsum(const float*, int64_t, int64_t, int32_t, float*, int32_t, void*, size_t, gcl_stream_t);

static int check_act(const char* who, int act, const float* slope) {
  GCL_CHECK_ARG(act == GCL_ACT_NONE || act == GCL_ACT_PRELU || act == GCL_ACT_SILU, "%s: unknown activation %d", who, act);
  GCL_CHECK_ARG(act != GCL_ACT_PRELU || slope, "%s: GCL_ACT_PRELU needs the slope pointer", who);
  return GCL_OK;
}

extern "C" int gcl_dense_fwd(const float* x, int64_t ldx, int32_t act, const float* slope, const float* W, int64_t ldw,
                             const float* bias, const float* addend, int64_t ldadd, float* y, int64_t ldy,
                             int64_t rows, int32_t Fin, int32_t Fout, gcl_stream_t stream) {
  GCL_CHECK_ARG(x && W && y, "dense_fwd: null argument");
  GCL_CHECK_ARG(rows >= 0 && Fin >= 1 && Fout >= 1 && ldx >= Fin && ldy >= Fout && ldw >= Fin && (!addend || ldadd >= Fout),
                "dense_fwd: bad shape rows=%lld ldx=%lld ldy=%lld ldw=%lld", (long long)rows, (long long)ldx,
                (long long)ldy, (long long)ldw);
  if (int rc = check_act("dense_fwd", act, slope)) return rc;
  const bool vec_x = (Fin % 4 == 0) && (ldx % 4 == 0) && gcl::aligned16(x);
  if (!addend && ldw == Fin && !use_valu() && gcl::x3_linear_fwd_applicable(x, ldx, y, ldy, Fin, Fout, act))
    return gcl::x3_linear_fwd(x, ldx, act, act == GCL_ACT_PRELU ? slope : nullptr, W, bias, y, ldy, rows, Fin, Fout,
                              (hipStream_t)stream);
  if (!addend && ldw == Fin && (use_valu() || panel_fits(Fin, Fout, vec_x, false, (ldw % 4 == 0) && gcl::aligned16(W))))
    return launch_linear<EPI_BIAS>(x, ldx, act == GCL_ACT_PRELU ? slope : nullptr, W, Fin, 0, bias, y, ldy, rows, Fin,
                                   Fout, nullptr, 0, nullptr, nullptr, nullptr, (hipStream_t)stream, act);
  return launch_gemm<EPI_BIAS>(x, ldx, act, act == GCL_ACT_PRELU ? slope : nullptr, W, ldw, 0, bias, y, ldy, rows, Fin,
                               Fout, nullptr, 0, addend, ldadd, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int gcl_linear_fwd(const float* x, int64_t ldx, const float* in_slope, const float* W, const float* bias,
                              float* y, int64_t ldy, int64_t rows, int32_t Fin, int32_t Fout, gcl_stream_t stream) {
  return gcl_dense_fwd(x, ldx, in_slope ? GCL_ACT_PRELU : GCL_ACT_NONE, in_slope, W, Fin, bias, nullptr, 0, y, ldy, rows,
                       Fin, Fout, stream);
}

// slope partials: one double per block of the dx kernel (the tiled kernel launches a block per tile)
static size_t slope_parts_cap(int64_t rows, int32_t Fin) {
  const int64_t g = (int64_t)gt_geom(rows, Fin).grid;
  return (size_t)(g > 4096 ? g : 4096);
}

extern "C" size_t gcl_linear_bwd_ws_bytes(int64_t rows, int32_t Fin, int32_t Fout) {
  size_t nc = (size_t)(Fin + 31) / 32;
  if (nc > 4) nc = 4;  // wider inputs are walked in 128-column chunks
  const size_t FinP = nc * 32;
  size_t no = (size_t)(Fout + 31) / 32;
  no = no <= 2 ? no : no <= 4 ? 4 : 8;
  const size_t FoutP = no * 32;
  const size_t dw = (size_t)kDwBlocks * FoutP * (FinP + 1) * sizeof(float);
  const size_t sl = slope_parts_cap(rows, Fin) * sizeof(double);
  return dw + sl;
}

extern "C" int gcl_dense_bwd_dx(const float* dy, int64_t lddy, const float* W, int64_t ldw, const float* z, int64_t ldz,
                                int32_t act, const float* slope, float* d_slope, const float* addend, int64_t ldadd,
                                float* dx, int64_t lddx, int64_t rows, int32_t Fin, int32_t Fout, void* ws,
                                size_t ws_bytes, gcl_stream_t stream) {
  GCL_CHECK_ARG(dy && W && dx, "dense_bwd_dx: null argument");
  GCL_CHECK_ARG(lddy >= Fout && lddx >= Fin && ldw >= Fin && (!addend || ldadd >= Fin), "dense_bwd_dx: leading dimension too small");
  if (int rc = check_act("dense_bwd_dx", act, slope)) return rc;
  GCL_CHECK_ARG(act == GCL_ACT_NONE || (z && ldz >= Fin), "dense_bwd_dx: activation given without its pre-activation z");
  hipStream_t st = (hipStream_t)stream;
  const float* sl = act == GCL_ACT_PRELU ? slope : nullptr;
  double* slope_part = nullptr;
  if (sl && d_slope) {
    GCL_CHECK_ARG(ws && ws_bytes >= slope_parts_cap(rows, Fin) * sizeof(double) && gcl::aligned16(ws), "dense_bwd_dx: workspace too small");
    slope_part = (double*)ws;
  }
  int nparts = 0, rc;
  const float* zz = act == GCL_ACT_NONE ? nullptr : z;
  const bool vec_x = (Fout % 4 == 0) && (lddy % 4 == 0) && gcl::aligned16(dy);
  // contraction over Fout: "weights" are W^T, i.e. Wl[j=c][k=o] = W[o*ldw + c]
  if (!addend && ldw == Fin && (use_valu() || panel_fits(Fout, Fin, vec_x, true, (ldw % 4 == 0) && gcl::aligned16(W))))
    rc = launch_linear<EPI_DX>(dy, lddy, nullptr, W, Fin, 1, nullptr, dx, lddx, rows, Fout, Fin, zz, ldz, sl, slope_part,
                               &nparts, st, act);
  else {
    // the tile kernel stages k-contiguous weight rows twice as cheaply as strided ones: transpose W
    // (a few hundred KB) into the workspace first when there is room
    const size_t off = (slope_parts_cap(rows, Fin) * sizeof(double) + 255) & ~(size_t)255;
    const size_t need = off + (size_t)Fin * Fout * sizeof(float);
    if (ws && ws_bytes >= need && gcl::aligned16(ws) && Fout % 4 == 0 && rows >= 4096) {
      float* Wt = reinterpret_cast<float*>(static_cast<char*>(ws) + off);
      hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)gcl::cdiv(Fin, 32), (unsigned)gcl::cdiv(Fout, 32)), dim3(256), 0,
                         st, W, ldw, Wt, Fout, Fin);
      rc = launch_gemm<EPI_DX>(dy, lddy, act, sl, Wt, Fout, 0, nullptr, dx, lddx, rows, Fout, Fin, zz, ldz, addend,
                               ldadd, slope_part, &nparts, st);
    } else {
      rc = launch_gemm<EPI_DX>(dy, lddy, act, sl, W, ldw, 1, nullptr, dx, lddx, rows, Fout, Fin, zz, ldz, addend, ldadd,
                               slope_part, &nparts, st);
    }
  }
  if (rc) return rc;
  if (slope_part && nparts > 0) {
    hipLaunchKernelGGL(reduce_scalar_kernel, dim3(1), dim3(64), 0, st, slope_part, nparts, d_slope);
    GCL_CHECK_LAUNCH();
  }
  return GCL_OK;
}

extern "C" int gcl_linear_bwd_dx(const float* dy, int64_t lddy, const float* W, const float* x, int64_t ldx,
                                 const float* in_slope, float* d_in_slope, float* dx, int64_t lddx, int64_t rows,
                                 int32_t Fin, int32_t Fout, void* ws, size_t ws_bytes, gcl_stream_t stream) {
  GCL_CHECK_ARG(!in_slope || (x && ldx >= Fin), "linear_bwd_dx: in_slope given without the forward input x");
  return gcl_dense_bwd_dx(dy, lddy, W, Fin, in_slope ? x : nullptr, ldx, in_slope ? GCL_ACT_PRELU : GCL_ACT_NONE,
                          in_slope, d_in_slope, nullptr, 0, dx, lddx, rows, Fin, Fout, ws, ws_bytes, stream);
}

// dW for <= 256 outputs and any number of inputs: ONE launch whose grid.y walks the 128-column input
// chunks (dY is re-read once per chunk), then one partial reduction per chunk
static int dw_block(const float* dy, int64_t lddy, const float* x, int64_t ldx, int akind, const float* in_slope,
                    float* dW, int64_t lddw, float* db, int64_t rows, int32_t Fin, int32_t Fout, int32_t accumulate,
                    void* ws, hipStream_t st) {
  const int nct = (Fin + 127) / 128;                      // input chunks
  const int NC = nct > 1 ? 4 : (Fin + 31) / 32;           // slabs of the (widest) chunk
  int NO = (Fout + 31) / 32;
  NO = NO <= 2 ? NO : NO <= 4 ? 4 : 8;  // instantiated: 1, 2, 4, 8 (extra slabs are zero)
  const int FinP = NC * 32, FoutP = NO * 32;
  const size_t lds = (size_t)kDwRT * (FoutP + FinP) * sizeof(float);
  // one partial tile per block: no more blocks than are resident at once (wide tiles fill the LDS
  // with one block per CU, and their partials are what the reduction then has to read back), and
  // at least four 64-row steps per block so the load pipeline has something to hide under
  const int64_t cap = (lds > 80 * 1024 ? gcl::kNumCU : kDwBlocks) / nct;
  int64_t nblk = gcl::cdiv(rows, (nct > 1 ? 4 : 2) * kDwRT);
  if (nblk > cap) nblk = cap;
  if (nblk < 1) nblk = 1;
  int64_t rpb = gcl::cdiv(rows, nblk);
  rpb = gcl::cdiv(rpb, kDwRT) * kDwRT;
  nblk = rows > 0 ? gcl::cdiv(rows, rpb) : 1;
  float* part = (float*)ws;
  float* dbpart = part + (size_t)kDwBlocks * FoutP * FinP;
  const int64_t tile_stride = (int64_t)nblk * FoutP * FinP;
  const bool vec = (lddy % 4 == 0) && (ldx % 4 == 0) && gcl::aligned16(dy) && gcl::aligned16(x);
#define GCL_DW3(NO_, NC_, V_)                                                                                     \
  do {                                                                                                            \
    auto kern = dw_mfma_kernel<NO_, NC_, V_>;                                                                     \
    { const int lrc_ = gcl::ensure_dyn_lds((const void*)kern, 160 * 1024); if (lrc_) return lrc_; } \
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk, (unsigned)nct), dim3(256), lds, st, dy, lddy, x, ldx, in_slope, part, \
                       db ? dbpart : nullptr, rows, Fin, Fout, rpb, akind, tile_stride);                          \
  } while (0)
#define GCL_DW2(NO_, NC_)              \
  do {                                 \
    if (vec) GCL_DW3(NO_, NC_, true);  \
    else GCL_DW3(NO_, NC_, false);     \
  } while (0)
#define GCL_DW(NO_)                    \
  do {                                 \
    switch (NC) {                      \
      case 1: GCL_DW2(NO_, 1); break;  \
      case 2: GCL_DW2(NO_, 2); break;  \
      case 3: GCL_DW2(NO_, 3); break;  \
      default: GCL_DW2(NO_, 4); break; \
    }                                  \
  } while (0)
  switch (NO) {
    case 1: GCL_DW(1); break;
    case 2: GCL_DW(2); break;
    case 4: GCL_DW(4); break;
    default: GCL_DW(8); break;
  }
#undef GCL_DW
#undef GCL_DW2
#undef GCL_DW3
  GCL_CHECK_LAUNCH();
  // final pass: the input chunks of a wide layer three to a launch (one segment each: chunk ct's records start
  // ct * tile_stride floats into the workspace) - a 256-wide layer used to end in two of these launches per call
  for (int ct = 0; ct < nct; ct += 3) {
    RedSeg seg[3] = {{nullptr, 0, 0, 1, 0, 0, 0}, {nullptr, 0, 0, 1, 0, 0, 0}, {nullptr, 0, 0, 1, 0, 0, 0}};
    for (int q = 0; q < 3 && ct + q < nct; ++q) {
      const int c = ct + q;
      const int fi = Fin - c * 128 < 128 ? Fin - c * 128 : 128;
      GCL_CHECK_ARG((int64_t)c * tile_stride < ((int64_t)1 << 31), "dense_bwd_dw: partial workspace offset overflows");
      seg[q] = RedSeg{dW + c * 128, (int32_t)(c * tile_stride), Fout * FinP, FinP, fi, (int32_t)lddw, accumulate};
    }
    launch_reduce_multi(part, (int)nblk, (int64_t)FoutP * FinP, seg[0], seg[1], seg[2], st);
    GCL_CHECK_LAUNCH();
  }
  if (db) return gcl::launch_reduce_parts(dbpart, (int)nblk, (int64_t)FoutP, FoutP, db, Fout, 1, Fout, accumulate, st);
  return GCL_OK;
}

extern "C" int gcl_dense_bwd_dw(const float* dy, int64_t lddy, const float* x, int64_t ldx, int32_t act,
                                const float* slope, float* dW, int64_t lddw, float* db, int64_t rows, int32_t Fin,
                                int32_t Fout, int32_t accumulate, void* ws, size_t ws_bytes, gcl_stream_t stream) {
  GCL_CHECK_ARG(dy && x && dW, "dense_bwd_dw: null argument");
  GCL_CHECK_ARG(lddy >= Fout && ldx >= Fin && lddw >= Fin, "dense_bwd_dw: leading dimension too small");
  GCL_CHECK_ARG(Fin >= 1 && Fout >= 1, "dense_bwd_dw: bad shape");
  GCL_CHECK_ARG(lddw < (1 << 30), "dense_bwd_dw: lddw too large");
  if (int rc = check_act("dense_bwd_dw", act, slope)) return rc;
  GCL_CHECK_ARG(ws && ws_bytes >= gcl_linear_bwd_ws_bytes(rows, Fin, Fout), "dense_bwd_dw: workspace too small");
  const float* sl = act == GCL_ACT_PRELU ? slope : nullptr;
  // wide layers: <= 256 outputs per launch, every 128-column input chunk inside it (grid.y)
  for (int o0 = 0; o0 < Fout; o0 += 256) {
    const int fo = Fout - o0 < 256 ? Fout - o0 : 256;
    int rc = dw_block(dy + o0, lddy, x, ldx, act, sl, dW + (int64_t)o0 * lddw, lddw, db ? db + o0 : nullptr, rows, Fin,
                      fo, accumulate, ws, (hipStream_t)stream);
    if (rc) return rc;
  }
  return GCL_OK;
}

extern "C" int gcl_linear_bwd_dw(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* in_slope,
                                 float* dW, float* db, int64_t rows, int32_t Fin, int32_t Fout, int32_t accumulate,
                                 void* ws, size_t ws_bytes, gcl_stream_t stream) {
  return gcl_dense_bwd_dw(dy, lddy, x, ldx, in_slope ? GCL_ACT_PRELU : GCL_ACT_NONE, in_slope, dW, Fin, db, rows, Fin,
                          Fout, accumulate, ws, ws_bytes, stream);
}

// Fused path geometry / workspace (see linear_bwd_fused_kernel)
static bool fused_ok(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* dx, int64_t lddx,
                     int32_t Fin, int32_t Fout) {
  static int off = -1;
  if (off < 0) {
    const char* e = getenv("GCL_NO_FUSED_BWD");
    off = (e && atoi(e) != 0) ? 1 : 0;
  }
  if (off || use_valu()) return false;
  // Fout need not be a multiple of 4 when the rows of dy are padded to one (lddy >= roundup(Fout, 4)): the 16-byte
  // loads then also fetch the padding columns, which meet zero weight rows (they must hold finite values)
  return Fout <= 64 && Fin <= 96 && (lddy >= ((Fout + 3) & ~3)) && (Fin % 4 == 0) && (lddy % 4 == 0) && (ldx % 4 == 0) &&
         (lddx >= Fin) && gcl::aligned16(dy) && gcl::aligned16(x) && dx != nullptr;
}

extern "C" size_t gcl_linear_bwd_all_ws_bytes(int64_t rows, int32_t Fin, int32_t Fout) {
  const size_t FiP = (size_t)((Fin + 31) / 32) * 32, FoP = (size_t)((Fout + 31) / 32) * 32;
  const size_t fused = (size_t)3 * gcl::kNumCU * (FoP * FiP + FoP + FiP) * sizeof(float) + (size_t)3 * gcl::kNumCU * 8 + 64;
  size_t sep = gcl_linear_bwd_ws_bytes(rows, Fin, Fout);
  const size_t cs = gcl_colsum_ws_bytes(rows, Fin);
  if (cs > sep) sep = cs;
  return fused > sep ? fused : sep;
}

static int bwd_all_impl(const float* dy, int64_t lddy, const float* W, const float* x, int64_t ldx,
                        const float* in_slope, float* d_in_slope, float* dx, int64_t lddx, float* dW,
                        float* db, float* colsum_dx, int64_t rows, int32_t Fin, int32_t Fout,
                        int32_t accumulate, void* ws, size_t ws_bytes, gcl_stream_t stream, gcl_reduce_job* job) {
  if (job) memset(job, 0, sizeof(*job));
  GCL_CHECK_ARG(dy && W && x && dx && dW, "linear_bwd_all: null argument");
  GCL_CHECK_ARG(ws && ws_bytes >= gcl_linear_bwd_all_ws_bytes(rows, Fin, Fout), "linear_bwd_all: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  // every destination has its own accumulate bit: dW, db and colsum_dx usually belong to DIFFERENT
  // parameters (colsum_dx is the bias gradient of the layer below) with different .grad states
  const int acc_dw = (accumulate & GCL_ACC_DW) ? 1 : 0, acc_db = (accumulate & GCL_ACC_DB) ? 1 : 0,
            acc_cs = (accumulate & GCL_ACC_COLSUM) ? 1 : 0;
  if (!fused_ok(dy, lddy, x, ldx, dx, lddx, Fin, Fout) || rows == 0) {
    int rc;
    if (db && acc_db != acc_dw) {  // separate kernels share one flag between dW and db: split the call
      rc = gcl_linear_bwd_dw(dy, lddy, x, ldx, in_slope, dW, nullptr, rows, Fin, Fout, acc_dw, ws, ws_bytes, stream);
      if (rc) return rc;
      rc = gcl_colsum(dy, lddy, rows, Fout, db, acc_db, ws, ws_bytes, stream);
    } else {
      rc = gcl_linear_bwd_dw(dy, lddy, x, ldx, in_slope, dW, db, rows, Fin, Fout, acc_dw, ws, ws_bytes, stream);
    }
    if (rc) return rc;
    rc = gcl_linear_bwd_dx(dy, lddy, W, in_slope ? x : nullptr, ldx, in_slope, d_in_slope, dx, lddx, rows, Fin, Fout,
                           ws, ws_bytes, stream);
    if (rc) return rc;
    if (colsum_dx) rc = gcl_colsum(dx, lddx, rows, Fin, colsum_dx, acc_cs, ws, ws_bytes, stream);
    return rc;  // (nothing deferred: job stays empty)
  }
  const int NO = (Fout + 31) / 32, NC = (Fin + 31) / 32;
  const int FoP = NO * 32, FiP = NC * 32;
  static const int no64 = [] { const char* e = getenv("GCL_NO_FUSED64"); return (e && atoi(e)) ? 1 : 0; }();
  const bool use64 = (NC == 2) && !no64;  // 64-row tiles, 3 blocks per CU
  // split-operand bf16 kernel (x3.h): same partial records, two blocks per CU
  const bool use_x3 = use64 && gcl::x3_linear_bwd_applicable(dy, lddy, x, ldx, dx, lddx, Fin, Fout);
  const int64_t ntiles = gcl::cdiv(rows, use64 ? 64 : 128);
  const int64_t cap = use64 ? 3 * gcl::kNumCU : gcl::kNumCU;
  const int nblk = use_x3 ? gcl::x3_linear_bwd_blocks(rows) : (int)(ntiles < cap ? ntiles : cap);
  const size_t rec_f = (size_t)FoP * FiP + FoP + FiP;  // floats per block record
  float* part_dw = (float*)ws;
  float* part_db = part_dw + (size_t)FoP * FiP;
  float* part_cs = part_db + FoP;
  double* part_sl = (double*)(((uintptr_t)(part_dw + (size_t)3 * gcl::kNumCU * rec_f) + 15) & ~(uintptr_t)15);
  const bool want_slope = in_slope && d_in_slope;
  if (use_x3) {
    if (int rc = gcl::x3_linear_bwd(dy, lddy, W, x, ldx, in_slope, dx, lddx, rows, Fin, Fout, part_dw, db ? part_db : nullptr,
                                    colsum_dx ? part_cs : nullptr, want_slope ? part_sl : nullptr, st))
      return rc;
  } else if (use64) {
    const size_t lds64 = ((size_t)FiP * (FoP + 2) + 64 * (size_t)(FoP + 2) + 64 * (size_t)FiP) * sizeof(float);
#define GCL_FB64(NO_)                                                                                             \
  do {                                                                                                            \
    auto kern = linear_bwd_fused64_kernel<NO_>;                                                                   \
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), lds64, st, dy, lddy, W, x, ldx, in_slope, dx, lddx, rows, Fin, \
                       Fout, part_dw, db ? part_db : nullptr, colsum_dx ? part_cs : nullptr,                      \
                       want_slope ? part_sl : nullptr);                                                           \
  } while (0)
    if (NO == 1) GCL_FB64(1);
    else GCL_FB64(2);
#undef GCL_FB64
  } else {
  const size_t lds = ((size_t)FiP * (FoP + 2) + 128 * (size_t)(FoP + 2) + 128 * (size_t)FiP) * sizeof(float);
#define GCL_FB(NO_, NC_)                                                                                          \
  do {                                                                                                            \
    auto kern = linear_bwd_fused_kernel<NO_, NC_>;                                                                \
    { const int lrc_ = gcl::ensure_dyn_lds((const void*)kern, 160 * 1024); if (lrc_) return lrc_; } \
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), lds, st, dy, lddy, W, x, ldx, in_slope, dx, lddx, rows, Fin,  \
                       Fout, part_dw, db ? part_db : nullptr, colsum_dx ? part_cs : nullptr,                      \
                       want_slope ? part_sl : nullptr);                                                           \
  } while (0)
  if (NO == 1 && NC == 1) GCL_FB(1, 1);
  else if (NO == 1 && NC == 2) GCL_FB(1, 2);
  else if (NO == 1 && NC == 3) GCL_FB(1, 3);
  else if (NO == 2 && NC == 1) GCL_FB(2, 1);
  else if (NO == 2 && NC == 2) GCL_FB(2, 2);
  else GCL_FB(2, 3);
#undef GCL_FB
  }
  GCL_CHECK_LAUNCH();
  {
    // per-block record: [dW tile FoP*FiP | db FoP | colsum FiP]; one launch reduces all three
    const int64_t rec = (int64_t)FoP * FiP + FoP + FiP;
    RedSeg s0{dW, 0, Fout * FiP, FiP, Fin, Fin, acc_dw};
    RedSeg s1{db, FoP * FiP, db ? Fout : 0, FoP, Fout, 0, acc_db};
    RedSeg s2{colsum_dx, FoP * FiP + FoP, colsum_dx ? Fin : 0, FiP, Fin, 0, acc_cs};
    if (job) {
      // deferred: describe the final pass instead of launching it.  db / colsum segments are widened to their padded
      // lengths (cols still limits what is written) so that every job takes the 16-byte reducer.
      s1.count = db ? FoP : 0;
      s2.count = colsum_dx ? FiP : 0;
      job->part = part_dw;
      job->nparts = nblk;
      job->pstride = rec;
      const RedSeg* ss[3] = {&s0, &s1, &s2};
      for (int q = 0; q < 3; ++q) {
        job->seg[q].out = ss[q]->out;
        job->seg[q].poff = ss[q]->poff;
        job->seg[q].count = ss[q]->count;
        job->seg[q].pld = ss[q]->pld;
        job->seg[q].cols = ss[q]->cols;
        job->seg[q].ldo = ss[q]->ldo;
        job->seg[q].acc = ss[q]->acc;
      }
      job->spart = want_slope ? part_sl : nullptr;
      job->ns = want_slope ? nblk : 0;
      job->sout = want_slope ? d_in_slope : nullptr;
      return GCL_OK;
    }
    // + the slope partials (one extra block of the same launch)
    launch_reduce_multi(part_dw, nblk, rec, s0, s1, s2, st, want_slope ? part_sl : nullptr, nblk, d_in_slope);
    GCL_CHECK_LAUNCH();
  }
  return GCL_OK;
}

extern "C" int gcl_linear_bwd_all(const float* dy, int64_t lddy, const float* W, const float* x, int64_t ldx,
                                  const float* in_slope, float* d_in_slope, float* dx, int64_t lddx, float* dW,
                                  float* db, float* colsum_dx, int64_t rows, int32_t Fin, int32_t Fout,
                                  int32_t accumulate, void* ws, size_t ws_bytes, gcl_stream_t stream) {
  return bwd_all_impl(dy, lddy, W, x, ldx, in_slope, d_in_slope, dx, lddx, dW, db, colsum_dx, rows, Fin, Fout,
                      accumulate, ws, ws_bytes, stream, nullptr);
}

extern "C" int gcl_linear_bwd_all_deferred(const float* dy, int64_t lddy, const float* W, const float* x, int64_t ldx,
                                           const float* in_slope, float* d_in_slope, float* dx, int64_t lddx,
                                           float* dW, float* db, float* colsum_dx, int64_t rows, int32_t Fin,
                                           int32_t Fout, int32_t accumulate, void* ws, size_t ws_bytes,
                                           gcl_stream_t stream, gcl_reduce_job* job) {
  GCL_CHECK_ARG(job, "linear_bwd_all_deferred: null job");
  return bwd_all_impl(dy, lddy, W, x, ldx, in_slope, d_in_slope, dx, lddx, dW, db, colsum_dx, rows, Fin, Fout,
                      accumulate, ws, ws_bytes, stream, job);
}

extern "C" int gcl_reduce_jobs(const gcl_reduce_job* jobs, int32_t n, gcl_stream_t stream) {
  GCL_CHECK_ARG(n >= 0 && (jobs || n == 0), "reduce_jobs: null argument");
  hipStream_t st = (hipStream_t)stream;
  // the running index is carried ACROSS launches: skipped (empty) jobs do not count towards the 16 of a launch, so
  // restarting at base + 16 would reduce the jobs behind a skipped one twice
  for (int32_t q = 0; q < n;) {
    RedJobsK J;
    memset(&J, 0, sizeof(J));
    int nb = 0, cnt = 0;
    bool any_slope = false;
    for (; q < n && cnt < kJobsPerLaunch; ++q) {
      const gcl_reduce_job& g = jobs[q];
      if (g.nparts <= 0) continue;  // that call reduced on the spot
      GCL_CHECK_ARG(g.part && g.pstride % 4 == 0 && gcl::aligned16(g.part), "reduce_jobs: bad partial buffer");
      RedJobK& k = J.j[cnt];
      k.part = g.part;
      k.pstride = g.pstride;
      RedSeg* ss[3] = {&k.s0, &k.s1, &k.s2};
      int total = 0;
      for (int t = 0; t < 3; ++t) {
        GCL_CHECK_ARG((g.seg[t].poff | g.seg[t].count) % 4 == 0, "reduce_jobs: segment not 16-byte granular");
        *ss[t] = RedSeg{g.seg[t].out, g.seg[t].poff, g.seg[t].out ? g.seg[t].count : 0, g.seg[t].pld > 0 ? g.seg[t].pld : 1,
                        g.seg[t].cols, g.seg[t].ldo, g.seg[t].acc};
        total += ss[t]->count;
      }
      k.spart = g.spart;
      k.sout = g.sout;
      k.nparts = g.nparts;
      k.ns = g.ns;
      k.blk0 = nb;
      any_slope = any_slope || (g.spart && g.sout && g.ns > 0);
      if (!(g.spart && g.sout && g.ns > 0)) k.spart = nullptr;
      nb += (int)gcl::cdiv(total, 32);
      ++cnt;
    }
    if (cnt == 0) continue;
    J.n = cnt;
    J.nblk = nb;
    hipLaunchKernelGGL(reduce_jobs_kernel, dim3((unsigned)nb + (any_slope ? 1u : 0u)), dim3(256), 0, st, J);
    GCL_CHECK_LAUNCH();
  }
  return GCL_OK;
}
