// Dense per-node transforms for K, N <= 64 on the bf16 matrix pipe with 3-way operand splitting (x3.h):
// fp32 in, fp32 out, fp32-GEMM accuracy, 0.375x the matrix-pipe time of the fp32-operand instruction.
// Same contracts as the fp32 kernels of linear.hip (nn.Linear + preceding PReLU of MLP.forward,
// src/models.py:106-109, and the `lin` GEMM of GCNConv / GATConv, src/models.py:419,425).
//
// Operand maps of v_mfma_f32_32x32x16_bf16 (cdna_hip_programming.md §3), lane l: r = l & 31, h = l >> 5:
//   A: A[row r][k = 8h + j], j = 0..7      B: B[k = 8h + j][col r]      D: reg q -> D[(q&3) + 8(q>>2) + 4h][col r]
// so both fragments are 16 contiguous bytes of a [row | col][k] bf16 image: one ds_read_b128 each.
#include <stdlib.h>

#include "x3.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#ifndef GCL_X3_ST_AUX
#define GCL_X3_ST_AUX 2  // cache-policy bits of the streamed-out stores: 2 = non-temporal (measured +1.1 % end to end: the outputs are
                         // consumed by a LATER kernel, keeping them out of the producer XCD's L2 leaves it to the inputs); 0 = default
#endif
constexpr unsigned kOOB = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, int64_t nbytes) {
  const int64_t cap = 0x7FFFFF00;
  const int n = (int)(nbytes < 0 ? 0 : (nbytes > cap ? cap : nbytes));
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, n, 0x00020000);
}
__device__ __forceinline__ void buf_st4(__amdgpu_buffer_rsrc_t r, unsigned off, float4 v) {
  u32x4 u = {__builtin_bit_cast(unsigned, v.x), __builtin_bit_cast(unsigned, v.y), __builtin_bit_cast(unsigned, v.z),
             __builtin_bit_cast(unsigned, v.w)};
  __builtin_amdgcn_raw_buffer_store_b128(u, r, off, 0, GCL_X3_ST_AUX);
}
__device__ float4 x3_zero4[1];
// rows that this kernel reads exactly once: non-temporal loads (measured +0.7 % end to end; -DGCL_X3_LD_PLAIN: plain)
__device__ __forceinline__ float4 ld_stream(const float4* p) {
#ifndef GCL_X3_LD_PLAIN
  typedef float lv4f __attribute__((ext_vector_type(4)));
  const lv4f v = __builtin_nontemporal_load(reinterpret_cast<const lv4f*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
#else
  return *p;
#endif
}
__device__ __forceinline__ int d_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

using namespace gcl::x3;  // pk_bf16, Pk3, split2, mfma_lo / _mid / _hi, bf16x8 (x3.h)

// LDS image of a [32 rows][64 k] bf16 piece: 144-byte rows (128 + 16): the 16 lanes one ds_read_b128 cycle
// serves ({0-3,12-15,20-27} ...) then sit on 16 distinct 4-bank groups.  Three piece images back to back.
constexpr int kRowB = 144;
constexpr int kImgB = 32 * kRowB;     // one piece
constexpr int kTileB = 3 * kImgB;     // hi | mid | lo of a 32-row tile (13.5 KiB)

__device__ unsigned long long x3_stamps[8 * 4096];  // diagnostic builds only (make STAMPS=1)
#ifdef GCL_STAMPS
#define X3_STAMP(slot)                                                           \
  do {                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                           \
    unsigned long long t_;                                                       \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
    __builtin_amdgcn_sched_barrier(0);                                           \
    stamp_acc[slot] += t_ - stamp_last;                                          \
    stamp_last = t_;                                                             \
  } while (0)
#else
#define X3_STAMP(slot) do {} while (0)
#endif

// ---------------------------------------------------------------------------------------------
// Forward:  Y[r, j] = sum_k act(X[r, k]) W[j, k] + bias[j].   4 waves per block, each wave owns 32 rows of the
// block's 128-row tile and ITS OWN LDS region: no block barrier after the set-up.  The weight fragments of
// all six piece combinations live in registers for the whole launch (NS * 4 * 3 fragments).  Per tile a wave
// (1) splits the rows it prefetched into its hi/mid/lo images, (2) issues the loads of its next tile,
// (3) runs NS * nks * 6 MFMAs on fragments read with ds_read_b128, (4) transposes the accumulators through
// the same LDS region and stores whole 16-byte row segments.
// ---------------------------------------------------------------------------------------------
template <int NS, bool SILU, int NKS>  // NKS: k-steps of 16 (2 for K <= 32, else 4)
__global__ __launch_bounds__(256, 2) void linear_x3_fwd_kernel(const float* __restrict__ X, int64_t ldx,
                                                               const float* __restrict__ slope_p, int32_t akind,
                                                               const float* __restrict__ W,
                                                               const float* __restrict__ bias, float* __restrict__ Y,
                                                               int64_t ldy, int64_t rows, int32_t K, int32_t N,
                                                               int32_t abl) {
  // abl: timing-only ablation bits of tools/ablate.sh (0 in the product): 1 = stores dropped (empty window),
  // 2 = no MFMA passes, 4 = loads read the zero page.  Runtime values: the code is the same.
  extern __shared__ __align__(16) unsigned char smem8[];
#ifdef GCL_STAMPS
  const unsigned long long stamp_t0 = __builtin_amdgcn_s_memtime();
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned char* Aw = smem8 + wave * kTileB;
  const int fo = (lane & 31) * kRowB + (lane >> 5) * 16;  // this lane's fragment inside a row-block, k-step 0
  const bool no_mfma = (abl & 2) != 0;
  const int64_t rows_ld = (abl & 4) ? 0 : rows;

  // ---- weight fragments: the block splits W once into piece images laid over the (still unused) tile
  // regions, every wave then lifts its NS * NKS * 3 fragments into registers ----
  bf16x8 wf[NS][NKS][3];
  {
    constexpr int kWimgB = NS * 32 * kRowB;  // one piece of the whole panel
    float2 wv[NS * 4];
#pragma unroll
    for (int i = 0; i < NS * 4; ++i) {  // NS*32 rows x 32 k-pairs over 256 threads, all loads in flight at once
      const int idx = i * 256 + tid;
      const int j = idx >> 5, k = (idx & 31) * 2;
      const bool okj = j < N;
      wv[i].x = (okj && k < K) ? W[(int64_t)j * K + k] : 0.f;
      wv[i].y = (okj && k + 1 < K) ? W[(int64_t)j * K + k + 1] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NS * 4; ++i) {
      const int idx = i * 256 + tid;
      const int j = idx >> 5, k = (idx & 31) * 2;
      const Pk3 p = split2(wv[i].x, wv[i].y);
      unsigned char* d = smem8 + j * kRowB + k * 2;
      *reinterpret_cast<unsigned*>(d) = p.h;
      *reinterpret_cast<unsigned*>(d + kWimgB) = p.m;
      *reinterpret_cast<unsigned*>(d + 2 * kWimgB) = p.l;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NS; ++t)
#pragma unroll
      for (int s = 0; s < NKS; ++s)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          wf[t][s][p] = *reinterpret_cast<const bf16x8*>(smem8 + p * kWimgB + t * 32 * kRowB + fo + s * 32);
    __syncthreads();  // the last block barrier: from here on every wave only touches its own region
  }

  const float pa = (akind == gcl::kActPrelu) ? *slope_p : 1.f;  // x > 0 ? x : x * pa ("none": pa = 1, exact)
  // bias of the four columns this lane stores (added after the transpose: one rounding on the finished sum)
  float4 bq;
  {
    const int oc = (lane % (NS * 8)) * 4;
    bq.x = (bias && oc < N) ? bias[oc] : 0.f;
    bq.y = (bias && oc + 1 < N) ? bias[oc + 1] : 0.f;
    bq.z = (bias && oc + 2 < N) ? bias[oc + 2] : 0.f;
    bq.w = (bias && oc + 3 < N) ? bias[oc + 3] : 0.f;
  }

  // staging map: 16 lanes x float4 = one 64-float row, 4 rows per wave-instruction, 8 instructions per tile
  const int rsub = lane >> 4, csub = lane & 15;
  const bool cok = csub * 4 < K;
  float4 pre[8];
  auto issue = [&](int64_t tile) {
    const int64_t r0 = tile * 128 + wave * 32;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int64_t row = r0 + it * 4 + rsub;
      const float4* p = (cok && row < rows_ld) ? reinterpret_cast<const float4*>(X + row * ldx + csub * 4) : x3_zero4;
      pre[it] = ld_stream(p);
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      float4 v = pre[it];
      if (SILU) {
        v.x = gcl::silu_f(v.x); v.y = gcl::silu_f(v.y); v.z = gcl::silu_f(v.z); v.w = gcl::silu_f(v.w);
      } else {  // PReLU and "none" (pa = 1) share one branch-free form
        const float mx = v.x * pa, my = v.y * pa, mz = v.z * pa, mw = v.w * pa;
        v.x = v.x > 0.f ? v.x : mx; v.y = v.y > 0.f ? v.y : my;
        v.z = v.z > 0.f ? v.z : mz; v.w = v.w > 0.f ? v.w : mw;
      }
      const Pk3 p01 = split2(v.x, v.y), p23 = split2(v.z, v.w);
      unsigned char* d = Aw + (it * 4 + rsub) * kRowB + csub * 8;
      *reinterpret_cast<u32x2*>(d) = u32x2{p01.h, p23.h};
      *reinterpret_cast<u32x2*>(d + kImgB) = u32x2{p01.m, p23.m};
      *reinterpret_cast<u32x2*>(d + 2 * kImgB) = u32x2{p01.l, p23.l};
    }
  };

  const int64_t ntiles = (rows + 127) / 128;
  issue(blockIdx.x);
#ifdef GCL_STAMPS
  unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_last = __builtin_amdgcn_s_memtime();
  stamp_acc[6] = stamp_last - stamp_t0;  // set-up (weight fragments)
  stamp_acc[7] = stamp_t0;               // absolute start: are all blocks resident at once?
#endif
  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // LDS operations of one wave complete in order
    X3_STAMP(0);
#ifdef GCL_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // split "waiting for the prefetched rows" from the commit work
#endif
    X3_STAMP(1);
    commit();
    X3_STAMP(2);
    issue(t + gridDim.x);  // rows past the end read the zero page
    X3_STAMP(3);

    f32x16 acc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;
    // fragments are re-read from LDS in every pass (6 * NKS ds_read_b128 per tile) rather than held: registers
    // are what limits this kernel to two waves per SIMD
    if (!no_mfma) {  // uniform
#pragma unroll
      for (int s = 0; s < NKS; ++s) {
        const unsigned char* ap = Aw + fo + s * 32;
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(ap);
        const bf16x8 am = *reinterpret_cast<const bf16x8*>(ap + kImgB);
        const bf16x8 al = *reinterpret_cast<const bf16x8*>(ap + 2 * kImgB);
#pragma unroll
        for (int c = 0; c < NS; ++c) acc[c] = mfma_lo(acc[c], ah, am, al, wf[c][s][0], wf[c][s][1], wf[c][s][2]);
      }
#pragma unroll
      for (int s = 0; s < NKS; ++s) {
        const unsigned char* ap = Aw + fo + s * 32;
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(ap);
        const bf16x8 am = *reinterpret_cast<const bf16x8*>(ap + kImgB);
#pragma unroll
        for (int c = 0; c < NS; ++c) acc[c] = mfma_mid(acc[c], ah, am, wf[c][s][0], wf[c][s][1]);
      }
#pragma unroll
      for (int s = 0; s < NKS; ++s) {
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(Aw + fo + s * 32);
#pragma unroll
        for (int c = 0; c < NS; ++c) acc[c] = mfma_hi(acc[c], ah, wf[c][s][0]);
      }
    }

#ifdef GCL_STAMPS
    asm volatile("" ::"v"(acc[0][0]));
#endif
    X3_STAMP(4);
    // transpose through the wave's region ([32][NS*32] fp32 <= 8 KiB of its 13.5) and store 16-byte row segments
    const int64_t r0 = t * 128 + wave * 32;
    const int64_t nr64 = rows - r0;
    const int nr = nr64 < 0 ? 0 : (nr64 < 32 ? (int)nr64 : 32);
    constexpr int OS = NS * 32, LPO = OS / 4, RPP = 64 / LPO;
    float* Ot = reinterpret_cast<float*>(Aw);
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) Ot[d_row(r, lane) * OS + s * 32 + (lane & 31)] = acc[s][r];
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(Y + r0 * ldy, (nr > 0 && !(abl & 1)) ? ((int64_t)(nr - 1) * ldy + N) * 4 : 0);
    const int orow = lane / LPO, ocol = (lane % LPO) * 4;
#pragma unroll
    for (int p = 0; p < 32 / RPP; ++p) {
      const int i = p * RPP + orow;
      float4 v = *reinterpret_cast<const float4*>(Ot + i * OS + ocol);
      v.x += bq.x; v.y += bq.y; v.z += bq.z; v.w += bq.w;
      buf_st4(ry, (ocol < N) ? (unsigned)((i * ldy + ocol) * 4) : kOOB, v);
    }
    X3_STAMP(5);
  }
#ifdef GCL_STAMPS
  if (lane == 0 && blockIdx.x < 1024)
    for (int i = 0; i < 8; ++i) x3_stamps[(blockIdx.x * 4 + wave) * 8 + i] = stamp_acc[i];
#endif
}

// ---------------------------------------------------------------------------------------------
// Fused backward of  y = act(P) W^T + b  for Fin in 33..64 (FiP = 64) and Fout <= 64, 64-row tiles, 4 waves:
//   dX = (dY W) * act'(P)      dW += dY^T act(P)      db += colsum(dY)      colsum_dx += colsum(dX)
//   d_slope += sum(dY W * P, P <= 0)
// Every thread splits the 4 + 4 float4 of dY and act(P) it loaded (coalesced, 16 lanes per row) into the two
// hi/mid/lo LDS images [64 rows][64 columns] of the tile.  Wave (rg, sg) then computes the dX block rows 32rg..,
// columns 32sg.. (A fragments = ds_read_b128 rows of the dY image, B = W^T fragments held in registers) and
// the dW block (so, sc) whose operands sum over the tile's ROWS: both are read from the same images with
// ds_read_b64_tr_b16 (the hardware transposing read: a 4-row x 16-column block per 16 lanes, delivered
// column-major), so no transposed copy of either tile exists.  dX goes through a fp32 staging tile and is
// finished (act', slope and column sums) by the thread that still holds the matching float4 of P, then stored
// as whole 16-byte row segments.  Two block barriers per tile.
// Partial-record layout and outputs are those of linear_bwd_fused64_kernel (linear.hip).
// ---------------------------------------------------------------------------------------------
// Backward images: [64 rows][64 bf16] with UNPADDED 128-byte rows and the eight 16-byte chunks of row r stored at
// chunk ^ swz(r), swz(r) = (bit1(r) << 2) | (bit3(r) << 1) | bit2(r).  One image serves both kinds of read without
// bank conflicts: the 16 rows one ds_read_b128 cycle serves ({0-3,12-15,20-27}, ...) differ in their low four bits, so
// (r & 1, swz(r)) - which 4-bank group of which half of the 64 banks - is distinct for all of them; and of the four
// rows x four chunks one half-wave of ds_read_b64_tr_b16 takes, rows 0/2 (same parity) land on chunk sets that differ
// in bit 2.  (With 144-byte padded rows a quarter of this kernel's LDS cycles were 2-way conflicts of the
// transposing reads: profiles/r02_pmc_linear_x3.txt.)
constexpr int kBRowB = 128;
constexpr int kImg64B = 64 * kBRowB;  // one piece of a 64-row tile
__device__ __forceinline__ int swz(int r) { return (((r >> 1) & 1) << 2) | (((r >> 3) & 1) << 1) | ((r >> 2) & 1); }
// byte offset of byte `b` (< 128) of row `r`
__device__ __forceinline__ int sw_off(int r, int b) { return r * kBRowB + ((((b >> 4) ^ swz(r)) << 4) | (b & 15)); }

typedef short s16x4 __attribute__((ext_vector_type(4)));
// rows rb..rb+3 (offset o0) and rb+4..rb+7 (offset o1: the swizzle differs) of 16 columns, delivered column-major
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* base, int o0, int o1) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + o0));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + o1));
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int NO>
__global__ __launch_bounds__(256, 2) void linear_x3_bwd_kernel(
    const float* __restrict__ dY, int64_t lddy, const float* __restrict__ W, const float* __restrict__ P,
    int64_t ldp, const float* __restrict__ in_slope, float* __restrict__ dX, int64_t lddx, int64_t rows, int32_t Fin,
    int32_t Fout, float* __restrict__ part_dw, float* __restrict__ part_db, float* __restrict__ part_cs,
    double* __restrict__ part_slope) {
  constexpr int FoP = NO * 32, FiP = 64, NKO = NO * 2, NT = NO * 2;
  extern __shared__ __align__(16) unsigned char smem8[];
  unsigned char* Yimg = smem8;                 // [3][64][128 B]  dY pieces (swizzled chunks)
  unsigned char* Pimg = smem8 + 3 * kImg64B;   // [3][64][128 B]  act(P) pieces
  float* Stg = reinterpret_cast<float*>(smem8 + 6 * kImg64B);  // [64][64] fp32: dX before act'
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, h = lane >> 5;
  const int rg = wave & 1, sg = wave >> 1;
  const int so = wave % NO, sc = wave / NO;
  const bool has_tile = wave < NT;

  // ---- W^T fragments of this wave's dX column slab: B[k = o][n = c] = W[o][c], image rows = c ----
  bf16x8 wtf[NKO][3];
  {
    float2 wv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {  // 64 columns x 32 o-pairs over 256 threads
      const int idx = i * 256 + tid;
      const int c = idx & 63, o = (idx >> 6) * 2;
      wv[i].x = (c < Fin && o < Fout) ? W[(int64_t)o * Fin + c] : 0.f;
      wv[i].y = (c < Fin && o + 1 < Fout) ? W[(int64_t)(o + 1) * Fin + c] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = i * 256 + tid;
      const int c = idx & 63, o = (idx >> 6) * 2;
      const Pk3 p = split2(wv[i].x, wv[i].y);
      unsigned char* d = Yimg + sw_off(c, o * 2);
      *reinterpret_cast<unsigned*>(d) = p.h;
      *reinterpret_cast<unsigned*>(d + kImg64B) = p.m;
      *reinterpret_cast<unsigned*>(d + 2 * kImg64B) = p.l;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < NKO; ++s)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        wtf[s][p] = *reinterpret_cast<const bf16x8*>(Yimg + p * kImg64B + sw_off(32 * sg + li, (16 * s + 8 * h) * 2));
    __syncthreads();
  }

  const bool act = in_slope != nullptr;
  const float slope = act ? *in_slope : 1.f;
  const int64_t ntiles = (rows + 63) / 64;

  // staging map: 16 lanes x float4 = one 64-float row; 4 rows per wave-instruction; a wave stages 16 rows
  const int rsub = lane >> 4, csub = lane & 15;
  const bool yok = csub * 4 < Fout, pok = csub * 4 < Fin;
  // element masks of the last partial float4 (Fout % 4 != 0: the padding columns of dY hold finite junk)
  const unsigned ym0 = csub * 4 < Fout ? ~0u : 0u, ym1 = csub * 4 + 1 < Fout ? ~0u : 0u, ym2 = csub * 4 + 2 < Fout ? ~0u : 0u,
                 ym3 = csub * 4 + 3 < Fout ? ~0u : 0u;
  float4 pre_y[4], pre_p[4];
  auto issue = [&](int64_t tile) {
    const int64_t r0 = tile * 64 + wave * 16;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int64_t row = r0 + it * 4 + rsub;
      const bool rok = row < rows;
      pre_y[it] = ld_stream((yok && rok) ? reinterpret_cast<const float4*>(dY + row * lddy + csub * 4) : x3_zero4);
      pre_p[it] = ld_stream((pok && rok) ? reinterpret_cast<const float4*>(P + row * ldp + csub * 4) : x3_zero4);
    }
  };

  // lane parts of the fragment addresses
  // lane parts of the fragment addresses.  Row fragment (ds_read_b128) of k-step s: row 32 rg + li, chunk 2 s + h.
  int fo[NKO];
#pragma unroll
  for (int s = 0; s < NKO; ++s) fo[s] = sw_off(rg * 32 + li, (2 * s + h) * 16);
  // Transposing read of k-step s, half j: lane 4q + pp of 16-lane group g supplies row 16 s + 8 (g >> 1) + 4 j + q,
  // bytes 64 slab + 32 (g & 1) + 8 pp .. +7; bits 4.. of the row do not enter the swizzle, so the k-step is an offset
  const int g = lane >> 4, i16 = lane & 15;
  int troY[2], troP[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = 8 * (g >> 1) + 4 * j + (i16 >> 2), b = 32 * (g & 1) + 8 * (i16 & 3);
    troY[j] = sw_off(r, 64 * so + b);
    troP[j] = sw_off(r, 64 * sc + b);
  }

  f32x16 dw_hi, dw_lo;  // running dW block: the hi x hi products and the five correction products apart
#pragma unroll
  for (int r = 0; r < 16; ++r) dw_hi[r] = 0.f, dw_lo[r] = 0.f;
  float4 db4 = make_float4(0.f, 0.f, 0.f, 0.f), cs4 = make_float4(0.f, 0.f, 0.f, 0.f);
  double slope_acc = 0.0;
  float4 cur_p[4];

  issue(blockIdx.x);
#ifdef GCL_STAMPS
  unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_last = __builtin_amdgcn_s_memtime();
#endif
  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
#ifdef GCL_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    X3_STAMP(0);
    // ---- split this thread's rows into the images ----
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      float4 y = pre_y[it];
      y.x = __uint_as_float(__float_as_uint(y.x) & ym0); y.y = __uint_as_float(__float_as_uint(y.y) & ym1);
      y.z = __uint_as_float(__float_as_uint(y.z) & ym2); y.w = __uint_as_float(__float_as_uint(y.w) & ym3);
      db4.x += y.x; db4.y += y.y; db4.z += y.z; db4.w += y.w;
      const float4 z = pre_p[it];
      cur_p[it] = z;
      float4 a;
      {
        const float mx = z.x * slope, my = z.y * slope, mz = z.z * slope, mw = z.w * slope;
        a.x = z.x > 0.f ? z.x : mx; a.y = z.y > 0.f ? z.y : my; a.z = z.z > 0.f ? z.z : mz; a.w = z.w > 0.f ? z.w : mw;
      }
      const Pk3 y01 = split2(y.x, y.y), y23 = split2(y.z, y.w), a01 = split2(a.x, a.y), a23 = split2(a.z, a.w);
      const int off = sw_off(wave * 16 + it * 4 + rsub, csub * 8);
      *reinterpret_cast<u32x2*>(Yimg + off) = u32x2{y01.h, y23.h};
      *reinterpret_cast<u32x2*>(Yimg + off + kImg64B) = u32x2{y01.m, y23.m};
      *reinterpret_cast<u32x2*>(Yimg + off + 2 * kImg64B) = u32x2{y01.l, y23.l};
      *reinterpret_cast<u32x2*>(Pimg + off) = u32x2{a01.h, a23.h};
      *reinterpret_cast<u32x2*>(Pimg + off + kImg64B) = u32x2{a01.m, a23.m};
      *reinterpret_cast<u32x2*>(Pimg + off + 2 * kImg64B) = u32x2{a01.l, a23.l};
    }
    X3_STAMP(1);
    __syncthreads();
    X3_STAMP(2);
    issue(t + gridDim.x);
    X3_STAMP(3);

    // ---- dX block (rows 32rg.., columns 32sg..), smallest piece products first ----
    {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int s = 0; s < NKO; ++s) {
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(Yimg + fo[s]);
        const bf16x8 am = *reinterpret_cast<const bf16x8*>(Yimg + fo[s] + kImg64B);
        const bf16x8 al = *reinterpret_cast<const bf16x8*>(Yimg + fo[s] + 2 * kImg64B);
        acc = mfma_lo(acc, ah, am, al, wtf[s][0], wtf[s][1], wtf[s][2]);
      }
#pragma unroll
      for (int s = 0; s < NKO; ++s) {
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(Yimg + fo[s]);
        const bf16x8 am = *reinterpret_cast<const bf16x8*>(Yimg + fo[s] + kImg64B);
        acc = mfma_mid(acc, ah, am, wtf[s][0], wtf[s][1]);
      }
#pragma unroll
      for (int s = 0; s < NKO; ++s) acc = mfma_hi(acc, *reinterpret_cast<const bf16x8*>(Yimg + fo[s]), wtf[s][0]);
#pragma unroll
      for (int r = 0; r < 16; ++r) Stg[(rg * 32 + d_row(r, lane)) * 64 + sg * 32 + li] = acc[r];
    }

    X3_STAMP(4);
    // ---- dW block (so, sc) over the 64 rows of the tile: both operands by transposing reads ----
    if (has_tile) {  // wave-uniform
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const unsigned char* ya = Yimg + s * 16 * kBRowB;
        const unsigned char* pa = Pimg + s * 16 * kBRowB;
        const bf16x8 ah = tr_frag(ya, troY[0], troY[1]), am = tr_frag(ya + kImg64B, troY[0], troY[1]),
                     al = tr_frag(ya + 2 * kImg64B, troY[0], troY[1]);
        const bf16x8 bh = tr_frag(pa, troP[0], troP[1]), bm = tr_frag(pa + kImg64B, troP[0], troP[1]),
                     bl = tr_frag(pa + 2 * kImg64B, troP[0], troP[1]);
        dw_lo = mfma_lo(dw_lo, ah, am, al, bh, bm, bl);
        dw_lo = mfma_mid(dw_lo, ah, am, bh, bm);
        dw_hi = mfma_hi(dw_hi, ah, bh);
      }
    }
#ifdef GCL_STAMPS
    asm volatile("" ::"v"(dw_hi[0]), "v"(dw_lo[0]));
#endif
    X3_STAMP(5);
    __syncthreads();
    X3_STAMP(6);

    // ---- finish dX: the thread that loaded P[row, 4 columns] owns that float4 of the result ----
    {
      const int64_t r0 = t * 64;
      const int64_t nr = (rows - r0) < 64 ? (rows - r0) : 64;
      const __amdgpu_buffer_rsrc_t rx = make_rsrc(dX + r0 * lddx, nr > 0 ? ((nr - 1) * lddx + Fin) * 4 : 0);
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = wave * 16 + it * 4 + rsub;
        float4 v = *reinterpret_cast<const float4*>(Stg + row * 64 + csub * 4);
        const float4 z = cur_p[it];
        const bool nx = act && z.x <= 0.f, ny = act && z.y <= 0.f, nz = act && z.z <= 0.f, nw = act && z.w <= 0.f;
        // fp64 across float4s: the slope gradient is a long signed sum with heavy cancellation
        slope_acc += (double)(((nx ? v.x * z.x : 0.f) + (ny ? v.y * z.y : 0.f)) + ((nz ? v.z * z.z : 0.f) + (nw ? v.w * z.w : 0.f)));
        v.x = nx ? v.x * slope : v.x; v.y = ny ? v.y * slope : v.y; v.z = nz ? v.z * slope : v.z; v.w = nw ? v.w * slope : v.w;
        cs4.x += v.x; cs4.y += v.y; cs4.z += v.z; cs4.w += v.w;
        buf_st4(rx, pok ? (unsigned)((row * lddx + csub * 4) * 4) : kOOB, v);
      }
    }
    X3_STAMP(7);
  }
#ifdef GCL_STAMPS
  if (lane == 0 && blockIdx.x < 1024)
    for (int i = 0; i < 8; ++i) x3_stamps[(blockIdx.x * 4 + wave) * 8 + i] = stamp_acc[i];
#endif

  // ---- per-block partials ----
  constexpr size_t REC = (size_t)FoP * FiP + FoP + FiP;
  float* out = part_dw + (size_t)blockIdx.x * REC;
  if (has_tile) {
#pragma unroll
    for (int r = 0; r < 16; ++r) out[(size_t)(so * 32 + d_row(r, lane)) * FiP + sc * 32 + li] = dw_hi[r] + dw_lo[r];
  }
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem8);  // [2][16][64]: db and colsum partials of the 16 row groups
  {
    float* rdb = red + (wave * 4 + rsub) * 64 + csub * 4;
    *reinterpret_cast<float4*>(rdb) = db4;
    *reinterpret_cast<float4*>(rdb + 16 * 64) = cs4;
  }
  for (int off = 32; off > 0; off >>= 1) slope_acc += __shfl_down(slope_acc, off, 64);
  double* dred = reinterpret_cast<double*>(red + 2 * 16 * 64);
  if (lane == 0) dred[wave] = slope_acc;
  __syncthreads();
  if (tid < 128) {
    const int which = tid >> 6, c = tid & 63;
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += red[(which * 16 + q) * 64 + c];
    if (which == 0) {
      if (part_db && c < FoP) part_db[(size_t)blockIdx.x * REC + c] = s;
    } else {
      if (part_cs) part_cs[(size_t)blockIdx.x * REC + c] = s;
    }
  }
  if (part_slope && tid == 0) part_slope[blockIdx.x] = (dred[0] + dred[1]) + (dred[2] + dred[3]);
}

int x3_ablate() {  // GCL_ABLATE: timing-only experiments (tools/ablate.sh) - honoured by the diagnostic build only
#ifndef GCL_STAMPS
  return 0;
#endif
  static const int v = [] { const char* e = getenv("GCL_ABLATE"); return e ? atoi(e) : 0; }();
  return v;
}

int x3_enabled() {
  static const int v = [] {
    const char* e = getenv("GCL_X3");
    return (e && atoi(e) == 0) ? 0 : 1;
  }();
  return v;
}

}  // namespace

#ifdef GCL_STAMPS
// diagnostic builds only: per-wave cycle sums [wave][8] = {loop top, wait loads, commit, issue, mfma, store, ..}
extern "C" int gcl_debug_read_stamps_x3(unsigned long long* host_out, int count) {
  GCL_CHECK_HIP(hipDeviceSynchronize());
  GCL_CHECK_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(x3_stamps), sizeof(unsigned long long) * count));
  return GCL_OK;
}
#endif

namespace gcl {

bool x3_linear_fwd_applicable(const float* x, int64_t ldx, const float* y, int64_t ldy, int K, int N, int akind) {
  if (!x3_enabled()) return false;
  if (K < 1 || K > 64 || N < 4 || N > 64 || (N & 3)) return false;
  if ((ldx & 3) || (ldy & 3) || !aligned16(x) || !aligned16(y)) return false;
  if (ldx * 4 * 35 >= ((int64_t)1 << 31) || ldy * 4 * 35 >= ((int64_t)1 << 31)) return false;  // 32-bit offsets inside a tile
  return akind == kActNone || akind == kActPrelu || akind == kActSilu;
}

int x3_linear_fwd(const float* x, int64_t ldx, int akind, const float* slope, const float* W, const float* bias,
                  float* y, int64_t ldy, int64_t rows, int K, int N, hipStream_t st) {
  if (rows == 0) return GCL_OK;
  const int64_t ntiles = cdiv(rows, 128);
  const int grid = (int)(ntiles < 2 * kNumCU ? ntiles : 2 * kNumCU);
  const size_t lds = 4 * (size_t)kTileB;
#define GCL_X3F3(NS_, SILU_, NKS_)                                                                                \
  hipLaunchKernelGGL((linear_x3_fwd_kernel<NS_, SILU_, NKS_>), dim3(grid), dim3(256), lds, st, x, ldx, slope, akind, \
                     W, bias, y, ldy, rows, K, N, x3_ablate())
#define GCL_X3F(NS_, SILU_)           \
  do {                                \
    if (K <= 32) GCL_X3F3(NS_, SILU_, 2); \
    else GCL_X3F3(NS_, SILU_, 4);     \
  } while (0)
  if (akind == kActSilu) {
    if (N > 32) GCL_X3F(2, true);
    else GCL_X3F(1, true);
  } else {
    if (N > 32) GCL_X3F(2, false);
    else GCL_X3F(1, false);
  }
#undef GCL_X3F3
#undef GCL_X3F
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

bool x3_linear_bwd_applicable(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* dx, int64_t lddx,
                              int Fin, int Fout) {
  if (!x3_enabled()) return false;
  if (Fin < 33 || Fin > 64 || (Fin & 3) || Fout < 1 || Fout > 64) return false;
  if (lddy < ((Fout + 3) & ~3) || (lddy & 3) || (ldx & 3) || (lddx & 3) || lddx < Fin) return false;
  if (!aligned16(dy) || !aligned16(x) || !aligned16(dx)) return false;
  return lddx * 4 * 70 < ((int64_t)1 << 31);  // 32-bit store offsets inside a tile
}

int x3_linear_bwd_blocks(int64_t rows) {
  const int64_t ntiles = cdiv(rows, 64);
  return (int)(ntiles < 2 * kNumCU ? ntiles : 2 * kNumCU);
}

int x3_linear_bwd(const float* dy, int64_t lddy, const float* W, const float* x, int64_t ldx, const float* in_slope,
                  float* dx, int64_t lddx, int64_t rows, int Fin, int Fout, float* part_dw, float* part_db,
                  float* part_cs, double* part_slope, hipStream_t st) {
  const int grid = x3_linear_bwd_blocks(rows);
  const size_t lds = 6 * (size_t)kImg64B + 64 * 64 * sizeof(float);
#define GCL_X3B(NO_)                                                                                              \
  do {                                                                                                            \
    auto kern = linear_x3_bwd_kernel<NO_>;                                                                        \
    { const int lrc_ = gcl::ensure_dyn_lds((const void*)kern, 160 * 1024); if (lrc_) return lrc_; } \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, dy, lddy, W, x, ldx, in_slope, dx, lddx, rows, Fin, Fout, \
                       part_dw, part_db, part_cs, part_slope);                                                    \
  } while (0)
  if (Fout > 32) GCL_X3B(2);
  else GCL_X3B(1);
#undef GCL_X3B
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

}  // namespace gcl
