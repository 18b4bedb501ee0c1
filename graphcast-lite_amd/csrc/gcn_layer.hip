// One GCNConv layer per launch, aggregate-first (src/models.py:419; backward of the same layer):
//
//   forward   y[b,i,:]  = (sum_{e in row i} w_e act(x[b, col_e, :])) W^T + bias
//   backward  dh        = A_hat^T dp                       (gathered through the sender-sorted CSR)
//             dx[b,i,:] = (dh[b,i,:] W) * act'(x[b,i,:])   dW += dh^T act(x)   cs += colsum(dx)   dslope += ...
//
// A_hat (act(X) W^T) = (A_hat act(X)) W^T, so aggregating the INPUT rows first keeps the intermediate
// h = act(X) W^T out of memory altogether: per layer and direction the linear + aggregate pair of round 1
// moved 2 x (read + write) of [B, n, F]; this kernel moves one read (the gather, mostly L2 hits) and one
// write.  Design (MI355X):
//   * WAVE-INDEPENDENT pipelines.  A wave owns 32 destination rows at a time ("wave tile"): it gathers and
//     sums their neighbour rows into registers (16 lanes x float4 per row, ELL prefix -> up to 8 rows in
//     flight per lane), drops the aggregated 32 x K tile into ITS OWN LDS region, runs the exact-fp32
//     v_mfma_f32_32x32x2_f32 chain against the block's weight panel, transposes the result through the
//     same LDS region and writes whole 16-byte row segments.  There is no block barrier after the weight
//     panel is staged: waves drift apart, so one wave's MFMA chain overlaps the other waves' gathers
//     and stores (a block-synchronous version runs its memory and matrix phases back to back).
//   * 12 waves (one 768-thread block) per CU: 118 KB of LDS, <= 168 VGPRs.
//   * XCD-aware persistent schedule: the blocks of XCD group g (blockIdx & 7) walk the samples
//     b = g, g+8, ... so one sample's x (2.6 MB at 64x32) stays in one 4 MiB L2 while its ~7x neighbour
//     re-reads happen.  Placement is a speed choice only.
//   * Rows with more in-edges than the ELL prefix finish in a per-tile fix-up loop (read-modify-write of
//     the wave's LDS tile) so the main path stays branch-free; sums keep PyG edge order either way.
// Shapes: Fin % 4 == 0, Fin <= 64, Fout <= 64, graphs without heavy rows (in-degree <= 64).
#include <stdlib.h>

#include "common.h"
#include "halo.h"
#include "x3.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned kOOB = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, int64_t nbytes) {
  const int64_t cap = 0x7FFFFF00;
  const int n = (int)(nbytes < 0 ? 0 : (nbytes > cap ? cap : nbytes));
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, n, 0x00020000);
}
template <int AUX = 0>  // cache-policy bits: 2 = non-temporal
__device__ __forceinline__ void buf_st4(__amdgpu_buffer_rsrc_t r, unsigned off, float4 v) {
  u32x4 u = {__builtin_bit_cast(unsigned, v.x), __builtin_bit_cast(unsigned, v.y), __builtin_bit_cast(unsigned, v.z),
             __builtin_bit_cast(unsigned, v.w)};
  __builtin_amdgcn_raw_buffer_store_b128(u, r, off, 0, AUX);
}
__device__ float4 gl_zero4[1];

__device__ __forceinline__ int d_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

using gcl::x3::bf16x8;  // exact 3-way bf16 operand split: x3.h
using gcl::x3::Pk3;
using gcl::x3::split2;
constexpr int kWRowB = 144;  // bytes per row of a weight piece image: 64 bf16 + 16 (conflict-free ds_read_b128 fragments)

template <int ACT>
__device__ __forceinline__ float act_t(float x, float a) {
  if (ACT == gcl::kActPrelu) return x > 0.f ? x : a * x;
  if (ACT == gcl::kActSilu) return gcl::silu_f(x);
  return x;
}

// Persistent schedule of one wave: work items are (sample, 32-row tile) pairs; the wave walks
// q = first, first + stride, ... in its item space.  (s, rt) is advanced incrementally - no division per tile.
struct Sched {
  int xcd_map, nRT, s, rt, ds, drt, xg;
  int64_t left;  // items left for this wave, including the current one
  __device__ __forceinline__ void init(int B_, int nRT_, int nwaves) {
    nRT = nRT_;
    const int wave = threadIdx.x >> 6;
    xcd_map = (B_ >= gcl::kNumXCD) && ((gridDim.x & (gcl::kNumXCD - 1)) == 0);
    unsigned first, stride, items;
    if (xcd_map) {
      xg = blockIdx.x & (gcl::kNumXCD - 1);
      const int nsx = (B_ - xg + gcl::kNumXCD - 1) / gcl::kNumXCD;  // samples xg, xg+8, ...
      items = (unsigned)nsx * (unsigned)nRT_;
      first = (blockIdx.x >> 3) * nwaves + wave;
      stride = (gridDim.x >> 3) * nwaves;
    } else {
      xg = 0;
      items = (unsigned)B_ * (unsigned)nRT_;
      first = blockIdx.x * nwaves + wave;
      stride = gridDim.x * nwaves;
    }
    s = first / (unsigned)nRT_;
    rt = first - s * nRT_;
    ds = stride / (unsigned)nRT_;
    drt = stride - ds * nRT_;
    left = first < items ? (int64_t)((items - 1 - first) / stride) + 1 : 0;
  }
  __device__ __forceinline__ int sample() const { return xcd_map ? xg + gcl::kNumXCD * s : s; }
  __device__ __forceinline__ int row0() const { return rt * 32; }
  __device__ __forceinline__ void advance() {
    s += ds;
    rt += drt;
    if (rt >= nRT) {
      rt -= nRT;
      s += 1;
    }
    left -= 1;
  }
};

// Stage a [NJ*32][KP] panel  P[j][k] = (j < N && k < K) ? W[TR ? k*ldw + j : j*ldw + k] : 0  (8 loads in flight)
template <bool TR>
__device__ __forceinline__ void stage_panel(float* Wl, const float* __restrict__ W, int ldw, int N, int K, int NJ32,
                                            int KE, int KP) {
  const int NT = blockDim.x, tid = threadIdx.x;
  const int total = NJ32 * KE;
  for (int base = 0; base < total; base += NT * 8) {
    float wv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + u * NT + tid;
      const int j = idx / KE, k = idx - j * KE;
      const bool ok = (idx < total) && (j < N) && (k < K);
      const float* src = ok ? (TR ? W + (int64_t)k * ldw + j : W + (int64_t)j * ldw + k)
                            : reinterpret_cast<const float*>(gl_zero4);
      wv[u] = *src;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + u * NT + tid;
      const int j = idx / KE, k = idx - j * KE;
      if (idx < total) Wl[j * KP + k] = wv[u];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Gather-aggregate the wave's 32 destination rows [r0, r0+32) of one sample into its LDS tile
//     At[i][c] = sum_{e in row r0+i} w_e act(Hb[col_e][c]),     i < 32, c < K.
// Lane = (row slot sub = lane / 16, channels 4 * (lane % 16) .. +3); 8 passes of 4 rows.
//   * The tile's metadata - rowptr[33], the ELL prefix columns and weights of its 32 rows (1 KB each,
//     contiguous in memory) - is fetched with ONE coalesced load per array a tile ahead (TileMeta, held in
//     9 registers meanwhile) and parked in the wave's LDS region; a pass reads its row's 8 (col, w) pairs
//     from there as broadcast 16-byte LDS reads: no per-pass index loads, no cross-lane shuffles.
//   * Branch-free passes: the row loads of pass p+1 are issued before pass p is summed, every LDS write is
//     unconditional (lanes beyond K write to a sink slot), so the compiler emits counted vmcnt waits.
//   * Rows with more edges than the ELL prefix are completed afterwards by a fix-up loop over the CSR
//     (read-modify-write of the tile, 4 loads in flight); sums keep PyG edge order (prefix, then the rest).
// Rows past n aggregate row n-1 (never stored).
// Per-wave LDS region (floats): tile / output staging [TILE_F] | sink [64] | rowptr [36] | ecol [256] | ew [256]
// ---------------------------------------------------------------------------------------------
constexpr int kStagger = 2;  // x s_sleep(64) = 2 x 4096 cycles per wave group
constexpr int kSinkF = 64, kRpF = 36, kMetaF = kRpF + 2 * 32 * gcl::kEll;

struct TileMeta {  // next tile's metadata in flight
  int4 ec;
  float4 ewv;
  int rp;
};

__device__ __forceinline__ TileMeta meta_issue(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ ecol,
                                               const float* __restrict__ ew, int r0, int n) {
  const int lane = threadIdx.x & 63;
  int row = r0 + (lane >> 1);  // lane covers half (4 entries) of one row's prefix
  row = row < n ? row : n - 1;
  row = row < 0 ? 0 : row;
  TileMeta m;
  m.ec = *reinterpret_cast<const int4*>(ecol + (int64_t)row * gcl::kEll + (lane & 1) * 4);
  m.ewv = *reinterpret_cast<const float4*>(ew + (int64_t)row * gcl::kEll + (lane & 1) * 4);
  int rr = r0 + (lane < 33 ? lane : 32);
  rr = rr < n ? rr : n;
  rr = rr < 0 ? 0 : rr;
  m.rp = rowptr[rr];
  return m;
}

__device__ __forceinline__ void meta_commit(float* __restrict__ Mt, const TileMeta& m) {
  const int lane = threadIdx.x & 63;
  int* rp = reinterpret_cast<int*>(Mt);
  int* ec = rp + kRpF;
  float* ewl = Mt + kRpF + 32 * gcl::kEll;
  *reinterpret_cast<int4*>(ec + lane * 4) = m.ec;
  *reinterpret_cast<float4*>(ewl + lane * 4) = m.ewv;
  rp[lane < 33 ? lane : 33] = m.rp;  // lanes 33..63 land on a spare slot
}

// One sample's rows are addressed as (wave-uniform base) + 32-bit byte offset.  (Not raw_buffer_load_b128: hipcc of
// ROCm 7.2 lowers that builtin to ONE dword load whose value is replicated into all four components.)
typedef __attribute__((address_space(1))) const char gchar;  // global address space: keeps the loads global_load, not flat_load
struct RowBase {
  gchar* p;
};
__device__ __forceinline__ RowBase sample_rows(const float* base) {
  const uint64_t q = reinterpret_cast<uint64_t>(base);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)q), hi = __builtin_amdgcn_readfirstlane((uint32_t)(q >> 32));
  return RowBase{(gchar*)(((uint64_t)hi << 32) | lo)};
}
__device__ __forceinline__ float4 row_ld4(RowBase r, unsigned off) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(1))) const v4f gv4f;
  const v4f v = *(gv4f*)(r.p + off);
  return make_float4(v.x, v.y, v.z, v.w);
}
// max(x, 0) as exactly one VALU instruction (fmaxf / fmed3 are lowered to a canonicalising v_max plus the max)
__device__ __forceinline__ float relu1(float x) {
  float r;
  asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
  return r;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// one neighbour row into the running sums:  a += w * act(v)
//   PReLU: act(x) = s x + (1 - s) max(x, 0)  ->  a += (w s) x + (w (1 - s)) max(x, 0): 4 v_max + 4 packed FMAs per
//   float4 instead of compare / select / multiply per element (the SIMD's VALU issue is the scarce resource here:
//   an fp32 MFMA holds it for its whole duration, so every VALU instruction of the gather is paid in full)
template <int ACT>
__device__ __forceinline__ void row_fma(f32x2& a01, f32x2& a23, const float4& v, float wk, float slope) {
  if (ACT == gcl::kActPrelu) {
    const float ws = wk * slope, w1 = wk - ws;
    a01.x = fmaf(ws, v.x, a01.x); a01.y = fmaf(ws, v.y, a01.y);
    a23.x = fmaf(ws, v.z, a23.x); a23.y = fmaf(ws, v.w, a23.y);
    a01.x = fmaf(w1, relu1(v.x), a01.x); a01.y = fmaf(w1, relu1(v.y), a01.y);
    a23.x = fmaf(w1, relu1(v.z), a23.x); a23.y = fmaf(w1, relu1(v.w), a23.y);
  } else if (ACT == gcl::kActSilu) {
    a01.x = fmaf(wk, gcl::silu_f(v.x), a01.x); a01.y = fmaf(wk, gcl::silu_f(v.y), a01.y);
    a23.x = fmaf(wk, gcl::silu_f(v.z), a23.x); a23.y = fmaf(wk, gcl::silu_f(v.w), a23.y);
  } else {
    a01.x = fmaf(wk, v.x, a01.x); a01.y = fmaf(wk, v.y, a01.y);
    a23.x = fmaf(wk, v.z, a23.x); a23.y = fmaf(wk, v.w, a23.y);
  }
}

// SAFE: mask padded prefix slots with a select (GCL_GRAPH_MEAN: a padded slot points at the row itself, which is
// NOT one of its neighbours there, so a non-finite value must not reach the sum through 0 * inf); GCN graphs
// carry a self-loop in every row, padded slots have weight 0 and need no mask.
template <int ACT, int EW, bool SAFE>
__device__ __forceinline__ void gather_tile(float* __restrict__ At, int KP, float* __restrict__ Sink,
                                            const float* __restrict__ Mt, RowBase rh, int64_t ldh,
                                            int K, const int32_t* __restrict__ col, const float* __restrict__ w,
                                            float slope) {
  const int lane = threadIdx.x & 63;
  const int sub = lane >> 4, l = lane & 15;
  const int c0 = l * 4;
  const bool cactive = c0 < K;
  const int cc = c0 < K - 4 ? c0 : K - 4;  // lanes beyond K re-read the last 16 bytes of the row (never kept)
  const int* rpl = reinterpret_cast<const int*>(Mt);
  const int* ecl = rpl + kRpF;
  const float* ewl = Mt + kRpF + 32 * gcl::kEll;
  const unsigned ldb = (unsigned)ldh * 4u, ccb = (unsigned)cc * 4u;

  auto rows_issue = [&](int p, float4 (&v)[EW]) {
    const int i = p * 4 + sub;
    int jj[8];
    if (EW > 4) {
      const int4 a = *reinterpret_cast<const int4*>(ecl + i * gcl::kEll), b = *reinterpret_cast<const int4*>(ecl + i * gcl::kEll + 4);
      jj[0] = a.x; jj[1] = a.y; jj[2] = a.z; jj[3] = a.w; jj[4] = b.x; jj[5] = b.y; jj[6] = b.z; jj[7] = b.w;
    } else {
      const int4 a = *reinterpret_cast<const int4*>(ecl + i * gcl::kEll);
      jj[0] = a.x; jj[1] = a.y; jj[2] = a.z; jj[3] = a.w; jj[4] = jj[5] = jj[6] = jj[7] = 0;
    }
#pragma unroll
    for (int k = 0; k < EW; ++k) v[k] = row_ld4(rh, __umul24((unsigned)jj[k], ldb) + ccb);  // n, ld < 2^24
  };
  auto rows_sum = [&](int p, const float4 (&v)[EW]) {
    const int i = p * 4 + sub;
    float ww[8];
    if (EW > 4) {
      const float4 a = *reinterpret_cast<const float4*>(ewl + i * gcl::kEll), b = *reinterpret_cast<const float4*>(ewl + i * gcl::kEll + 4);
      ww[0] = a.x; ww[1] = a.y; ww[2] = a.z; ww[3] = a.w; ww[4] = b.x; ww[5] = b.y; ww[6] = b.z; ww[7] = b.w;
    } else {
      const float4 a = *reinterpret_cast<const float4*>(ewl + i * gcl::kEll);
      ww[0] = a.x; ww[1] = a.y; ww[2] = a.z; ww[3] = a.w; ww[4] = ww[5] = ww[6] = ww[7] = 0.f;
    }
    const int deg = rpl[i + 1] - rpl[i];
    f32x2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < EW; ++k) {
      if (SAFE) {
        float4 vk = v[k];
        const bool in = k < deg;
        vk.x = in ? vk.x : 0.f; vk.y = in ? vk.y : 0.f; vk.z = in ? vk.z : 0.f; vk.w = in ? vk.w : 0.f;
        row_fma<ACT>(a01, a23, vk, ww[k], slope);
      } else {
        row_fma<ACT>(a01, a23, v[k], ww[k], slope);  // padded slots: weight 0 on a finite (own, self-looped) row
      }
    }
    float2* d = reinterpret_cast<float2*>(cactive ? At + i * KP + c0 : Sink + l * 4);
    d[0] = make_float2(a01.x, a01.y);
    d[1] = make_float2(a23.x, a23.y);
    return deg;
  };

  // three passes of row loads in flight per lane (ring of three register sets)
  float4 va[EW], vb[EW], vc[EW];
  int maxdeg = 0;
  rows_issue(0, va);
  rows_issue(1, vb);
  rows_issue(2, vc);
  maxdeg = max(maxdeg, rows_sum(0, va));
  rows_issue(3, va);
  maxdeg = max(maxdeg, rows_sum(1, vb));
  rows_issue(4, vb);
  maxdeg = max(maxdeg, rows_sum(2, vc));
  rows_issue(5, vc);
  maxdeg = max(maxdeg, rows_sum(3, va));
  rows_issue(6, va);
  maxdeg = max(maxdeg, rows_sum(4, vb));
  rows_issue(7, vb);
  maxdeg = max(maxdeg, rows_sum(5, vc));
  maxdeg = max(maxdeg, rows_sum(6, va));
  maxdeg = max(maxdeg, rows_sum(7, vb));

  if (__any(maxdeg > EW)) {  // wave-uniform: some row of this tile has more edges than the prefix
#pragma unroll 1
    for (int p = 0; p < 8; ++p) {
      const int i = p * 4 + sub;
      const int st = rpl[i], end = rpl[i + 1];
      if (!__any(end - st > EW)) continue;
      float2* d = reinterpret_cast<float2*>(cactive ? At + i * KP + c0 : Sink + l * 4);
      const float2 t0 = d[0], t1 = d[1];
      f32x2 a01 = {t0.x, t0.y}, a23 = {t1.x, t1.y};
      for (int e = st + EW; __any(e < end); e += 4) {  // 4 neighbour rows in flight per lane
        int j[4];
        float wk[4];
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bool ok = e + u < end;
          j[u] = col[ok ? e + u : 0];  // (rows past n have st == E': never index with it)
          wk[u] = ok ? w[ok ? e + u : 0] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = row_ld4(rh, __umul24((unsigned)j[u], ldb) + ccb);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float4 vu = v[u];
          if (SAFE) {
            const bool ok = e + u < end;
            vu.x = ok ? vu.x : 0.f; vu.y = ok ? vu.y : 0.f; vu.z = ok ? vu.z : 0.f; vu.w = ok ? vu.w : 0.f;
          }
          row_fma<ACT>(a01, a23, vu, wk[u], slope);  // slots past the row's end: weight 0 on row col[0] (finite unless the input is not)
        }
      }
      d[0] = make_float2(a01.x, a01.y);
      d[1] = make_float2(a23.x, a23.y);
    }
  }
}

// acc[s] += At[32][K] x Wl[s*32..+31][K]^T  (fragment scheme of linear.hip: 8-byte reads, k pairs)
template <int NS>
__device__ __forceinline__ void mfma_tile(f32x16 (&acc)[NS], const float* __restrict__ At, const float* __restrict__ Wl,
                                          int KP, int nq) {
  const int lane = threadIdx.x & 63;
  const float* ap = At + (lane & 31) * KP + 2 * (lane >> 5);
  const float* bp = Wl + (lane & 31) * KP + 2 * (lane >> 5);
  float2 a_c = *reinterpret_cast<const float2*>(ap);
  float2 b_c[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) b_c[s] = *reinterpret_cast<const float2*>(bp + s * 32 * KP);
  for (int q = 0; q < nq; ++q) {
    const int qn = (q + 1 < nq) ? q + 1 : q;  // fragments of the next k pair are read before this pair's MFMAs issue
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
}

// Weight piece images for the split-operand MFMA: Wimg[p][j][k] (p = hi | mid | lo) = piece p of W[j][k], rows of
// kWRowB bytes; block-cooperative, 8 loads in flight per thread.
__device__ __forceinline__ void stage_panel_x3(unsigned char* Wimg, const float* __restrict__ W, int N, int K, int NJ32) {
  const int NT = blockDim.x, tid = threadIdx.x;
  const int total = NJ32 * 32;  // (row, k-pair)
  const int pieceB = NJ32 * kWRowB;
  for (int base = 0; base < total; base += NT * 4) {
    float2 wv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = base + u * NT + tid;
      const int j = idx >> 5, k = (idx & 31) * 2;
      const bool ok = idx < total && j < N;
      wv[u].x = (ok && k < K) ? W[(int64_t)j * K + k] : 0.f;
      wv[u].y = (ok && k + 1 < K) ? W[(int64_t)j * K + k + 1] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = base + u * NT + tid;
      if (idx < total) {
        const int j = idx >> 5, k = (idx & 31) * 2;
        const Pk3 p = split2(wv[u].x, wv[u].y);
        unsigned char* d = Wimg + j * kWRowB + k * 2;
        *reinterpret_cast<unsigned*>(d) = p.h;
        *reinterpret_cast<unsigned*>(d + pieceB) = p.m;
        *reinterpret_cast<unsigned*>(d + 2 * pieceB) = p.l;
      }
    }
  }
}

// acc[c] += At[32][K] x W[c*32..+31][K]^T on the bf16 pipe (x3.h): the fp32 tile is read in the bf16 operand layout
// (lane (r, h): 8 consecutive k of row r = four 8-byte reads, conflict-free on the KP = K + 2 stride), split in
// registers into hi / mid / lo fragments, and multiplied with the weight piece fragments (ds_read_b128) in three
// passes, smallest products first.  K % 16 == 0.
template <int NS>
__device__ __forceinline__ void mfma_tile_x3(f32x16 (&acc)[NS], const float* __restrict__ At, int KP,
                                             const unsigned char* __restrict__ Wimg, int nks) {
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const float* ap = At + r * KP + 8 * h;
  const int pieceB = NS * 32 * kWRowB;
  const unsigned char* wp = Wimg + r * kWRowB + h * 16;
  bf16x8 ah[4], am[4], al[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    if (s < nks) {
      const float2 f0 = *reinterpret_cast<const float2*>(ap + 16 * s), f1 = *reinterpret_cast<const float2*>(ap + 16 * s + 2);
      const float2 f2 = *reinterpret_cast<const float2*>(ap + 16 * s + 4), f3 = *reinterpret_cast<const float2*>(ap + 16 * s + 6);
      const Pk3 p0 = split2(f0.x, f0.y), p1 = split2(f1.x, f1.y), p2 = split2(f2.x, f2.y), p3 = split2(f3.x, f3.y);
      typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
      ah[s] = __builtin_bit_cast(bf16x8, u32x4_t{p0.h, p1.h, p2.h, p3.h});
      am[s] = __builtin_bit_cast(bf16x8, u32x4_t{p0.m, p1.m, p2.m, p3.m});
      al[s] = __builtin_bit_cast(bf16x8, u32x4_t{p0.l, p1.l, p2.l, p3.l});
    }
  }
  auto wfrag = [&](int c, int s, int p) {
    return *reinterpret_cast<const bf16x8*>(wp + p * pieceB + c * 32 * kWRowB + s * 32);
  };
#pragma unroll
  for (int s = 0; s < 4; ++s)
    if (s < nks)
#pragma unroll
      for (int c = 0; c < NS; ++c) {
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[s], wfrag(c, s, 0), acc[c], 0, 0, 0);
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], wfrag(c, s, 2), acc[c], 0, 0, 0);
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[s], wfrag(c, s, 1), acc[c], 0, 0, 0);
      }
#pragma unroll
  for (int s = 0; s < 4; ++s)
    if (s < nks)
#pragma unroll
      for (int c = 0; c < NS; ++c) {
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[s], wfrag(c, s, 0), acc[c], 0, 0, 0);
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], wfrag(c, s, 1), acc[c], 0, 0, 0);
      }
#pragma unroll
  for (int s = 0; s < 4; ++s)
    if (s < nks)
#pragma unroll
      for (int c = 0; c < NS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], wfrag(c, s, 0), acc[c], 0, 0, 0);
}

// Transpose the 32 x (NS*32) accumulator tile through the wave's LDS region Ot[32][NS*32] and write whole
// 16-byte row segments:  Y[r0 + i][0 .. Nst) (rows >= nr and columns >= Nst dropped by the range check).
template <int NS, int AUX = 0>
__device__ __forceinline__ void store_tile(const f32x16 (&acc)[NS], float* __restrict__ Ot, float* Ybase, int64_t ldy,
                                           int nr, int Nst, float4 bq = make_float4(0.f, 0.f, 0.f, 0.f)) {
  constexpr int OS = NS * 32;          // floats per staged row
  constexpr int LPO = OS / 4;          // lanes per row (8 or 16)
  constexpr int RPP = 64 / LPO;        // rows per pass
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int r = 0; r < 16; ++r) Ot[d_row(r, lane) * OS + s * 32 + (lane & 31)] = acc[s][r];
  const __amdgpu_buffer_rsrc_t ry = make_rsrc(Ybase, nr > 0 ? ((int64_t)(nr - 1) * ldy + Nst) * 4 : 0);
  const int orow = lane / LPO, ocol = (lane % LPO) * 4;
#pragma unroll
  for (int p = 0; p < 32 / RPP; ++p) {
    const int i = p * RPP + orow;
    float4 v = *reinterpret_cast<const float4*>(Ot + i * OS + ocol);
    v.x += bq.x; v.y += bq.y; v.z += bq.z; v.w += bq.w;  // (zero unless the bias was kept out of the accumulator)
    buf_st4<AUX>(ry, (ocol < Nst) ? (unsigned)((i * ldy + ocol) * 4) : kOOB, v);
  }
}

__device__ unsigned long long gl_stamps[8 * 4096];  // diagnostic builds only (make STAMPS=1)
#ifdef GCL_STAMPS
#define GCL_STAMP(slot)                                                                  \
  do {                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                   \
    unsigned long long t_;                                                               \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");          \
    __builtin_amdgcn_sched_barrier(0);                                                   \
    stamp_acc[slot] += t_ - stamp_last;                                                  \
    stamp_last = t_;                                                                     \
  } while (0)
#else
#define GCL_STAMP(slot) do {} while (0)
#endif

// floats of one wave's LDS region
__host__ __device__ constexpr int wave_region_f(int tile_f) { return tile_f + kSinkF + kMetaF; }

// X3: the dense part runs on the bf16 matrix pipe with exact 3-way operand splitting (fp32 accuracy, 0.375x the matrix
// time, and - unlike the fp32-operand MFMA - it leaves the SIMD's issue port to the other waves' gathers meanwhile).
template <int NS, int ACT, int EW, int NW, bool SAFE, bool X3>
__global__ __launch_bounds__(NW * 64, (NW + 3) / 4) void gcn_fwd_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ w,
    const int32_t* __restrict__ ecol, const float* __restrict__ ew, const float* __restrict__ X, int64_t ldx,
    int64_t bsx, const float* __restrict__ slope_p, const float* __restrict__ W, const float* __restrict__ bias,
    float* __restrict__ Y, int64_t ldy, int64_t bsy, int32_t n, int32_t B, int32_t K, int32_t N, int32_t Nst,
    int32_t nRT, int32_t n_out) {
  extern __shared__ __align__(16) float smem[];
  const int KP = K + 2;  // K % 4 == 0: even stride with KP/2 odd -> conflict-free 8-byte fragment reads
  float* Wl = smem;      // fp32: [NS*32][KP]; X3: three bf16 piece images [NS*32][kWRowB bytes]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int TILE_F = 32 * (KP > NS * 32 ? KP : NS * 32);  // the [32][KP] tile, reused as the [32][NS*32] output staging
  const size_t panel_f = X3 ? (size_t)3 * NS * 32 * kWRowB / 4 : (size_t)NS * 32 * KP;
  float* At = smem + panel_f + (size_t)wave * wave_region_f(TILE_F);
  float* Sink = At + TILE_F;
  float* Mt = Sink + kSinkF;
  if (X3) stage_panel_x3(reinterpret_cast<unsigned char*>(Wl), W, N, K, NS * 32);
  else stage_panel<false>(Wl, W, K, N, K, NS * 32, K, KP);
  const float slope = (ACT == gcl::kActPrelu) ? *slope_p : 1.f;
  float bj[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int j = s * 32 + (lane & 31);
    bj[s] = (bias && j < N && !X3) ? bias[j] : 0.f;
  }
  float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);  // X3: the bias of the four columns this lane stores, added after the transpose
  if (X3 && bias) {
    const int oc = (lane % (NS * 8)) * 4;
    bq.x = oc < N ? bias[oc] : 0.f; bq.y = oc + 1 < N ? bias[oc + 1] : 0.f;
    bq.z = oc + 2 < N ? bias[oc + 2] : 0.f; bq.w = oc + 3 < N ? bias[oc + 3] : 0.f;
  }
  __syncthreads();  // the only block barrier: from here on every wave works on its own LDS region

  Sched sc;
  sc.init(B, nRT, NW);
  int b = sc.sample(), r0 = sc.row0();
  if (sc.left <= 0) b = 0, r0 = 0;
  TileMeta mt = meta_issue(rowptr, ecol, ew, r0, n);
#ifdef GCL_STAMPS
  unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_last = __builtin_amdgcn_s_memtime();
#endif
  // Stagger: the waves run the same program, so without it they fall into lockstep - all twelve gather at once
  // (the CU's load path is shared, each crawls), then all want the matrix pipe at once.  Wave groups 1 and 2
  // start a third / two thirds of a tile period later; equal periods keep the offset.  (Speed only.)
  {
    const int grp = __builtin_amdgcn_readfirstlane(wave) >> 2;
    for (int i = 0; i < grp * kStagger; ++i) __builtin_amdgcn_s_sleep(64);
  }
  while (sc.left > 0) {
    GCL_STAMP(0);
    meta_commit(Mt, mt);
    sc.advance();
    const int bn = sc.left > 0 ? sc.sample() : b, rn = sc.left > 0 ? sc.row0() : r0;
    mt = meta_issue(rowptr, ecol, ew, rn, n);  // next tile's metadata: in flight during this tile
    GCL_STAMP(1);
    {
      // prefix width of THIS tile: the widest row decides how many neighbour loads every lane issues
      const int* rpl = reinterpret_cast<const int*>(Mt);
      int dmax = rpl[(lane & 31) + 1] - rpl[lane & 31];
#pragma unroll
      for (int off = 16; off > 0; off >>= 1) dmax = max(dmax, __shfl_xor(dmax, off, 64));
      dmax = __builtin_amdgcn_readfirstlane(dmax);
      const RowBase rh = sample_rows(X + (int64_t)b * bsx);
      if (EW >= 8 && dmax > 4) gather_tile<ACT, 8, SAFE>(At, KP, Sink, Mt, rh, ldx, K, col, w, slope);
      else if (EW >= 4 && dmax > 2) gather_tile<ACT, 4, SAFE>(At, KP, Sink, Mt, rh, ldx, K, col, w, slope);
      else if (EW >= 2 && dmax > 1) gather_tile<ACT, 2, SAFE>(At, KP, Sink, Mt, rh, ldx, K, col, w, slope);
      else gather_tile<ACT, 1, SAFE>(At, KP, Sink, Mt, rh, ldx, K, col, w, slope);
    }
    GCL_STAMP(2);
    f32x16 acc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[s][r] = bj[s];  // the bias rides in the accumulator's initial value
    if (X3) mfma_tile_x3<NS>(acc, At, KP, reinterpret_cast<const unsigned char*>(Wl), K >> 4);
    else mfma_tile<NS>(acc, At, Wl, KP, K >> 2);
#ifdef GCL_STAMPS
    asm volatile("" ::"v"(acc[0][0]));
#endif
    GCL_STAMP(3);
    store_tile<NS>(acc, At, Y + (int64_t)b * bsy + (int64_t)r0 * ldy, ldy, n_out - r0 < 32 ? n_out - r0 : 32, Nst, bq);
    GCL_STAMP(4);
    b = bn;
    r0 = rn;
  }
#ifdef GCL_STAMPS
  if (lane == 0 && blockIdx.x < 4096 / NW) {
    for (int i = 0; i < 8; ++i) gl_stamps[(blockIdx.x * NW + wave) * 8 + i] = stamp_acc[i];
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------
// The same layer on graphs that carry a source-tile layout (common.h gcl_halo: mesh rows in tile order): the gather
// stage is the one of agg_halo_loop_kernel (aggregate.hip) - a block stages a 64-row tile's own rows and its halo
// once by LDS-DMA and forms the sums from LDS - instead of one L2 gather per edge, which bounds gcn_fwd_kernel on the
// mesh graph (145 us per layer against 61 us for the source-tile aggregation of the same rows).
// Block = 4 waves, persistent over a contiguous range of (tile, sample) items, tile-major (list entries and edge
// records of a tile stay in registers for all samples of the XCD group).  Per item:
//   wait DMA | act(x) in place on the pieces the wave's own DMAs delivered | barrier | sums of the wave's 16 rows (fused
//   multiply-adds in CSR order: gather_tile's arithmetic on a pre-activated input, bit for bit) -> At[pair][32][KP] | barrier (At complete, image free) | DMA of the NEXT item (inline asm: in flight
//   under what follows) | wave (pair p, half h): acc = At[p] x W[h*32 .. +31]^T on the bf16 pipe (x3) | barrier |
//   transpose through At[p]'s half h, 16-byte row stores.
// The wave's weight fragments (all pieces, all k-steps) live in registers, so LDS holds only At (16.9 KB) and the
// image (31 KB): three blocks per CU, <= 168 VGPRs.
// ---------------------------------------------------------------------------------------------------------
// the wave's weight fragments: rows [j0, j0 + 32) of W (output columns), all k-steps, split once per launch into
// hi / mid / lo bf16 pieces and kept in registers (48 VGPRs) - no LDS image, which is what lets three blocks share a CU
struct WFrag {
  bf16x8 f[4][3];  // [k-step][piece hi | mid | lo]
};
__device__ __forceinline__ WFrag load_wfrag(const float* __restrict__ W, int j0, int N, int K) {
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const int j = j0 + r;
  WFrag wf;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = s * 16 + h * 8 + u;
      v[u] = (j < N && k < K) ? W[(int64_t)j * K + k] : 0.f;
    }
    const Pk3 p0 = split2(v[0], v[1]), p1 = split2(v[2], v[3]), p2 = split2(v[4], v[5]), p3 = split2(v[6], v[7]);
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    wf.f[s][0] = __builtin_bit_cast(bf16x8, u32x4_t{p0.h, p1.h, p2.h, p3.h});
    wf.f[s][1] = __builtin_bit_cast(bf16x8, u32x4_t{p0.m, p1.m, p2.m, p3.m});
    wf.f[s][2] = __builtin_bit_cast(bf16x8, u32x4_t{p0.l, p1.l, p2.l, p3.l});
  }
  return wf;
}

struct NoBetween {
  __device__ __forceinline__ void operator()(int) const {}
};
// `between(s)` runs after the operands of k-step s are split (the source-tile kernel issues two LDS-DMA pieces of the
// next item there: a DMA instruction that waits for a free slot in the memory pipeline then overlaps the splits and
// the matrix chain of the same wave instead of standing in front of them)
template <class F = NoBetween>
__device__ __forceinline__ void mfma_half_x3(f32x16& acc, const float* __restrict__ At, int KP, const WFrag& wf, int nks,
                                             F between = F()) {
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const float* ap = At + r * KP + 8 * h;
  bf16x8 ah[4], am[4], al[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    if (s < nks) {
      const float2 f0 = *reinterpret_cast<const float2*>(ap + 16 * s), f1 = *reinterpret_cast<const float2*>(ap + 16 * s + 2);
      const float2 f2 = *reinterpret_cast<const float2*>(ap + 16 * s + 4), f3 = *reinterpret_cast<const float2*>(ap + 16 * s + 6);
      const Pk3 p0 = split2(f0.x, f0.y), p1 = split2(f1.x, f1.y), p2 = split2(f2.x, f2.y), p3 = split2(f3.x, f3.y);
      typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
      ah[s] = __builtin_bit_cast(bf16x8, u32x4_t{p0.h, p1.h, p2.h, p3.h});
      am[s] = __builtin_bit_cast(bf16x8, u32x4_t{p0.m, p1.m, p2.m, p3.m});
      al[s] = __builtin_bit_cast(bf16x8, u32x4_t{p0.l, p1.l, p2.l, p3.l});
    }
    between(s);
  }
  // the order of mfma_tile_x3 (smallest products first), so both kernels produce the same bits
#pragma unroll
  for (int s = 0; s < 4; ++s)
    if (s < nks) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[s], wf.f[s][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], wf.f[s][2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[s], wf.f[s][1], acc, 0, 0, 0);
    }
#pragma unroll
  for (int s = 0; s < 4; ++s)
    if (s < nks) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[s], wf.f[s][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], wf.f[s][1], acc, 0, 0, 0);
    }
#pragma unroll
  for (int s = 0; s < 4; ++s)
    if (s < nks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], wf.f[s][0], acc, 0, 0, 0);
}

// TAB: the layer's input rows are read THROUGH a row table instead of from a materialised [B, n, K] tensor: row i of
// sample b is row tab[i] of sample b of X when tab[i] >= 0, and the batch-invariant row ~tab[i] of X viewed as one flat
// row list otherwise (the compact pipeline's mesh latents: models.py::_forward_compact, functional.MeshLatFn - 81 % of
// the mesh rows at 64x32 are batch-invariant and then come from a 2 MB region that stays in the L2).  The table is
// applied once per tile (list entries and own rows), so an item costs two more vector instructions per DMA piece.
template <int ACT, int MAXPW, int STAUX = 0, bool DIRECT = false, bool INTER = false, bool TAB = false>
__global__ __launch_bounds__(256, 3) void gcn_halo_fwd_kernel(
    const int32_t* __restrict__ list, const int32_t* __restrict__ cnt, const int2* __restrict__ rec,
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ opos, const float* __restrict__ w, int32_t smax,
    const float* __restrict__ X, int64_t ldx, int64_t bsx, const float* __restrict__ slope_p, const float* __restrict__ W,
    const float* __restrict__ bias, float* __restrict__ Y, int64_t ldy, int64_t bsy, int32_t n, int32_t B, int32_t K,
    int32_t N, int32_t Nst, int32_t ntiles, const int32_t* __restrict__ tab = nullptr) {
  using gcl::halo::glds16;
  using gcl::halo::row_bcast;
  extern __shared__ __align__(16) float smem[];
  constexpr int T = 64, LPR = 16, RPW = 4, NIT = 4, SH = 8;
  typedef float v4f __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) const v4f* lds4_t;
  const int KP = K + 2;
  const int AtF = 32 * KP > 2048 ? 32 * KP : 2048;                    // floats per pair: the [32][KP] tile, later two [32][32] output stagings
  float* At = smem;                                                   // [2][AtF]
  float* imgf = At + 2 * AtF;                                         // [(smax + 1) * 16] float4
  float4* img = reinterpret_cast<float4*>(imgf);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int sub = lane >> 4, l = lane & 15, c0 = l * 4;
  const bool cactive = c0 < K;
  const unsigned cb = (unsigned)(cactive ? c0 : 0) * 4u, ldb = (unsigned)ldx * 4u;
  const int hstride = smax - T;
  const unsigned lds_img = (unsigned)(size_t)((gcl::halo::lptr_t)img);
  const unsigned lb = lds_img + (unsigned)l * 16u;
  const int zrow = smax << SH;
  const int pair = wave >> 1, half = wave & 1;
  const WFrag wf = load_wfrag(W, half * 32, N, K);
  if (threadIdx.x < LPR) img[smax * LPR + threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
  const float slope = (ACT == gcl::kActPrelu) ? *slope_p : 1.f;
  float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);  // bias of the four columns this lane stores (added after the transpose)
  if (bias) {
    const int oc = half * 32 + (lane & 7) * 4;
    bq.x = oc < N ? bias[oc] : 0.f; bq.y = oc + 1 < N ? bias[oc + 1] : 0.f;
    bq.z = oc + 2 < N ? bias[oc + 2] : 0.f; bq.w = oc + 3 < N ? bias[oc + 3] : 0.f;
  }
  float bcol = 0.f;  // DIRECT: bias of the ONE output column this lane's accumulator registers belong to
  if (bias && half * 32 + (lane & 31) < N) bcol = bias[half * 32 + (lane & 31)];
  asm volatile("" : "+v"(bq.x), "+v"(bq.y), "+v"(bq.z), "+v"(bq.w), "+v"(bcol));  // hipcc's wait for these loads: here, not in the loop
  __syncthreads();

  const int xcd = blockIdx.x & (gcl::kNumXCD - 1);
  const int J = gridDim.x >> 3, j = blockIdx.x >> 3;
  const int nsamp = (B - xcd + gcl::kNumXCD - 1) / gcl::kNumXCD;
  const int items = nsamp * ntiles;  // item = tile * nsamp + sample index
  const int base = items / J, extra = items - base * J;
  int m = j * base + min(j, extra);
  const int mend = m + base + (j < extra ? 1 : 0);
  if (m >= mend) return;

  int jj[MAXPW], nhalo = 0, tile = -1;
  int own[TAB ? NIT : 1];  // TAB: table entries of the wave's own rows
  int2 rc[NIT];
  auto new_tile = [&](int t) {  // list entries (scalar loads) and edge records of tile t: kept for all samples of the group
    tile = t;
    nhalo = cnt[t] / RPW;
    const int32_t* __restrict__ tl = list + (int64_t)t * hstride;
#pragma unroll
    for (int q = 0; q < MAXPW; ++q) {
      const int e0 = min((wave + 4 * q) * RPW, hstride - RPW);
      int jv = tl[e0];
#pragma unroll
      for (int r = 1; r < RPW; ++r) {
        const int jr = tl[e0 + r];
        jv = sub == r ? jr : jv;
      }
      jj[q] = TAB ? tab[jv] : jv;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int row = t * T + wave * 16 + sub + it * RPW;
      rc[it] = rec[(int64_t)(row < n ? row : n - 1) * gcl::kHaloRec + l];
      if (TAB) own[it] = tab[row < n ? row : n - 1];
    }
  };
  // byte offset of a staged source row: plain row id (relative to the sample's base) or table entry (relative to X)
  auto src_off = [&](int src, unsigned sboff) -> unsigned {
    if (!TAB) return __umul24(src, ldb) + cb;
    return (src >= 0 ? sboff + __umul24(src, ldb) : __umul24(~src, ldb)) + cb;
  };
  auto stage = [&](int mm) {  // all DMA pieces of item mm (its tile is `tile`)
    const int s = mm - tile * nsamp;
    const char* Xc = reinterpret_cast<const char*>(X + (TAB ? 0 : (int64_t)(xcd + gcl::kNumXCD * s) * bsx));
    const unsigned sboff = TAB ? (unsigned)(xcd + gcl::kNumXCD * s) * (unsigned)bsx * 4u : 0u;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int row = tile * T + wave * 16 + sub + it * RPW;
      glds16(Xc, src_off(TAB ? own[TAB ? it : 0] : (row < n ? row : n - 1), sboff), lds_img + (unsigned)((wave * 16 + it * RPW) << SH));
    }
#pragma unroll
    for (int q = 0; q < MAXPW; ++q) {
      const int p = wave + 4 * q;
      if (p < nhalo) glds16(Xc, src_off(jj[q], sboff), lds_img + (unsigned)((T + p * RPW) << SH));
    }
  };
  // INTER: the same pieces two at a time, piece index 0..3 = own rows, 4.. = halo
  const char* Xn = nullptr;
  unsigned sbn = 0;
  auto stage_pair = [&](int s2) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int pc = s2 * 2 + u;
      if (pc < NIT) {
        const int row = tile * T + wave * 16 + sub + pc * RPW;
        glds16(Xn, src_off(TAB ? own[TAB ? pc : 0] : (row < n ? row : n - 1), sbn), lds_img + (unsigned)((wave * 16 + pc * RPW) << SH));
      } else if (pc - NIT < MAXPW) {
        const int q = pc - NIT, p = wave + 4 * q;
        if (p < nhalo) glds16(Xn, src_off(jj[q < MAXPW ? q : 0], sbn), lds_img + (unsigned)((T + p * RPW) << SH));
      }
    }
  };
  new_tile(m / nsamp);
#pragma unroll
  for (int it = 0; it < NIT; ++it) asm volatile("" : "+v"(rc[it].x), "+v"(rc[it].y));  // records are in before any DMA is issued
  if (TAB) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) asm volatile("" : "+v"(own[TAB ? it : 0]));
  }
  stage(m);
#ifdef GCL_STAMPS
  unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_last = __builtin_amdgcn_s_memtime();
#endif

  while (true) {
    const int s = m - tile * nsamp;
    const int b = xcd + gcl::kNumXCD * s;
    const int trow = tile * T;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of the item (and its stores of the previous one)
    if (ACT != gcl::kActNone) {
      // the input activation ONCE per staged element, in place, by the lane whose DMA delivered the piece (no other wave
      // has touched the image yet): the sums then weigh act(x) like plain rows.  A row is read by 7.4 edges on the mesh,
      // so activating per edge (gather_tile's row_fma<ACT>) cost 10 extra vector instructions per slot and lane.
      typedef __attribute__((address_space(3))) v4f* lds4w_t;
      auto act_piece = [&](unsigned ad) {
        v4f v = *(lds4_t)ad;
        if (ACT == gcl::kActPrelu) {
          v.x = v.x > 0.f ? v.x : slope * v.x; v.y = v.y > 0.f ? v.y : slope * v.y;
          v.z = v.z > 0.f ? v.z : slope * v.z; v.w = v.w > 0.f ? v.w : slope * v.w;
        } else {
          v.x = gcl::silu_f(v.x); v.y = gcl::silu_f(v.y); v.z = gcl::silu_f(v.z); v.w = gcl::silu_f(v.w);
        }
        *(lds4w_t)ad = v;
      };
      const unsigned la = lds_img + (unsigned)lane * 16u;
#pragma unroll
      for (int it = 0; it < NIT; ++it) act_piece(la + (unsigned)((wave * 16 + it * RPW) << SH));
#pragma unroll
      for (int q = 0; q < MAXPW; ++q) {
        const int p = wave + 4 * q;
        if (p < nhalo) act_piece(la + (unsigned)((T + p * RPW) << SH));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                       // image complete
    GCL_STAMP(0);  // wait for the item + barrier
    // ---- sums of this wave's 16 rows: a = sum_e w_e act(x_src), CSR order, fused multiply-adds (row_fma)
    float* Ap = At + pair * AtF;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int row = trow + wave * 16 + it * RPW + sub;
      const int rx = rc[it].x, rw = rc[it].y;
      const int rxb = (rx & gcl::kHaloPosMask) << SH;
      f32x2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
#define GCL_FUSED_SLOT(Kk)                                              \
  {                                                                     \
    const unsigned ad = (unsigned)row_bcast<Kk>(rxb) + lb;              \
    const float wk = __int_as_float(row_bcast<Kk>(rw));                 \
    const v4f v = *(lds4_t)ad;                                          \
    row_fma<gcl::kActNone>(a01, a23, make_float4(v.x, v.y, v.z, v.w), wk, 1.f); \
  }
      GCL_FUSED_SLOT(0) GCL_FUSED_SLOT(1) GCL_FUSED_SLOT(2) GCL_FUSED_SLOT(3)
      GCL_FUSED_SLOT(4) GCL_FUSED_SLOT(5) GCL_FUSED_SLOT(6) GCL_FUSED_SLOT(7)
      const int last = row_bcast<15>(rx);
      if (__any(row_bcast<8>(rxb) != zrow)) {
        GCL_FUSED_SLOT(8) GCL_FUSED_SLOT(9) GCL_FUSED_SLOT(10) GCL_FUSED_SLOT(11)
        GCL_FUSED_SLOT(12) GCL_FUSED_SLOT(13) GCL_FUSED_SLOT(14) GCL_FUSED_SLOT(15)
        if (__any((last & gcl::kHaloMore) != 0)) {
          const int rcl = row < n ? row : n - 1;
          const int end = (last & gcl::kHaloMore) ? rowptr[rcl + 1] : 0;
          for (int e = rowptr[rcl] + gcl::kHaloRec; e < end; ++e) {
            const v4f v = *(lds4_t)(((unsigned)opos[e] << SH) + lb);
            row_fma<gcl::kActNone>(a01, a23, make_float4(v.x, v.y, v.z, v.w), w[e], 1.f);
          }
        }
      }
#undef GCL_FUSED_SLOT
      if (cactive) {
        float2* d = reinterpret_cast<float2*>(Ap + (half * 16 + it * RPW + sub) * KP + c0);
        d[0] = make_float2(a01.x, a01.y);
        d[1] = make_float2(a23.x, a23.y);
      }
    }
    GCL_STAMP(1);  // sums -> At
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // At complete; every wave is done with the image
    GCL_STAMP(2);  // barrier
    // ---- the next item's DMA: in flight under the dense part and the stores of this one
    const bool more = m + 1 < mend;
    if (more) {
      const int tn = (m + 1) / nsamp;
      if (tn != tile) new_tile(tn);
      if (!INTER) stage(m + 1);
      else if (TAB) Xn = reinterpret_cast<const char*>(X), sbn = (unsigned)(xcd + gcl::kNumXCD * (m + 1 - tile * nsamp)) * (unsigned)bsx * 4u;
      else Xn = reinterpret_cast<const char*>(X + (int64_t)(xcd + gcl::kNumXCD * (m + 1 - tile * nsamp)) * bsx);
    }
    GCL_STAMP(3);  // DMA issue of the next item
    // ---- dense part: wave (pair, half) = rows [pair*32, +32) x output columns [half*32, +32)
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (INTER) {
      static_assert(!INTER || NIT + MAXPW <= 8, "the interleaved issue places two pieces behind each of four k-steps");
      mfma_half_x3(acc, Ap, KP, wf, K >> 4, [&](int s2) {  // four callback slots whatever K is (K = 48: slot 3 has no k-step)
        if (more) stage_pair(s2);
      });
    } else
    mfma_half_x3(acc, Ap, KP, wf, K >> 4);
#ifdef GCL_STAMPS
    asm volatile("" ::"v"(acc[0]));
#endif
    GCL_STAMP(4);  // split + MFMA
    if (DIRECT) {
      // straight from the accumulator: register r of lane L is element (d_row(r, L), L & 31) of the 32 x 32 block, so one
      // store instruction writes two rows x 32 consecutive columns = two whole 128-byte lines - no staging tile, no
      // third barrier (the next item's sums cannot touch At before every wave has passed the next barrier 1)
      const int r0 = trow + pair * 32;
      const int nr = n - r0 < 32 ? n - r0 : 32;
      const int col = half * 32 + (lane & 31);
      float* Yb = Y + (int64_t)b * bsy + (int64_t)r0 * ldy;
      const __amdgpu_buffer_rsrc_t ry = make_rsrc(Yb, nr > 0 ? ((int64_t)(nr - 1) * ldy + Nst) * 4 : 0);
      const unsigned cofs = col < Nst ? (unsigned)col * 4u : kOOB;
      const unsigned ldyb = (unsigned)ldy * 4u;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned off = (unsigned)d_row(r, lane) * ldyb + cofs;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[r] + bcol), ry, off, 0, STAUX);
      }
      GCL_STAMP(6);
    } else {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // both waves of the pair have read At[pair]: its halves become the output staging
    GCL_STAMP(5);  // barrier
    {
      f32x16 accv[1] = {acc};
      const int r0 = trow + pair * 32;
      const int nr = n - r0 < 32 ? n - r0 : 32;
      int nst = Nst - half * 32;
      nst = nst < 0 ? 0 : (nst > 32 ? 32 : nst);
      store_tile<1, STAUX>(accv, Ap + half * (32 * 32), Y + (int64_t)b * bsy + (int64_t)r0 * ldy + half * 32, ldy, nr, nst, bq);
    }
    GCL_STAMP(6);  // transpose + stores
    }
#ifdef GCL_STAMPS
    stamp_acc[7] += 1;
#endif
    if (!more) break;
    ++m;
  }
#ifdef GCL_STAMPS
  if (lane == 0 && blockIdx.x < 4096 / 4)
    for (int i = 0; i < 8; ++i) gl_stamps[(blockIdx.x * 4 + wave) * 8 + i] = stamp_acc[i];
#endif
}

int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

}  // namespace

#ifdef GCL_STAMPS
// diagnostic builds only: per-wave cycle sums [wave][8] = {loop top, meta, gather, mfma, store, ...}
extern "C" int gcl_debug_read_stamps(unsigned long long* host_out, int count) {
  GCL_CHECK_HIP(hipDeviceSynchronize());
  GCL_CHECK_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(gl_stamps), sizeof(unsigned long long) * count));
  return GCL_OK;
}
#endif

// Source-tile form of the layer (gcn_halo_fwd_kernel): graphs with a tile layout (mesh rows in tile order), 64-wide rows,
// split-operand dense part.  Returns GCL_OK with *launched = false when the shape / graph is outside its range.
// tab != nullptr: the input rows are read through the row table (see the kernel), x_rows = rows of X as one flat list.
static int halo_layer_launch(const gcl_graph_t* g, const float* x, int64_t ldx, int64_t bsx, const int32_t* tab,
                             int64_t x_rows, int32_t act, const float* slope, const float* W, const float* bias, float* y,
                             int64_t ldy, int64_t bsy, int32_t B, int32_t Fin, int32_t Fout, int32_t Fout_store,
                             int32_t rows_out, hipStream_t st, bool* launched) {
  *launched = false;
  const int32_t n = g->n;
  const int halo_on = env_int("GCL_GCN_HALO", 1);  // read per call: the parity test compares the two kernels
  static const int x3_on = env_int("GCL_X3", 1) && env_int("GCL_X3_GCN", 1);
  const gcl_halo& hl = g->halo[0][0];
  const int hp4 = hl.T == 64 ? (int)gcl::cdiv((hl.smax - 64) / 4, 4) : 99;
  const int KPh = Fin + 2;
  const size_t ldsh = (size_t)2 * (32 * KPh > 2048 ? 32 * KPh : 2048) * 4 + (size_t)(hl.smax + 1) * 256;
  // 32-bit byte offsets: inside one sample without a table, inside the whole of X with one
  const bool offs_ok = tab ? (x_rows * ldx * 4 < ((int64_t)1 << 32) && (int64_t)B * bsx * 4 < ((int64_t)1 << 32) && x_rows < (1 << 24))
                           : ((int64_t)n * ldx * 4 < ((int64_t)1 << 31));
  // only where the gather is the bulk of the layer (>= 6 edges per row: the mesh graph has 7.4): on the decoder graph
  // of the 512x256 configs (3.3 edges per row) the per-edge kernel's twelve wave-independent pipelines run 1.6x faster
  if (!(halo_on && x3_on && g->kind == GCL_GRAPH_GCN && hl.T == 64 && hp4 <= 8 && Fin % 16 == 0 && Fin > 32 && rows_out == n &&
        g->e >= 6 * (int64_t)n && ldsh <= 80 * 1024 && offs_ok && n < (1 << 24) && ldx * 4 < (1 << 24) &&
        (int64_t)n * ldy * 4 < ((int64_t)1 << 31)))
    return GCL_OK;
  if (tab && hp4 > 4) return GCL_OK;  // the table variant exists for the interleaved-issue form only
  static const int bpc_env = env_int("GCL_GCN_HALO_BPC", 0);
  const int per_cu = (int)((160 * 1024) / ldsh);
  const int Jx = 32 * (bpc_env > 0 ? bpc_env : per_cu);
  dim3 grid((unsigned)(gcl::kNumXCD * Jx));
  auto go = [&](auto kern) -> int {
    const int rc = gcl::ensure_dyn_lds((const void*)kern, ldsh);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, grid, dim3(256), ldsh, st, hl.list, hl.cnt, reinterpret_cast<const int2*>(hl.rec), g->rowptr,
                       hl.opos, g->w, hl.smax, x, ldx, bsx, slope, W, bias, y, ldy, bsy, n, B, Fin, Fout, Fout_store,
                       hl.ntiles, tab);
    return GCL_OK;
  };
  int rc;
  // hp4 <= 4 (the icosphere meshes: <= 64 halo rows per tile): non-temporal stores straight from the accumulator and
  // the next item's DMA pieces issued between the k-steps of the dense part (in the step 111.5 -> 102.6 us per mesh
  // layer; GCL_GCN_HALO_FORM=0 selects the first form - DMA issue up front, staged 16-byte stores - for comparison)
  static const int form_env = env_int("GCL_GCN_HALO_FORM", 1);
  if (tab)
    rc = act == GCL_ACT_PRELU  ? go(&gcn_halo_fwd_kernel<gcl::kActPrelu, 4, 2, true, true, true>)
         : act == GCL_ACT_SILU ? go(&gcn_halo_fwd_kernel<gcl::kActSilu, 4, 2, true, true, true>)
                               : go(&gcn_halo_fwd_kernel<gcl::kActNone, 4, 2, true, true, true>);
  else if (hp4 <= 4 && form_env)
    rc = act == GCL_ACT_PRELU  ? go(&gcn_halo_fwd_kernel<gcl::kActPrelu, 4, 2, true, true>)
         : act == GCL_ACT_SILU ? go(&gcn_halo_fwd_kernel<gcl::kActSilu, 4, 2, true, true>)
                               : go(&gcn_halo_fwd_kernel<gcl::kActNone, 4, 2, true, true>);
  else if (hp4 <= 4)
    rc = act == GCL_ACT_PRELU  ? go(&gcn_halo_fwd_kernel<gcl::kActPrelu, 4>)
         : act == GCL_ACT_SILU ? go(&gcn_halo_fwd_kernel<gcl::kActSilu, 4>)
                               : go(&gcn_halo_fwd_kernel<gcl::kActNone, 4>);
  else
    rc = act == GCL_ACT_PRELU  ? go(&gcn_halo_fwd_kernel<gcl::kActPrelu, 8, 2, true>)
         : act == GCL_ACT_SILU ? go(&gcn_halo_fwd_kernel<gcl::kActSilu, 8, 2, true>)
                               : go(&gcn_halo_fwd_kernel<gcl::kActNone, 8, 2, true>);
  if (rc) return rc;
  GCL_CHECK_LAUNCH();
  *launched = true;
  return GCL_OK;
}

extern "C" int gcl_gcn_layer_fwd(const gcl_graph_t* g, const float* x, int64_t ldx, int64_t bsx, int32_t act,
                                 const float* slope, const float* W, const float* bias, float* y, int64_t ldy,
                                 int64_t bsy, int32_t B, int32_t Fin, int32_t Fout, int32_t Fout_store,
                                 gcl_stream_t stream) {
  return gcl_gcn_layer_fwd_rows(g, x, ldx, bsx, act, slope, W, bias, y, ldy, bsy, B, Fin, Fout, Fout_store,
                                g ? g->n : 0, stream);
}

extern "C" int gcl_gcn_layer_fwd_rows(const gcl_graph_t* g, const float* x, int64_t ldx, int64_t bsx, int32_t act,
                                      const float* slope, const float* W, const float* bias, float* y, int64_t ldy,
                                      int64_t bsy, int32_t B, int32_t Fin, int32_t Fout, int32_t Fout_store,
                                      int32_t rows_out, gcl_stream_t stream) {
  GCL_CHECK_ARG(g && x && W && y, "gcn_layer_fwd: null argument");
  GCL_CHECK_ARG(g->kind == GCL_GRAPH_GCN || g->kind == GCL_GRAPH_MEAN, "gcn_layer_fwd: graph carries no edge weights");
  GCL_CHECK_ARG(B > 0 && Fin >= 4 && Fin <= 64 && Fin % 4 == 0 && Fout >= 1 && Fout <= 64,
                "gcn_layer_fwd: unsupported Fin=%d Fout=%d (Fin %% 4 == 0, both <= 64)", Fin, Fout);
  GCL_CHECK_ARG(Fout_store >= Fout && Fout_store % 4 == 0 && Fout_store <= 64 && ldy >= Fout_store,
                "gcn_layer_fwd: Fout_store=%d must be a multiple of 4 in [Fout, min(64, ldy)]", Fout_store);
  GCL_CHECK_ARG(ldx >= Fin && ldx % 4 == 0 && bsx % 4 == 0 && gcl::aligned16(x), "gcn_layer_fwd: x rows must be 16-B aligned");
  GCL_CHECK_ARG(ldy % 4 == 0 && bsy % 4 == 0 && gcl::aligned16(y), "gcn_layer_fwd: y rows must be 16-B aligned");
  GCL_CHECK_ARG(g->n_heavy == 0, "gcn_layer_fwd: the graph has rows with more than %d in-edges", gcl::kHeavy);
  GCL_CHECK_ARG(act == GCL_ACT_NONE || act == GCL_ACT_SILU || (act == GCL_ACT_PRELU && slope),
                "gcn_layer_fwd: bad activation %d", act);
  GCL_CHECK_ARG(x != y, "gcn_layer_fwd: in-place is not supported");
  const int32_t n = g->n;
  GCL_CHECK_ARG(rows_out >= 1 && rows_out <= n, "gcn_layer_fwd: rows_out=%d outside [1, n=%d]", rows_out, n);
  hipStream_t st = (hipStream_t)stream;
  {
    bool launched = false;
    const int rc = halo_layer_launch(g, x, ldx, bsx, nullptr, 0, act, slope, W, bias, y, ldy, bsy, B, Fin, Fout, Fout_store,
                                     rows_out, st, &launched);
    if (rc || launched) return rc;
  }
  const int32_t nRT = (int32_t)gcl::cdiv(rows_out, 32);  // only the tiles that hold requested rows are computed
  const int NS = Fout_store > 32 ? 2 : 1;
  constexpr int NW12 = 12, NW8 = 8;
  const int KP = Fin + 2;
  static const int x3_env = env_int("GCL_X3", 1) && env_int("GCL_X3_GCN", 1);
  const size_t wave1_b = (size_t)wave_region_f(32 * (KP > NS * 32 ? KP : NS * 32)) * sizeof(float);
  // the split-operand variant holds its A fragments in registers: 8 waves per block (256 VGPRs each) instead of 12
  const bool x3 = x3_env != 0 && (Fin % 16 == 0);
  const int NWr = x3 ? NW8 : NW12;
  const size_t lds = (x3 ? (size_t)3 * NS * 32 * kWRowB : (size_t)NS * 32 * KP * sizeof(float)) + NWr * wave1_b;
  GCL_CHECK_ARG((int64_t)n * ldx * 4 < (int64_t)1 << 31 && n < (1 << 24) && ldx * 4 < (1 << 24),
                "gcn_layer_fwd: one sample of x must stay below 2 GiB (n, row bytes < 2^24)");
  const int64_t tiles = (int64_t)B * nRT;
  int64_t grid = gcl::kNumCU;
  if (tiles < grid * NWr) grid = gcl::cdiv(tiles, NWr);
  if (B >= gcl::kNumXCD) grid = gcl::cdiv(grid, gcl::kNumXCD) * gcl::kNumXCD;  // XCD-aware schedule needs a multiple of 8
  const int ewidth = g->ell_cover;  // smallest prefix width that covers (almost) every row: the fix-up loop is the slow path
#define GCL_GF4(NS_, ACT_, EW_)                \
  do {                                         \
    if (x3) GCL_GF5(NS_, ACT_, EW_, true);     \
    else GCL_GF5(NS_, ACT_, EW_, false);       \
  } while (0)
#define GCL_GF5(NS_, ACT_, EW_, X3_)                                                                                 \
  do {                                                                                                              \
    constexpr int NW = X3_ ? NW8 : NW12;                                                                            \
    auto kern = gcn_fwd_kernel<NS_, ACT_, EW_, NW, false, X3_>;                                                              \
    { const int lrc_ = gcl::ensure_dyn_lds((const void*)kern, 160 * 1024); if (lrc_) return lrc_; } \
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NW * 64), lds, st, g->rowptr, g->col, g->w, g->ecol, g->ew, x, \
                       ldx, bsx, slope, W, bias, y, ldy, bsy, n, B, Fin, Fout, Fout_store, nRT, rows_out);                     \
  } while (0)
#define GCL_GF3(NS_, ACT_)                  \
  do {                                      \
    switch (ewidth) {                       \
      case 8: GCL_GF4(NS_, ACT_, 8); break; \
      case 4: GCL_GF4(NS_, ACT_, 4); break; \
      default: GCL_GF4(NS_, ACT_, 2); break; \
    }                                       \
  } while (0)
#define GCL_GF2(NS_)                                         \
  do {                                                       \
    if (act == GCL_ACT_PRELU) GCL_GF3(NS_, gcl::kActPrelu);  \
    else if (act == GCL_ACT_SILU) GCL_GF3(NS_, gcl::kActSilu); \
    else GCL_GF3(NS_, gcl::kActNone);                        \
  } while (0)
  if (g->kind == GCL_GRAPH_MEAN) {  // no self-loops: masked (SAFE) variant, activation-free, widest prefix
    GCL_CHECK_ARG(act == GCL_ACT_NONE, "gcn_layer_fwd: mean-aggregation graphs take no activation");
    constexpr int NW = NW12;
    const size_t lds = (size_t)NS * 32 * KP * sizeof(float) + NW * wave1_b;
    int64_t grid = gcl::kNumCU;
    if (tiles < grid * NW) grid = gcl::cdiv(tiles, NW);
    if (B >= gcl::kNumXCD) grid = gcl::cdiv(grid, gcl::kNumXCD) * gcl::kNumXCD;
    if (NS == 2) {
      auto kern = gcn_fwd_kernel<2, gcl::kActNone, 8, NW, true, false>;
      { const int lrc_ = gcl::ensure_dyn_lds((const void*)kern, 160 * 1024); if (lrc_) return lrc_; }
      hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NW * 64), lds, st, g->rowptr, g->col, g->w, g->ecol, g->ew, x, ldx,
                         bsx, slope, W, bias, y, ldy, bsy, n, B, Fin, Fout, Fout_store, nRT, rows_out);
    } else {
      auto kern = gcn_fwd_kernel<1, gcl::kActNone, 8, NW, true, false>;
      { const int lrc_ = gcl::ensure_dyn_lds((const void*)kern, 160 * 1024); if (lrc_) return lrc_; }
      hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NW * 64), lds, st, g->rowptr, g->col, g->w, g->ecol, g->ew, x, ldx,
                         bsx, slope, W, bias, y, ldy, bsy, n, B, Fin, Fout, Fout_store, nRT, rows_out);
    }
  } else if (NS == 2) GCL_GF2(2);
  else GCL_GF2(1);
#undef GCL_GF2
#undef GCL_GF3
#undef GCL_GF4
#undef GCL_GF5
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

// The layer with its input rows read through a row table (gcn_halo_fwd_kernel<.., TAB>): row i of sample b is row
// tab[i] of sample b of x when tab[i] >= 0, the batch-invariant flat row ~tab[i] of x otherwise.  Only the source-tile
// form supports it: GCL_EUNSUPPORTED elsewhere (gcl_gcn_layer_fwd_tab_ok tells beforehand), and the caller
// materialises the rows (gcl_gather2_rows) and calls gcl_gcn_layer_fwd instead.
extern "C" int gcl_gcn_layer_fwd_tab(const gcl_graph_t* g, const float* x, int64_t ldx, int64_t bsx, int64_t x_rows,
                                     const int32_t* tab, int32_t act, const float* slope, const float* W,
                                     const float* bias, float* y, int64_t ldy, int64_t bsy, int32_t B, int32_t Fin,
                                     int32_t Fout, int32_t Fout_store, gcl_stream_t stream) {
  GCL_CHECK_ARG(g && x && W && y && tab, "gcn_layer_fwd_tab: null argument");
  GCL_CHECK_ARG(g->kind == GCL_GRAPH_GCN, "gcn_layer_fwd_tab: graph carries no GCN edge weights");
  GCL_CHECK_ARG(B > 0 && Fin >= 4 && Fin <= 64 && Fin % 4 == 0 && Fout >= 1 && Fout <= 64,
                "gcn_layer_fwd_tab: unsupported Fin=%d Fout=%d (Fin %% 4 == 0, both <= 64)", Fin, Fout);
  GCL_CHECK_ARG(Fout_store >= Fout && Fout_store % 4 == 0 && Fout_store <= 64 && ldy >= Fout_store,
                "gcn_layer_fwd_tab: Fout_store=%d must be a multiple of 4 in [Fout, min(64, ldy)]", Fout_store);
  GCL_CHECK_ARG(ldx >= Fin && ldx % 4 == 0 && bsx % 4 == 0 && gcl::aligned16(x) && x_rows >= 1,
                "gcn_layer_fwd_tab: x rows must be 16-B aligned");
  GCL_CHECK_ARG(ldy % 4 == 0 && bsy % 4 == 0 && gcl::aligned16(y), "gcn_layer_fwd_tab: y rows must be 16-B aligned");
  GCL_CHECK_ARG(act == GCL_ACT_NONE || act == GCL_ACT_SILU || (act == GCL_ACT_PRELU && slope),
                "gcn_layer_fwd_tab: bad activation %d", act);
  bool launched = false;
  const int rc = halo_layer_launch(g, x, ldx, bsx, tab, x_rows, act, slope, W, bias, y, ldy, bsy, B, Fin, Fout, Fout_store,
                                   g->n, (hipStream_t)stream, &launched);
  if (rc) return rc;
  if (!launched) {
    gcl::set_error("gcn_layer_fwd_tab: this graph / shape has no source-tile form (Fin=%d Fout=%d)", Fin, Fout);
    return GCL_EUNSUPPORTED;
  }
  return GCL_OK;
}

extern "C" int gcl_gcn_layer_fwd_tab_ok(const gcl_graph_t* g, int64_t ldx, int64_t bsx, int64_t x_rows, int32_t B,
                                        int32_t Fin, int32_t Fout) {
  if (!g || g->kind != GCL_GRAPH_GCN || g->n_heavy) return 0;
  static const int x3_on = env_int("GCL_X3", 1) && env_int("GCL_X3_GCN", 1);
  const gcl_halo& hl = g->halo[0][0];
  if (!env_int("GCL_GCN_HALO", 1) || !x3_on || hl.T != 64) return 0;
  const int hp4 = (int)gcl::cdiv((hl.smax - 64) / 4, 4);
  const size_t ldsh = (size_t)2 * (32 * (Fin + 2) > 2048 ? 32 * (Fin + 2) : 2048) * 4 + (size_t)(hl.smax + 1) * 256;
  return hp4 <= 4 && Fin % 16 == 0 && Fin > 32 && Fin <= 64 && Fout >= 1 && Fout <= 64 && g->e >= 6 * (int64_t)g->n &&
                 ldsh <= 80 * 1024 && x_rows * ldx * 4 < ((int64_t)1 << 32) && (int64_t)B * bsx * 4 < ((int64_t)1 << 32) &&
                 x_rows < (1 << 24) && g->n < (1 << 24) && ldx * 4 < (1 << 24)
             ? 1
             : 0;
}
