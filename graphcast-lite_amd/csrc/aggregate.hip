// CSR segmented reduction  y[b,i,:] = sum_{e in row i} w_e * h[b, col_e, :] (+ bias)
//
// Replaces PyG propagate's index_select -> multiply -> scatter_add_ (GCNConv src/models.py:419,
// SimpleConv mean src/models.py:414).  HBM-bound: per (sample,row) it reads deg rows of h (all but
// the first touch of each row served from the XCD's L2) and writes one row of y.
//
// Mapping (wave64): a destination row is owned by LPR lanes, each lane holding 4 consecutive
// channels (one 16-B load per neighbour row per lane = the widest coalesced access); a wave
// carries 64/LPR destination rows, a 256-thread block 4 waves.  Column indices and weights of a
// row are fetched LPR at a time (one coalesced load per lane group) and broadcast with
// ds_bpermute, so the inner loop issues up to 4 independent 16-B neighbour loads per lane before
// the first FMA.  Rows keep PyG edge order inside the CSR, so the sum runs in the reference's
// order (self-loop last).
//
// XCD placement: blocks are dealt round-robin over the 8 XCDs (MI355X_MICROARCH.md), so block id
// b&7 names an XCD group.  All row-blocks of sample s are given ids congruent to s mod 8: one
// sample's h (2.6 MB at 64x32/F=64) then lives in ONE 4 MiB L2 and the ~7x neighbour re-reads
// never leave the XCD.  Placement is a speed choice only; results do not depend on it.
#include <stdlib.h>

#include <algorithm>

#include "common.h"
#include "halo.h"

namespace {

// EW = number of ELL-prefix neighbours every row reads unconditionally (1, 2, 4 or 8; chosen per
// graph so that most rows fit).  Per batch the dependency chain is: {rowptr pair, ELL entries}
// (one round trip, prefetched one batch ahead) -> EW neighbour rows issued together -> store.
// Rows with more than EW in-edges finish in a CSR loop (wave-uniform test).  Padded slots load a
// valid row and are masked with a select (not a multiply), so non-finite values cannot leak.
template <int LPR, bool VL, bool VS, int EW, int ITER>
__global__ __launch_bounds__(256) void agg_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                  const float* __restrict__ w, const int32_t* __restrict__ ecol,
                                                  const float* __restrict__ ew, const float* __restrict__ H,
                                                  int64_t ldh, int64_t bsh, const float* __restrict__ bias,
                                                  float* __restrict__ Y, int64_t ldy, int64_t bsy, int32_t n,
                                                  int32_t B, int32_t F, int32_t nRB, int32_t xcd_map,
                                                  int32_t nt_store, const int32_t* __restrict__ order16) {
  constexpr int RPW = 64 / LPR;
  constexpr int EL = LPR < gcl::kEll ? LPR : gcl::kEll;  // ELL entries a lane group can hold
  static_assert(EW <= EL, "ELL width exceeds the lanes of a row group");
  const int bid = blockIdx.x;
  int b, rb;
  if (xcd_map) {
    const int xcd = bid & (gcl::kNumXCD - 1);
    const int slot = bid >> 3;
    b = xcd + gcl::kNumXCD * (slot / nRB);
    rb = slot % nRB;
  } else {
    b = bid / nRB;
    rb = bid % nRB;
  }
  if (b >= B) return;
  if (order16) {  // the k-th block runs the rows of the k-th 16-row group of the graph's processing order (common.h)
    constexpr int RPB = RPW * 4 * ITER, K = 16 / RPB;  // launched only with RPB <= 16
    rb = order16[rb / (K > 0 ? K : 1)] * (K > 0 ? K : 1) + rb % (K > 0 ? K : 1);
  }
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int sub = lane / LPR;
  const int l = lane % LPR;
  const int gbase = sub * LPR;  // first lane of this row's group
  const int c0 = l * 4;
  const bool cactive = c0 < F;
  const int cc = cactive ? c0 : 0;  // inactive channel lanes re-read channel 0 (never stored)
  const float* __restrict__ Hb = H + (int64_t)b * bsh;
  float* __restrict__ Yb = Y + (int64_t)b * bsy;

  float bz0 = 0.f, bz1 = 0.f, bz2 = 0.f, bz3 = 0.f;
  if (bias && cactive) {
    bz0 = bias[c0];
    if (VS || c0 + 1 < F) bz1 = bias[c0 + 1];
    if (VS || c0 + 2 < F) bz2 = bias[c0 + 2];
    if (VS || c0 + 3 < F) bz3 = bias[c0 + 3];
  }

  auto ld4 = [&](int j, float& x0, float& x1, float& x2, float& x3) {
    const float* p = Hb + (int64_t)j * ldh + cc;
    if (VL) {
      const float4 v = *reinterpret_cast<const float4*>(p);
      x0 = v.x; x1 = v.y; x2 = v.z; x3 = v.w;
    } else {
      x0 = p[0];
      x1 = (cc + 1 < F) ? p[1] : 0.f;
      x2 = (cc + 2 < F) ? p[2] : 0.f;
      x3 = (cc + 3 < F) ? p[3] : 0.f;
    }
  };

  // metadata of one batch: loaded unconditionally (rows past the end are clamped, masked at store)
  const int row_first = (rb * 4 + wave) * (RPW * ITER) + sub;
  auto meta = [&](int row, int& start, int& end, int& cj, float& wj) {
    const int rc = row < n ? row : n - 1;
    start = rowptr[rc];
    end = rowptr[rc + 1];
    cj = ecol[(int64_t)rc * gcl::kEll + (l & (EL - 1))];
    wj = ew[(int64_t)rc * gcl::kEll + (l & (EL - 1))];
  };
  int start, end, cj;
  float wj;
  meta(row_first, start, end, cj, wj);

#pragma unroll 1
  for (int it = 0; it < ITER; ++it) {
    const int row = row_first + it * RPW;
    int nstart = 0, nend = 0, ncj = 0;
    float nwj = 0.f;
    if (it + 1 < ITER) meta(row + RPW, nstart, nend, ncj, nwj);  // next batch, in flight during this one

    const int deg_all = end - start;
    const bool heavy = deg_all > gcl::kHeavy;  // done by agg_heavy_kernel (one block per row)
    const int deg = heavy ? 0 : deg_all;
    if (heavy) end = start;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    {
      int jj[EW];
      float ww[EW], v0[EW], v1[EW], v2[EW], v3[EW];
#pragma unroll
      for (int k = 0; k < EW; ++k) {
        jj[k] = __shfl(cj, gbase + k, 64);
        ww[k] = __shfl(wj, gbase + k, 64);
      }
#pragma unroll
      for (int k = 0; k < EW; ++k) ld4(jj[k], v0[k], v1[k], v2[k], v3[k]);  // EW rows in flight
#pragma unroll
      for (int k = 0; k < EW; ++k) {
        const bool in = k < deg;
        a0 += in ? ww[k] * v0[k] : 0.f;
        a1 += in ? ww[k] * v1[k] : 0.f;
        a2 += in ? ww[k] * v2[k] : 0.f;
        a3 += in ? ww[k] * v3[k] : 0.f;
      }
    }
    if (__any(deg > EW)) {  // wave-uniform: some row of this wave has more edges than the prefix
      for (int base = start + EW; base < end; base += LPR) {
        const int mine = base + l;
        int oj = 0;
        float ow = 0.f;
        if (mine < end) {
          oj = col[mine];
          ow = w[mine];
        }
        const int cnt = min(LPR, end - base);
        for (int k = 0; k < cnt; k += 4) {
          const int j0 = __shfl(oj, gbase + k, 64);
          const int j1 = __shfl(oj, gbase + ((k + 1) & (LPR - 1)), 64);
          const int j2 = __shfl(oj, gbase + ((k + 2) & (LPR - 1)), 64);
          const int j3 = __shfl(oj, gbase + ((k + 3) & (LPR - 1)), 64);
          const float w0 = __shfl(ow, gbase + k, 64);
          const float w1 = __shfl(ow, gbase + ((k + 1) & (LPR - 1)), 64);
          const float w2 = __shfl(ow, gbase + ((k + 2) & (LPR - 1)), 64);
          const float w3 = __shfl(ow, gbase + ((k + 3) & (LPR - 1)), 64);
          float p0 = 0, p1 = 0, p2 = 0, p3 = 0, q0 = 0, q1 = 0, q2 = 0, q3 = 0;
          float r0 = 0, r1 = 0, r2 = 0, r3 = 0, s0 = 0, s1 = 0, s2 = 0, s3 = 0;
          ld4(j0, p0, p1, p2, p3);
          if (k + 1 < cnt) ld4(j1, q0, q1, q2, q3);
          if (k + 2 < cnt) ld4(j2, r0, r1, r2, r3);
          if (k + 3 < cnt) ld4(j3, s0, s1, s2, s3);
          a0 += w0 * p0; a1 += w0 * p1; a2 += w0 * p2; a3 += w0 * p3;
          if (k + 1 < cnt) { a0 += w1 * q0; a1 += w1 * q1; a2 += w1 * q2; a3 += w1 * q3; }
          if (k + 2 < cnt) { a0 += w2 * r0; a1 += w2 * r1; a2 += w2 * r2; a3 += w2 * r3; }
          if (k + 3 < cnt) { a0 += w3 * s0; a1 += w3 * s1; a2 += w3 * s2; a3 += w3 * s3; }
        }
      }
    }
    if (row < n && cactive && !heavy) {
      a0 += bz0; a1 += bz1; a2 += bz2; a3 += bz3;
      float* __restrict__ yp = Yb + (int64_t)row * ldy + c0;
      if (VS) {
        // streamed-out rows are not re-read by this kernel: keep them from evicting the gathered
        // rows of h out of the XCD's L2 (non-temporal store)
        if (nt_store) {  // ONE 16-byte non-temporal store per lane (four dword stores write partial sectors)
          typedef float v4f __attribute__((ext_vector_type(4)));
          v4f v = {a0, a1, a2, a3};
          __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(yp));
        } else {
          *reinterpret_cast<float4*>(yp) = make_float4(a0, a1, a2, a3);
        }
      } else {
        yp[0] = a0;
        if (c0 + 1 < F) yp[1] = a1;
        if (c0 + 2 < F) yp[2] = a2;
        if (c0 + 3 < F) yp[3] = a3;
      }
    }
    start = nstart; end = nend; cj = ncj; wj = nwj;
  }
}

// One block per (heavy row, sample): the 256/LPR lane groups stride over the row's edges (4 loads
// in flight each) and are combined through LDS in a fixed order.  Only rows with more than kHeavy
// edges take this path (polar mesh nodes of E_G2M / E_M2G at 512x256: up to 943 edges), so a
// single wave no longer serialises hundreds of dependent gathers at the tail of the launch.
template <int LPR, bool VL, bool VS>
__global__ __launch_bounds__(256) void agg_heavy_kernel(const int32_t* __restrict__ rows_heavy,
                                                        const int32_t* __restrict__ rowptr,
                                                        const int32_t* __restrict__ col, const float* __restrict__ w,
                                                        const float* __restrict__ H, int64_t ldh, int64_t bsh,
                                                        const float* __restrict__ bias, float* __restrict__ Y,
                                                        int64_t ldy, int64_t bsy, int32_t F) {
  constexpr int NG = 256 / LPR;
  __shared__ float red[NG][LPR * 4 + 1];
  const int row = rows_heavy[blockIdx.x];
  const int b = blockIdx.y;
  const int g = threadIdx.x / LPR, l = threadIdx.x % LPR;
  const int c0 = l * 4;
  const bool cactive = c0 < F;
  const int cc = cactive ? c0 : 0;
  const float* __restrict__ Hb = H + (int64_t)b * bsh;
  const int start = rowptr[row], end = rowptr[row + 1];
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  auto ld4 = [&](int j, float& x0, float& x1, float& x2, float& x3) {
    const float* p = Hb + (int64_t)j * ldh + cc;
    if (VL) {
      const float4 v = *reinterpret_cast<const float4*>(p);
      x0 = v.x; x1 = v.y; x2 = v.z; x3 = v.w;
    } else {
      x0 = p[0];
      x1 = (cc + 1 < F) ? p[1] : 0.f;
      x2 = (cc + 2 < F) ? p[2] : 0.f;
      x3 = (cc + 3 < F) ? p[3] : 0.f;
    }
  };
  for (int e = start + g; e < end; e += 4 * NG) {
    const int e1 = e + NG, e2 = e + 2 * NG, e3 = e + 3 * NG;
    const int j0 = col[e], j1 = col[e1 < end ? e1 : e], j2 = col[e2 < end ? e2 : e], j3 = col[e3 < end ? e3 : e];
    const float w0 = w[e], w1 = e1 < end ? w[e1] : 0.f, w2 = e2 < end ? w[e2] : 0.f, w3 = e3 < end ? w[e3] : 0.f;
    float p0, p1, p2, p3, q0, q1, q2, q3, r0, r1, r2, r3, s0, s1, s2, s3;
    ld4(j0, p0, p1, p2, p3);
    ld4(j1, q0, q1, q2, q3);
    ld4(j2, r0, r1, r2, r3);
    ld4(j3, s0, s1, s2, s3);
    a0 += w0 * p0; a1 += w0 * p1; a2 += w0 * p2; a3 += w0 * p3;
    if (e1 < end) { a0 += w1 * q0; a1 += w1 * q1; a2 += w1 * q2; a3 += w1 * q3; }
    if (e2 < end) { a0 += w2 * r0; a1 += w2 * r1; a2 += w2 * r2; a3 += w2 * r3; }
    if (e3 < end) { a0 += w3 * s0; a1 += w3 * s1; a2 += w3 * s2; a3 += w3 * s3; }
  }
  red[g][c0] = a0; red[g][c0 + 1] = a1; red[g][c0 + 2] = a2; red[g][c0 + 3] = a3;
  __syncthreads();
  if (g == 0 && cactive) {
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
#pragma unroll
    for (int q = 0; q < NG; ++q) { t0 += red[q][c0]; t1 += red[q][c0 + 1]; t2 += red[q][c0 + 2]; t3 += red[q][c0 + 3]; }
    if (bias) {
      t0 += bias[c0];
      if (c0 + 1 < F) t1 += bias[c0 + 1];
      if (c0 + 2 < F) t2 += bias[c0 + 2];
      if (c0 + 3 < F) t3 += bias[c0 + 3];
    }
    float* yp = Y + (int64_t)b * bsy + (int64_t)row * ldy + c0;
    if (VS) {
      *reinterpret_cast<float4*>(yp) = make_float4(t0, t1, t2, t3);
    } else {
      yp[0] = t0;
      if (c0 + 1 < F) yp[1] = t1;
      if (c0 + 2 < F) yp[2] = t2;
      if (c0 + 3 < F) yp[3] = t3;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Source-tile ("halo") aggregation.  agg_kernel gathers every edge's source row through the L1/TA path: B*E' rows
// per launch, ~3.4x the algorithmic bytes on the mesh graph, and that path (not HBM) bounds it.  Here a block owns
// a tile of T consecutive rows at a time: it stages the tile's own rows and the DISTINCT source rows outside the
// tile (its halo; the list is precomputed per graph: gcl_halo) once by LDS-DMA and forms every sum from LDS, so the
// vector-memory path carries each source once per tile instead of once per edge.  With tiles that share sources well
// (mesh nodes numbered by recursive coordinate bisection: 1.7 staged rows per destination row at T = 64 instead of
// 7.4 edge reads) the kernel runs at 0.74 of the 8 TB/s HBM roofline instead of 0.50-0.58.
//
// Sums: a row is owned by LPR lanes (4 channels each), its first 16 edge records {image position, weight} sit one per
// lane and are broadcast with DPP row_newbcast (no LDS traffic; broadcast + address is ONE v_add_u32_dpp), the source
// values come from ds_read_b128 (conflict-free for any row set: 16 lanes x 16 B span all 64 banks).  Slots 0..7 are
// read unconditionally (padding points at a zero row with weight 0), slots 8..15 when some row of the wave-instruction
// has that many edges, rows with more than 16 edges finish from the CSR arrays.
// Arithmetic is that of agg_kernel<.., EW = 8>, bit for bit: product and sum rounded separately for the first eight
// edges of a row (PyG's multiply, then scatter_add), fused multiply-add for the rest, in CSR order.
//
// What was tried on the way (profiles/r03_halo_*.txt, tools/probes/tile_copy_probe.hip): one tile per block (64 us:
// half of a block's life passes before its last load is issued), that form with 8 / 16 waves per tile (71 / 88 us),
// two images per block with the next tile's DMA under this tile's sums (62 us at two blocks per CU: sums at two waves
// per SIMD run at single-wave issue rate), a loader wave + summing waves (69 us: one wave issues 30 DMAs per tile one
// after the other), fused multiply-add for all slots (no change: the sums are not bound by the FP pipe).  A probe with
// this structure but no lists, synthetic records and exactly 8 slots copies the same bytes in 50 us; re-reading a
// tile's 128-byte-per-row records for every sample costs 17 us of that difference and the scalar-loaded list 6 us -
// which is why a block here keeps both in registers across the samples of its XCD group.
// ---------------------------------------------------------------------------------------------------------
using gcl::halo::mul_then_add;
using gcl::halo::row_bcast;
typedef gcl::halo::gptr_t gcl_gptr_t;
typedef gcl::halo::lptr_t gcl_lptr_t;

#ifdef GCL_STAMPS
__device__ unsigned long long agg_stamps[8 * 4096];  // diagnostic builds only (make STAMPS=1): per wave, cycles per phase
#define GCL_AGG_STAMP(i)                                   \
  {                                                        \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
    st_acc[i] += t_ - st_last;                             \
    st_last = t_;                                          \
  }
#else
#define GCL_AGG_STAMP(i)
#endif

// ---------------------------------------------------------------------------------------------------------
// The same aggregation with PERSISTENT blocks.  Stamps of agg_halo_kernel (tools/stamps_agg.py) put half of a
// block's life before its last load is issued - kernel arguments, block -> (sample, tile) arithmetic, the tile's
// list behind a scalar-load round trip - and LDS (one 31 KB image per block) caps a CU at five blocks, so that
// start-up is paid in throughput.  A probe with the kernel's structure (tools/probes/tile_copy_probe.hip) adds
// that re-reading a tile's edge records (128 B per row) for every sample costs 17 us of a 67 us launch and the
// scalar-loaded list 6 us, although both are the same for all samples.  So a block here walks a CONTIGUOUS range
// of the (tile, sample) items of its XCD group, tile-major: arguments and addressing are set up once, a tile's
// list entries and edge records are fetched once and kept in registers for all of the group's samples, and the
// stores of one item are still in flight while the loads of the next are issued.  One image per block, two
// barriers per item (loaded / free again).
// ---------------------------------------------------------------------------------------------------------
template <int LPR, int T, int MAXPW>
__global__ __launch_bounds__(256) void agg_halo_loop_kernel(const int32_t* __restrict__ list, const int32_t* __restrict__ cnt,
                                                            const int2* __restrict__ rec, const int32_t* __restrict__ rowptr,
                                                            const int32_t* __restrict__ opos, const float* __restrict__ w,
                                                            int32_t smax, const float* __restrict__ H, int64_t ldh,
                                                            int64_t bsh, const float* __restrict__ bias,
                                                            float* __restrict__ Y, int64_t ldy, int64_t bsy, int32_t n,
                                                            int32_t B, int32_t F, int32_t ntiles, int32_t nt_store) {
  extern __shared__ float4 img[];  // (smax + 1) * LPR float4: own rows, halo rows, zero row
  constexpr int RPW = 64 / LPR;
  constexpr int NW = 4;              // waves per block
  constexpr int NIT = T / NW / RPW;  // row groups (wave-instructions) per wave and tile
  constexpr int SH = LPR == 16 ? 8 : LPR == 32 ? 9 : 10;  // log2 of the bytes of an image row
  typedef float v4f __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) const v4f* lds4_t;
  const int xcd = blockIdx.x & (gcl::kNumXCD - 1);
  const int J = gridDim.x >> 3, j = blockIdx.x >> 3;
  const int nsamp = (B - xcd + gcl::kNumXCD - 1) / gcl::kNumXCD;  // samples of this XCD group: xcd, xcd + 8, ...
  const int items = nsamp * ntiles;                               // item = tile * nsamp + sample index
  // contiguous share of block j: items [m, mend) (the first items % J blocks take one more)
  const int base = items / J, extra = items - base * J;
  int m = j * base + min(j, extra);
  const int mend = m + base + (j < extra ? 1 : 0);
  if (m >= mend) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int sub = lane / LPR;
  const int l = lane % LPR;
  const int c0 = l * 4;
  const bool cactive = c0 < F;
  const unsigned cb = (unsigned)(cactive ? c0 : 0) * 4u;
  const unsigned ldb = (unsigned)ldh * 4u;
  const int hstride = smax - T;
  const unsigned lb = (unsigned)(size_t)((gcl_lptr_t)img) + (unsigned)l * 16u;
  const int zrow = smax << SH;

  if (threadIdx.x < LPR) img[smax * LPR + threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
  float bz0 = 0.f, bz1 = 0.f, bz2 = 0.f, bz3 = 0.f;
  if (bias && cactive) {
    const float4 bv = *reinterpret_cast<const float4*>(bias + c0);
    bz0 = bv.x; bz1 = bv.y; bz2 = bv.z; bz3 = bv.w;
  }

  // list entries of a tile for this wave's halo pieces (scalar loads: wave-uniform addresses)
  auto fetch_list = [&](int tile, int (&jj)[MAXPW], int& nhalo) {
    nhalo = cnt[tile] / RPW;
    const int32_t* __restrict__ tl = list + (int64_t)tile * hstride;
#pragma unroll
    for (int q = 0; q < MAXPW; ++q) {
      const int e0 = min((wave + NW * q) * RPW, hstride - RPW);
      int j = tl[e0];
#pragma unroll
      for (int r = 1; r < RPW; ++r) {
        const int jr = tl[e0 + r];
        j = sub == r ? jr : j;
      }
      jj[q] = j;
    }
  };
  int jj[MAXPW], nhalo = 0, tile = -1;
  int2 rc[NIT];
#ifdef GCL_STAMPS
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime();
#endif

  while (true) {
    const int tnew = m / nsamp;
    const int s = m - tnew * nsamp;
    const int b = xcd + gcl::kNumXCD * s;
    const bool newtile = tnew != tile;  // wave-uniform: its list entries and edge records serve all samples of the group
    if (newtile) {
      tile = tnew;
      fetch_list(tile, jj, nhalo);
    }
    GCL_AGG_STAMP(0)  // new tile: list + records
    const char* Hc = reinterpret_cast<const char*>(H + (int64_t)b * bsh);
    float* __restrict__ Yb = Y + (int64_t)b * bsy;
    const int row0 = tile * T + wave * (T / NW) + sub;
    // stage: own rows, halo rows (LDS-DMA)
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int row = row0 + it * RPW;
      const char* src = Hc + (__umul24(row < n ? row : n - 1, ldb) + cb);
      __builtin_amdgcn_global_load_lds((gcl_gptr_t)src, (gcl_lptr_t)(img + (wave * (T / NW) + it * RPW) * LPR), 16, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < MAXPW; ++q) {
      const int p = wave + NW * q;
      if (p < nhalo) {
        const char* src = Hc + (__umul24(jj[q], ldb) + cb);
        __builtin_amdgcn_global_load_lds((gcl_gptr_t)src, (gcl_lptr_t)(img + (T + p * RPW) * LPR), 16, 0, 0);
      }
    }
    if (newtile) {  // records AFTER the DMAs: hipcc waits for them where this block ends, i.e. together with the tile
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int row = row0 + it * RPW;
        rc[it] = rec[(int64_t)(row < n ? row : n - 1) * gcl::kHaloRec + (l & (gcl::kHaloRec - 1))];
      }
    }
    GCL_AGG_STAMP(1)  // DMA issue
    __syncthreads();  // vmcnt(0) (this wave's DMA, new records; the previous item's stores as well) + barrier: image complete
    GCL_AGG_STAMP(2)  // landed + barrier

#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int row = row0 + it * RPW;
      const int rx = rc[it].x, rw = rc[it].y;
      const int rxb = (rx & gcl::kHaloPosMask) << SH;
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#define GCL_HALO_MULADD(K)                                          \
  {                                                                 \
    const unsigned ad = (unsigned)row_bcast<K>(rxb) + lb;           \
    const float wk = __int_as_float(row_bcast<K>(rw));              \
    const v4f v = *(lds4_t)ad;                                      \
    a0 = mul_then_add(wk, v.x, a0);                                 \
    a1 = mul_then_add(wk, v.y, a1);                                 \
    a2 = mul_then_add(wk, v.z, a2);                                 \
    a3 = mul_then_add(wk, v.w, a3);                                 \
  }
#define GCL_HALO_FMA(K)                                             \
  {                                                                 \
    const unsigned ad = (unsigned)row_bcast<K>(rxb) + lb;           \
    const float wk = __int_as_float(row_bcast<K>(rw));              \
    const v4f v = *(lds4_t)ad;                                      \
    a0 = __fmaf_rn(wk, v.x, a0);                                    \
    a1 = __fmaf_rn(wk, v.y, a1);                                    \
    a2 = __fmaf_rn(wk, v.z, a2);                                    \
    a3 = __fmaf_rn(wk, v.w, a3);                                    \
  }
      GCL_HALO_MULADD(0) GCL_HALO_MULADD(1) GCL_HALO_MULADD(2) GCL_HALO_MULADD(3)
      GCL_HALO_MULADD(4) GCL_HALO_MULADD(5) GCL_HALO_MULADD(6) GCL_HALO_MULADD(7)
      const int last = row_bcast<15>(rx);
      if (__any(row_bcast<8>(rxb) != zrow)) {
        GCL_HALO_FMA(8) GCL_HALO_FMA(9) GCL_HALO_FMA(10) GCL_HALO_FMA(11)
        GCL_HALO_FMA(12) GCL_HALO_FMA(13) GCL_HALO_FMA(14) GCL_HALO_FMA(15)
        if (__any((last & gcl::kHaloMore) != 0)) {
          const int rcl = row < n ? row : n - 1;
          const int end = (last & gcl::kHaloMore) ? rowptr[rcl + 1] : 0;
          for (int e = rowptr[rcl] + gcl::kHaloRec; e < end; ++e) {
            const float wk = w[e];
            const v4f v = *(lds4_t)(((unsigned)opos[e] << SH) + lb);
            a0 = __fmaf_rn(wk, v.x, a0);
            a1 = __fmaf_rn(wk, v.y, a1);
            a2 = __fmaf_rn(wk, v.z, a2);
            a3 = __fmaf_rn(wk, v.w, a3);
          }
        }
      }
#undef GCL_HALO_MULADD
#undef GCL_HALO_FMA
      if (row < n && cactive && !(last & gcl::kHaloSkip)) {
        a0 += bz0; a1 += bz1; a2 += bz2; a3 += bz3;
        float* __restrict__ yp = Yb + (int64_t)row * ldy + c0;
        v4f v = {a0, a1, a2, a3};
        if (nt_store) __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(yp));
        else *reinterpret_cast<v4f*>(yp) = v;
      }
    }
    GCL_AGG_STAMP(3)  // sums + store issue
#ifdef GCL_STAMPS
    st_acc[5] += 1;
#endif
    if (++m >= mend) break;
    __syncthreads();  // every wave is done reading the image
    GCL_AGG_STAMP(4)  // image free again
  }
#ifdef GCL_STAMPS
  if (lane == 0 && (blockIdx.x * NW + wave) < 4096)
    for (int i = 0; i < 8; ++i) agg_stamps[(blockIdx.x * NW + wave) * 8 + i] = st_acc[i];
#endif
}

#ifdef GCL_STAMPS
extern "C" int gcl_debug_read_agg_stamps(unsigned long long* host_out, int count) {
  GCL_CHECK_HIP(hipDeviceSynchronize());
  GCL_CHECK_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(agg_stamps), sizeof(unsigned long long) * count));
  return GCL_OK;
}
#endif

int agg_env(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

struct AggArgs {
  const int32_t *rowptr, *col, *ecol, *heavy;
  const float *w, *ew;
  int ell_width, n_heavy;
  const gcl_halo* halo;  // [2]: T = 64, T = 32 (T == 0: not built)
  const int32_t* order16 = nullptr;  // processing order of 16-row groups (common.h), or nullptr
  int32_t n_order16 = 0;
};

// Source-tile path: returns GCL_OK after launching, or -1 when this call is not eligible (agg_kernel runs instead).
template <int LPR>
int launch_agg_halo(const AggArgs& ga, const float* h, int64_t ldh, int64_t bsh, const float* bias, float* y,
                    int64_t ldy, int64_t bsy, int32_t n, int32_t B, int32_t F, hipStream_t st) {
  const int enabled = agg_env("GCL_AGG_HALO", 1);  // read per call: the parity tests switch it to compare the two kernels
  static const int force_t = agg_env("GCL_AGG_HALO_T", 0);    // tuning: tile height
  static const int bpc_env = agg_env("GCL_AGG_HALO_BPC", 0);  // tuning: blocks per CU
  static const int nt = agg_env("GCL_AGG_NT", 1);
  if (!enabled || ga.ell_width != 8) return -1;  // the arithmetic below is that of agg_kernel<.., EW = 8>
  if ((int64_t)n * ldh * 4 >= (int64_t)1 << 31 || ldh * 4 >= (int64_t)1 << 24 || n >= 1 << 24) return -1;  // 32-bit row offsets
  constexpr int RPW = 64 / LPR;
  const gcl_halo* hl = nullptr;
  for (int t = 0; t < 2; ++t) {
    const gcl_halo& c = ga.halo[t];
    if (c.T == 0 || (force_t && c.T != force_t)) continue;
    const int64_t bytes = (int64_t)(c.smax + 1) * LPR * 16;
    if (bytes > 80 * 1024 || gcl::cdiv((c.smax - c.T) / RPW, 4) > 32) continue;  // at least two blocks per CU
    hl = &c;
    break;
  }
  if (!hl) return -1;
  const int64_t lds = (int64_t)(hl->smax + 1) * LPR * 16;
  // (wide rows, F = 128: 66 KB images, two blocks per CU.  Launched back to back at B = 8 the per-edge kernel on the same
  // tile-ordered graph is as fast, 79 vs 82 us - the whole batch sits in the Infinity Cache then; inside the training
  // step it is not, and the staged form wins, 88 vs 102 us per launch, so there is no batch-size switch here.)
  // persistent blocks, as many per CU as LDS allows; every block walks its share of the (tile, sample) items of its XCD group
  const int per_cu = (int)std::min<int64_t>(8, (160 * 1024) / lds);
  const int J = 32 * (bpc_env > 0 ? bpc_env : per_cu);  // blocks per XCD
  const int mpw = (int)gcl::cdiv((hl->smax - hl->T) / RPW, 4);  // halo pieces per wave of the fullest tile
  dim3 grid((unsigned)(gcl::kNumXCD * J)), block(256);
  auto go = [&](auto kern) -> int {
    const int rc = gcl::ensure_dyn_lds(reinterpret_cast<const void*>(kern), (size_t)lds);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, grid, block, (size_t)lds, st, hl->list, hl->cnt, reinterpret_cast<const int2*>(hl->rec),
                       ga.rowptr, hl->opos, ga.w, hl->smax, h, ldh, bsh, bias, y, ldy, bsy, n, B, F, hl->ntiles, nt);
    return GCL_OK;
  };
  int rc;
  if (hl->T == 64)
    rc = mpw <= 4 ? go(&agg_halo_loop_kernel<LPR, 64, 4>) : mpw <= 8 ? go(&agg_halo_loop_kernel<LPR, 64, 8>)
       : mpw <= 16 ? go(&agg_halo_loop_kernel<LPR, 64, 16>) : go(&agg_halo_loop_kernel<LPR, 64, 32>);
  else
    rc = mpw <= 4 ? go(&agg_halo_loop_kernel<LPR, 32, 4>) : mpw <= 8 ? go(&agg_halo_loop_kernel<LPR, 32, 8>)
       : mpw <= 16 ? go(&agg_halo_loop_kernel<LPR, 32, 16>) : go(&agg_halo_loop_kernel<LPR, 32, 32>);
  if (rc) return rc;
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

template <int LPR>
int launch_agg_heavy(const AggArgs& ga, const float* h, int64_t ldh, int64_t bsh, const float* bias, float* y,
                     int64_t ldy, int64_t bsy, int32_t B, int32_t F, bool vl, bool vs, hipStream_t st) {
  dim3 hgrid((unsigned)ga.n_heavy, (unsigned)B), block(256);
#define GCL_AGGH(VL_, VS_)                                                                                       \
  hipLaunchKernelGGL((agg_heavy_kernel<LPR, VL_, VS_>), hgrid, block, 0, st, ga.heavy, ga.rowptr, ga.col, ga.w, h, \
                     ldh, bsh, bias, y, ldy, bsy, F)
  if (vl && vs) GCL_AGGH(true, true);
  else if (vl) GCL_AGGH(true, false);
  else if (vs) GCL_AGGH(false, true);
  else GCL_AGGH(false, false);
#undef GCL_AGGH
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

template <int LPR>
int launch_agg(const AggArgs& ga, const float* h, int64_t ldh, int64_t bsh, const float* bias, float* y, int64_t ldy,
               int64_t bsy, int32_t n, int32_t B, int32_t F, hipStream_t st) {
  constexpr int RPW = 64 / LPR;
  constexpr int EL = LPR < gcl::kEll ? LPR : gcl::kEll;
  // vector loads need 16-B aligned rows and a padded tail (ldh >= roundup(F,4))
  const bool vl = (ldh % 4 == 0) && (bsh % 4 == 0) && gcl::aligned16(h) && ldh >= ((F + 3) / 4) * 4;
  const bool vs = (ldy % 4 == 0) && (bsy % 4 == 0) && gcl::aligned16(y) && (F % 4 == 0);
  if constexpr (LPR >= 16) {
    if (vl && vs && (!bias || gcl::aligned16(bias))) {
      const int hr = launch_agg_halo<LPR>(ga, h, ldh, bsh, bias, y, ldy, bsy, n, B, F, st);
      if (hr >= 0) {
        if (hr != GCL_OK || ga.n_heavy == 0) return hr;
        return launch_agg_heavy<LPR>(ga, h, ldh, bsh, bias, y, ldy, bsy, B, F, true, true, st);
      }
    }
  }
  static const int iter_env = agg_env("GCL_AGG_ITER", 0);  // tuning overrides (0 = per-graph default)
  static const int ew_env = agg_env("GCL_AGG_EW", 0);
  static const int nt = agg_env("GCL_AGG_NT", 1);  // non-temporal output stores (measured: -7..-12 %)
  int ewidth = ew_env > 0 ? ew_env : ga.ell_width;
  if (ewidth > EL) ewidth = EL;
  ewidth = ewidth >= 8 ? 8 : ewidth >= 4 ? 4 : ewidth >= 2 ? 2 : 1;
  // measured on MI355X (profiles/r01_c_*): dense prefixes (mesh) run best with one batch per wave,
  // near-diagonal bipartite graphs with two (next batch's metadata prefetched)
  const int iter = (iter_env == 1 || iter_env == 2 || iter_env == 4 || iter_env == 8) ? iter_env
                                                                                       : (ewidth >= 4 ? 1 : 2);
  int32_t nRB = (int32_t)gcl::cdiv(n, RPW * 4 * iter);
  if (ga.order16 && RPW * 4 * iter <= 16) nRB = ga.n_order16 * (16 / (RPW * 4 * iter));  // whole 16-row groups (rows >= n are masked)
  const int xcd_map = B >= gcl::kNumXCD ? 1 : 0;
  const int64_t nb = xcd_map ? (int64_t)gcl::kNumXCD * gcl::cdiv(B, gcl::kNumXCD) * nRB : (int64_t)B * nRB;
  GCL_CHECK_ARG(nb < (int64_t)INT32_MAX, "aggregate: grid too large");
  dim3 grid((unsigned)nb), block(256);
  const int32_t* order16 = (RPW * 4 * iter <= 16) ? ga.order16 : nullptr;
#define GCL_AGG4(VL_, VS_, EW_, IT_)                                                                             \
  hipLaunchKernelGGL((agg_kernel<LPR, VL_, VS_, EW_, IT_>), grid, block, 0, st, ga.rowptr, ga.col, ga.w, ga.ecol, \
                     ga.ew, h, ldh, bsh, bias, y, ldy, bsy, n, B, F, nRB, xcd_map, nt, order16)
#define GCL_AGG3(VL_, VS_, EW_)              \
  do {                                       \
    if (iter == 8) GCL_AGG4(VL_, VS_, EW_, 8); \
    else if (iter == 4) GCL_AGG4(VL_, VS_, EW_, 4); \
    else if (iter == 2) GCL_AGG4(VL_, VS_, EW_, 2); \
    else GCL_AGG4(VL_, VS_, EW_, 1);         \
  } while (0)
#define GCL_AGG2(VL_, VS_)                                        \
  do {                                                            \
    if constexpr (EL >= 8) { if (ewidth == 8) { GCL_AGG3(VL_, VS_, 8); break; } } \
    if (ewidth >= 4) GCL_AGG3(VL_, VS_, 4);                       \
    else if (ewidth == 2) GCL_AGG3(VL_, VS_, 2);                  \
    else GCL_AGG3(VL_, VS_, 1);                                   \
  } while (0)
  if (vl && vs) GCL_AGG2(true, true);
  else if (vl) GCL_AGG2(true, false);
  else if (vs) GCL_AGG2(false, true);
  else GCL_AGG2(false, false);
#undef GCL_AGG2
#undef GCL_AGG3
#undef GCL_AGG4
  GCL_CHECK_LAUNCH();
  if (ga.n_heavy > 0) return launch_agg_heavy<LPR>(ga, h, ldh, bsh, bias, y, ldy, bsy, B, F, vl, vs, st);
  return GCL_OK;
}

}  // namespace

extern "C" int gcl_aggregate(const gcl_graph_t* g, int32_t transpose, const float* h, int64_t ldh, int64_t bsh,
                             const float* bias, float* y, int64_t ldy, int64_t bsy, int32_t B, int32_t F,
                             gcl_stream_t stream) {
  GCL_CHECK_ARG(g && h && y, "aggregate: null argument");
  GCL_CHECK_ARG(B > 0 && F > 0 && F <= 256, "aggregate: unsupported B=%d F=%d (F must be in 1..256)", B, F);
  GCL_CHECK_ARG(ldh >= F && ldy >= F, "aggregate: leading dimension smaller than F");
  GCL_CHECK_ARG(g->kind != GCL_GRAPH_GAT, "aggregate: GAT graphs carry no edge weights; use gcl_gat_fwd");
  GCL_CHECK_ARG(h != y, "aggregate: in-place aggregation is not supported");
  AggArgs ga;
  ga.rowptr = transpose ? g->trowptr : g->rowptr;
  ga.col = transpose ? g->tcol : g->col;
  ga.w = transpose ? g->tw : g->w;
  ga.ecol = transpose ? g->tecol : g->ecol;
  ga.ew = transpose ? g->tew : g->ew;
  ga.ell_width = transpose ? g->tell_width : g->ell_width;
  ga.order16 = g->order16[transpose ? 1 : 0];
  ga.n_order16 = g->n_order16;
  ga.heavy = transpose ? g->theavy : g->heavy;
  ga.n_heavy = transpose ? g->n_theavy : g->n_heavy;
  ga.halo = g->halo[transpose ? 1 : 0];
  GCL_CHECK_ARG(B <= 65535 || ga.n_heavy == 0, "aggregate: batch too large for the heavy-row launch");
  hipStream_t st = (hipStream_t)stream;
  const int lanes = (F + 3) / 4;
  if (lanes <= 4) return launch_agg<4>(ga, h, ldh, bsh, bias, y, ldy, bsy, g->n, B, F, st);
  if (lanes <= 8) return launch_agg<8>(ga, h, ldh, bsh, bias, y, ldy, bsy, g->n, B, F, st);
  if (lanes <= 16) return launch_agg<16>(ga, h, ldh, bsh, bias, y, ldy, bsy, g->n, B, F, st);
  if (lanes <= 32) return launch_agg<32>(ga, h, ldh, bsh, bias, y, ldy, bsy, g->n, B, F, st);
  return launch_agg<64>(ga, h, ldh, bsh, bias, y, ldy, bsy, g->n, B, F, st);
}
