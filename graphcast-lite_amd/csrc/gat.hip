// GATConv(heads=H, concat=False) attention + aggregation, forward and backward, and the
// SparseGATConv prune.  Reference call sites: src/models.py:425 (GATConv), :130-151 (SparseGATConv);
// arithmetic per SURVEY.md Appendix A.2 (PyG 2.5.3):
//   e_ij = LeakyReLU_0.2(a_s[j] + a_d[i]);  alpha = softmax over in-edges of i (segment max
//   subtracted, +1e-16 in the denominator);  y[i] = mean_h sum_e alpha_e h[j_e,h,:] + bias.
//
// Mapping: a destination row is owned by LPR lanes x 4 channels covering all H*C channels, so a
// lane's 4 channels lie in ONE head.  Each lane evaluates the softmax of its own head over the
// row's <= ~13 in-edges itself (the per-edge scalars a_s[col_e,h] are same-address loads across the
// head's lanes and come from L1/L2): the neighbour softmax needs no cross-lane traffic at all, and
// the only shuffles are the head-mean in the epilogue (forward) and the C-channel dot products
// (backward).  Constraints per launch: C % 4 == 0, C <= 256; other head counts / widths run as head chunks (gat_check).
#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "common.h"
#include "halo.h"

namespace {

constexpr float kNegSlope = 0.2f;
__device__ __forceinline__ float leaky(float x) { return x > 0.f ? x : kNegSlope * x; }

// Lane -> channel map of the per-edge kernels.  A head occupies Cp / 4 lanes, Cp = the head width C rounded up to a
// power-of-two number of 4-channel lanes (C = 48 -> 16 lanes, C = 96 -> 32), so the in-head reductions stay xor
// butterflies for ANY C % 4 == 0; the lanes of a head beyond C are idle (they contribute zeros).  Cp == C for the
// power-of-two widths, where the map is the plain l * 4.
struct LaneMap {
  int h, cc, c0;  // head of the launch's chunk, channel inside the head, column of h / att (= h * C + cc)
  bool cact;      // the lane owns four real channels
  bool hact;      // the lane lies in a real head (it may still be one of the head's idle lanes)
};
__device__ __forceinline__ LaneMap lane_map(int l, int H, int C, int Cp) {
  LaneMap m;
  const int pc = l * 4;
  const int hd = pc / Cp;
  m.cc = pc - hd * Cp;
  m.hact = hd < H;
  m.cact = m.hact && m.cc < C;
  m.h = m.hact ? hd : 0;
  if (!m.cact) m.cc = 0;
  m.c0 = m.h * C + m.cc;
  return m;
}

// a_s[r,h] = <h[r,h,:], att_src[h,:]>, a_d likewise.  rows = B*n flattened via (ld, bs).
template <int LPR>
__global__ __launch_bounds__(256) void gat_scores_kernel(const float* __restrict__ Hf, int64_t ldh, int64_t bsh,
                                                         const float* __restrict__ att_s,
                                                         const float* __restrict__ att_d, float* __restrict__ a_s,
                                                         float* __restrict__ a_d, int32_t n, int32_t B, int32_t H,
                                                         int32_t C, int32_t Cp, int32_t Hs, int32_t h0) {
  // H = heads of THIS launch (a chunk of the layer's heads when H_total * C > 256 or H_total is not a power of
  // two); per-head arrays are indexed [.., Hs] at head offset h0 (Hs = H_total, h0 = first head of the chunk)
  constexpr int RPW = 64 / LPR, RPB = RPW * 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / LPR, l = lane % LPR;
  const int lph = Cp >> 2;  // lanes per head
  const LaneMap lm = lane_map(l, H, C, Cp);
  const int c0 = lm.c0;
  const bool cact = lm.cact;
  float s0 = 0, s1 = 0, s2 = 0, s3 = 0, d0 = 0, d1 = 0, d2 = 0, d3 = 0;
  if (cact) {
    s0 = att_s[c0]; s1 = att_s[c0 + 1]; s2 = att_s[c0 + 2]; s3 = att_s[c0 + 3];
    d0 = att_d[c0]; d1 = att_d[c0 + 1]; d2 = att_d[c0 + 2]; d3 = att_d[c0 + 3];
  }
  const int64_t rows = (int64_t)B * n;
  for (int64_t r = (int64_t)blockIdx.x * RPB + wave * RPW + sub; r < rows; r += (int64_t)gridDim.x * RPB) {
    const int64_t b = r / n;
    const int i = (int)(r - b * n);
    float ps = 0.f, pd = 0.f;
    if (cact) {
      const float4 v = *reinterpret_cast<const float4*>(Hf + b * bsh + (int64_t)i * ldh + c0);
      ps = v.x * s0 + v.y * s1 + v.z * s2 + v.w * s3;
      pd = v.x * d0 + v.y * d1 + v.z * d2 + v.w * d3;
    }
    for (int off = lph >> 1; off > 0; off >>= 1) {
      ps += __shfl_xor(ps, off, 64);
      pd += __shfl_xor(pd, off, 64);
    }
    if (cact && (l % lph) == 0) {
      const int h = lm.h;
      a_s[r * Hs + h0 + h] = ps;
      a_d[r * Hs + h0 + h] = pd;
    }
  }
}

template <int LPR>
__global__ __launch_bounds__(256) void gat_fwd_kernel(const int32_t* __restrict__ rowptr,
                                                      const int32_t* __restrict__ col,
                                                      const int32_t* __restrict__ ecol, const float* __restrict__ Hf,
                                                      int64_t ldh, int64_t bsh, const float* __restrict__ a_s,
                                                      const float* __restrict__ a_d, const float* __restrict__ bias,
                                                      float* __restrict__ alpha, float* __restrict__ Y, int64_t ldy,
                                                      int64_t bsy, int32_t n, int64_t Ep, int32_t B, int32_t H,
                                                      int32_t C, int32_t Cp, int32_t nRB, int32_t xcd_map, int32_t Hs,
                                                      int32_t h0, int32_t Htot, int32_t yacc) {
  constexpr int RPW = 64 / LPR, RPB = RPW * 4;
  constexpr int EL = LPR < gcl::kEll ? LPR : gcl::kEll;
  const int bid = blockIdx.x;
  int b, rb;
  if (xcd_map) {
    const int slot = bid >> 3;
    b = (bid & 7) + 8 * (slot / nRB);
    rb = slot % nRB;
  } else {
    b = bid / nRB;
    rb = bid % nRB;
  }
  if (b >= B) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / LPR, l = lane % LPR;
  const int gbase = sub * LPR;
  const int row = rb * RPB + wave * RPW + sub;
  const int lph = Cp >> 2;
  const LaneMap lm = lane_map(l, H, C, Cp);
  const bool cact = lm.cact;
  const bool active = (row < n) && cact;
  const int cc = lm.c0;  // column of h (0 for an idle lane)
  const int h = lm.h;
  const int rc = row < n ? row : n - 1;
  const float* __restrict__ Hb = Hf + (int64_t)b * bsh;
  const float* __restrict__ as_b = a_s + (int64_t)b * n * Hs + h0;
  // metadata: one round trip (the ELL prefix does not depend on rowptr)
  const int start = rowptr[rc], end = rowptr[rc + 1];
  const int cj = ecol[(int64_t)rc * gcl::kEll + (l & (EL - 1))];
  const float ad = a_d[((int64_t)b * n + rc) * Hs + h0 + h];
  const int deg = end - start;
  const bool leader = active && (l % lph) == 0;
  float a0 = 0, a1 = 0, a2 = 0, a3 = 0;

  if (!__any(deg > EL)) {
    // fast path: <= EL in-edges -> indices, scores and neighbour rows are all in flight together;
    // the softmax of this lane's head lives in registers (no cross-lane traffic)
    int jj[EL];
    float sc[EL];
#pragma unroll
    for (int k = 0; k < EL; ++k) jj[k] = __shfl(cj, gbase + k, 64);
    float4 v[EL];
#pragma unroll
    for (int k = 0; k < EL; ++k) v[k] = *reinterpret_cast<const float4*>(Hb + (int64_t)jj[k] * ldh + cc);
    if (lph >= EL) {
      // A head spans >= EL lanes: lane hl < EL of the head owns in-edge hl - ONE gathered score load,
      // ONE exp and ONE coalesced alpha store per lane instead of EL of each; the EL softmax weights
      // are then broadcast inside the head for the weighted sum.
      const int hl = l % lph, hbase = gbase + (l / lph) * lph;
      const bool mine = hl < EL && hl < deg;
      const int jm = __shfl(cj, gbase + (hl & (EL - 1)), 64);
      float e = mine ? leaky(as_b[(int64_t)jm * Hs + h] + ad) : -INFINITY;
      float m = e;
#pragma unroll
      for (int off = EL >> 1; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));  // lanes 0..EL-1 of the head
      m = __shfl(m, hbase, 64);
      const float ex = mine ? expf(e - m) : 0.f;
      float den = ex;
#pragma unroll
      for (int off = EL >> 1; off > 0; off >>= 1) den += __shfl_xor(den, off, 64);
      den = __shfl(den, hbase, 64);
      const float almine = ex * (1.f / (den + 1e-16f));
      if (alpha && mine && row < n && lm.hact) alpha[((int64_t)b * Ep + start + hl) * Hs + h0 + h] = almine;
#pragma unroll
      for (int k = 0; k < EL; ++k) {
        const float al = __shfl(almine, hbase + k, 64);
        const bool in = k < deg;  // select, not multiply: a padded slot's row must not leak non-finite values
        a0 += in ? al * v[k].x : 0.f;
        a1 += in ? al * v[k].y : 0.f;
        a2 += in ? al * v[k].z : 0.f;
        a3 += in ? al * v[k].w : 0.f;
      }
    } else {
#pragma unroll
    for (int k = 0; k < EL; ++k) sc[k] = as_b[(int64_t)jj[k] * Hs + h];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < EL; ++k) {
      sc[k] = leaky(sc[k] + ad);
      m = (k < deg) ? fmaxf(m, sc[k]) : m;
    }
    float den = 0.f;
#pragma unroll
    for (int k = 0; k < EL; ++k) {
      sc[k] = (k < deg) ? expf(sc[k] - m) : 0.f;
      den += sc[k];
    }
    const float inv = 1.f / (den + 1e-16f);
#pragma unroll
    for (int k = 0; k < EL; ++k) {
      const float al = sc[k] * inv;
      const bool in = k < deg;
      a0 += in ? al * v[k].x : 0.f;
      a1 += in ? al * v[k].y : 0.f;
      a2 += in ? al * v[k].z : 0.f;
      a3 += in ? al * v[k].w : 0.f;
      if (leader && alpha && in) alpha[((int64_t)b * Ep + start + k) * Hs + h0 + h] = al;
    }
    }
  } else {
    float m = -INFINITY;
    for (int e = start; e < end; ++e) m = fmaxf(m, leaky(as_b[(int64_t)col[e] * Hs + h] + ad));
    float den = 0.f;
    for (int e = start; e < end; ++e) den += expf(leaky(as_b[(int64_t)col[e] * Hs + h] + ad) - m);
    const float inv = 1.f / (den + 1e-16f);
    for (int e = start; e < end; ++e) {
      const int j = col[e];
      const float al = expf(leaky(as_b[(int64_t)j * Hs + h] + ad) - m) * inv;
      const float4 v = *reinterpret_cast<const float4*>(Hb + (int64_t)j * ldh + cc);
      a0 += al * v.x; a1 += al * v.y; a2 += al * v.z; a3 += al * v.w;
      if (leader && alpha) alpha[((int64_t)b * Ep + e) * Hs + h0 + h] = al;
    }
  }
  // mean over heads: lanes holding the same channel of different heads are lph apart
  for (int off = lph; off < lph * H; off <<= 1) {
    a0 += __shfl_xor(a0, off, 64); a1 += __shfl_xor(a1, off, 64);
    a2 += __shfl_xor(a2, off, 64); a3 += __shfl_xor(a3, off, 64);
  }
  if (active && h == 0) {
    const float s = 1.f / (float)Htot;  // mean over ALL heads of the layer; later chunks add onto the first one's result
    const int c0 = lm.cc;  // head 0: the output channel
    float* yp = Y + (int64_t)b * bsy + (int64_t)row * ldy + c0;
    float o0 = a0 * s, o1 = a1 * s, o2 = a2 * s, o3 = a3 * s;
    if (bias) { o0 += bias[c0]; o1 += bias[c0 + 1]; o2 += bias[c0 + 2]; o3 += bias[c0 + 3]; }
    if (yacc) {
      const float4 old = *reinterpret_cast<const float4*>(yp);
      o0 += old.x; o1 += old.y; o2 += old.z; o3 += old.w;
    }
    *reinterpret_cast<float4*>(yp) = make_float4(o0, o1, o2, o3);  // 16-B rows: checked on the host
  }
}

// Backward, destination side: per in-edge e of row i (head h)
//   dalpha_e = <dy[i]/H, h[j_e,h,:]>, t = sum alpha_e dalpha_e, de_e = alpha_e (dalpha_e - t) * LeakyReLU'
// writes de[b, slot, h] and da_d[b, i, h] = sum_e de_e.
template <int LPR>
__global__ __launch_bounds__(256) void gat_bwd_dst_kernel(const int32_t* __restrict__ rowptr,
                                                          const int32_t* __restrict__ col,
                                                          const int32_t* __restrict__ ecol,
                                                          const float* __restrict__ dY, int64_t lddy, int64_t bsdy,
                                                          const float* __restrict__ Hf, int64_t ldh, int64_t bsh,
                                                          const float* __restrict__ a_s, const float* __restrict__ a_d,
                                                          const float* __restrict__ alpha, float* __restrict__ de,
                                                          float* __restrict__ dad, int32_t n, int64_t Ep, int32_t B,
                                                          int32_t H, int32_t C, int32_t Cp, int32_t nRB, int32_t xcd_map,
                                                          int32_t Hs, int32_t h0, int32_t Htot) {
  constexpr int RPW = 64 / LPR, RPB = RPW * 4;
  constexpr int EL = LPR < gcl::kEll ? LPR : gcl::kEll;
  // all row blocks of a sample on ONE XCD (blocks are dealt round-robin): its h rows stay in that L2
  int b, rb;
  if (xcd_map) {
    const int slot = blockIdx.x >> 3;
    b = (blockIdx.x & 7) + 8 * (slot / nRB);
    rb = slot % nRB;
  } else {
    b = blockIdx.x / nRB;
    rb = blockIdx.x % nRB;
  }
  if (b >= B) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / LPR, l = lane % LPR;
  const int gbase = sub * LPR;
  const int row = rb * RPB + wave * RPW + sub;
  const int lph = Cp >> 2;
  const LaneMap lm = lane_map(l, H, C, Cp);
  const bool cact = lm.cact;
  const bool active = (row < n) && cact;
  const int cc0 = lm.c0;
  const int h = lm.h;
  const int cc = lm.cc;  // channel inside the head = channel of dy
  const int rc = row < n ? row : n - 1;
  const float* __restrict__ Hb = Hf + (int64_t)b * bsh;
  const float* __restrict__ as_b = a_s + (int64_t)b * n * Hs + h0;
  const int start = rowptr[rc], end = rowptr[rc + 1];
  const int cj = ecol[(int64_t)rc * gcl::kEll + (l & (EL - 1))];
  const float ad = a_d[((int64_t)b * n + rc) * Hs + h0 + h];
  const float* dp = dY + (int64_t)b * bsdy + (int64_t)rc * lddy + cc;
  const float sH = 1.f / (float)Htot;
  const float g0 = cact ? dp[0] * sH : 0.f, g1 = cact ? dp[1] * sH : 0.f;
  const float g2 = cact ? dp[2] * sH : 0.f, g3 = cact ? dp[3] * sH : 0.f;
  const int deg = end - start;
  const bool leader = active && (l % lph) == 0;
  float sum_de = 0.f;

  if (!__any(deg > EL)) {
    int jj[EL];
    float4 v[EL];
    float al[EL], pre[EL], d[EL];
#pragma unroll
    for (int k = 0; k < EL; ++k) jj[k] = __shfl(cj, gbase + k, 64);
#pragma unroll
    for (int k = 0; k < EL; ++k) v[k] = *reinterpret_cast<const float4*>(Hb + (int64_t)jj[k] * ldh + cc0);
#pragma unroll
    for (int k = 0; k < EL; ++k) {
      const int e = (k < deg) ? start + k : start;
      al[k] = alpha[((int64_t)b * Ep + e) * Hs + h0 + h];
      pre[k] = as_b[(int64_t)jj[k] * Hs + h] + ad;
    }
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < EL; ++k) {
      float dk = g0 * v[k].x + g1 * v[k].y + g2 * v[k].z + g3 * v[k].w;
      for (int off = lph >> 1; off > 0; off >>= 1) dk += __shfl_xor(dk, off, 64);
      d[k] = dk;
      al[k] = (k < deg) ? al[k] : 0.f;
      t += al[k] * dk;
    }
#pragma unroll
    for (int k = 0; k < EL; ++k) {
      const float dev = al[k] * (d[k] - t) * (pre[k] > 0.f ? 1.f : kNegSlope);
      sum_de += (k < deg) ? dev : 0.f;
      if (leader && k < deg) de[((int64_t)b * Ep + start + k) * Hs + h0 + h] = dev;
    }
  } else {
    // generic path: the trip count is uniform across the lanes that shuffle together (one row group)
    float t = 0.f;
    for (int e = start; e < end; ++e) {
      const float4 v = *reinterpret_cast<const float4*>(Hb + (int64_t)col[e] * ldh + cc0);
      float d = g0 * v.x + g1 * v.y + g2 * v.z + g3 * v.w;
      for (int off = lph >> 1; off > 0; off >>= 1) d += __shfl_xor(d, off, 64);
      t += alpha[((int64_t)b * Ep + e) * Hs + h0 + h] * d;
    }
    for (int e = start; e < end; ++e) {
      const int j = col[e];
      const float4 v = *reinterpret_cast<const float4*>(Hb + (int64_t)j * ldh + cc0);
      float d = g0 * v.x + g1 * v.y + g2 * v.z + g3 * v.w;
      for (int off = lph >> 1; off > 0; off >>= 1) d += __shfl_xor(d, off, 64);
      const float a_ = alpha[((int64_t)b * Ep + e) * Hs + h0 + h];
      const float pr = as_b[(int64_t)j * Hs + h] + ad;
      const float dev = a_ * (d - t) * (pr > 0.f ? 1.f : kNegSlope);
      sum_de += dev;
      if (leader) de[((int64_t)b * Ep + e) * Hs + h0 + h] = dev;
    }
  }
  if (leader) dad[((int64_t)b * n + row) * Hs + h0 + h] = sum_de;
}

// Backward, source side (transposed CSR): for source row j, head h
//   da_s[j] = sum_{e: src=j} de_e
//   dh[j,h,:] = sum_{e: src=j} alpha_e dy[dst_e]/H + da_s[j] att_src[h,:] + da_d[j] att_dst[h,:]
template <int LPR>
__global__ __launch_bounds__(256) void gat_bwd_src_kernel(const int32_t* __restrict__ trowptr,
                                                          const int32_t* __restrict__ tcol,
                                                          const int32_t* __restrict__ tslot,
                                                          const int32_t* __restrict__ tecol,
                                                          const int32_t* __restrict__ teslot,
                                                          const float* __restrict__ dY, int64_t lddy, int64_t bsdy,
                                                          const float* __restrict__ alpha, const float* __restrict__ de,
                                                          const float* __restrict__ dad, const float* __restrict__ att_s,
                                                          const float* __restrict__ att_d, float* __restrict__ das,
                                                          float* __restrict__ dH, int64_t lddh, int64_t bsdh,
                                                          int32_t n, int64_t Ep, int32_t B, int32_t H, int32_t C,
                                                          int32_t Cp, int32_t nRB, int32_t vdy, int32_t xcd_map, int32_t Hs,
                                                          int32_t h0, int32_t Htot) {
  constexpr int RPW = 64 / LPR, RPB = RPW * 4;
  constexpr int EL = LPR < gcl::kEll ? LPR : gcl::kEll;
  int b, rb;
  if (xcd_map) {
    const int slot = blockIdx.x >> 3;
    b = (blockIdx.x & 7) + 8 * (slot / nRB);
    rb = slot % nRB;
  } else {
    b = blockIdx.x / nRB;
    rb = blockIdx.x % nRB;
  }
  if (b >= B) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / LPR, l = lane % LPR;
  const int gbase = sub * LPR;
  const int row = rb * RPB + wave * RPW + sub;
  const int lph = Cp >> 2;
  const LaneMap lm = lane_map(l, H, C, Cp);
  const bool active = (row < n) && lm.cact;
  const int c0 = lm.c0, h = lm.h, cc = lm.cc;
  const int rc = row < n ? row : n - 1;
  const float s = 1.f / (float)Htot;
  const int start = trowptr[rc], end = trowptr[rc + 1];
  const int ci = tecol[(int64_t)rc * gcl::kEll + (l & (EL - 1))];
  const int si = teslot[(int64_t)rc * gcl::kEll + (l & (EL - 1))];
  const int deg = end - start;
  const float* dYb = dY + (int64_t)b * bsdy;
  float a0 = 0, a1 = 0, a2 = 0, a3 = 0, sde = 0.f;
  auto ld_dy = [&](int i) -> float4 {
    const float* p = dYb + (int64_t)i * lddy + cc;
    if (vdy) return *reinterpret_cast<const float4*>(p);
    return make_float4(p[0], p[1], p[2], p[3]);
  };
  if (!__any(deg > EL)) {
    int ii[EL], ss[EL];
#pragma unroll
    for (int k = 0; k < EL; ++k) {
      ii[k] = __shfl(ci, gbase + k, 64);
      ss[k] = __shfl(si, gbase + k, 64);
    }
    float4 v[EL];
    float al[EL], dk[EL];
#pragma unroll
    for (int k = 0; k < EL; ++k) v[k] = ld_dy(ii[k]);
#pragma unroll
    for (int k = 0; k < EL; ++k) {
      const int64_t sl = ((int64_t)b * Ep + ss[k]) * Hs + h0 + h;
      al[k] = alpha[sl];
      dk[k] = de[sl];
    }
#pragma unroll
    for (int k = 0; k < EL; ++k) {
      const bool in = k < deg;
      const float a_ = al[k] * s;
      sde += in ? dk[k] : 0.f;
      a0 += in ? a_ * v[k].x : 0.f;
      a1 += in ? a_ * v[k].y : 0.f;
      a2 += in ? a_ * v[k].z : 0.f;
      a3 += in ? a_ * v[k].w : 0.f;
    }
  } else {
    for (int e = start; e < end; ++e) {
      const int i = tcol[e];
      const int64_t sl = ((int64_t)b * Ep + tslot[e]) * Hs + h0 + h;
      const float a_ = alpha[sl] * s;
      sde += de[sl];
      const float4 v = ld_dy(i);
      a0 += a_ * v.x; a1 += a_ * v.y; a2 += a_ * v.z; a3 += a_ * v.w;
    }
  }
  if (!active) return;
  const float dd = dad[((int64_t)b * n + row) * Hs + h0 + h];
  a0 += sde * att_s[c0] + dd * att_d[c0];
  a1 += sde * att_s[c0 + 1] + dd * att_d[c0 + 1];
  a2 += sde * att_s[c0 + 2] + dd * att_d[c0 + 2];
  a3 += sde * att_s[c0 + 3] + dd * att_d[c0 + 3];
  float* o = dH + (int64_t)b * bsdh + (int64_t)row * lddh + c0;
  *reinterpret_cast<float4*>(o) = make_float4(a0, a1, a2, a3);  // 16-B rows: checked on the host
  if ((l % lph) == 0) das[((int64_t)b * n + row) * Hs + h0 + h] = sde;
}

// part[block][2][HC]: sum_rows da_s[r,h] * h[r,hc]  |  sum_rows da_d[r,h] * h[r,hc]
template <int LPR>
__global__ __launch_bounds__(256) void gat_datt_kernel(const float* __restrict__ Hf, int64_t ldh, int64_t bsh,
                                                       const float* __restrict__ das, const float* __restrict__ dad,
                                                       float* __restrict__ part, int32_t n, int32_t B, int32_t H,
                                                       int32_t C, int32_t Cp, int32_t Hs, int32_t h0) {
  constexpr int RPW = 64 / LPR, RPB = RPW * 4;
  __shared__ float red[RPB][LPR * 4 * 2 + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / LPR, l = lane % LPR;
  const int HC = H * C;
  const LaneMap lm = lane_map(l, H, C, Cp);
  const bool cact = lm.cact;
  const int h = lm.h, c0 = lm.c0;
  float s0 = 0, s1 = 0, s2 = 0, s3 = 0, d0 = 0, d1 = 0, d2 = 0, d3 = 0;
  const int64_t rows = (int64_t)B * n;
  if (cact)
    for (int64_t r = (int64_t)blockIdx.x * RPB + wave * RPW + sub; r < rows; r += (int64_t)gridDim.x * RPB) {
      const int64_t b = r / n;
      const int i = (int)(r - b * n);
      const float4 v = *reinterpret_cast<const float4*>(Hf + b * bsh + (int64_t)i * ldh + c0);
      const float ws = das[r * Hs + h0 + h], wd = dad[r * Hs + h0 + h];
      s0 += ws * v.x; s1 += ws * v.y; s2 += ws * v.z; s3 += ws * v.w;
      d0 += wd * v.x; d1 += wd * v.y; d2 += wd * v.z; d3 += wd * v.w;
    }
  float* r_ = red[wave * RPW + sub];
  if (cact) {  // every column < HC is owned by exactly one lane of the row group; the idle lanes of a padded head write nothing
    r_[c0] = s0; r_[c0 + 1] = s1; r_[c0 + 2] = s2; r_[c0 + 3] = s3;
    r_[LPR * 4 + c0] = d0; r_[LPR * 4 + c0 + 1] = d1; r_[LPR * 4 + c0 + 2] = d2; r_[LPR * 4 + c0 + 3] = d3;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 2 * LPR * 4; idx += 256) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < RPB; ++q) s += red[q][idx];
    const int which = idx / (LPR * 4), c = idx % (LPR * 4);
    if (c < HC) part[((size_t)blockIdx.x * 2 + which) * HC + c] = s;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Source-tile forward for one head (the reference's GAT / SparseGAT configs run heads = 1: configs[2], [4]) on graphs
// that carry a tile layout (common.h gcl_halo; mesh rows in tile order): a block stages a 64-row tile's own rows and
// its halo rows of h once by LDS-DMA (the staging of agg_halo_loop_kernel, aggregate.hip) and does EVERYTHING of the
// layer from LDS - the scores a_s = <h, att_src>, a_d = <h, att_dst> of the staged rows (no separate scores pass
// over h), the softmax over a row's in-edges in registers, and the weighted sum - instead of one L2 gather per edge
// plus one same-address score load per edge.  Persistent blocks, contiguous tile-major (tile, sample) items: a
// tile's list entries, edge records and CSR offsets stay in registers for all samples of the XCD group.
//   item: DMA | barrier | scores of the staged rows -> sS[], sD[] (own rows also to a_src / a_dst for the backward)
//         | barrier | per row: e_k = LeakyReLU(sS[pos_k] + sD[row]), m, sum exp, alpha_k; y = sum alpha_k h[pos_k] + b;
//         lane k stores alpha_k | barrier (image free)
// Padded record slots point at the zero row, whose score is -inf: exp() makes them 0 without a select.
// ---------------------------------------------------------------------------------------------------------
// value of the lane N places further on in the same 16-lane DPP row (rotation): folds into v_max_f32_dpp / v_add_f32_dpp
template <int N>
__device__ __forceinline__ float row_ror(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + N, 0xf, 0xf, true));
}

// TAB: the rows of h are read through a row table (row i of sample b = row tab[i] of sample b of Hf when tab[i] >= 0,
// the batch-invariant flat row ~tab[i] of Hf otherwise): the compact pipeline's first GAT layer transforms only the
// compact rows (functional.LatSource) and the [B, M, C] tensor of transformed mesh rows is never written.
template <int LPR, int MAXPW, bool TAB = false>
__global__ __launch_bounds__(256) void gat_halo_fwd_kernel(const int32_t* __restrict__ list, const int32_t* __restrict__ cnt,
                                                           const int2* __restrict__ rec, const int32_t* __restrict__ rowptr,
                                                           const int32_t* __restrict__ opos, int32_t smax,
                                                           const float* __restrict__ Hf, int64_t ldh, int64_t bsh,
                                                           const float* __restrict__ att_s, const float* __restrict__ att_d,
                                                           const float* __restrict__ bias, float* __restrict__ a_src,
                                                           float* __restrict__ a_dst, float* __restrict__ alpha,
                                                           float* __restrict__ Y, int64_t ldy, int64_t bsy, int32_t n,
                                                           int64_t Ep, int32_t B, int32_t C, int32_t ntiles,
                                                           const int32_t* __restrict__ tab = nullptr) {
  using gcl::halo::row_bcast;
  extern __shared__ float4 img[];  // [(smax + 1) * LPR] staged rows + zero row | sS[smax + 1] | sD[64]
  constexpr int T = 64, NW = 4;
  constexpr int RPW = 64 / LPR, NIT = T / NW / RPW;
  constexpr int SH = LPR == 16 ? 8 : 9;
  typedef float v4f __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) const v4f* lds4_t;
  typedef __attribute__((address_space(3))) const float* ldsf_t;
  float* sS = reinterpret_cast<float*>(img + (smax + 1) * LPR);
  float* sD = sS + (smax + 1);
  const int xcd = blockIdx.x & (gcl::kNumXCD - 1);
  const int J = gridDim.x >> 3, j = blockIdx.x >> 3;
  const int nsamp = (B - xcd + gcl::kNumXCD - 1) / gcl::kNumXCD;
  const int items = nsamp * ntiles;
  const int base = items / J, extra = items - base * J;
  int m = j * base + min(j, extra);
  const int mend = m + base + (j < extra ? 1 : 0);
  if (m >= mend) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int sub = lane / LPR, l = lane % LPR, c0 = l * 4;
  const unsigned cb = (unsigned)c0 * 4u, ldb = (unsigned)ldh * 4u;
  const int hstride = smax - T;
  const unsigned lds0 = (unsigned)(size_t)((gcl::halo::lptr_t)img);
  const unsigned lb = lds0 + (unsigned)l * 16u;
  const unsigned ldsS = (unsigned)(size_t)((gcl::halo::lptr_t)sS);

  if (threadIdx.x < LPR) img[smax * LPR + threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (threadIdx.x == 0) sS[smax] = -INFINITY;  // the score of a padded slot
  const float4 as4 = *reinterpret_cast<const float4*>(att_s + c0), ad4 = *reinterpret_cast<const float4*>(att_d + c0);
  float4 bz = make_float4(0.f, 0.f, 0.f, 0.f);
  if (bias) bz = *reinterpret_cast<const float4*>(bias + c0);
  asm volatile("" : "+v"(bz.x), "+v"(bz.y), "+v"(bz.z), "+v"(bz.w));

  int jj[MAXPW], nhalo = 0, tile = -1;
  int own[TAB ? NIT : 1];
  int2 rc[NIT];
  int rstart[NIT];
  while (true) {
    const int tnew = m / nsamp;
    const int s = m - tnew * nsamp;
    const int b = xcd + gcl::kNumXCD * s;
    const bool newtile = tnew != tile;
    if (newtile) {
      tile = tnew;
      nhalo = cnt[tile] / RPW;
      const int32_t* __restrict__ tl = list + (int64_t)tile * hstride;
#pragma unroll
      for (int q = 0; q < MAXPW; ++q) {
        const int e0 = min((wave + NW * q) * RPW, hstride - RPW);
        int jv = tl[e0];
#pragma unroll
        for (int r = 1; r < RPW; ++r) {
          const int jr = tl[e0 + r];
          jv = sub == r ? jr : jv;
        }
        jj[q] = TAB ? tab[jv] : jv;
      }
      if (TAB) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const int row = tile * T + wave * (T / NW) + sub + it * RPW;
          own[TAB ? it : 0] = tab[row < n ? row : n - 1];
        }
        // the table entries arrive HERE, inside the new-tile branch: left to hipcc, the wait for these vector loads sits
        // at the join and every item then waits for the previous item's stores before it may issue its DMA
#pragma unroll
        for (int it = 0; it < NIT; ++it) asm volatile("" : "+v"(own[TAB ? it : 0]));
#pragma unroll
        for (int q = 0; q < MAXPW; ++q) asm volatile("" : "+v"(jj[q]));
      }
    }
    const char* Hc = reinterpret_cast<const char*>(Hf + (int64_t)b * bsh);
    const char* H0 = reinterpret_cast<const char*>(Hf);
    const unsigned sboff = (unsigned)b * (unsigned)bsh * 4u;
    const int row0 = tile * T + wave * (T / NW) + sub;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int row = row0 + it * RPW;
      const int sr = TAB ? own[TAB ? it : 0] : (row < n ? row : n - 1);
      // TAB: one base and a 32-bit byte offset (the whole of Hf stays below 4 GiB: checked on the host)
      const char* src = TAB ? H0 + ((sr >= 0 ? sboff + __umul24(sr, ldb) : __umul24(~sr, ldb)) + cb) : Hc + (__umul24(sr, ldb) + cb);
      __builtin_amdgcn_global_load_lds((gcl::halo::gptr_t)src, (gcl::halo::lptr_t)(img + (wave * (T / NW) + it * RPW) * LPR), 16, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < MAXPW; ++q) {
      const int p = wave + NW * q;
      if (p < nhalo) {
        const int sr = jj[q];
        const char* src = TAB ? H0 + ((sr >= 0 ? sboff + __umul24(sr, ldb) : __umul24(~sr, ldb)) + cb) : Hc + (__umul24(sr, ldb) + cb);
        __builtin_amdgcn_global_load_lds((gcl::halo::gptr_t)src, (gcl::halo::lptr_t)(img + (T + p * RPW) * LPR), 16, 0, 0);
      }
    }
    if (newtile) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int row = row0 + it * RPW;
        const int rcl = row < n ? row : n - 1;
        rc[it] = rec[(int64_t)rcl * gcl::kHaloRec + (l & (gcl::kHaloRec - 1))];
        rstart[it] = rowptr[rcl];
      }
    }
    __syncthreads();  // image complete (vmcnt(0) + barrier)

    // ---- scores of the staged rows: groups of RPW image rows, dealt round-robin to the waves
    {
      const int ngroups = T / RPW + nhalo;
      for (int gq = wave; gq < ngroups; gq += NW) {
        const int p = gq * RPW + sub;  // image row
        const v4f v = *(lds4_t)(lds0 + ((unsigned)p << SH) + (unsigned)l * 16u);
        float ps = v.x * as4.x + v.y * as4.y + v.z * as4.z + v.w * as4.w;
        float pd = v.x * ad4.x + v.y * ad4.y + v.z * ad4.z + v.w * ad4.w;
#pragma unroll
        for (int off = LPR >> 1; off > 0; off >>= 1) {
          ps += __shfl_xor(ps, off, 64);
          pd += __shfl_xor(pd, off, 64);
        }
        if (l == 0) {
          sS[p] = ps;
          if (p < T) {
            sD[p] = pd;
            const int row = tile * T + p;
            if (row < n) {
              a_src[(int64_t)b * n + row] = ps;
              a_dst[(int64_t)b * n + row] = pd;
            }
          }
        }
      }
    }
    __syncthreads();

    // ---- softmax over the in-edges and weighted sum, per row group
    float* __restrict__ Yb = Y + (int64_t)b * bsy;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int row = row0 + it * RPW;
      const int rloc = wave * (T / NW) + it * RPW + sub;
      const int rx = rc[it].x;
      const int rxp = rx & gcl::kHaloPosMask;
      const int rxb = rxp << SH;
      const int last = row_bcast<15>(rx);
      const float adst = sD[rloc];
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      if (!__any((last & gcl::kHaloMore) != 0)) {
        // lane k of a 16-lane DPP row owns in-edge slot k: ONE score, ONE exp per lane; the row maximum and the
        // denominator are four rotate-and-combine steps each (v_max_f32_dpp / v_add_f32_dpp row_ror), every lane ends
        // up with both; the 16 weights are then broadcast slot by slot for the weighted sum
        const float e = leaky(*(ldsf_t)(ldsS + (unsigned)rxp * 4u) + adst);  // padded slot: -inf
        float mx = e;
        mx = fmaxf(mx, row_ror<8>(mx)); mx = fmaxf(mx, row_ror<4>(mx));
        mx = fmaxf(mx, row_ror<2>(mx)); mx = fmaxf(mx, row_ror<1>(mx));
        const float ex = expf(e - mx);
        float den = ex;
        den += row_ror<8>(den); den += row_ror<4>(den); den += row_ror<2>(den); den += row_ror<1>(den);
        const float almine = ex * (1.f / (den + 1e-16f));
        const int ali = __float_as_int(almine);
        const bool wide = __any(row_bcast<8>(rxb) != (smax << SH));
#define GCL_GAT_ACC(Kk)                                                          \
  {                                                                              \
    const unsigned adr = (unsigned)row_bcast<Kk>(rxb) + lb;                      \
    const v4f v = *(lds4_t)adr;                                                  \
    const float al = __int_as_float(row_bcast<Kk>(ali));                         \
    a0 += al * v.x; a1 += al * v.y; a2 += al * v.z; a3 += al * v.w;              \
  }
        GCL_GAT_ACC(0) GCL_GAT_ACC(1) GCL_GAT_ACC(2) GCL_GAT_ACC(3)
        GCL_GAT_ACC(4) GCL_GAT_ACC(5) GCL_GAT_ACC(6) GCL_GAT_ACC(7)
        if (wide) {
          GCL_GAT_ACC(8) GCL_GAT_ACC(9) GCL_GAT_ACC(10) GCL_GAT_ACC(11)
          GCL_GAT_ACC(12) GCL_GAT_ACC(13) GCL_GAT_ACC(14) GCL_GAT_ACC(15)
        }
#undef GCL_GAT_ACC
        if (alpha && row < n && l < gcl::kHaloRec && rxp != smax) alpha[(int64_t)b * Ep + rstart[it] + l] = almine;
      } else {
        // a row of this group has more than 16 in-edges: the CSR arrays give every edge's image position
        const int rcl = row < n ? row : n - 1;
        const int st = rowptr[rcl], en = rowptr[rcl + 1];
        float mx = -INFINITY;
        for (int e = st; e < en; ++e) mx = fmaxf(mx, leaky(sS[opos[e]] + adst));
        float den = 0.f;
        for (int e = st; e < en; ++e) den += expf(leaky(sS[opos[e]] + adst) - mx);
        const float inv = 1.f / (den + 1e-16f);
        for (int e = st; e < en; ++e) {
          const int pk = opos[e];
          const float al = expf(leaky(sS[pk] + adst) - mx) * inv;
          const v4f v = *(lds4_t)(((unsigned)pk << SH) + lb);
          a0 += al * v.x; a1 += al * v.y; a2 += al * v.z; a3 += al * v.w;
          if (alpha && l == 0 && row < n) alpha[(int64_t)b * Ep + e] = al;
        }
      }
      if (row < n) {
        float* __restrict__ yp = Yb + (int64_t)row * ldy + c0;
        *reinterpret_cast<float4*>(yp) = make_float4(a0 + bz.x, a1 + bz.y, a2 + bz.z, a3 + bz.w);
      }
    }
    if (++m >= mend) break;
    __syncthreads();  // every wave is done with the image and the score arrays
  }
}

// ---------------------------------------------------------------------------------------------------------
// Source-tile backward for one head (the counterpart of gat_halo_fwd_kernel), two kernels:
//   dst side (forward tile layout; the image holds h rows): per in-edge slot k of row i
//       dalpha_k = <dy_i, h[pos_k]>,  t = sum_k alpha_k dalpha_k,  de_k = alpha_k (dalpha_k - t) LeakyReLU'(a_s[src_k] + a_d[i])
//     lane k of the row's 16-lane DPP row owns slot k (its alpha, its de); the dot products are reduced with four
//     rotate-and-add steps, so every lane has dalpha_k and lane k keeps it.  a_s of the staged rows arrives by two
//     4-byte LDS-DMAs (own rows: contiguous; halo rows: by the list).  Writes de[b, slot] and da_d[b, i].
//   src side (transposed tile layout; the image holds dy rows): per out-edge slot k of source row j
//       dh_j = sum_k alpha_k dy[dst_k] + (sum_k de_k) att_src + da_d[j] att_dst,  da_s[j] = sum_k de_k
//     alpha_k / de_k are fetched by the edge's forward CSR slot, which the transposed edge records of a GAT graph
//     carry in place of a weight.
// Rows with more than 16 edges take the CSR loop; heavy rows (> 64) keep such graphs on the per-edge kernels.
// ---------------------------------------------------------------------------------------------------------
template <int LPR, int MAXPW, bool TAB = false>
__global__ __launch_bounds__(256, (TAB && LPR == 16) ? 4 : 1) void gat_halo_bwd_dst_kernel(
    const int32_t* __restrict__ list, const int32_t* __restrict__ cnt, const int2* __restrict__ rec,
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ opos, int32_t smax, const float* __restrict__ dY,
    int64_t lddy, int64_t bsdy, const float* __restrict__ Hf, int64_t ldh, int64_t bsh, const float* __restrict__ a_s,
    const float* __restrict__ a_d, const float* __restrict__ alpha, float* __restrict__ de, float* __restrict__ dad,
    float* __restrict__ part, int32_t n, int64_t Ep, int32_t B, int32_t C, int32_t ntiles,
    const int32_t* __restrict__ tab = nullptr) {
  using gcl::halo::row_bcast;
  extern __shared__ float4 img[];  // [(smax + 1) * LPR] staged h rows + zero row | sS[64 + 128 + 8]: a_s of the staged rows (4-byte DMAs of 64 lanes)
  constexpr int T = 64, NW = 4;
  constexpr int RPW = 64 / LPR, NIT = T / NW / RPW;
  constexpr int SH = LPR == 16 ? 8 : 9;
  typedef float v4f __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) const v4f* lds4_t;
  typedef __attribute__((address_space(3))) const float* ldsf_t;
  float* sS = reinterpret_cast<float*>(img + (smax + 1) * LPR);
  const int xcd = blockIdx.x & (gcl::kNumXCD - 1);
  const int J = gridDim.x >> 3, j = blockIdx.x >> 3;
  const int nsamp = (B - xcd + gcl::kNumXCD - 1) / gcl::kNumXCD;
  const int items = nsamp * ntiles;
  const int base = items / J, extra = items - base * J;
  int m = j * base + min(j, extra);
  const int mend = m + base + (j < extra ? 1 : 0);
  if (m >= mend) {  // a block without items still owns a (zero) partial record
    for (int idx = threadIdx.x; idx < 2 * LPR * 4; idx += 256) part[(size_t)blockIdx.x * (2 * LPR * 4) + idx] = 0.f;
    return;
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int sub = lane / LPR, l = lane % LPR, c0 = l * 4;
  const unsigned cb = (unsigned)c0 * 4u, ldb = (unsigned)ldh * 4u;
  const int hstride = smax - T;
  const unsigned lds0 = (unsigned)(size_t)((gcl::halo::lptr_t)img);
  const unsigned lb = lds0 + (unsigned)l * 16u;
  const unsigned ldsS = (unsigned)(size_t)((gcl::halo::lptr_t)sS);
  if (threadIdx.x < LPR) img[smax * LPR + threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (threadIdx.x == 0) sS[smax] = 0.f;

  // gradients of the attention vectors, accumulated over every row this block handles (the rows are all in the image):
  //   d att_src += sum_edges de_e h[src_e],   d att_dst += sum_rows da_d[i] h[i]      -> part[block][2][C]
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
  int jj[MAXPW], nhalo = 0, tile = -1;
  int hl0 = 0, hl1 = 0;  // wave 0: the halo list, one entry per lane (and lane + 64)
  int own[TAB ? NIT : 1];
  int2 rc[NIT];
  int rstart[NIT];
  while (true) {
    const int tnew = m / nsamp;
    const int s = m - tnew * nsamp;
    const int b = xcd + gcl::kNumXCD * s;
    const bool newtile = tnew != tile;
    if (newtile) {
      tile = tnew;
      nhalo = cnt[tile] / RPW;
      const int32_t* __restrict__ tl = list + (int64_t)tile * hstride;
#pragma unroll
      for (int q = 0; q < MAXPW; ++q) {
        const int e0 = min((wave + NW * q) * RPW, hstride - RPW);
        int jv = tl[e0];
#pragma unroll
        for (int r = 1; r < RPW; ++r) {
          const int jr = tl[e0 + r];
          jv = sub == r ? jr : jv;
        }
        jj[q] = TAB ? tab[jv] : jv;
      }
      if (wave == 0) {
        hl0 = tl[min(lane, hstride - 1)];
        hl1 = tl[min(lane + 64, hstride - 1)];
      }
      if (TAB) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const int row = tile * T + wave * (T / NW) + sub + it * RPW;
          own[TAB ? it : 0] = tab[row < n ? row : n - 1];
        }
        // the table entries arrive HERE, inside the new-tile branch: left to hipcc, the wait for these vector loads sits
        // at the join and every item then waits for the previous item's stores before it may issue its DMA
#pragma unroll
        for (int it = 0; it < NIT; ++it) asm volatile("" : "+v"(own[TAB ? it : 0]));
#pragma unroll
        for (int q = 0; q < MAXPW; ++q) asm volatile("" : "+v"(jj[q]));
      }
    }
    const char* Hc = reinterpret_cast<const char*>(Hf + (int64_t)b * bsh);
    const char* H0 = reinterpret_cast<const char*>(Hf);
    const unsigned sboff = (unsigned)b * (unsigned)bsh * 4u;
    const int row0 = tile * T + wave * (T / NW) + sub;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int row = row0 + it * RPW;
      const int sr = TAB ? own[TAB ? it : 0] : (row < n ? row : n - 1);
      // TAB: one base and a 32-bit byte offset (the whole of Hf stays below 4 GiB: checked on the host)
      const char* src = TAB ? H0 + ((sr >= 0 ? sboff + __umul24(sr, ldb) : __umul24(~sr, ldb)) + cb) : Hc + (__umul24(sr, ldb) + cb);
      __builtin_amdgcn_global_load_lds((gcl::halo::gptr_t)src, (gcl::halo::lptr_t)(img + (wave * (T / NW) + it * RPW) * LPR), 16, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < MAXPW; ++q) {
      const int p = wave + NW * q;
      if (p < nhalo) {
        const int sr = jj[q];
        const char* src = TAB ? H0 + ((sr >= 0 ? sboff + __umul24(sr, ldb) : __umul24(~sr, ldb)) + cb) : Hc + (__umul24(sr, ldb) + cb);
        __builtin_amdgcn_global_load_lds((gcl::halo::gptr_t)src, (gcl::halo::lptr_t)(img + (T + p * RPW) * LPR), 16, 0, 0);
      }
    }
    if (wave == 0) {  // a_s of the staged rows: 4 bytes per lane
      const float* asb = a_s + (int64_t)b * n;
      const int r = tile * T + lane;
      __builtin_amdgcn_global_load_lds((gcl::halo::gptr_t)(asb + (r < n ? r : n - 1)), (gcl::halo::lptr_t)sS, 4, 0, 0);
      if (nhalo > 0) __builtin_amdgcn_global_load_lds((gcl::halo::gptr_t)(asb + hl0), (gcl::halo::lptr_t)(sS + T), 4, 0, 0);
      if (nhalo * RPW > 64) __builtin_amdgcn_global_load_lds((gcl::halo::gptr_t)(asb + hl1), (gcl::halo::lptr_t)(sS + T + 64), 4, 0, 0);
    }
    if (newtile) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int row = row0 + it * RPW;
        const int rcl = row < n ? row : n - 1;
        rc[it] = rec[(int64_t)rcl * gcl::kHaloRec + (l & (gcl::kHaloRec - 1))];
        rstart[it] = rowptr[rcl];
      }
    }
    // this item's per-row operands: dy row, a_d, the lane's alpha
    float4 gy[NIT];
    float adv[NIT], alv[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int row = row0 + it * RPW;
      const int rcl = row < n ? row : n - 1;
      gy[it] = *reinterpret_cast<const float4*>(dY + (int64_t)b * bsdy + (int64_t)rcl * lddy + c0);
      adv[it] = a_d[(int64_t)b * n + rcl];
      const bool valid = (rc[it].x & gcl::kHaloPosMask) != smax;
      alv[it] = alpha[(int64_t)b * Ep + rstart[it] + (valid ? (l & (gcl::kHaloRec - 1)) : 0)];
    }
    __syncthreads();  // image + scores complete (vmcnt(0) + barrier)

#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int row = row0 + it * RPW;
      const int rx = rc[it].x;
      const int rxp = rx & gcl::kHaloPosMask;
      const int rxb = rxp << SH;
      const int last = row_bcast<15>(rx);
      const float4 g4 = gy[it];
      const float adst = adv[it];
      const bool valid = rxp != smax;
      if (!__any((last & gcl::kHaloMore) != 0)) {
        const int k15 = l & (gcl::kHaloRec - 1);
        float mine = 0.f;
        const bool wide = __any(row_bcast<8>(rxb) != (smax << SH));
#define GCL_GAT_DOT(Kk)                                                                   \
  {                                                                                       \
    const unsigned adr = (unsigned)row_bcast<Kk>(rxb) + lb;                               \
    const v4f v = *(lds4_t)adr;                                                           \
    float d = g4.x * v.x + g4.y * v.y + g4.z * v.z + g4.w * v.w;                          \
    d += row_ror<8>(d); d += row_ror<4>(d); d += row_ror<2>(d); d += row_ror<1>(d);       \
    if (LPR == 32) d += __shfl_xor(d, 16, 64);                                            \
    mine = k15 == Kk ? d : mine;                                                          \
  }
        GCL_GAT_DOT(0) GCL_GAT_DOT(1) GCL_GAT_DOT(2) GCL_GAT_DOT(3)
        GCL_GAT_DOT(4) GCL_GAT_DOT(5) GCL_GAT_DOT(6) GCL_GAT_DOT(7)
        if (wide) {
          GCL_GAT_DOT(8) GCL_GAT_DOT(9) GCL_GAT_DOT(10) GCL_GAT_DOT(11)
          GCL_GAT_DOT(12) GCL_GAT_DOT(13) GCL_GAT_DOT(14) GCL_GAT_DOT(15)
        }
#undef GCL_GAT_DOT
        const float al = valid ? alv[it] : 0.f;
        float t = al * mine;
        t += row_ror<8>(t); t += row_ror<4>(t); t += row_ror<2>(t); t += row_ror<1>(t);
        const float pre = *(ldsf_t)(ldsS + (unsigned)rxp * 4u) + adst;
        const float dev = valid ? al * (mine - t) * (pre > 0.f ? 1.f : kNegSlope) : 0.f;
        float sd = dev;
        sd += row_ror<8>(sd); sd += row_ror<4>(sd); sd += row_ror<2>(sd); sd += row_ror<1>(sd);
        if (row < n) {
          if (valid && l < gcl::kHaloRec) de[(int64_t)b * Ep + rstart[it] + l] = dev;
          if (l == 0) dad[(int64_t)b * n + row] = sd;
        }
        {
          const float live = row < n ? 1.f : 0.f;  // rows past the end repeat the last row's records
          const int dvi = __float_as_int(dev * live);
          const v4f hi = *(lds4_t)(lds0 + ((unsigned)(wave * (T / NW) + it * RPW + sub) << SH) + (unsigned)l * 16u);
          const float sdl = sd * live;
          d0 += sdl * hi.x; d1 += sdl * hi.y; d2 += sdl * hi.z; d3 += sdl * hi.w;
#define GCL_GAT_DS(Kk)                                                           \
  {                                                                              \
    const unsigned adr = (unsigned)row_bcast<Kk>(rxb) + lb;                      \
    const v4f v = *(lds4_t)adr;                                                  \
    const float dk = __int_as_float(row_bcast<Kk>(dvi));                         \
    s0 += dk * v.x; s1 += dk * v.y; s2 += dk * v.z; s3 += dk * v.w;              \
  }
          GCL_GAT_DS(0) GCL_GAT_DS(1) GCL_GAT_DS(2) GCL_GAT_DS(3)
          GCL_GAT_DS(4) GCL_GAT_DS(5) GCL_GAT_DS(6) GCL_GAT_DS(7)
          if (wide) {
            GCL_GAT_DS(8) GCL_GAT_DS(9) GCL_GAT_DS(10) GCL_GAT_DS(11)
            GCL_GAT_DS(12) GCL_GAT_DS(13) GCL_GAT_DS(14) GCL_GAT_DS(15)
          }
#undef GCL_GAT_DS
        }
      } else {
        const int rcl = row < n ? row : n - 1;
        const int st = rowptr[rcl], en = rowptr[rcl + 1];
        float t = 0.f;
        for (int e = st; e < en; ++e) {
          const v4f v = *(lds4_t)(((unsigned)opos[e] << SH) + lb);
          float d = g4.x * v.x + g4.y * v.y + g4.z * v.z + g4.w * v.w;
          for (int off = LPR >> 1; off > 0; off >>= 1) d += __shfl_xor(d, off, 64);
          t += alpha[(int64_t)b * Ep + e] * d;
        }
        float sd = 0.f;
        for (int e = st; e < en; ++e) {
          const int pk = opos[e];
          const v4f v = *(lds4_t)(((unsigned)pk << SH) + lb);
          float d = g4.x * v.x + g4.y * v.y + g4.z * v.z + g4.w * v.w;
          for (int off = LPR >> 1; off > 0; off >>= 1) d += __shfl_xor(d, off, 64);
          const float pr = sS[pk] + adst;
          const float dev = alpha[(int64_t)b * Ep + e] * (d - t) * (pr > 0.f ? 1.f : kNegSlope);
          sd += dev;
          if (l == 0 && row < n) de[(int64_t)b * Ep + e] = dev;
          if (row < n) { s0 += dev * v.x; s1 += dev * v.y; s2 += dev * v.z; s3 += dev * v.w; }
        }
        if (l == 0 && row < n) dad[(int64_t)b * n + row] = sd;
        if (row < n) {
          const v4f hi = *(lds4_t)(lds0 + ((unsigned)(wave * (T / NW) + it * RPW + sub) << SH) + (unsigned)l * 16u);
          d0 += sd * hi.x; d1 += sd * hi.y; d2 += sd * hi.z; d3 += sd * hi.w;
        }
      }
    }
    if (++m >= mend) break;
    __syncthreads();
  }
  // per-block partial of the attention-vector gradients: the 16 row groups (4 waves x RPW ... ) of a channel quad, in LDS
  __syncthreads();
  float* red = reinterpret_cast<float*>(img);  // [4 * RPW][2][LPR * 4]
  {
    float* r_ = red + (size_t)(wave * RPW + sub) * (2 * LPR * 4);
    *reinterpret_cast<float4*>(r_ + c0) = make_float4(s0, s1, s2, s3);
    *reinterpret_cast<float4*>(r_ + LPR * 4 + c0) = make_float4(d0, d1, d2, d3);
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 2 * LPR * 4; idx += 256) {
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < 4 * RPW; ++q) acc += red[(size_t)q * (2 * LPR * 4) + idx];
    part[(size_t)blockIdx.x * (2 * LPR * 4) + idx] = acc;
  }
}

template <int LPR, int MAXPW>
__global__ __launch_bounds__(256) void gat_halo_bwd_src_kernel(
    const int32_t* __restrict__ list, const int32_t* __restrict__ cnt, const int2* __restrict__ rec,
    const int32_t* __restrict__ trowptr, const int32_t* __restrict__ opos, const int32_t* __restrict__ tslot, int32_t smax,
    const float* __restrict__ dY, int64_t lddy, int64_t bsdy, const float* __restrict__ alpha, const float* __restrict__ de,
    const float* __restrict__ dad, const float* __restrict__ att_s, const float* __restrict__ att_d, float* __restrict__ das,
    float* __restrict__ dH, int64_t lddh, int64_t bsdh, float* __restrict__ part_b, int32_t n, int64_t Ep, int32_t B,
    int32_t C, int32_t ntiles) {
  using gcl::halo::row_bcast;
  extern __shared__ float4 img[];  // [(smax + 1) * LPR] staged dy rows + zero row
  constexpr int T = 64, NW = 4;
  constexpr int RPW = 64 / LPR, NIT = T / NW / RPW;
  constexpr int SH = LPR == 16 ? 8 : 9;
  typedef float v4f __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) const v4f* lds4_t;
  const int xcd = blockIdx.x & (gcl::kNumXCD - 1);
  const int J = gridDim.x >> 3, j = blockIdx.x >> 3;
  const int nsamp = (B - xcd + gcl::kNumXCD - 1) / gcl::kNumXCD;
  const int items = nsamp * ntiles;
  const int base = items / J, extra = items - base * J;
  int m = j * base + min(j, extra);
  const int mend = m + base + (j < extra ? 1 : 0);
  if (m >= mend) {
    if (part_b)
      for (int idx = threadIdx.x; idx < LPR * 4; idx += 256) part_b[(size_t)blockIdx.x * (LPR * 4) + idx] = 0.f;
    return;
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int sub = lane / LPR, l = lane % LPR, c0 = l * 4;
  const unsigned cb = (unsigned)c0 * 4u, ldb = (unsigned)lddy * 4u;
  float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;  // d bias = column sums of dy: the tile's own rows are in the image
  const int hstride = smax - T;
  const unsigned lds0 = (unsigned)(size_t)((gcl::halo::lptr_t)img);
  const unsigned lb = lds0 + (unsigned)l * 16u;
  if (threadIdx.x < LPR) img[smax * LPR + threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 as4 = *reinterpret_cast<const float4*>(att_s + c0), ad4 = *reinterpret_cast<const float4*>(att_d + c0);

  int jj[MAXPW], nhalo = 0, tile = -1;
  int2 rc[NIT];
  while (true) {
    const int tnew = m / nsamp;
    const int s = m - tnew * nsamp;
    const int b = xcd + gcl::kNumXCD * s;
    const bool newtile = tnew != tile;
    if (newtile) {
      tile = tnew;
      nhalo = cnt[tile] / RPW;
      const int32_t* __restrict__ tl = list + (int64_t)tile * hstride;
#pragma unroll
      for (int q = 0; q < MAXPW; ++q) {
        const int e0 = min((wave + NW * q) * RPW, hstride - RPW);
        int jv = tl[e0];
#pragma unroll
        for (int r = 1; r < RPW; ++r) {
          const int jr = tl[e0 + r];
          jv = sub == r ? jr : jv;
        }
        jj[q] = jv;
      }
    }
    const char* Dc = reinterpret_cast<const char*>(dY + (int64_t)b * bsdy);
    const int row0 = tile * T + wave * (T / NW) + sub;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int row = row0 + it * RPW;
      const char* src = Dc + (__umul24(row < n ? row : n - 1, ldb) + cb);
      __builtin_amdgcn_global_load_lds((gcl::halo::gptr_t)src, (gcl::halo::lptr_t)(img + (wave * (T / NW) + it * RPW) * LPR), 16, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < MAXPW; ++q) {
      const int p = wave + NW * q;
      if (p < nhalo) {
        const char* src = Dc + (__umul24(jj[q], ldb) + cb);
        __builtin_amdgcn_global_load_lds((gcl::halo::gptr_t)src, (gcl::halo::lptr_t)(img + (T + p * RPW) * LPR), 16, 0, 0);
      }
    }
    if (newtile) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int row = row0 + it * RPW;
        rc[it] = rec[(int64_t)(row < n ? row : n - 1) * gcl::kHaloRec + (l & (gcl::kHaloRec - 1))];
      }
    }
    // this item's per-edge operands of the lane's slot (by forward CSR slot) and the row's da_d
    float alv[NIT], dev[NIT], ddv[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int row = row0 + it * RPW;
      const bool valid = (rc[it].x & gcl::kHaloPosMask) != smax;
      const int64_t sl = (int64_t)b * Ep + (valid ? rc[it].y : 0);
      alv[it] = alpha[sl];
      dev[it] = de[sl];
      ddv[it] = dad[(int64_t)b * n + (row < n ? row : n - 1)];
    }
    __syncthreads();

    float* __restrict__ Ob = dH + (int64_t)b * bsdh;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int row = row0 + it * RPW;
      const int rx = rc[it].x;
      const int rxp = rx & gcl::kHaloPosMask;
      const int rxb = rxp << SH;
      const int last = row_bcast<15>(rx);
      const bool valid = rxp != smax;
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, sde;
      if (!__any((last & gcl::kHaloMore) != 0)) {
        const int ali = __float_as_int(valid ? alv[it] : 0.f);
        sde = valid ? dev[it] : 0.f;
        sde += row_ror<8>(sde); sde += row_ror<4>(sde); sde += row_ror<2>(sde); sde += row_ror<1>(sde);
        const bool wide = __any(row_bcast<8>(rxb) != (smax << SH));
#define GCL_GAT_ACC(Kk)                                                          \
  {                                                                              \
    const unsigned adr = (unsigned)row_bcast<Kk>(rxb) + lb;                      \
    const v4f v = *(lds4_t)adr;                                                  \
    const float al = __int_as_float(row_bcast<Kk>(ali));                         \
    a0 += al * v.x; a1 += al * v.y; a2 += al * v.z; a3 += al * v.w;              \
  }
        GCL_GAT_ACC(0) GCL_GAT_ACC(1) GCL_GAT_ACC(2) GCL_GAT_ACC(3)
        GCL_GAT_ACC(4) GCL_GAT_ACC(5) GCL_GAT_ACC(6) GCL_GAT_ACC(7)
        if (wide) {
          GCL_GAT_ACC(8) GCL_GAT_ACC(9) GCL_GAT_ACC(10) GCL_GAT_ACC(11)
          GCL_GAT_ACC(12) GCL_GAT_ACC(13) GCL_GAT_ACC(14) GCL_GAT_ACC(15)
        }
#undef GCL_GAT_ACC
      } else {
        const int rcl = row < n ? row : n - 1;
        sde = 0.f;
        for (int e = trowptr[rcl]; e < trowptr[rcl + 1]; ++e) {
          const int64_t sl = (int64_t)b * Ep + tslot[e];
          const float al = alpha[sl];
          sde += de[sl];
          const v4f v = *(lds4_t)(((unsigned)opos[e] << SH) + lb);
          a0 += al * v.x; a1 += al * v.y; a2 += al * v.z; a3 += al * v.w;
        }
      }
      if (row < n) {
        if (part_b) {
          const v4f own = *(lds4_t)(lds0 + ((unsigned)(wave * (T / NW) + it * RPW + sub) << SH) + (unsigned)l * 16u);
          b0 += own.x; b1 += own.y; b2 += own.z; b3 += own.w;
        }
        const float dd = ddv[it];
        a0 += sde * as4.x + dd * ad4.x;
        a1 += sde * as4.y + dd * ad4.y;
        a2 += sde * as4.z + dd * ad4.z;
        a3 += sde * as4.w + dd * ad4.w;
        *reinterpret_cast<float4*>(Ob + (int64_t)row * lddh + c0) = make_float4(a0, a1, a2, a3);
        if (l == 0) das[(int64_t)b * n + row] = sde;
      }
    }
    if (++m >= mend) break;
    __syncthreads();
  }
  if (part_b) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(img);  // [4 * RPW][LPR * 4]
    *reinterpret_cast<float4*>(red + (size_t)(wave * RPW + sub) * (LPR * 4) + c0) = make_float4(b0, b1, b2, b3);
    __syncthreads();
    for (int idx = threadIdx.x; idx < LPR * 4; idx += 256) {
      float acc = 0.f;
#pragma unroll
      for (int q = 0; q < 4 * RPW; ++q) acc += red[(size_t)q * (LPR * 4) + idx];
      part_b[(size_t)blockIdx.x * (LPR * 4) + idx] = acc;
    }
  }
}

__global__ __launch_bounds__(256) void alpha_reorder_kernel(const int32_t* __restrict__ eperm,
                                                            const float* __restrict__ a_slots,
                                                            float* __restrict__ a_edges, int64_t Ep, int32_t H) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= Ep * H) return;
  const int64_t s = idx / H;
  const int k = (int)(idx - s * H);
  a_edges[(int64_t)eperm[s] * H + k] = a_slots[idx];
}

// keep-mask as one 64-bit ballot word per wave of edges
__global__ __launch_bounds__(256) void prune_ballot_kernel(const float* __restrict__ a_edges, float thr, int64_t Ep,
                                                           unsigned long long* __restrict__ words) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool keep = idx < Ep && a_edges[idx] >= thr;
  const unsigned long long m = __ballot(keep);
  if ((threadIdx.x & 63) == 0 && (idx >> 6) < (Ep + 63) / 64) words[idx >> 6] = m;
}

constexpr int kGatBlocks = 512;

// Head chunking: the kernels map ALL channels of the heads they handle onto one lane group (<= 64 lanes x 4
// channels), so one launch takes a power-of-two number of heads Hc with Hc * C <= 256.  A layer with more heads, or
// a head count that is not a power of two (the reference reports 8 and 33 heads at C = 64, README.md:148-150),
// runs as several launches over column blocks of h, with per-head arrays strided by the total head count.
// head width rounded up to a power-of-two number of 4-channel lanes (lane_map)
inline int padded_width(int C) {
  int lph = 1;
  while (lph * 4 < C) lph *= 2;
  return lph * 4;
}
int gat_check(const gcl_graph_t* g, int32_t H, int32_t C, int* chunk) {
  GCL_CHECK_ARG(g, "gat: null graph");
  GCL_CHECK_ARG(H >= 1 && C >= 4 && (C % 4) == 0, "gat: C must be a positive multiple of 4 (H=%d C=%d)", H, C);
  if (C > 256) {
    gcl::set_error("gat: unsupported head width C=%d (C <= 256)", C);
    return GCL_EUNSUPPORTED;
  }
  const int Cp = padded_width(C);
  int hc = 1;
  while (hc * 2 <= H && hc * 2 * Cp <= 256) hc *= 2;
  *chunk = hc;  // the last chunk(s) of a non-power-of-two H are smaller powers of two
  return GCL_OK;
}
inline int lpr_for(int heads, int C) {
  const int lanes = heads * (padded_width(C) / 4);
  return lanes <= 4 ? 4 : lanes <= 8 ? 8 : lanes <= 16 ? 16 : lanes <= 32 ? 32 : 64;
}
// heads of the chunk that starts at h0
inline int chunk_heads(int H, int h0, int maxc) {
  int hc = maxc;
  while (hc > H - h0) hc >>= 1;
  return hc;
}

}  // namespace

#define GCL_DISPATCH_LPR(lpr, CALL) \
  switch (lpr) {                    \
    case 4: CALL(4); break;         \
    case 8: CALL(8); break;         \
    case 16: CALL(16); break;       \
    case 32: CALL(32); break;       \
    default: CALL(64); break;       \
  }

static int gat_fwd_impl(const gcl_graph_t* g, const float* h, int64_t ldh, int64_t bsh, const int32_t* tab,
                        const float* att_src, const float* att_dst, const float* bias, float* a_src, float* a_dst,
                        float* alpha, float* y, int64_t ldy, int64_t bsy, int32_t B, int32_t H, int32_t C,
                        gcl_stream_t stream) {
  int maxc = 0;
  int rc = gat_check(g, H, C, &maxc);
  if (rc) return rc;
  GCL_CHECK_ARG(h && att_src && att_dst && a_src && a_dst && y, "gat_fwd: null argument");
  GCL_CHECK_ARG(B > 0 && ldh >= H * C && ldy >= C, "gat_fwd: bad shape");
  GCL_CHECK_ARG((ldh % 4) == 0 && (bsh % 4) == 0 && gcl::aligned16(h), "gat_fwd: h must be 16-B aligned with ld %% 4 == 0");
  GCL_CHECK_ARG((ldy % 4) == 0 && (bsy % 4) == 0 && gcl::aligned16(y), "gat_fwd: y must be 16-B aligned with ld %% 4 == 0");
  GCL_CHECK_ARG(g->kind == GCL_GRAPH_GAT, "gat_fwd: graph was not created with GCL_GRAPH_GAT");
  hipStream_t st = (hipStream_t)stream;
  {
    // one head on a graph with a tile layout: everything of the layer from one LDS image per tile (gat_halo_fwd_kernel)
    const char* ev = getenv("GCL_GAT_HALO");  // read per call: the parity test compares the two forms
    const gcl_halo& hl = g->halo[0][0];
    const int lprh = C / 4;
    const int64_t ldsb = (int64_t)(hl.smax + 1) * lprh * 16 + (int64_t)(hl.smax + 1 + 64) * 4;
    if (!(ev && atoi(ev) == 0) && H == 1 && (C == 64 || C == 128) && hl.T == 64 && g->n_heavy == 0 && ldsb <= 80 * 1024 &&
        (int64_t)g->n * ldh * 4 < ((int64_t)1 << 31) && ldh * 4 < (1 << 24) && g->n < (1 << 24) && gcl::aligned16(att_src) &&
        gcl::aligned16(att_dst) && (!bias || gcl::aligned16(bias))) {
      const int rpw = 64 / lprh;
      const int mpw = (int)gcl::cdiv((hl.smax - 64) / rpw, 4);
      if (mpw <= 16) {
        const int per_cu = (int)std::min<int64_t>(8, (160 * 1024) / ldsb);
        dim3 grid((unsigned)(gcl::kNumXCD * 32 * per_cu));
        auto go = [&](auto kern) -> int {
          const int rc2 = gcl::ensure_dyn_lds((const void*)kern, (size_t)ldsb);
          if (rc2) return rc2;
          hipLaunchKernelGGL(kern, grid, dim3(256), (size_t)ldsb, st, hl.list, hl.cnt, reinterpret_cast<const int2*>(hl.rec),
                             g->rowptr, hl.opos, hl.smax, h, ldh, bsh, att_src, att_dst, bias, a_src, a_dst, alpha, y, ldy,
                             bsy, g->n, g->e, B, C, hl.ntiles, tab);
          return GCL_OK;
        };
        int rc2;
        if (tab) {
          if (lprh == 16) rc2 = mpw <= 4 ? go(&gat_halo_fwd_kernel<16, 4, true>) : mpw <= 8 ? go(&gat_halo_fwd_kernel<16, 8, true>) : go(&gat_halo_fwd_kernel<16, 16, true>);
          else rc2 = mpw <= 4 ? go(&gat_halo_fwd_kernel<32, 4, true>) : mpw <= 8 ? go(&gat_halo_fwd_kernel<32, 8, true>) : go(&gat_halo_fwd_kernel<32, 16, true>);
        } else if (lprh == 16) rc2 = mpw <= 4 ? go(&gat_halo_fwd_kernel<16, 4>) : mpw <= 8 ? go(&gat_halo_fwd_kernel<16, 8>) : go(&gat_halo_fwd_kernel<16, 16>);
        else rc2 = mpw <= 4 ? go(&gat_halo_fwd_kernel<32, 4>) : mpw <= 8 ? go(&gat_halo_fwd_kernel<32, 8>) : go(&gat_halo_fwd_kernel<32, 16>);
        if (rc2) return rc2;
        GCL_CHECK_LAUNCH();
        return GCL_OK;
      }
    }
  }
  if (tab) {
    gcl::set_error("gat_fwd_tab: this graph / shape has no source-tile form (H=%d C=%d)", H, C);
    return GCL_EUNSUPPORTED;
  }
  for (int h0 = 0; h0 < H;) {
    const int hc = chunk_heads(H, h0, maxc);
    const int lpr = lpr_for(hc, C);
    const float* hh = h + (int64_t)h0 * C;  // column block of this chunk's heads
    const int rpb = (64 / lpr) * 4;
    const int64_t rows = (int64_t)B * g->n;
    int64_t nbs = gcl::cdiv(rows, rpb);
    if (nbs > 4096) nbs = 4096;
#define CALL(L)                                                                                                     \
  hipLaunchKernelGGL((gat_scores_kernel<L>), dim3((unsigned)nbs), dim3(256), 0, st, hh, ldh, bsh, att_src + h0 * C, \
                     att_dst + h0 * C, a_src, a_dst, g->n, B, hc, C, padded_width(C), H, h0)
    GCL_DISPATCH_LPR(lpr, CALL)
#undef CALL
    GCL_CHECK_LAUNCH();
    const int32_t nRB = (int32_t)gcl::cdiv(g->n, rpb);
    const int xcd_map = B >= 8 ? 1 : 0;
    const int64_t nb = xcd_map ? (int64_t)8 * gcl::cdiv(B, 8) * nRB : (int64_t)B * nRB;
#define CALL(L)                                                                                                       \
  hipLaunchKernelGGL((gat_fwd_kernel<L>), dim3((unsigned)nb), dim3(256), 0, st, g->rowptr, g->col, g->ecol, hh, ldh,  \
                     bsh, a_src, a_dst, h0 == 0 ? bias : nullptr, alpha, y, ldy, bsy, g->n, g->e, B, hc, C,           \
                     padded_width(C), nRB, xcd_map, H, h0, H, h0 > 0 ? 1 : 0)
    GCL_DISPATCH_LPR(lpr, CALL)
#undef CALL
    GCL_CHECK_LAUNCH();
    h0 += hc;
  }
  return GCL_OK;
}

extern "C" int gcl_gat_fwd(const gcl_graph_t* g, const float* h, int64_t ldh, int64_t bsh, const float* att_src,
                           const float* att_dst, const float* bias, float* a_src, float* a_dst, float* alpha,
                           float* y, int64_t ldy, int64_t bsy, int32_t B, int32_t H, int32_t C, gcl_stream_t stream) {
  return gat_fwd_impl(g, h, ldh, bsh, nullptr, att_src, att_dst, bias, a_src, a_dst, alpha, y, ldy, bsy, B, H, C, stream);
}

// The rows of h read through a row table (gat_halo_*_kernel<.., TAB>): row i of sample b is row tab[i] of sample b of h
// when tab[i] >= 0, the batch-invariant flat row ~tab[i] of h otherwise.  One head on a source-tile graph only
// (gcl_gat_tab_ok tells beforehand); GCL_EUNSUPPORTED elsewhere.
extern "C" int gcl_gat_fwd_tab(const gcl_graph_t* g, const float* h, int64_t ldh, int64_t bsh, const int32_t* tab,
                               const float* att_src, const float* att_dst, const float* bias, float* a_src,
                               float* a_dst, float* alpha, float* y, int64_t ldy, int64_t bsy, int32_t B, int32_t H,
                               int32_t C, gcl_stream_t stream) {
  GCL_CHECK_ARG(tab, "gat_fwd_tab: null row table");
  GCL_CHECK_ARG((int64_t)B * bsh * 4 < ((int64_t)1 << 32), "gat_fwd_tab: h must stay below 4 GiB (32-bit row offsets)");
  return gat_fwd_impl(g, h, ldh, bsh, tab, att_src, att_dst, bias, a_src, a_dst, alpha, y, ldy, bsy, B, H, C, stream);
}

extern "C" int gcl_gat_tab_ok(const gcl_graph_t* g, int64_t ldh, int32_t H, int32_t C) {
  if (!g || g->kind != GCL_GRAPH_GAT || H != 1 || !(C == 64 || C == 128) || g->n_heavy || g->n_theavy) return 0;
  const char* ev = getenv("GCL_GAT_HALO");
  if (ev && atoi(ev) == 0) return 0;
  const gcl_halo& hf = g->halo[0][0];
  const gcl_halo& ht = g->halo[1][0];
  if (hf.T != 64 || ht.T != 64) return 0;
  const int lprh = C / 4, rpw = 64 / lprh;
  const int64_t ldsf = (int64_t)(hf.smax + 1) * lprh * 16 + (int64_t)(hf.smax + 1 + 64) * 4;
  const int64_t ldsd = (int64_t)(hf.smax + 1) * lprh * 16 + (int64_t)(64 + 128 + 8) * 4;
  const int64_t ldss = (int64_t)(ht.smax + 1) * lprh * 16;
  const int mpd = (int)gcl::cdiv((hf.smax - 64) / rpw, 4), mps = (int)gcl::cdiv((ht.smax - 64) / rpw, 4);
  return ldsf <= 80 * 1024 && ldsd <= 80 * 1024 && ldss <= 80 * 1024 && hf.smax - 64 <= 128 && mpd <= 16 && mps <= 16 &&
                 ldh * 4 < (1 << 24) && g->n < (1 << 24)
             ? 1
             : 0;
}

extern "C" size_t gcl_gat_bwd_ws_bytes(int64_t e_prime, int32_t n, int32_t B, int32_t H, int32_t C) {
  const size_t de = (size_t)B * e_prime * H;
  const size_t nodes = (size_t)B * n * H * 2;  // da_d, da_s
  const size_t parts = (size_t)(kGatBlocks > 2048 ? kGatBlocks : 2048) * 2 * H * C + (size_t)2048 * C;  // partial records (<= 2048 blocks) + colsum scratch / d_bias partials
  return (de + nodes + parts) * sizeof(float) + 256;
}

extern "C" int gcl_colsum(const float*, int64_t, int64_t, int32_t, float*, int32_t, void*, size_t, gcl_stream_t);
extern "C" size_t gcl_colsum_ws_bytes(int64_t, int32_t);

static int gat_bwd_impl(const gcl_graph_t* g, const float* dy, int64_t lddy, int64_t bsdy, const float* h, int64_t ldh,
                        int64_t bsh, const int32_t* tab, const float* att_src, const float* att_dst, const float* a_src,
                        const float* a_dst, const float* alpha, float* dh, int64_t lddh, int64_t bsdh, float* d_att_src,
                        float* d_att_dst, float* d_bias, int32_t accumulate, int32_t B, int32_t H, int32_t C, void* ws,
                        size_t ws_bytes, gcl_stream_t stream) {
  int maxc = 0;
  int rc = gat_check(g, H, C, &maxc);
  if (rc) return rc;
  GCL_CHECK_ARG(dy && h && att_src && att_dst && a_src && a_dst && alpha && dh && d_att_src && d_att_dst,
                "gat_bwd: null argument");
  GCL_CHECK_ARG(B > 0 && ldh >= H * C && lddh >= H * C && lddy >= C, "gat_bwd: bad shape");
  GCL_CHECK_ARG((ldh % 4) == 0 && (bsh % 4) == 0 && gcl::aligned16(h), "gat_bwd: h must be 16-B aligned with ld %% 4 == 0");
  GCL_CHECK_ARG((lddh % 4) == 0 && (bsdh % 4) == 0 && gcl::aligned16(dh), "gat_bwd: dh must be 16-B aligned with ld %% 4 == 0");
  GCL_CHECK_ARG(g->kind == GCL_GRAPH_GAT, "gat_bwd: graph was not created with GCL_GRAPH_GAT");
  GCL_CHECK_ARG(bsdy == (int64_t)g->n * lddy || B == 1, "gat_bwd: dy must be row-contiguous across the batch");
  GCL_CHECK_ARG(ws && ws_bytes >= gcl_gat_bwd_ws_bytes(g->e, g->n, B, H, C), "gat_bwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  float* de = (float*)ws;
  float* dad = de + (size_t)B * g->e * H;
  float* das = dad + (size_t)B * g->n * H;
  float* part = das + (size_t)B * g->n * H;
  float* cs_ws = part + (size_t)(kGatBlocks > 2048 ? kGatBlocks : 2048) * 2 * H * C;
  const int xcd_map = B >= 8 ? 1 : 0;
  const int vdy = (lddy % 4 == 0) && (bsdy % 4 == 0) && (C % 4 == 0) && gcl::aligned16(dy);
  const int64_t rows = (int64_t)B * g->n;
  bool halo_done = false;
  int halo_parts = 0;  // partial records of the attention-vector gradients written by the staged dst-side kernel
  int halo_bparts = 0;  // partial records of d_bias (column sums of dy) written by the staged src-side kernel
  {
    // one head on a graph with tile layouts in both directions: both edge passes from LDS images (gat_halo_bwd_*_kernel)
    const char* ev = getenv("GCL_GAT_HALO");
    const gcl_halo& hf = g->halo[0][0];
    const gcl_halo& ht = g->halo[1][0];
    const int lprh = C / 4;
    const int64_t ldsd = (int64_t)(hf.smax + 1) * lprh * 16 + (int64_t)(64 + 128 + 8) * 4;  // image + a_s of up to 64 + 128 staged rows
    const int64_t ldss = (int64_t)(ht.smax + 1) * lprh * 16;
    if (!(ev && atoi(ev) == 0) && H == 1 && (C == 64 || C == 128) && hf.T == 64 && ht.T == 64 && g->n_heavy == 0 &&
        g->n_theavy == 0 && hf.smax - 64 <= 128 && ldsd <= 80 * 1024 && ldss <= 80 * 1024 && vdy &&
        (int64_t)g->n * ldh * 4 < ((int64_t)1 << 31) && ldh * 4 < (1 << 24) && (int64_t)g->n * lddy * 4 < ((int64_t)1 << 31) &&
        lddy * 4 < (1 << 24) && g->n < (1 << 24) && gcl::aligned16(att_src) && gcl::aligned16(att_dst)) {
      const int rpw = 64 / lprh;
      const int mpd = (int)gcl::cdiv((hf.smax - 64) / rpw, 4), mps = (int)gcl::cdiv((ht.smax - 64) / rpw, 4);
      if (mpd <= 16 && mps <= 16) {
        auto god = [&](auto kern) -> int {
          const int rc2 = gcl::ensure_dyn_lds((const void*)kern, (size_t)ldsd);
          if (rc2) return rc2;
          const int per_cu = (int)std::min<int64_t>(8, (160 * 1024) / ldsd);
          hipLaunchKernelGGL(kern, dim3((unsigned)(gcl::kNumXCD * 32 * per_cu)), dim3(256), (size_t)ldsd, st, hf.list, hf.cnt,
                             reinterpret_cast<const int2*>(hf.rec), g->rowptr, hf.opos, hf.smax, dy, lddy, bsdy, h, ldh, bsh,
                             a_src, a_dst, alpha, de, dad, part, g->n, g->e, B, C, hf.ntiles, tab);
          halo_parts = (int)(gcl::kNumXCD * 32 * per_cu);
          return GCL_OK;
        };
        auto gos = [&](auto kern) -> int {
          const int rc2 = gcl::ensure_dyn_lds((const void*)kern, (size_t)ldss);
          if (rc2) return rc2;
          const int per_cu = (int)std::min<int64_t>(8, (160 * 1024) / ldss);
          hipLaunchKernelGGL(kern, dim3((unsigned)(gcl::kNumXCD * 32 * per_cu)), dim3(256), (size_t)ldss, st, ht.list, ht.cnt,
                             reinterpret_cast<const int2*>(ht.rec), g->trowptr, ht.opos, g->tslot, ht.smax, dy, lddy, bsdy, alpha,
                             de, dad, att_src, att_dst, das, dh, lddh, bsdh, d_bias ? cs_ws : nullptr, g->n, g->e, B, C, ht.ntiles);
          halo_bparts = d_bias ? (int)(gcl::kNumXCD * 32 * per_cu) : 0;
          return GCL_OK;
        };
        int rc2;
        if (tab) {
          if (lprh == 16) rc2 = mpd <= 4 ? god(&gat_halo_bwd_dst_kernel<16, 4, true>) : mpd <= 8 ? god(&gat_halo_bwd_dst_kernel<16, 8, true>) : god(&gat_halo_bwd_dst_kernel<16, 16, true>);
          else rc2 = mpd <= 4 ? god(&gat_halo_bwd_dst_kernel<32, 4, true>) : mpd <= 8 ? god(&gat_halo_bwd_dst_kernel<32, 8, true>) : god(&gat_halo_bwd_dst_kernel<32, 16, true>);
        } else if (lprh == 16) rc2 = mpd <= 4 ? god(&gat_halo_bwd_dst_kernel<16, 4>) : mpd <= 8 ? god(&gat_halo_bwd_dst_kernel<16, 8>) : god(&gat_halo_bwd_dst_kernel<16, 16>);
        else rc2 = mpd <= 4 ? god(&gat_halo_bwd_dst_kernel<32, 4>) : mpd <= 8 ? god(&gat_halo_bwd_dst_kernel<32, 8>) : god(&gat_halo_bwd_dst_kernel<32, 16>);
        if (rc2) return rc2;
        GCL_CHECK_LAUNCH();
        if (lprh == 16) rc2 = mps <= 4 ? gos(&gat_halo_bwd_src_kernel<16, 4>) : mps <= 8 ? gos(&gat_halo_bwd_src_kernel<16, 8>) : gos(&gat_halo_bwd_src_kernel<16, 16>);
        else rc2 = mps <= 4 ? gos(&gat_halo_bwd_src_kernel<32, 4>) : mps <= 8 ? gos(&gat_halo_bwd_src_kernel<32, 8>) : gos(&gat_halo_bwd_src_kernel<32, 16>);
        if (rc2) return rc2;
        GCL_CHECK_LAUNCH();
        halo_done = true;
      }
    }
  }
  if (tab && !halo_done) {
    gcl::set_error("gat_bwd_tab: this graph / shape has no source-tile form (H=%d C=%d)", H, C);
    return GCL_EUNSUPPORTED;
  }
  for (int h0 = 0; h0 < H;) {
    const int hc = chunk_heads(H, h0, maxc);
    const int lpr = lpr_for(hc, C);
    const int rpb = (64 / lpr) * 4;
    const int32_t nRB = (int32_t)gcl::cdiv(g->n, rpb);
    const unsigned nb = (unsigned)(xcd_map ? (int64_t)8 * gcl::cdiv(B, 8) * nRB : (int64_t)B * nRB);
    const float* hh = h + (int64_t)h0 * C;
    if (!halo_done) {
#define CALL(L)                                                                                                    \
  hipLaunchKernelGGL((gat_bwd_dst_kernel<L>), dim3(nb), dim3(256), 0, st, g->rowptr, g->col, g->ecol, dy, lddy,     \
                     bsdy, hh, ldh, bsh, a_src, a_dst, alpha, de, dad, g->n, g->e, B, hc, C, padded_width(C), nRB,  \
                     xcd_map, H, h0, H)
      GCL_DISPATCH_LPR(lpr, CALL)
#undef CALL
      GCL_CHECK_LAUNCH();
#define CALL(L)                                                                                                       \
  hipLaunchKernelGGL((gat_bwd_src_kernel<L>), dim3(nb), dim3(256), 0, st, g->trowptr, g->tcol, g->tslot, g->tecol,    \
                     g->teslot, dy, lddy, bsdy, alpha, de, dad, att_src + h0 * C, att_dst + h0 * C, das,              \
                     dh + (int64_t)h0 * C, lddh, bsdh, g->n, g->e, B, hc, C, padded_width(C), nRB, vdy, xcd_map, H,   \
                     h0, H)
      GCL_DISPATCH_LPR(lpr, CALL)
#undef CALL
      GCL_CHECK_LAUNCH();
    }
    int64_t nbd = gcl::cdiv(rows, rpb);
    if (nbd > kGatBlocks) nbd = kGatBlocks;
    if (halo_done) {
      nbd = halo_parts;  // the staged dst-side kernel already left one record per block (H = 1: [2][C])
    } else {
#define CALL(L)                                                                                                    \
  hipLaunchKernelGGL((gat_datt_kernel<L>), dim3((unsigned)nbd), dim3(256), 0, st, hh, ldh, bsh, das, dad, part, g->n, \
                     B, hc, C, padded_width(C), H, h0)
      GCL_DISPATCH_LPR(lpr, CALL)
#undef CALL
      GCL_CHECK_LAUNCH();
    }
    const int HC = hc * C;
    rc = gcl::launch_reduce_parts2(part, (int)nbd, 2 * HC, HC, HC, d_att_src + h0 * C, d_att_dst + h0 * C, HC, accumulate,
                                   st);
    if (rc) return rc;
    h0 += hc;
  }
  if (d_bias && halo_bparts > 0) {
    rc = gcl::launch_reduce_parts(cs_ws, halo_bparts, C, C, d_bias, C, 1, C, accumulate, st);
    if (rc) return rc;
  } else if (d_bias) {
    // dy rows are contiguous across the batch (checked above): one flat column sum
    rc = gcl_colsum(dy, lddy, rows, C, d_bias, accumulate, cs_ws, gcl_colsum_ws_bytes(rows, C), stream);
    if (rc) return rc;
  }
  return GCL_OK;
}

extern "C" int gcl_gat_bwd(const gcl_graph_t* g, const float* dy, int64_t lddy, int64_t bsdy, const float* h,
                           int64_t ldh, int64_t bsh, const float* att_src, const float* att_dst, const float* a_src,
                           const float* a_dst, const float* alpha, float* dh, int64_t lddh, int64_t bsdh,
                           float* d_att_src, float* d_att_dst, float* d_bias, int32_t accumulate, int32_t B, int32_t H,
                           int32_t C, void* ws, size_t ws_bytes, gcl_stream_t stream) {
  return gat_bwd_impl(g, dy, lddy, bsdy, h, ldh, bsh, nullptr, att_src, att_dst, a_src, a_dst, alpha, dh, lddh, bsdh,
                      d_att_src, d_att_dst, d_bias, accumulate, B, H, C, ws, ws_bytes, stream);
}

// dh[b, i] is the gradient of the TABLE-READ row i of sample b (dense [B, n, C]); the caller folds it back onto the
// rows behind the table (gather for tab >= 0, batch sum for the shared rows).
extern "C" int gcl_gat_bwd_tab(const gcl_graph_t* g, const float* dy, int64_t lddy, int64_t bsdy, const float* h,
                               int64_t ldh, int64_t bsh, const int32_t* tab, const float* att_src, const float* att_dst,
                               const float* a_src, const float* a_dst, const float* alpha, float* dh, int64_t lddh,
                               int64_t bsdh, float* d_att_src, float* d_att_dst, float* d_bias, int32_t accumulate,
                               int32_t B, int32_t H, int32_t C, void* ws, size_t ws_bytes, gcl_stream_t stream) {
  GCL_CHECK_ARG(tab, "gat_bwd_tab: null row table");
  GCL_CHECK_ARG((int64_t)B * bsh * 4 < ((int64_t)1 << 32), "gat_bwd_tab: h must stay below 4 GiB (32-bit row offsets)");
  return gat_bwd_impl(g, dy, lddy, bsdy, h, ldh, bsh, tab, att_src, att_dst, a_src, a_dst, alpha, dh, lddh, bsdh, d_att_src,
                      d_att_dst, d_bias, accumulate, B, H, C, ws, ws_bytes, stream);
}

extern "C" int gcl_gat_alpha_to_edge_order(const gcl_graph_t* g, const float* alpha_slots, float* alpha_edges,
                                           int32_t H, gcl_stream_t stream) {
  GCL_CHECK_ARG(g && alpha_slots && alpha_edges && H >= 1, "gat_alpha_to_edge_order: bad argument");
  const int64_t total = g->e * H;
  hipLaunchKernelGGL(alpha_reorder_kernel, dim3((unsigned)gcl::cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     g->eperm, alpha_slots, alpha_edges, g->e, H);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

extern "C" size_t gcl_gat_prune_ws_bytes(int64_t e_prime) { return (size_t)((e_prime + 63) / 64 + 1) * 8; }

extern "C" int gcl_gat_prune(const gcl_graph_t* g, const float* alpha_edges, float threshold, int64_t* edge_index_out,
                             int64_t* kept, void* ws, size_t ws_bytes, gcl_stream_t stream) {
  GCL_CHECK_ARG(g && alpha_edges && edge_index_out && kept, "gat_prune: null argument");
  GCL_CHECK_ARG(ws && ws_bytes >= gcl_gat_prune_ws_bytes(g->e), "gat_prune: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int64_t Ep = g->e, nw = (Ep + 63) / 64;
  hipLaunchKernelGGL(prune_ballot_kernel, dim3((unsigned)gcl::cdiv(nw * 64, 256)), dim3(256), 0, st, alpha_edges,
                     threshold, Ep, (unsigned long long*)ws);
  GCL_CHECK_LAUNCH();
  std::vector<unsigned long long> words((size_t)nw);
  GCL_CHECK_HIP(hipMemcpyAsync(words.data(), ws, (size_t)nw * 8, hipMemcpyDeviceToHost, st));
  GCL_CHECK_HIP(hipStreamSynchronize(st));
  // prefix compaction of the ballot words in PyG edge order
  int64_t k = 0;
  for (int64_t wi = 0; wi < nw; ++wi) k += __builtin_popcountll(words[(size_t)wi]);
  int64_t o = 0;
  for (int64_t e = 0; e < Ep; ++e)
    if ((words[(size_t)(e >> 6)] >> (e & 63)) & 1ull) {
      edge_index_out[o] = g->h_edges[e];
      edge_index_out[k + o] = g->h_edges[Ep + e];
      ++o;
    }
  *kept = k;
  return GCL_OK;
}
