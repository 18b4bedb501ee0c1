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
#include "common.h"

namespace {

template <int LPR, bool VL, bool VS>
__global__ __launch_bounds__(256) void agg_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                  const float* __restrict__ w, const float* __restrict__ H,
                                                  int64_t ldh, int64_t bsh, const float* __restrict__ bias,
                                                  float* __restrict__ Y, int64_t ldy, int64_t bsy, int32_t n,
                                                  int32_t B, int32_t F, int32_t nRB, int32_t xcd_map) {
  constexpr int RPW = 64 / LPR;
  constexpr int RPB = RPW * 4;
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
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int sub = lane / LPR;
  const int l = lane % LPR;
  const int gbase = sub * LPR;  // first lane of this row's group
  const int row = rb * RPB + wave * RPW + sub;
  const bool ractive = row < n;
  const int c0 = l * 4;
  const bool cactive = c0 < F;
  const float* __restrict__ Hb = H + (int64_t)b * bsh;

  int start = 0, end = 0;
  if (ractive) {
    start = rowptr[row];
    end = rowptr[row + 1];
  }
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;

  auto ld4 = [&](int j, float& x0, float& x1, float& x2, float& x3) {
    const float* p = Hb + (int64_t)j * ldh + c0;
    if (VL) {
      const float4 v = *reinterpret_cast<const float4*>(p);
      x0 = v.x; x1 = v.y; x2 = v.z; x3 = v.w;
    } else {
      x0 = p[0];
      x1 = (c0 + 1 < F) ? p[1] : 0.f;
      x2 = (c0 + 2 < F) ? p[2] : 0.f;
      x3 = (c0 + 3 < F) ? p[3] : 0.f;
    }
  };

  for (int base = start; base < end; base += LPR) {
    const int mine = base + l;
    int cj = 0;
    float wj = 0.f;
    if (mine < end) {
      cj = col[mine];
      wj = w[mine];
    }
    const int cnt = min(LPR, end - base);
    for (int k = 0; k < cnt; k += 4) {
      // indices/weights of up to 4 neighbours, broadcast from the lanes of this row's group
      const int j0 = __shfl(cj, gbase + k, 64);
      const int j1 = __shfl(cj, gbase + ((k + 1) & (LPR - 1)), 64);
      const int j2 = __shfl(cj, gbase + ((k + 2) & (LPR - 1)), 64);
      const int j3 = __shfl(cj, gbase + ((k + 3) & (LPR - 1)), 64);
      const float w0 = __shfl(wj, gbase + k, 64);
      const float w1 = __shfl(wj, gbase + ((k + 1) & (LPR - 1)), 64);
      const float w2 = __shfl(wj, gbase + ((k + 2) & (LPR - 1)), 64);
      const float w3 = __shfl(wj, gbase + ((k + 3) & (LPR - 1)), 64);
      float p0 = 0, p1 = 0, p2 = 0, p3 = 0, q0 = 0, q1 = 0, q2 = 0, q3 = 0;
      float r0 = 0, r1 = 0, r2 = 0, r3 = 0, s0 = 0, s1 = 0, s2 = 0, s3 = 0;
      if (cactive) {
        ld4(j0, p0, p1, p2, p3);
        if (k + 1 < cnt) ld4(j1, q0, q1, q2, q3);
        if (k + 2 < cnt) ld4(j2, r0, r1, r2, r3);
        if (k + 3 < cnt) ld4(j3, s0, s1, s2, s3);
      }
      a0 += w0 * p0; a1 += w0 * p1; a2 += w0 * p2; a3 += w0 * p3;
      if (k + 1 < cnt) { a0 += w1 * q0; a1 += w1 * q1; a2 += w1 * q2; a3 += w1 * q3; }
      if (k + 2 < cnt) { a0 += w2 * r0; a1 += w2 * r1; a2 += w2 * r2; a3 += w2 * r3; }
      if (k + 3 < cnt) { a0 += w3 * s0; a1 += w3 * s1; a2 += w3 * s2; a3 += w3 * s3; }
    }
  }

  if (!ractive || !cactive) return;
  if (bias) {
    if (VS || c0 + 3 < F) {
      a0 += bias[c0]; a1 += bias[c0 + 1]; a2 += bias[c0 + 2]; a3 += bias[c0 + 3];
    } else {
      a0 += bias[c0];
      if (c0 + 1 < F) a1 += bias[c0 + 1];
      if (c0 + 2 < F) a2 += bias[c0 + 2];
    }
  }
  float* __restrict__ yp = Y + (int64_t)b * bsy + (int64_t)row * ldy + c0;
  if (VS) {
    *reinterpret_cast<float4*>(yp) = make_float4(a0, a1, a2, a3);
  } else {
    yp[0] = a0;
    if (c0 + 1 < F) yp[1] = a1;
    if (c0 + 2 < F) yp[2] = a2;
    if (c0 + 3 < F) yp[3] = a3;
  }
}

template <int LPR>
int launch_agg(const int32_t* rowptr, const int32_t* col, const float* w, const float* h, int64_t ldh, int64_t bsh,
               const float* bias, float* y, int64_t ldy, int64_t bsy, int32_t n, int32_t B, int32_t F,
               hipStream_t st) {
  constexpr int RPB = (64 / LPR) * 4;
  const int32_t nRB = (int32_t)gcl::cdiv(n, RPB);
  const int xcd_map = B >= gcl::kNumXCD ? 1 : 0;
  const int64_t nb = xcd_map ? (int64_t)gcl::kNumXCD * gcl::cdiv(B, gcl::kNumXCD) * nRB : (int64_t)B * nRB;
  GCL_CHECK_ARG(nb < (int64_t)INT32_MAX, "aggregate: grid too large");
  // vector loads need 16-B aligned rows and a padded tail (ldh >= roundup(F,4))
  const bool vl = (ldh % 4 == 0) && (bsh % 4 == 0) && gcl::aligned16(h) && ldh >= ((F + 3) / 4) * 4;
  const bool vs = (ldy % 4 == 0) && (bsy % 4 == 0) && gcl::aligned16(y) && (F % 4 == 0);
  dim3 grid((unsigned)nb), block(256);
#define GCL_AGG(VL_, VS_)                                                                                     \
  hipLaunchKernelGGL((agg_kernel<LPR, VL_, VS_>), grid, block, 0, st, rowptr, col, w, h, ldh, bsh, bias, y, ldy, \
                     bsy, n, B, F, nRB, xcd_map)
  if (vl && vs) GCL_AGG(true, true);
  else if (vl) GCL_AGG(true, false);
  else if (vs) GCL_AGG(false, true);
  else GCL_AGG(false, false);
#undef GCL_AGG
  GCL_CHECK_LAUNCH();
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
  const int32_t* rp = transpose ? g->trowptr : g->rowptr;
  const int32_t* cl = transpose ? g->tcol : g->col;
  const float* ww = transpose ? g->tw : g->w;
  hipStream_t st = (hipStream_t)stream;
  const int lanes = (F + 3) / 4;
  if (lanes <= 4) return launch_agg<4>(rp, cl, ww, h, ldh, bsh, bias, y, ldy, bsy, g->n, B, F, st);
  if (lanes <= 8) return launch_agg<8>(rp, cl, ww, h, ldh, bsh, bias, y, ldy, bsy, g->n, B, F, st);
  if (lanes <= 16) return launch_agg<16>(rp, cl, ww, h, ldh, bsh, bias, y, ldy, bsy, g->n, B, F, st);
  if (lanes <= 32) return launch_agg<32>(rp, cl, ww, h, ldh, bsh, bias, y, ldy, bsy, g->n, B, F, st);
  return launch_agg<64>(rp, cl, ww, h, ldh, bsh, bias, y, ldy, bsy, g->n, B, F, st);
}
