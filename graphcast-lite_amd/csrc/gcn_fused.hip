// One GCNConv layer forward in ONE kernel (src/models.py:419, PyG GCNConv: lin -> propagate -> bias).
//
//   y[b,i,:] = ( sum_{e in row i} w_e * act(x[b, col_e, :]) ) W^T + bias
//
// i.e. aggregate first, transform second - the same result as PyG's transform-then-aggregate up to
// fp32 rounding (the aggregation is linear), but without the intermediate h = act(x) W^T ever going
// to memory: per layer one gathered read of x and one write of y instead of read x / write h /
// gathered read h / write y.  A block owns 32 rows of one sample: each of its 4 waves gathers 8 rows
// (two batches of the aggregation kernel's ELL-prefix + CSR-tail scheme, 16 lanes x float4 per row)
// into the block's LDS tile, then waves 0 and 1 multiply the tile by one 32-column slab each of the
// weight panel the block staged (v_mfma_f32_32x32x2_f32, exact fp32), add the bias and store.
// For Fin, Fout <= 64, Fin % 4 == 0 and graphs without heavy rows (in-degree <= 64).
// STATUS: correct (tests/test_hip_ops.py::test_gcn_layer_fwd_one_kernel) but not yet faster than
// gcl_linear_fwd + gcl_aggregate on MI355X (4.46 vs 4.01 ms per baseline step; a 128-row-per-block
// version: 4.72), so the Python side only uses it when GCL_FUSED_GCN=1.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ int d_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

constexpr unsigned kOOB = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, int64_t nbytes) {
  const int64_t cap = 0x7FFFFF00;
  const int n = (int)(nbytes < 0 ? 0 : (nbytes > cap ? cap : nbytes));
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, n, 0x00020000);
}
__device__ __forceinline__ void buf_st1(__amdgpu_buffer_rsrc_t r, unsigned off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, off, 0, 0);
}

constexpr int kKP = 66;   // LDS row stride (floats): even, kKP/2 odd -> conflict-free 8-byte fragment reads
constexpr int kTR = 32;   // rows per block: 8 per wave (two gather batches), one shared 32-row MFMA tile

template <int EW>
__global__ __launch_bounds__(256) void gcn_fwd_fused_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ w,
    const int32_t* __restrict__ ecol, const float* __restrict__ ew, const float* __restrict__ X, int64_t ldx,
    int64_t bsx, int32_t akind, const float* __restrict__ slope_p, const float* __restrict__ W,
    const float* __restrict__ bias, float* __restrict__ Y, int64_t ldy, int64_t bsy, int32_t n, int32_t B,
    int32_t Fin, int32_t Fout, int32_t nRB, int32_t xcd_map) {
  __shared__ __align__(16) float Wl[64 * kKP];
  __shared__ __align__(16) float Xt[32 * kKP];
  constexpr int LPR = 16;
  const int bid = blockIdx.x;
  int b, rb;
  if (xcd_map) {  // all row blocks of a sample on one XCD: its x rows stay in that L2
    const int slot = bid >> 3;
    b = (bid & (gcl::kNumXCD - 1)) + gcl::kNumXCD * (slot / nRB);
    rb = slot % nRB;
  } else {
    b = bid / nRB;
    rb = bid % nRB;
  }
  if (b >= B) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int idx = tid; idx < 64 * 64; idx += 256) {  // weight panel, zero padded to 64 x 64
    const int j = idx >> 6, k = idx & 63;
    Wl[j * kKP + k] = (j < Fout && k < Fin) ? W[(int64_t)j * Fin + k] : 0.f;
  }
  const int sub = lane / LPR, l = lane % LPR, gbase = sub * LPR, c0 = l * 4;
  const bool cactive = c0 < Fin;
  const int cc = cactive ? c0 : 0;  // inactive channel lanes re-read channel 0 and contribute zeros
  const float slope = (akind == gcl::kActPrelu && slope_p) ? *slope_p : 1.f;
  const float* __restrict__ Xb = X + (int64_t)b * bsx;
  const int row0 = rb * kTR;          // first row of the block's tile
  const int wrow0 = row0 + wave * 8;  // first row this wave gathers
  float* Xw = Xt;

  auto ld4 = [&](int j, float& x0, float& x1, float& x2, float& x3) {
    const float4 v = *reinterpret_cast<const float4*>(Xb + (int64_t)j * ldx + cc);
    x0 = v.x; x1 = v.y; x2 = v.z; x3 = v.w;
    if (akind != gcl::kActNone) {  // block-uniform
      x0 = gcl::act_f(x0, slope, akind); x1 = gcl::act_f(x1, slope, akind);
      x2 = gcl::act_f(x2, slope, akind); x3 = gcl::act_f(x3, slope, akind);
    }
  };
  auto meta = [&](int row, int& start, int& end, int& cj, float& wj) {
    const int rc = row < n ? row : n - 1;
    start = rowptr[rc];
    end = rowptr[rc + 1];
    cj = ecol[(int64_t)rc * gcl::kEll + (l & (gcl::kEll - 1))];
    wj = ew[(int64_t)rc * gcl::kEll + (l & (gcl::kEll - 1))];
  };
  int start, end, cj;
  float wj;
  meta(wrow0 + sub, start, end, cj, wj);

#pragma unroll 1
  for (int it = 0; it < 2; ++it) {
    const int row = wrow0 + it * 4 + sub;
    int nstart = 0, nend = 0, ncj = 0;
    float nwj = 0.f;
    if (it + 1 < 2) meta(row + 4, nstart, nend, ncj, nwj);  // next batch's metadata, in flight during this one
    const int deg = end - start;
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
    if (__any(deg > EW)) {  // wave-uniform: CSR tail of the rows with more edges than the prefix
      for (int base = start + EW; base < end; base += LPR) {
        const int mine = base + l;
        int oj = 0;
        float ow = 0.f;
        if (mine < end) {
          oj = col[mine];
          ow = w[mine];
        }
        const int cnt = min(LPR, end - base);
        for (int k = 0; k < cnt; ++k) {
          const int j = __shfl(oj, gbase + k, 64);
          const float wk = __shfl(ow, gbase + k, 64);
          float p0, p1, p2, p3;
          ld4(j, p0, p1, p2, p3);
          a0 += wk * p0; a1 += wk * p1; a2 += wk * p2; a3 += wk * p3;
        }
      }
    }
    const bool live = (row < n) && cactive;
    float2* d = reinterpret_cast<float2*>(Xw + (wave * 8 + it * 4 + sub) * kKP + c0);
    d[0] = make_float2(live ? a0 : 0.f, live ? a1 : 0.f);
    d[1] = make_float2(live ? a2 : 0.f, live ? a3 : 0.f);
    start = nstart; end = nend; cj = ncj; wj = nwj;
  }
  __syncthreads();  // weight panel staged by all waves; each wave's own tile complete

  if (wave >= 2) return;  // waves 0 and 1 each multiply the 32-row tile by one 32-column slab of W
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const float* ap = Xw + (lane & 31) * kKP + 2 * (lane >> 5);
  const float* bp = Wl + (wave * 32 + (lane & 31)) * kKP + 2 * (lane >> 5);
  const int nq = (Fin + 3) >> 2;
  for (int q = 0; q < nq; ++q) {
    const float2 a = *reinterpret_cast<const float2*>(ap + 4 * q);
    const float2 b0 = *reinterpret_cast<const float2*>(bp + 4 * q);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc, 0, 0, 0);
  }

  // epilogue: lane owns output column j of this wave's slab and 16 of the tile's 32 rows
  const int nr = (n - row0) < 32 ? (n - row0) : 32;
  const int64_t wbytes = nr > 0 ? ((int64_t)(nr - 1) * ldy + Fout) * 4 : 0;
  const __amdgpu_buffer_rsrc_t ry = make_rsrc(Y + (int64_t)b * bsy + (int64_t)row0 * ldy, wbytes);
  const int j = wave * 32 + (lane & 31);
  const bool jok = j < Fout;
  const float bj = (bias && jok) ? bias[j] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int rr = d_row(r, lane);
    buf_st1(ry, jok ? (unsigned)((rr * ldy + j) * 4) : kOOB, acc[r] + bj);
  }
}

}  // namespace

extern "C" int gcl_gcn_layer_fwd(const gcl_graph_t* g, const float* x, int64_t ldx, int64_t bsx, int32_t act,
                                 const float* slope, const float* W, const float* bias, float* y, int64_t ldy,
                                 int64_t bsy, int32_t B, int32_t Fin, int32_t Fout, gcl_stream_t stream) {
  GCL_CHECK_ARG(g && x && W && y, "gcn_layer_fwd: null argument");
  GCL_CHECK_ARG(g->kind == GCL_GRAPH_GCN || g->kind == GCL_GRAPH_MEAN, "gcn_layer_fwd: graph carries no edge weights");
  GCL_CHECK_ARG(B > 0 && Fin >= 4 && Fin <= 64 && Fin % 4 == 0 && Fout >= 1 && Fout <= 64,
                "gcn_layer_fwd: unsupported Fin=%d Fout=%d (Fin %% 4 == 0, both <= 64)", Fin, Fout);
  GCL_CHECK_ARG(ldx >= Fin && ldy >= Fout && ldx % 4 == 0 && bsx % 4 == 0 && gcl::aligned16(x),
                "gcn_layer_fwd: x rows must be 16-B aligned");
  GCL_CHECK_ARG(g->n_heavy == 0, "gcn_layer_fwd: the graph has rows with more than %d in-edges", gcl::kHeavy);
  GCL_CHECK_ARG(act == GCL_ACT_NONE || act == GCL_ACT_SILU || (act == GCL_ACT_PRELU && slope),
                "gcn_layer_fwd: bad activation %d", act);
  GCL_CHECK_ARG(x != y, "gcn_layer_fwd: in-place is not supported");
  const int32_t n = g->n;
  const int32_t nRB = (int32_t)gcl::cdiv(n, kTR);
  const int xcd_map = B >= gcl::kNumXCD ? 1 : 0;
  const int64_t nb = xcd_map ? (int64_t)gcl::kNumXCD * gcl::cdiv(B, gcl::kNumXCD) * nRB : (int64_t)B * nRB;
  GCL_CHECK_ARG(nb < (int64_t)INT32_MAX, "gcn_layer_fwd: grid too large");
  int ewidth = g->ell_width >= 8 ? 8 : g->ell_width >= 4 ? 4 : g->ell_width >= 2 ? 2 : 1;
#define GCL_GF(EW_)                                                                                                  \
  hipLaunchKernelGGL((gcn_fwd_fused_kernel<EW_>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, g->rowptr,   \
                     g->col, g->w, g->ecol, g->ew, x, ldx, bsx, act, slope, W, bias, y, ldy, bsy, n, B, Fin, Fout, nRB, \
                     xcd_map)
  switch (ewidth) {
    case 8: GCL_GF(8); break;
    case 4: GCL_GF(4); break;
    case 2: GCL_GF(2); break;
    default: GCL_GF(1); break;
  }
#undef GCL_GF
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}
