// Edge-wise glue of the InteractionNet processor (src/models.py:206-236): everything of one
// message-passing step that is not a dense contraction.  All of it is HBM-bound row traffic over
// the edge state [B, E, D] (D = latent, 1 KiB rows at 256): rows are moved as float4 by
// D/4-lane groups, indices are read once per row, no atomics (segments come from a CSR, so every
// output row has exactly one writer and a fixed summation order).
//
//   gcl_segment_reduce : out[b,i,:] = (sum | mean) over k in [rowptr[i], rowptr[i+1]) of
//                        src[b, perm ? perm[k] : k, :]
//        forward : scatter(edge_update, receivers, reduce="mean")   (src/models.py:221)
//        backward: gradients of the x[senders] / x[receivers] gathers (src/models.py:216)
//   gcl_edge_combine   : out[b,e,:] = base[b,e,:] + extra[b,e,:] + A[b,ia[e],:] * sa[ia[e]] + C[b,ic[e],:]
//        forward : hidden = e W_e^T + (x W_s^T)[senders] + (x W_r^T)[receivers]  (the first edge-MLP
//                  layer on cat([x_s, x_r, e]), src/models.py:216-217, split by operand)
//        backward: d(edge_update) = d(new_edge) + d(aggregated)[receivers] / deg
//   gcl_act_fwd / gcl_act_bwd : elementwise SiLU / PReLU where an activation output must be
//        materialised (edge_encoder output, src/models.py:251-254)
#include "common.h"

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 fma4(float4 a, float s, float4 b) {
  return make_float4(fmaf(a.x, s, b.x), fmaf(a.y, s, b.y), fmaf(a.z, s, b.z), fmaf(a.w, s, b.w));
}

// LPR lanes per row (power of two <= 64); a row of D floats is ceil(D/4/LPR) float4 per lane.
template <int LPR>
__global__ __launch_bounds__(256) void segment_reduce_kernel(const float* __restrict__ src, int64_t lds_, int64_t bss,
                                                             const int32_t* __restrict__ perm,
                                                             const int32_t* __restrict__ rowptr, int32_t mean,
                                                             float* __restrict__ out, int64_t ldo, int64_t bso,
                                                             int32_t B, int32_t n, int32_t D4) {
  constexpr int GPB = 256 / LPR;  // row groups per block
  const int sub = threadIdx.x % LPR, grp = threadIdx.x / LPR;
  const int64_t total = (int64_t)B * n;
  for (int64_t row = (int64_t)blockIdx.x * GPB + grp; row < total; row += (int64_t)gridDim.x * GPB) {
    const int b = (int)(row / n), i = (int)(row - (int64_t)b * n);
    const int k0 = rowptr[i], k1 = rowptr[i + 1];
    const float* sb = src + (int64_t)b * bss;
    const float inv = (mean && k1 > k0) ? 1.f / (float)(k1 - k0) : 1.f;
    for (int c = sub; c < D4; c += LPR) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      int k = k0;
      for (; k + 4 <= k1; k += 4) {  // four independent rows in flight
        const int e0 = perm ? perm[k] : k, e1 = perm ? perm[k + 1] : k + 1;
        const int e2 = perm ? perm[k + 2] : k + 2, e3 = perm ? perm[k + 3] : k + 3;
        const float4 v0 = ld4(sb + (int64_t)e0 * lds_ + 4 * c), v1 = ld4(sb + (int64_t)e1 * lds_ + 4 * c);
        const float4 v2 = ld4(sb + (int64_t)e2 * lds_ + 4 * c), v3 = ld4(sb + (int64_t)e3 * lds_ + 4 * c);
        acc = add4(add4(add4(add4(acc, v0), v1), v2), v3);  // edge order
      }
      for (; k < k1; ++k) {
        const int e = perm ? perm[k] : k;
        acc = add4(acc, ld4(sb + (int64_t)e * lds_ + 4 * c));
      }
      acc.x *= inv; acc.y *= inv; acc.z *= inv; acc.w *= inv;
      st4(out + (int64_t)b * bso + (int64_t)i * ldo + 4 * c, acc);
    }
  }
}

template <int LPR>
__global__ __launch_bounds__(256) void edge_combine_kernel(const float* __restrict__ base, const float* __restrict__ extra,
                                                           const float* __restrict__ A, int64_t lda, int64_t bsa,
                                                           const int32_t* __restrict__ ia, const float* __restrict__ sa,
                                                           const float* __restrict__ Cc, int64_t ldc, int64_t bsc,
                                                           const int32_t* __restrict__ ic, float* __restrict__ out,
                                                           int32_t B, int64_t E, int32_t D4) {
  constexpr int GPB = 256 / LPR;
  const int sub = threadIdx.x % LPR, grp = threadIdx.x / LPR;
  const int64_t total = (int64_t)B * E;
  const int64_t D = (int64_t)D4 * 4;
  for (int64_t row = (int64_t)blockIdx.x * GPB + grp; row < total; row += (int64_t)gridDim.x * GPB) {
    const int64_t b = row / E, e = row - b * E;
    const int ja = A ? ia[e] : 0, jc = Cc ? ic[e] : 0;
    const float s = (A && sa) ? sa[ja] : 1.f;
    for (int c = sub; c < D4; c += LPR) {
      float4 v = base ? ld4(base + row * D + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (extra) v = add4(v, ld4(extra + row * D + 4 * c));
      if (A) v = fma4(ld4(A + b * bsa + (int64_t)ja * lda + 4 * c), s, v);
      if (Cc) v = add4(v, ld4(Cc + b * bsc + (int64_t)jc * ldc + 4 * c));
      st4(out + row * D + 4 * c, v);
    }
  }
}

__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n4,
                                                      int32_t kind, const float* __restrict__ slope) {
  const float a = slope ? *slope : 1.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 v = ld4(x + 4 * i);
    v.x = gcl::act_f(v.x, a, kind); v.y = gcl::act_f(v.y, a, kind);
    v.z = gcl::act_f(v.z, a, kind); v.w = gcl::act_f(v.w, a, kind);
    st4(y + 4 * i, v);
  }
}

// dx = dy * act'(x); PReLU slope partial (fp64) per block
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                      float* __restrict__ dx, int64_t n4, int32_t kind,
                                                      const float* __restrict__ slope, double* __restrict__ part) {
  __shared__ double red[4];
  const float a = slope ? *slope : 1.f;
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 z = ld4(x + 4 * i);
    float4 g = ld4(dy + 4 * i);
    if (kind == gcl::kActSilu) {
      g.x *= gcl::dsilu_f(z.x); g.y *= gcl::dsilu_f(z.y); g.z *= gcl::dsilu_f(z.z); g.w *= gcl::dsilu_f(z.w);
    } else {
      acc += (z.x <= 0.f ? (double)(g.x * z.x) : 0.0) + (z.y <= 0.f ? (double)(g.y * z.y) : 0.0) +
             (z.z <= 0.f ? (double)(g.z * z.z) : 0.0) + (z.w <= 0.f ? (double)(g.w * z.w) : 0.0);
      g.x = z.x <= 0.f ? g.x * a : g.x; g.y = z.y <= 0.f ? g.y * a : g.y;
      g.z = z.z <= 0.f ? g.z * a : g.z; g.w = z.w <= 0.f ? g.w * a : g.w;
    }
    st4(dx + 4 * i, g);
  }
  if (part) {
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

__global__ void act_slope_reduce_kernel(const double* __restrict__ part, int32_t n, float* __restrict__ d_slope) {
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) acc += part[i];
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (threadIdx.x == 0) *d_slope += (float)acc;
}

int lanes_per_row(int D4) { return D4 >= 64 ? 64 : D4 > 16 ? 32 : D4 > 8 ? 16 : 8; }
unsigned rows_grid(int64_t rows, int gpb) {
  const int64_t want = gcl::cdiv(rows, gpb);
  const int64_t cap = (int64_t)gcl::kNumCU * 32;
  return (unsigned)(want < 1 ? 1 : want < cap ? want : cap);
}

}  // namespace

extern "C" int gcl_segment_reduce(const float* src, int64_t ld_src, int64_t bs_src, const int32_t* perm,
                                  const int32_t* rowptr, int32_t mean, float* out, int64_t ld_out, int64_t bs_out,
                                  int32_t B, int32_t n, int32_t D, gcl_stream_t stream) {
  GCL_CHECK_ARG(src && rowptr && out, "segment_reduce: null argument");
  GCL_CHECK_ARG(B > 0 && n > 0 && D > 0 && D % 4 == 0, "segment_reduce: D=%d must be a positive multiple of 4", D);
  GCL_CHECK_ARG(ld_src % 4 == 0 && ld_out % 4 == 0 && bs_src % 4 == 0 && bs_out % 4 == 0 && ld_src >= D && ld_out >= D &&
                    gcl::aligned16(src) && gcl::aligned16(out),
                "segment_reduce: rows must be 16-B aligned");
  const int D4 = D / 4, lpr = lanes_per_row(D4);
  const unsigned grid = rows_grid((int64_t)B * n, 256 / lpr);
#define GCL_SR(L_)                                                                                                 \
  hipLaunchKernelGGL(segment_reduce_kernel<L_>, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, ld_src, bs_src, \
                     perm, rowptr, mean, out, ld_out, bs_out, B, n, D4)
  switch (lpr) {
    case 64: GCL_SR(64); break;
    case 32: GCL_SR(32); break;
    case 16: GCL_SR(16); break;
    default: GCL_SR(8); break;
  }
#undef GCL_SR
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

extern "C" int gcl_edge_combine(const float* base, const float* extra, const float* A, int64_t lda, int64_t bsa,
                                const int32_t* ia, const float* sa, const float* Cc, int64_t ldc, int64_t bsc,
                                const int32_t* ic, float* out, int32_t B, int64_t E, int32_t D, gcl_stream_t stream) {
  GCL_CHECK_ARG(out && (base || extra || A || Cc), "edge_combine: nothing to combine");
  GCL_CHECK_ARG((!A || ia) && (!Cc || ic), "edge_combine: gathered operand without its index array");
  GCL_CHECK_ARG(B > 0 && E > 0 && D > 0 && D % 4 == 0, "edge_combine: D=%d must be a positive multiple of 4", D);
  GCL_CHECK_ARG((!A || (lda % 4 == 0 && bsa % 4 == 0 && lda >= D && gcl::aligned16(A))) &&
                    (!Cc || (ldc % 4 == 0 && bsc % 4 == 0 && ldc >= D && gcl::aligned16(Cc))) && gcl::aligned16(out) &&
                    (!base || gcl::aligned16(base)) && (!extra || gcl::aligned16(extra)),
                "edge_combine: rows must be 16-B aligned");
  const int D4 = D / 4, lpr = lanes_per_row(D4);
  const unsigned grid = rows_grid((int64_t)B * E, 256 / lpr);
#define GCL_EC(L_)                                                                                                    \
  hipLaunchKernelGGL(edge_combine_kernel<L_>, dim3(grid), dim3(256), 0, (hipStream_t)stream, base, extra, A, lda, bsa, \
                     ia, sa, Cc, ldc, bsc, ic, out, B, E, D4)
  switch (lpr) {
    case 64: GCL_EC(64); break;
    case 32: GCL_EC(32); break;
    case 16: GCL_EC(16); break;
    default: GCL_EC(8); break;
  }
#undef GCL_EC
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

extern "C" int gcl_act_fwd(const float* x, float* y, int64_t count, int32_t act, const float* slope,
                           gcl_stream_t stream) {
  GCL_CHECK_ARG(x && y && count >= 0 && count % 4 == 0 && gcl::aligned16(x) && gcl::aligned16(y),
                "act_fwd: count must be a multiple of 4 and the buffers 16-B aligned");
  GCL_CHECK_ARG(act == GCL_ACT_SILU || (act == GCL_ACT_PRELU && slope), "act_fwd: unsupported activation %d", act);
  if (count == 0) return GCL_OK;
  const int64_t n4 = count / 4;
  const int64_t want = gcl::cdiv(n4, 256);
  hipLaunchKernelGGL(act_fwd_kernel, dim3((unsigned)(want < 8192 ? want : 8192)), dim3(256), 0, (hipStream_t)stream, x, y,
                     n4, act, slope);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

extern "C" size_t gcl_act_bwd_ws_bytes(void) { return 2048 * sizeof(double); }

extern "C" int gcl_act_bwd(const float* x, const float* dy, float* dx, int64_t count, int32_t act, const float* slope,
                           float* d_slope, void* ws, size_t ws_bytes, gcl_stream_t stream) {
  GCL_CHECK_ARG(x && dy && dx && count >= 0 && count % 4 == 0 && gcl::aligned16(x) && gcl::aligned16(dy) && gcl::aligned16(dx),
                "act_bwd: count must be a multiple of 4 and the buffers 16-B aligned");
  GCL_CHECK_ARG(act == GCL_ACT_SILU || (act == GCL_ACT_PRELU && slope), "act_bwd: unsupported activation %d", act);
  if (count == 0) return GCL_OK;
  const bool want_slope = act == GCL_ACT_PRELU && d_slope;
  GCL_CHECK_ARG(!want_slope || (ws && ws_bytes >= gcl_act_bwd_ws_bytes()), "act_bwd: workspace too small");
  const int64_t n4 = count / 4;
  const int64_t want = gcl::cdiv(n4, 256);
  const unsigned grid = (unsigned)(want < 2048 ? want : 2048);
  hipLaunchKernelGGL(act_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, dy, dx, n4, act, slope,
                     want_slope ? (double*)ws : nullptr);
  GCL_CHECK_LAUNCH();
  if (want_slope) {
    hipLaunchKernelGGL(act_slope_reduce_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const double*)ws, (int)grid,
                       d_slope);
    GCL_CHECK_LAUNCH();
  }
  return GCL_OK;
}
