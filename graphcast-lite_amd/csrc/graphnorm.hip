// PyG LayerNorm(mode="graph") forward + backward (reference construction
// src/models.py:102-104,368-374): statistics over ALL n*F elements of one sample's tensor,
//   y = (x - mean) / (std_biased + eps) * gamma + beta        (eps added to the std, not inside sqrt)
// per sample under batching.  Three launches per direction: per-(sample, chunk) partial sums in
// fp64 (the variance is formed as E[x^2] - mean^2, so the sums must not lose bits), a tiny
// fixed-order reduction of the partials to per-sample sums, then the apply pass.
//   stats[b] = (mean, 1/(std + eps))
// backward, with g = dy*gamma, d = std + eps, N = n*F:
//   dx = (g - mean(g)) / d - (x - mean) * sum(g (x - mean)) / (N std d^2)
//   dgamma[c] = sum dy * xhat,  dbeta[c] = sum dy
#include <algorithm>

#include "common.h"

namespace {

constexpr int kGnChunks = 2048;  // at most this many partial sums per sample (a sample of 84M elements still fills the chip)
static inline int gn_chunks(int64_t n, int F) {  // ~16K elements per chunk
  const int64_t c = (n * F + 16383) / 16384;
  return (int)(c < 1 ? 1 : c > kGnChunks ? kGnChunks : c);
}

// part[b][chunk][2] = (sum a, sum b) over the chunk's rows, with
//   MODE 0: a = x, b = x^2          MODE 1: a = dy*gamma, b = dy*gamma*(x - mean)
// VEC: rows are read as float4 (F % 4 == 0, 16-B aligned rows).
template <int MODE, bool VEC>
__global__ __launch_bounds__(256) void gn_partial_kernel(const float* __restrict__ X, int64_t ldx, int64_t bsx,
                                                         const float* __restrict__ dY, int64_t lddy, int64_t bsdy,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ stats, double* __restrict__ part,
                                                         int32_t n, int32_t F) {
  __shared__ double red[2][4];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int rows_per = (n + (int)gridDim.x - 1) / (int)gridDim.x;
  const int r0 = min(n, chunk * rows_per), r1 = min(n, r0 + rows_per);
  const float mean = (MODE == 1) ? stats[2 * b] : 0.f;
  double sa = 0.0, sb = 0.0;
  if (VEC) {
    const int F4 = F >> 2;
    const int64_t total = (int64_t)(r1 - r0) * F4;
    for (int64_t idx = threadIdx.x; idx < total; idx += 256) {
      const int r = r0 + (int)(idx / F4), c = (int)(idx % F4) * 4;
      const float4 x = *reinterpret_cast<const float4*>(X + (int64_t)b * bsx + (int64_t)r * ldx + c);
      if (MODE == 0) {
        sa += ((double)x.x + (double)x.y) + ((double)x.z + (double)x.w);
        sb += ((double)x.x * (double)x.x + (double)x.y * (double)x.y) + ((double)x.z * (double)x.z + (double)x.w * (double)x.w);
      } else {
        const float4 d = *reinterpret_cast<const float4*>(dY + (int64_t)b * bsdy + (int64_t)r * lddy + c);
        const float4 gm = *reinterpret_cast<const float4*>(gamma + c);
        const float g0 = d.x * gm.x, g1 = d.y * gm.y, g2 = d.z * gm.z, g3 = d.w * gm.w;
        sa += ((double)g0 + (double)g1) + ((double)g2 + (double)g3);
        sb += ((double)g0 * (double)(x.x - mean) + (double)g1 * (double)(x.y - mean)) +
              ((double)g2 * (double)(x.z - mean) + (double)g3 * (double)(x.w - mean));
      }
    }
  } else {
    const int64_t total = (int64_t)(r1 - r0) * F;
    for (int64_t idx = threadIdx.x; idx < total; idx += 256) {
      const int r = r0 + (int)(idx / F), c = (int)(idx % F);
      const float x = X[(int64_t)b * bsx + (int64_t)r * ldx + c];
      if (MODE == 0) {
        sa += (double)x;
        sb += (double)x * (double)x;
      } else {
        const float g = dY[(int64_t)b * bsdy + (int64_t)r * lddy + c] * gamma[c];
        sa += (double)g;
        sb += (double)g * (double)(x - mean);
      }
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    sa += __shfl_down(sa, off, 64);
    sb += __shfl_down(sb, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = sa;
    red[1][threadIdx.x >> 6] = sb;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double* p = part + ((int64_t)b * kGnChunks + chunk) * 2;
    p[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    p[1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

// fin[b][2] = the per-sample sums, reduced from the chunk partials in a fixed order
__global__ __launch_bounds__(256) void gn_finalize_kernel(const double* __restrict__ part, double* __restrict__ fin,
                                                          int32_t nchunks) {
  __shared__ double red[2][4];
  const int b = blockIdx.x;
  double s0 = 0.0, s1 = 0.0;
  for (int k = threadIdx.x; k < nchunks; k += 256) {
    s0 += part[((int64_t)b * kGnChunks + k) * 2];
    s1 += part[((int64_t)b * kGnChunks + k) * 2 + 1];
  }
  for (int off = 32; off > 0; off >>= 1) {
    s0 += __shfl_down(s0, off, 64);
    s1 += __shfl_down(s1, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = s0;
    red[1][threadIdx.x >> 6] = s1;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    fin[2 * b] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    fin[2 * b + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

__device__ __forceinline__ void gn_sum_parts(const double* fin, int b, double& s0, double& s1) {
  s0 = fin[2 * b];
  s1 = fin[2 * b + 1];
}

template <bool VEC>
__global__ __launch_bounds__(256) void gn_fwd_apply_kernel(const float* __restrict__ X, int64_t ldx, int64_t bsx,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps,
                                                           const double* __restrict__ part, float* __restrict__ Y,
                                                           int64_t ldy, int64_t bsy, float* __restrict__ stats,
                                                           int32_t n, int32_t F) {
  const int b = blockIdx.y;
  double s0, s1;
  gn_sum_parts(part, b, s0, s1);
  const double N = (double)n * (double)F;
  const double mean_d = s0 / N;
  double var = s1 / N - mean_d * mean_d;
  if (var < 0.0) var = 0.0;
  const float mean = (float)mean_d;
  const float rinv = (float)(1.0 / (sqrt(var) + (double)eps));
  if (blockIdx.x == 0 && threadIdx.x == 0 && stats) {
    stats[2 * b] = mean;
    stats[2 * b + 1] = rinv;
  }
  if (VEC) {  // 16-byte rows: one float4 of x, gamma, beta per step
    const int F4 = F >> 2;
    const int64_t total4 = (int64_t)n * F4;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total4; idx += (int64_t)gridDim.x * 256) {
      const int r = (int)(idx / F4), c = (int)(idx % F4) * 4;
      const float4 x = *reinterpret_cast<const float4*>(X + (int64_t)b * bsx + (int64_t)r * ldx + c);
      const float4 g = *reinterpret_cast<const float4*>(gamma + c);
      const float4 be = *reinterpret_cast<const float4*>(beta + c);
      float4 y;
      y.x = (x.x - mean) * rinv * g.x + be.x;
      y.y = (x.y - mean) * rinv * g.y + be.y;
      y.z = (x.z - mean) * rinv * g.z + be.z;
      y.w = (x.w - mean) * rinv * g.w + be.w;
      *reinterpret_cast<float4*>(Y + (int64_t)b * bsy + (int64_t)r * ldy + c) = y;
    }
    return;
  }
  const int64_t total = (int64_t)n * F;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int r = (int)(idx / F), c = (int)(idx % F);
    const float x = X[(int64_t)b * bsx + (int64_t)r * ldx + c];
    Y[(int64_t)b * bsy + (int64_t)r * ldy + c] = (x - mean) * rinv * gamma[c] + beta[c];
  }
}

// dx and per-block column partials of (dy*xhat | dy): cpart[block][2][F]
template <bool VEC>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float* __restrict__ dY, int64_t lddy, int64_t bsdy,
                                                           const float* __restrict__ X, int64_t ldx, int64_t bsx,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ stats, float eps,
                                                           const double* __restrict__ part, float* __restrict__ dX,
                                                           int64_t lddx, int64_t bsdx, float* __restrict__ cpart,
                                                           int32_t n, int32_t F, int32_t rows_per_block) {
  extern __shared__ float cs[];  // [256 / F][2][F]: one slot per thread, summed in a fixed order
  const int b = blockIdx.y;
  double s0, s1;
  gn_sum_parts(part, b, s0, s1);
  const double N = (double)n * (double)F;
  const float mean = stats[2 * b], rinv = stats[2 * b + 1];
  const double d = 1.0 / (double)rinv;       // std + eps
  const double sd = d - (double)eps;         // std
  const float gmean = (float)(s0 / N);
  const float k2 = sd > 0.0 ? (float)(s1 / (N * sd * d * d)) : 0.f;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(n, r0 + rows_per_block);
  if (VEC) {
    // thread t owns the float4 column t % (F/4) of rows r0 + t / (F/4), stepping 256 / (F/4) rows
    // (F/4 divides 256): four fixed columns per thread, 16-byte accesses
    const int F4 = F >> 2;
    const int c = (threadIdx.x % F4) * 4, rsub = threadIdx.x / F4, rstep = 256 / F4;
    const float4 gm = *reinterpret_cast<const float4*>(gamma + c);
    float4 pg = make_float4(0.f, 0.f, 0.f, 0.f), pb = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = r0 + rsub; r < r1; r += rstep) {
      const float4 x = *reinterpret_cast<const float4*>(X + (int64_t)b * bsx + (int64_t)r * ldx + c);
      const float4 dy = *reinterpret_cast<const float4*>(dY + (int64_t)b * bsdy + (int64_t)r * lddy + c);
      const float4 xc = make_float4(x.x - mean, x.y - mean, x.z - mean, x.w - mean);
      float4 o;
      o.x = (dy.x * gm.x - gmean) * rinv - xc.x * k2;
      o.y = (dy.y * gm.y - gmean) * rinv - xc.y * k2;
      o.z = (dy.z * gm.z - gmean) * rinv - xc.z * k2;
      o.w = (dy.w * gm.w - gmean) * rinv - xc.w * k2;
      *reinterpret_cast<float4*>(dX + (int64_t)b * bsdx + (int64_t)r * lddx + c) = o;
      pg.x += dy.x * xc.x * rinv; pg.y += dy.y * xc.y * rinv; pg.z += dy.z * xc.z * rinv; pg.w += dy.w * xc.w * rinv;
      pb.x += dy.x; pb.y += dy.y; pb.z += dy.z; pb.w += dy.w;
    }
    *reinterpret_cast<float4*>(cs + rsub * 2 * F + c) = pg;
    *reinterpret_cast<float4*>(cs + rsub * 2 * F + F + c) = pb;
    __syncthreads();
    float* out = cpart + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 * F;
    for (int q = threadIdx.x; q < 2 * F; q += 256) {
      float t = 0.f;
      for (int k = 0; k < rstep; ++k) t += cs[k * 2 * F + q];
      out[q] = t;
    }
    return;
  }
  // thread t owns column t % F of rows r0 + t / F, stepping 256 / F rows (F <= 256): its column is fixed
  const int c = threadIdx.x % F, rsub = threadIdx.x / F, rstep = 256 / F;
  float pg = 0.f, pb = 0.f;
  if (rsub < rstep)
    for (int r = r0 + rsub; r < r1; r += rstep) {
      const float x = X[(int64_t)b * bsx + (int64_t)r * ldx + c];
      const float dy = dY[(int64_t)b * bsdy + (int64_t)r * lddy + c];
      const float xc = x - mean;
      const float g = dy * gamma[c];
      dX[(int64_t)b * bsdx + (int64_t)r * lddx + c] = (g - gmean) * rinv - xc * k2;
      pg += dy * xc * rinv;
      pb += dy;
    }
  if (rsub < rstep) {
    cs[rsub * 2 * F + c] = pg;
    cs[rsub * 2 * F + F + c] = pb;
  }
  __syncthreads();
  float* out = cpart + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 * F;
  for (int q = threadIdx.x; q < 2 * F; q += 256) {
    float t = 0.f;
    for (int k = 0; k < rstep; ++k) t += cs[k * 2 * F + q];
    out[q] = t;
  }
}

}  // namespace

extern "C" size_t gcl_graphnorm_ws_bytes(int32_t B, int32_t n, int32_t F) {
  const size_t parts = (size_t)B * (kGnChunks + 1) * 2 * sizeof(double);  // chunk partials + per-sample sums
  const size_t nblk = (size_t)gcl::cdiv(n, 256);
  const size_t cparts = (size_t)B * nblk * 2 * (size_t)F * sizeof(float);
  return parts + cparts + 64;
}

extern "C" int gcl_graphnorm_fwd(const float* x, int64_t ldx, int64_t bsx, const float* gamma, const float* beta,
                                 float eps, float* y, int64_t ldy, int64_t bsy, float* stats, int32_t B, int32_t n,
                                 int32_t F, void* ws, size_t ws_bytes, gcl_stream_t stream) {
  GCL_CHECK_ARG(x && gamma && beta && y, "graphnorm_fwd: null argument");
  GCL_CHECK_ARG(B > 0 && n > 0 && F > 0 && ldx >= F && ldy >= F, "graphnorm_fwd: bad shape");
  GCL_CHECK_ARG(ws && ws_bytes >= gcl_graphnorm_ws_bytes(B, n, F) && gcl::aligned16(ws), "graphnorm_fwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  double* part = (double*)ws;
  double* fin = part + (size_t)B * kGnChunks * 2;
  const int nch = gn_chunks(n, F);
  const bool vec = (F % 4 == 0) && (ldx % 4 == 0) && (bsx % 4 == 0) && gcl::aligned16(x);
  if (vec)
    hipLaunchKernelGGL((gn_partial_kernel<0, true>), dim3(nch, B), dim3(256), 0, st, x, ldx, bsx, nullptr, 0, 0,
                       nullptr, nullptr, part, n, F);
  else
    hipLaunchKernelGGL((gn_partial_kernel<0, false>), dim3(nch, B), dim3(256), 0, st, x, ldx, bsx, nullptr, 0, 0,
                       nullptr, nullptr, part, n, F);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(256), 0, st, part, fin, nch);
  const bool vec_all = vec && (ldy % 4 == 0) && (bsy % 4 == 0) && gcl::aligned16(y) && gcl::aligned16(gamma) &&
                       gcl::aligned16(beta);
  if (vec_all) {
    const unsigned nb = (unsigned)std::min<int64_t>(gcl::cdiv((int64_t)n * (F / 4), 256), 4096);
    hipLaunchKernelGGL(gn_fwd_apply_kernel<true>, dim3(nb, B), dim3(256), 0, st, x, ldx, bsx, gamma, beta, eps, fin, y,
                       ldy, bsy, stats, n, F);
  } else {
    const unsigned nb = (unsigned)std::min<int64_t>(gcl::cdiv((int64_t)n * F, 256), 1024);
    hipLaunchKernelGGL(gn_fwd_apply_kernel<false>, dim3(nb, B), dim3(256), 0, st, x, ldx, bsx, gamma, beta, eps, fin, y,
                       ldy, bsy, stats, n, F);
  }
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

extern "C" int gcl_graphnorm_bwd(const float* dy, int64_t lddy, int64_t bsdy, const float* x, int64_t ldx,
                                 int64_t bsx, const float* gamma, const float* stats, float eps, float* dx,
                                 int64_t lddx, int64_t bsdx, float* dgamma, float* dbeta, int32_t accumulate,
                                 int32_t B, int32_t n, int32_t F, void* ws, size_t ws_bytes, gcl_stream_t stream) {
  GCL_CHECK_ARG(dy && x && gamma && stats && dx && dgamma && dbeta, "graphnorm_bwd: null argument");
  GCL_CHECK_ARG(B > 0 && n > 0 && F > 0 && F <= 256 && ldx >= F && lddy >= F && lddx >= F, "graphnorm_bwd: bad shape (F <= 256)");
  GCL_CHECK_ARG(ws && ws_bytes >= gcl_graphnorm_ws_bytes(B, n, F) && gcl::aligned16(ws), "graphnorm_bwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  double* part = (double*)ws;
  double* fin = part + (size_t)B * kGnChunks * 2;
  const int nch = gn_chunks(n, F);
  float* cpart = (float*)(fin + (size_t)B * 2);
  const bool vec = (F % 4 == 0) && (ldx % 4 == 0) && (bsx % 4 == 0) && (lddy % 4 == 0) && (bsdy % 4 == 0) &&
                   gcl::aligned16(x) && gcl::aligned16(dy) && gcl::aligned16(gamma);
  if (vec)
    hipLaunchKernelGGL((gn_partial_kernel<1, true>), dim3(nch, B), dim3(256), 0, st, x, ldx, bsx, dy, lddy, bsdy,
                       gamma, stats, part, n, F);
  else
    hipLaunchKernelGGL((gn_partial_kernel<1, false>), dim3(nch, B), dim3(256), 0, st, x, ldx, bsx, dy, lddy, bsdy,
                       gamma, stats, part, n, F);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(256), 0, st, part, fin, nch);
  const int rows_per_block = 256;
  const unsigned nblk = (unsigned)gcl::cdiv(n, rows_per_block);
  const int F4 = F / 4;
  const bool vec_apply = vec && F4 >= 1 && (256 % F4 == 0) && (lddx % 4 == 0) && (bsdx % 4 == 0) && gcl::aligned16(dx);
  if (vec_apply)
    hipLaunchKernelGGL(gn_bwd_apply_kernel<true>, dim3(nblk, B), dim3(256), 2 * (size_t)F * (256 / F4) * sizeof(float), st,
                       dy, lddy, bsdy, x, ldx, bsx, gamma, stats, eps, fin, dx, lddx, bsdx, cpart, n, F, rows_per_block);
  else
    hipLaunchKernelGGL(gn_bwd_apply_kernel<false>, dim3(nblk, B), dim3(256), 2 * (size_t)F * (256 / F) * sizeof(float), st,
                       dy, lddy, bsdy, x, ldx, bsx, gamma, stats, eps, fin, dx, lddx, bsdx, cpart, n, F, rows_per_block);
  GCL_CHECK_LAUNCH();
  const int nparts = (int)(nblk * B);
  return gcl::launch_reduce_parts2(cpart, nparts, 2 * F, F, F, dgamma, dbeta, F, accumulate, st);
}
