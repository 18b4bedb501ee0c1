// Input assembly, loss (+gradient), Adam and strided row copies: the elementwise ends of the path.
#include "common.h"

namespace {

// (b, i, c) of a flat index over [B][N][C] that a thread walks with a fixed stride, kept incrementally.  The GPU has
// no integer divide: the naive `idx % C, (idx / C) % N, idx / (C * N)` costs two 64-bit divisions (~40 instructions
// each) per ELEMENT and made these streaming kernels instruction-bound (2.2-2.7 TB/s); here they are paid once per thread.
struct Walk3 {
  int c, i, C, N, sc, si;
  int64_t b, sb;
  __device__ __forceinline__ Walk3(int64_t idx, int64_t stride, int N_, int C_) : C(C_), N(N_) {
    c = (int)(idx % C_);
    const int64_t r = idx / C_;
    i = (int)(r % N_);
    b = r / N_;
    sc = (int)(stride % C_);
    const int64_t t = stride / C_;
    si = (int)(t % N_);
    sb = t / N_;
  }
  __device__ __forceinline__ void next() {
    c += sc;
    i += si;
    b += sb;
    if (c >= C) { c -= C; ++i; }
    if (i >= N) { i -= N; ++b; }
  }
};

// out[b, i, :] = i < G ? [x[b,i,:Cdyn] | gstat[i,:Cs]] : [0 | mstat[i-G,:Cs]]
// (src/models.py:776-806; the reference allocates the zero block and concatenates 3x per forward)
__global__ __launch_bounds__(256) void assemble_kernel(const float* __restrict__ x, const float* __restrict__ gs,
                                                       const float* __restrict__ ms, float* __restrict__ out,
                                                       int64_t ldo, int32_t B, int32_t G, int32_t M, int32_t Cdyn,
                                                       int32_t Cs, const float* __restrict__ tail, int32_t r) {
  const int C = Cdyn + Cs;
  const int64_t total = (int64_t)B * (G + M) * C;
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (Walk3 w(idx, stride, G + M, C); idx < total; idx += stride, w.next()) {
    const int c = w.c, i = w.i;
    const int64_t b = w.b;
    float v;
    if (i < G)
      v = c < Cdyn ? x[(b * G + i) * Cdyn + c] : gs[(int64_t)i * Cs + (c - Cdyn)];
    else if (i >= G + M - r)  // per-sample tail rows, given whole
      v = tail[(b * r + (i - (G + M - r))) * C + c];
    else
      v = c < Cdyn ? 0.f : ms[(int64_t)(i - G) * Cs + (c - Cdyn)];
    out[(b * (G + M) + i) * ldo + c] = v;
  }
}

// the same with one 16-byte store per thread (C % 4 == 0, 16-byte aligned output rows): the element kernel above is
// bound by its 4-byte stores (2.8 TB/s)
__global__ __launch_bounds__(256) void assemble4_kernel(const float* __restrict__ x, const float* __restrict__ gs,
                                                        const float* __restrict__ ms, float* __restrict__ out,
                                                        int64_t ldo, int32_t B, int32_t G, int32_t M, int32_t Cdyn,
                                                        int32_t Cs, const float* __restrict__ tail, int32_t r) {
  const int C4 = (Cdyn + Cs) >> 2;
  const int64_t total = (int64_t)B * (G + M) * C4;
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (Walk3 w(idx, stride, G + M, C4); idx < total; idx += stride, w.next()) {
    const int c0 = w.c * 4, i = w.i;
    const int64_t b = w.b;
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = c0 + q;
      if (i < G)
        v[q] = c < Cdyn ? x[(b * G + i) * Cdyn + c] : gs[(int64_t)i * Cs + (c - Cdyn)];
      else if (i >= G + M - r)
        v[q] = tail[(b * r + (i - (G + M - r))) * (Cdyn + Cs) + c];
      else
        v[q] = c < Cdyn ? 0.f : ms[(int64_t)(i - G) * Cs + (c - Cdyn)];
    }
    *reinterpret_cast<float4*>(out + (b * (G + M) + i) * ldo + c0) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// loss partials + gradient (src/train.py:203-213,85-102)
__global__ __launch_bounds__(256) void wmse_kernel(const float* __restrict__ delta, int64_t ldd, int64_t bsd,
                                                   const float* __restrict__ xl, int64_t ldx, int64_t bsx,
                                                   const float* __restrict__ y, int64_t ldy, int64_t bsy,
                                                   const float* __restrict__ node_w, const float* __restrict__ chan_w,
                                                   float inv_wsum, float grad_scale, float* __restrict__ dd,
                                                   float* __restrict__ out_state, float* __restrict__ part,
                                                   int32_t B, int32_t G, int32_t C) {
  __shared__ float red[4];
  const int64_t total = (int64_t)B * G * C;
  float acc = 0.f;
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (Walk3 wk(idx, stride, G, C); idx < total; idx += stride, wk.next()) {
    const int c = wk.c, g = wk.i;
    const int64_t b = wk.b;
    float o = delta[b * bsd + (int64_t)g * ldd + c];
    if (xl) o += xl[b * bsx + (int64_t)g * ldx + c];
    const float t = y[b * bsy + (int64_t)g * ldy + c];
    float w = 1.f;
    if (chan_w) w *= chan_w[c];
    if (node_w) w *= node_w[g];
    const float d = o - t;
    acc += (d * d) * w;
    if (dd) dd[idx] = 2.f * w * d * inv_wsum * grad_scale;
    if (out_state) out_state[idx] = o;
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void wmse_final_kernel(const float* __restrict__ part, int32_t nparts, float inv_wsum,
                                  const float* __restrict__ loss_prev, float* __restrict__ loss) {
  __shared__ double red[4];
  double s = 0.0;
  for (int p = threadIdx.x; p < nparts; p += 256) s += (double)part[p];
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float l = (float)((red[0] + red[1] + red[2] + red[3]) * (double)inv_wsum);
    *loss = loss_prev ? *loss_prev + l : l;  // running sum over the autoregressive steps of one batch
  }
}

// torch.optim.Adam (no amsgrad, no maximize): src/main.py:212
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t count,
                                                   float lr, float b1, float b2, float eps, float wd, float bc1,
                                                   float bc2_sqrt, float gscale) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
    float gi = g[i] * gscale;
    if (wd != 0.f) gi += wd * p[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] -= (lr / bc1) * (mi / denom);
  }
}

// Device-resident step counter: lets a captured hipGraph replay the optimiser without any
// step-dependent host scalar baked into it.  state[0] = step (as float bits of an int), bc = {bc1, sqrt(bc2)}.
__global__ void adam_tick_kernel(int32_t* __restrict__ step, float* __restrict__ bc, float b1, float b2) {
  const int t = *step + 1;
  *step = t;
  bc[0] = (float)(1.0 - pow((double)b1, (double)t));
  bc[1] = (float)sqrt(1.0 - pow((double)b2, (double)t));
}

__global__ __launch_bounds__(256) void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ m, float* __restrict__ v, int64_t count,
                                                       float lr, float b1, float b2, float eps, float wd,
                                                       const float* __restrict__ bc, float gscale) {
  const float bc1 = bc[0], bc2_sqrt = bc[1];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
    float gi = g[i] * gscale;
    if (wd != 0.f) gi += wd * p[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] -= (lr / bc1) * (mi / denom);
  }
}

__global__ __launch_bounds__(256) void copy_rows_kernel(const float* __restrict__ src, int64_t lds, int64_t bss,
                                                        float* __restrict__ dst, int64_t ldd, int64_t bsd, int32_t B,
                                                        int32_t rows, int32_t F, int32_t vec) {
  if (vec) {
    const int nv = F >> 2;
    const int64_t total = (int64_t)B * rows * nv;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
      const int c = (int)(idx % nv) * 4;
      const int64_t br = idx / nv;
      const int r = (int)(br % rows);
      const int64_t b = br / rows;
      *reinterpret_cast<float4*>(dst + b * bsd + (int64_t)r * ldd + c) =
          *reinterpret_cast<const float4*>(src + b * bss + (int64_t)r * lds + c);
    }
  } else {
    const int64_t total = (int64_t)B * rows * F;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
      const int c = (int)(idx % F);
      const int64_t br = idx / F;
      const int r = (int)(br % rows);
      const int64_t b = br / rows;
      dst[b * bsd + (int64_t)r * ldd + c] = src[b * bss + (int64_t)r * lds + c];
    }
  }
}

// dst[b,i,:] = a[b, map_a[i], :] if map_a[i] >= 0, else b_[b, map_b[i], :] if map_b[i] >= 0, else 0.
// A source with batch stride 0 is broadcast over the batch.  sum_batch: dst[0,i,:] = sum_b a[b, map_a[i], :];
// sum_batch = R > 1 deals the nd summed rows R to a destination sample: row i goes to dst[i / R, i % R, :].
// Row glue of the compact pipeline (stage split / concat of src/models.py:837-838,860-862 restricted
// to the rows that matter); 16-B accesses, one row per F/4 lanes.
__global__ __launch_bounds__(256) void gather2_kernel(const float* __restrict__ a, int64_t lda, int64_t bsa,
                                                      const int32_t* __restrict__ map_a,
                                                      const float* __restrict__ b_, int64_t ldb, int64_t bsb,
                                                      const int32_t* __restrict__ map_b, float* __restrict__ dst,
                                                      int64_t ldd, int64_t bsd, int32_t B, int32_t nd, int32_t F4,
                                                      int32_t sum_batch) {
  const int64_t total = (int64_t)(sum_batch ? 1 : B) * nd * F4;
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (Walk3 w(idx, stride, nd, F4); idx < total; idx += stride, w.next()) {
    const int c = w.c * 4, i = w.i;
    const int64_t b = w.b;
    const int ja = map_a ? map_a[i] : i;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (sum_batch) {
      if (ja >= 0)
        for (int bb = 0; bb < B; ++bb) {
          const float4 t = *reinterpret_cast<const float4*>(a + bb * bsa + (int64_t)ja * lda + c);
          v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
    } else if (ja >= 0) {
      v = *reinterpret_cast<const float4*>(a + b * bsa + (int64_t)ja * lda + c);
    } else if (b_ && map_b) {
      const int jb = map_b[i];
      if (jb >= 0) v = *reinterpret_cast<const float4*>(b_ + b * bsb + (int64_t)jb * ldb + c);
    }
    if (sum_batch > 1) {
      const int bq = i / sum_batch;
      *reinterpret_cast<float4*>(dst + (int64_t)bq * bsd + (int64_t)(i - bq * sum_batch) * ldd + c) = v;
    } else {
      *reinterpret_cast<float4*>(dst + b * bsd + (int64_t)i * ldd + c) = v;
    }
  }
}

// One autoregressive advance (scripts/predict.py:512-535, src/train.py:203-228): residual add,
// static / forcing channel overwrite, output append and window shift in ONE pass.
//   step_out = residual ? state[..., obs-1, :] + delta : delta
//   step_out[c] = state[..., obs-1, c] for static channels, y_step[c] for forcing channels
//   out[b, g, out_off + c] = step_out;  new_state[..., k, :] = state[..., k+1, :], last slot = step_out
// chan_kind[c]: 0 = predicted, 1 = static (carry forward), 2 = forcing (from y_step when given).
__global__ __launch_bounds__(256) void ar_advance_kernel(const float* __restrict__ state,
                                                         const float* __restrict__ delta, int64_t ldd, int64_t bsd,
                                                         const float* __restrict__ y_step, int64_t ldy, int64_t bsy,
                                                         const int32_t* __restrict__ chan_kind,
                                                         float* __restrict__ new_state, float* __restrict__ out,
                                                         int64_t ldo, int64_t bso, int32_t out_off, int32_t B,
                                                         int32_t G, int32_t obs, int32_t C, int32_t residual) {
  const int64_t total = (int64_t)B * G * obs * C;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int c = (int)(idx % C);
    const int k = (int)((idx / C) % obs);
    const int64_t bg = idx / ((int64_t)C * obs);
    if (k < obs - 1) {
      new_state[idx] = state[idx + C];  // shift the window by one step
    } else {
      const int g = (int)(bg % G);
      const int64_t b = bg / G;
      const float xl = state[idx];
      float v = delta[b * bsd + (int64_t)g * ldd + c];
      if (residual) v += xl;
      const int kind = chan_kind ? chan_kind[c] : 0;
      if (kind == 1) v = xl;
      else if (kind == 2 && y_step) v = y_step[b * bsy + (int64_t)g * ldy + c];
      new_state[idx] = v;
      if (out) out[b * bso + (int64_t)g * ldo + out_off + c] = v;
    }
  }
}

// Input windows from a device-resident fp16 time series (src/data/dataloader_chunked.py:179-223):
//   X[b, g, o*C + c] = (float(series[t0[b] + o,        lon, lat, c]) - mean[c]) / std[c]
//   Y[b, g, p*C + c] = (float(series[t0[b] + obs + p,  lon, lat, c]) - mean[c]) / std[c]
// with g = lat * n_lon + lon (lat-major, lon fastest - the node order of the graphs) and only the
// first C of the Ct stored channels used.  fp16 -> fp32 is exact and the subtraction / IEEE
// division are the reference's numpy operations, so the result is bit-identical to the loader's.
// One thread per output element: writes are fully coalesced, reads are C-channel runs.
__global__ __launch_bounds__(256) void window_pack_kernel(const _Float16* __restrict__ series, int64_t T, int32_t n_lon,
                                                          int32_t n_lat, int32_t Ct, const int64_t* __restrict__ t0,
                                                          const float* __restrict__ mean, const float* __restrict__ stdv,
                                                          int32_t C, int32_t first, int32_t frames,
                                                          float* __restrict__ out, int32_t B) {
  const int64_t G = (int64_t)n_lon * n_lat;
  const int64_t row = (int64_t)frames * C;
  const int64_t total = (int64_t)B * G * row;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int k = (int)(idx % row);
    const int64_t bg = idx / row;
    const int64_t g = bg % G;
    const int b = (int)(bg / G);
    const int f = k / C, c = k - f * C;
    const int lat = (int)(g / n_lon), lon = (int)(g - (int64_t)lat * n_lon);
    const int64_t t = t0[b] + first + f;
    float v = __builtin_nanf("");  // a window that leaves the series is an error made visible
    if (t >= 0 && t < T) {
      const float x = (float)series[(((int64_t)t * n_lon + lon) * n_lat + lat) * Ct + c];
      v = (x - mean[c]) / stdv[c];
    }
    out[idx] = v;
  }
}

// Backward of one autoregressive training step (loss of the step + window advance; src/train.py:203-228).
//   forward:  pred = residual ? x_last + delta : delta;  loss += sum(w (pred - y)^2) * inv;   dd = d loss / d pred
//             new_state = shift(state), last slot = pred (static channels: x_last, forcing channels: y)
//   backward: d_delta[c]     = g_loss * dd[c] + m_delta[c] * g_new[last, c]
//             d_state[t, c]  = (t >= 1 ? g_new[t-1, c] : 0)
//                              + (t == obs-1 ? (residual ? g_loss * dd[c] : 0) + m_state[c] * g_new[last, c] : 0)
//   with, per channel kind (0 predicted, 1 static, 2 forcing-from-y):  m_delta = {1, 0, 0},  m_state = {residual, 1, 0};
//   a forcing channel WITHOUT y behaves as a predicted one.  g_loss is a device scalar, g_new may be NULL (last step).
__global__ __launch_bounds__(256) void ar_step_bwd_kernel(const float* __restrict__ dd, const float* __restrict__ g_loss,
                                                          const float* __restrict__ g_new,
                                                          const int32_t* __restrict__ chan_kind, int32_t has_y,
                                                          int32_t residual, float* __restrict__ d_delta,
                                                          float* __restrict__ d_state, int32_t B, int32_t G, int32_t obs,
                                                          int32_t C) {
  const float gl = g_loss ? *g_loss : 1.f;
  const int64_t total = (int64_t)B * G * obs * C;
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (Walk3 w(idx, stride, obs, C); idx < total; idx += stride, w.next()) {
    const int c = w.c, k = w.i;
    const int64_t bg = w.b;
    float ds = (g_new && k >= 1) ? g_new[idx - C] : 0.f;
    if (k == obs - 1) {
      int kind = chan_kind ? chan_kind[c] : 0;
      if (kind == 2 && !has_y) kind = 0;
      const float gd = gl * dd[bg * C + c];
      const float gn = g_new ? g_new[idx] : 0.f;
      d_delta[bg * C + c] = gd + (kind == 0 ? gn : 0.f);
      ds += (residual ? gd : 0.f) + ((kind == 1 || (kind == 0 && residual)) ? gn : 0.f);
    }
    if (d_state) d_state[idx] = ds;
  }
}

// dst[b, i, c] = (i < rows_src && c < F_src) ? src[b, i, c] : 0   for i < rows_dst, c < F_dst: widens a gradient to the
// zero-padded layout of a layer whose output was returned as a row / column slice (the decoder returns the grid rows
// and the first 33 / 19 columns of its last, 36- / 20-wide conv: src/models.py:870-872).
__global__ __launch_bounds__(256) void pad_rows_kernel(const float* __restrict__ src, int64_t lds, int64_t bss,
                                                       int32_t rows_src, int32_t F_src, float* __restrict__ dst,
                                                       int64_t ldd, int64_t bsd, int32_t rows_dst, int32_t F_dst,
                                                       int32_t B) {
  const int64_t total = (int64_t)B * rows_dst * F_dst;
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (Walk3 w(idx, stride, rows_dst, F_dst); idx < total; idx += stride, w.next()) {
    const int c = w.c, r = w.i;
    const int64_t b = w.b;
    const bool in = r < rows_src && c < F_src;
    dst[b * bsd + (int64_t)r * ldd + c] = in ? src[b * bss + (int64_t)r * lds + c] : 0.f;
  }
}

inline unsigned grid_for(int64_t total, int cap = 4096) {
  int64_t nb = gcl::cdiv(total > 0 ? total : 1, 256);
  return (unsigned)(nb > cap ? cap : nb);
}
constexpr int kLossBlocks = 1024;

}  // namespace

extern "C" int gcl_assemble_input(const float* x, const float* gs, const float* ms, float* out, int64_t ldo,
                                  int32_t B, int32_t G, int32_t M, int32_t Cdyn, int32_t Cs, gcl_stream_t stream) {
  return gcl_assemble_input_tail(x, gs, ms, nullptr, 0, out, ldo, B, G, M, Cdyn, Cs, stream);
}

extern "C" int gcl_assemble_input_tail(const float* x, const float* gs, const float* ms, const float* tail, int32_t r,
                                       float* out, int64_t ldo, int32_t B, int32_t G, int32_t M, int32_t Cdyn, int32_t Cs,
                                       gcl_stream_t stream) {
  GCL_CHECK_ARG(x && gs && (ms || r == M) && out, "assemble_input: null argument");
  GCL_CHECK_ARG(B > 0 && G > 0 && M >= 0 && Cdyn >= 0 && Cs >= 0 && ldo >= Cdyn + Cs, "assemble_input: bad shape");
  GCL_CHECK_ARG(r >= 0 && r <= M && (r == 0 || tail), "assemble_input: tail rows r=%d outside [0, M=%d] or no tail", r, M);
  const int64_t total = (int64_t)B * (G + M) * (Cdyn + Cs);
  if (((Cdyn + Cs) & 3) == 0 && (ldo & 3) == 0 && gcl::aligned16(out))
    hipLaunchKernelGGL(assemble4_kernel, dim3(grid_for(total / 4, 8192)), dim3(256), 0, (hipStream_t)stream, x, gs, ms, out,
                       ldo, B, G, M, Cdyn, Cs, tail, r);
  else
    hipLaunchKernelGGL(assemble_kernel, dim3(grid_for(total, 8192)), dim3(256), 0, (hipStream_t)stream, x, gs, ms, out,
                     ldo, B, G, M, Cdyn, Cs, tail, r);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

extern "C" size_t gcl_wmse_ws_bytes(int32_t B, int32_t G, int32_t C) {
  (void)B; (void)G; (void)C;
  return (size_t)kLossBlocks * sizeof(float);
}

extern "C" int gcl_wmse_fwd_bwd(const float* delta, int64_t ldd, int64_t bsd, const float* x_last, int64_t ldx,
                                int64_t bsx, const float* y, int64_t ldy, int64_t bsy, const float* node_w,
                                const float* chan_w, float inv_wsum, float grad_scale, float* d_delta,
                                float* out_state, const float* loss_prev, float* loss_out, int32_t B, int32_t G,
                                int32_t C, void* ws, size_t ws_bytes, gcl_stream_t stream) {
  GCL_CHECK_ARG(delta && y && loss_out, "wmse: null argument");
  GCL_CHECK_ARG(B > 0 && G > 0 && C > 0 && ldd >= C && ldy >= C && (!x_last || ldx >= C), "wmse: bad shape");
  GCL_CHECK_ARG(ws && ws_bytes >= gcl_wmse_ws_bytes(B, G, C), "wmse: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = (int64_t)B * G * C;
  const unsigned nb = grid_for(total, kLossBlocks);
  hipLaunchKernelGGL(wmse_kernel, dim3(nb), dim3(256), 0, st, delta, ldd, bsd, x_last, ldx, bsx, y, ldy, bsy, node_w,
                     chan_w, inv_wsum, grad_scale, d_delta, out_state, (float*)ws, B, G, C);
  hipLaunchKernelGGL(wmse_final_kernel, dim3(1), dim3(256), 0, st, (const float*)ws, (int)nb, inv_wsum, loss_prev,
                     loss_out);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

extern "C" int gcl_adam_step(float* p, const float* g, float* m, float* v, int64_t count, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int32_t step, float grad_scale,
                             gcl_stream_t stream) {
  GCL_CHECK_ARG(p && g && m && v, "adam: null argument");
  GCL_CHECK_ARG(count >= 0 && step >= 1, "adam: bad count/step");
  if (count == 0) return GCL_OK;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(count, 1024)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, count,
                     lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), grad_scale);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

extern "C" int gcl_copy_rows(const float* src, int64_t lds, int64_t bss, float* dst, int64_t ldd, int64_t bsd,
                             int32_t B, int32_t rows, int32_t F, gcl_stream_t stream) {
  GCL_CHECK_ARG(src && dst, "copy_rows: null argument");
  GCL_CHECK_ARG(B > 0 && rows >= 0 && F > 0 && lds >= F && ldd >= F, "copy_rows: bad shape");
  if (rows == 0) return GCL_OK;
  const int vec = (F % 4 == 0) && (lds % 4 == 0) && (ldd % 4 == 0) && (bss % 4 == 0) && (bsd % 4 == 0) &&
                  gcl::aligned16(src) && gcl::aligned16(dst);
  const int64_t total = (int64_t)B * rows * (vec ? F / 4 : F);
  hipLaunchKernelGGL(copy_rows_kernel, dim3(grid_for(total, 8192)), dim3(256), 0, (hipStream_t)stream, src, lds, bss,
                     dst, ldd, bsd, B, rows, F, vec);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

extern "C" int gcl_gather2_rows(const float* a, int64_t lda, int64_t bsa, const int32_t* map_a, const float* b,
                                int64_t ldb, int64_t bsb, const int32_t* map_b, float* dst, int64_t ldd, int64_t bsd,
                                int32_t B, int32_t nd, int32_t F, int32_t sum_batch, gcl_stream_t stream) {
  GCL_CHECK_ARG(a && dst, "gather2_rows: null argument");
  GCL_CHECK_ARG(B > 0 && nd >= 0 && F > 0 && F % 4 == 0, "gather2_rows: F must be a positive multiple of 4 (F=%d)", F);
  GCL_CHECK_ARG(lda % 4 == 0 && bsa % 4 == 0 && ldd % 4 == 0 && bsd % 4 == 0 && gcl::aligned16(a) && gcl::aligned16(dst) &&
                    (!b || (ldb % 4 == 0 && bsb % 4 == 0 && gcl::aligned16(b))),
                "gather2_rows: rows must be 16-B aligned");
  if (nd == 0) return GCL_OK;
  const int64_t total = (int64_t)(sum_batch ? 1 : B) * nd * (F / 4);
  hipLaunchKernelGGL(gather2_kernel, dim3(grid_for(total, 8192)), dim3(256), 0, (hipStream_t)stream, a, lda, bsa, map_a,
                     b, ldb, bsb, map_b, dst, ldd, bsd, B, nd, F / 4, sum_batch);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

extern "C" int gcl_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t count, float lr, float beta1,
                                 float beta2, float eps, float weight_decay, int32_t* step_dev, float* bc_dev,
                                 float grad_scale, gcl_stream_t stream) {
  GCL_CHECK_ARG(p && g && m && v && step_dev && bc_dev, "adam_dev: null argument");
  GCL_CHECK_ARG(count >= 0, "adam_dev: bad count");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, st, step_dev, bc_dev, beta1, beta2);
  if (count > 0)
    hipLaunchKernelGGL(adam_dev_kernel, dim3(grid_for(count, 1024)), dim3(256), 0, st, p, g, m, v, count, lr, beta1,
                       beta2, eps, weight_decay, bc_dev, grad_scale);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

extern "C" int gcl_ar_step_bwd(const float* dd, const float* g_loss, const float* g_new, const int32_t* chan_kind,
                               int32_t has_y, int32_t residual, float* d_delta, float* d_state, int32_t B, int32_t G,
                               int32_t obs, int32_t C, gcl_stream_t stream) {
  GCL_CHECK_ARG(dd && d_delta, "ar_step_bwd: null argument");
  GCL_CHECK_ARG(B > 0 && G > 0 && obs >= 1 && C > 0, "ar_step_bwd: bad shape");
  const int64_t total = (int64_t)B * G * obs * C;
  hipLaunchKernelGGL(ar_step_bwd_kernel, dim3(grid_for(total, 8192)), dim3(256), 0, (hipStream_t)stream, dd, g_loss,
                     g_new, chan_kind, has_y, residual, d_delta, d_state, B, G, obs, C);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

extern "C" int gcl_pad_rows(const float* src, int64_t lds, int64_t bss, int32_t rows_src, int32_t F_src, float* dst,
                            int64_t ldd, int64_t bsd, int32_t rows_dst, int32_t F_dst, int32_t B, gcl_stream_t stream) {
  GCL_CHECK_ARG(src && dst, "pad_rows: null argument");
  GCL_CHECK_ARG(B > 0 && rows_src >= 0 && rows_dst >= rows_src && F_src >= 0 && F_dst >= F_src && lds >= F_src && ldd >= F_dst,
                "pad_rows: bad shape");
  const int64_t total = (int64_t)B * rows_dst * F_dst;
  if (total > 0)
    hipLaunchKernelGGL(pad_rows_kernel, dim3(grid_for(total, 8192)), dim3(256), 0, (hipStream_t)stream, src, lds, bss,
                       rows_src, F_src, dst, ldd, bsd, rows_dst, F_dst, B);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

extern "C" int gcl_zero(void* ptr, size_t nbytes, gcl_stream_t stream) {
  GCL_CHECK_ARG(ptr || nbytes == 0, "zero: null argument");
  if (nbytes) GCL_CHECK_HIP(hipMemsetAsync(ptr, 0, nbytes, (hipStream_t)stream));
  return GCL_OK;
}

extern "C" int gcl_ar_advance(const float* state, const float* delta, int64_t ldd, int64_t bsd, const float* y_step,
                              int64_t ldy, int64_t bsy, const int32_t* chan_kind, float* new_state, float* out,
                              int64_t ldo, int64_t bso, int32_t out_off, int32_t B, int32_t G, int32_t obs, int32_t C,
                              int32_t residual, gcl_stream_t stream) {
  GCL_CHECK_ARG(state && delta && new_state, "ar_advance: null argument");
  GCL_CHECK_ARG(ldd >= C, "ar_advance: delta row stride smaller than C");
  GCL_CHECK_ARG(state != new_state, "ar_advance: the window shift cannot be done in place");
  GCL_CHECK_ARG(B > 0 && G > 0 && obs >= 1 && C > 0, "ar_advance: bad shape");
  GCL_CHECK_ARG(!out || (ldo >= out_off + C), "ar_advance: output row too short for out_off + C");
  const int64_t total = (int64_t)B * G * obs * C;
  hipLaunchKernelGGL(ar_advance_kernel, dim3(grid_for(total, 8192)), dim3(256), 0, (hipStream_t)stream, state, delta,
                     ldd, bsd, y_step, ldy, bsy, chan_kind, new_state, out, ldo, bso, out_off, B, G, obs, C, residual);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}

extern "C" int gcl_window_pack(const uint16_t* series, int64_t T, int32_t n_lon, int32_t n_lat, int32_t Ct,
                               const int64_t* t0, const float* mean, const float* stdv, int32_t C, int32_t obs,
                               int32_t pred, float* X, float* Y, int32_t B, gcl_stream_t stream) {
  GCL_CHECK_ARG(series && t0 && mean && stdv && X, "window_pack: null argument");
  GCL_CHECK_ARG(T > 0 && n_lon > 0 && n_lat > 0 && Ct > 0 && C > 0 && C <= Ct && obs > 0 && pred >= 0 && B > 0,
                "window_pack: bad shape (T=%lld grid %dx%d C=%d of %d obs=%d pred=%d)", (long long)T, n_lon, n_lat, C, Ct,
                obs, pred);
  GCL_CHECK_ARG(pred == 0 || Y, "window_pack: pred > 0 needs the Y buffer");
  const int64_t G = (int64_t)n_lon * n_lat;
  hipLaunchKernelGGL(window_pack_kernel, dim3(grid_for((int64_t)B * G * obs * C, 16384)), dim3(256), 0,
                     (hipStream_t)stream, (const _Float16*)series, T, n_lon, n_lat, Ct, t0, mean, stdv, C, 0, obs, X, B);
  if (pred > 0)
    hipLaunchKernelGGL(window_pack_kernel, dim3(grid_for((int64_t)B * G * pred * C, 16384)), dim3(256), 0,
                       (hipStream_t)stream, (const _Float16*)series, T, n_lon, n_lat, Ct, t0, mean, stdv, C, obs, pred, Y, B);
  GCL_CHECK_LAUNCH();
  return GCL_OK;
}
