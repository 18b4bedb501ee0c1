// Wide dense transforms:  Y = act(X) Wl^T (+ bias) (+ addend)   for K, N beyond what the
// resident-panel kernel of linear.hip can hold in LDS (a 256x256 fp32 panel alone is 256 KiB).
//
// This is the dense work of the InteractionNet processor (src/models.py:166-236: edge / node MLPs
// at latent 256) and of the 256-wide encoder / decoder MLPs of the v2 configs.  Classic tiled
// contraction on the exact-fp32 matrix instruction v_mfma_f32_32x32x2_f32:
//   block = 4 waves, 128 x 128 output tile, wave = 64 x 64 (2 x 2 MFMA tiles, 64 accumulators);
//   K is walked in chunks of 32 through a double-buffered LDS stage ([128][34] + [128][34] floats
//   per buffer = 68 KiB for both: two blocks per CU, two waves per SIMD);
//   the global loads of chunk c+1 are issued before the 64 MFMAs of chunk c and committed to the
//   other buffer after them: one barrier per chunk; blocks are persistent and the first chunk of
//   the NEXT tile is fetched under the last chunk of the current one, so only the first tile of a
//   block pays the load latency;
//   fragments are 8-byte LDS reads (two k-steps each) at row stride 34 (stride/2 odd: conflict-free).
// Tiles are dealt XCD-aware: the column tiles of one row tile get consecutive slots of the SAME
// XCD, so the second read of an X tile hits that XCD's L2.
// Included by linear.hip (same translation unit: shares its helpers and launch plumbing).
#pragma once

constexpr int kGtTN = 128, kGtKC = 32, kGtKP = kGtKC + 2;
// MI = 32-row MFMA tiles per wave: the block tile is (64*MI) x 128.  MI = 1 (64-row tiles) is used when
// 128-row tiles would leave the persistent grid a short, badly balanced queue (a few hundred tiles).
constexpr size_t gt_lds(int MI) { return (size_t)2 * (64 * MI + kGtTN) * kGtKP * sizeof(float); }

// EPI_BIAS: akind/in_slope describe the activation applied to X on load.
// EPI_DX  : X is dY (no activation), Wl = W^T (TRANS), akind/in_slope describe the activation whose
//           derivative at Z multiplies the result; slope_part[block] gets the PReLU slope partial.
template <int EPI, bool TRANS, int MI>
__global__ __launch_bounds__(256, (MI == 1 ? 3 : 2)) void gemm_tile_kernel(
    const float* __restrict__ X, int64_t ldx, int32_t akind, const float* __restrict__ in_slope,
    const float* __restrict__ W, int64_t ldw, const float* __restrict__ bias, float* __restrict__ Y, int64_t ldy,
    int64_t rows, int32_t K, int32_t N, const float* __restrict__ Z, int64_t ldz, const float* __restrict__ add,
    int64_t ldadd, double* __restrict__ slope_part, int32_t nt, int64_t total, int32_t per_xcd) {
  constexpr int TM = 64 * MI, TN = kGtTN, KC = kGtKC, KP = kGtKP;
  constexpr int NX = 2 * MI;  // float4 loads per thread for the X chunk
  extern __shared__ __align__(16) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // persistent over the tiles of this block's XCD: slot, slot + nslots, ...
  const int xcd = blockIdx.x & 7, nslots = gridDim.x >> 3;
  const int64_t Lbase = (int64_t)xcd * per_xcd;
  const int64_t Lend = (Lbase + per_xcd) < total ? (Lbase + per_xcd) : total;
  int64_t L = Lbase + (blockIdx.x >> 3);
  double slope_acc = 0.0;
  if (L >= Lend) {  // block-uniform
    if (EPI == EPI_DX && slope_part && tid == 0) slope_part[blockIdx.x] = 0.0;
    return;
  }
  const float slope = in_slope ? *in_slope : 1.f;
  const bool xact = EPI == EPI_BIAS && akind != gcl::kActNone;
  const bool silu = akind == gcl::kActSilu;
  const bool has_z = EPI == EPI_DX && Z != nullptr;
  const bool has_add = add != nullptr;

  float4 pre[NX + 4];  // [0..NX) X, [NX..NX+4) W of the next chunk
  const float4* zero = gcl_zero4;
  auto issue = [&](int64_t r0, int n0, int k0) {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int idx = tid + 256 * i;
      const int row = idx >> 3, c4 = idx & 7;
      const bool ok = (r0 + row < rows) && (k0 + 4 * c4 < K);
      const float4* p = ok ? reinterpret_cast<const float4*>(X + (r0 + row) * ldx + k0 + 4 * c4) : zero;
      pre[i] = *p;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 256 * i;
      if (TRANS) {
        const int k = idx >> 5, j4 = idx & 31;
        const bool ok = (k0 + k < K) && (n0 + 4 * j4 < N);
        const float4* p = ok ? reinterpret_cast<const float4*>(W + (int64_t)(k0 + k) * ldw + n0 + 4 * j4) : zero;
        pre[NX + i] = *p;
      } else {
        const int j = idx >> 3, c4 = idx & 7;
        const bool ok = (n0 + j < N) && (k0 + 4 * c4 < K);
        const float4* p = ok ? reinterpret_cast<const float4*>(W + (int64_t)(n0 + j) * ldw + k0 + 4 * c4) : zero;
        pre[NX + i] = *p;
      }
    }
  };
  auto commit = [&](float* buf) {
    float* Xl = buf;
    float* Wl = buf + TM * KP;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int idx = tid + 256 * i;
      const int row = idx >> 3, c4 = idx & 7;
      float4 v = pre[i];
      if (xact) {
        v.x = gcl::act_f(v.x, slope, akind); v.y = gcl::act_f(v.y, slope, akind);
        v.z = gcl::act_f(v.z, slope, akind); v.w = gcl::act_f(v.w, slope, akind);
      }
      float2* d = reinterpret_cast<float2*>(Xl + row * KP + 4 * c4);
      d[0] = make_float2(v.x, v.y);
      d[1] = make_float2(v.z, v.w);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 256 * i;
      const float4 v = pre[NX + i];
      if (TRANS) {
        const int k = idx >> 5, j4 = idx & 31;
        float* d = Wl + (4 * j4) * KP + k;
        d[0] = v.x; d[KP] = v.y; d[2 * KP] = v.z; d[3 * KP] = v.w;
      } else {
        const int j = idx >> 3, c4 = idx & 7;
        float2* d = reinterpret_cast<float2*>(Wl + j * KP + 4 * c4);
        d[0] = make_float2(v.x, v.y);
        d[1] = make_float2(v.z, v.w);
      }
    }
  };

  const int nchunks = (K + KC - 1) / KC;
  int64_t r0 = (L / nt) * TM;
  int n0 = (int)(L % nt) * TN;
  int sel = 0;
  issue(r0, n0, 0);
  commit(smem);
  __syncthreads();
  for (;;) {
    const int64_t Ln = L + nslots;
    const bool more = Ln < Lend;  // block-uniform
    const int64_t nr0 = more ? (Ln / nt) * TM : 0;
    const int nn0 = more ? (int)(Ln % nt) * TN : 0;

    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int c = 0; c < nchunks; ++c) {
      float* buf = smem + (size_t)sel * (TM + TN) * KP;
      const bool lastc = c + 1 == nchunks;
      // the next chunk - of this tile or the first of the next tile - is in flight under the MFMAs
      if (!lastc) issue(r0, n0, (c + 1) * KC);
      else if (more) issue(nr0, nn0, 0);
      const float* ap = buf + (wm * 32 * MI + (lane & 31)) * KP + 2 * (lane >> 5);
      const float* bp = buf + TM * KP + (wn * 64 + (lane & 31)) * KP + 2 * (lane >> 5);
      // fragments of k-pair q+1 are read before the MFMAs of pair q issue (explicit double buffer)
      float2 a_c[MI], b_c[2];
#pragma unroll
      for (int i = 0; i < MI; ++i) a_c[i] = *reinterpret_cast<const float2*>(ap + i * 32 * KP);
      b_c[0] = *reinterpret_cast<const float2*>(bp);
      b_c[1] = *reinterpret_cast<const float2*>(bp + 32 * KP);
#pragma unroll
      for (int q = 0; q < KC / 4; ++q) {
        float2 a_n[MI], b_n[2];
        const int qn = (q + 1 < KC / 4) ? q + 1 : q;  // the last iteration re-reads its own pair (unused)
#pragma unroll
        for (int i = 0; i < MI; ++i) a_n[i] = *reinterpret_cast<const float2*>(ap + i * 32 * KP + 4 * qn);
        b_n[0] = *reinterpret_cast<const float2*>(bp + 4 * qn);
        b_n[1] = *reinterpret_cast<const float2*>(bp + 32 * KP + 4 * qn);
        __builtin_amdgcn_sched_barrier(0);  // keep these reads ABOVE the MFMAs they are meant to hide under
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c[i].x, b_c[0].x, acc[i][0], 0, 0, 0);
          acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c[i].x, b_c[1].x, acc[i][1], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c[i].y, b_c[0].y, acc[i][0], 0, 0, 0);
          acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c[i].y, b_c[1].y, acc[i][1], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) a_c[i] = a_n[i];
        b_c[0] = b_n[0];
        b_c[1] = b_n[1];
      }
      if (!lastc || more) commit(smem + (size_t)(sel ^ 1) * (TM + TN) * KP);
      __syncthreads();
      sel ^= 1;
    }

    // epilogue: lane owns column j and 16 rows of each of its 4 MFMA tiles; branch-free accesses.
    // The store / load offsets are the same for every tile: an opaque copy of the lane id keeps
    // the compiler from hoisting ~200 registers of them out of the persistent loop.
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int64_t nr = (rows - r0) < TM ? (rows - r0) : TM;
    const int ncols = (N - n0) < TN ? (N - n0) : TN;
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(Y + r0 * ldy + n0, win_bytes(nr, ldy, ncols));
    const __amdgpu_buffer_rsrc_t rz = make_rsrc(has_z ? Z + r0 * ldz + n0 : Y, has_z ? win_bytes(nr, ldz, ncols) : 0);
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(has_add ? add + r0 * ldadd + n0 : Y, has_add ? win_bytes(nr, ldadd, ncols) : 0);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = wn * 64 + j * 32 + (lane_e & 31);
        const bool jok = col < ncols;
        const float bj = (EPI == EPI_BIAS && bias && jok) ? bias[n0 + col] : 0.f;
        float zv[16], av[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rr = wm * 32 * MI + i * 32 + d_row(r, lane_e);
          zv[r] = buf_ld1(rz, jok ? (unsigned)((rr * ldz + col) * 4) : kOOB);    // 0 when absent
          av[r] = buf_ld1(ra, jok ? (unsigned)((rr * ldadd + col) * 4) : kOOB);  // 0 when absent
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rr = wm * 32 * MI + i * 32 + d_row(r, lane_e);
          float v = acc[i][j][r];
          if (EPI == EPI_DX) {
            if (silu) {  // block-uniform
              v *= has_z ? gcl::dsilu_f(zv[r]) : 1.f;
            } else {
              const bool neg = has_z && (zv[r] <= 0.f);
              slope_acc += neg ? (double)(v * zv[r]) : 0.0;
              v = neg ? v * slope : v;
            }
          } else {
            v += bj;
          }
          v += av[r];
          buf_st1(ry, jok ? (unsigned)((rr * ldy + col) * 4) : kOOB, v);
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the four slabs' Z / addend loads from piling up in registers
      }
    }
    if (!more) break;
    L = Ln;
    r0 = nr0;
    n0 = nn0;
  }
  if (EPI == EPI_DX && slope_part) {
    for (int off = 32; off > 0; off >>= 1) slope_acc += __shfl_down(slope_acc, off, 64);
    double* dred = reinterpret_cast<double*>(smem);  // all fragment reads ended at the last barrier
    if (lane == 0) dred[wave] = slope_acc;
    __syncthreads();
    if (tid == 0) slope_part[blockIdx.x] = (dred[0] + dred[1]) + (dred[2] + dred[3]);
  }
}

// ---------------------------------------------------------------------------------------------
// The same contract on the bf16 matrix pipe with exact 3-way operand splitting (x3.h): 128 x 128 tiles, K walked in
// chunks of 16 (one bf16 k-step).  A thread splits the 2 + 2 float4 of X and W it loaded into hi / mid / lo images
// ([256 rows][16 bf16] with 48-byte rows: conflict-free ds_read_b128 fragments), double-buffered (72 KiB: two blocks
// per CU); a wave (64 x 64 = 2 x 2 MFMA tiles) reads 6 + 6 fragments per chunk and issues 24 MFMAs of 32 cycles
// where the fp32-operand kernel issues 32 of 64.  The accumulator persists over the chunks, so the hi x hi products
// and the five correction products are kept in two accumulators (see x3.h) and added in the epilogue.
// EPI / activation / addend / Z semantics are those of gemm_tile_kernel; non-transposed weights only.
// ---------------------------------------------------------------------------------------------
constexpr int kX3KC = 16, kX3RowB = 48;
constexpr size_t gtx3_lds() { return (size_t)2 * 3 * 256 * kX3RowB; }

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_tile_x3_kernel(
    const float* __restrict__ X, int64_t ldx, int32_t akind, const float* __restrict__ in_slope,
    const float* __restrict__ W, int64_t ldw, const float* __restrict__ bias, float* __restrict__ Y, int64_t ldy,
    int64_t rows, int32_t K, int32_t N, const float* __restrict__ Z, int64_t ldz, const float* __restrict__ add,
    int64_t ldadd, double* __restrict__ slope_part, int32_t nt, int64_t total, int32_t per_xcd) {
  using namespace gcl::x3;
  constexpr int TM = 128, TN = 128, KC = kX3KC, RB = kX3RowB;
  constexpr int IMG = 256 * RB;  // one piece of one buffer: X rows 0..127, W rows 128..255
  extern __shared__ __align__(16) float smem[];
  unsigned char* sm8 = reinterpret_cast<unsigned char*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int xcd = blockIdx.x & 7, nslots = gridDim.x >> 3;
  const int64_t Lbase = (int64_t)xcd * per_xcd;
  const int64_t Lend = (Lbase + per_xcd) < total ? (Lbase + per_xcd) : total;
  int64_t L = Lbase + (blockIdx.x >> 3);
  double slope_acc = 0.0;
  if (L >= Lend) {  // block-uniform
    if (EPI == EPI_DX && slope_part && tid == 0) slope_part[blockIdx.x] = 0.0;
    return;
  }
  const float slope = in_slope ? *in_slope : 1.f;
  const bool xact = EPI == EPI_BIAS && akind != gcl::kActNone;
  const bool silu = akind == gcl::kActSilu;
  const bool has_z = EPI == EPI_DX && Z != nullptr;
  const bool has_add = add != nullptr;

  // staging map: a row chunk is 16 floats = 4 float4; 128 rows x 4 = 512 float4 per operand, 2 per thread
  float4 pre[4];  // [0..2) X, [2..4) W of the next chunk
  const float4* zero = gcl_zero4;
  auto issue = [&](int64_t r0, int n0, int k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + 256 * i;
      const int row = idx >> 2, c4 = idx & 3;
      const bool okx = (r0 + row < rows) && (k0 + 4 * c4 < K);
      pre[i] = *(okx ? reinterpret_cast<const float4*>(X + (r0 + row) * ldx + k0 + 4 * c4) : zero);
      const bool okw = (n0 + row < N) && (k0 + 4 * c4 < K);
      pre[2 + i] = *(okw ? reinterpret_cast<const float4*>(W + (int64_t)(n0 + row) * ldw + k0 + 4 * c4) : zero);
    }
  };
  auto commit = [&](unsigned char* buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + 256 * i;
      const int row = idx >> 2, c4 = idx & 3;
      float4 v = pre[i];
      if (xact) {
        v.x = gcl::act_f(v.x, slope, akind); v.y = gcl::act_f(v.y, slope, akind);
        v.z = gcl::act_f(v.z, slope, akind); v.w = gcl::act_f(v.w, slope, akind);
      }
      const float4 w4 = pre[2 + i];
      const Pk3 x01 = split2(v.x, v.y), x23 = split2(v.z, v.w), w01 = split2(w4.x, w4.y), w23 = split2(w4.z, w4.w);
      typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
      unsigned char* dx_ = buf + row * RB + c4 * 8;
      unsigned char* dw_ = buf + (128 + row) * RB + c4 * 8;
      *reinterpret_cast<u32x2*>(dx_) = u32x2{x01.h, x23.h};
      *reinterpret_cast<u32x2*>(dx_ + IMG) = u32x2{x01.m, x23.m};
      *reinterpret_cast<u32x2*>(dx_ + 2 * IMG) = u32x2{x01.l, x23.l};
      *reinterpret_cast<u32x2*>(dw_) = u32x2{w01.h, w23.h};
      *reinterpret_cast<u32x2*>(dw_ + IMG) = u32x2{w01.m, w23.m};
      *reinterpret_cast<u32x2*>(dw_ + 2 * IMG) = u32x2{w01.l, w23.l};
    }
  };

  const int nchunks = (K + KC - 1) / KC;
  int64_t r0 = (L / nt) * TM;
  int n0 = (int)(L % nt) * TN;
  int sel = 0;
  issue(r0, n0, 0);
  commit(sm8);
  __syncthreads();
  const int fa = (wm * 64 + (lane & 31)) * RB + (lane >> 5) * 16;          // A fragment of row tile 0 (tile 1: + 32 rows)
  const int fb = (128 + wn * 64 + (lane & 31)) * RB + (lane >> 5) * 16;    // B fragment of column tile 0
  for (;;) {
    const int64_t Ln = L + nslots;
    const bool more = Ln < Lend;  // block-uniform
    const int64_t nr0 = more ? (Ln / nt) * TM : 0;
    const int nn0 = more ? (int)(Ln % nt) * TN : 0;

    f32x16 ahi[2][2], alo[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) ahi[i][j][r] = 0.f, alo[i][j][r] = 0.f;

    for (int c = 0; c < nchunks; ++c) {
      const unsigned char* buf = sm8 + (size_t)sel * 3 * IMG;
      const bool lastc = c + 1 == nchunks;
      if (!lastc) issue(r0, n0, (c + 1) * KC);
      else if (more) issue(nr0, nn0, 0);
      bf16x8 a[2][3], b[2][3];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          a[i][p] = *reinterpret_cast<const bf16x8*>(buf + p * IMG + fa + i * 32 * RB);
          b[i][p] = *reinterpret_cast<const bf16x8*>(buf + p * IMG + fb + i * 32 * RB);
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          alo[i][j] = mfma_lo(alo[i][j], a[i][0], a[i][1], a[i][2], b[j][0], b[j][1], b[j][2]);
          alo[i][j] = mfma_mid(alo[i][j], a[i][0], a[i][1], b[j][0], b[j][1]);
          ahi[i][j] = mfma_hi(ahi[i][j], a[i][0], b[j][0]);
        }
      if (!lastc || more) commit(sm8 + (size_t)(sel ^ 1) * 3 * IMG);
      __syncthreads();
      sel ^= 1;
    }

    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int64_t nr = (rows - r0) < TM ? (rows - r0) : TM;
    const int ncols = (N - n0) < TN ? (N - n0) : TN;
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(Y + r0 * ldy + n0, win_bytes(nr, ldy, ncols));
    const __amdgpu_buffer_rsrc_t rz = make_rsrc(has_z ? Z + r0 * ldz + n0 : Y, has_z ? win_bytes(nr, ldz, ncols) : 0);
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(has_add ? add + r0 * ldadd + n0 : Y, has_add ? win_bytes(nr, ldadd, ncols) : 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = wn * 64 + j * 32 + (lane_e & 31);
        const bool jok = col < ncols;
        const float bj = (EPI == EPI_BIAS && bias && jok) ? bias[n0 + col] : 0.f;
        float zv[16], av[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rr = wm * 64 + i * 32 + d_row(r, lane_e);
          zv[r] = buf_ld1(rz, jok ? (unsigned)((rr * ldz + col) * 4) : kOOB);    // 0 when absent
          av[r] = buf_ld1(ra, jok ? (unsigned)((rr * ldadd + col) * 4) : kOOB);  // 0 when absent
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rr = wm * 64 + i * 32 + d_row(r, lane_e);
          float v = ahi[i][j][r] + alo[i][j][r];
          if (EPI == EPI_DX) {
            if (silu) {  // block-uniform
              v *= has_z ? gcl::dsilu_f(zv[r]) : 1.f;
            } else {
              const bool neg = has_z && (zv[r] <= 0.f);
              slope_acc += neg ? (double)(v * zv[r]) : 0.0;
              v = neg ? v * slope : v;
            }
          } else {
            v += bj;
          }
          v += av[r];
          buf_st1(ry, jok ? (unsigned)((rr * ldy + col) * 4) : kOOB, v);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (!more) break;
    L = Ln;
    r0 = nr0;
    n0 = nn0;
  }
  if (EPI == EPI_DX && slope_part) {
    for (int off = 32; off > 0; off >>= 1) slope_acc += __shfl_down(slope_acc, off, 64);
    double* dred = reinterpret_cast<double*>(smem);  // all fragment reads ended at the last barrier
    if (lane == 0) dred[wave] = slope_acc;
    __syncthreads();
    if (tid == 0) slope_part[blockIdx.x] = (dred[0] + dred[1]) + (dred[2] + dred[3]);
  }
}

struct GtGeom {
  int mi;
  int nt;
  int64_t total;
  int per_xcd;
  unsigned grid;
};
static inline GtGeom gt_geom(int64_t rows, int N) {
  GtGeom g;
  g.nt = (N + kGtTN - 1) / kGtTN;
  // 128-row tiles unless they would give the 512 persistent blocks fewer than ~3 tiles each
  g.mi = (gcl::cdiv(rows, 128) * g.nt < 3 * 2 * gcl::kNumCU) ? 1 : 2;  // env GCL_GT_MI overrides (experiments)
  static const int force = [] { const char* e = getenv("GCL_GT_MI"); return e ? atoi(e) : 0; }();
  if (force == 1 || force == 2) g.mi = force;
  const int TM = 64 * g.mi;
  g.total = gcl::cdiv(rows, TM) * g.nt;
  // whole row tiles per XCD so that the column tiles of a row tile share an L2
  const int64_t row_tiles_per_xcd = gcl::cdiv(gcl::cdiv(rows, TM), gcl::kNumXCD);
  g.per_xcd = (int)(row_tiles_per_xcd * g.nt);
  // persistent: the resident blocks per CU (LDS-limited: 2 at 128-row tiles, 3 at 64-row tiles)
  const int cap = (g.mi == 1 ? 3 : 2) * gcl::kNumCU / gcl::kNumXCD;
  const int slots = g.per_xcd < cap ? g.per_xcd : cap;
  g.grid = (unsigned)(slots * gcl::kNumXCD);
  return g;
}
