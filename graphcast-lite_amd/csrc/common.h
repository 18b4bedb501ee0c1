// Shared helpers for libgcl_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/gcl.h"

namespace gcl {

void set_error(const char* fmt, ...);

#define GCL_CHECK_ARG(cond, ...)        \
  do {                                  \
    if (!(cond)) {                      \
      gcl::set_error(__VA_ARGS__);      \
      return GCL_EINVAL;                \
    }                                   \
  } while (0)

#define GCL_CHECK_HIP(expr)                                                              \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      gcl::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return GCL_EHIP;                                                                   \
    }                                                                                    \
  } while (0)

#define GCL_CHECK_LAUNCH()                                                     \
  do {                                                                         \
    hipError_t _e = hipGetLastError();                                         \
    if (_e != hipSuccess) {                                                    \
      gcl::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
      return GCL_EHIP;                                                         \
    }                                                                          \
  } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

constexpr int kWave = 64;     // CDNA wavefront
constexpr int kNumXCD = 8;    // MI355X: 8 XCDs, blocks are dealt round-robin over them
constexpr int kNumCU = 256;
constexpr int kEll = 8;      // stride of the ELL prefix arrays
constexpr int kHeavy = 64;   // rows with more edges than this get a whole block
constexpr int kHaloRec = 16;             // edge records per row held in registers (one per lane of a 16-lane row group)
constexpr int kHaloMore = 1 << 17;       // flag in slot 15: the row has more than 16 edges (finish from the CSR arrays)
constexpr int kHaloSkip = 1 << 16;       // flag in slot 15: heavy row, done by agg_heavy_kernel (do not store)
constexpr int kHaloPosMask = 0xFFFF;

__device__ __forceinline__ float prelu_f(float x, float a) { return x > 0.f ? x : a * x; }
// Activation kinds of the dense kernels (GCL_ACT_* of gcl.h).  SiLU: x * sigmoid(x).
constexpr int kActNone = 0, kActPrelu = 1, kActSilu = 2;
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float silu_f(float x) { return x * sigmoid_f(x); }
__device__ __forceinline__ float act_f(float x, float a, int kind) { return kind == kActSilu ? silu_f(x) : prelu_f(x, a); }
// d act(z) / dz
__device__ __forceinline__ float dsilu_f(float z) {
  const float s = sigmoid_f(z);
  return s * (1.f + z * (1.f - s));
}

// out[i*ldo + j] (+)= sum_p part[p*pstride + i*pld + j], i < R, j < C (parallel over partials, fixed order)
int launch_reduce_parts(const float* part, int nparts, int64_t pstride, int pld, float* out, int ldo, int R, int C,
                        int accumulate, hipStream_t st);
// two vectors of one partial record in ONE launch: out0[j] (+)= sum_p part[p*pstride + j],
// out1[j] (+)= sum_p part[p*pstride + off1 + j], j < C (each vector padded to pld entries in the record)
int launch_reduce_parts2(const float* part, int nparts, int64_t pstride, int pld, int off1, float* out0, float* out1,
                         int C, int accumulate, hipStream_t st);

// Kernels that ask for more than 64 KiB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize raised first.
// The attribute belongs to the (function, DEVICE) pair, so the "already done" record is kept per device: a process
// that drives several GPUs (or a later hipSetDevice) sets it again where it is still missing.
int ensure_dyn_lds(const void* func, size_t bytes);

int launch_reduce_parts3(const float* part, int nparts, int64_t pstride, int pld, float* out0, int acc0, float* out1,
                         int acc1, float* out2, int acc2, int C, hipStream_t st);

}  // namespace gcl

// Source-tile ("halo") layout of one CSR direction: the rows are cut into tiles of T consecutive rows and each
// tile carries the list of DISTINCT source rows its edges read, so a block stages every source once in LDS
// (LDS-DMA row gather) and forms the sums from LDS (csrc/aggregate.hip, agg_halo_kernel).  Built on the host
// when the graph is created, and only when the tiling shares sources well enough to pay (graph.hip).
struct gcl_halo {
  int32_t T = 0, ntiles = 0;
  int32_t smax = 0;          // rows of a tile image: T own rows + the largest halo count (rounded up to 8); row `smax` is the zero row
  int32_t* list = nullptr;   // [ntiles * (smax - T)] sources of a tile OUTSIDE the tile (its halo), ascending, padded by repeating the last
  int32_t* cnt = nullptr;    // [ntiles] halo entries to stage (multiple of 8)
  int32_t* rec = nullptr;    // [n * 16 * 2] {image position | flags (slot 15), weight bits} of the first 16 edges of a row
                             // (GCL_GRAPH_GAT, transposed direction: the edge's forward CSR slot instead of the weight)
  int32_t* opos = nullptr;   // [E'] image position of every CSR slot (rows with more than 16 edges)
};

// Device-side graph arrays (owned by the handle).
struct gcl_graph {
  int32_t n = 0;
  int64_t e = 0;  // E'
  int32_t kind = 0;
  int32_t max_in_deg = 0;
  int32_t max_out_deg = 0;
  int32_t *rowptr = nullptr, *col = nullptr, *eperm = nullptr;
  int32_t *trowptr = nullptr, *tcol = nullptr, *tslot = nullptr;
  float *w = nullptr, *tw = nullptr;
  // fixed-stride prefix of every row (first kEll edges in CSR order; padding: col = row, w = 0)
  int32_t *ecol = nullptr, *tecol = nullptr, *teslot = nullptr;  // teslot: forward slot of a transposed prefix edge
  float *ew = nullptr, *tew = nullptr;
  int32_t ell_width = 8, tell_width = 8;  // how many prefix entries the kernels read unconditionally
  int32_t ell_cover = 8, tell_cover = 8;  // smallest of {2,4,8} covering >= 98 % of the rows (the one-kernel GCN layer)
  // rows with more than kHeavy edges: skipped by the row-group kernel, done by one block each
  int32_t *heavy = nullptr, *theavy = nullptr;
  int32_t n_heavy = 0, n_theavy = 0;
  // source-tile layouts: [direction: 0 forward, 1 transpose][0: T = 64, 1: T = 32]; T == 0 when not built
  gcl_halo halo[2][2];
  // Processing order of the per-edge kernels on large graphs: order16[d][k] = the k-th 16-row group to be processed.
  // A group is placed right after the LAST group it reads from (self-loops aside), so on the bipartite encoder /
  // decoder graphs a block of mesh rows runs while the grid rows it gathers are still in the XCD's L2 (and the other
  // way round for the transposed graph); nullptr on graphs small enough for the L2 anyway.  [0] forward, [1] transpose.
  int32_t* order16[2] = {nullptr, nullptr};
  int32_t n_order16 = 0;
  // host copy of the PyG-order edge list with loops (for export / prune)
  int64_t* h_edges = nullptr;  // [2, e]
};
