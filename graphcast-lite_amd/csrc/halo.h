// Device helpers shared by the source-tile ("halo") kernels: csrc/aggregate.hip (agg_halo_loop_kernel) and
// csrc/gcn_layer.hip (gcn_halo_fwd_kernel).  Layout of the tile images and edge records: common.h, gcl_halo.
#pragma once
#include "common.h"

namespace gcl {
namespace halo {

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// lane K of every 16-lane row, to all lanes of that row (DPP row_newbcast; bound_ctrl: no `old` operand to set up).
// `row_bcast<K>(x) + y` compiles to ONE v_add_u32_dpp.
template <int K>
__device__ __forceinline__ int row_bcast(int v) {
  return __builtin_amdgcn_update_dpp(0, v, 0x150 + K, 0xf, 0xf, true);
}

// product and sum rounded separately (PyG: message = w * x_j, then scatter_add); `x * y` alone may be contracted
__device__ __forceinline__ float mul_then_add(float wk, float v, float a) {
#pragma clang fp contract(off)
  const float t = wk * v;
  return a + t;
}

// One LDS-DMA piece as inline asm (invisible to hipcc's wait bookkeeping: the caller counts it): 64 lanes x 16 B from
// sbase + voff[lane] to LDS bytes [lds_byte, lds_byte + 1024).  M0 carries the LDS base; saved / restored here.
__device__ __forceinline__ void glds16(const char* sbase, unsigned voff, unsigned lds_byte) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_byte)
      : "memory");
}

}  // namespace halo
}  // namespace gcl
