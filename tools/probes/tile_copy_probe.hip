// What does the tile structure of the source-tile aggregation cost against a plain stream copy?  Copies
// [B][n][64] floats in 64-row tiles with the aggregation's block -> (sample, tile) map, four ways:
//   P1 registers only, P2 registers -> LDS -> barrier -> registers, P3 LDS-DMA -> barrier -> registers,
//   P4 = P3 with 120-row images (own rows + 56 more rows that hit L2), P0 = flat one-float4-per-thread copy.
// Development probe, not part of the product.   hipcc -O3 --offload-arch=gfx950 -o tile_copy_probe tile_copy_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__global__ __launch_bounds__(256) void p0(const v4f* __restrict__ s, v4f* __restrict__ d, size_t n4) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) __builtin_nontemporal_store(s[i], d + i);
}

template <int K> __device__ __forceinline__ int bc(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x150 + K, 0xf, 0xf, true); }
typedef __attribute__((address_space(3))) const v4f* lds4_t;
__device__ __forceinline__ v4f mul_then_add(float w, v4f x, v4f a) {
#pragma clang fp contract(off)
  const v4f t = w * x;
  return a + t;
}

template <int MODE, int XCD, int HALO, int COMP = 0, int RECS = 0, int LIST = 0>
__global__ __launch_bounds__(256) void ptile(const float* __restrict__ H, float* __restrict__ Y, int n, int B, int ntiles,
                                             const int2* __restrict__ rec = nullptr, const int* __restrict__ list = nullptr) {
  extern __shared__ v4f img[];
  const int bid = blockIdx.x;
  int b, tile;
  if (XCD) {
    const int slot = bid >> 3;
    b = (bid & 7) + 8 * (slot / ntiles);
    tile = slot % ntiles;
  } else {
    b = bid / ntiles;
    tile = bid % ntiles;
  }
  if (b >= B) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane >> 4, l = lane & 15;
  const float* Hb = H + (size_t)b * n * 64;
  float* Yb = Y + (size_t)b * n * 64;
  const int row0 = tile * 64 + wave * 16 + sub;
  v4f v[4];
  if (MODE == 1 || MODE == 2) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = min(row0 + it * 4, n - 1);
      v[it] = *reinterpret_cast<const v4f*>(Hb + (size_t)row * 64 + l * 4);
    }
  }
  if (MODE == 2) {
#pragma unroll
    for (int it = 0; it < 4; ++it) img[(wave * 16 + it * 4) * 16 + lane] = v[it];
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) v[it] = img[((wave * 16 + it * 4 + sub + 17) & 63) * 16 + l];
  }
  if (MODE == 3) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = min(row0 + it * 4, n - 1);
      __builtin_amdgcn_global_load_lds((gptr_t)(Hb + (size_t)row * 64 + l * 4), (lptr_t)(img + (wave * 16 + it * 4) * 16), 16, 0, 0);
    }
    int lj[HALO > 0 ? HALO : 1];
    if (LIST) {  // the rows come from a per-tile list read with scalar loads (wave-uniform addresses)
      const int uw = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
      for (int q = 0; q < HALO; ++q) {
        const int* tl = list + (size_t)tile * (HALO * 16) + (uw + 4 * q) * 4;
        int j = tl[0];
        j = sub == 1 ? tl[1] : j;
        j = sub == 2 ? tl[2] : j;
        j = sub == 3 ? tl[3] : j;
        lj[q] = j;
      }
    }
    int2 rcv[4];
    if (RECS) {
#pragma unroll
      for (int it = 0; it < 4; ++it) rcv[it] = rec[(size_t)min(row0 + it * 4, n - 1) * 16 + l];
    }
#pragma unroll
    for (int q = 0; q < HALO; ++q) {  // extra rows of the neighbouring tiles (L2 hits mostly)
      const int row = LIST ? lj[q] : min(max(tile * 64 - 28 + (wave + 4 * q) * 4 + sub + (q >= HALO / 2 ? 64 : 0), 0), n - 1);
      __builtin_amdgcn_global_load_lds((gptr_t)(Hb + (size_t)row * 64 + l * 4), (lptr_t)(img + (64 + (wave + 4 * q) * 4) * 16), 16, 0, 0);
    }
    __syncthreads();
    if (COMP == 0) {
#pragma unroll
      for (int it = 0; it < 4; ++it) v[it] = img[((wave * 16 + it * 4 + sub + 17) & 63) * 16 + l];
    } else {
      // the aggregation's arithmetic on made-up edge records: COMP reads of 16 B per row group and edge, weight and
      // position broadcast by DPP, product and sum rounded separately
      const unsigned lb = (unsigned)(size_t)((lptr_t)img) + l * 16;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int r = wave * 16 + it * 4 + sub;
        const int rxb = RECS ? (rcv[it].x & 0xffff) << 8 : ((r * 7 + l * 13) % (64 + HALO * 16)) << 8;
        const int rw = RECS ? rcv[it].y : __float_as_int(0.125f + l);
        v4f a = {0.f, 0.f, 0.f, 0.f};
#define S(K) if (K < COMP) { const unsigned ad = (unsigned)bc<K>(rxb) + lb; const float wk = __int_as_float(bc<K>(rw)); const v4f x = *(lds4_t)ad; \
        a = mul_then_add(wk, x, a); }
        S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11)
#undef S
        v[it] = a;
      }
    }
  }
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = row0 + it * 4;
    if (row < n) __builtin_nontemporal_store(v[it], reinterpret_cast<v4f*>(Yb + (size_t)row * 64 + l * 4));
  }
}

template <typename F>
void timeit(const char* name, F launch, double bytes) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  std::vector<float> ts;
  for (int it = 0; it < 25; ++it) {
    hipEventRecord(a);
    launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    if (it >= 5) ts.push_back(ms * 1e3f);
  }
  std::sort(ts.begin(), ts.end());
  printf("%-52s median %7.1f us  min %7.1f us  = %7.1f GB/s\n", name, ts[ts.size() / 2], ts[0], bytes / ts[ts.size() / 2] / 1e3);
}

int main() {
  const int B = 64, n = 10242, ntiles = (n + 63) / 64;
  const size_t bytes = (size_t)B * n * 64 * 4;
  float *s, *d;
  hipMalloc(&s, bytes);
  hipMalloc(&d, bytes);
  {
    std::vector<float> hbuf(bytes / 4);
    unsigned x = 12345;
    for (auto& f : hbuf) {
      x = x * 1664525u + 1013904223u;
      f = (float)(int)(x >> 8) * (1.0f / 8388608.0f) - 1.0f;
    }
    hipMemcpy(s, hbuf.data(), bytes, hipMemcpyHostToDevice);  // random data (constant bytes copy ~15 % faster than they should)
  }
  const size_t n4 = bytes / 16;
  const double tot = 2.0 * bytes;
  timeit("P0 flat copy, one float4 per thread, nt store", [&] { hipLaunchKernelGGL(p0, dim3((n4 + 255) / 256), dim3(256), 0, 0, (const v4f*)s, (v4f*)d, n4); }, tot);
  const int nb = 8 * ((B + 7) / 8) * ntiles;
#define RUN(MODE, XCD, HALO, LDS, NAME) \
  timeit(NAME, [&] { hipLaunchKernelGGL((ptile<MODE, XCD, HALO>), dim3(nb), dim3(256), LDS, 0, s, d, n, B, ntiles); }, tot)
  RUN(1, 0, 0, 0, "P1 tile copy through registers, sample-major blocks");
  RUN(1, 1, 0, 0, "P1 tile copy through registers, XCD map");
  RUN(2, 1, 0, 16384, "P2 regs -> LDS -> barrier -> regs, XCD map, 16 KB LDS");
  RUN(2, 1, 0, 31 * 1024, "P2 same with 31 KB LDS per block (5 blocks / CU)");
  RUN(3, 1, 0, 16384, "P3 LDS-DMA -> barrier -> regs, XCD map, 16 KB LDS");
  RUN(3, 1, 0, 31 * 1024, "P3 same with 31 KB LDS per block (5 blocks / CU)");
  RUN(3, 1, 4, 31 * 1024, "P4 LDS-DMA own rows + 56 neighbour rows, 31 KB LDS");
  RUN(3, 0, 4, 31 * 1024, "P4 same, sample-major blocks");
#define RUNC(COMP, NAME) \
  timeit(NAME, [&] { hipLaunchKernelGGL((ptile<3, 1, 4, COMP>), dim3(nb), dim3(256), 31 * 1024, 0, s, d, n, B, ntiles); }, tot)
  RUNC(1, "P5 = P4 + 1 edge read per row (DPP + ds_read_b128 + mul, add)");
  RUNC(4, "P5 with 4 edges per row");
  RUNC(8, "P5 with 8 edges per row");
  RUNC(12, "P5 with 12 edges per row");
  // records and lists as the real kernel has them (synthetic content of the same shape)
  int2* rec;
  int* list;
  {
    std::vector<int> hr((size_t)n * 32), hl((size_t)ntiles * 64);
    for (int i = 0; i < n; ++i)
      for (int k = 0; k < 16; ++k) {
        hr[((size_t)i * 16 + k) * 2] = (i * 7 + k * 13) % 120;
        float wv = 0.125f + k;
        hr[((size_t)i * 16 + k) * 2 + 1] = *reinterpret_cast<int*>(&wv);
      }
    for (int t = 0; t < ntiles; ++t)
      for (int k = 0; k < 64; ++k) hl[(size_t)t * 64 + k] = std::min(std::max(t * 64 - 28 + k + (k >= 28 ? 64 : 0), 0), n - 1);
    hipMalloc(&rec, hr.size() * 4);
    hipMalloc(&list, hl.size() * 4);
    hipMemcpy(rec, hr.data(), hr.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(list, hl.data(), hl.size() * 4, hipMemcpyHostToDevice);
  }
#define RUNF(COMP, RECS, LIST, NAME) \
  timeit(NAME, [&] { hipLaunchKernelGGL((ptile<3, 1, 4, COMP, RECS, LIST>), dim3(nb), dim3(256), 31 * 1024, 0, s, d, n, B, ntiles, rec, list); }, tot)
  RUNF(8, 1, 0, "P6 = P5(8 edges) + edge records loaded per row (128 B / row)");
  RUNF(8, 0, 1, "P7 = P5(8 edges) + halo rows from a list (scalar loads)");
  RUNF(8, 1, 1, "P8 = P5(8 edges) + records + list");
  return 0;
}
