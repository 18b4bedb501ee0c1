// Stream-copy ceiling of the box for the aggregation's footprint (168 MB read + 168 MB written), so that the
// roofline kernel can be read against what a pure copy reaches here.  Development probe, not part of the product.
//   hipcc -O3 --offload-arch=gfx950 -o copy_probe copy_probe.hip && ./copy_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));

template <bool NTL, bool NTS, int UNROLL>
__global__ __launch_bounds__(256) void copy_kernel(const v4f* __restrict__ src, v4f* __restrict__ dst, size_t n4) {
  const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
  for (size_t i = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x; i < n4; i += stride) {
    v4f v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
      if (i + u * 256 < n4) v[u] = NTL ? __builtin_nontemporal_load(src + i + u * 256) : src[i + u * 256];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
      if (i + u * 256 < n4) {
        if (NTS) __builtin_nontemporal_store(v[u], dst + i + u * 256);
        else dst[i + u * 256] = v[u];
      }
  }
}

template <bool NTL, bool NTS, int UNROLL>
void run(const char* name, const v4f* s, v4f* d, size_t n4, int blocks) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  std::vector<float> ts;
  for (int it = 0; it < 25; ++it) {
    hipEventRecord(a);
    hipLaunchKernelGGL((copy_kernel<NTL, NTS, UNROLL>), dim3(blocks), dim3(256), 0, 0, s, d, n4);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    if (it >= 5) ts.push_back(ms * 1e3f);
  }
  std::sort(ts.begin(), ts.end());
  printf("%-28s blocks %6d: median %7.1f us  min %7.1f us  = %7.1f GB/s (read+write)\n", name, blocks, ts[ts.size() / 2], ts[0],
         2.0 * n4 * 16 / ts[ts.size() / 2] / 1e3);
}

int main(int argc, char** argv) {
  const size_t bytes = argc > 1 ? (size_t)atol(argv[1]) : (size_t)64 * 10242 * 64 * 4;
  const size_t n4 = bytes / 16;
  v4f *s, *d;
  hipMalloc(&s, bytes);
  hipMalloc(&d, bytes);
  hipMemset(s, 1, bytes);
  printf("# copy of %.1f MB\n", bytes / 1e6);
  for (int blocks : {1024, 2048, 4096, 8192, 16384}) {
    run<false, false, 1>("plain u1", s, d, n4, blocks);
    run<false, true, 1>("nt store u1", s, d, n4, blocks);
    run<true, true, 1>("nt load+store u1", s, d, n4, blocks);
    run<false, true, 4>("nt store u4", s, d, n4, blocks);
    run<true, true, 4>("nt load+store u4", s, d, n4, blocks);
  }
  size_t full = (n4 + 255) / 256;
  run<false, true, 1>("one float4 per thread, nt st", s, d, n4, (int)full);
  run<false, false, 1>("one float4 per thread", s, d, n4, (int)full);
  return 0;
}
