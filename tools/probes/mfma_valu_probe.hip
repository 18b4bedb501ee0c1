// Probe: can fp32 MFMA chains of one wave and VALU work of ANOTHER wave on the same SIMD overlap?
// 512-thread blocks (2 waves per SIMD), one block per CU.  Waves 0-3 run MFMA, waves 4-7 run VALU FMAs.
// Modes: 1 = MFMA only, 2 = VALU only, 3 = both (different waves), 4 = every wave alternates MFMA / VALU phases.
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_valu_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int NOPS = 0>  // NOPS: s_nop 7 (8 idle cycles) after every MFMA; KIND 0: 32x32x2 f32, 1: 16x16x4 f32, 2: 32x32x16 bf16 (nm counts 32-cycle MFMAs), 3: 16x16x32 bf16
__global__ __launch_bounds__(512, 2) void probe(float* out, int mode, int iters, int nm, int nv) {
  extern __shared__ float smem[];
  const int wave = threadIdx.x >> 6;
  // 5: only waves 4-7 run (VALU), 6: only waves 0-3 run (MFMA): the one-wave-per-SIMD baselines of mode 3
  // 7: mode 3 with the VALU waves at s_setprio 3;  8: roles swapped (older waves 0-3 VALU, younger 4-7 MFMA)
  // 9: mode 4 (every wave alternates) with s_setprio 3 around the VALU phase and 0 around the MFMA phase
  const bool swap = mode == 8;
  const bool do_m = mode == 1 || ((mode == 3 || mode == 6 || mode == 7) && wave < 4) || (swap && wave >= 4) || mode == 4 || mode == 9;
  const bool do_v = mode == 2 || ((mode == 3 || mode == 5 || mode == 7) && wave >= 4) || (swap && wave < 4) || mode == 4 || mode == 9;
  if (mode == 7 && __builtin_amdgcn_readfirstlane(wave) >= 4) __builtin_amdgcn_s_setprio(3);
  f32x16 a0 = {0}, a1 = {0};
  f32x4 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
  float x = threadIdx.x * 1e-3f, y = 1.0001f, v0 = x, v1 = x + 1, v2 = x + 2, v3 = x + 3, v4 = x + 4, v5 = x + 5, v6 = x + 6, v7 = x + 7;
  for (int it = 0; it < iters; ++it) {
    if (do_m) {
      for (int k = 0; k < nm; k += 2) {
        if (KIND == 0) {
          a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
          if (NOPS > 0) asm volatile("s_nop %0" ::"n"(NOPS - 1));
          a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
          if (NOPS > 0) asm volatile("s_nop %0" ::"n"(NOPS - 1));
        } else if (KIND == 2) {
          bf16x8 p = {(short)threadIdx.x, 1, 2, 3, 4, 5, 6, 7}, q = {7, 6, 5, 4, 3, 2, 1, (short)threadIdx.x};
          a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p, q, a0, 0, 0, 0);
          if (NOPS > 0) asm volatile("s_nop %0" ::"n"(NOPS - 1));
          a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(q, p, a1, 0, 0, 0);
          if (NOPS > 0) asm volatile("s_nop %0" ::"n"(NOPS - 1));
        } else if (KIND == 3) {
          bf16x8 p = {(short)threadIdx.x, 1, 2, 3, 4, 5, 6, 7}, q = {7, 6, 5, 4, 3, 2, 1, (short)threadIdx.x};
          c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p, q, c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q, p, c1, 0, 0, 0);
          c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p, p, c2, 0, 0, 0);
          c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q, q, c3, 0, 0, 0);
        } else {
          c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, c1, 0, 0, 0);
          c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, c2, 0, 0, 0);
          c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, y, c3, 0, 0, 0);
        }
      }
    }
    if (do_v) {
      if (mode == 9) __builtin_amdgcn_s_setprio(3);
      for (int k = 0; k < nv; k += 8) {
        v0 = fmaf(v0, y, x); v1 = fmaf(v1, y, x); v2 = fmaf(v2, y, x); v3 = fmaf(v3, y, x);
        v4 = fmaf(v4, y, x); v5 = fmaf(v5, y, x); v6 = fmaf(v6, y, x); v7 = fmaf(v7, y, x);
      }
      if (mode == 9) __builtin_amdgcn_s_setprio(0);
    }
  }
  float s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r];
  s += c0[0] + c1[1] + c2[2] + c3[3];
  const unsigned long long t_end = __builtin_amdgcn_s_memtime();
  if (s == 12345.678f) out[threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) reinterpret_cast<unsigned long long*>(out)[256 + blockIdx.x * 8 + wave] = t_end - t_begin;
}

// Interleaved stream of ONE wave: every MFMA is followed by NV independent v_fma_f32 of the same wave (the guide's
// "fillers in the MFMA gap").  WPS waves per SIMD run the same stream.
template <int KIND, int NV, int WPS>
__global__ __launch_bounds__(256 * WPS) void probe_il(float* out, int iters) {
  f32x16 a0 = {0}, a1 = {0};
  float x = threadIdx.x * 1e-3f, y = 1.0001f;
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = x + i;
  bf16x8 p = {(short)threadIdx.x, 1, 2, 3, 4, 5, 6, 7}, q = {7, 6, 5, 4, 3, 2, 1, (short)threadIdx.x};
  const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (KIND == 0) {
        if (k & 1) a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
        else a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a0, 0, 0, 0);
      } else {
        if (k & 1) a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p, q, a1, 0, 0, 0);
        else a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(q, p, a0, 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 7]) : "v"(y), "v"(x));
    }
  }
  const unsigned long long t_end = __builtin_amdgcn_s_memtime();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r];
  if (s == 12345.678f) out[threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) reinterpret_cast<unsigned long long*>(out)[256 + blockIdx.x * 16 + (threadIdx.x >> 6)] = t_end - t_begin;
}

template <int KIND, int NV, int WPS>
void run_il(const char* name) {
  float* out;
  hipMalloc(&out, 1 << 20);
  static unsigned long long host[256 * 16];
  const int iters = 400;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe_il<KIND, NV, WPS>), dim3(256), dim3(256 * WPS), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(host, reinterpret_cast<unsigned long long*>(out) + 256, sizeof(host), hipMemcpyDeviceToHost);
  unsigned long long mx = 0;
  for (int i = 0; i < 256 * 16; ++i) if ((i & 15) < 4 * WPS) mx = host[i] > mx ? host[i] : mx;
  printf("%s + %2d v_fma per MFMA, %d wave(s)/SIMD: %.1f cycles of SIMD time per MFMA (%.1f us)\n", name, NV, WPS,
         (double)mx / (iters * 16.0 * WPS), ms * 1e3);
  hipFree(out);
}

template <int KIND, int NOPS = 0>
void run(const char* name, int nm = 64) {
  float* out;
  hipMalloc(&out, 1 << 20);
  static unsigned long long host[256 * 8];
  hipFuncSetAttribute((const void*)probe<KIND, NOPS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int iters = 200, nv = 1024;  // per iteration: 64 MFMAs (4096 or 2048 pipe cycles), 1024 VALU FMAs
  for (int mode : {6, 3, 5}) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL((probe<KIND, NOPS>), dim3(256), dim3(512), 100 * 1024, 0, out, mode, iters, nm, nv);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(host, reinterpret_cast<unsigned long long*>(out) + 256, sizeof(host), hipMemcpyDeviceToHost);
    unsigned long long mx = 0;
    for (int i = 0; i < 256 * 8; ++i) mx = host[i] > mx ? host[i] : mx;
    printf("%s mode %d: %.1f us  (per iteration %.0f ns, %.0f shader cycles of the slowest wave -> %.2f GHz)\n", name, mode,
           ms * 1e3, ms * 1e6 / iters, (double)mx / iters, (double)mx / (ms * 1e6));
  }
}
int main() {
  run_il<0, 0, 1>("f32_32x32x2 "); run_il<0, 4, 1>("f32_32x32x2 "); run_il<0, 8, 1>("f32_32x32x2 ");
  run_il<0, 12, 1>("f32_32x32x2 "); run_il<0, 14, 1>("f32_32x32x2 "); run_il<0, 16, 1>("f32_32x32x2 "); run_il<0, 24, 1>("f32_32x32x2 ");
  run_il<0, 0, 2>("f32_32x32x2 "); run_il<0, 8, 2>("f32_32x32x2 "); run_il<0, 12, 2>("f32_32x32x2 "); run_il<0, 16, 2>("f32_32x32x2 ");
  run_il<2, 0, 1>("bf16_32x32x16"); run_il<2, 4, 1>("bf16_32x32x16"); run_il<2, 6, 1>("bf16_32x32x16"); run_il<2, 8, 1>("bf16_32x32x16");
  run_il<2, 4, 2>("bf16_32x32x16"); run_il<2, 6, 2>("bf16_32x32x16"); run_il<2, 8, 2>("bf16_32x32x16");
  if (getenv("PROBE_IL_ONLY")) return 0;
  run<0>("mfma_f32_32x32x2 ");
  run<0, 10>("mfma_f32_32x32x2 + s_nop 9 ", 64);
  run<0, 12>("mfma_f32_32x32x2 + s_nop 11", 64);
  run<0, 13>("mfma_f32_32x32x2 + s_nop 12", 64);
  run<0, 14>("mfma_f32_32x32x2 + s_nop 13", 64);
  run<0, 15>("mfma_f32_32x32x2 + s_nop 14", 64);
  run<0, 16>("mfma_f32_32x32x2 + s_nop 15", 64);
  run<2, 4>("mfma_bf16_32x32x16 x128 + s_nop 3", 128);
  run<2, 5>("mfma_bf16_32x32x16 x128 + s_nop 4", 128);
  run<2, 6>("mfma_bf16_32x32x16 x128 + s_nop 5", 128);
  run<2, 7>("mfma_bf16_32x32x16 x128 + s_nop 6", 128);
  run<2>("mfma_bf16_32x32x16 x128", 128);  // 128 x 32 cycles = 4096 matrix-pipe cycles per wave and iteration
  run<2>("mfma_bf16_32x32x16 x48 ", 48);   // the 3xbf16 budget of a 32x64x64 fp32 tile: 48 x 32 = 1536 cycles

  return 0;
}
