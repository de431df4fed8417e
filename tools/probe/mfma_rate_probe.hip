// MFMA issue rate of one SIMD as a function of (a) waves per SIMD issuing MFMAs, (b) whether consecutive MFMAs share
// their A/B operand registers, (c) the accumulator count.  256 blocks (one per CU when 512 threads; two per CU may
// share a CU when 256 threads -- the grid is sized so that every CU gets exactly `waves_per_simd` MFMA waves per SIMD).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe/mfma_rate_probe.hip -o tools/probe/bin/mfma_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// MODE 0: same a, b for every MFMA, 6 accumulators.  MODE 1: 2x3 tile (2 A fragments x 3 B fragments -> 6 accumulators),
// 3 A planes, like the GEMM kernels' k16 step: 18 MFMAs, each A fragment used 3 times in a row.
// MODE 2: as 1 but B-major order (each B fragment used 6 times in a row, A changes every MFMA).
template <int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void k_(float* out, int iters, int active_waves) {
  const int wid = threadIdx.x >> 6;
  float r = 0.f;
  if (wid < active_waves) {
    f32x16 acc[6];
    for (int i = 0; i < 6; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    bf16x8 a[6], b[3];
    for (int q = 0; q < 6; ++q) for (int e = 0; e < 8; ++e) a[q][e] = (__bf16)(float)((threadIdx.x + e + q) & 7);
    for (int q = 0; q < 3; ++q) for (int e = 0; e < 8; ++e) b[q][e] = (__bf16)(float)((e + q) & 3);
    for (int it = 0; it < iters; ++it) {
      if (MODE == 0) {
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
          for (int i = 0; i < 6; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc[i], 0, 0, 0);
      } else if (MODE == 1) {
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
              acc[i * 3 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q * 2 + i], b[j], acc[i * 3 + j], 0, 0, 0);
      } else {
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
          for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int i = 0; i < 2; ++i)
              acc[i * 3 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q * 2 + i], b[j], acc[i * 3 + j], 0, 0, 0);
      }
      // keep the operands opaque so the loop is not collapsed
#pragma unroll
      for (int q = 0; q < 6; ++q) asm volatile("" : "+v"(a[q]));
#pragma unroll
      for (int q = 0; q < 3; ++q) asm volatile("" : "+v"(b[q]));
    }
    for (int i = 0; i < 6; ++i) r += acc[i][0];
  }
  out[blockIdx.x * THREADS + threadIdx.x] = r;
}

template <int MODE, int THREADS>
static void run(const char* name, float* out, int blocks, int active) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 400;            // x 18 MFMAs
  float ms = 0.f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0, nullptr);
    hipLaunchKernelGGL((k_<MODE, THREADS>), dim3(blocks), dim3(THREADS), 0, nullptr, out, iters, active);
    hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double n = 18.0 * iters;    // MFMAs per active wave
  printf("%-64s blocks %4d thr %3d active waves %d : %8.1f us  -> %6.1f ns per MFMA per wave\n", name, blocks, THREADS, active,
         ms * 1e3, ms * 1e6 / n);
}

int main() {
  float* out; hipMalloc(&out, 2048 * 512 * 4);
  run<0, 256>("same A/B, 1 wave/SIMD", out, 256, 4);
  run<1, 256>("2x3 tile A-major, 1 wave/SIMD", out, 256, 4);
  run<2, 256>("2x3 tile B-major, 1 wave/SIMD", out, 256, 4);
  run<0, 512>("same A/B, 2 waves/SIMD (one block)", out, 256, 8);
  run<1, 512>("2x3 tile A-major, 2 waves/SIMD (one block)", out, 256, 8);
  run<2, 512>("2x3 tile B-major, 2 waves/SIMD (one block)", out, 256, 8);
  run<0, 512>("same A/B, 512-thread block, only waves 0-3 active", out, 256, 4);
  run<1, 512>("2x3 tile, 512-thread block, only waves 0-3 active", out, 256, 4);
  run<0, 256>("same A/B, 2 blocks of 256 per CU", out, 512, 4);
  run<1, 256>("2x3 tile, 2 blocks of 256 per CU", out, 512, 4);
  run<1, 256>("2x3 tile, 4 blocks of 256 per CU", out, 1024, 4);
  run<1, 128>("2x3 tile, 128-thread blocks (2 waves), 256 blocks", out, 256, 2);
  run<1, 64>("2x3 tile, 64-thread blocks, 256 blocks", out, 256, 1);
  run<1, 64>("2x3 tile, 64-thread blocks, 1024 blocks", out, 1024, 1);
  run<1, 64>("2x3 tile, 64-thread blocks, 2048 blocks", out, 2048, 1);
  return 0;
}
