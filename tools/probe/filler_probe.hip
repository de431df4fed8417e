// How many single-issue VALU instructions hide behind one v_mfma_f32_32x32x16_bf16 when they follow it in the SAME
// wave's program order (FILL per MFMA, compile-time, pinned with sched_barrier), for one and two waves per SIMD?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe/filler_probe.hip -o tools/probe/bin/filler_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int FILL, int KIND, int THREADS>
__global__ __launch_bounds__(THREADS) void k_(float* out, int iters) {
  f32x16 acc[6];
  for (int i = 0; i < 6; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  bf16x8 a[6], b[3];
  for (int q = 0; q < 6; ++q) for (int e = 0; e < 8; ++e) a[q][e] = (__bf16)(float)((threadIdx.x + e + q) & 7);
  for (int q = 0; q < 3; ++q) for (int e = 0; e < 8; ++e) b[q][e] = (__bf16)(float)((e + q) & 3);
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = (float)(threadIdx.x * 8 + i) * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 18; ++g) {
      acc[g % 6] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[g % 6], b[g % 3], acc[g % 6], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < FILL; ++u) {
        float& v = x[(g * FILL + u) & 7];
        if (KIND == 0) v = __fmaf_rn(v, 1.0001f, 0.5f);                                           // v_fma_f32
        else if (KIND == 1) v = __uint_as_float(__float_as_uint(v) & 0xffff0000u) + 1.0f;         // and + add (2 instr)
        else { typedef float f2 __attribute__((ext_vector_type(2))); f2 t = {v, x[((g * FILL + u) + 1) & 7]}; t = t * 1.0001f; v = t.x; x[((g * FILL + u) + 1) & 7] = t.y; }  // v_pk_mul_f32
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float r = 0.f;
  for (int i = 0; i < 6; ++i) r += acc[i][0];
  for (int i = 0; i < 8; ++i) r += x[i];
  out[blockIdx.x * THREADS + threadIdx.x] = r;
}

template <int FILL, int KIND, int THREADS>
static void run(float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 400;
  float ms = 0.f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0, nullptr);
    hipLaunchKernelGGL((k_<FILL, KIND, THREADS>), dim3(256), dim3(THREADS), 0, nullptr, out, iters);
    hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const char* kinds[3] = {"v_fma_f32", "v_and + v_add (2 instr per filler)", "v_pk_mul_f32"};
  printf("%d waves/SIMD, %d x %-36s per MFMA: %8.1f us -> %6.1f ns per MFMA per SIMD\n", THREADS / 256, FILL, kinds[KIND],
         ms * 1e3, ms * 1e6 / (18.0 * iters) / (THREADS / 256));
}

int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  run<0, 0, 256>(out); run<2, 0, 256>(out); run<4, 0, 256>(out); run<6, 0, 256>(out); run<8, 0, 256>(out);
  run<0, 0, 512>(out); run<1, 0, 512>(out); run<2, 0, 512>(out); run<3, 0, 512>(out); run<4, 0, 512>(out);
  run<5, 0, 512>(out); run<6, 0, 512>(out); run<8, 0, 512>(out);
  run<1, 1, 512>(out); run<2, 1, 512>(out); run<3, 1, 512>(out);
  run<1, 2, 512>(out); run<2, 2, 512>(out); run<4, 2, 512>(out);
  return 0;
}
