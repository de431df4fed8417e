// Pure matrix-core loops for the power trace (tools/power_mfma_workload.py): what rate of v_mfma_f32_32x32x16_f16 /
// v_mfma_i32_32x32x32_i8 does the part SUSTAIN under its power cap when nothing but MFMAs is issued?  Shared library, one launch
// function; 6 independent accumulators per wave (no issue stalls on the accumulator chain), operands fixed in registers.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC tools/probe/mfma_power_probe.hip -o tools/probe/bin/libmfma_power.so
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ __launch_bounds__(512) void mfma_loop(float* out, int iters) {
  float r = 0.f;
  if (KIND == 0) {
    f32x16 acc[6];
    for (int i = 0; i < 6; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(float)((threadIdx.x + e) & 7); b[e] = (_Float16)(float)(e & 3); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 6; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
      asm volatile("" : "+v"(a), "+v"(b));
    }
    for (int i = 0; i < 6; ++i) r += acc[i][0];
  } else {
    i32x16 acc[6];
    for (int i = 0; i < 6; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0;
    i32x4 a = {(int)threadIdx.x, 0x01020304, 0x01010101, 0x02020202}, b = {0x01010101, 0x02020202, 0x01010101, 0x03030303};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 6; ++i) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[i], 0, 0, 0);
      asm volatile("" : "+v"(a), "+v"(b));
    }
    for (int i = 0; i < 6; ++i) r += (float)acc[i][0];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// kind 0: fp16 32x32x16 (32 768 flop per wave-instruction), 1: int8 32x32x32 (65 536 op); 24 MFMAs per iteration and wave
extern "C" int mfma_power_launch(int kind, float* out, int blocks, int threads, int iters, void* stream) {
  if (kind == 0) hipLaunchKernelGGL(mfma_loop<0>, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, out, iters);
  else hipLaunchKernelGGL(mfma_loop<1>, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, out, iters);
  return (int)hipGetLastError();
}
