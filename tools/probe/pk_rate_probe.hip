// Is packed fp32 VALU math (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two elements per lane and instruction) full-rate in a
// PURE VALU stream?  (Beside MFMAs it is an anti-lever: ~10 cycles each in an MFMA shadow, tools/probe/filler_probe.hip.)  The
// quantiser epilogues of the int8 forward / recompute-backward kernels are VALU-issue-bound stretches WITHOUT MFMAs, 2-3 waves
// per SIMD; about half of their ~15-45 instructions per element are plain mul / add / fma.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/pk_rate_probe.hip -o tools/probe/bin/pk_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x2 __attribute__((ext_vector_type(2)));

// MODE 0: v_fma_f32, 1: v_pk_fma_f32, 2: v_pk_mul_f32, 3: v_pk_add_f32, 4: v_mul_f32 + v_med3 + v_rndne (a non-packable mix),
// 5: alternating v_pk_fma_f32 / v_med3_f32
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  constexpr int ILP = 8;
  f32x2 x[ILP];
  const f32x2 av = {a, a}, bv = {b, b};
#pragma unroll
  for (int i = 0; i < ILP; ++i) { x[i][0] = threadIdx.x * 0.001f + i; x[i][1] = threadIdx.x * 0.002f + i; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < ILP; ++i) {
        if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i][0]) : "v"(a), "v"(b));
        if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(av), "v"(bv));
        if (MODE == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(av));
        if (MODE == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(bv));
        if (MODE == 4) {
          if (r % 3 == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i][0]) : "v"(a));
          if (r % 3 == 1) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x[i][0]) : "v"(a), "v"(b));
          if (r % 3 == 2) asm volatile("v_rndne_f32 %0, %0" : "+v"(x[i][0]));
        }
        if (MODE == 5) {
          if (r & 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(av), "v"(bv));
          else asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x[i][0]) : "v"(a), "v"(b));
        }
      }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < ILP; ++i) s += x[i][0] + x[i][1];
  if (s == 12345.678f) out[0] = s;
}

template <int MODE>
static void run(const char* name, int waves_per_simd) {
  float* out; hipMalloc(&out, 4);
  const int iters = 20000;
  const int blocks = 256 * waves_per_simd;
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, 100, 0.999f, 0.001f);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, iters, 0.999f, 0.001f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_simd = (double)iters * 8 * 8 * waves_per_simd;
  printf("%-34s %d wave(s)/SIMD: %.2f ns per instruction and SIMD\n", name, waves_per_simd, ms * 1e6 / instr_per_simd);
  hipFree(out);
}

int main() {
  for (int w = 1; w <= 3; ++w) {
    run<0>("v_fma_f32", w);
    run<1>("v_pk_fma_f32 (2 elements)", w);
    run<2>("v_pk_mul_f32 (2 elements)", w);
    run<3>("v_pk_add_f32 (2 elements)", w);
    run<4>("v_mul / v_med3 / v_rndne mix", w);
    run<5>("v_pk_fma_f32 / v_med3_f32 alternating", w);
  }
  return 0;
}
