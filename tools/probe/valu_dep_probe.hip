// VALU issue rate of ONE wave per SIMD: a chain of dependent v_fma_f32 against eight independent chains, and the same for
// v_exp_f32 / v_rcp_f32 -- decides whether the quantiser epilogues (long per-element dependent chains, 2-3 waves per SIMD)
// are bound by instruction count or by dependent-issue latency.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/valu_dep_probe.hip -o tools/probe/bin/valu_dep_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int ILP, int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  float x[ILP];
#pragma unroll
  for (int i = 0; i < ILP; ++i) x[i] = threadIdx.x * 0.001f + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < ILP; ++i) {
        if (MODE == 0) x[i] = __builtin_fmaf(x[i], a, b);
        else if (MODE == 1) x[i] = __builtin_amdgcn_exp2f(x[i]) * a;
        else x[i] = __builtin_amdgcn_rcpf(x[i]) + b;
        asm volatile("" : "+v"(x[i]));
      }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < ILP; ++i) s += x[i];
  if (s == 12345.678f) out[0] = s;
}

template <int ILP, int MODE>
static void run(const char* name, int waves_per_simd) {
  float* out; hipMalloc(&out, 4);
  const int iters = 20000;
  const int blocks = 256 * waves_per_simd;
  hipLaunchKernelGGL((k<ILP, MODE>), dim3(blocks), dim3(256), 0, 0, out, 100, 0.999f, 0.001f);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); 
  hipLaunchKernelGGL((k<ILP, MODE>), dim3(blocks), dim3(256), 0, 0, out, iters, 0.999f, 0.001f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_wave = (double)iters * 8 * ILP * (MODE == 0 ? 1 : 2);
  printf("%-28s ILP %d, %d wave(s)/SIMD: %.2f ns per instruction and wave  (%.1f cycles at 2.4 GHz)\n", name, ILP, waves_per_simd,
         ms * 1e6 / instr_per_wave, ms * 1e6 / instr_per_wave * 2.4);
  hipFree(out);
}

int main() {
  run<1, 0>("v_fma_f32 dependent", 1);  run<8, 0>("v_fma_f32 independent", 1);
  run<1, 0>("v_fma_f32 dependent", 2);  run<1, 0>("v_fma_f32 dependent", 3);  run<8, 0>("v_fma_f32 independent", 3);
  run<1, 1>("v_exp_f32+mul dependent", 1);  run<8, 1>("v_exp_f32+mul independent", 1);  run<1, 1>("v_exp_f32+mul dependent", 3);
  run<1, 2>("v_rcp_f32+add dependent", 1);  run<8, 2>("v_rcp_f32+add independent", 1);
  return 0;
}
