// Is a word updated by device-scope atomicMax in kernel B seen by EVERY workgroup of the next kernel C (plain / scalar load), on all
// XCDs, when kernel A zero-filled it just before?  Mimics ops.absmax -> F16 GEMM prologue.  hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void zero_k(unsigned* w, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) w[i] = 0u; }
__global__ void amax_k(const float* x, int n, unsigned* w) {
  float m = 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) m = fmaxf(m, fabsf(x[i]));
  unsigned b = __float_as_uint(m);
  for (int o = 32; o > 0; o >>= 1) { unsigned t = __shfl_xor(b, o, 64); b = t > b ? t : b; }
  if ((threadIdx.x & 63) == 0 && b) atomicMax(w, b);
}
template <int MODE>
__global__ void read_k(const unsigned* w, unsigned* out) {
  unsigned v;
  if (MODE == 0) v = *w;                                                                     // uniform: scalar load
  else v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (threadIdx.x == 0) out[blockIdx.x] = v;
}
int main() {
  const int n = 1 << 22, NB = 1024, WORDS = 4096;
  float* x; unsigned *w, *out;
  hipMalloc(&x, n * 4); hipMalloc(&w, WORDS * 4); hipMalloc(&out, NB * 4);
  std::vector<float> hx(n);
  for (int i = 0; i < n; ++i) hx[i] = (float)((i * 2654435761u) >> 8) / 16777216.f;
  std::vector<unsigned> ho(NB);
  for (int mode = 0; mode < 2; ++mode) {
    long bad = 0, iters = 3000;
    for (long it = 0; it < iters; ++it) {
      hx[(it * 7919) % n] = 2.0f + it;                       // a new maximum every iteration
      hipMemcpy(x + (it * 7919) % n, &hx[(it * 7919) % n], 4, hipMemcpyHostToDevice);
      const int slot = it % WORDS;
      hipLaunchKernelGGL(zero_k, dim3(WORDS / 256), dim3(256), 0, 0, w, WORDS);
      hipLaunchKernelGGL(amax_k, dim3(512), dim3(256), 0, 0, x, n, w + slot);
      if (mode == 0) hipLaunchKernelGGL(read_k<0>, dim3(NB), dim3(64), 0, 0, w + slot, out);
      else hipLaunchKernelGGL(read_k<1>, dim3(NB), dim3(64), 0, 0, w + slot, out);
      hipMemcpy(ho.data(), out, NB * 4, hipMemcpyDeviceToHost);
      const unsigned expect = __builtin_bit_cast(unsigned, 2.0f + it);
      for (int b = 0; b < NB; ++b) if (ho[b] != expect) { if (bad < 5) printf("mode %d it %ld block %d: got %08x expect %08x\n", mode, it, b, ho[b], expect); ++bad; }
    }
    printf("mode %d (%s): %ld stale reads in %ld iterations x %d blocks\n", mode, mode ? "agent-scope atomic load" : "plain load", bad, iters, NB);
  }
  return 0;
}
