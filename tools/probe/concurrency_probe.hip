// Do two kernels from two streams share the chip?  Kernel A: nA workgroups (512 threads, 112 KB LDS: one per CU) that spin
// for `us` microseconds; kernel B: nB of the same on a second stream.  Prints wall time of A alone, B alone, A then B on one
// stream, and A | B on two streams (optionally B on a low-priority stream).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(512) void spin(float* out, long long ticks) {
  __shared__ float lds[112 * 256];
  lds[threadIdx.x] = (float)threadIdx.x;
  __syncthreads();
  const long long t0 = wall_clock64();
  float acc = lds[(threadIdx.x * 7) & 255];
  while (wall_clock64() - t0 < ticks) acc = acc * 1.0001f + 0.5f;
  if (acc == 12345.678f) out[blockIdx.x] = acc;
}

static double run(hipStream_t sa, hipStream_t sb, int nA, int nB, long long ticks, float* out, bool same_stream) {
  hipEvent_t e0, e1, eb;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&eb));
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, sa));
  if (!same_stream) CK(hipStreamWaitEvent(sb, e0, 0));
  if (nA) hipLaunchKernelGGL(spin, dim3(nA), dim3(512), 0, sa, out, ticks);
  if (nB) hipLaunchKernelGGL(spin, dim3(nB), dim3(512), 0, same_stream ? sa : sb, out, ticks);
  if (!same_stream) { CK(hipEventRecord(eb, sb)); CK(hipStreamWaitEvent(sa, eb, 0)); }
  CK(hipEventRecord(e1, sa));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3;
}

int main() {
  float* out; CK(hipMalloc(&out, 4096));
  hipStream_t sa, sb, slow;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CK(hipStreamCreateWithPriority(&slow, hipStreamNonBlocking, lo));
  printf("priority range: least %d greatest %d\n", lo, hi);
  const long long ticks = 100 * 100;      // wall_clock64 runs at 100 MHz: 100 us
  for (int rep = 0; rep < 2; ++rep) {
    printf("A=198 alone            %8.1f us\n", run(sa, sb, 198, 0, ticks, out, true));
    printf("B=58 alone             %8.1f us\n", run(sa, sb, 0, 58, ticks, out, true));
    printf("A=198 then B=58 (1 st) %8.1f us\n", run(sa, sb, 198, 58, ticks, out, true));
    printf("A=198 | B=58  (2 st)   %8.1f us\n", run(sa, sb, 198, 58, ticks, out, false));
    printf("A=198 | B=58  (B low)  %8.1f us\n", run(sa, slow, 198, 58, ticks, out, false));
    printf("A=198 | B=256 (2 st)   %8.1f us\n", run(sa, sb, 198, 256, ticks, out, false));
    printf("A=198 then B=256 (1st) %8.1f us\n", run(sa, sb, 198, 256, ticks, out, true));
    printf("A=256 | B=256 (2 st)   %8.1f us\n", run(sa, sb, 256, 256, ticks, out, false));
  }
  return 0;
}
