// Does v_mfma_f32_32x32x16_f16 keep fp16 DENORMAL inputs (or flush them)?  And: is a two-plane fp16 split of an fp32
// value (hi = rtz_f16(x), lo = rtz_f16(x - hi)) times a small integer exact to ~2^-22?   hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void probe(const float* a_in, const float* b_in, float* out) {
  // A: 32 x 16 (row = lane & 31, k = 8 * (lane >> 5) + e), B: 16 x 32 (col = lane & 31): out = A.B
  const int lane = threadIdx.x;
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = (_Float16)a_in[(lane & 31) * 16 + 8 * (lane >> 5) + e];
    b[e] = (_Float16)b_in[(lane & 31) * 16 + 8 * (lane >> 5) + e];
  }
  f32x16 acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  for (int e = 0; e < 16; ++e) out[lane * 16 + e] = acc[e];
}
int main() {
  float ha[32 * 16], hb[32 * 16], ho[64 * 16];
  for (int i = 0; i < 512; ++i) { ha[i] = 0.f; hb[i] = 0.f; }
  // row 0 of A: one denormal fp16 value 2^-20 at k = 0; row 1: 2^-24 (smallest denormal); row 2: normal 2^-14
  ha[0 * 16 + 0] = ldexpf(1.f, -20);
  ha[1 * 16 + 0] = ldexpf(1.f, -24);
  ha[2 * 16 + 0] = ldexpf(1.f, -14);
  ha[3 * 16 + 0] = 3.f;                 // B denormal test row
  for (int c = 0; c < 32; ++c) hb[c * 16 + 0] = (c == 1) ? ldexpf(1.f, -20) : 1024.f;
  float *da, *db, *dout;
  hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dout, sizeof ho);
  hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dout);
  hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
  // acc layout 32x32: lane l holds column (l & 31), rows (e & 3) + 8 * (e >> 2) + 4 * (l >> 5)
  auto at = [&](int row, int col) { int lh = (row >> 2) & 1, e = (row & 3) + 4 * (row >> 3); return ho[(col + 32 * lh) * 16 + e]; };
  printf("A denormal 2^-20 * 1024 = %g (expect %g)\n", at(0, 0), ldexpf(1.f, -10));
  printf("A denormal 2^-24 * 1024 = %g (expect %g)\n", at(1, 0), ldexpf(1.f, -14));
  printf("A normal   2^-14 * 1024 = %g (expect %g)\n", at(2, 0), ldexpf(1.f, -4));
  printf("B denormal 3 * 2^-20    = %g (expect %g)\n", at(3, 1), 3 * ldexpf(1.f, -20));
  return 0;
}
