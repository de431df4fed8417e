// Does VALU work of one wave overlap the MFMA stream of its SIMD partner?  512-thread blocks (two waves per SIMD):
// waves 0-3 run `nm` MFMAs (6 independent accumulators), waves 4-7 run `nv` integer/fp VALU ops (8 independent chains).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512) void k_(float* out, int nm, int nv, int mode, int vkind) {
  const int wid = threadIdx.x >> 6;
  float r = 0.f;
  bool do_m = (mode == 0) ? (wid < 4) : (mode == 1 ? (wid < 4) : (mode == 3));
  bool do_v = (mode == 0) ? (wid >= 4) : (mode == 2 ? (wid >= 4) : (mode == 3));
  if (mode == 4 || mode == 5) { do_m = wid >= 4; do_v = wid < 4; }       // roles swapped: MFMA on the younger waves
  if (mode == 5 && do_v) __builtin_amdgcn_s_setprio(3);
  if (mode == 6) { do_m = wid < 4; do_v = wid >= 4; if (do_v) __builtin_amdgcn_s_setprio(3); }
  if (mode == 7) {     // every wave: the same MFMA count as mode 3 (nm each), with `vkind` fp32 VALU ops after every MFMA in program order
    f32x16 acc[6];
    for (int i = 0; i < 6; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x + e); b[e] = (__bf16)(float)(e + 1); }
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = (float)(threadIdx.x * 8 + i) * 1e-3f;
    for (int it = 0; it < nm / 6; ++it) {
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (u < vkind) {
            const float h = __uint_as_float(__float_as_uint(x[u]) & 0xffff0000u);
            x[u] = __fsub_rn(x[u], h) + 1.0f;
          }
        }
      }
    }
    for (int i = 0; i < 6; ++i) r += acc[i][0];
    for (int i = 0; i < 8; ++i) r += x[i];
    out[blockIdx.x * 512 + threadIdx.x] = r;
    return;
  }
  if (do_m) {
    f32x16 acc[6];
    for (int i = 0; i < 6; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x + e); b[e] = (__bf16)(float)(e + 1); }
    for (int it = 0; it < nm / 6; ++it) {
#pragma unroll
      for (int i = 0; i < 6; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 6; ++i) r += acc[i][0];
  }
  if (do_v && vkind == 1) {      // integer-only VALU
    unsigned x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 8 + i;
    for (int it = 0; it < nv / 16; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const unsigned h = x[i] & 0xffff0000u;
        x[i] = ((x[i] ^ h) + 0x12345u) ^ (x[i] >> 3);
      }
    }
    for (int i = 0; i < 8; ++i) r += (float)x[i];
  } else if (do_v && vkind == 2) {   // LDS reads (ds_read_b128), no VALU
    __shared__ float4 lds[2048];
    lds[threadIdx.x] = make_float4(1.f, 2.f, 3.f, 4.f); lds[threadIdx.x + 512] = lds[threadIdx.x];
    lds[threadIdx.x + 1024] = lds[threadIdx.x]; lds[threadIdx.x + 1536] = lds[threadIdx.x];
    __syncthreads();
    float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
    int idx = threadIdx.x;
    for (int it = 0; it < nv / 48; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float4 v = lds[(idx + i * 64) & 2047];
        a4.x += v.x;
      }
      idx += 7;
    }
    r += a4.x;
  } else if (do_v) {
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = (float)(threadIdx.x * 8 + i) * 1e-3f;
    for (int it = 0; it < nv / 16; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float h = __uint_as_float(__float_as_uint(x[i]) & 0xffff0000u);
        x[i] = __fsub_rn(x[i], h) + 1.0f;
      }
    }
    for (int i = 0; i < 8; ++i) r += x[i];
  }
  out[blockIdx.x * 512 + threadIdx.x] = r;
}

int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int nm = 6000, nv = 48000;
  const char* names[7] = {"MFMA waves 0-3 + partner waves 4-7", "MFMA only (waves 0-3)", "partner only (waves 4-7)", "every wave: MFMA then partner work", "MFMA waves 4-7 + partner waves 0-3", "same, partner at s_setprio 3", "MFMA waves 0-3 + partner 4-7 at s_setprio 3"};
  const char* kinds[3] = {"fp32 VALU (and/sub/add)", "int VALU (and/xor/add/shift)", "LDS ds_read_b128"};
  for (int vkind = 0; vkind < 3; ++vkind) {
    printf("-- partner work: %s\n", kinds[vkind]);
    for (int mode = 0; mode < 7; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, nullptr);
        hipLaunchKernelGGL(k_, dim3(256), dim3(512), 0, nullptr, out, nm, nv, mode, vkind);
        hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("%-36s %8.1f us\n", names[mode], ms * 1e3);
    }
  }
  printf("-- every wave (2 per SIMD): %d MFMAs, k x 3 fp32 VALU instructions after each MFMA in program order\n", nm);
  for (int k = 0; k <= 8; k += 2) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0, nullptr);
      hipLaunchKernelGGL(k_, dim3(256), dim3(512), 0, nullptr, out, nm, nv, 7, k);
      hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("k = %d (%2d VALU per MFMA)               %8.1f us\n", k, 3 * k, ms * 1e3);
  }
  return 0;
}
