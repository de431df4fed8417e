// Stand-alone timing of the int8 forward GEMM (ofq_qgemm_i8_nt / _q) on the DeiT-S layer shapes, with optional
// experiment macros in the library TU (-DI8X_NO_F32_STORE, -DI8X_NO_KLOOP ...).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -w tools/probe/i8_probe.hip -o tools/probe/bin/i8_probe
#include "../../ofq_amd/csrc/libofq.hip"
#include <cstdio>
#include <vector>

int main() {
  const int M = 128 * 198, S = 198;
  struct Sh { const char* name; int N, K, codes, gelu, rowmul; } shapes[] = {
      {"v     (codes, col mode)", 384, 384, 2, 0, 1}, {"qkx   (codes, row mode x6)", 2304, 384, 1, 0, 6},
      {"proj  (no codes)", 384, 384, 0, 0, 1},        {"fc1   (codes, GELU)", 1536, 384, 1, 1, 1},
      {"fc2   (no codes)", 384, 1536, 0, 0, 1}};
  for (auto& sh : shapes) {
    const int N = sh.N, K = sh.K;
    std::vector<int8_t> ha((size_t)M * K), hb((size_t)N * K);
    unsigned x = 777u;
    auto rnd = [&]() { x = x * 1664525u + 1013904223u; return (x >> 8) * (1.0f / 16777216.0f); };
    for (auto& v : ha) v = (int8_t)((int)(rnd() * 4.f) - 2);
    for (auto& v : hb) v = (int8_t)((int)(rnd() * 4.f) - 2);
    const int qS = sh.codes == 2 ? N : S * sh.rowmul;
    std::vector<float> hs(S), hcs(N), hq(qS);
    for (auto& v : hs) v = 0.05f + 0.1f * rnd();
    for (auto& v : hcs) v = 0.01f + 0.02f * rnd();
    for (auto& v : hq) v = 0.2f + 0.2f * rnd();
    int8_t *A, *B, *Q; float *C, *s, *cs, *qs, *bias;
    hipMalloc(&A, ha.size()); hipMalloc(&B, hb.size()); hipMalloc(&Q, (size_t)M * N); hipMalloc(&C, (size_t)M * N * 4);
    hipMalloc(&s, S * 4); hipMalloc(&cs, N * 4); hipMalloc(&qs, qS * 4); hipMalloc(&bias, N * 4);
    hipMemcpy(A, ha.data(), ha.size(), hipMemcpyHostToDevice); hipMemcpy(B, hb.data(), hb.size(), hipMemcpyHostToDevice);
    hipMemcpy(s, hs.data(), S * 4, hipMemcpyHostToDevice); hipMemcpy(cs, hcs.data(), N * 4, hipMemcpyHostToDevice);
    hipMemcpy(qs, hq.data(), qS * 4, hipMemcpyHostToDevice); hipMemset(bias, 0, N * 4);
    auto run = [&]() {
      if (sh.codes)
        return ofq_qgemm_i8_nt_q(A, B, C, bias, cs, 1.f, nullptr, s, S, 0.01f, M, N, K, K, K, N, Q, N, qs, qS, 0.01f, bias,
                                 sh.gelu ? 0 : -2, sh.gelu ? 3 : 1, sh.gelu, sh.rowmul, sh.rowmul > 1 ? N / sh.rowmul : N,
                                 sh.codes == 2, nullptr);
      return ofq_qgemm_i8_nt(A, B, C, bias, cs, 1.f, nullptr, s, S, 0.01f, M, N, K, K, K, N, nullptr);
    };
    for (int it = 0; it < 3; ++it) { int rc = run(); if (rc) { printf("rc=%d\n", rc); return 1; } }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, nullptr);
    const int iters = 20;
    for (int it = 0; it < iters; ++it) run();
    hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)M * N * (4 + (sh.codes ? 1 : 0)) + (double)M * K + (double)N * K;
    std::vector<float> h(2); hipMemcpy(h.data(), C, 8, hipMemcpyDeviceToHost);
    printf("%-28s N=%4d K=%4d  %7.1f us  %5.2f TB/s  %6.1f TOPS  C[0..1]=%g %g\n", sh.name, N, K, ms * 1e3 / iters,
           bytes / (ms / iters) / 1e9, 2.0 * M * N * K / (ms / iters) / 1e9, h[0], h[1]);
#ifdef I8X_TIMING
    unsigned long long d[8];
    hipMemcpyFromSymbol(d, HIP_SYMBOL(g_i8_dbg), sizeof(d));
    printf("   mid-grid workgroup, cycles: prologue %llu | k-loop %llu | row terms + barrier %llu | epilogue %llu | code tile out %llu | total %llu\n",
           d[1] - d[0], d[2] - d[1], d[3] - d[2], d[4] - d[3], d[5] - d[4], d[5] - d[0]);
#endif
    hipFree(A); hipFree(B); hipFree(Q); hipFree(C); hipFree(s); hipFree(cs); hipFree(qs); hipFree(bias);
  }
  return 0;
}
