// Stand-alone timing harness for the weight-gradient code GEMM (no Python): builds the library TU with optional
// experiment macros and times ofq_qgemm_bf16s_tn on the DeiT-S layer shapes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-DTNW_...] tools/probe/tn_probe.hip -o tools/probe/bin/tn_probe
#include "../../ofq_amd/csrc/libofq.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

int main(int argc, char** argv) {
  const int Ktok = 128 * 197, S = 197;
  const int shapes[4][2] = {{1152, 384}, {384, 384}, {1536, 384}, {384, 1536}};
  for (int si = 0; si < 4; ++si) {
    const int M = shapes[si][0], N = shapes[si][1];
    std::vector<float> hdy((size_t)Ktok * M), hs(S);
    std::vector<int8_t> hc((size_t)Ktok * N);
    unsigned x = 12345u;
    auto rnd = [&]() { x = x * 1664525u + 1013904223u; return (x >> 8) * (1.0f / 16777216.0f); };
    for (auto& v : hdy) v = (rnd() - 0.5f) * 1e-3f;
    for (auto& v : hc) v = (int8_t)((int)(rnd() * 4.f) - 2);
    for (auto& v : hs) v = 0.1f + rnd();
    float *dy, *s, *dW, *db, *baft; int8_t* codes; void* ws;
    hipMalloc(&dy, hdy.size() * 4); hipMalloc(&s, S * 4); hipMalloc(&dW, (size_t)M * N * 4); hipMalloc(&db, M * 4);
    hipMalloc(&baft, N * 4); hipMalloc(&codes, hc.size());
    hipMemcpy(dy, hdy.data(), hdy.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(s, hs.data(), S * 4, hipMemcpyHostToDevice);
    hipMemcpy(codes, hc.data(), hc.size(), hipMemcpyHostToDevice);
    hipMemset(baft, 0, N * 4);
    const int tiles = (M / 128) * (N / 384);
    int split = 256 / tiles; if (argc > 1) split = atoi(argv[1]);
    const size_t wsb = ofq_qgemm_bf16s_tn_ws_bytes(M, N, split);
    hipMalloc(&ws, wsb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) ofq_qgemm_bf16s_tn(dy, codes, dW, s, S, 0.01f, db, 1, baft, Ktok, M, N, M, N, split, ws, wsb, nullptr);
    hipEventRecord(e0, nullptr);
    const int iters = 20;
    for (int it = 0; it < iters; ++it) {
      int rc = ofq_qgemm_bf16s_tn(dy, codes, dW, s, S, 0.01f, db, 1, baft, Ktok, M, N, M, N, split, ws, wsb, nullptr);
      if (rc) { printf("rc=%d\n", rc); return 1; }
    }
    hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<float> h(8); hipMemcpy(h.data(), dW, 32, hipMemcpyDeviceToHost);
    printf("o=%4d c=%4d split=%3d  %7.1f us  %6.1f TF  dW[0..1]=%g %g\n", M, N, split, ms * 1e3 / iters,
           2.0 * Ktok * M * N / (ms / iters) / 1e9, h[0], h[1]);
#ifdef TNW_TIMING
    unsigned long long h_dbg[8][8];
    hipMemcpyFromSymbol(h_dbg, HIP_SYMBOL(g_tnw_dbg), sizeof(h_dbg));
    for (int w = 0; w < 8; ++w) {
      const double n = (double)h_dbg[w][7];
      printf("   wave %d: cycles/k-step  mfma-issue %5.0f | loadwait %5.0f  Asplit+write %5.0f  Bconv+write %5.0f | gload %5.0f  barrier %5.0f | total %5.0f\n",
             w, h_dbg[w][0] / n, h_dbg[w][4] / n, h_dbg[w][5] / n, h_dbg[w][1] / n, h_dbg[w][2] / n, h_dbg[w][3] / n, h_dbg[w][6] / n);
    }
#endif
    {   // input-gradient GEMM on the same layer: dX[Ktok, N] = (dy * ks[o]) @ wT[N, M]^T
      unsigned short* wT; float *ksc, *dX;
      hipMalloc(&wT, (size_t)N * M * 2); hipMalloc(&ksc, M * 4); hipMalloc(&dX, (size_t)Ktok * N * 4);
      std::vector<unsigned short> hw((size_t)N * M);
      for (auto& v : hw) { float f = (float)(2 * ((int)(rnd() * 4.f) - 2) + 1); unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
      std::vector<float> hk(M); for (auto& v : hk) v = 0.01f + 0.1f * rnd();
      hipMemcpy(wT, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
      hipMemcpy(ksc, hk.data(), M * 4, hipMemcpyHostToDevice);
      for (int it = 0; it < 3; ++it) ofq_qgemm_bf16s_nt(dy, wT, dX, ksc, 0.25f, 0, 3, Ktok, N, M, M, M, N, nullptr);
      hipEventRecord(e0, nullptr);
      for (int it = 0; it < iters; ++it) ofq_qgemm_bf16s_nt(dy, wT, dX, ksc, 0.25f, 0, 3, Ktok, N, M, M, M, N, nullptr);
      hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(h.data(), dX, 32, hipMemcpyDeviceToHost);
      printf("   NT dX N=%4d K=%4d          %7.1f us  %6.1f TF  dX[0..1]=%g %g\n", N, M, ms * 1e3 / iters,
             2.0 * Ktok * M * N / (ms / iters) / 1e9, h[0], h[1]);
      hipFree(wT); hipFree(ksc); hipFree(dX);
    }
    hipFree(dy); hipFree(s); hipFree(dW); hipFree(db); hipFree(baft); hipFree(codes); hipFree(ws);
  }
  return 0;
}
