// Experiment (round 4): the streaming dX kernel with FOUR waves per workgroup, one per SIMD, each owning all 128 rows x 96 columns
// (72 MFMAs, 30 fragment reads and ~190 other instructions per wave and k-step instead of 36 / 18 / ~127 on two waves per SIMD):
// does a single in-order stream per SIMD with fewer non-MFMA instructions per MFMA beat the two-wave form on a power-bound
// kernel?  Whole tiles only, one K-segment, k-scale vector required.  Same MFMA order per accumulator as the shipped kernel, so
// the two outputs must be the same bits.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/probe/nt_w4_probe.hip -o tools/probe/bin/nt_w4_probe
#include "../../ofq_amd/csrc/libofq.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

template <int NJ>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void nt_w4_kernel(QNtSkArgs p) {
  constexpr int BM = 128, BN = 128 * NJ, NS = 3, MI = 4;
  constexpr int PLANE = BM * QBS_LD;
  constexpr int STAGE = NS * PLANE + BN * QBS_LD;
  constexpr int NA = 4;                 // dY float4 per thread and k-step: rows (tid >> 3) + 32 i
  constexpr int NB = 2 * NJ;            // 16-byte weight chunks per thread and k-step: rows (tid >> 2) + 64 i
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int G = gridDim.x, w = blockIdx.x;
  const int nkt = p.nkt;
  const unsigned u_begin = 2u * (unsigned)((p.units * (unsigned long long)w) / (unsigned long long)G);
  const unsigned u_end = 2u * (unsigned)((p.units * (unsigned long long)(w + 1)) / (unsigned long long)G);
  const unsigned kqa4 = (unsigned)(tid & 7) * 16u, kqb2 = (unsigned)(tid & 3) * 16u;
  const int kqa = (tid & 7) * 4, kqb = (tid & 3) * 8;

  f32x16q acc[MI][NJ];
  f32x4v ra[2][NA], rks[2];
  i32x4 rb[NB];
  using Slot0 = std::integral_constant<int, 0>;
  using Slot1 = std::integral_constant<int, 1>;

  unsigned la_u = u_begin, lb_u = u_begin;
  int la_kt = (int)(u_begin % (unsigned)nkt), lb_kt = la_kt;
  int la_tile = (int)(u_begin / (unsigned)nkt), lb_tile = la_tile;
  unsigned voA[NA], voB[NB];
  auto set_a_tile = [&](int tile) {
    const int m0 = (tile / p.tiles_n) * BM;
#pragma unroll
    for (int i = 0; i < NA; ++i) voA[i] = (unsigned)min(m0 + ((tid >> 3) + 32 * i), p.M - 1) * p.seg[0].lda4 + kqa4;
  };
  auto set_b_tile = [&](int tile) {
    const int n0 = (tile % p.tiles_n) * BN;
#pragma unroll
    for (int i = 0; i < NB; ++i) voB[i] = (unsigned)min(n0 + ((tid >> 2) + 64 * i), p.N - 1) * p.seg[0].ldb2 + kqb2;
  };
  set_a_tile(la_tile);
  set_b_tile(lb_tile);
  auto adv_a = [&]() {
    if (la_u + 1 < u_end) {
      ++la_u;
      if (++la_kt == nkt) { la_kt = 0; set_a_tile(++la_tile); }
    }
  };
  auto adv_b = [&]() {
    if (lb_u + 1 < u_end) {
      ++lb_u;
      if (++lb_kt == nkt) { lb_kt = 0; set_b_tile(++lb_tile); }
    }
  };
  auto a_base = [&](int kt) -> const char* { return reinterpret_cast<const char*>(p.seg[0].A) + (size_t)kt * (QBS_BK * 4); };
  auto s_base = [&](int kt) -> const char* { return reinterpret_cast<const char*>(p.seg[0].s) + (size_t)kt * (QBS_BK * 4); };
  auto b_base = [&](int kt) -> const char* { return reinterpret_cast<const char*>(p.seg[0].B) + (size_t)kt * (QBS_BK * 2); };
  auto gload = [&](auto SLOT) {
    constexpr int sl = decltype(SLOT)::value;
    const char* ab = a_base(la_kt);
    rks[sl] = *reinterpret_cast<const f32x4v*>(s_base(la_kt) + kqa4);
#pragma unroll
    for (int i = 0; i < NA; ++i) ra[sl][i] = *reinterpret_cast<const f32x4v*>(ab + voA[i]);
    adv_a();
  };
  auto gload_b = [&]() {
    const char* bb = b_base(lb_kt);
#pragma unroll
    for (int i = 0; i < NB; ++i) rb[i] = *reinterpret_cast<const i32x4*>(bb + voB[i]);
    adv_b();
  };
  auto lstore = [&](unsigned char* sb, auto SLOT) {
    constexpr int sl = decltype(SLOT)::value;
    const f32x2v k01 = {rks[sl][0], rks[sl][1]}, k23 = {rks[sl][2], rks[sl][3]};
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int row = (tid >> 3) + 32 * i;
      const f32x2v a01 = {ra[sl][i][0], ra[sl][i][1]}, a23 = {ra[sl][i][2], ra[sl][i][3]};
      unsigned lo[NS], hi[NS];
      split_pair_bf16<NS>(a01 * k01, lo);
      split_pair_bf16<NS>(a23 * k23, hi);
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        uint2 wv;
        wv.x = lo[q];
        wv.y = hi[q];
        *reinterpret_cast<uint2*>(&sb[q * PLANE + row * QBS_LD + kqa * 2]) = wv;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i)
      *reinterpret_cast<i32x4*>(&sb[NS * PLANE + ((tid >> 2) + 64 * i) * QBS_LD + kqb * 2]) = rb[i];
    __builtin_amdgcn_sched_barrier(0);
  };

  // 72 MFMAs per wave and k-step in six groups (ks, plane) of 4 x NJ; the A fragments of the NEXT group and the B fragments
  // of the second half are read behind the first MFMAs of the running group; staging pieces as in the shipped kernel
  constexpr int NM = 2 * NS * MI * NJ, GRP = MI * NJ, NPA = 17, NP = NA * NPA + NB + NB + 1 + NA;
  auto step = [&](const unsigned char* cur, unsigned char* nxt, auto SLOT) {
    constexpr int sl = decltype(SLOT)::value;
    const unsigned char* a = &cur[l31 * QBS_LD + lh * 16];
    const unsigned char* b = &cur[NS * PLANE + (wn * 32 * NJ + l31) * QBS_LD + lh * 16];
    bf16x8 av[2][MI], bv[2][NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) bv[0][j] = *reinterpret_cast<const bf16x8*>(b + j * 32 * QBS_LD);
#pragma unroll
    for (int i = 0; i < MI; ++i) av[0][i] = *reinterpret_cast<const bf16x8*>(a + i * 32 * QBS_LD);
    __builtin_amdgcn_sched_barrier(0);
    float ksv[4], x_ = 0.f, r1_ = 0.f, p0v[2], p1v[2], r2v[2];
    unsigned lo[NS], hi[NS];
    const char* bb2 = b_base(lb_kt);
    const char* ab3 = a_base(la_kt);
    const char* sb3 = s_base(la_kt) + kqa4;
    auto piece = [&](auto P_) {
      constexpr int P = decltype(P_)::value;
      if constexpr (P < NA * NPA) {
        constexpr int i = P / NPA, r = P % NPA;
        if constexpr (r == 0 && i == 0) {
          asm volatile("" : "+v"(rks[sl]), "+v"(ra[sl][0]), "+v"(ra[sl][1]), "+v"(ra[sl][2]), "+v"(ra[sl][3]));
#pragma unroll
          for (int e = 0; e < 4; ++e) ksv[e] = rks[sl][e];
        }
        if constexpr (r < 14) {
          constexpr int pr = r / 7, rr = r % 7;
          if constexpr (rr < 6) {
            constexpr int el = rr / 3, st = rr % 3, e = pr * 2 + el;
            if constexpr (st == 0) valu_mul_hi16(ra[sl][i][e], ksv[e], x_, p0v[el]);
            if constexpr (st == 1) valu_sub_hi16(x_, p0v[el], r1_, p1v[el]);
            if constexpr (st == 2) { r2v[el] = valu_sub(r1_, p1v[el]); }
          } else {
            valu_pack3_hi16(p0v, p1v, r2v, pr == 0 ? lo : hi);
          }
        } else {
          constexpr int q = r - 14;
          uint2 wv;
          wv.x = lo[q];
          wv.y = hi[q];
          *reinterpret_cast<uint2*>(&nxt[q * PLANE + ((tid >> 3) + 32 * i) * QBS_LD + kqa * 2]) = wv;
        }
      } else if constexpr (P < NA * NPA + NB) {
        constexpr int i = P - NA * NPA;
        asm volatile("" : "+v"(rb[i]));
        *reinterpret_cast<i32x4*>(&nxt[NS * PLANE + ((tid >> 2) + 64 * i) * QBS_LD + kqb * 2]) = rb[i];
      } else if constexpr (P < NA * NPA + 2 * NB) {
        constexpr int i = P - NA * NPA - NB;
        rb[i] = *reinterpret_cast<const i32x4*>(bb2 + voB[i]);
      } else {
        constexpr int wq = P - NA * NPA - 2 * NB;
        if constexpr (wq == 0) rks[sl] = *reinterpret_cast<const f32x4v*>(sb3);
        else ra[sl][wq - 1] = *reinterpret_cast<const f32x4v*>(ab3 + voA[wq - 1]);
      }
    };
    static_for<NM>([&](auto G_) {
      constexpr int Gi = decltype(G_)::value;
      constexpr int grp = Gi / GRP, ks = grp / NS, q = grp % NS, i = (Gi / NJ) % MI, j = Gi % NJ, in = Gi % GRP;
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[grp & 1][i], bv[ks][j], acc[i][j], 0, 0, 0);
      if constexpr (grp + 1 < 2 * NS && in < MI) {               // A fragments of the next group (ks', q')
        constexpr int g2 = grp + 1, ks2 = g2 / NS, q2 = g2 % NS;
        av[g2 & 1][in] = *reinterpret_cast<const bf16x8*>(a + q2 * PLANE + in * 32 * QBS_LD + ks2 * 32);
      }
      if constexpr (grp == 0 && in >= MI && in < MI + NJ) bv[1][in - MI] = *reinterpret_cast<const bf16x8*>(b + (in - MI) * 32 * QBS_LD + 32);
      constexpr int P0 = Gi * NP / NM, P1 = (Gi + 1) * NP / NM;
      static_for<P1 - P0>([&](auto D_) { piece(std::integral_constant<int, P0 + decltype(D_)::value>{}); });
      __builtin_amdgcn_sched_barrier(0);
    });
    lds_barrier();
    adv_b();
    adv_a();
  };

#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  gload(Slot0());
  gload_b();
  gload(Slot1());
  lstore(smem, Slot0());
  gload_b();
  gload(Slot0());
  lds_barrier();

  unsigned u = u_begin;
  while (u < u_end) {
    const int tile = (int)(u / (unsigned)nkt);
    const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    for (int n2 = nkt >> 1; n2 > 0; --n2) {
      step(smem, smem + STAGE, Slot1());
      step(smem + STAGE, smem, Slot0());
    }
    int l31e = l31, lhe = lh;
    asm volatile("" : "+v"(l31e), "+v"(lhe));
    const float alpha = p.seg[0].alpha;
    float* Cs = p.C + (int64_t)m0 * p.ldc + n0;
    const int ldc = (int)p.ldc;
    const int nl0 = wn * 32 * NJ + l31e;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int eb = 0; eb < 4; ++eb) {
        const int mlb = (i * 32 + 8 * eb + 4 * lhe) * ldc + nl0;
#pragma unroll
        for (int ee = 0; ee < 4; ++ee)
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            Cs[mlb + ee * ldc + j * 32] = acc[i][j][eb * 4 + ee] * alpha;
            acc[i][j][eb * 4 + ee] = 0.f;
          }
      }
    u += (unsigned)nkt;
  }
}

static float med(std::vector<float> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
  const int M = 128 * 198, N = 384, G = 198;
  for (int K : {1536, 2304, 384}) {
    std::vector<float> hdy((size_t)M * K), hk(K);
    std::vector<unsigned short> hw((size_t)N * K);
    unsigned x = 12345u;
    auto rnd = [&]() { x = x * 1664525u + 1013904223u; return (x >> 8) * (1.0f / 16777216.0f); };
    for (auto& v : hdy) v = (rnd() - 0.5f) * 1e-3f;
    for (auto& v : hk) v = 0.01f + 0.1f * rnd();
    for (auto& v : hw) { float f = (float)(2 * ((int)(rnd() * 4.f) - 2) + 1); unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
    float *dy, *ks, *c0, *c1; unsigned short* wT; void* ws;
    hipMalloc(&dy, hdy.size() * 4); hipMalloc(&ks, K * 4); hipMalloc(&wT, hw.size() * 2);
    hipMalloc(&c0, (size_t)M * N * 4); hipMalloc(&c1, (size_t)M * N * 4);
    const size_t wsb = ofq_qgemm_bf16s_nt_sk_ws_bytes(256);
    hipMalloc(&ws, wsb); hipMemset(ws, 0, wsb);
    hipMemcpy(dy, hdy.data(), hdy.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(ks, hk.data(), K * 4, hipMemcpyHostToDevice);
    hipMemcpy(wT, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    hipMemset(c0, 0xff, (size_t)M * N * 4); hipMemset(c1, 0xff, (size_t)M * N * 4);
    ofq_nt_seg seg = {dy, wT, ks, K, K, K, 0.25f};
    QNtSkArgs a = {};
    a.seg[0].A = dy; a.seg[0].B = wT; a.seg[0].s = ks; a.seg[0].lda4 = (unsigned)K * 4; a.seg[0].ldb2 = (unsigned)K * 2;
    a.seg[0].nkt = K / 32; a.seg[0].alpha = 0.25f;
    a.C = c1; a.ldc = N; a.M = M; a.N = N; a.nkt = K / 32; a.tiles_n = 1; a.accumulate = 0;
    a.units = (unsigned long long)(M / 128) * (unsigned long long)(K / 64);
    auto shipped = [&]() { return ofq_qgemm_bf16s_nt_sk(&seg, 1, c0, 0, M, N, N, -G, ws, wsb, nullptr); };
    auto w4 = [&]() { hipLaunchKernelGGL(nt_w4_kernel<3>, dim3(G), dim3(256), 0, nullptr, a); return (int)hipGetLastError(); };
    if (shipped() || w4()) { printf("launch failed\n"); return 1; }
    hipDeviceSynchronize();
    std::vector<float> h0((size_t)M * N), h1((size_t)M * N);
    hipMemcpy(h0.data(), c0, h0.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(h1.data(), c1, h1.size() * 4, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (size_t i = 0; i < h0.size(); ++i) bad += memcmp(&h0[i], &h1[i], 4) != 0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> t0, t1;
    for (int round = 0; round < 5; ++round) {
      for (int which = 0; which < 2; ++which) {
        hipEventRecord(e0, nullptr);
        for (int it = 0; it < 10; ++it) which ? w4() : shipped();
        hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        (which ? t1 : t0).push_back(ms * 100.f);
      }
    }
    printf("K=%4d: shipped 8-wave %7.1f us   4-wave %7.1f us   elements that differ: %zu of %zu\n", K, med(t0), med(t1), bad, h0.size());
    hipFree(dy); hipFree(ks); hipFree(wT); hipFree(c0); hipFree(c1); hipFree(ws);
  }
  return 0;
}
