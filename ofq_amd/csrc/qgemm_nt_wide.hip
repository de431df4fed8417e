// Wide input-gradient kernel for the linear layers: 8 waves own a 128 x (128*NJ) tile of dX, so a row panel of dY is
// scaled and split into its three bf16 planes once per NJ column blocks (once in total when N <= 384, the qkv / fc1 /
// proj case) instead of once per 128 columns.  Same pipeline as the wide dW kernel: double-buffered LDS, one LDS-only
// barrier per k-step, two register prefetch slots.
// LSQ = true: the epilogue is the backward of the layer's own input quantiser (ofq_lsq_bwd's arithmetic, element for
// element) applied to the dX tile while it is still in registers: dx, the per-row step-gradient partials [M][tiles_n]
// and the per-column offset-gradient partials [tiles_m][2][N] are written instead of dX, so dX never travels to HBM and
// back (8 of the 16 B/element of the unfused pair).
template <int NJ, bool LSQ, bool F16 = false>
__global__ __launch_bounds__(512) void qgemm_bf16s_nt_wide_kernel(QGemmArgs p) {
  constexpr int BM = 128, BN = 128 * NJ, NS = F16 ? 2 : 3;
  constexpr int PLANE = BM * QBS_LD;
  constexpr int STAGE = NS * PLANE + BN * QBS_LD;
  constexpr int NB = NJ;                               // 16-byte chunks of the weight tile per thread (BN*4/512)
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];
  int tm, tn, gby;
  qgemm_tile_id(p, tm, tn, gby);
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 2, wn = wid & 3;
  const int l31 = lane & 31, lh = lane >> 5;
  const float* A = (const float*)p.A;
  const unsigned short* B = (const unsigned short*)p.B;
  const int K = p.K;
  const int nkt = (K + QBS_BK - 1) / QBS_BK;

  // A: 128 rows x 32 fp32 = 1024 float4 -> 2 per thread (row = f >> 3);  B: BN rows x 32 bf16 -> NJ x 16 B per thread
  const float* pa[2];
  bool okA[2];
  const unsigned short* pb[NB];
  bool okB[NB];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (tid + 512 * i) >> 3;
    okA[i] = (m0 + row) < p.M;
#ifdef NTW_SAME_ROWS
    pa[i] = A + (int64_t)row * p.lda + (tid & 7) * 4;         // experiment: every workgroup reads rows 0..127 (L2 hits)
#else
    pa[i] = A + (int64_t)min(m0 + row, p.M - 1) * p.lda + (tid & 7) * 4;
#endif
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int row = (tid + 512 * i) >> 2;
    okB[i] = (n0 + row) < p.N;
    pb[i] = B + (int64_t)min(n0 + row, p.N - 1) * p.ldb + (tid & 3) * 8;
  }
  const int kqa = (tid & 7) * 4, kqb = (tid & 3) * 8;
  float sE = 1.f, inv_sE = 1.f;        // F16: the launch's power of two (see qgemm_bf16s_nt_wide_sk_kernel)
  if constexpr (F16) {
    const float m = p.s ? block512_absmax(p.s, K, reinterpret_cast<float*>(smem), tid) : 1.f;
    const float a = ofq_amax_load(p.amax);
    f16_plane_scale(a == a ? a * m : a, sE, inv_sE);
  }
  f32x4v ra[2][2], rks[2];
  i32x4 rb[NB];                                         // weights are L2-resident: one step of prefetch is enough
  bool rka[2], rkb;
  auto gload = [&](int kt, auto SLOT) {
    constexpr int sl = decltype(SLOT)::value;
    const int k0 = kt * QBS_BK;
    rka[sl] = (k0 + kqa) < K;                           // K % 8 == 0 (host check): chunks are all-in or all-out
    const int ka = rka[sl] ? k0 : 0;
    // no scale vector: the load still happens (from the dY panel, any valid address) and the value is replaced at the
    // LDS store; a load under `if (p.s)` ends in a register copy that has to wait for it, i.e. s_waitcnt vmcnt(0) here
    rks[sl] = *reinterpret_cast<const f32x4v*>(p.s ? p.s + ka + kqa : pa[0]);
#pragma unroll
    for (int i = 0; i < 2; ++i) ra[sl][i] = *reinterpret_cast<const f32x4v*>(pa[i] + ka);
  };
  auto gload_b = [&](int kt) {
    const int k0 = kt * QBS_BK;
    rkb = (k0 + kqb) < K;
    const int kb = rkb ? k0 : 0;
#pragma unroll
    for (int i = 0; i < NB; ++i) rb[i] = *reinterpret_cast<const i32x4*>(pb[i] + kb);
  };
  auto lstore = [&](unsigned char* sb, auto SLOT) {
    constexpr int sl = decltype(SLOT)::value;
    // pin the slot's registers here: the k-loop body is one basic block, and without an ordered use the selects on the
    // loaded values are placed right behind the loads' issue (one k-step early), where they wait for them
    asm volatile("" : "+v"(rks[sl]), "+v"(ra[sl][0]), "+v"(ra[sl][1]));
    f32x4v ks = rks[sl];
    if (!p.s) ks = f32x4v{1.f, 1.f, 1.f, 1.f};
    if constexpr (F16) ks = ks * sE;
    if (!rka[sl]) ks = f32x4v{0.f, 0.f, 0.f, 0.f};           // beyond K: zero pieces (register select, the loads are done)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (tid + 512 * i) >> 3;
      const float z = okA[i] ? 1.f : 0.f;
      const f32x2v k01 = {ks[0] * z, ks[1] * z}, k23 = {ks[2] * z, ks[3] * z};
      const f32x2v a01 = {ra[sl][i][0], ra[sl][i][1]}, a23 = {ra[sl][i][2], ra[sl][i][3]};
      unsigned lo[NS], hi[NS];
      if constexpr (F16) {
        const f32x2v x01 = a01 * k01, x23 = a23 * k23;
        split2_f16(x01[0], x01[1], lo[0], lo[1]);
        split2_f16(x23[0], x23[1], hi[0], hi[1]);
      } else {
        split_pair_bf16<NS>(a01 * k01, lo);
        split_pair_bf16<NS>(a23 * k23, hi);
      }
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        uint2 w;
        w.x = lo[q];
        w.y = hi[q];
        *reinterpret_cast<uint2*>(&sb[q * PLANE + row * QBS_LD + kqa * 2]) = w;
      }
      __builtin_amdgcn_sched_barrier(0);     // one row chunk at a time: interleaving them only costs registers
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) asm volatile("" : "+v"(rb[i]));
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int row = (tid + 512 * i) >> 2;
      const int m = (okB[i] && rkb) ? -1 : 0;
      *reinterpret_cast<i32x4*>(&sb[NS * PLANE + row * QBS_LD + kqb * 2]) = rb[i] & m;
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  f32x16q acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  auto compute = [&](const unsigned char* sb) {
    const unsigned char* a = &sb[(wm * 64 + l31) * QBS_LD + lh * 16];
    const unsigned char* b = &sb[NS * PLANE + (wn * 32 * NJ + l31) * QBS_LD + lh * 16];
    bf16x8 av[QBS_BK / 16][NS][2], bv[QBS_BK / 16][NJ];
#pragma unroll
    for (int ks = 0; ks < QBS_BK / 16; ++ks) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) bv[ks][j] = *reinterpret_cast<const bf16x8*>(b + j * 32 * QBS_LD + ks * 32);
#pragma unroll
      for (int q = 0; q < NS; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i)
          av[ks][q][i] = *reinterpret_cast<const bf16x8*>(a + q * PLANE + i * 32 * QBS_LD + ks * 32);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < QBS_BK / 16; ++ks)
#pragma unroll
      for (int q = 0; q < NS; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc[i][j] = mfma_16b<F16>(av[ks][q][i], bv[ks][j], acc[i][j]);
    __builtin_amdgcn_sched_barrier(0);       // the staging below consumes global loads: keep its waits behind the MFMAs
  };

  using Slot0 = std::integral_constant<int, 0>;
  using Slot1 = std::integral_constant<int, 1>;
  // The k-loop body is branch-free: past the last tile the loads repeat tile nkt-1 and the staging writes a stage that
  // nobody reads.  With `if (kt + 3 < nkt)` guards around the loads the compiler's wait-count pass merges the "not
  // issued" path with the steady state and ends up waiting for every outstanding load at the top of each k-step
  // (s_waitcnt vmcnt(0) right after the barrier), which cancels the two-step prefetch of the dY panel.
  const int klast = nkt - 1;
  // Issue order = consumption order (vmcnt counts in order): weights of tile kt+2, then the dY panel of tile kt+3, both
  // after the staging of tile kt+1; the staging of the next step then waits with 3 / 6 younger loads still in flight.
#ifdef NTW_SERIAL_STAGING
  auto step = [&](int kt, const unsigned char* cur, unsigned char* nxt, auto SLOT) {
    compute(cur);
    lstore(nxt, SLOT);
    gload_b(min(kt + 2, klast));
    gload(min(kt + 3, klast), SLOT);
    lds_barrier();
  };
#else
  // One k-step with the staging of tile kt+1 and the loads of tiles kt+2 / kt+3 cut into NP small pieces that are
  // spread behind the 12*NJ MFMAs of tile kt (see the note at static_for above).  Piece list, per 128-row chunk i of the
  // dY panel (17 pieces): for each half (x,y) / (z,w) of the float4 -- per element: [x = a*ks, p0 = hi16(x)],
  // [r1 = x - p0, p1 = hi16(r1)], [r2 = r1 - p1] (r2 has <= 8 significant bits: it is its own bf16 plane), then one
  // piece packing the three planes of the pair -- and three LDS stores; then the weight chunks (store each, no masks:
  // rows past N only feed columns that are never written, k past K is zeroed through ks), then the six loads in
  // consumption order.
  constexpr int NM = 4 * NS * NJ, NPA = F16 ? 10 : 17, NP = 2 * NPA + NB + NB + 3;
  auto step = [&](int kt, const unsigned char* cur, unsigned char* nxt, auto SLOT) {
    constexpr int sl = decltype(SLOT)::value;
    const unsigned char* a = &cur[(wm * 64 + l31) * QBS_LD + lh * 16];
    const unsigned char* b = &cur[NS * PLANE + (wn * 32 * NJ + l31) * QBS_LD + lh * 16];
    // fragments of the first 16-deep MFMA step up front, those of the second one behind the MFMAs that used up their
    // registers (9 instead of 18 LDS reads between the barrier and the first MFMA, 24 fewer live VGPRs)
    static_assert(QBS_BK == 32, "two MFMA steps per k-step");
    bf16x8 av[NS][2], bv[2][NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) bv[0][j] = *reinterpret_cast<const bf16x8*>(b + j * 32 * QBS_LD);
#pragma unroll
    for (int q = 0; q < NS; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i) av[q][i] = *reinterpret_cast<const bf16x8*>(a + q * PLANE + i * 32 * QBS_LD);
    __builtin_amdgcn_sched_barrier(0);
    float ksv[4], x_ = 0.f, r1_ = 0.f, p0v[2], p1v[2], r2v[2];
    unsigned lo[NS], hi[NS];
    const int kb2 = min(kt + 2, klast) * QBS_BK, ka3 = min(kt + 3, klast) * QBS_BK;
    auto piece = [&](auto P_) {
      constexpr int P = decltype(P_)::value;
      if constexpr (P < 2 * NPA) {
        constexpr int i = P / NPA, r = P % NPA;
        if constexpr (r == 0 && i == 0) {          // first touch of the slot: the wait for its loads lands here
          asm volatile("" : "+v"(rks[sl]), "+v"(ra[sl][0]), "+v"(ra[sl][1]));
#pragma unroll
          for (int e = 0; e < 4; ++e) ksv[e] = rka[sl] ? (p.s ? rks[sl][e] : 1.f) * (F16 ? sE : 1.f) : 0.f;
        }
        if constexpr (F16) {
          if constexpr (r < 8) {
            constexpr int pr = r / 4, st = r % 4, e = pr * 2;
            if constexpr (st == 0) valu_mul2(ra[sl][i][e], ksv[e], ra[sl][i][e + 1], ksv[e + 1], x_, r1_);
            if constexpr (st == 1) (pr == 0 ? lo : hi)[0] = valu_cvt_pk_f16(x_, r1_);
            if constexpr (st == 2) valu_resid2_f16((pr == 0 ? lo : hi)[0], x_, r1_, p0v[0], p0v[1]);
            if constexpr (st == 3) (pr == 0 ? lo : hi)[1] = valu_cvt_pk_f16(p0v[0], p0v[1]);
          } else {
            constexpr int q = r - 8;
            const int row = (tid + 512 * i) >> 3;
            uint2 w;
            w.x = lo[q];
            w.y = hi[q];
            *reinterpret_cast<uint2*>(&nxt[q * PLANE + row * QBS_LD + kqa * 2]) = w;
          }
        } else if constexpr (r < 14) {
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
          const int row = (tid + 512 * i) >> 3;
          uint2 w;
          w.x = lo[q];
          w.y = hi[q];
          *reinterpret_cast<uint2*>(&nxt[q * PLANE + row * QBS_LD + kqa * 2]) = w;
        }
      } else if constexpr (P < 2 * NPA + NB) {
        constexpr int i = P - 2 * NPA;
        const int row = (tid + 512 * i) >> 2;
        asm volatile("" : "+v"(rb[i]));
        *reinterpret_cast<i32x4*>(&nxt[NS * PLANE + row * QBS_LD + kqb * 2]) = rb[i];
      } else if constexpr (P < 2 * NPA + 2 * NB) {
        constexpr int i = P - 2 * NPA - NB;
        rb[i] = *reinterpret_cast<const i32x4*>(pb[i] + ((kb2 + kqb) < K ? kb2 : 0));
      } else {
        constexpr int w = P - 2 * NPA - 2 * NB;
        if constexpr (w == 0) {
          rka[sl] = (ka3 + kqa) < K;
          rks[sl] = *reinterpret_cast<const f32x4v*>(p.s ? p.s + (rka[sl] ? ka3 : 0) + kqa : pa[0]);
        } else {
          ra[sl][w - 1] = *reinterpret_cast<const f32x4v*>(pa[w - 1] + (rka[sl] ? ka3 : 0));
        }
      }
    };
    static_for<NM>([&](auto G_) {
      constexpr int G = decltype(G_)::value;
      constexpr int ks = G / (2 * NS * NJ), q = (G / (2 * NJ)) % NS, i = (G / NJ) % 2, j = G % NJ;
      acc[i][j] = mfma_16b<F16>(av[q][i], bv[ks][j], acc[i][j]);
      if constexpr (ks == 0) {
        if constexpr (G < NJ) bv[1][G] = *reinterpret_cast<const bf16x8*>(b + G * 32 * QBS_LD + 32);
        if constexpr (j == NJ - 1) av[q][i] = *reinterpret_cast<const bf16x8*>(a + q * PLANE + i * 32 * QBS_LD + 32);
      }
      constexpr int P0 = G * NP / NM, P1 = (G + 1) * NP / NM;
#ifndef NTW_X_NO_STAGING
      static_for<P1 - P0>([&](auto D_) { piece(std::integral_constant<int, P0 + decltype(D_)::value>{}); });
#endif
      __builtin_amdgcn_sched_barrier(0);
    });
#ifndef NTW_X_NO_BARRIER
    lds_barrier();
#endif
  };
#endif
  gload(0, Slot0());
  gload_b(0);
  gload(min(1, klast), Slot1());
  lstore(smem, Slot0());
  gload_b(min(1, klast));
  gload(min(2, klast), Slot0());
  lds_barrier();
  {
    int kt = 0;
    for (; kt + 1 < nkt; kt += 2) {
      step(kt, smem, smem + STAGE, Slot1());
      step(kt + 1, smem + STAGE, smem, Slot0());
    }
    if (kt < nkt) step(kt, smem, smem + STAGE, Slot1());
  }

  if constexpr (!LSQ) {
    const float alpha_e = p.alpha * inv_sE;
    // optional per-column scale and bias (the W8A8 stem's forward: C = cs[n] * (A . codes^T) + bias[n]; p.cs / p.bias)
    float csv[NJ], cbv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int nc = min(n0 + wn * 32 * NJ + j * 32 + l31, p.N - 1);
      csv[j] = p.cs ? p.cs[nc] * alpha_e : alpha_e;
      cbv[j] = p.bias ? p.bias[nc] : 0.f;
    }
    // interior tiles (every tile of the DeiT shapes): uniform tile base + one 32-bit lane offset per access, no
    // per-element bounds checks (each one is an exec-mask branch around a single store)
    const bool interior = (m0 + BM <= p.M) && (n0 + BN <= p.N) && (int64_t)BM * p.ldc < (1ll << 28);
    if (interior) {
      float* Cs = p.C + (int64_t)m0 * p.ldc + n0;
      const int ldc = (int)p.ldc;
      const int nl0 = wn * 32 * NJ + l31;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int eb = 0; eb < 4; ++eb) {
          const int mlb = (wm * 64 + i * 32 + 8 * eb + 4 * lh) * ldc + nl0;
          float old[4][NJ];
          if (p.accumulate) {
#pragma unroll
            for (int ee = 0; ee < 4; ++ee)
#pragma unroll
              for (int j = 0; j < NJ; ++j) old[ee][j] = Cs[mlb + ee * ldc + j * 32];
          }
#pragma unroll
          for (int ee = 0; ee < 4; ++ee)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
              const float v = acc[i][j][eb * 4 + ee] * csv[j] + cbv[j];
              Cs[mlb + ee * ldc + j * 32] = p.accumulate ? v + old[ee][j] : v;
            }
        }
    } else if (!p.accumulate) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int n = n0 + wn * 32 * NJ + j * 32 + l31;
        if (n >= p.N) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
            if (m < p.M) p.C[(int64_t)m * p.ldc + n] = acc[i][j][e] * csv[j] + cbv[j];
          }
      }
    } else {
      // C += ...: the old values are fetched unconditionally (clamped addresses), a quad of rows at a time, so that no
      // load sits behind a per-element condition (that costs one memory round trip per element)
      int ncc[NJ];
      bool nok[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int n = n0 + wn * 32 * NJ + j * 32 + l31;
        nok[j] = n < p.N;
        ncc[j] = min(n, p.N - 1);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int eb = 0; eb < 4; ++eb) {
          float old[4][NJ];
#pragma unroll
          for (int ee = 0; ee < 4; ++ee) {
            const int mc = min(m0 + wm * 64 + i * 32 + ee + 8 * eb + 4 * lh, p.M - 1);
#pragma unroll
            for (int j = 0; j < NJ; ++j) old[ee][j] = p.C[(int64_t)mc * p.ldc + ncc[j]];
          }
#pragma unroll
          for (int ee = 0; ee < 4; ++ee) {
            const int m = m0 + wm * 64 + i * 32 + ee + 8 * eb + 4 * lh;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
              if (m < p.M && nok[j]) p.C[(int64_t)m * p.ldc + ncc[j]] = (acc[i][j][eb * 4 + ee] * csv[j] + cbv[j]) + old[ee][j];
          }
        }
    }
  } else {
    float* fsm = reinterpret_cast<float*>(smem);       // the k-loop's last barrier has released the staging buffers
    float* row_a = fsm;                                // [128]  effective LSQ step of the tile rows
    float* rowred = fsm + BM;                          // [4][128] step-gradient partials per column-wave
    float* colred = fsm + 5 * BM;                      // [2][2][BN] offset-gradient partials per row-wave
    if (tid < BM) row_a[tid] = ofq_lsq_eff_scale(p.ls[min(m0 + tid, p.M - 1) % p.lS], p.lgscale);
    __syncthreads();
    const float alpha_l = p.alpha * inv_sE;             // (F16: back from the launch's power-of-two scale)
    int ncol[NJ];
    bool nok[NJ];
    float b4v[NJ], cb4[NJ], cg[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      ncol[j] = n0 + wn * 32 * NJ + j * 32 + l31;
      nok[j] = ncol[j] < p.N;
      b4v[j] = (nok[j] && p.lb4) ? p.lb4[ncol[j]] : 0.f;
      cb4[j] = cg[j] = 0.f;
    }
    int ncc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) ncc[j] = min(ncol[j], p.N - 1);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int eb = 0; eb < 4; ++eb) {
        // the 4 x NJ inputs of this row quad are loaded unconditionally (clamped addresses) before any of them is used:
        // a load behind a per-element condition would cost one memory round trip per element
        float xv[4][NJ];
#pragma unroll
        for (int ee = 0; ee < 4; ++ee) {
          const int mc = min(m0 + wm * 64 + i * 32 + ee + 8 * eb + 4 * lh, p.M - 1);
#pragma unroll
          for (int j = 0; j < NJ; ++j) xv[ee][j] = p.lx[(int64_t)mc * p.ldlx + ncc[j]];
        }
#pragma unroll
        for (int ee = 0; ee < 4; ++ee) {
          const int e = eb * 4 + ee;
          const int ml = wm * 64 + i * 32 + ee + 8 * eb + 4 * lh;
          const int m = m0 + ml;
          const bool mok = m < p.M;
          const float al = row_a[ml];
          float rds = 0.f;
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const bool ok = mok && nok[j];
            const float ge = ok ? acc[i][j][e] * alpha_l : 0.f;
            const float xin = xv[ee][j];
            const float xe = p.lgelu ? ofq_gelu(xin) : xin;
            float q, v;
            ofq_lsq_quant(__fadd_rn(xe, b4v[j]), al, p.llo, p.lhi, q, v);
            const bool inr = (v >= p.llo) && (v <= p.lhi);
            const float dq = inr ? ofq_div(__fmul_rn(ge, al), al) : 0.f;       // autograd order: (g*a)/a
            rds += ge * (inr ? (q - v) : q);
            cb4[j] += dq;
            cg[j] += ge;
            if (ok) p.C[(int64_t)m * p.ldc + ncol[j]] = p.lgelu ? dq * ofq_gelu_grad(xin) : dq;
          }
#pragma unroll
          for (int o = 16; o > 0; o >>= 1) rds += __shfl_xor(rds, o, 64);      // the 32 lanes of this half-wave share the row
          if (l31 == 0) rowred[wn * BM + ml] = rds;
        }
      }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      cb4[j] += __shfl_xor(cb4[j], 32, 64);
      cg[j] += __shfl_xor(cg[j], 32, 64);
      if (lh == 0) {
        const int nl = wn * 32 * NJ + j * 32 + l31;
        colred[(wm * 2 + 0) * BN + nl] = cb4[j];
        colred[(wm * 2 + 1) * BN + nl] = cg[j];
      }
    }
    __syncthreads();
    if (tid < BM && m0 + tid < p.M)
      p.lrow[(int64_t)(m0 + tid) * p.tiles_n + tn] = (rowred[tid] + rowred[BM + tid]) + (rowred[2 * BM + tid] + rowred[3 * BM + tid]);
    for (int idx = tid; idx < 2 * BN; idx += 512) {
      const int ac = idx / BN, nl = idx - ac * BN;
      if (n0 + nl < p.N)
        p.lcol[((int64_t)tm * 2 + ac) * p.N + n0 + nl] = colred[ac * BN + nl] + colred[(2 + ac) * BN + nl];
    }
  }
}

extern "C" int ofq_qgemm_bf16s_nt(const float* A, const void* B_bf16, float* C, const float* k_scale, float alpha,
                                  int accumulate, int nsplit, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                                  int64_t ldc, const void* amax, const float* col_scale, const float* col_bias,
                                  ofq_stream_t stream) {
  if (!A || !B_bf16 || !C || M <= 0 || N <= 0 || K <= 0) return OFQ_EINVAL;
  if ((K & 7) || (lda & 3) || (ldb & 7) || !al16(A) || !al16(B_bf16) || (k_scale && !al16(k_scale)) || M >= (1ll << 30) ||
      N >= (1ll << 30) || (nsplit != 2 && nsplit != 3) || (amax && nsplit != 2))
    return OFQ_EINVAL;
  QGemmArgs a = {};
  a.A = A; a.B = B_bf16; a.C = C; a.s = k_scale; a.amax = (const unsigned*)amax; a.cs = col_scale; a.bias = col_bias;
  a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.M = (int)M; a.N = (int)N; a.K = (int)K;
  a.tiles_m = (int)ceil_div(M, 128); a.tiles_n = (int)ceil_div(N, 128); a.alpha = alpha; a.accumulate = accumulate; a.nb1 = 1;
  const int nj = N > 256 ? 3 : 2;
  // few rows (late Swin stages): 128-column tiles of the 4-wave kernel give 2-3x more workgroups, which matters more than
  // the shared split (measured: 185 vs 207 us at M=6272, N=768, K=3072; an 8-wave 128x128 variant lost to it as well)
  const bool too_few = (int64_t)a.tiles_m * ceil_div(N, 128 * nj) < 160 && (int64_t)a.tiles_m * a.tiles_n >= 192 && !col_scale && !col_bias;
  if ((nsplit == 3 || amax) && N > 128 && !too_few) {      // wide tiles: the dY panel is split once per 384 (256) columns
    a.tiles_n = (int)ceil_div(N, 128 * nj);
    dim3 gridw((unsigned)(a.tiles_m * a.tiles_n));
    if (amax) {
      if (nj == 3) hipLaunchKernelGGL((qgemm_bf16s_nt_wide_kernel<3, false, true>), gridw, dim3(512), 0, (hipStream_t)stream, a);
      else hipLaunchKernelGGL((qgemm_bf16s_nt_wide_kernel<2, false, true>), gridw, dim3(512), 0, (hipStream_t)stream, a);
    } else {
      if (nj == 3) hipLaunchKernelGGL((qgemm_bf16s_nt_wide_kernel<3, false>), gridw, dim3(512), 0, (hipStream_t)stream, a);
      else hipLaunchKernelGGL((qgemm_bf16s_nt_wide_kernel<2, false>), gridw, dim3(512), 0, (hipStream_t)stream, a);
    }
    OFQ_LAUNCH_CHECK();
    return 0;
  }
  if (col_scale || col_bias) return OFQ_EINVAL;      // the column epilogue exists in the wide kernels only (N > 128)
  dim3 grid((unsigned)(a.tiles_m * a.tiles_n));
  if (amax) hipLaunchKernelGGL((qgemm_bf16s_nt_kernel<2, false, 1, 5, true>), grid, dim3(256), 0, (hipStream_t)stream, a);
  else if (nsplit == 3) hipLaunchKernelGGL((qgemm_bf16s_nt_kernel<3, false>), grid, dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((qgemm_bf16s_nt_kernel<2, false>), grid, dim3(256), 0, (hipStream_t)stream, a);
  OFQ_LAUNCH_CHECK();
  return 0;
}
