// ------------------------------------------------------------------------------------------------ int8 forward
#define QI8_BK 64                 // bytes of k per LDS stage (two 32x32x32 MFMA steps)
#define QI8_LD (QI8_BK + 16)      // padded LDS row (bytes)

// workgroup barrier that orders LDS traffic only: global prefetch loads stay in flight across it (__syncthreads would
// drain vmcnt and expose the HBM latency once per k-step)
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// Epilogue of one interior 128x128 tile of the linear-layer GEMM: y = cs[n] * (a_eff[m] * I + r[n]) + bias[n] stored as
// fp32, plus (optionally) the next quantiser's int8 levels into the LDS code tile.  Straight-line code specialised on the
// by-product mode: no per-element bounds / mode branches, row pointers from scalar arithmetic (the lane adds one 32-bit
// offset), LDS addresses as immediates, the level through ofq_lsq_level_rcp.  This epilogue is VALU-bound, so
// instructions per element are what counts.  row_a: [3][128] floats in LDS (effective input step, the by-product's
// per-row step and its reciprocal), ctile: [128][128] bytes in LDS.
__device__ __forceinline__ void i8_epi0_interior_tile(const QGemmArgs& p, const i32x16 (&acc)[2][2], float* Cb, const float* row_a,
                                                      signed char* ctile, const float (&csn)[2], const float (&rn)[2],
                                                      const float (&bz)[2], const float (&qb)[2], const float (&qsc)[2], int m0,
                                                      int n0, int wm, int wn, int l31, int lh) {
  constexpr int BM = 128, BN = 128;
    const int wm_s = __builtin_amdgcn_readfirstlane(wm), wn_s = __builtin_amdgcn_readfirstlane(wn);
    const float* row_c = row_a + 2 * BM;
    float qrc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) qrc[j] = (p.qout && p.qcolmode) ? __fdiv_rn(1.f, qsc[j]) : 1.f;
    const unsigned lane_off4 = 4u * ((unsigned)(4 * lh) * (unsigned)p.ldc + (unsigned)(n0 + wn_s * 64 + l31));
    float* Cb_t = Cb + (int64_t)(m0 + wm_s * 64) * p.ldc;
    signed char* ct = ctile + (wm_s * 64 + 4 * lh) * BN + wn_s * 64 + l31;
    const float* ra_t = row_a + wm_s * 64 + 4 * lh;
    const float qlo = p.qlo, qhi = p.qhi;
    const float half_m_tol = 0.5f - ofq_lsq_level_tol(qlo, qhi);
    auto tile = [&](auto QMODE_, auto QGELU_, auto STORE_) {
      constexpr int QMODE = decltype(QMODE_)::value;          // 0 none, 1 per-row step, 2 per-column step
      constexpr bool QGELU = decltype(QGELU_)::value;
      constexpr bool STORE_Y = decltype(STORE_)::value;       // false: only the by-product codes leave the kernel
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int eg = 0; eg < 4; ++eg) {                      // 4 rows x 2 columns per lane share one exactness check
          // QGELU: the level is decided on ofq_gelu_fast (common.h); its error bound, in level units, widens the
          // half-integer margin of the group, and a flagged group is redone with erff and the IEEE division
          float yq[4][2], xq[4][2], qv[4][2], rbv[4];
          float dmax = 0.f, rmax = 0.f;
#pragma unroll
          for (int ee = 0; ee < 4; ++ee) {
            const int e = eg * 4 + ee;
            const int r = i * 32 + ee + 8 * eg;
            const float ae = ra_t[r];
            rbv[ee] = QMODE == 1 ? ra_t[BM + r] : 1.f;
            const float rrb = QMODE == 1 ? ra_t[2 * BM + r] : 1.f;
            if (QGELU && QMODE == 1) rmax = fmaxf(rmax, rrb);
            float* rowp = Cb_t + (int64_t)r * p.ldc;           // uniform: lives in an SGPR pair
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const float yv = __fadd_rn(__fmul_rn(csn[j], __fadd_rn(__fmul_rn(ae, (float)acc[i][j][e]), rn[j])), bz[j]);
              // scalar row base + 32-bit lane offset + immediate: no per-element address arithmetic on the VALU
#ifndef I8X_NO_F32_STORE
              if (STORE_Y) {
                if (j == 0) asm volatile("global_store_dword %0, %1, %2" ::"v"(lane_off4), "v"(yv), "s"(rowp));
                else asm volatile("global_store_dword %0, %1, %2 offset:128" ::"v"(lane_off4), "v"(yv), "s"(rowp));
              }
#else
              if (yv == 123.456f) asm volatile("global_store_dword %0, %1, %2" ::"v"(lane_off4), "v"(yv), "s"(rowp));
#endif
              if (QMODE != 0) {
                if (QGELU) {
                  yq[ee][j] = yv;
                  qv[ee][j] = ofq_lsq_level_rcp_d(__fadd_rn(ofq_gelu_fast(yv), qb[j]), QMODE == 2 ? qrc[j] : rrb, qlo, qhi, dmax);
                } else {
                  xq[ee][j] = __fadd_rn(yv, qb[j]);
                  qv[ee][j] = ofq_lsq_level_rcp_d(xq[ee][j], QMODE == 2 ? qrc[j] : rrb, qlo, qhi, dmax);
                }
              }
            }
          }
          if (QMODE != 0) {
            float thr = half_m_tol;
            if (QGELU) thr = __builtin_fmaf(-OFQ_GELU_FAST_EPS, QMODE == 2 ? fmaxf(qrc[0], qrc[1]) : rmax, half_m_tol);
            if (__builtin_amdgcn_ballot_w64(!(dmax < thr)) != 0ull) {
#pragma unroll
              for (int ee = 0; ee < 4; ++ee)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                  qv[ee][j] = ofq_lsq_level_exact(QGELU ? __fadd_rn(ofq_gelu(yq[ee][j]), qb[j]) : xq[ee][j],
                                                  QMODE == 2 ? qsc[j] : rbv[ee], qlo, qhi);
            }
#pragma unroll
            for (int ee = 0; ee < 4; ++ee)
#pragma unroll
#ifndef I8X_NO_CODE_LDS
              for (int j = 0; j < 2; ++j) ct[(i * 32 + ee + 8 * eg) * BN + j * 32] = (signed char)(int)qv[ee][j];
#else
              for (int j = 0; j < 2; ++j) if (qv[ee][j] == 77.f) ct[(i * 32 + ee + 8 * eg) * BN + j * 32] = (signed char)1;
#endif
          }
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using Y = std::true_type;
    using N = std::false_type;
    if (!p.qout) tile(I0(), N(), Y());
    else if (p.C) {
      if (p.qcolmode) { if (p.qgelu) tile(I2(), Y(), Y()); else tile(I2(), N(), Y()); }
      else { if (p.qgelu) tile(I1(), Y(), Y()); else tile(I1(), N(), Y()); }
    } else {
      if (p.qcolmode) { if (p.qgelu) tile(I2(), Y(), N()); else tile(I2(), N(), N()); }
      else { if (p.qgelu) tile(I1(), Y(), N()); else tile(I1(), N(), N()); }
    }
}

#ifdef I8X_TIMING
__device__ unsigned long long g_i8_dbg[8];          // phase timestamps of one mid-grid workgroup (tools/probe/i8_probe.hip)
#define I8_T(slot) do { if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) g_i8_dbg[slot] = __builtin_readcyclecounter(); } while (0)
#else
#define I8_T(slot) do {} while (0)
#endif

#ifndef I8_WPE
#define I8_WPE 3                   // waves per SIMD the int8 kernel is compiled for (tools/probe/i8_probe.hip overrides it)
#endif
// The k-loop of the int8 kernels: acc[2][2] (2x2 blocks of 32x32 per wave, 2x2 waves) += A[m0.., :] . B[n0.., :]^T over K.
// Shared by the forward kernel and by the backward kernel that recomputes a layer output from the codes.
// T = 2: 128 x 128 tile (each of the 2 x 2 waves owns 64 x 64 = 2 x 2 MFMA blocks); T = 1: 64 x 64 tile (one 32 x 32 block per
// wave) for the 49-token Swin windows, where a 128 x 128 tile is 85 % padding.
// General form: the workgroup tile is (64 CA) x (64 CB) (CA / CB = 16-byte chunks per thread and k-step of the A / B operand),
// its four waves are arranged WGM x (4 / WGM), each owning MI x NJ MFMA blocks of 32 x 32.
template <int CA, int CB, int WGM, int MI, int NJ>
__device__ __forceinline__ void i8_mainloop_g(const QGemmArgs& p, const unsigned char* A, const unsigned char* B, int m0, int n0,
                                              unsigned char (*smem)[(64 * CA + 64 * CB) * QI8_LD], i32x16 (&acc)[MI][NJ]) {
  constexpr int BM = 64 * CA, BN = 64 * CB, WGN = 4 / WGM;
  static_assert(WGM * 32 * MI == BM && WGN * 32 * NJ == BN, "wave layout does not tile the workgroup tile");
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WGN, wn = wid % WGN;
  const int l31 = lane & 31, lh = lane >> 5;
  const int K = p.K;
  const int nkt = (K + QI8_BK - 1) / QI8_BK;

  // staging: rows x 64 B per operand = 4 * rows x 16 B -> CA / CB chunks per thread (chunk i of a thread: row (tid + 256 i) / 4,
  // the same 16-byte column kq for every i)
  int64_t offA[CA], offB[CB];
  bool okA[CA], okB[CB];
  const int kq = (tid & 3) * 16;
#pragma unroll
  for (int i = 0; i < CA; ++i) {
    const int row = (tid + 256 * i) >> 2;
    okA[i] = (m0 + row) < p.M;
    offA[i] = (int64_t)min(m0 + row, p.M - 1) * p.lda + kq;
  }
#pragma unroll
  for (int i = 0; i < CB; ++i) {
    const int row = (tid + 256 * i) >> 2;
    okB[i] = (n0 + row) < p.N;
    offB[i] = (int64_t)min(n0 + row, p.N - 1) * p.ldb + kq;
  }
  // Two register slots: the loads of tile kt+3 are issued behind the staging of tile kt+1 and are first touched (masked)
  // two k-steps later, so a k-step never waits for the HBM / L2 latency of its own loads (k-steps are only 8 MFMAs
  // long here).  The loop body is branch-free (tiles past the end repeat the last one into a stage nobody reads): guards
  // around the loads make the compiler's wait-count pass wait for every outstanding load at each k-step.
  i32x4 ra[2][CA], rb[2][CB];
  const int klast = nkt - 1;
  const bool nomask = (m0 + BM <= p.M) && (n0 + BN <= p.N) && (K % QI8_BK) == 0;
  auto gload = [&](int kt, auto SLOT) {
    constexpr int sl = decltype(SLOT)::value;
    const int k0 = kt * QI8_BK;
    const bool kin = (k0 + kq) < K;                      // K % 16 == 0 (host check): a chunk is all in or all out
#pragma unroll
    for (int i = 0; i < CA; ++i) ra[sl][i] = *reinterpret_cast<const i32x4*>(A + offA[i] + (kin ? k0 : -kq));
#pragma unroll
    for (int i = 0; i < CB; ++i) rb[sl][i] = *reinterpret_cast<const i32x4*>(B + offB[i] + (kin ? k0 : -kq));
  };
  auto lstore = [&](unsigned char* sb, int kt, auto SLOT) {
    constexpr int sl = decltype(SLOT)::value;
#pragma unroll
    for (int i = 0; i < CA; ++i) asm volatile("" : "+v"(ra[sl][i]));
#pragma unroll
    for (int i = 0; i < CB; ++i) asm volatile("" : "+v"(rb[sl][i]));
    const int k0 = kt * QI8_BK;
    if (nomask) {      // interior tile, no k tail (every tile of the DeiT-S shapes): 16 v_and + the mask selects per k-step gone
#pragma unroll
      for (int i = 0; i < CA; ++i) *reinterpret_cast<i32x4*>(&sb[((tid + 256 * i) >> 2) * QI8_LD + kq]) = ra[sl][i];
#pragma unroll
      for (int i = 0; i < CB; ++i) *reinterpret_cast<i32x4*>(&sb[(BM + ((tid + 256 * i) >> 2)) * QI8_LD + kq]) = rb[sl][i];
    } else {
      const bool kin = (k0 + kq) < K;
#pragma unroll
      for (int i = 0; i < CA; ++i)
        *reinterpret_cast<i32x4*>(&sb[((tid + 256 * i) >> 2) * QI8_LD + kq]) = ra[sl][i] & ((okA[i] && kin) ? -1 : 0);
#pragma unroll
      for (int i = 0; i < CB; ++i)
        *reinterpret_cast<i32x4*>(&sb[(BM + ((tid + 256 * i) >> 2)) * QI8_LD + kq]) = rb[sl][i] & ((okB[i] && kin) ? -1 : 0);
    }
  };

#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0;

  auto compute = [&](const unsigned char* sb) {
    const unsigned char* a = &sb[(wm * 32 * MI + l31) * QI8_LD + lh * 16];
    const unsigned char* b = &sb[(BM + wn * 32 * NJ + l31) * QI8_LD + lh * 16];
#pragma unroll
    for (int ks = 0; ks < QI8_BK / 32; ++ks) {
      i32x4 av[MI], bv[NJ];
#pragma unroll
      for (int i = 0; i < MI; ++i) av[i] = *reinterpret_cast<const i32x4*>(a + i * 32 * QI8_LD + ks * 32);
#pragma unroll
      for (int j = 0; j < NJ; ++j) bv[j] = *reinterpret_cast<const i32x4*>(b + j * 32 * QI8_LD + ks * 32);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  using Slot0 = std::integral_constant<int, 0>;
  using Slot1 = std::integral_constant<int, 1>;
  auto step = [&](int kt, const unsigned char* cur, unsigned char* nxt, auto SLOT) {
    compute(cur);
    lstore(nxt, min(kt + 1, klast), SLOT);
    gload(min(kt + 3, klast), SLOT);
    lds_barrier();
  };
  gload(0, Slot0());
  gload(min(1, klast), Slot1());
  lstore(smem[0], 0, Slot0());
  gload(min(2, klast), Slot0());
  lds_barrier();
  I8_T(1);
#ifndef I8X_NO_KLOOP
  {
    int kt = 0;
    for (; kt + 1 < nkt; kt += 2) {
      step(kt, smem[0], smem[1], Slot1());
      step(kt + 1, smem[1], smem[0], Slot0());
    }
    if (kt < nkt) step(kt, smem[0], smem[1], Slot1());
  }
#endif
}

// square form used by the GEMM kernels: 64 T x 64 T tile, 2 x 2 waves of T x T blocks
template <int T>
__device__ __forceinline__ void i8_mainloop(const QGemmArgs& p, const unsigned char* A, const unsigned char* B, int m0, int n0,
                                            unsigned char (*smem)[(128 * T) * QI8_LD], i32x16 (&acc)[T][T]) {
  i8_mainloop_g<T, T, 2, T, T>(p, A, B, m0, n0, smem, acc);
}

// EPI 0: linear layer   1: QKR attention scores   2: P*V
template <int EPI, int T = 2>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(I8_WPE, I8_WPE))) void qgemm_i8_nt_kernel(QGemmArgs p) {
  static_assert(T == 2 || EPI != 0, "the linear-layer epilogue is written for 128 x 128 tiles");
  // T = 3 ("tall"): 256 x 64 tile, the four waves stacked (each 64 x 64) -- P.V, whose output has one head's 64 channels:
  // a 128 x 128 tile there leaves two of the four waves without columns and takes two workgroups per (batch, head)
  constexpr bool TALL = T == 3;
  constexpr int TT = TALL ? 2 : T;                       // 32 x 32 blocks per wave and direction
  constexpr int BM = TALL ? 256 : 64 * T, BN = TALL ? 64 : 64 * T;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2][(BM + BN) * QI8_LD];
  I8_T(0);
  int tm, tn, gby;
  qgemm_tile_id(p, tm, tn, gby);
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = TALL ? wid : wid >> 1, wn = TALL ? 0 : wid & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  const int b0 = gby / p.nb1, b1 = gby % p.nb1;
  const unsigned char* A = (const unsigned char*)p.A + b0 * p.sA0 + b1 * p.sA1;
  const unsigned char* B = (const unsigned char*)p.B + b0 * p.sB0 + b1 * p.sB1;
  const int K = p.K;
  // Epilogue parameters (per-row steps / offsets, per-column scales) are requested before the k-loop: a workgroup lives
  // for one 128x128 tile only, and every dependent round trip to memory after the loop (row terms -> barrier -> column
  // terms) is paid in full ~14 times per CU.  All loads are unconditional on clamped indices (a load under a condition
  // ends in a register copy that waits for it); optional vectors fall back to a valid address and are ignored later.
  float pre_ra, pre_rb = 0.f, pre_c[TT][5];
  {
    const int m = min(m0 + (tid & (BM - 1)), p.M - 1);
    pre_ra = p.s[m % p.S];
    if (EPI == 0) {
      const float* qsp = (p.qout && !p.qcolmode) ? p.qs + ((int64_t)m * p.qrowmul + n0 / p.qcoldiv) % p.qS : p.s;
      pre_rb = *qsp;
    }
    if (EPI == 1) pre_rb = p.u[((int64_t)b0 * p.M + m) * p.nb1 + b1];
    if (EPI == 2) pre_rb = p.rp[((int64_t)b0 * p.nb1 + b1) * p.M + m];
    if (EPI == 0) {
#pragma unroll
      for (int j = 0; j < TT; ++j) {
        const int nc = min(n0 + wn * 32 * TT + j * 32 + l31, p.N - 1);
        pre_c[j][0] = p.cs[nc];
        pre_c[j][1] = (p.r ? p.r : p.cs)[nc];
        pre_c[j][2] = (p.bias ? p.bias : p.cs)[nc];
        pre_c[j][3] = ((p.qout && p.qb4) ? p.qb4 : p.cs)[nc];
        pre_c[j][4] = ((p.qout && p.qcolmode) ? p.qs : p.cs)[nc];
      }
    }
  }

  i32x16 acc[TT][TT];
  if constexpr (TALL) i8_mainloop_g<4, 1, 4, 2, 2>(p, A, B, m0, n0, smem, acc);
  else i8_mainloop<T>(p, A, B, m0, n0, smem, acc);
  I8_T(2);
  float* Cb = p.C + b0 * p.sC0 + b1 * p.sC1;
  int ncol[TT];
#pragma unroll
  for (int j = 0; j < TT; ++j) ncol[j] = n0 + wn * 32 * TT + j * 32 + l31;
  // per-row epilogue terms of the 128 tile rows go through LDS once (the k-loop's last barrier has released smem):
  // row_a = effective LSQ step of the row, row_b = the row's offset term (u / rp); every lane then reads 32 of them as
  // broadcasts instead of issuing 32 dependent global loads + integer modulos
  float* row_a = reinterpret_cast<float*>(&smem[0][0]);
  float* row_b = row_a + BM;
  if (tid < BM) {
    row_a[tid] = ofq_lsq_eff_scale(pre_ra, p.gscale);
    if (EPI == 0 && p.qout && !p.qcolmode) {
      const float rbv = ofq_lsq_eff_scale(pre_rb, p.qgscale);
      row_b[tid] = rbv;
      row_a[2 * BM + tid] = __fdiv_rn(1.f, rbv);                 // row_c: reciprocal steps for the fast level path
    }
    if (EPI == 1 || EPI == 2) row_b[tid] = pre_rb;
  }
  __syncthreads();
  I8_T(3);
  if constexpr (EPI == 0) {
    // y = cs[n] * (a_eff[m % S] * I + r[n]) + bias[n]
    float csn[2], rn[2], bz[2], qb[2], qsc[2];
    signed char* ctile = reinterpret_cast<signed char*>(&smem[0][0]) + 2048;      // [128][128] codes, behind row_a / row_b
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      csn[j] = pre_c[j][0] * p.alpha;
      rn[j] = p.r ? pre_c[j][1] : 0.f;
      bz[j] = p.bias ? pre_c[j][2] : 0.f;
      qb[j] = (p.qout && p.qb4) ? pre_c[j][3] : 0.f;
      qsc[j] = (p.qout && p.qcolmode) ? ofq_lsq_eff_scale(pre_c[j][4], p.qgscale) : 1.f;
    }
    // Interior tiles (every tile of the DeiT-S shapes) take a straight-line epilogue specialised on the by-product mode:
    // no per-element bounds / mode branches, row pointers from scalar arithmetic (the lane adds one 32-bit offset),
    // LDS addresses as immediates, and the level through ofq_lsq_level_rcp.  This epilogue is VALU-bound (the qkx
    // GEMM spent ~70 of its 119 us in it), so instructions per element are what counts.
    const bool interior = (m0 + BM <= p.M) && (n0 + BN <= p.N) && 4 * (4 * p.ldc + p.N) < (int64_t)0x7fffffff;
    if (interior) {
      i8_epi0_interior_tile(p, acc, Cb, row_a, ctile, csn, rn, bz, qb, qsc, m0, n0, wm, wn, l31, lh);
    } else {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (m >= p.M) continue;
        const float ae = row_a[m - m0];
#pragma unroll
        for (int j = 0; j < 2; ++j)
          if (ncol[j] < p.N) {
            const float yv = __fadd_rn(__fmul_rn(csn[j], __fadd_rn(__fmul_rn(ae, (float)acc[i][j][e]), rn[j])), bz[j]);
            if (p.C) Cb[(int64_t)m * p.ldc + ncol[j]] = yv;
            if (p.qout) {
              const float xe = p.qgelu ? ofq_gelu(yv) : yv;
              float q, v;
              ofq_lsq_quant(__fadd_rn(xe, qb[j]), p.qcolmode ? qsc[j] : row_b[m - m0], p.qlo, p.qhi, q, v);
              ctile[(m - m0) * BN + (ncol[j] - n0)] = (signed char)(int)q;
            }
          }
      }
    }
    I8_T(4);
    if (p.qout) {       // the code tile goes out in 64-byte row pieces (two threads per row) instead of single bytes
      __syncthreads();
      const int row = tid >> 1, c0 = (tid & 1) * 64;
      if (m0 + row < p.M) {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const int n = n0 + c0 + 16 * q4;
          if (n < p.N)       // N % 16 == 0 (host check)
            *reinterpret_cast<i32x4*>(p.qout + (int64_t)(m0 + row) * p.ldq + n) =
                *reinterpret_cast<const i32x4*>(ctile + row * BN + c0 + 16 * q4);
        }
      }
    }
    I8_T(5);
  } else if constexpr (EPI == 1) {
    // S[n,m] = ax[n] * (aq[m,h] * I + u[b,n,h]) + aq[m,h] * tq[b,m,h] + z[h]      (x_hat . qkx_hat^T, attention.py:210)
    float aq[TT], tqa[TT];
    const float zz = p.z[b1];
#pragma unroll
    for (int j = 0; j < TT; ++j) {
      const int nc = min(ncol[j], p.N - 1);
      aq[j] = ofq_lsq_eff_scale(p.s2[nc * p.s2s0 + b1 * p.s2s1], p.gscale2);
      tqa[j] = __fadd_rn(__fmul_rn(aq[j], p.tq[((int64_t)b0 * p.N + nc) * p.nb1 + b1]), zz);
    }
#pragma unroll
    for (int i = 0; i < TT; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * 32 * TT + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (m >= p.M) continue;
        const float ax = row_a[m - m0];
        const float uu = row_b[m - m0];
#pragma unroll
        for (int j = 0; j < TT; ++j)
          if (ncol[j] < p.N)
            Cb[(int64_t)m * p.ldc + ncol[j]] =
                __fadd_rn(__fmul_rn(ax, __fadd_rn(__fmul_rn(aq[j], (float)acc[i][j][e]), uu)), tqa[j]);
      }
  } else {
    // O[n,c] = ap[n] * (av[c] * I + bav[c] * rp[n])                                 (P_hat . V_hat, attention.py:219)
    float av[TT], bv2[TT];
#pragma unroll
    for (int j = 0; j < TT; ++j) {
      const int nc = min(ncol[j], p.N - 1) + b1 * p.N;
      av[j] = ofq_lsq_eff_scale(p.s2[nc], p.gscale2);
      bv2[j] = p.z ? p.z[nc] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < TT; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * 32 * TT + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (m >= p.M) continue;
        const float ap = row_a[m - m0];
        const float rpm = row_b[m - m0];
#pragma unroll
        for (int j = 0; j < TT; ++j)
          if (ncol[j] < p.N)
            Cb[(int64_t)m * p.ldc + ncol[j]] =
                __fmul_rn(ap, __fadd_rn(__fmul_rn(av[j], (float)acc[i][j][e]), __fmul_rn(bv2[j], rpm)));
      }
  }
}
