// Wide form of the plane-product GEMM (the fp32 KD teacher's linear layers, ofq_gemm_bf16x3x3_nt): 128 x 384 tile, eight
// waves of 64 x 96, k-steps of 16 through a double-buffered LDS ring (one stage: three planes of A, 128 x 16, and three
// planes of B, 384 x 16: 74 KB), two register prefetch slots.  Per k-step a wave issues 9 (6) plane products x 6 blocks =
// 54 (36) MFMAs against ~75 other instructions -- the split of the activation panel is paid once per 384 columns and
// nine products -- where the 128 x 128 single-buffered kernel ran staging and MFMAs in separate phases on 594 workgroups for
// 512 slots (N = 384).  Same plane order per element as the narrow kernel: sum over k-steps of sum_{q + r < PMAX} A_q B_r.
#define QPW_BK 16
#define QPW_LD (QPW_BK * 2 + 16)       // 48-byte LDS rows: the 16 lanes of a ds_read_b128 pass land on disjoint banks
template <int PMAX>
__global__ __launch_bounds__(512) void gemm_bf16x3x3_wide_kernel(QGemmArgs p) {
  constexpr int BM = 128, BN = 384, NS = 3, NB = 3;
  constexpr int PLANE_A = BM * QPW_LD, PLANE_B = BN * QPW_LD;
  constexpr int STAGE = NS * PLANE_A + NB * PLANE_B;
  constexpr int CHB = NB * BN * 2 / 512 + ((NB * BN * 2) % 512 ? 1 : 0);        // 16-byte B chunks per thread and k-step (5)
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
  const int nkt = (K + QPW_BK - 1) / QPW_BK;
  // A: 128 rows x 16 fp32 = 512 float4: one per thread;  B: 3 planes x 384 rows x 16 bf16 = 2304 16-byte chunks
  const int arow = tid >> 2, akq = (tid & 3) * 4;
  const bool okA = (m0 + arow) < p.M;
  const float* pa = A + (int64_t)min(m0 + arow, p.M - 1) * p.lda + akq;
  const unsigned short* pb[CHB];
  int boff[CHB];
  bool okB[CHB];
#pragma unroll
  for (int i = 0; i < CHB; ++i) {
    const int f = tid + 512 * i;
    const int pl = min(f / (2 * BN), NB - 1), rem = f % (2 * BN), row = rem >> 1, half = rem & 1;
    okB[i] = f < NB * 2 * BN && (n0 + row) < p.N;
    pb[i] = B + (int64_t)pl * p.sBp + (int64_t)min(n0 + row, p.N - 1) * p.ldb + half * 8;
    boff[i] = NS * PLANE_A + pl * PLANE_B + row * QPW_LD + half * 16;
  }
  f32x4v ra[2];
  i32x4 rb[2][CHB];
  bool rka[2], rkb[2][2];
  auto gload = [&](int kt, auto SLOT) {
    constexpr int sl = decltype(SLOT)::value;
    const int k0 = kt * QPW_BK;
    rka[sl] = (k0 + akq) < K;                              // K % 8 == 0 (host check)
    ra[sl] = *reinterpret_cast<const f32x4v*>(pa + (rka[sl] ? k0 : 0));
    rkb[sl][0] = k0 < K;
    rkb[sl][1] = (k0 + 8) < K;
#pragma unroll
    for (int i = 0; i < CHB; ++i) rb[sl][i] = *reinterpret_cast<const i32x4*>(pb[i] + (k0 < K ? k0 : 0));
  };
  auto lstore = [&](unsigned char* sb, auto SLOT) {
    constexpr int sl = decltype(SLOT)::value;
    asm volatile("" : "+v"(ra[sl]));
    const float z = (okA && rka[sl]) ? 1.f : 0.f;
    const f32x2v a01 = {ra[sl][0] * z, ra[sl][1] * z}, a23 = {ra[sl][2] * z, ra[sl][3] * z};
    unsigned lo[NS], hi[NS];
    split_pair_bf16<NS>(a01, lo);
    split_pair_bf16<NS>(a23, hi);
#pragma unroll
    for (int q = 0; q < NS; ++q) {
      uint2 w;
      w.x = lo[q];
      w.y = hi[q];
      *reinterpret_cast<uint2*>(&sb[q * PLANE_A + arow * QPW_LD + akq * 2]) = w;
    }
#pragma unroll
    for (int i = 0; i < CHB; ++i) {
      asm volatile("" : "+v"(rb[sl][i]));
      const int f = tid + 512 * i;
      if (f < NB * 2 * BN) {
        const int m = (okB[i] && rkb[sl][(f % (2 * BN)) & 1]) ? -1 : 0;
        *reinterpret_cast<i32x4*>(&sb[boff[i]]) = rb[sl][i] & m;
      }
    }
  };
  f32x16q acc[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  auto compute = [&](const unsigned char* sb) {
    const unsigned char* a = &sb[(wm * 64 + l31) * QPW_LD + lh * 16];
    const unsigned char* b = &sb[NS * PLANE_A + (wn * 96 + l31) * QPW_LD + lh * 16];
    bf16x8 av[NS][2], bv[NB][3];
#pragma unroll
    for (int q = 0; q < NS; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i) av[q][i] = *reinterpret_cast<const bf16x8*>(a + q * PLANE_A + i * 32 * QPW_LD);
#pragma unroll
    for (int r = 0; r < NB; ++r)
#pragma unroll
      for (int j = 0; j < 3; ++j) bv[r][j] = *reinterpret_cast<const bf16x8*>(b + r * PLANE_B + j * 32 * QPW_LD);
#pragma unroll
    for (int q = 0; q < NS; ++q)
#pragma unroll
      for (int r = 0; r < NB; ++r) {
        if (q + r >= PMAX) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[q][i], bv[r][j], acc[i][j], 0, 0, 0);
      }
    __builtin_amdgcn_sched_barrier(0);
  };
  using Slot0 = std::integral_constant<int, 0>;
  using Slot1 = std::integral_constant<int, 1>;
  const int klast = nkt - 1;
  // loads of tile kt + 3 behind the staging of tile kt + 1, as in the other wide kernels (branch-free: past the end the
  // last tile is loaded again into a stage nobody reads)
#ifdef QPW_SERIAL_STAGING
  auto step = [&](int kt, const unsigned char* cur, unsigned char* nxt, auto SLOT) {
    compute(cur);
    lstore(nxt, SLOT);
    gload(min(kt + 3, klast), SLOT);
    lds_barrier();
  };
#else
  // The staging of tile kt + 1 and the loads of tile kt + 3 cut into pieces of one or two instructions behind the MFMAs
  // of tile kt (see the note at static_for: the first two instructions behind an MFMA of the same wave are free).  Run as
  // a block after the MFMAs, staging + loads + barrier + the first fragment reads left the matrix pipe idle for a third
  // of every k-step (both waves of a SIMD reach that phase together): 283 us at K = 1536 against 185 us of MFMA time.
  // Pieces: the A float4 of this thread (its three planes: per element [x = a*z, p0], [r1, p1], [r2], a pack per pair, three
  // stores), the B chunks (mask + store each), then the loads in consumption order.
  constexpr int NPROD = PMAX >= 5 ? 9 : 6, NM = NPROD * 6;
  constexpr int NPA = 17, NP = NPA + CHB + CHB + 1;
  auto step = [&](int kt, const unsigned char* cur, unsigned char* nxt, auto SLOT) {
    constexpr int sl = decltype(SLOT)::value;
    const unsigned char* a = &cur[(wm * 64 + l31) * QPW_LD + lh * 16];
    const unsigned char* b = &cur[NS * PLANE_A + (wn * 96 + l31) * QPW_LD + lh * 16];
    // A fragments ping-pong between two register pairs (plane q + 1 is read behind the first MFMA of plane q): 16 instead
    // of 24 VGPRs -- the kernel sits at the 256-register limit of two waves per SIMD
    bf16x8 avq[2][2], bv[NB][3];
#pragma unroll
    for (int i = 0; i < 2; ++i) avq[0][i] = *reinterpret_cast<const bf16x8*>(a + i * 32 * QPW_LD);
#pragma unroll
    for (int r = 0; r < NB; ++r)
#pragma unroll
      for (int j = 0; j < 3; ++j) bv[r][j] = *reinterpret_cast<const bf16x8*>(b + r * PLANE_B + j * 32 * QPW_LD);
    float z_ = 0.f, x_ = 0.f, r1_ = 0.f, p0v[2], p1v[2], r2v[2];
    unsigned lo[NS], hi[NS];
    const int k3 = min(kt + 3, klast) * QPW_BK;
    auto piece = [&](auto P_) {
      constexpr int P = decltype(P_)::value;
      if constexpr (P < NPA) {
        if constexpr (P == 0) {
          asm volatile("" : "+v"(ra[sl]));                     // first touch of the slot: the wait for its loads lands here
          z_ = (okA && rka[sl]) ? 1.f : 0.f;
        }
        if constexpr (P < 14) {
          constexpr int pr = P / 7, rr = P % 7;
          if constexpr (rr < 6) {
            constexpr int el = rr / 3, st = rr % 3, e = pr * 2 + el;
            if constexpr (st == 0) valu_mul_hi16(ra[sl][e], z_, x_, p0v[el]);
            if constexpr (st == 1) valu_sub_hi16(x_, p0v[el], r1_, p1v[el]);
            if constexpr (st == 2) { r2v[el] = valu_sub(r1_, p1v[el]); }
          } else {
            valu_pack3_hi16(p0v, p1v, r2v, pr == 0 ? lo : hi);
          }
        } else {
          constexpr int q = P - 14;
          uint2 w;
          w.x = lo[q];
          w.y = hi[q];
          *reinterpret_cast<uint2*>(&nxt[q * PLANE_A + arow * QPW_LD + akq * 2]) = w;
        }
      } else if constexpr (P < NPA + CHB) {
        constexpr int i = P - NPA;
        asm volatile("" : "+v"(rb[sl][i]));
        const int f = tid + 512 * i;
        if (CHB * 512 == NB * 2 * BN || f < NB * 2 * BN) {
          const int m = (okB[i] && rkb[sl][(f % (2 * BN)) & 1]) ? -1 : 0;
          *reinterpret_cast<i32x4*>(&nxt[boff[i]]) = rb[sl][i] & m;
        }
      } else if constexpr (P < NPA + 2 * CHB) {
        constexpr int i = P - NPA - CHB;
        if constexpr (i == 0) {
          rkb[sl][0] = k3 < K;
          rkb[sl][1] = (k3 + 8) < K;
        }
        rb[sl][i] = *reinterpret_cast<const i32x4*>(pb[i] + (k3 < K ? k3 : 0));
      } else {
        rka[sl] = (k3 + akq) < K;
        ra[sl] = *reinterpret_cast<const f32x4v*>(pa + (rka[sl] ? k3 : 0));
      }
    };
    static_for<NM>([&](auto G_) {
      constexpr int G = decltype(G_)::value;
      // product list in the narrow kernel's order: q outer, r inner, skipping q + r >= PMAX
      constexpr int pidx = G / 6, ij = G % 6, i = ij / 3, j = ij % 3;
      constexpr int q = PMAX >= 5 ? pidx / 3 : (pidx < 3 ? 0 : (pidx < 5 ? 1 : 2));
      constexpr int r = PMAX >= 5 ? pidx % 3 : (pidx < 3 ? pidx : (pidx < 5 ? pidx - 3 : 0));
      constexpr int Gq0 = PMAX >= 5 ? q * 18 : (q == 0 ? 0 : (q == 1 ? 18 : 30));       // first MFMA of plane q
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(avq[q & 1][i], bv[r][j], acc[i][j], 0, 0, 0);
      if constexpr (G == Gq0 + 5 && q + 1 < NS && (PMAX >= 5 || q + 1 < PMAX)) {
        // plane q's predecessor has been used up six MFMAs ago: its registers take plane q + 1
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
          avq[(q + 1) & 1][ii] = *reinterpret_cast<const bf16x8*>(a + (q + 1) * PLANE_A + ii * 32 * QPW_LD);
      }
      constexpr int P0 = G * NP / NM, P1 = (G + 1) * NP / NM;
      static_for<P1 - P0>([&](auto D_) { piece(std::integral_constant<int, P0 + decltype(D_)::value>{}); });
      __builtin_amdgcn_sched_barrier(0);
    });
    lds_barrier();
  };
#endif
  gload(0, Slot0());
  gload(min(1, klast), Slot1());
  lstore(smem, Slot0());
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
  // epilogue: + bias, uniform tile base + 32-bit lane offsets (host check: 128 * ldc < 2^28)
  float* Cs = p.C + (int64_t)m0 * p.ldc + n0;
  const int ldc = (int)p.ldc;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int nl = wn * 96 + j * 32 + l31;
    const bool nok = (n0 + nl) < p.N;
    const float bz = (p.bias && nok) ? p.bias[n0 + nl] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ml = wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (nok && (m0 + ml) < p.M) Cs[ml * ldc + nl] = acc[i][j][e] + bz;
      }
  }
}

// C[m][n] = sum_k A[m][k] * B[n][k] + bias[n] with BOTH operands fp32: A is split into three bf16 planes in the kernel, B is
// given pre-split (three bf16 planes [3][N][ldb], B = B_0 + B_1 + B_2 exactly: ofq_split_f32_bf16x3).  products = 9: all
// plane pairs (the exact product up to fp32 accumulation); 6: the leading ones (dropped terms <= 2^-24 relative).
extern "C" int ofq_gemm_bf16x3x3_nt(const float* A, const void* B_planes, float* C, const float* bias, int products, int64_t M,
                                    int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int64_t plane_stride,
                                    ofq_stream_t stream) {
  if (!A || !B_planes || !C || M <= 0 || N <= 0 || K <= 0 || (products != 6 && products != 9)) return OFQ_EINVAL;
  if ((K & 7) || (lda & 3) || (ldb & 7) || (plane_stride & 7) || !al16(A) || !al16(B_planes) || M >= (1ll << 30) || N >= (1ll << 30))
    return OFQ_EINVAL;
  QGemmArgs a = {};
  a.A = A; a.B = B_planes; a.C = C; a.bias = bias; a.sBp = plane_stride;
  a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.M = (int)M; a.N = (int)N; a.K = (int)K;
  a.tiles_m = (int)ceil_div(M, 128); a.tiles_n = (int)ceil_div(N, 128); a.alpha = 1.f; a.nb1 = 1;
  // measured at 25 344 rows (tools/plane_gemm_bench.py, nine products, wide / narrow us): N = 384, K = 384: 80 / 108;
  // N = 384, K = 1536: 238 / 346; N = 1152, K = 384: 234 / 239; N = 1536, K = 384: 300 / 281 -- the wide tile wins where a
  // workgroup's k-loop is long or the narrow grid (594 workgroups on 512 slots) quantises badly
  if (N >= 256 && 128 * ldc < (1ll << 28) && (K & 15) == 0 && (K >= 1024 || N <= 384)) {
    a.tiles_n = (int)ceil_div(N, 384);
    dim3 gridw((unsigned)(a.tiles_m * a.tiles_n));
    if (products == 9) hipLaunchKernelGGL(gemm_bf16x3x3_wide_kernel<5>, gridw, dim3(512), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(gemm_bf16x3x3_wide_kernel<3>, gridw, dim3(512), 0, (hipStream_t)stream, a);
    OFQ_LAUNCH_CHECK();
    return 0;
  }
  dim3 grid((unsigned)(a.tiles_m * a.tiles_n));
  if (products == 9) hipLaunchKernelGGL((qgemm_bf16s_nt_kernel<3, false, 3, 5>), grid, dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((qgemm_bf16s_nt_kernel<3, false, 3, 3>), grid, dim3(256), 0, (hipStream_t)stream, a);
  OFQ_LAUNCH_CHECK();
  return 0;
}

// x (fp32, n elements) -> three bf16 planes p0, p1, p2 with x = p0 + p1 + p2 exactly (each plane the round-to-nearest bf16
// of what is left): the one-off split of a frozen fp32 weight matrix
__global__ __launch_bounds__(256) void split_f32_bf16x3_kernel(const float* __restrict__ x, unsigned short* __restrict__ planes,
                                                               int64_t n, int64_t plane_stride) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float r = x[i];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const __bf16 h = (__bf16)r;                              // RNE
    planes[q * plane_stride + i] = __builtin_bit_cast(unsigned short, h);
    r = __fsub_rn(r, (float)h);                              // exact: the residual fits in fp32
  }
}
extern "C" int ofq_split_f32_bf16x3(const float* x, void* planes, int64_t n, int64_t plane_stride, ofq_stream_t stream) {
  if (!x || !planes || n <= 0 || plane_stride < n) return OFQ_EINVAL;
  hipLaunchKernelGGL(split_f32_bf16x3_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, x,
                     (unsigned short*)planes, n, plane_stride);
  OFQ_LAUNCH_CHECK();
  return 0;
}
