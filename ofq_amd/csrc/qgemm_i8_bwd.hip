// ------------------------------------------------------------------------------------------------ int8 recompute + LSQ backward
// Backward of "linear layer -> input quantiser of its only consumer" without the saved activation: the forward GEMM
// (ofq_qgemm_i8_nt_q with C = NULL) wrote only the consumer's int8 codes; here the fp32 layer output y is RECOMPUTED
// from the same integer codes with the same k-loop and the same epilogue expression (bit-identical to what the forward
// would have stored), and the consumer quantiser's backward (ofq_lsq_bwd's arithmetic, element for element) runs on it in
// registers:   dy = d/dy [ LSQ([gelu](y) + b4; step) ],  plus the partials of d(step), d(b4), d(baft).
// HBM traffic per element: 4 B read (gy) + 4 B written (dy), against 4 B written forward + 8 B read + 4 B written backward
// for the stored-activation pair; the extra int8 GEMM (K = C) is free next to that.
// QMODE 1: per-row step, index ((m * qrowmul + n / qcoldiv) % qS); QMODE 2: per-column step.
// Partials: lrow[m][2 * tiles_n] (row mode: sum of dsc over each 64-column half tile), lcol[tiles_m][nacc][N].
// (An "interior" instantiation -- no bounds selects, rows addressed through a uniform scalar base plus one 32-bit lane offset,
// 145 instead of 153 us for qkx -- existed in round 3 and was removed in round 4: at the full DeiT-S size it returned exact
// zeros for 16-lane groups of dy on tile rows 13 / 77 (accumulator element 5 of the first row block, upper half-wave: the
// gradients requested BEFORE the k-loop) in ~50 of 25 216 rows, differently from launch to launch, and kept doing so with its
// loads, selects and stores replaced one by one by this form's; reading its ISA against this one's (same barriers, same
// s_waitcnt structure around the LDS reuse and the gradient loads, different register allocation: its row offsets live in
// v[136:137] and are overwritten by the last gradient loads) did not show the cause.  This form is held to bit-identical
// repeats at full size by tests/test_fullsize_gpu.py::test_recompute_backward_stress.)
// DQKX (QKR attention, qkx = x_hat . W_qk^T -> its quantiser -> scores): the incoming gradient is not read from memory but
// FORMED here, gy[(b, m), (h, c)] = sum_n dS[b, h, n, m] * (a_eff[n] * qx[b, n, c] + bax[c]) -- the product the stream kernel
// (qgemm_bf16s_tn_wide_stream_kernel, two fp16 planes) writes as `dqkx`, in its order of operations, so the fused launch
// returns what the pair returns bit for bit while the 4 B/element of dqkx are neither written nor read.  The A operand of
// the recompute (the x codes) is the B operand of that product.  A 128-row tile of the flat (image, token) rows holds rows
// of at most two images (N >= 128): every row reads its own image's dS panel, the code operand is staged once per image and
// a 32-row block that straddles the boundary multiplies twice with the other image's rows zeroed.
template <int QMODE, bool GELU, bool DQKX = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void qgemm_i8_lsqbwd_kernel(QGemmArgs p) {
  constexpr int BM = 128, BN = 128;
  static_assert(!DQKX || (QMODE == 1 && !GELU), "the fused attention form is the per-row quantiser without a GELU");
  __shared__ __attribute__((aligned(16))) unsigned char smem[2][(BM + BN) * QI8_LD];
  int tm, tn, gby;
  qgemm_tile_id(p, tm, tn, gby);
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  const unsigned char* A = (const unsigned char*)p.A;
  const unsigned char* B = (const unsigned char*)p.B;
  float pre_ra, pre_rb = 1.f, pre_c[2][5];
  float dymax = 0.f;
  int ncol[2];
  {
    const int m = min(m0 + (tid & (BM - 1)), p.M - 1);
    pre_ra = p.s[m % p.S];
    if (QMODE == 1) pre_rb = p.qs[((int64_t)m * p.qrowmul + n0 / p.qcoldiv) % p.qS];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      ncol[j] = n0 + wn * 64 + j * 32 + l31;
      const int nc = min(ncol[j], p.N - 1);
      pre_c[j][0] = p.cs[nc];
      pre_c[j][1] = (p.r ? p.r : p.cs)[nc];
      pre_c[j][2] = (p.bias ? p.bias : p.cs)[nc];
      pre_c[j][3] = (p.qb4 ? p.qb4 : p.cs)[nc];
      pre_c[j][4] = (QMODE == 2 ? p.qs : p.cs)[nc];
    }
  }
  // the incoming gradients of the first half of the wave tile are requested BEFORE the k-loop (32 registers ride through
  // it), those of the second half right after it: the epilogue never waits for a cold HBM round trip
  const float* G = p.lx;
  const int ncl[2] = {min(ncol[0], p.N - 1), min(ncol[1], p.N - 1)};
  const bool cok[2] = {ncol[0] < p.N, ncol[1] < p.N};
  float g0[16][2], g1[16][2];
  auto gload = [&](float (&g)[16][2], int i) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = min(m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh, p.M - 1);
#pragma unroll
      for (int j = 0; j < 2; ++j) g[e][j] = G[(int64_t)m * p.ldlx + ncl[j]];
    }
  };
  f32x16q gacc[2][2];
  if constexpr (DQKX) {
    unsigned char* sm = &smem[0][0];
    constexpr int PLANE = QTN_BK * QTN_LD;                    // [32 k][128 fp16 + pad]: A hi, A lo, codes of image 0, of image 1
    static_assert(4 * PLANE <= 2 * (BM + BN) * QI8_LD, "the four planes overlay the int8 staging buffers");
    const int Ntok = p.dN, ldS = (int)p.ldS;
    const int hh = n0 / p.qcoldiv, c0 = n0 - hh * p.qcoldiv;
    const int bA = m0 / Ntok;
    const int bound = (bA + 1) * Ntok - m0;                   // tile row at which the next image begins
    const bool two = bound < BM && (m0 + bound) < p.M;
    float sE, inv_sE;                                         // the launch's power of two, as the stream kernel forms it
    {
      const float mx = fmaxf(block_absmax<256>(p.s, p.S, reinterpret_cast<float*>(sm), tid), 1e-5f) * 1.0001f;
      const float am = ofq_amax_load(p.amax);
      f16_plane_scale(am == am ? am * mx : am, sE, inv_sE);
    }
    const unsigned c64 = 0x64646464u;
    // staging maps: A chunk [32 n][128 m] fp32 -- rows ak + 8 i, tile rows at .. at + 3 as two pairs (a pair never straddles
    // two images: N is even); code chunk [32 n][128 c] int8, 16 B per thread and image
    const int ak = tid >> 5, at = (tid & 31) * 4;
    const float* Ap[2];
    bool aok[2];
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      const int r = m0 + at + 2 * pr;
      aok[pr] = r < p.M;
      const int rc = aok[pr] ? r : 0;
      const int b = rc / Ntok, i = rc - b * Ntok;
      Ap[pr] = p.dS + ((int64_t)(b * p.dH + hh) * Ntok) * ldS + i;
    }
    const int bk = tid >> 3, bc = (tid & 7) * 16;
    const unsigned char* Xc0 = A + (int64_t)bA * Ntok * p.lda + c0 + bc;
    const unsigned char* Xc1 = A + (int64_t)(two ? bA + 1 : bA) * Ntok * p.lda + c0 + bc;
    // NSL register slots: the loads of k-step kt + NSL are issued while step kt is multiplied, so a step never waits for the
    // round trip of its own operands (a k-step is 16 MFMAs per wave: far shorter than an L2 miss)
    constexpr int NSL = 2;
    f32x2v ra[NSL][4][2];
    float rs[NSL][4];
    i32x4 rb0[NSL], rb1[NSL];
    auto gload1 = [&](int kt, auto SLOT) {
      constexpr int sl = decltype(SLOT)::value;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int n = kt * QTN_BK + ak + 8 * i;
        const int nc = min(n, Ntok - 1);
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) ra[sl][i][pr] = *reinterpret_cast<const f32x2v*>(Ap[pr] + (int64_t)nc * ldS);
        rs[sl][i] = p.s[min(n, p.S - 1)];
      }
      const int64_t nb = (int64_t)min(kt * QTN_BK + bk, Ntok - 1) * p.lda;
      rb0[sl] = *reinterpret_cast<const i32x4*>(Xc0 + nb);
      rb1[sl] = *reinterpret_cast<const i32x4*>(Xc1 + nb);
    };
    // column sums of the raw dS (the offset term), in the stream kernel's order: its thread a_k adds rows a_k, a_k + 16 of
    // every k-step; rows ak, ak + 16 go to csA, rows ak + 8, ak + 24 to csB (its thread a_k = ak + 8)
    float csA[4] = {0.f, 0.f, 0.f, 0.f}, csB[4] = {0.f, 0.f, 0.f, 0.f};
    auto lstore1 = [&](int kt, auto SLOT) {
      constexpr int sl = decltype(SLOT)::value;
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(ra[sl][i][0]), "+v"(ra[sl][i][1]), "+v"(rs[sl][i]));
      asm volatile("" : "+v"(rb0[sl]), "+v"(rb1[sl]));
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool nok = (kt * QTN_BK + ak + 8 * i) < Ntok;
        const float se = valu_mul(valu_eff_scale(rs[sl][i], p.gscale), sE);      // (unconditional: a select, not a branch)
        const float sc = nok ? se : 0.f;
        unsigned hi[2], lo[2];
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          const float okf = (aok[pr] && nok) ? 1.f : 0.f;
          const float scp = aok[pr] ? sc : 0.f;
          float* cs = (i & 1) ? csB : csA;
          cs[2 * pr] = valu_fma(ra[sl][i][pr][0], okf, cs[2 * pr]);
          cs[2 * pr + 1] = valu_fma(ra[sl][i][pr][1], okf, cs[2 * pr + 1]);
          const f32x2v x = ra[sl][i][pr] * scp;
          split2_f16(x[0], x[1], hi[pr], lo[pr]);
        }
        uint2 w;
        w.x = hi[0]; w.y = hi[1];
        *reinterpret_cast<uint2*>(&sm[(ak + 8 * i) * QTN_LD + at * 2]) = w;
        w.x = lo[0]; w.y = lo[1];
        *reinterpret_cast<uint2*>(&sm[PLANE + (ak + 8 * i) * QTN_LD + at * 2]) = w;
      }
      unsigned bw[8];
#pragma unroll
      for (int d = 0; d < 4; ++d) valu_cvt4_i8_f16((unsigned)rb0[sl][d], c64, bw[2 * d], bw[2 * d + 1]);
      unsigned char* dst = &sm[2 * PLANE + bk * QTN_LD + bc * 2];
      *reinterpret_cast<uint4*>(dst) = make_uint4(bw[0], bw[1], bw[2], bw[3]);
      *reinterpret_cast<uint4*>(dst + 16) = make_uint4(bw[4], bw[5], bw[6], bw[7]);
      if (two) {
#pragma unroll
        for (int d = 0; d < 4; ++d) valu_cvt4_i8_f16((unsigned)rb1[sl][d], c64, bw[2 * d], bw[2 * d + 1]);
        *reinterpret_cast<uint4*>(dst + PLANE) = make_uint4(bw[0], bw[1], bw[2], bw[3]);
        *reinterpret_cast<uint4*>(dst + PLANE + 16) = make_uint4(bw[4], bw[5], bw[6], bw[7]);
      }
    };
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) gacc[i][j][e] = 0.f;
    const int p16 = lane & 15;
    const int fr_off = (8 * lh + (p16 >> 2)) * QTN_LD + (16 * ((lane >> 4) & 1) + 4 * (p16 & 3)) * 2;
    // which image a 32-row block of this wave multiplies with: 0 the first, 1 the second, 2 both (rows of the other zeroed)
    int mode[2];
    const int wm_s = __builtin_amdgcn_readfirstlane(wm);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int R0 = wm_s * 64 + i * 32;
      mode[i] = (!two || R0 + 32 <= bound) ? 0 : (R0 >= bound ? 1 : 2);
    }
    const int nkt1 = (Ntok + QTN_BK - 1) / QTN_BK;
    auto mma1 = [&]() {
#pragma unroll
      for (int ks = 0; ks < QTN_BK / 16; ++ks) {
        bf16x8 bv0[2], bv1[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const unsigned char* bp = &sm[2 * PLANE + ks * 16 * QTN_LD + fr_off + (wn * 64 + j * 32) * 2];
          bv0[j] = tr_frag(bp);
          bv1[j] = two ? tr_frag(bp + PLANE) : bv0[j];
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const bf16x8 av = tr_frag(&sm[q * PLANE + ks * 16 * QTN_LD + fr_off + (wm * 64 + i * 32) * 2]);
            if (mode[i] == 2) {
              const bool first = (wm * 64 + i * 32 + l31) < bound;
              const i32x4 ai = __builtin_bit_cast(i32x4, av);
              const i32x4 z4 = {0, 0, 0, 0};
              const bf16x8 a0 = __builtin_bit_cast(bf16x8, first ? ai : z4), a1 = __builtin_bit_cast(bf16x8, first ? z4 : ai);
#pragma unroll
              for (int j = 0; j < 2; ++j) {
                gacc[i][j] = mfma_16b<true>(a0, bv0[j], gacc[i][j]);
                gacc[i][j] = mfma_16b<true>(a1, bv1[j], gacc[i][j]);
              }
            } else {
#pragma unroll
              for (int j = 0; j < 2; ++j) gacc[i][j] = mfma_16b<true>(av, mode[i] ? bv1[j] : bv0[j], gacc[i][j]);
            }
          }
        }
      }
    };
    // one k-step: the loads are unconditional (clamped to the last step: a load under a condition makes the wait-count pass
    // drain every outstanding load at the join), only the staging and the MFMAs of a step past the end are skipped
    auto stepk = [&](int kt, auto SLOT) {
      const bool live = kt < nkt1;                             // (uniform)
      if (live) {
        lstore1(kt, SLOT);
        __syncthreads();
      }
      gload1(min(kt + NSL, nkt1 - 1), SLOT);
      __builtin_amdgcn_sched_barrier(0);
      if (live) {
        mma1();
        __syncthreads();
      }
    };
    static_for<NSL>([&](auto S_) { gload1(min((int)decltype(S_)::value, nkt1 - 1), S_); });
    for (int kt = 0; kt < nkt1; kt += NSL)
      static_for<NSL>([&](auto S_) { stepk(kt + decltype(S_)::value, S_); });
    // gy = acc / 2^E + colsum_n(dS)[m] * bax[c]      (the stream kernel's epilogue expression)
    float4* red = reinterpret_cast<float4*>(sm);
    float* red1 = reinterpret_cast<float*>(sm) + 16 * 32 * 4;
    if (p.z) {
      red[ak * 32 + (tid & 31)] = make_float4(csA[0], csA[1], csA[2], csA[3]);
      red[(ak + 8) * 32 + (tid & 31)] = make_float4(csB[0], csB[1], csB[2], csB[3]);
      __syncthreads();
      if (ak == 0) {
        float4 tt = red[tid & 31];
#pragma unroll
        for (int g = 1; g < 16; ++g) {
          const float4 u = red[g * 32 + (tid & 31)];
          tt.x += u.x; tt.y += u.y; tt.z += u.z; tt.w += u.w;
        }
        *reinterpret_cast<float4*>(red1 + at) = tt;
      }
      __syncthreads();
    }
    float bfv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) bfv[j] = p.z ? p.z[c0 + wn * 64 + j * 32 + l31] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float rsum = p.z ? red1[wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh] : 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j)
          gacc[i][j][e] = p.z ? gacc[i][j][e] * inv_sE + rsum * bfv[j] : gacc[i][j][e] * inv_sE;
      }
    __syncthreads();                                           // red1 is read: the int8 k-loop may overwrite the buffers
  } else {
    gload(g0, 0);
  }
  i32x16 acc[2][2];
  i8_mainloop<2>(p, A, B, m0, n0, smem, acc);
  if constexpr (!DQKX) gload(g1, 1);

  float* row_a = reinterpret_cast<float*>(&smem[0][0]);       // [128] effective input step of the row
  float* row_b = row_a + BM;                                  // [128] effective step of the consumer quantiser (row mode)
  float* row_c = row_a + 2 * BM;                              // [128] its correctly rounded reciprocal
  float* colred = row_a + 3 * BM;                             // [2 wm][3 acc][128] column partials
  if (tid < BM) {
    row_a[tid] = ofq_lsq_eff_scale(pre_ra, p.gscale);
    const float rb = ofq_lsq_eff_scale(pre_rb, p.qgscale);
    row_b[tid] = rb;
    row_c[tid] = __fdiv_rn(1.f, rb);
  }
  __syncthreads();
  float csn[2], rn[2], bz[2], qb[2], qsc[2], qrc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    csn[j] = pre_c[j][0] * p.alpha;
    rn[j] = p.r ? pre_c[j][1] : 0.f;
    bz[j] = p.bias ? pre_c[j][2] : 0.f;
    qb[j] = p.qb4 ? pre_c[j][3] : 0.f;
    qsc[j] = QMODE == 2 ? ofq_lsq_eff_scale(pre_c[j][4], p.qgscale) : 1.f;
    qrc[j] = QMODE == 2 ? __fdiv_rn(1.f, qsc[j]) : 1.f;
  }
  const float lo = p.qlo, hi = p.qhi;
  const float tol = ofq_lsq_level_tol(lo, hi);
  const float half_m_tol = 0.5f - tol;
  float cb4[2] = {0.f, 0.f}, cba[2] = {0.f, 0.f}, cds[2] = {0.f, 0.f};
  float rds[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) rds[k] = 0.f;
  auto half = [&](const float (&g)[16][2], auto I_) {
    constexpr int i = decltype(I_)::value;
#pragma unroll
    for (int eg = 0; eg < 4; ++eg) {
      // 4 rows x 2 columns share one exactness check (see ofq_lsq_bwd_fast): a wave that raises it redoes the group with
      // the IEEE division sequences, so every value equals ofq_lsq_bwd's bit for bit
      float yv[4][2], xin[4][2], gev[4][2], alv[4][2], dq[4][2], dsc[4][2];
      OfqLsqFlags fl;
#pragma unroll
      for (int ee = 0; ee < 4; ++ee) {
        const int e = eg * 4 + ee;
        const int mr = wm * 64 + i * 32 + ee + 8 * eg + 4 * lh;
        const bool mok = (m0 + mr) < p.M;
        const float ae = row_a[mr];
        const float alr = row_b[mr], rar = row_c[mr];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          yv[ee][j] = __fadd_rn(__fmul_rn(csn[j], __fadd_rn(__fmul_rn(ae, (float)acc[i][j][e]), rn[j])), bz[j]);
          xin[ee][j] = __fadd_rn(GELU ? ofq_gelu(yv[ee][j]) : yv[ee][j], qb[j]);
          gev[ee][j] = (mok && cok[j]) ? (DQKX ? gacc[i][j][e] : g[e][j]) : 0.f;
          alv[ee][j] = QMODE == 2 ? qsc[j] : alr;
          ofq_lsq_bwd_fast(xin[ee][j], gev[ee][j], alv[ee][j], QMODE == 2 ? qrc[j] : rar, lo, hi, fl, dq[ee][j], dsc[ee][j]);
        }
      }
      if (__builtin_amdgcn_ballot_w64(ofq_lsq_flags_risky(fl, half_m_tol, tol)) != 0ull) {
#pragma unroll
        for (int ee = 0; ee < 4; ++ee)
#pragma unroll
          for (int j = 0; j < 2; ++j) ofq_lsq_bwd_exact(xin[ee][j], gev[ee][j], alv[ee][j], lo, hi, dq[ee][j], dsc[ee][j]);
      }
#pragma unroll
      for (int ee = 0; ee < 4; ++ee) {
        const int e = eg * 4 + ee;
        const int m = m0 + wm * 64 + i * 32 + ee + 8 * eg + 4 * lh;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          cb4[j] += dq[ee][j];
          cba[j] += gev[ee][j];
          if (QMODE == 2) cds[j] += dsc[ee][j]; else rds[i * 16 + e] += dsc[ee][j];
          const float dy = GELU ? dq[ee][j] * ofq_gelu_grad(yv[ee][j]) : dq[ee][j];
          if (m < p.M && cok[j]) { p.C[(int64_t)m * p.ldc + ncol[j]] = dy; dymax = fmaxf(dymax, fabsf(dy)); }
        }
      }
    }
  };
  half(g0, std::integral_constant<int, 0>());
  half(g1, std::integral_constant<int, 1>());
  if (p.amax_out) ofq_amax_publish(p.amax_out, dymax);       // max |dy| of the written elements (two-plane GEMMs downstream)
  // ---- row partials (row mode): sum over the 32 lanes that hold the columns of one row; transpose-reduce, 31 exchanges:
  // after the step with mask w a lane keeps the half of its slots selected by its bit w, so lane l31 ends with slot l31
  if (QMODE == 1) {
#pragma unroll
    for (int w = 16; w >= 1; w >>= 1) {
      const bool up = (l31 & w) != 0;
#pragma unroll
      for (int k = 0; k < w; ++k) {
        const float send = up ? rds[k] : rds[k + w];
        const float keep = up ? rds[k + w] : rds[k];
        rds[k] = keep + __shfl_xor(send, w, 64);
      }
    }
    const int e = l31 & 15, i = l31 >> 4;
    const int m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
    if (m < p.M) p.lrow[(int64_t)m * (2 * p.tiles_n) + 2 * tn + wn] = rds[0];
  }
  // ---- column partials: lane pair (lh) -> wave pair (wm) through LDS -> lcol[tm][acc][n]
  constexpr int NACC = QMODE == 2 ? 3 : 2;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    cb4[j] += __shfl_xor(cb4[j], 32, 64);
    cba[j] += __shfl_xor(cba[j], 32, 64);
    if (QMODE == 2) cds[j] += __shfl_xor(cds[j], 32, 64);
    if (lh == 0) {
      const int c = wn * 64 + j * 32 + l31;
      colred[(wm * 3 + 0) * BN + c] = cb4[j];
      colred[(wm * 3 + 1) * BN + c] = cba[j];
      if (QMODE == 2) colred[(wm * 3 + 2) * BN + c] = cds[j];
    }
  }
  __syncthreads();
  for (int idx = tid; idx < NACC * BN; idx += 256) {
    const int c = idx % BN, ac = idx / BN;
    if (n0 + c < p.N)
      p.lcol[((int64_t)tm * NACC + ac) * p.N + n0 + c] = colred[ac * BN + c] + colred[(3 + ac) * BN + c];
  }
}

static void i8_lsqbwd_ws(int64_t M, int64_t N, int colmode, size_t* rowf, size_t* colf) {
  const int64_t tiles_m = ceil_div(M, 128), tiles_n = ceil_div(N, 128);
  *rowf = colmode ? 0 : (size_t)M * 2 * tiles_n;
  *colf = (size_t)tiles_m * (colmode ? 3 : 2) * N;
}
extern "C" size_t ofq_qgemm_i8_lsq_bwd_ws_bytes(int64_t M, int64_t N, int q_colmode) {
  if (M <= 0 || N <= 0) return 0;
  size_t rf, cf;
  i8_lsqbwd_ws(M, N, q_colmode, &rf, &cf);
  return (rf + cf) * sizeof(float) + 256;
}

// (dS != NULL: the fused attention form, see qgemm_i8_lsqbwd_kernel<1, false, true>; gy is not read then)
static int i8_lsq_bwd_launch(const int8_t* A, const int8_t* B, const float* bias, const float* col_scale, float col_mult,
                             const float* r, const float* lsq_s, int64_t S, float gscale, int64_t M, int64_t N, int64_t K,
                             int64_t lda, int64_t ldb, const float* gy, int64_t ldg, float* dy, int64_t ldd,
                             const float* q_s, int64_t q_S, float q_gscale, const float* q_b4, int q_lo, int q_hi,
                             int q_gelu, int q_rowmul, int64_t q_coldiv, int q_colmode, float* ds, float* db4, float* dbaft,
                             void* ws, size_t ws_bytes, void* amax_out, const float* dS, int64_t ldS, int64_t H, int64_t Ntok,
                             const float* bax, const void* amax_in, ofq_stream_t stream) {
  if (!A || !B || !col_scale || !lsq_s || (!gy && !dS) || !dy || !q_s || !ws || M <= 0 || N <= 0 || K <= 0 || S <= 0 || q_S <= 0)
    return OFQ_EINVAL;
  if ((K & 15) || (lda & 15) || (ldb & 15) || !al16(A) || !al16(B) || M >= (1ll << 30) || N >= (1ll << 30) || (gy && ldg < N) || ldd < N)
    return OFQ_EINVAL;
  if (q_rowmul < 1 || q_coldiv < 1 || (q_rowmul > 1 && (q_coldiv % 128 || q_rowmul * q_coldiv != N)) || (q_colmode && q_S != N))
    return OFQ_EINVAL;
  const int64_t T = q_colmode ? 1 : q_S / q_rowmul;          // quantiser rows per batch element (tokens)
  if (!q_colmode && (T * q_rowmul != q_S || M % T)) return OFQ_EINVAL;
  size_t rf, cf;
  i8_lsqbwd_ws(M, N, q_colmode, &rf, &cf);
  if (ws_bytes < (rf + cf) * sizeof(float)) return OFQ_ENOWS;
  QGemmArgs a = {};
  a.A = A; a.B = B; a.C = dy; a.bias = bias; a.cs = col_scale; a.r = r; a.s = lsq_s;
  a.lda = lda; a.ldb = ldb; a.ldc = ldd; a.M = (int)M; a.N = (int)N; a.K = (int)K; a.S = (int)S;
  a.tiles_m = (int)ceil_div(M, 128); a.tiles_n = (int)ceil_div(N, 128); a.gscale = gscale; a.alpha = col_mult; a.nb1 = 1;
  a.qs = q_s; a.qS = (int)q_S; a.qgscale = q_gscale; a.qb4 = q_b4; a.qlo = (float)q_lo; a.qhi = (float)q_hi; a.qgelu = q_gelu;
  a.qrowmul = q_rowmul; a.qcoldiv = (int)(q_coldiv > (1ll << 30) ? (1ll << 30) : q_coldiv); a.qcolmode = q_colmode;
  a.lx = gy; a.ldlx = ldg; a.lrow = (float*)ws; a.lcol = (float*)ws + rf; a.amax_out = (unsigned*)amax_out;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)(a.tiles_m * a.tiles_n)), block(256);
  auto launch = [&](auto QM, auto GE) {
    constexpr int qm = decltype(QM)::value;
    constexpr bool ge = decltype(GE)::value;
    hipLaunchKernelGGL((qgemm_i8_lsqbwd_kernel<qm, ge>), grid, block, 0, st, a);
  };
  if (dS) {
    // x codes [B Ntok][K] (lda == row stride), steps per token (S == Ntok), qkx columns (head, channel): N == H K, the quantiser's
    // rows are (token, head); a 128-row tile must not reach into a third image, a row pair not into a second one
    if (q_colmode || q_gelu || !amax_in || H <= 0 || Ntok < 128 || (Ntok & 1) || S != Ntok || M % Ntok || N != H * K || q_coldiv != K ||
        (K % 128) || (ldS & 1) || ldS < Ntok || q_rowmul != H || H * Ntok * ldS >= (1ll << 31) || (((uintptr_t)dS) & 7))
      return OFQ_EINVAL;
    a.dS = dS; a.ldS = ldS; a.dH = (int)H; a.dN = (int)Ntok; a.z = bax; a.amax = (const unsigned*)amax_in;
    hipLaunchKernelGGL((qgemm_i8_lsqbwd_kernel<1, false, true>), grid, block, 0, st, a);
  } else if (q_colmode) {
    if (q_gelu) launch(std::integral_constant<int, 2>(), std::true_type());
    else launch(std::integral_constant<int, 2>(), std::false_type());
  } else {
    if (q_gelu) launch(std::integral_constant<int, 1>(), std::true_type());
    else launch(std::integral_constant<int, 1>(), std::false_type());
  }
  OFQ_LAUNCH_CHECK();
  // second stage (fixed order, no atomics): ds over batches and half tiles / over row tiles; db4, dbaft over row tiles
  const int nacc = q_colmode ? 3 : 2;
  SumJobs jobs = {};
  int64_t maxc = 0;
  if (ds) {
    if (q_colmode) jobs.j[0] = {a.lcol + 2 * N, ds, N, a.tiles_m, nacc * N, 1, q_gscale, 0, 0};
    else {
      const int64_t nparts = 2 * (int64_t)a.tiles_n, pph = nparts / q_rowmul;
      jobs.j[0] = {a.lrow, ds, q_S, M / T, T * nparts, (int)pph, q_gscale, q_rowmul, nparts};
    }
    maxc = jobs.j[0].ncols;
  }
  if (db4) { jobs.j[1] = {a.lcol, db4, N, a.tiles_m, nacc * N, 1, 1.0f, 0, 0}; if (N > maxc) maxc = N; }
  if (dbaft) { jobs.j[2] = {a.lcol + N, dbaft, N, a.tiles_m, nacc * N, 1, 1.0f, 0, 0}; if (N > maxc) maxc = N; }
  if (maxc > 0) {
    strided_sum_launch(jobs, maxc, 3, st);
    OFQ_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int ofq_qgemm_i8_lsq_bwd(const int8_t* A, const int8_t* B, const float* bias, const float* col_scale, float col_mult,
                                    const float* r, const float* lsq_s, int64_t S, float gscale, int64_t M, int64_t N, int64_t K,
                                    int64_t lda, int64_t ldb, const float* gy, int64_t ldg, float* dy, int64_t ldd,
                                    const float* q_s, int64_t q_S, float q_gscale, const float* q_b4, int q_lo, int q_hi,
                                    int q_gelu, int q_rowmul, int64_t q_coldiv, int q_colmode, float* ds, float* db4, float* dbaft,
                                    void* ws, size_t ws_bytes, void* amax_out, ofq_stream_t stream) {
  if (!gy) return OFQ_EINVAL;
  return i8_lsq_bwd_launch(A, B, bias, col_scale, col_mult, r, lsq_s, S, gscale, M, N, K, lda, ldb, gy, ldg, dy, ldd, q_s, q_S, q_gscale,
                           q_b4, q_lo, q_hi, q_gelu, q_rowmul, q_coldiv, q_colmode, ds, db4, dbaft, ws, ws_bytes, amax_out, nullptr, 0, 0,
                           0, nullptr, nullptr, stream);
}

// QKR attention: backward of [scores <- qkx quantiser <- qkx = x_hat . W_qk^T] from dS in one launch (attention.py:200-210,
// lsq.py:571-602): dy = d/d(qkx) of the scores through the quantiser, with the quantiser's step / offset gradients -- what
// ofq_qattn_dqkx_bf16s (two-plane form) followed by ofq_qgemm_i8_lsq_bwd returns, bit for bit, without the dqkx tensor.
extern "C" int ofq_qattn_dqkx_lsq_bwd(const int8_t* xcodes, const int8_t* wcodes, const float* bias, const float* col_scale,
                                      float col_mult, const float* r, const float* sx, float gscale_x, const float* bax,
                                      const float* dS, int64_t ldS, const void* amax, int64_t B, int64_t H, int64_t Ntok, int64_t C,
                                      int64_t lda, int64_t ldb, float* dy, int64_t ldd, const float* q_s, int64_t q_S,
                                      float q_gscale, const float* q_b4, int q_lo, int q_hi, float* ds, float* db4, float* dbaft,
                                      void* ws, size_t ws_bytes, void* amax_out, ofq_stream_t stream) {
  if (!dS || !amax || B <= 0 || H <= 0 || Ntok <= 0 || C <= 0) return OFQ_EINVAL;
  return i8_lsq_bwd_launch(xcodes, wcodes, bias, col_scale, col_mult, r, sx, Ntok, gscale_x, B * Ntok, H * C, C, lda, ldb, nullptr, 0, dy,
                           ldd, q_s, q_S, q_gscale, q_b4, q_lo, q_hi, 0, (int)H, C, 0, ds, db4, dbaft, ws, ws_bytes, amax_out, dS, ldS, H,
                           Ntok, bax, amax, stream);
}
