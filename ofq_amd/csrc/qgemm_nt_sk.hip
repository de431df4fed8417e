// ---- streaming / stream-K form of the wide input-gradient kernel -------------------------------------------------------
// The one-tile-per-workgroup launch above pays, per tile, a first memory round trip and a 196 KB store burst at the same moment
// as every other workgroup (measured: ~19 us of a 37 us launch at K = 384 are launch edges), and N = 1536 makes 792 tiles =
// 3.09 per CU.  Here `gridDim.x` persistent workgroups (one per CU) share the launch's tiles x k-step PAIRS evenly: workgroup
// w walks the units [U w / G, U (w+1) / G) in order as ONE continuous k-step stream -- the load cursors run three / two steps
// ahead of the MFMAs straight through tile boundaries, a boundary costs its epilogue only.  The launcher (nt_sk_grid) picks G
// so that no tile is cut whenever a divisor of the tile count fills three quarters of the chip (198 for the DeiT-S shapes:
// every bit equals the one-tile-per-workgroup kernel's); otherwise G = the CU count and tiles are CUT.  A tile that lies
// inside one run is stored directly.  A tile cut by a run boundary is finished by its OWNER, the workgroup that holds its
// k = 0 piece (the tail of that workgroup's run): the other holders (the heads of the following workgroups' runs -- done
// early) publish their fp32 partial tile to their slot of the workspace (write-through stores, drained, then a flag), the
// owner adds them in workgroup order and stores.  The cut points are a function of (M, N, K, G) only and the order of
// the additions is fixed, so results are bit-identical from launch to launch; they are NOT bit-identical to the
// one-tile-per-workgroup kernel (another association of the same fp32 sums) unless no tile is cut.
// Up to two K-SEGMENTS: C = sum_seg alpha_seg * (A_seg * ks_seg) . B_seg^T -- the input gradients that two layers send to the
// same tensor (v and W_qk of the QKR attention both consume x_hat) as one GEMM over the concatenated contraction.
// Flags: one int per workgroup, zero before the first launch (caller), set by the publisher, reset by the owner.  Every
// spin is bounded: a timeout raises the error word behind the flags and the kernel finishes with wrong numbers, not a hang.
struct QNtSkSeg {
  const float* A; const unsigned short* B; const float* s;
  const unsigned* amax;      // F16 form: bits of (an upper bound of) max |A| of this segment (device scalar)
  unsigned lda4, ldb2;       // row pitch of A / B in bytes
  int nkt;                   // k-steps of QBS_BK in this segment
  float alpha;
  int hi_only;               // segment 1, F16 form: only A's leading plane multiplies this segment's B (a three-product forward)
};
struct QNtSkArgs {
  QNtSkSeg seg[2];
  float* C; int64_t ldc;
  const float* col_bias;     // optional: + col_bias[n] on the finished tile (a forward product: y = x . W^T + b)
  int M, N, nkt, tiles_n, accumulate;
  unsigned long long units;  // tiles * nkt / 2: the workgroups share PAIRS of k-steps (every piece starts on LDS stage 0)
  float* ws; int* flags;     // [G][8 * 3 * 512 * 4] partial tiles; flags[w], error word flags[4096]
};
#define QNT_SK_SPIN_LIMIT (1 << 21)

// F16: two fp16 planes of the scaled panel (see split2_f16) against fp16 weight codes instead of three bf16 planes against
// bf16 codes: 8 NJ MFMAs, 6 VALU per element pair and two plane stores per k-step instead of 12 NJ, 13 and three.
template <int NJ, int NSEG, bool F16>
__global__ __launch_bounds__(512) void qgemm_bf16s_nt_wide_sk_kernel(QNtSkArgs p) {
  constexpr int BM = 128, BN = 128 * NJ, NS = F16 ? 2 : 3;
  constexpr int PLANE = BM * QBS_LD;
  constexpr int STAGE = NS * PLANE + BN * QBS_LD;
  constexpr int NB = NJ;
  constexpr int SLOT_F4 = 8 * NJ * 512;                  // float4 per partial-tile slot
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 2, wn = wid & 3;
  const int l31 = lane & 31, lh = lane >> 5;
  const int G = gridDim.x, w = blockIdx.x;
  const int nkt = p.nkt;                                  // even (host check): units are PAIRS of k-steps
  // this workgroup's run of k-steps, in the order tile 0 steps 0 .. nkt-1, tile 1 ...: an even number, from an even step
  const unsigned u_begin = 2u * (unsigned)((p.units * (unsigned long long)w) / (unsigned long long)G);
  const unsigned u_end = 2u * (unsigned)((p.units * (unsigned long long)(w + 1)) / (unsigned long long)G);
  const unsigned kqa4 = (unsigned)(tid & 7) * 16u, kqb2 = (unsigned)(tid & 3) * 16u;   // byte offset of this lane's chunk in a k-step
  const int kqa = (tid & 7) * 4, kqb = (tid & 3) * 8;
  // F16: one power of two for the launch, from the segments' amax words and the largest k-scale (all workgroups compute
  // the same value from the same inputs: the partial tiles of a cut tile are in the same units)
  float sE = 1.f, inv_sE = 1.f;
  if constexpr (F16) {
    float t = 0.f;
#pragma unroll
    for (int sg = 0; sg < NSEG; ++sg) {
      const float m = p.seg[sg].s ? block512_absmax(p.seg[sg].s, p.seg[sg].nkt * QBS_BK, reinterpret_cast<float*>(smem), tid) : 1.f;
      const float a = ofq_amax_load(p.seg[sg].amax);
      t = fmaxf(t, a * m * (NSEG > 1 ? fabsf(p.seg[sg].alpha) : 1.f));
      if (!(a == a)) t = a;                                  // a NaN bound stays one
    }
    f16_plane_scale(t, sE, inv_sE);
  }

  f32x16q acc[2][NJ];
  f32x4v ra[2][2], rks[2];
  i32x4 rb[NB];
  float rkm[2] = {1.f, 1.f};                              // alpha of the segment a slot's panel belongs to (NSEG > 1)
  bool rhs[2] = {false, false};                           // that segment has a k-scale vector
  using Slot0 = std::integral_constant<int, 0>;
  using Slot1 = std::integral_constant<int, 1>;

  // ---- load cursors: the (tile, k-step) the NEXT dY-panel loads / weight loads fetch.  They run three / two steps ahead of
  // the MFMAs straight through piece and tile boundaries (a boundary costs its epilogue, not a pipeline restart) and stop on
  // the run's last step.  Tile bases are uniform (scalar registers); a lane adds one 32-bit offset per row, recomputed when
  // a cursor enters a new tile (rows clamped to the matrix: rows past M / N only feed elements that are never stored).
  unsigned la_u = u_begin, lb_u = u_begin;
  int la_kt = (int)(u_begin % (unsigned)nkt), lb_kt = la_kt;
  int la_tile = (int)(u_begin / (unsigned)nkt), lb_tile = la_tile;
  unsigned voA[NSEG][2], voB[NSEG][NB];
  auto set_a_tile = [&](int tile) {
    const int m0 = (tile / p.tiles_n) * BM;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned row = (unsigned)min(m0 + ((tid + 512 * i) >> 3), p.M - 1);
#pragma unroll
      for (int sg = 0; sg < NSEG; ++sg) voA[sg][i] = row * p.seg[sg].lda4 + kqa4;
    }
  };
  auto set_b_tile = [&](int tile) {
    const int n0 = (tile % p.tiles_n) * BN;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const unsigned row = (unsigned)min(n0 + ((tid + 512 * i) >> 2), p.N - 1);
#pragma unroll
      for (int sg = 0; sg < NSEG; ++sg) voB[sg][i] = row * p.seg[sg].ldb2 + kqb2;
    }
  };
  set_a_tile(la_tile);
  set_b_tile(lb_tile);
  auto adv_a = [&]() {
    if (la_u + 1 < u_end) {
      ++la_u;
      if (++la_kt == nkt) {
        la_kt = 0;
        set_a_tile(++la_tile);
      }
    }
  };
  auto adv_b = [&]() {
    if (lb_u + 1 < u_end) {
      ++lb_u;
      if (++lb_kt == nkt) {
        lb_kt = 0;
        set_b_tile(++lb_tile);
      }
    }
  };
  // uniform part of the addresses of tile step kt: segment base + k offset
  auto seg_of = [&](int kt) -> int { return (NSEG > 1 && kt >= p.seg[0].nkt) ? 1 : 0; };
  auto a_base = [&](int kt, int sg) -> const char* {
    return reinterpret_cast<const char*>(p.seg[sg].A) + (size_t)(kt - (sg ? p.seg[0].nkt : 0)) * (QBS_BK * 4);
  };
  auto s_base = [&](int kt, int sg) -> const char* {
    return reinterpret_cast<const char*>(p.seg[sg].s) + (size_t)(kt - (sg ? p.seg[0].nkt : 0)) * (QBS_BK * 4);
  };
  auto b_base = [&](int kt, int sg) -> const char* {
    return reinterpret_cast<const char*>(p.seg[sg].B) + (size_t)(kt - (sg ? p.seg[0].nkt : 0)) * (QBS_BK * 2);
  };
  // prologue-style (un-interleaved) loads / staging: the first three steps of the run only
  auto gload = [&](auto SLOT) {
    constexpr int sl = decltype(SLOT)::value;
    const int sg = seg_of(la_kt);
    const char* ab = a_base(la_kt, sg);
    rhs[sl] = p.seg[sg].s != nullptr;
    // no scale vector: the load still happens (any valid address) and the value is replaced at the staging
    rks[sl] = *reinterpret_cast<const f32x4v*>(rhs[sl] ? s_base(la_kt, sg) + kqa4 : ab + voA[NSEG > 1 ? sg : 0][0]);
    if constexpr (NSEG > 1) rkm[sl] = p.seg[sg].alpha;
#pragma unroll
    for (int i = 0; i < 2; ++i) ra[sl][i] = *reinterpret_cast<const f32x4v*>(ab + voA[NSEG > 1 ? sg : 0][i]);
    adv_a();
  };
  auto gload_b = [&]() {
    const int sg = seg_of(lb_kt);
    const char* bb = b_base(lb_kt, sg);
#pragma unroll
    for (int i = 0; i < NB; ++i) rb[i] = *reinterpret_cast<const i32x4*>(bb + voB[NSEG > 1 ? sg : 0][i]);
    adv_b();
  };
  // scale of slot sl's panel: ks (or 1), times the segment's alpha when there are two segments (alpha is a power of two in
  // every caller, 1 / 2^bits: folding it here or applying it in the epilogue gives the same bits)
  auto slot_scale = [&](auto SLOT, float (&ksv)[4]) {
    constexpr int sl = decltype(SLOT)::value;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = rhs[sl] ? rks[sl][e] : 1.f;
      if constexpr (NSEG > 1) v *= rkm[sl];
      if constexpr (F16) v *= sE;
      ksv[e] = v;
    }
  };
  auto lstore = [&](unsigned char* sb, auto SLOT) {
    constexpr int sl = decltype(SLOT)::value;
    asm volatile("" : "+v"(rks[sl]), "+v"(ra[sl][0]), "+v"(ra[sl][1]));
    float ksv[4];
    slot_scale(SLOT, ksv);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (tid + 512 * i) >> 3;
      const f32x2v k01 = {ksv[0], ksv[1]}, k23 = {ksv[2], ksv[3]};
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
        uint2 wv;
        wv.x = lo[q];
        wv.y = hi[q];
        *reinterpret_cast<uint2*>(&sb[q * PLANE + row * QBS_LD + kqa * 2]) = wv;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) asm volatile("" : "+v"(rb[i]));
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int row = (tid + 512 * i) >> 2;
      *reinterpret_cast<i32x4*>(&sb[NS * PLANE + row * QBS_LD + kqb * 2]) = rb[i];
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  // one k-step of the stream: the MFMAs of the step staged in `cur`; behind them, in small pieces, the staging of the next
  // step (register slot SLOT -> `nxt`), the weight loads at the B cursor (two steps ahead) and the dY-panel loads at the A
  // cursor (three ahead).  Same piece list as qgemm_bf16s_nt_wide_kernel.
  // F16 piece list per row chunk (10 pieces): per pair [x0 = a0*ks0, x1 = a1*ks1] [h = cvt_pk(x0, x1)] [r0 = x0 - h.lo, r1 = x1 - h.hi]
  // [l = cvt_pk(r0, r1)], then the two plane stores
  constexpr int NM = 4 * NS * NJ, NPA = F16 ? 10 : 17, NP = 2 * NPA + NB + NB + 3;
  // ho (uniform; two-segment fp16 form): the k-step in `cur` belongs to a segment whose B multiplies A's LEADING plane only --
  // the trailing plane's MFMAs are branched over (one body: a second instantiation of this step costs 200 spilled registers)
  auto step = [&](const unsigned char* cur, unsigned char* nxt, auto SLOT, const bool ho) {
    constexpr int sl = decltype(SLOT)::value;
    constexpr int NSE = NS, NME = NM;
    const unsigned char* a = &cur[(wm * 64 + l31) * QBS_LD + lh * 16];
    const unsigned char* b = &cur[NS * PLANE + (wn * 32 * NJ + l31) * QBS_LD + lh * 16];
    static_assert(QBS_BK == 32, "two MFMA steps per k-step");
    bf16x8 av[NS][2], bv[2][NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) bv[0][j] = *reinterpret_cast<const bf16x8*>(b + j * 32 * QBS_LD);
#pragma unroll
    for (int q = 0; q < NSE; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i) av[q][i] = *reinterpret_cast<const bf16x8*>(a + q * PLANE + i * 32 * QBS_LD);
    __builtin_amdgcn_sched_barrier(0);
    float ksv[4], x_ = 0.f, r1_ = 0.f, p0v[2], p1v[2], r2v[2];
    unsigned lo[NS], hi[NS];
    const int sgb = seg_of(lb_kt), sga = seg_of(la_kt);
    const char* bb2 = b_base(lb_kt, sgb);
    const char* ab3 = a_base(la_kt, sga);
    const bool has_s3 = p.seg[sga].s != nullptr;
    const char* sb3 = has_s3 ? s_base(la_kt, sga) + kqa4 : ab3;
    const float alpha3 = p.seg[sga].alpha;
    auto piece = [&](auto P_) {
      constexpr int P = decltype(P_)::value;
      if constexpr (P < 2 * NPA) {
        constexpr int i = P / NPA, r = P % NPA;
        if constexpr (r == 0 && i == 0) {          // first touch of the slot: the wait for its loads lands here
          asm volatile("" : "+v"(rks[sl]), "+v"(ra[sl][0]), "+v"(ra[sl][1]));
          slot_scale(SLOT, ksv);
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
            uint2 wv;
            wv.x = lo[q];
            wv.y = hi[q];
            *reinterpret_cast<uint2*>(&nxt[q * PLANE + row * QBS_LD + kqa * 2]) = wv;
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
          uint2 wv;
          wv.x = lo[q];
          wv.y = hi[q];
          *reinterpret_cast<uint2*>(&nxt[q * PLANE + row * QBS_LD + kqa * 2]) = wv;
        }
      } else if constexpr (P < 2 * NPA + NB) {
        constexpr int i = P - 2 * NPA;
        const int row = (tid + 512 * i) >> 2;
        asm volatile("" : "+v"(rb[i]));
        *reinterpret_cast<i32x4*>(&nxt[NS * PLANE + row * QBS_LD + kqb * 2]) = rb[i];
      } else if constexpr (P < 2 * NPA + 2 * NB) {
        constexpr int i = P - 2 * NPA - NB;
        rb[i] = *reinterpret_cast<const i32x4*>(bb2 + voB[NSEG > 1 ? sgb : 0][i]);
      } else {
        constexpr int wq = P - 2 * NPA - 2 * NB;
        if constexpr (wq == 0) {
          rks[sl] = *reinterpret_cast<const f32x4v*>(has_s3 ? sb3 : sb3 + voA[NSEG > 1 ? sga : 0][0]);
          rhs[sl] = has_s3;
          if constexpr (NSEG > 1) rkm[sl] = alpha3;
        } else {
          ra[sl][wq - 1] = *reinterpret_cast<const f32x4v*>(ab3 + voA[NSEG > 1 ? sga : 0][wq - 1]);
        }
      }
    };
    static_for<NME>([&](auto G_) {
      constexpr int Gi = decltype(G_)::value;
      constexpr int ks = Gi / (2 * NSE * NJ), q = (Gi / (2 * NJ)) % NSE, i = (Gi / NJ) % 2, j = Gi % NJ;
      if constexpr (NSEG > 1 && F16 && q == 1) {
        if (!ho) acc[i][j] = mfma_16b<F16>(av[q][i], bv[ks][j], acc[i][j]);
      } else {
        acc[i][j] = mfma_16b<F16>(av[q][i], bv[ks][j], acc[i][j]);
      }
      if constexpr (ks == 0) {
        if constexpr (Gi < NJ) bv[1][Gi] = *reinterpret_cast<const bf16x8*>(b + Gi * 32 * QBS_LD + 32);
        if constexpr (j == NJ - 1) av[q][i] = *reinterpret_cast<const bf16x8*>(a + q * PLANE + i * 32 * QBS_LD + 32);
      }
      constexpr int P0 = Gi * NP / NME, P1 = (Gi + 1) * NP / NME;
      static_for<P1 - P0>([&](auto D_) { piece(std::integral_constant<int, P0 + decltype(D_)::value>{}); });
      __builtin_amdgcn_sched_barrier(0);
    });
    lds_barrier();
    adv_b();
    adv_a();
  };

#if defined(NTSK_CLOCK_PROBE) || defined(NTSK_PHASE_PROBE)      // tools/nt_sk_sweep.py: shader clock the chip holds while G workgroups run this kernel
  const unsigned long long pc0 = __builtin_readcyclecounter(), pr0 = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll
  for (int i = 0; i < 2; ++i)
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
    const int kb = (int)(u - (unsigned)tile * (unsigned)nkt);
    const int ke = min(nkt, kb + (int)(u_end - u));
    const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    int kc = kb;                                              // k-step being multiplied (pairs never straddle the segments)
    for (int n2 = (ke - kb) >> 1; n2 > 0; --n2, kc += 2) {
      bool ho = false;
      if constexpr (NSEG > 1 && F16) ho = p.seg[1].hi_only && kc >= p.seg[0].nkt;
      step(smem, smem + STAGE, Slot1(), ho);
      step(smem + STAGE, smem, Slot0(), ho);
    }

    int l31e = l31, lhe = lh, tide = tid;
    asm volatile("" : "+v"(l31e), "+v"(lhe), "+v"(tide));      // keep the epilogue's lane offsets out of the k-loop's registers
#ifdef NTSK_PHASE_PROBE      // tools/nt_sk_phases.py: where the hand-off of a cut tile spends its time (10 ns ticks since kernel start)
#define NTSK_STAMP(slot) do { if (tid == 0) p.flags[4200 + w * 8 + (slot)] = (int)(__builtin_amdgcn_s_memrealtime() - pr0); } while (0)
    NTSK_STAMP(kb != 0 ? 0 : 2);
#else
#define NTSK_STAMP(slot) do {} while (0)
#endif
    if (kb != 0) {
      // ---- not the owner: publish the partial tile (register order, one float4 per lane and store: coalesced) ----
      // (the address is a VGPR pair: a scalar base would be restored from spill lanes by v_readlane right in front of the
      // asm statement, and the hazard recogniser does not pad a VALU-written SGPR in front of an opaque memory instruction)
      const char* slot = reinterpret_cast<const char*>(p.ws) + (size_t)w * (SLOT_F4 * 16) + (size_t)tide * 16u;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4v v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
            const char* sp = slot + (size_t)((i * NJ + j) * 4 + q) * (512 * 16);
            asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(sp), "v"(v) : "memory");
          }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      // (flags[4097]: fault injection for the tests -- the workgroup named there, +1, never publishes; zero in every real run)
      if (tid == 0 && p.flags[4097] != w + 1) __hip_atomic_store(p.flags + w, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      NTSK_STAMP(1);
    } else {
      if (ke != nkt) {
        // ---- owner of a cut tile: add the partials of the following workgroups, in order ----
        const unsigned tile_end = (unsigned)(tile + 1) * (unsigned)nkt;
        for (int x = w + 1; x < G; ++x) {
          const unsigned ux = 2u * (unsigned)((p.units * (unsigned long long)x) / (unsigned long long)G);
          if (ux >= tile_end) break;
          bool tmo = false;
          if (tid == 0) {
            int it = 0;
            bool got = true;
            while (__hip_atomic_load(p.flags + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
              if (++it > QNT_SK_SPIN_LIMIT) {
                // STICKY error word: this launch and every later one on this workspace is suspect until the caller has re-zeroed
                // the flag area (ofq_qgemm_bf16s_nt_sk_reset).  The publisher's flag is NOT reset here: it may still arrive,
                // and a flag cleared now and set later would be taken for the NEXT launch's partial.
                __hip_atomic_store(p.flags + 4096, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                got = false;
                break;
              }
              __builtin_amdgcn_s_sleep(8);
            }
            if (got) __hip_atomic_store(p.flags + x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (x == w + 1) NTSK_STAMP(3);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            tmo = !got;
          }
          if (__syncthreads_or(tmo ? 1 : 0)) continue;      // timed out: the tile goes without this partial (error word raised)
          const f32x4v* src = reinterpret_cast<const f32x4v*>(p.ws) + (size_t)x * SLOT_F4 + tide;
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
              f32x4v v[4];
#pragma unroll
              for (int q = 0; q < 4; ++q) v[q] = __builtin_nontemporal_load(src + ((i * NJ + j) * 4 + q) * 512);
#pragma unroll
              for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] += v[q][e];
            }
        }
      }
      if (ke != nkt) NTSK_STAMP(4);
      // ---- store the finished tile ----
      const float alpha = (NSEG > 1 ? 1.f : p.seg[0].alpha) * inv_sE;
      float cbv[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) cbv[j] = p.col_bias ? p.col_bias[min(n0 + wn * 32 * NJ + j * 32 + l31e, p.N - 1)] : 0.f;
      const bool interior = (m0 + BM <= p.M) && (n0 + BN <= p.N) && (int64_t)BM * p.ldc < (1ll << 28);
      if (interior) {
        float* Cs = p.C + (int64_t)m0 * p.ldc + n0;
        const int ldc = (int)p.ldc;
        const int nl0 = wn * 32 * NJ + l31e;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int eb = 0; eb < 4; ++eb) {
            const int mlb = (wm * 64 + i * 32 + 8 * eb + 4 * lhe) * ldc + nl0;
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
                const float v = acc[i][j][eb * 4 + ee] * alpha + cbv[j];
                Cs[mlb + ee * ldc + j * 32] = p.accumulate ? v + old[ee][j] : v;
              }
          }
      } else {
        int ncc[NJ];
        bool nok[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int n = n0 + wn * 32 * NJ + j * 32 + l31e;
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
              const int mc = min(m0 + wm * 64 + i * 32 + ee + 8 * eb + 4 * lhe, p.M - 1);
#pragma unroll
              for (int j = 0; j < NJ; ++j) old[ee][j] = p.accumulate ? p.C[(int64_t)mc * p.ldc + ncc[j]] : 0.f;
            }
#pragma unroll
            for (int ee = 0; ee < 4; ++ee) {
              const int m = m0 + wm * 64 + i * 32 + ee + 8 * eb + 4 * lhe;
#pragma unroll
              for (int j = 0; j < NJ; ++j)
                if (m < p.M && nok[j]) p.C[(int64_t)m * p.ldc + ncc[j]] = (acc[i][j][eb * 4 + ee] * alpha + cbv[j]) + old[ee][j];
            }
          }
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    u += (unsigned)(ke - kb);
  }
#ifdef NTSK_PHASE_PROBE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  NTSK_STAMP(5);
#endif
#ifdef NTSK_CLOCK_PROBE
  if (w == G / 2 && tid == 0) {
    p.flags[4098] = (int)(__builtin_readcyclecounter() - pc0);
    p.flags[4099] = (int)(__builtin_amdgcn_s_memrealtime() - pr0);
  }
#endif
}

// workspace: [QNT_SK_FLAG_BYTES of flags: one int per workgroup, then the error word (and the probe words)] [one partial-tile
// slot per workgroup]; the flags sit in front so that their place does not depend on the number of workgroups of a launch
#define QNT_SK_FLAG_BYTES 32768
#define QNT_SK_SLOT_BYTES (8 * 3 * 512 * 16)
extern "C" size_t ofq_qgemm_bf16s_nt_sk_ws_bytes(int num_wgs) {
  if (num_wgs < 0) num_wgs = -num_wgs;
  return num_wgs == 0 || num_wgs > 4096 ? 0 : QNT_SK_FLAG_BYTES + (size_t)num_wgs * QNT_SK_SLOT_BYTES;
}

// Number of workgroups for `tiles` tiles of `nkt` k-steps on a chip of `num_wgs` CUs.  Measured on MI355X (tools/nt_sk_sweep.py):
// the chip is power-bound in this kernel -- 2.04 GHz with 99 workgroups, 1.74 with 198, 1.72 with 256 -- and a cut tile costs
// its holders ~20 us (196 KB published write-through, the owner's acquire + read + store all behind the last k-step), so
// 256 workgroups with cuts only tie with 198 whole tiles (K = 2304: 122.0 vs 120.8 us).  Hence: whole tiles whenever a
// divisor of the tile count fills at least three quarters of the chip (no tile is cut: the k order of every tile, and so
// every bit of the result, is that of the one-tile-per-workgroup kernel), cuts otherwise.
static int nt_sk_grid(int64_t tiles, int num_wgs) {
  for (int64_t g = num_wgs; 4 * g >= 3 * (int64_t)num_wgs && g >= 1; --g)
    if (tiles % g == 0) return (int)g;
  return num_wgs;                 // cut tiles (also when there are fewer tiles than CUs: late Swin stages, 98 tiles of 96 k-steps)
}

// 1: the streaming launch is expected to beat the one-tile-per-workgroup launch (a workgroup gets at least 24 k-steps: what
// it saves are the launch edges between consecutive tiles; a 12-step launch of 198 tiles is 29.6 us either way), 0: not
extern "C" int ofq_qgemm_bf16s_nt_sk_pays(int64_t M, int64_t N, int64_t K, int num_wgs) {
  if (num_wgs <= 1 || N <= 128 || (K % (2 * QBS_BK)) != 0) return 0;
  const int nj = N > 256 ? 3 : 2;
  const int64_t tiles = ceil_div(M, 128) * ceil_div(N, 128 * nj);
  const int g = nt_sk_grid(tiles, num_wgs);
  if (tiles * (K / QBS_BK) < 24 * (int64_t)g) return 0;
  if (tiles % g == 0) return 1;
  const int64_t rounds = ceil_div(tiles, num_wgs);                      // cuts: only when the whole-tile launch fills badly
  return (double)tiles / (double)(rounds * num_wgs) < 0.85 ? 1 : 0;
}

extern "C" int ofq_qgemm_bf16s_nt_sk(const ofq_nt_seg* segs, int nseg, float* C, int accumulate, int64_t M, int64_t N, int64_t ldc,
                                     int num_wgs, void* ws, size_t ws_bytes, const float* col_bias, ofq_stream_t stream) {
  const bool forced = num_wgs < 0;               // exactly -num_wgs workgroups (tests, tools/nt_sk_sweep.py)
  if (forced) num_wgs = -num_wgs;
  if (!segs || (nseg != 1 && nseg != 2) || !C || !ws || M <= 0 || N <= 128 || num_wgs <= 0 || num_wgs > 4096) return OFQ_EINVAL;
  if (ws_bytes < ofq_qgemm_bf16s_nt_sk_ws_bytes(num_wgs)) return OFQ_ENOWS;
  if (M >= (1ll << 30) || N >= (1ll << 30) || !al16(ws) || ldc < N) return OFQ_EINVAL;
  QNtSkArgs a = {};
  int nkt = 0;
  for (int i = 0; i < nseg; ++i) {
    const ofq_nt_seg& sg = segs[i];
    if (!sg.A || !sg.B_bf16 || sg.K <= 0 || (sg.K % QBS_BK) || (sg.lda & 3) || (sg.ldb & 7) || sg.lda < sg.K || sg.ldb < sg.K ||
        !al16(sg.A) || !al16(sg.B_bf16) ||
        (sg.k_scale && !al16(sg.k_scale)) || M * sg.lda * 4 >= (1ll << 32) || N * sg.ldb * 2 >= (1ll << 32))
      return OFQ_EINVAL;
    a.seg[i].A = sg.A; a.seg[i].B = (const unsigned short*)sg.B_bf16; a.seg[i].s = sg.k_scale; a.seg[i].amax = (const unsigned*)sg.amax;
    if ((sg.amax != nullptr) != (segs[0].amax != nullptr)) return OFQ_EINVAL;      // one operand format per launch
    a.seg[i].lda4 = (unsigned)(sg.lda * 4); a.seg[i].ldb2 = (unsigned)(sg.ldb * 2);
    a.seg[i].nkt = (int)(sg.K / QBS_BK); a.seg[i].alpha = sg.alpha;
    a.seg[i].hi_only = sg.hi_only;
    // (a hi-only segment: the second of two, fp16 form, behind a whole number of k-step PAIRS)
    if (sg.hi_only && (i != 1 || !sg.amax || (segs[0].K % (2 * QBS_BK)))) return OFQ_EINVAL;
    nkt += a.seg[i].nkt;
  }
  const int nj = N > 256 ? 3 : 2;
  const int64_t tiles_m = ceil_div(M, 128), tiles_n = ceil_div(N, 128 * nj);
  if (tiles_m * tiles_n * nkt >= (1ll << 31) || (nkt & 1)) return OFQ_EINVAL;      // units are pairs of k-steps
  a.C = C; a.ldc = ldc; a.M = (int)M; a.N = (int)N; a.nkt = nkt; a.tiles_n = (int)tiles_n; a.accumulate = accumulate;
  a.col_bias = col_bias;
  a.units = (unsigned long long)(tiles_m * tiles_n) * (unsigned long long)(nkt / 2);
  a.flags = (int*)ws;
  a.ws = (float*)((char*)ws + QNT_SK_FLAG_BYTES);
  int g = forced ? num_wgs : nt_sk_grid(tiles_m * tiles_n, num_wgs);
  if ((unsigned long long)g > a.units) g = (int)a.units;
  const dim3 grid((unsigned)g), block(512);
  hipStream_t st = (hipStream_t)stream;
  const bool f16 = segs[0].amax != nullptr;        // two fp16 planes against fp16 codes (B_bf16 then holds fp16 values)
  if (nj == 3) {
    if (nseg == 1) {
      if (f16) hipLaunchKernelGGL((qgemm_bf16s_nt_wide_sk_kernel<3, 1, true>), grid, block, 0, st, a);
      else hipLaunchKernelGGL((qgemm_bf16s_nt_wide_sk_kernel<3, 1, false>), grid, block, 0, st, a);
    } else {
      if (f16) hipLaunchKernelGGL((qgemm_bf16s_nt_wide_sk_kernel<3, 2, true>), grid, block, 0, st, a);
      else hipLaunchKernelGGL((qgemm_bf16s_nt_wide_sk_kernel<3, 2, false>), grid, block, 0, st, a);
    }
  } else {
    if (nseg == 1) {
      if (f16) hipLaunchKernelGGL((qgemm_bf16s_nt_wide_sk_kernel<2, 1, true>), grid, block, 0, st, a);
      else hipLaunchKernelGGL((qgemm_bf16s_nt_wide_sk_kernel<2, 1, false>), grid, block, 0, st, a);
    } else {
      if (f16) hipLaunchKernelGGL((qgemm_bf16s_nt_wide_sk_kernel<2, 2, true>), grid, block, 0, st, a);
      else hipLaunchKernelGGL((qgemm_bf16s_nt_wide_sk_kernel<2, 2, false>), grid, block, 0, st, a);
    }
  }
  OFQ_LAUNCH_CHECK();
  return 0;
}

// The error word of a stream-K workspace, surfaced inside the step: loss[0] becomes NaN when a hand-off of an earlier launch on
// this workspace timed out (one thread; captured with the step, so a replayed step poisons its loss as well).
__global__ void nt_sk_check_kernel(const int* __restrict__ flags, float* __restrict__ loss) {
  if (__hip_atomic_load(flags + 4096, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) loss[0] = __builtin_nanf("");
}
extern "C" int ofq_qgemm_bf16s_nt_sk_check(const void* ws, float* loss, ofq_stream_t stream) {
  if (!ws || !loss) return OFQ_EINVAL;
  hipLaunchKernelGGL(nt_sk_check_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (const int*)ws, loss);
  OFQ_LAUNCH_CHECK();
  return 0;
}
// Back to the state of a fresh workspace: every flag, the error word and the test words zero (asynchronous, on `stream`).
extern "C" int ofq_qgemm_bf16s_nt_sk_reset(void* ws, ofq_stream_t stream) {
  if (!ws) return OFQ_EINVAL;
  return hipMemsetAsync(ws, 0, QNT_SK_FLAG_BYTES, (hipStream_t)stream) == hipSuccess ? 0 : OFQ_EINVAL;
}
