// ------------------------------------------------------------------------------------------------ bf16-split NN (attention dxq)
// dxq[b,n,c] = sum_h sum_m (dS[b,h,n,m] * aq[m,h]) * qq[b,m,h,c]      (autograd of attention.py:210 wrt x_hat)
// A = dS is K-contiguous (m), scaled along k by the (token, head) LSQ step gathered with stride H and split into three
// bf16 planes [row][k]; B = the qkx codes, contiguous along c, staged as a [k][c] bf16 plane and read with the
// LDS transpose read.  The head sum is a k-batch loop inside the kernel, so dxq is written once.
struct QNnArgs {
  const float* A; const int8_t* B; float* C;
  const float* s;        // LSQ steps of qkx: index k*ks_stride + kb
  const unsigned* amax;  // wide kernel, two-plane fp16 form: bits of an upper bound of max |A| over the columns that are read
  int64_t lda, ldb, ldc, sA0, sB0, sC0, sAk, sBk;
  int M, N, K, nkb, ks_stride, tiles_m, tiles_n, accumulate;
  float gscale;
  int64_t sA1, sB1, sC1;  // inner batch (grid.y = batches * nb1): plain attention runs one head per batch entry
  int nb1;                // 0 / 1: no inner batch
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void qgemm_bf16s_nn_kernel(QNnArgs p) {
  constexpr int BM = 128, NS = 3;
  constexpr int PLANE_A = BM * QBS_LD;              // [row][k] bf16, 80 B rows
  constexpr int PLANE_B = QTN_BK * QTN_LD;          // [k][c]  bf16, 320 B rows
  __shared__ __attribute__((aligned(16))) unsigned char smem[NS * PLANE_A + PLANE_B];
  int tile, gby;
  xcd_remap_grid(tile, gby);
  const int nb1 = p.nb1 > 1 ? p.nb1 : 1;
  const int b0 = gby / nb1, b1 = gby % nb1;
  const int tm = tile / p.tiles_n, tn = tile % p.tiles_n;
  const int m0 = tm * BM, n0 = tn * 128;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  const float* Ab = p.A + b0 * p.sA0 + b1 * p.sA1;
  const int8_t* Bb = p.B + b0 * p.sB0 + b1 * p.sB1;
  const int K = p.K;
  const int nkt = (K + QBS_BK - 1) / QBS_BK;
  const int T = nkt * p.nkb;

  int64_t offA[4];
  bool okA[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int f = tid + 256 * i;
    const int row = f >> 3;
    okA[i] = (m0 + row) < p.M;
    offA[i] = (int64_t)min(m0 + row, p.M - 1) * p.lda + (f & 7) * 4;
  }
  const int kqa = (tid & 7) * 4;
  const int b_k = tid >> 3, b_c = (tid & 7) * 16;
  const bool b_ok = (n0 + b_c) < p.N;
  // gload only issues the loads (raw values + the masks as flags); scaling, masking and the split happen at the LDS
  // store of the next iteration, behind the MFMAs of this one.  Touching a loaded value inside gload (mask select,
  // effective-step arithmetic) puts the wait for it in front of the MFMAs, i.e. no overlap inside the workgroup.
  f32x4v ra[4];
  float rsv[4];
  i32x4 rb;
  bool rkin = false, rbk = false;
  auto gload = [&](int t) {
    const int kb = t / nkt, kt = t - kb * nkt;
    const int k0 = kt * QBS_BK;
    const bool kina = (k0 + kqa) < K;
    const int kbase = kina ? k0 + kqa : 0;
    const float* sp = p.s + kb;
#pragma unroll
    for (int e = 0; e < 4; ++e) rsv[e] = sp[(int64_t)min(kbase + e, K - 1) * p.ks_stride];
    const float* At = Ab + kb * p.sAk;
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const f32x4v*>(At + offA[i] + (kina ? k0 : -kqa));
    rkin = kina;
    const int k = k0 + b_k;
    rb = *reinterpret_cast<const i32x4*>(Bb + kb * p.sBk + (int64_t)min(k, K - 1) * p.ldb + (b_ok ? n0 + b_c : 0));
    rbk = b_ok && k < K;
  };
  auto lstore = [&]() {
#pragma unroll
    for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(rsv[e]));
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(ra[i]));
    asm volatile("" : "+v"(rb));
    float ks[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) ks[e] = rkin ? ofq_lsq_eff_scale(rsv[e], p.gscale) : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = tid + 256 * i;
      const int row = f >> 3;
      const float z = okA[i] ? 1.f : 0.f;
      const f32x2v k01 = {ks[0] * z, ks[1] * z}, k23 = {ks[2] * z, ks[3] * z};
      const f32x2v a01 = {ra[i][0], ra[i][1]}, a23 = {ra[i][2], ra[i][3]};
      unsigned lo[NS], hi[NS];
      split_pair_bf16<NS>(a01 * k01, lo);
      split_pair_bf16<NS>(a23 * k23, hi);
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        uint2 w;
        w.x = lo[q];
        w.y = hi[q];
        *reinterpret_cast<uint2*>(&smem[q * PLANE_A + row * QBS_LD + kqa * 2]) = w;
      }
    }
    const i32x4 rbm = rb & (rbk ? -1 : 0);
    unsigned w[8];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const int word = rbm[d];
      w[2 * d] = i8x2_to_bf16x2((int)(signed char)(word & 0xff), (int)(signed char)((word >> 8) & 0xff));
      w[2 * d + 1] = i8x2_to_bf16x2((int)(signed char)((word >> 16) & 0xff), (int)(signed char)((word >> 24) & 0xff));
    }
    unsigned char* dst = &smem[NS * PLANE_A + b_k * QTN_LD + b_c * 2];
    *reinterpret_cast<uint4*>(dst) = make_uint4(w[0], w[1], w[2], w[3]);
    *reinterpret_cast<uint4*>(dst + 16) = make_uint4(w[4], w[5], w[6], w[7]);
  };

  f32x16q acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int p16 = lane & 15;
  const int fr_off = (8 * lh + (p16 >> 2)) * QTN_LD + (16 * ((lane >> 4) & 1) + 4 * (p16 & 3)) * 2;
  const bool wave_on = (m0 + wm * 64 < p.M) && (n0 + wn * 64 < p.N);     // plain dq: 64 columns, half the waves only stage
  gload(0);
  for (int t = 0; t < T; ++t) {
    lstore();
    __syncthreads();
    gload(min(t + 1, T - 1));          // unconditional (the last one is never stored): no guard, no merged wait state
    __builtin_amdgcn_sched_barrier(0);  // ... and ahead of the MFMAs (the scheduler otherwise sinks the loads below them)
    const unsigned char* a = &smem[(wm * 64 + l31) * QBS_LD + lh * 16];
    if (wave_on)
#pragma unroll
    for (int ks = 0; ks < QBS_BK / 16; ++ks) {
      bf16x8 bv[2];
#pragma unroll
      for (int j = 0; j < 2; ++j)
        bv[j] = tr_frag(&smem[NS * PLANE_A + ks * 16 * QTN_LD + fr_off + (wn * 64 + j * 32) * 2]);
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        bf16x8 av[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) av[i] = *reinterpret_cast<const bf16x8*>(a + q * PLANE_A + i * 32 * QBS_LD + ks * 32);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  float* Cb = p.C + b0 * p.sC0 + b1 * p.sC1;
  if (!p.accumulate) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn * 64 + j * 32 + l31;
      if (n >= p.N) continue;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
          if (m < p.M) Cb[(int64_t)m * p.ldc + n] = acc[i][j][e];
        }
    }
  } else {      // old values fetched unconditionally on clamped addresses, eight at a time (see the wide dX kernel)
    int ncc[2];
    bool nok[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn * 64 + j * 32 + l31;
      nok[j] = n < p.N;
      ncc[j] = min(n, p.N - 1);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int eb = 0; eb < 4; ++eb) {
        float old[4][2];
#pragma unroll
        for (int ee = 0; ee < 4; ++ee) {
          const int mc = min(m0 + wm * 64 + i * 32 + ee + 8 * eb + 4 * lh, p.M - 1);
#pragma unroll
          for (int j = 0; j < 2; ++j) old[ee][j] = Cb[(int64_t)mc * p.ldc + ncc[j]];
        }
#pragma unroll
        for (int ee = 0; ee < 4; ++ee) {
          const int m = m0 + wm * 64 + i * 32 + ee + 8 * eb + 4 * lh;
#pragma unroll
          for (int j = 0; j < 2; ++j)
            if (m < p.M && nok[j]) Cb[(int64_t)m * p.ldc + ncc[j]] = acc[i][j][eb * 4 + ee] + old[ee][j];
        }
      }
  }
}
