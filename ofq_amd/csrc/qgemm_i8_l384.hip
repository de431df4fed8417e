// ------------------------------------------------------------------------------------------------ int8 forward, fc2-shaped layers
// A linear layer whose output is 384 wide and whose contraction is long (DeiT-S fc2: K = 1536) in ONE 128 x 384 tile per
// workgroup, the k-loop fed by LDS-DMA (round 6).  What round 6's counters and the 64-row-tile experiment said
// about the light-epilogue launches of qgemm_i8_nt_kernel<0>: they are bound by operand staging -- every 128 x 128 tile
// re-reads its weight tile, and a k-step of 64 bytes waits ~0.8 us for loads that two register slots cannot keep far enough ahead,
// against 0.11 us of MFMAs in it.  Here
//   * one workgroup (8 waves, 2 x 4, each 64 rows x 96 columns = 2 x 3 MFMA blocks) owns all 384 columns of a 128-row panel: the
//     activation panel is read once, the weights (590 KB, L2-resident) once per panel instead of once per tile;
//   * operands go global -> LDS by global_load_lds_dwordx4 (no staging registers, no ds_write pass) into a ring of FOUR 32 KB
//     stages, three k-steps ahead of the MFMAs; a wave-instruction lands as 1 KB = 16 rows x 64 B, lane-linear, so the bank
//     swizzle is applied to the SOURCE address (chunk c of row r sits in slot c ^ ((r >> 1) & 3)) and again on the fragment reads;
//   * hipcc does not count asm loads: the waits are written here (s_waitcnt vmcnt(8): of the twelve LDS-DMA pieces a wave has in
//     flight at the top of a k-step the four oldest have landed), one barrier per k-step (everybody's pieces of this stage are in
//     LDS, everybody has finished reading the stage that is refilled next); steps past the end re-fetch the last stage into a free
//     slot so that the count stays the same to the end.
// The epilogue is qgemm_i8_nt_kernel<0>'s expression, element for element: y = cs[n] * (a_eff[m % S] * I + r[n]) + bias[n]: the same
// bits as the 128 x 128 kernel (tests/test_kernels_gpu.py::test_i8_forward_128x384_lds_dma_kernel_equals_the_128x128_kernel).
// Measured (tools/i8_fused_bench.py, profiles/r06_i8_l384.txt): fc2 35.6 -> 28.8 us.  The K = 384 layers of the same width (v, proj)
// stay on the 128 x 128 kernel: this form runs them in 16.7 / 24 us against 18 / 22 -- their k-loop is six steps, and with it
// removed altogether they still take 15 us (a launch, 49 MB of traffic, the epilogue); a variant that runs two 192-column halves
// as one k-step stream, the first half's stores draining under the second half's loop, was built, is bit-identical and is slower
// (20 / 26 / 40 us): all of that is in DESIGN 4.
#define L384_BK 64
#define L384_NST 4
#define L384_STAGE ((128 + 384) * L384_BK)        // 32768 bytes: pieces 0..7 the activation rows, 8..31 the weight rows

// one LDS-DMA piece: 64 lanes x 16 B from (sbase + voff) to LDS [m0 .. m0 + 1024)
__device__ __forceinline__ void l384_glds(unsigned voff, const void* sbase, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(sbase) : "memory");
}

__global__ __launch_bounds__(512) void qgemm_i8_l384_kernel(QGemmArgs p) {
  constexpr int BM = 128;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[L384_NST * L384_STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 2, wn = w & 3;
  const int l31 = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.x * BM;
  const int nkt = p.K / L384_BK;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;      // (LDS byte address of the ring)

  // ---- epilogue parameters: requested now, consumed behind the k-loop (an ordinary load consumed inside the loop would make
  // hipcc drain the whole queue there)
  float pre_ra, pre_c[3][3];
  {
    const int m = min(m0 + (tid & (BM - 1)), p.M - 1);
    pre_ra = p.s[m % p.S];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int nc = wn * 96 + j * 32 + l31;
      pre_c[j][0] = p.cs[nc];
      pre_c[j][1] = (p.r ? p.r : p.cs)[nc];
      pre_c[j][2] = (p.bias ? p.bias : p.cs)[nc];
    }
  }

  // ---- producer side: wave w issues pieces 4 w .. 4 w + 3 of every stage (waves 0, 1: activation rows; 2 .. 7: weight rows)
  const int r16 = lane >> 2;
  const unsigned chunk = (unsigned)((lane & 3) ^ ((r16 >> 1) & 3));    // the source chunk that belongs in this lane's LDS slot
  unsigned voff[4];
  const bool isA = w < 2;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int pc = w * 4 + q;
    if (isA) voff[q] = (unsigned)min(m0 + pc * 16 + r16, p.M - 1) * (unsigned)p.lda + chunk * 16u;
    else voff[q] = (unsigned)((pc - 8) * 16 + r16) * (unsigned)p.ldb + chunk * 16u;
  }
  const unsigned char* gsrc = isA ? (const unsigned char*)p.A : (const unsigned char*)p.B;
  const unsigned piece0 = lds0 + (unsigned)w * 4096u;
  auto issue = [&](int kt) {
    const unsigned char* sb = gsrc + (int64_t)kt * L384_BK;            // (uniform: an SGPR pair)
    const unsigned dst = piece0 + (unsigned)(kt & (L384_NST - 1)) * (unsigned)L384_STAGE;
#pragma unroll
    for (int q = 0; q < 4; ++q) l384_glds(voff[q], sb, dst + (unsigned)q * 1024u);
  };

  // ---- consumer side: swizzled fragment addresses (row l31 of a 32-row block, 16 bytes of k at chunk 2 ks + lh)
  const int swz = ((l31 & 15) >> 1) & 3;
  unsigned cofs[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) cofs[ks] = (unsigned)(((ks * 2 + lh) ^ swz) << 4);
  unsigned offA[2], offB[3];
#pragma unroll
  for (int i = 0; i < 2; ++i) offA[i] = (unsigned)(((wm * 64 + i * 32 + l31) >> 4) * 1024 + (l31 & 15) * 64);
#pragma unroll
  for (int j = 0; j < 3; ++j) offB[j] = (unsigned)(8 * 1024 + ((wn * 96 + j * 32 + l31) >> 4) * 1024 + (l31 & 15) * 64);

  i32x16 acc[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0;

  issue(0);
  issue(min(1, nkt - 1));
  issue(min(2, nkt - 1));
  for (int kt = 0; kt < nkt; ++kt) {
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                   // this wave's four pieces of stage kt are in LDS ...
    __builtin_amdgcn_s_barrier();                                      // ... and so are everybody's; stage kt - 1 is free
    asm volatile("" ::: "memory");
    issue(min(kt + 3, nkt - 1));
    const unsigned char* sb = smem + (kt & (L384_NST - 1)) * L384_STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      i32x4 av[2], bv[3];
#pragma unroll
      for (int i = 0; i < 2; ++i) av[i] = *reinterpret_cast<const i32x4*>(sb + offA[i] + cofs[ks]);
#pragma unroll
      for (int j = 0; j < 3; ++j) bv[j] = *reinterpret_cast<const i32x4*>(sb + offB[j] + cofs[ks]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // (the re-fetches of the last steps)
  __syncthreads();

  // ---- epilogue: qgemm_i8_nt_kernel<0>'s arithmetic
  float* row_a = reinterpret_cast<float*>(smem);
  if (tid < BM) row_a[tid] = ofq_lsq_eff_scale(pre_ra, p.gscale);
  __syncthreads();
  float csn[3], rn[3], bz[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    csn[j] = pre_c[j][0] * p.alpha;
    rn[j] = p.r ? pre_c[j][1] : 0.f;
    bz[j] = p.bias ? pre_c[j][2] : 0.f;
  }
  const int colb = wn * 96 + l31;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int r = wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
      if (m0 + r >= p.M) continue;
      const float ae = row_a[r];
      float* rowp = p.C + (int64_t)(m0 + r) * p.ldc + colb;
#pragma unroll
      for (int j = 0; j < 3; ++j)
        rowp[j * 32] = __fadd_rn(__fmul_rn(csn[j], __fadd_rn(__fmul_rn(ae, (float)acc[i][j][e]), rn[j])), bz[j]);
    }
}

// may this launch take the 128 x 384 kernel?  (a 384-wide layer without a by-product, a long contraction)
static bool i8_l384_ok(const QGemmArgs& a) {
  const char* e = getenv("OFQ_I8_L384");                       // test hook / A-B switch (read at every launch: a test compares the
  const int mode = e ? atoi(e) : 1;                            // two kernels bit for bit in one process; 2: also the short contractions)
  if (mode == 0) return false;
  if (a.N != 384 || (a.K % L384_BK) || a.K < (mode == 2 ? 3 : 12) * L384_BK || !a.C || a.qout) return false;
  if ((int64_t)a.M * a.lda >= (1ll << 31) || (int64_t)a.N * a.ldb >= (1ll << 31)) return false;
  return true;
}
static void i8_l384_launch(const QGemmArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(qgemm_i8_l384_kernel, dim3((unsigned)ceil_div(a.M, 128)), dim3(512), 0, st, a);
}
