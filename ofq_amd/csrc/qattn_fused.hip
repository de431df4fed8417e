// Scores GEMM + softmax + unsigned LSQ in one kernel (reference attention.py:96-99 / :207-216): the score matrix S never
// travels to HBM (122 MB written and read back per DeiT-S block at 128 images).
//   phase 1  a workgroup owns 64 query rows x all keys (<= 256) of one (batch, head): exact int8 MFMA product of the codes
//            (i8_mainloop_g<1,4,1,2,2>: four waves side by side, 64 x 64 each), the scores epilogue of qgemm_i8_nt_kernel<1>
//            (same expression, same rounding), S -> LDS tile [64][260] fp32 (it overlays the k-loop's staging buffers)
//   phase 2  16 lanes per row, four rows per wave at a time: ofq_softmax_lsq_fwd's arithmetic, element for element (max,
//            expf, sum by 16-lane butterflies; IEEE divisions); prob (fp32, kept for the backward), the uint8 codes of P_hat
//            and their row sums (operands of the int8 P.V GEMM) leave with 16-byte / 4-byte stores
// HBM: the int8 operands in, 4 B + 1 B per score out.
#include "common.h"

struct QSsArgs {
  QGemmArgs g;                 // scores operands: A / B codes, strides, M = N = tokens, K, s / s2 / u / tq / z, gscale / gscale2
  const float* sm_s;           // softmax quantiser steps [S]
  const float* addend;         // optional [add_period][S][ld] (Swin: relative-position bias + shift mask)
  float* prob; unsigned char* codes; float* rowsum;
  int64_t ld, add_period;
  int S;
  float sm_gscale, alpha, hi;
};

// QSS_SLD = floats per row of the S tile in LDS.  260 (256 + 4: consecutive rows start 16 B apart in the banks) serves any
// N <= 256 with two workgroups per CU; 200 serves N <= 200 (DeiT: 198 tokens) in 51.4 KB, i.e. three workgroups per CU --
// the k-loop of one workgroup then runs under the softmax arithmetic of the other two.  Phase 2 reads whole float4 chunks
// up to column 255: with the short stride they run into the following rows (masked by column) and, for the last row,
// into 256 B of slack behind the tile.
// CB = 4: 256-key panel, four waves side by side (DeiT); CB = 1: 64-key panel, 2 x 2 waves of one 32 x 32 block each -- the
// 49-token Swin windows, where a 256-key panel would be three quarters padding (one window and head per workgroup).
template <int QSS_SLD, int CB = 4, bool TAIL = false>
__global__ __launch_bounds__(256) void qattn_scores_softmax_kernel(QSsArgs q) {
  const QGemmArgs& p = q.g;
  constexpr int BM = 64, BN = 64 * CB;
  constexpr int WGM = CB == 4 ? 1 : 2, MI = CB == 4 ? 2 : 1, NJ = CB == 4 ? 2 : 1;      // wave grid WGM x (4 / WGM), blocks per wave
  constexpr int KCH = CB;                                 // 64-column chunks of a score row in phase 2
  constexpr int STAGE = (BM + BN) * QI8_LD;
  constexpr int TILE_BYTES = BM * QSS_SLD * 4 + 256;
  __shared__ __attribute__((aligned(16))) unsigned char smem_raw[(2 * STAGE > TILE_BYTES ? 2 * STAGE : TILE_BYTES) + 2 * BM * 4];
  unsigned char (*smem)[STAGE] = reinterpret_cast<unsigned char (*)[STAGE]>(smem_raw);
  float* stile = reinterpret_cast<float*>(smem_raw);
  float* row_a = reinterpret_cast<float*>(smem_raw + (2 * STAGE > TILE_BYTES ? 2 * STAGE : TILE_BYTES));
  float* row_b = row_a + BM;
  // the row tiles of one (batch, head) share its key panel, the heads of one batch element its query rows: keep them on
  // one XCD, next to each other in time (hardware order alone spreads consecutive workgroups over the 8 L2s)
  int tm, gby;
  xcd_remap_grid(tm, gby);
  const int m0 = tm * BM;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int b0 = gby / p.nb1, b1 = gby % p.nb1;
  const unsigned char* A = (const unsigned char*)p.A + b0 * p.sA0 + b1 * p.sA1;
  const unsigned char* B = (const unsigned char*)p.B + b0 * p.sB0 + b1 * p.sB1;
  // epilogue parameters are requested before the k-loop (see qgemm_i8_nt_kernel)
  float pre_ra, pre_rb, pre_s2[NJ], pre_tq[NJ];
  const int wm = wid / (4 / WGM), wn = wid % (4 / WGM);
  {
    const int m = min(m0 + (tid & (BM - 1)), p.M - 1);
    pre_ra = p.s[m % p.S];
    pre_rb = p.u[((int64_t)b0 * p.M + m) * p.nb1 + b1];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int nc = min(wn * 32 * NJ + j * 32 + l31, p.N - 1);
      pre_s2[j] = p.s2[nc * p.s2s0 + b1 * p.s2s1];
      pre_tq[j] = p.tq[((int64_t)b0 * p.N + nc) * p.nb1 + b1];
    }
  }
  const float zz = p.z[b1];
  i32x16 acc[MI][NJ];
  i8_mainloop_g<1, CB, WGM, MI, NJ>(p, A, B, m0, 0, smem, acc);
  if (tid < BM) {
    row_a[tid] = ofq_lsq_eff_scale(pre_ra, p.gscale);
    row_b[tid] = pre_rb;
  }
  __syncthreads();             // (also: every wave has left the k-loop, the staging buffers may be overwritten)
  {
    // S[n,m] = ax[n] * (aq[m] * I + u[n]) + aq[m] * tq[m] + z   -- qgemm_i8_nt_kernel<1>'s expression
    float aq[NJ], tqa[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      aq[j] = ofq_lsq_eff_scale(pre_s2[j], p.gscale2);
      tqa[j] = __fadd_rn(__fmul_rn(aq[j], pre_tq[j]), zz);
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int mr = wm * 32 * MI + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        const float ax = row_a[mr], uu = row_b[mr];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int col = wn * 32 * NJ + j * 32 + l31;
          if (QSS_SLD >= BN || col < QSS_SLD)        // (short stride: columns past it belong to the next row)
            stile[mr * QSS_SLD + col] = __fadd_rn(__fmul_rn(ax, __fadd_rn(__fmul_rn(aq[j], (float)acc[i][j][e]), uu)), tqa[j]);
        }
      }
  }
  __syncthreads();
  // ---- phase 2: softmax + LSQ (ofq_softmax_lsq_fwd's arithmetic).  16 lanes per row, 4 rows per wave at a time: lane l of
  // a group owns the float4 chunks l, l + 16, l + 32, l + 48 of its row (every store instruction covers 256 contiguous
  // bytes per row), the three row reductions are 16-lane butterflies, and four independent rows per wave keep the
  // dependent chain max -> exp -> sum -> divide -> quantise fed.
  // TAIL (the 198-token DeiT rows, N <= 208): the fourth chunk would be columns 192..255, i.e. two busy lanes and fourteen
  // that run 4 elements of padding each -- a quarter of this VALU-bound phase.  Instead lane l takes the ONE column
  // 192 + l; its exponential is handed back to the lane that owned it in the chunk layout for the row sum, so the sum is
  // formed in the same order and the probabilities stay bit-identical to ofq_softmax_lsq_fwd.  Row groups that lie entirely
  // below the matrix (the fourth 64-row tile holds rows 192..197) are skipped.
  constexpr int KF = TAIL ? KCH - 1 : KCH;                // full float4 chunks per lane
  const int n = p.N;
  const int lr = lane & 15, rg = lane >> 4;
  for (int it = 0; it < 4; ++it) {
    if (m0 + wid * 16 + it * 4 >= p.M) continue;           // wave-uniform: none of the four rows exists
    const int mr = wid * 16 + it * 4 + rg;
    const bool rok = (m0 + mr) < p.M;
    const int64_t R = (int64_t)gby * p.M + min(m0 + mr, p.M - 1);      // row of the (B, H, N) x ld matrices
    const int ct = 64 * KF + lr;                            // the tail column of this lane
    float t[KCH][4];
    float tt = -INFINITY;
#pragma unroll
    for (int k = 0; k < KF; ++k) {
      const int c0 = 4 * lr + 64 * k;
      const float4 vin = *reinterpret_cast<const float4*>(stile + mr * QSS_SLD + c0);
      float4 ain = make_float4(0.f, 0.f, 0.f, 0.f);
      if (q.addend && c0 < q.ld) ain = *reinterpret_cast<const float4*>(q.addend + (((R / q.S) % q.add_period) * q.S + (R % q.S)) * q.ld + c0);
      const float vv[4] = {vin.x, vin.y, vin.z, vin.w}, aa[4] = {ain.x, ain.y, ain.z, ain.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        t[k][e] = -INFINITY;
        if (c0 + e < n) {
          t[k][e] = __fmul_rn(vv[e], q.alpha);
          if (q.addend) t[k][e] = __fadd_rn(t[k][e], aa[e]);
        }
      }
    }
    if (TAIL && ct < n) {
      tt = __fmul_rn(stile[mr * QSS_SLD + ct], q.alpha);
      if (q.addend) tt = __fadd_rn(tt, q.addend[(((R / q.S) % q.add_period) * q.S + (R % q.S)) * q.ld + ct]);
    }
    float m = TAIL ? tt : -INFINITY;
#pragma unroll
    for (int k = 0; k < KF; ++k) m = fmaxf(m, fmaxf(fmaxf(t[k][0], t[k][1]), fmaxf(t[k][2], t[k][3])));
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < KF; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        t[k][e] = (4 * lr + 64 * k + e < n) ? expf(t[k][e] - m) : 0.f;
        sum += t[k][e];
      }
    if (TAIL) {
      tt = ct < n ? expf(tt - m) : 0.f;
      // lane l < 4 owned columns 192 + 4 l .. + 3 in the chunk layout: their exponentials join its partial sum in that order
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = __shfl(tt, (lane & 48) + ((4 * lr + e) & 15), 64);
        sum += lr < 4 ? v : 0.f;
      }
    }
    sum = ofq_group_sum<16>(sum);
    const float a = ofq_lsq_eff_scale(q.sm_s[R % q.S], q.sm_gscale);
    // p = e / sum and the level rint(clamp(p / a)) without the per-element IEEE division sequences: one exact reciprocal of
    // the row's sum and of its step, then ofq_div_by_rcp (correctly rounded quotient) and ofq_lsq_level_rcp (exact level
    // unless the product sits next to a half-integer: a group that flags one redoes its 16 levels with the division)
    const float rsum = __fdiv_rn(1.f, sum), ra = __fdiv_rn(1.f, a);
    const float half_m_tol = 0.5f - ofq_lsq_level_tol(0.f, q.hi);
    float qsum = 0.f;
    float pr[KCH][4], qq[KCH][4];
    float prt = 0.f, qqt = 0.f;
    float dmax = 0.f;
#pragma unroll
    for (int k = 0; k < KF; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pr[k][e] = ofq_div_by_rcp(t[k][e], sum, rsum);
        qq[k][e] = ofq_lsq_level_rcp_d(pr[k][e], ra, 0.f, q.hi, dmax);
      }
    if (TAIL) {
      prt = ofq_div_by_rcp(tt, sum, rsum);
      qqt = ofq_lsq_level_rcp_d(prt, ra, 0.f, q.hi, dmax);
    }
    if (__builtin_amdgcn_ballot_w64(!(dmax < half_m_tol)) != 0ull) {
#pragma unroll
      for (int k = 0; k < KF; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) qq[k][e] = ofq_lsq_level_exact(pr[k][e], a, 0.f, q.hi);
      if (TAIL) qqt = ofq_lsq_level_exact(prt, a, 0.f, q.hi);
    }
#pragma unroll
    for (int k = 0; k < KF; ++k) {
      const int c0 = 4 * lr + 64 * k;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (c0 + e >= n) { pr[k][e] = 0.f; qq[k][e] = 0.f; }
        qsum += qq[k][e];
      }
      if (rok && c0 < q.ld) {
        *reinterpret_cast<float4*>(q.prob + R * q.ld + c0) = make_float4(pr[k][0], pr[k][1], pr[k][2], pr[k][3]);
        *reinterpret_cast<uchar4*>(q.codes + R * q.ld + c0) = make_uchar4((unsigned char)(int)qq[k][0], (unsigned char)(int)qq[k][1],
                                                                            (unsigned char)(int)qq[k][2], (unsigned char)(int)qq[k][3]);
      }
    }
    if (TAIL) {
      if (ct >= n) { prt = 0.f; qqt = 0.f; }
      qsum += qqt;
      if (rok && ct < q.ld) {
        q.prob[R * q.ld + ct] = prt;
        q.codes[R * q.ld + ct] = (unsigned char)(int)qqt;
      }
    }
    qsum = ofq_group_sum<16>(qsum);
    if (lr == 0 && rok) q.rowsum[R] = qsum;
  }
}

// QKR (plain = 0): A = x codes [B][N][CK], B = qkx codes [B][N][H][CK], sq one step per (token, head);
// plain (plain = 1): A = q codes, B = k codes [B][N][H*CK] with CK = head dim, sq one step per token.
extern "C" int ofq_qattn_scores_softmax_i8(const int8_t* acodes, const int8_t* bcodes, const float* sa, float gscale_a,
                                           const float* sb, float gscale_b, const float* u, const float* tq, const float* z,
                                           int plain, const float* sm_s, float sm_gscale, float alpha, int hi,
                                           const float* addend, int64_t add_period, float* prob, uint8_t* codes, float* rowsum,
                                           int64_t B, int64_t H, int64_t N, int64_t CK, int64_t ld, ofq_stream_t stream) {
  if (!acodes || !bcodes || !sa || !sb || !u || !tq || !z || !sm_s || !prob || !codes || !rowsum) return OFQ_EINVAL;
  if (B <= 0 || H <= 0 || N <= 0 || N > 256 || ld < N || ld > 256 || (ld & 3) || (CK & 15) || hi < 1 || hi > 255) return OFQ_EINVAL;
  if (addend && add_period <= 0) return OFQ_EINVAL;
  QSsArgs q = {};
  QGemmArgs& a = q.g;
  a.A = acodes; a.B = bcodes; a.s = sa; a.s2 = sb; a.u = u; a.tq = tq; a.z = z;
  if (plain) {
    const int64_t C = H * CK;
    a.lda = C; a.ldb = C; a.sA0 = N * C; a.sA1 = CK; a.sB0 = N * C; a.sB1 = CK; a.s2s0 = 1; a.s2s1 = 0;
  } else {
    a.lda = CK; a.ldb = H * CK; a.sA0 = N * CK; a.sA1 = 0; a.sB0 = N * H * CK; a.sB1 = CK; a.s2s0 = (int)H; a.s2s1 = 1;
  }
  a.M = (int)N; a.N = (int)N; a.K = (int)CK; a.S = (int)N; a.nb1 = (int)H; a.gscale = gscale_a; a.gscale2 = gscale_b;
  a.tiles_m = (int)ceil_div(N, 64); a.tiles_n = 1;
  q.sm_s = sm_s; q.addend = addend; q.prob = prob; q.codes = codes; q.rowsum = rowsum; q.ld = ld;
  q.add_period = addend ? add_period : 1; q.S = (int)N; q.sm_gscale = sm_gscale; q.alpha = alpha; q.hi = (float)hi;
  if (N <= 64 && ld <= 64)          // Swin windows: one (window, head) per workgroup, 64-key panel (S tile rows of 68 floats)
    hipLaunchKernelGGL((qattn_scores_softmax_kernel<68, 1>), dim3((unsigned)a.tiles_m, (unsigned)(B * H)), dim3(256), 0, (hipStream_t)stream, q);
  else if (N <= 200 && ld <= 208)     // (DeiT: 198 tokens, ld 208) three full chunks + a one-column tail per lane
    hipLaunchKernelGGL((qattn_scores_softmax_kernel<200, 4, true>), dim3((unsigned)a.tiles_m, (unsigned)(B * H)), dim3(256), 0, (hipStream_t)stream, q);
  else if (N <= 200)
    hipLaunchKernelGGL(qattn_scores_softmax_kernel<200>, dim3((unsigned)a.tiles_m, (unsigned)(B * H)), dim3(256), 0, (hipStream_t)stream, q);
  else
    hipLaunchKernelGGL(qattn_scores_softmax_kernel<260>, dim3((unsigned)a.tiles_m, (unsigned)(B * H)), dim3(256), 0, (hipStream_t)stream, q);
  OFQ_LAUNCH_CHECK();
  return 0;
}

// =====================================================================================================================
// dP GEMM + softmax-LSQ backward in one kernel (autograd of attention.py:213-219: attn @ v, the unsigned LSQ of the
// probabilities, the softmax, the scale).  Unfused, dP[b,h,n,m] = dO[b,n,h,:] . V_hat[b,m,h,:] is written (122 MB per
// DeiT-S block at 128 images), read back by the softmax backward next to the saved probabilities, and overwritten with dS.
// Here a workgroup owns 64 query rows x all keys (<= 256) of one (batch, head):
//   phase 1  dP tile = (dO * av_eff) split into three bf16 planes  x  the v codes (bf16), K = head dim in chunks of 32,
//            four waves side by side (64 x 64 each); w[n] = dO[n,:] . bav (the offset term of V_hat, a row constant of
//            dP) is accumulated from the same registers.  dP + w -> LDS tile [64][QSS_SLD] (over the staging buffers)
//   phase 2  16 lanes per row, four rows per wave at a time: ofq_softmax_lsq_bwd's arithmetic on (dP row, saved prob row):
//            LSQ backward (ofq_lsq_bwd_fast, exact redo for a group that flags a tie / range edge), row dot, softmax
//            backward, * alpha; dS leaves with 16-byte stores, the step-gradient partial of the row with one float.
// HBM: dO, the v codes and prob in, dS out -- dP never exists in memory (-244 MB per block), and the row dots w
// (ofq_rowdot_f32_seg) are a by-product.
struct QDpArgs {
  const float* dO; const int8_t* vcodes; const float* sv; const float* bav;
  const float* prob; const float* sm_s; float* dS; float* rowpart; float* ds_rowsum;
  unsigned* amax;        // optional: bits of max |dS| (see ofq_amax_publish; the pad columns are written as zeros)
  int64_t ld;
  int B, H, N, d, C, S;
  float gscale_v, sm_gscale, alpha, hi;
};

template <int QSS_SLD, bool TAIL = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(QSS_SLD <= 200 ? 3 : 2, QSS_SLD <= 200 ? 3 : 2)))
void qattn_dp_softmax_bwd_kernel(QDpArgs q) {
  constexpr int BM = 64, BN = 256, NS = 3, KCH = 4;
  constexpr int KF = TAIL ? KCH - 1 : KCH;                // full float4 chunks per lane; TAIL: + the one column 192 + lane
                                                          // (see qattn_scores_softmax_kernel)
  constexpr int PLANE = BM * QBS_LD;                      // one bf16 plane of the dO tile: 64 rows x 32 k (+ pad)
  constexpr int STAGE = NS * PLANE + BN * QBS_LD;
  constexpr int TILE_BYTES = BM * QSS_SLD * 4 + 256;
  constexpr int SBYTES = STAGE > TILE_BYTES ? STAGE : TILE_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[SBYTES + BM * 4];
  float* stile = reinterpret_cast<float*>(smem);
  float* row_w = reinterpret_cast<float*>(smem + SBYTES);
  // the row tiles of one (batch, head) share its key panel, the heads of one batch element its query rows: keep them on
  // one XCD, next to each other in time (hardware order alone spreads consecutive workgroups over the 8 L2s)
  int tm, gby;
  xcd_remap_grid(tm, gby);
  const int m0 = tm * BM;
  const int b = gby / q.H, h = gby % q.H;
  const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int N = q.N, d = q.d, C = q.C;
  const float* dOb = q.dO + ((int64_t)b * N) * C + h * d;
  const int8_t* Vb = q.vcodes + ((int64_t)b * N) * C + h * d;
  const float* svh = q.sv + h * d;
  const float* bavh = q.bav ? q.bav + h * d : nullptr;
  const int nkc = (d + QBS_BK - 1) / QBS_BK;

  // the saved probabilities of this thread's phase-2 rows are requested first: they arrive under phase 1
  const int lr = lane & 15, rg = lane >> 4;
  float4 pin[4][KF];
  float pint[4] = {0.f, 0.f, 0.f, 0.f};
  const int ct = 64 * KF + lr;                             // this lane's tail column
  auto load_prob = [&](auto IT0) {                        // rows IT0, IT0 + 1 of this thread's four phase-2 rows
    constexpr int it0 = decltype(IT0)::value;
#pragma unroll
    for (int it = it0; it < it0 + 2; ++it) {
      const int64_t R = (int64_t)gby * N + min(m0 + wn * 16 + it * 4 + rg, N - 1);
#pragma unroll
      for (int k = 0; k < KF; ++k) {
        const int c0 = 4 * lr + 64 * k;
        pin[it][k] = *reinterpret_cast<const float4*>(q.prob + R * q.ld + (c0 < q.ld ? c0 : 0));
      }
      if (TAIL) pint[it] = q.prob[R * q.ld + (ct < q.ld ? ct : 0)];
    }
  };
  load_prob(std::integral_constant<int, 0>());            // (the other two rows once the accumulators are out of the way)
  f32x16q acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  float wpart[2] = {0.f, 0.f};
  const int kqa = (tid & 7) * 4;
  for (int kc = 0; kc < nkc; ++kc) {
    const int k0 = kc * QBS_BK;
    // ---- stage the dO chunk (64 rows x 32 k: two float4 per thread), scaled along k by the effective v step and split
    {
      const bool kin = (k0 + kqa) < d;                    // d % 4 == 0: chunks are all-in or all-out
      const int kk = kin ? k0 + kqa : 0;
      const f32x4v sraw = *reinterpret_cast<const f32x4v*>(svh + kk);
      f32x4v braw = {0.f, 0.f, 0.f, 0.f};
      if (bavh) braw = *reinterpret_cast<const f32x4v*>(bavh + kk);
      f32x4v ra[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = (tid + 256 * i) >> 3;
        ra[i] = *reinterpret_cast<const f32x4v*>(dOb + (int64_t)min(m0 + row, N - 1) * C + kk);
      }
      float ks[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float t = q.gscale_v > 0.f ? ofq_lsq_eff_scale(sraw[e], q.gscale_v) : sraw[e];
        ks[e] = kin ? t : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = (tid + 256 * i) >> 3;
        const float z = (m0 + row) < N ? 1.f : 0.f;
        if (kin) wpart[i] += z * ((ra[i][0] * braw[0] + ra[i][1] * braw[1]) + (ra[i][2] * braw[2] + ra[i][3] * braw[3]));
        const f32x2v k01 = {ks[0] * z, ks[1] * z}, k23 = {ks[2] * z, ks[3] * z};
        const f32x2v a01 = {ra[i][0], ra[i][1]}, a23 = {ra[i][2], ra[i][3]};
        unsigned lo[NS], hi[NS];
        split_pair_bf16<NS>(a01 * k01, lo);
        split_pair_bf16<NS>(a23 * k23, hi);
#pragma unroll
        for (int p = 0; p < NS; ++p) {
          uint2 w;
          w.x = lo[p];
          w.y = hi[p];
          *reinterpret_cast<uint2*>(&smem[p * PLANE + row * QBS_LD + kqa * 2]) = w;
        }
      }
    }
    // ---- stage the v codes (256 keys x 32 k int8 -> bf16: two 16-code halves per thread)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int f = tid + 256 * i;
      const int row = f >> 1, half = f & 1;
      const bool ok = row < N && (k0 + half * 16) < d;    // d % 16 == 0
      const i32x4 c = *reinterpret_cast<const i32x4*>(Vb + (int64_t)min(row, N - 1) * C + (ok ? k0 + half * 16 : 0));
      unsigned w[8];
#pragma unroll
      for (int u = 0; u < 4; ++u) valu_cvt4_i8_bf16(ok ? (unsigned)c[u] : 0u, w[2 * u], w[2 * u + 1]);
      unsigned char* dst = &smem[NS * PLANE + row * QBS_LD + half * 32];
      *reinterpret_cast<uint4*>(dst) = make_uint4(w[0], w[1], w[2], w[3]);
      *reinterpret_cast<uint4*>(dst + 16) = make_uint4(w[4], w[5], w[6], w[7]);
    }
    __syncthreads();
    const unsigned char* a = &smem[l31 * QBS_LD + lh * 16];
    const unsigned char* bb = &smem[NS * PLANE + (wn * 64 + l31) * QBS_LD + lh * 16];
#pragma unroll
    for (int ks2 = 0; ks2 < QBS_BK / 16; ++ks2) {
      bf16x8 bv[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) bv[j] = *reinterpret_cast<const bf16x8*>(bb + j * 32 * QBS_LD + ks2 * 32);
#pragma unroll
      for (int p = 0; p < NS; ++p) {
        bf16x8 av[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) av[i] = *reinterpret_cast<const bf16x8*>(a + p * PLANE + i * 32 * QBS_LD + ks2 * 32);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();                                       // the staging buffers are rewritten (next chunk / the dP tile)
  }
  // w[row]: the eight threads of a row (consecutive lanes) hold its partial dots
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float w = wpart[i];
    w += __shfl_xor(w, 1, 64);
    w += __shfl_xor(w, 2, 64);
    w += __shfl_xor(w, 4, 64);
    if ((tid & 7) == 0) row_w[(tid + 256 * i) >> 3] = w;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int mr = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
      const float ww = row_w[mr];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = wn * 64 + j * 32 + l31;
        if (QSS_SLD >= BN || col < QSS_SLD) stile[mr * QSS_SLD + col] = acc[i][j][e] + ww;
      }
    }
  load_prob(std::integral_constant<int, 2>());
  __syncthreads();
  // ---- phase 2: LSQ backward + softmax backward per row (ofq_softmax_lsq_bwd's arithmetic)
  const float tol = ofq_lsq_level_tol(0.f, q.hi), half_m_tol = 0.5f - tol;
  float omax = 0.f;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    if (m0 + wn * 16 + it * 4 >= N) continue;              // wave-uniform: none of the four rows exists
    const int mr = wn * 16 + it * 4 + rg;
    const bool rok = (m0 + mr) < N;
    const int row = min(m0 + mr, N - 1);
    const int64_t R = (int64_t)gby * N + row;
    const float a = ofq_lsq_eff_scale(q.sm_s[row], q.sm_gscale);          // S == N: the step of query token `row`
    const float ra = __fdiv_rn(1.f, a);
    constexpr int NE = 4 * KF + (TAIL ? 1 : 0);
    float p[NE], g[NE], dq[NE];
    float rowds = 0.f, dot = 0.f;
    OfqLsqFlags fl;                                        // extrema instead of per-element booleans (common.h)
#pragma unroll
    for (int k = 0; k < KF; ++k) {
      const int c0 = 4 * lr + 64 * k;
      const float4 gin = *reinterpret_cast<const float4*>(stile + mr * QSS_SLD + c0);
      const float pp[4] = {pin[it][k].x, pin[it][k].y, pin[it][k].z, pin[it][k].w}, gg[4] = {gin.x, gin.y, gin.z, gin.w};
      const bool full = 64 * (k + 1) <= N;                  // wave-uniform
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool in = full || (c0 + e) < N;
        p[4 * k + e] = in ? pp[e] : 0.f;
        g[4 * k + e] = in ? gg[e] : 0.f;
      }
    }
    if (TAIL) {
      const bool in = ct < N;
      p[NE - 1] = in ? pint[it] : 0.f;
      g[NE - 1] = in ? stile[mr * QSS_SLD + ct] : 0.f;
    }
    // ofq_lsq_bwd_fast specialised for an unsigned quantiser on a non-negative input (a probability): v = p * ra >= 0 = lo
    // holds exactly, in the division form as well, so closeness to the LOWER edge decides nothing (the generic routine
    // flags it -- and a softmax row is full of probabilities below 1e-6 and of padding zeros: every group would take the
    // exact redo)
#pragma unroll
    for (int x = 0; x < NE; ++x) {
      const float v = __fmul_rn(p[x], ra);
      const float u = fminf(v, q.hi);
      const float qq = rintf(u);
      const bool inr = v <= q.hi;
      fl.dmax = fmaxf(fl.dmax, fabsf(__fsub_rn(u, qq)));
      fl.emin = fminf(fl.emin, fabsf(__fsub_rn(v, q.hi)));
      const float t = __fmul_rn(g[x], a);
      float d0 = __fmul_rn(t, ra);
      d0 = __fmaf_rn(__fmaf_rn(-a, d0, t), ra, d0);
      d0 = __fmaf_rn(__fmaf_rn(-a, d0, t), ra, d0);
      const unsigned tb = __float_as_uint(t) & 0x7fffffffu;
      fl.umin = min(fl.umin, tb - 1u);
      fl.umax = max(fl.umax, tb);
      dq[x] = inr ? d0 : 0.f;
      rowds += g[x] * (inr ? (qq - v) : qq);
      dot += dq[x] * p[x];
    }
    if (__builtin_amdgcn_ballot_w64(ofq_lsq_flags_risky(fl, half_m_tol, tol)) != 0ull) {       // rare: a value next to a rounding tie / the upper edge
      rowds = 0.f;
      dot = 0.f;
#pragma unroll
      for (int x = 0; x < NE; ++x) {
        float dsc;
        ofq_lsq_bwd_exact(p[x], g[x], a, 0.f, q.hi, dq[x], dsc);
        rowds += dsc;
        dot += dq[x] * p[x];
      }
    }
    rowds = ofq_group_sum<16>(rowds);
    dot = ofq_group_sum<16>(dot);
    if (lr == 0 && rok) q.rowpart[R] = rowds;
    float rsum = 0.f;
#pragma unroll
    for (int k = 0; k < KF; ++k) {
      const int c0 = 4 * lr + 64 * k;
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = (dq[4 * k + e] - dot) * p[4 * k + e] * q.alpha;              // (padding columns: p = 0)
        rsum += o[e];
      }
      if (rok && c0 < q.ld) {
        *reinterpret_cast<float4*>(q.dS + R * q.ld + c0) = make_float4(o[0], o[1], o[2], o[3]);
        omax = ofq_absmax4(omax, o[0], o[1], o[2], o[3]);
      }
    }
    if (TAIL) {
      const float o = (dq[NE - 1] - dot) * p[NE - 1] * q.alpha;
      rsum += o;
      if (rok && ct < q.ld) { q.dS[R * q.ld + ct] = o; omax = fmaxf(omax, fabsf(o)); }
    }
    if (q.ds_rowsum) {
      rsum = ofq_group_sum<16>(rsum);
      if (lr == 0 && rok) q.ds_rowsum[R] = rsum;
    }
  }
  if (q.amax) ofq_amax_publish(q.amax, omax);
}

extern "C" size_t ofq_qattn_dp_softmax_bwd_ws_bytes(int64_t B, int64_t H, int64_t N) {
  return (size_t)(B * H * N) * sizeof(float) + 256;
}

// dS = d(loss)/d(scores) from dO = d(loss)/d(attention output): dP = dO . V_hat^T per head, then the backward of
// P_hat = LSQ_unsigned(softmax(alpha * S)) on the saved probabilities.  ds[S] = the softmax quantiser's step gradient
// (sum over batch, heads of the row partials, times sm_gscale), ds_rowsum (optional) = row sums of dS.
// N <= 256 keys, ld <= 256, ld % 4 == 0, d % 16 == 0, C % 4 == 0.
extern "C" int ofq_qattn_dp_softmax_bwd(const float* dO, const int8_t* vcodes, const float* sv, float gscale_v, const float* bav,
                                        const float* prob, const float* sm_s, float sm_gscale, float alpha, int hi, float* dS,
                                        float* ds, float* ds_rowsum, int64_t B, int64_t H, int64_t N, int64_t d, int64_t ld,
                                        void* ws, size_t ws_bytes, void* amax_out, ofq_stream_t stream) {
  if (!dO || !vcodes || !sv || !prob || !sm_s || !dS || !ws || B <= 0 || H <= 0 || N <= 0) return OFQ_EINVAL;
  if (N > 256 || ld < N || ld > 256 || (ld & 3) || (d & 15) || d <= 0 || hi < 1 || hi > 255) return OFQ_EINVAL;
  const int64_t C = H * d;
  if (!al16(dO) || !al16(vcodes) || !al16(sv) || (bav && !al16(bav)) || !al16(prob) || !al16(dS)) return OFQ_EINVAL;
  if (ws_bytes < ofq_qattn_dp_softmax_bwd_ws_bytes(B, H, N)) return OFQ_ENOWS;
  QDpArgs q = {};
  q.dO = dO; q.vcodes = vcodes; q.sv = sv; q.bav = bav; q.prob = prob; q.sm_s = sm_s; q.dS = dS; q.rowpart = (float*)ws;
  q.ds_rowsum = ds_rowsum; q.amax = (unsigned*)amax_out; q.ld = ld; q.B = (int)B; q.H = (int)H; q.N = (int)N; q.d = (int)d; q.C = (int)C; q.S = (int)N;
  q.gscale_v = gscale_v; q.sm_gscale = sm_gscale; q.alpha = alpha; q.hi = (float)hi;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)ceil_div(N, 64), (unsigned)(B * H));
  if (N <= 200 && ld <= 208) hipLaunchKernelGGL((qattn_dp_softmax_bwd_kernel<200, true>), grid, dim3(256), 0, st, q);
  else if (N <= 200) hipLaunchKernelGGL(qattn_dp_softmax_bwd_kernel<200>, grid, dim3(256), 0, st, q);     // 51 KB: three per CU
  else hipLaunchKernelGGL(qattn_dp_softmax_bwd_kernel<260>, grid, dim3(256), 0, st, q);
  OFQ_LAUNCH_CHECK();
  if (ds) {
    SumJobs jobs = {};
    jobs.j[0] = {(const float*)ws, ds, N, (B * H * N) / N, N, 1, sm_gscale, 0, 0};
    strided_sum_launch(jobs, N, 1, st);
    OFQ_LAUNCH_CHECK();
  }
  return 0;
}
