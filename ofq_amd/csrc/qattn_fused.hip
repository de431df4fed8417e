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
template <int QSS_SLD, int CB = 4>
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
  const int tm = blockIdx.x, gby = blockIdx.y;
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
  // dependent chain max -> exp -> sum -> divide -> quantise fed
  const int n = p.N;
  const int lr = lane & 15, rg = lane >> 4;
  for (int it = 0; it < 4; ++it) {
    const int mr = wid * 16 + it * 4 + rg;
    const bool rok = (m0 + mr) < p.M;
    const int64_t R = (int64_t)gby * p.M + min(m0 + mr, p.M - 1);      // row of the (B, H, N) x ld matrices
    float t[KCH][4];
#pragma unroll
    for (int k = 0; k < KCH; ++k) {
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
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < KCH; ++k) m = fmaxf(m, fmaxf(fmaxf(t[k][0], t[k][1]), fmaxf(t[k][2], t[k][3])));
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < KCH; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        t[k][e] = (4 * lr + 64 * k + e < n) ? expf(t[k][e] - m) : 0.f;
        sum += t[k][e];
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
    bool risky = false;
#pragma unroll
    for (int k = 0; k < KCH; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pr[k][e] = ofq_div_by_rcp(t[k][e], sum, rsum);
        qq[k][e] = ofq_lsq_level_rcp(pr[k][e], ra, 0.f, q.hi, half_m_tol, risky);
      }
    if (__builtin_amdgcn_ballot_w64(risky) != 0ull) {
#pragma unroll
      for (int k = 0; k < KCH; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) qq[k][e] = ofq_lsq_level_exact(pr[k][e], a, 0.f, q.hi);
    }
#pragma unroll
    for (int k = 0; k < KCH; ++k) {
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
  else if (N <= 200)
    hipLaunchKernelGGL(qattn_scores_softmax_kernel<200>, dim3((unsigned)a.tiles_m, (unsigned)(B * H)), dim3(256), 0, (hipStream_t)stream, q);
  else
    hipLaunchKernelGGL(qattn_scores_softmax_kernel<260>, dim3((unsigned)a.tiles_m, (unsigned)(B * H)), dim3(256), 0, (hipStream_t)stream, q);
  OFQ_LAUNCH_CHECK();
  return 0;
}
