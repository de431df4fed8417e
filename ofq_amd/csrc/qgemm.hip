// Exact low-precision GEMMs for the fake-quantised linear layers (reference: F.linear qlinear.py:69 and its
// autograd products).  Fake-quant operands are scale x small integer (+ a per-channel offset), so the scales are
// factored out of the contraction and the matrix cores run on the integer codes, which int8 / bf16 represent
// exactly; accumulation (i32, or fp32 over integers < 2^24) is exact as well.
//
//  ofq_qgemm_i8_nt    forward   y[m,n] = cs[n] * (a_eff[m % S] * sum_k qa[m,k] * qw[n,k] + r[n]) + bias[n]
//                     qa = LSQ codes of the input, qw = 2L+1 (StatsQ) or the LSQ level of the weight, a_eff the
//                     effective LSQ step, cs[n] = s[n]/(2*nlev) (StatsQ) or the weight step, r[n] = sum_k baft[k] qw[n,k]
//                     (the post-quantiser offset's contribution).  v_mfma_i32_32x32x32_i8, exact.
//  ofq_qgemm_bf16s_nt backward  dx[m,n] = alpha * sum_k (dy[m,k] * ks[k]) * qwT[n,k]
//                     dy fp32 is scaled along k by the weight scale and split into three bf16 pieces
//                     (dy*ks = hi + mid + lo exactly: 3 x 8 significand bits), qwT are the weight codes as bf16;
//                     three v_mfma_f32_32x32x16_bf16 per k-step give the fp32-exact product at 3/16 of the fp32-MFMA cost.
// The 4-wave kernels: 256 threads = 2x2 waves, 128x128 tile, operands K-contiguous ("NT"), branch-free staging with
// clamped addresses, LDS rows padded by 16 B so that ds_read_b128 fragments are conflict-free.  The linear layers'
// gradient GEMMs use the 8-wave "wide" kernels further down (128 x 384 tiles, double-buffered LDS, LDS-only barriers,
// two-step register prefetch): the fp32 -> 3 x bf16 split of a dY panel is paid once per 384 output columns.
// Workgroup order is XCD-aware over the whole (tile, batch) grid (xcd_remap_grid).  Epilogue rule learnt the hard way:
// no load behind a per-element condition (one memory round trip each) -- per-row terms go through LDS, old values
// for C += ... are fetched unconditionally on clamped addresses.
#include "common.h"

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef float f32x16q __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

static bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
__device__ __forceinline__ bool al16_dev(const void* p) { return (((uintptr_t)p) & 15) == 0; }

struct QGemmArgs {
  const void* A; const void* B; float* C;
  const float* bias;     // [N] optional
  const float* cs;       // i8 linear: column scale [N]
  const float* r;        // i8 linear: offset term [N] (optional)
  const float* s;        // i8: LSQ step vector of the rows [S];  bf16s: k-scale ks (optional)
  const unsigned* amax;  // bf16s, two-plane fp16 form: bits of an upper bound of max |A| (device word)
  unsigned* amax_out;    // i8 recompute backward, optional: bits of max |dy| written (ofq_amax_publish)
  // attention epilogues (i8) / extras (bf16s)
  const float* s2;       // second LSQ step vector (columns: qkx steps [N*nb1] / v steps [C])
  const float* u;        // scores: [nb0][M][nb1]      bf16s-nt: per-row addend [nb0][M][nb1]
  const float* tq;       // scores: [nb0][N][nb1]
  const float* z;        // scores: [nb1]              pv: column offset vector bav [C]
  const float* rp;       // pv: row sums of the P codes [nb0][nb1][M]
  int64_t lda, ldb, ldc;
  int64_t sA0, sA1, sB0, sB1, sC0, sC1;   // batch strides (elements)
  int64_t sK1;           // bf16s: offset of the k-scale vector per inner batch index b1
  int64_t sBp;           // bf16s with a plane-split fp32 B operand: elements between consecutive bf16 planes
  int M, N, K, S, nb1;
  int s2s0, s2s1;        // scores: the column step is s2[n * s2s0 + b1 * s2s1] (QKR: one per (token, head); plain: per token)
  int tiles_m, tiles_n, accumulate, b_is_i8;
  float gscale, gscale2, alpha;
  // i8 linear, optional by-product: the int8 codes of the NEXT layer's input quantiser applied to this output,
  //   q = LSQ([gelu](y) + qb4[n]; step qs[m % qS]) -- what ofq_lsq_fwd would compute from the stored y, bit for bit
  //   per-row step (qcolmode 0): index (m * qrowmul + n0 / qcoldiv) % qS -- qrowmul > 1 when the output row holds
  //   qrowmul quantiser rows side by side (qkx: heads); per-column step (qcolmode 1): index n
  // bf16s-nt wide, optional fused LSQ backward of the layer's input quantiser (see qgemm_bf16s_nt_wide_kernel<NJ, true>)
  const float* lx; const float* ls; const float* lb4; float* lrow; float* lcol;
  int64_t ldlx;
  int lS, lgelu;
  float lgscale, llo, lhi;
  int8_t* qout; const float* qs; const float* qb4;
  // i8 recompute backward, fused attention form (qgemm_i8_lsqbwd_kernel<1, false, true>): the incoming gradient is not read
  // but formed in the kernel, gy[b N + m, h C + c] = sum_n dS[b, h, n, m] * (a_eff[n] * A[b N + n, c] + z[c])
  const float* dS; int64_t ldS; int dH, dN;
  int64_t ldq;
  int qS, qgelu, qrowmul, qcoldiv, qcolmode;
  float qgscale, qlo, qhi;
};

// Hardware dispatch order is x-fastest and workgroup w lands on XCD w % 8 (each XCD has its own L2).  Give each XCD a
// contiguous run of the (batch, tile) sequence, so that workgroups which share operands -- the tiles of one batch
// element, the column tiles of one row panel -- run on the same XCD at about the same time and hit its L2 instead
// of fetching the shared operand once per XCD (the attention dxq GEMM fetched 3x its algorithmic bytes before this).
__device__ __forceinline__ void xcd_remap_grid(int& bx, int& by) {
  const int nx = gridDim.x;
  const int total = nx * gridDim.y;
  int L = blockIdx.y * nx + blockIdx.x;
  const int q = total >> 3, r = total & 7;
  const int xcd = L & 7, loc = L >> 3;
  L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  bx = L % nx;
  by = L / nx;
}

__device__ __forceinline__ void qgemm_tile_id(const QGemmArgs& p, int& tm, int& tn, int& by) {
  int tile;
  xcd_remap_grid(tile, by);
  tm = tile / p.tiles_n;
  tn = tile % p.tiles_n;
}

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

// ------------------------------------------------------------------------------------------------ bf16-split backward
#define QBS_BK 32                      // k per stage
#define QBS_LD (QBS_BK * 2 + 16)       // padded LDS row in bytes (bf16)

__device__ __forceinline__ unsigned pack_hi16(float lo_elem, float hi_elem) {
  // two fp32 whose low 16 bits are zero -> one dword of two bf16 (element order: lo_elem first)
  return (__float_as_uint(lo_elem) >> 16) | (__float_as_uint(hi_elem) & 0xffff0000u);
}
__device__ __forceinline__ float trunc_bf16(float x) { return __uint_as_float(__float_as_uint(x) & 0xffff0000u); }

typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
// three-way bf16 split of two neighbouring fp32 (one packed dword per plane).  Written on explicit 2-vectors so that the
// pairs are the naturally aligned (x,y) / (z,w) halves of the loaded float4: left to the SLP vectoriser the pairing was
// (x,w) / (y,z), whose register shuffles (v_mov of freshly loaded registers) forced early s_waitcnt vmcnt on the prefetch
template <int NS>
__device__ __forceinline__ void split_pair_bf16(f32x2v x, unsigned (&out)[NS]) {
  f32x2v rem = x;
#pragma unroll
  for (int q = 0; q < NS; ++q) {
    const u32x2v hb = __builtin_bit_cast(u32x2v, rem) & 0xffff0000u;
    out[q] = (hb.x >> 16) | hb.y;
    rem = rem - __builtin_bit_cast(f32x2v, hb);
  }
}

// ---- staging work interleaved into the MFMA stream -------------------------------------------------------------------
// tools/probe/filler_probe.hip: single-issue VALU instructions that follow an MFMA in the SAME wave's program order hide
// in its shadow -- with two waves per SIMD the first two per MFMA are free and each further one costs ~2.2 cycles
// instead of 4 -- while one v_pk_mul_f32 there costs ~10 cycles.  So the k-loops below place a few scalar staging
// instructions behind every MFMA (pinned with sched_barrier) instead of running the staging as a block after the MFMAs,
// and the split is written with these one-instruction wrappers so that the SLP vectoriser cannot re-pack it.
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

__device__ __forceinline__ float valu_mul(float a, float b) { float d; asm("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float valu_sub(float a, float b) { float d; asm("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float valu_add(float a, float b) { float d; asm("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float valu_hi16(float a) { float d; asm("v_and_b32 %0, 0xffff0000, %1" : "=v"(d) : "v"(a)); return d; }
__device__ __forceinline__ float valu_max(float a, float b) { float d; asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float valu_fma(float a, float b, float c) { float d; asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
// byte BYTE of w, sign-extended, as fp32 (one SDWA convert)
template <int BYTE>
__device__ __forceinline__ float valu_cvt_i8(unsigned w) {
  float d;
  if constexpr (BYTE == 0) asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0" : "=v"(d) : "v"(w));
  if constexpr (BYTE == 1) asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "=v"(d) : "v"(w));
  if constexpr (BYTE == 2) asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2" : "=v"(d) : "v"(w));
  if constexpr (BYTE == 3) asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3" : "=v"(d) : "v"(w));
  return d;
}
// four int8 codes of one dword -> two packed bf16 pairs (exact), one asm statement (4 SDWA converts + 2 v_perm_b32)
__device__ __forceinline__ void valu_cvt4_i8_bf16(unsigned w, unsigned& d01, unsigned& d23) {
  float t0, t1, t2, t3;
  asm("v_cvt_f32_i32_sdwa %2, sext(%6) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0\n\t"
      "v_cvt_f32_i32_sdwa %3, sext(%6) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n\t"
      "v_cvt_f32_i32_sdwa %4, sext(%6) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n\t"
      "v_cvt_f32_i32_sdwa %5, sext(%6) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3\n\t"
      "v_perm_b32 %0, %3, %2, %7\n\t"
      "v_perm_b32 %1, %5, %4, %7"
      : "=&v"(d01), "=&v"(d23), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
      : "v"(w), "s"(0x07060302u));
}
// ofq_lsq_eff_scale, instruction for instruction: a = max(s, 1e-5); t = a * g; (a - t) + t
__device__ __forceinline__ float valu_eff_scale(float s, float g) {
  float a, t, d;
  asm("v_max_f32 %0, 0x3727c5ac, %3\n\tv_mul_f32 %1, %0, %4\n\tv_sub_f32 %2, %0, %1\n\tv_add_f32 %2, %2, %1"
      : "=&v"(a), "=&v"(t), "=&v"(d) : "v"(s), "v"(g));
  return d;
}
// two-instruction steps of the split as ONE asm statement: between separate asm statements the hazard recogniser pads
// with s_nop (it cannot see what they are), which costs an issue slot each
__device__ __forceinline__ void valu_mul_hi16(float a, float b, float& x, float& p0) {        // x = a*b; p0 = hi16(x)
  asm("v_mul_f32 %0, %2, %3\n\tv_and_b32 %1, 0xffff0000, %0" : "=&v"(x), "=v"(p0) : "v"(a), "v"(b));
}
__device__ __forceinline__ void valu_sub_hi16(float a, float b, float& r, float& p) {          // r = a-b; p = hi16(r)
  asm("v_sub_f32 %0, %2, %3\n\tv_and_b32 %1, 0xffff0000, %0" : "=&v"(r), "=v"(p) : "v"(a), "v"(b));
}
// (lo_elem >> 16) | (hi_elem & 0xffff0000): the bf16 pair of two fp32 (truncating), one v_perm_b32
__device__ __forceinline__ void valu_pack3_hi16(const float (&a)[2], const float (&b)[2], const float (&c)[2], unsigned* d) {
  asm("v_perm_b32 %0, %4, %3, %9\n\tv_perm_b32 %1, %6, %5, %9\n\tv_perm_b32 %2, %8, %7, %9"
      : "=&v"(d[0]), "=&v"(d[1]), "=v"(d[2])
      : "v"(a[0]), "v"(a[1]), "v"(b[0]), "v"(b[1]), "v"(c[0]), "v"(c[1]), "s"(0x07060302u));
}
__device__ __forceinline__ unsigned valu_pack_hi16(float lo_elem, float hi_elem) {
  unsigned d;
  asm("v_perm_b32 %0, %1, %2, %3" : "=v"(d) : "v"(hi_elem), "v"(lo_elem), "s"(0x07060302u));
  return d;
}

// ---- two fp16 planes instead of three bf16 planes (round 5) -----------------------------------------------------------
// x (fp32, pre-scaled by a power of two so that the largest |x| of the launch sits below 2^15) = hi + lo + err with
// hi = rne_f16(x), lo = rne_f16(x - hi): x - hi is exact in fp32 (<= 13 significant bits), so |err| <= 2^-24 |x| as long
// as lo stays a normal fp16 (|x| >= 2^-3 of the scaled range), and <= 2^-25 absolute below that (fp16 denormals are kept by
// v_cvt_pk_f16_f32, v_fma_mix_f32 and v_mfma_f32_32x32x16_f16 alike: tools/probe/f16_mfma_probe.hip).  With the launch
// maximum scaled to [2^14, 2^15) that is: fp32 precision for every element within 2^-17 of the maximum, an absolute
// error of 2^-39 of the maximum below -- fp32-grade against the tensor scale, at two MFMAs per k-step instead of three.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <bool F16>
__device__ __forceinline__ f32x16q mfma_16b(bf16x8 a, bf16x8 b, f32x16q c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void valu_mul2(float a0, float b0, float a1, float b1, float& x0, float& x1) {
  asm("v_mul_f32 %0, %2, %3\n\tv_mul_f32 %1, %4, %5" : "=&v"(x0), "=v"(x1) : "v"(a0), "v"(b0), "v"(a1), "v"(b1));
}
__device__ __forceinline__ unsigned valu_cvt_pk_f16(float x0, float x1) {                       // (lo half: x0)
  unsigned d;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(d) : "v"(x0), "v"(x1));
  return d;
}
// r0 = x0 - f32(h.lo), r1 = x1 - f32(h.hi): one mixed-precision fma each, exact
__device__ __forceinline__ void valu_resid2_f16(unsigned h, float x0, float x1, float& r0, float& r1) {
  asm("v_fma_mix_f32 %0, %2, -1.0, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mix_f32 %1, %2, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "=&v"(r0), "=v"(r1) : "v"(h), "v"(x0), "v"(x1));
}
__device__ __forceinline__ void split2_f16(float x0, float x1, unsigned& hi, unsigned& lo) {
  float r0, r1;
  hi = valu_cvt_pk_f16(x0, x1);
  valu_resid2_f16(hi, x0, x1, r0, r1);
  lo = valu_cvt_pk_f16(r0, r1);
}
// four int8 codes of one dword -> two packed fp16 pairs (exact): bytes ^ 0x80 are the codes + 128 as unsigned, 0x64xx is
// the fp16 1024 + xx, minus 1152 gives the code.  c64 = 0x64646464 (a VGPR: v_perm_b32 may read one scalar operand only)
__device__ __forceinline__ void valu_cvt4_i8_f16(unsigned w, unsigned c64, unsigned& d01, unsigned& d23) {
  unsigned t;
  asm("v_xor_b32 %2, 0x80808080, %3\n\t"
      "v_perm_b32 %0, %4, %2, %5\n\t"
      "v_perm_b32 %1, %4, %2, %6\n\t"
      "v_pk_add_f16 %0, %0, %7\n\t"
      "v_pk_add_f16 %1, %1, %7"
      : "=&v"(d01), "=&v"(d23), "=&v"(t) : "v"(w), "v"(c64), "s"(0x04010400u), "s"(0x04030402u), "v"(0xE480E480u));
}
// The power of two that brings t (an upper bound of max |x| of the launch) into [2^14, 2^15), and its inverse; 1 for t = 0,
// inf or nan (a non-finite gradient stays non-finite through the product, as it would in fp32)
__device__ __forceinline__ void f16_plane_scale(float t, float& sE, float& inv_sE) {
  const int ex = (int)((__float_as_uint(t) >> 23) & 0xffu) - 127;
  int E = 14 - ex;
  E = E < -100 ? -100 : (E > 100 ? 100 : E);
  if (!(t > 0.f) || ((__float_as_uint(t) >> 23) & 0xffu) == 0xffu) E = 0;
  sE = __uint_as_float((unsigned)(E + 127) << 23);
  inv_sE = __uint_as_float((unsigned)(127 - E) << 23);
}
// max |v[0 .. n)| over an NT-thread workgroup, through NT / 64 floats of LDS at `red` (two barriers; every thread gets the result)
template <int NT = 512>
__device__ __forceinline__ float block_absmax(const float* __restrict__ v, int n, float* red, int tid) {
  float m = 0.f;
  for (int k = tid; k < n; k += NT) m = fmaxf(m, fabsf(v[k]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  float r = red[0];
#pragma unroll
  for (int i = 1; i < NT / 64; ++i) r = fmaxf(r, red[i]);
  __syncthreads();
  return r;
}
__device__ __forceinline__ float block512_absmax(const float* __restrict__ v, int n, float* red, int tid) {
  return block_absmax<512>(v, n, red, tid);
}

__device__ __forceinline__ unsigned i8x2_to_bf16x2(int b0, int b1) {
  // two small signed integers -> packed bf16 pair (exact for |v| <= 256)
  return (__float_as_uint((float)b0) >> 16) | (__float_as_uint((float)b1) & 0xffff0000u);
}

// NB > 1 (B_I8 = false): B is an fp32 matrix given as NB bf16 planes B = B_0 + B_1 + B_2 (plane r at B + r * sBp); the
// products A_q . B_r with q + r < PMAX are accumulated: PMAX = 5 keeps all nine (the exact product of the two fp32 values
// up to fp32 accumulation), PMAX = 3 the six leading ones (the dropped terms are <= 2^-24 of the product).  This is the
// GEMM of the frozen fp32 KD teacher, whose weights are split once: 6 / 9 bf16 MFMAs per k-step amortise the split of the
// activations that bounds the three-product kernels, at 16x the fp32-MFMA rate per instruction.
template <int NSPLIT, bool B_I8, int NB = 1, int PMAX = 5, bool F16 = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NB == 1 ? 3 : 2, NB == 1 ? 3 : 2))) void qgemm_bf16s_nt_kernel(QGemmArgs p) {
  static_assert(NB == 1 || !B_I8, "plane-split B is an fp32 operand");
  static_assert(!F16 || (NSPLIT == 2 && !B_I8 && NB == 1), "two fp16 planes: the linear layers' dX (B = fp16 codes)");
  constexpr int BM = 128, BN = 128;
  constexpr int PLANE = BM * QBS_LD;                 // bytes per bf16 plane of A
  constexpr int PLANE_B = BN * QBS_LD;
  __shared__ __attribute__((aligned(16))) unsigned char smem[NSPLIT * PLANE + NB * PLANE_B];
  int tm, tn, gby;
  qgemm_tile_id(p, tm, tn, gby);
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  const int b0 = gby / p.nb1, b1 = gby % p.nb1;
  const float* A = (const float*)p.A + b0 * p.sA0 + b1 * p.sA1;
  const unsigned short* B = (const unsigned short*)p.B + (B_I8 ? 0 : b0 * p.sB0 + b1 * p.sB1);
  const signed char* B8 = (const signed char*)p.B + b0 * p.sB0 + b1 * p.sB1;
  const float* ksp = p.s ? p.s + b1 * p.sK1 : nullptr;
  const int K = p.K;
  const int nkt = (K + QBS_BK - 1) / QBS_BK;

  // A: 128 rows x 32 fp32 = 1024 float4 -> 4 per thread (row = f >> 3, kq = f & 7)
  // B: 128 rows x 32 bf16 = 512 x 16 B  -> 2 per thread (row = f >> 2, kq = f & 3)
  int64_t offA[4], offB[2];
  bool okA[4], okB[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int f = tid + 256 * i;
    const int row = f >> 3;
    okA[i] = (m0 + row) < p.M;
    offA[i] = (int64_t)min(m0 + row, p.M - 1) * p.lda + (f & 7) * 4;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int f = tid + 256 * i;
    const int row = f >> 2;
    okB[i] = (n0 + row) < p.N;
    offB[i] = (int64_t)min(n0 + row, p.N - 1) * p.ldb + (f & 3) * 8;
  }
  const int kqa = (tid & 7) * 4;       // same for all 4 chunks (256 % 8 == 0)
  const int kqb = (tid & 3) * 8;
  // gload only issues the loads; scaling, masking, the int8 -> bf16 conversion and the split happen at the LDS store of
  // the next iteration, behind the MFMAs of this one (a value touched inside gload is waited for in front of them)
  float sE = 1.f, inv_sE = 1.f;        // F16: the launch's power of two (see qgemm_bf16s_nt_wide_sk_kernel)
  if constexpr (F16) {
    const float m = ksp ? block_absmax<256>(ksp, K, reinterpret_cast<float*>(smem), tid) : 1.f;
    const float a = ofq_amax_load(p.amax);
    f16_plane_scale(a == a ? a * m : a, sE, inv_sE);
  }
  f32x4v ra[4], rks;
  i32x4 rb[NB][2];
  bool rkina = false, rkinb = false;
  auto gload = [&](int kt) {
    const int k0 = kt * QBS_BK;
    const bool kina = (k0 + kqa) < K, kinb = (k0 + kqb) < K;       // K % 8 == 0 (host check)
    rks = *reinterpret_cast<const f32x4v*>(ksp ? ksp + (kina ? k0 + kqa : 0) : A + offA[0]);      // no ksp: any valid address
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const f32x4v*>(A + offA[i] + (kina ? k0 : -kqa));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (!B_I8) {
#pragma unroll
        for (int r = 0; r < NB; ++r) rb[r][i] = *reinterpret_cast<const i32x4*>(B + r * p.sBp + offB[i] + (kinb ? k0 : -kqb));
      } else {
        const u32x2v v = *reinterpret_cast<const u32x2v*>(B8 + offB[i] + (kinb ? k0 : -kqb));
        rb[0][i].x = (int)v[0];
        rb[0][i].y = (int)v[1];
      }
    }
    rkina = kina;
    rkinb = kinb;
  };
  auto lstore = [&]() {
    asm volatile("" : "+v"(rks));
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(ra[i]));
    float ks[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t = ksp ? rks[e] : 1.f;
      if (B_I8 && p.gscale2 > 0.f) t = ofq_lsq_eff_scale(t, p.gscale2);      // raw LSQ step -> effective value
      if constexpr (F16) t *= sE;
      ks[e] = rkina ? t : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = tid + 256 * i;
      const int row = f >> 3;
      const float z = okA[i] ? 1.f : 0.f;
      const f32x2v k01 = {ks[0] * z, ks[1] * z}, k23 = {ks[2] * z, ks[3] * z};
      const f32x2v a01 = {ra[i][0], ra[i][1]}, a23 = {ra[i][2], ra[i][3]};
      unsigned lo[NSPLIT], hi[NSPLIT];
      if constexpr (F16) {
        const f32x2v x01 = a01 * k01, x23 = a23 * k23;
        split2_f16(x01[0], x01[1], lo[0], lo[1]);
        split2_f16(x23[0], x23[1], hi[0], hi[1]);
      } else {
        split_pair_bf16<NSPLIT>(a01 * k01, lo);
        split_pair_bf16<NSPLIT>(a23 * k23, hi);
      }
#pragma unroll
      for (int sidx = 0; sidx < NSPLIT; ++sidx) {
        uint2 w;
        w.x = lo[sidx];
        w.y = hi[sidx];
        *reinterpret_cast<uint2*>(&smem[sidx * PLANE + row * QBS_LD + kqa * 2]) = w;
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int f = tid + 256 * i;
      const int row = f >> 2;
      const int m = (okB[i] && rkinb) ? -1 : 0;
      i32x4 w;
      if (!B_I8) {
#pragma unroll
        for (int r = 1; r < NB; ++r) {
          asm volatile("" : "+v"(rb[r][i]));
          *reinterpret_cast<i32x4*>(&smem[NSPLIT * PLANE + r * PLANE_B + row * QBS_LD + kqb * 2]) = rb[r][i] & m;
        }
        asm volatile("" : "+v"(rb[0][i]));
        w = rb[0][i] & m;
      } else {   // 8 int8 codes -> 8 bf16
        int w0 = rb[0][i].x, w1 = rb[0][i].y;
        asm volatile("" : "+v"(w0), "+v"(w1));
        w0 &= m;
        w1 &= m;
        w.x = (int)i8x2_to_bf16x2((int)(signed char)(w0 & 0xff), (int)(signed char)((w0 >> 8) & 0xff));
        w.y = (int)i8x2_to_bf16x2((int)(signed char)((w0 >> 16) & 0xff), (int)(signed char)((w0 >> 24) & 0xff));
        w.z = (int)i8x2_to_bf16x2((int)(signed char)(w1 & 0xff), (int)(signed char)((w1 >> 8) & 0xff));
        w.w = (int)i8x2_to_bf16x2((int)(signed char)((w1 >> 16) & 0xff), (int)(signed char)((w1 >> 24) & 0xff));
      }
      *reinterpret_cast<i32x4*>(&smem[NSPLIT * PLANE + row * QBS_LD + kqb * 2]) = w;
    }
  };

  f32x16q acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  gload(0);
  for (int kt = 0; kt < nkt; ++kt) {
    lstore();
    __syncthreads();
    gload(min(kt + 1, nkt - 1));        // unconditional (the last one is never stored) and pinned ahead of the MFMAs
    __builtin_amdgcn_sched_barrier(0);
    const unsigned char* a = &smem[(wm * 64 + l31) * QBS_LD + lh * 16];
    const unsigned char* b = &smem[NSPLIT * PLANE + (wn * 64 + l31) * QBS_LD + lh * 16];
#pragma unroll
    for (int ks = 0; ks < QBS_BK / 16; ++ks) {
      bf16x8 bv[NB][2];
#pragma unroll
      for (int r = 0; r < NB; ++r)
#pragma unroll
        for (int j = 0; j < 2; ++j) bv[r][j] = *reinterpret_cast<const bf16x8*>(b + r * PLANE_B + j * 32 * QBS_LD + ks * 32);
#pragma unroll
      for (int sidx = 0; sidx < NSPLIT; ++sidx) {
        bf16x8 av[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
          av[i] = *reinterpret_cast<const bf16x8*>(a + sidx * PLANE + i * 32 * QBS_LD + ks * 32);
#pragma unroll
        for (int r = 0; r < NB; ++r) {
          if (sidx + r >= PMAX) continue;
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = mfma_16b<F16>(av[i], bv[r][j], acc[i][j]);
        }
      }
    }
    __syncthreads();
  }

  // the per-row addend goes through LDS once (the loop's last barrier released smem) instead of 32 conditional global
  // loads per lane; old values for C += ... are fetched unconditionally on clamped addresses, a row quad at a time
  float* row_u = reinterpret_cast<float*>(smem);
  const bool has_u = B_I8 && p.u != nullptr;
  if (has_u) {
    if (tid < BM) row_u[tid] = p.u[((int64_t)b0 * p.M + min(m0 + tid, p.M - 1)) * p.nb1 + b1];
    __syncthreads();
  }
  float* Cb = p.C + b0 * p.sC0 + b1 * p.sC1;
  int ncc[2];
  bool nok[2];
  float cbias[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn * 64 + j * 32 + l31;
    nok[j] = n < p.N;
    ncc[j] = min(n, p.N - 1);
    cbias[j] = (NB > 1 && p.bias) ? p.bias[ncc[j]] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int eb = 0; eb < 4; ++eb) {
      float old[4][2];
      if (p.accumulate) {
#pragma unroll
        for (int ee = 0; ee < 4; ++ee) {
          const int mc = min(m0 + wm * 64 + i * 32 + ee + 8 * eb + 4 * lh, p.M - 1);
#pragma unroll
          for (int j = 0; j < 2; ++j) old[ee][j] = Cb[(int64_t)mc * p.ldc + ncc[j]];
        }
      }
#pragma unroll
      for (int ee = 0; ee < 4; ++ee) {
        const int ml = wm * 64 + i * 32 + ee + 8 * eb + 4 * lh;
        const int m = m0 + ml;
        const float uu = has_u ? row_u[ml] : 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j)
          if (m < p.M && nok[j]) {
            float v = acc[i][j][eb * 4 + ee] * (F16 ? p.alpha * inv_sE : p.alpha) + uu;
            if (NB > 1) v += cbias[j];
            if (p.accumulate) v += old[ee][j];
            Cb[(int64_t)m * p.ldc + ncc[j]] = v;
          }
      }
    }
}

// ------------------------------------------------------------------------------------------------ bf16-split dW (TN)
// dW[o,c] = sum_m (dY[m,o] * a_eff[m % S]) * qx[m,c]  +  db[o] * baft[c]          (autograd of F.linear wrt the weight:
// dY^T @ X_hat with X_hat = a_eff*qx + baft).  Both operands are contiguous along the NON-contracted dimension, so
// they are staged in their natural [k][t] layout (dY split into three bf16 planes, the int8 codes widened to bf16)
// and the MFMA fragments are fetched with the gfx950 LDS transpose read ds_read_b64_tr_b16: within a 16-lane group
// lane p supplies the 8-byte chunk (row k0 + p/4, cols t0 + 4*(p%4) .. +3) and receives column t0 + p, rows k0..k0+3.
// LDS rows are padded to 320 B so the two 16-lane groups of a half-wave (4 rows x 32 B each) hit disjoint banks.
// Split-K over the token dimension; partials are reduced in a fixed order together with the rank-1 offset term.
#include <type_traits>
#define QTN_BK 32
#define QTN_LD 320                      // bytes per LDS row: 128 bf16 + 64 B pad
typedef short s16x4 __attribute__((ext_vector_type(4)));

struct QTnArgs {
  const float* A;        // dY   [Ktok][M]  (M = out features)
  const int8_t* B;       // codes [Ktok][N] (N = in features)
  float* ws;             // [split][M][N]
  float* csum;           // [split][M] column sums of dY over this split's tokens (optional)
  const float* s;        // LSQ step vector [S]
  const unsigned* amax;  // two-plane fp16 form (wide kernels): bits of an upper bound of max |A| (device word); NULL: three bf16 planes
  // direct (batched, un-split) mode: C written by the GEMM kernel itself
  float* C;              // NULL = split-K mode
  const float* baft;     // direct mode: + colsum_k(A)[m] * baft[n + b1 * sBf1]
  int64_t sBf1;          // (plain attention: the offset vector of head b1 starts at b1 * d)
  int64_t lda, ldb, ldc;
  int64_t sA0, sA1, sB0, sB1, sC0, sC1;
  int M, N, Ktok, S, split, tiles_m, tiles_n, nb1, Mstore, Nstore, trans_out;
  float gscale;
  // stream kernel, stacked form (dqkx): the output rows of the stk_h inner batch entries (heads) are laid end to end,
  // stk_mp rows apiece (stk_valid of them real), and tiled as ONE matrix of stk_h * stk_mp rows -- the heads share the B
  // operand.  Row r belongs to head r / stk_mp = (r * stk_magic) >> 20 (host-verified for every row of the launch).
  int stk_mp, stk_h, stk_valid;
  unsigned stk_magic;
};

__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* base) {
  // two transpose reads: k rows 0..3 and 4..7 of this lane's k-group
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 4 * QTN_LD));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// W4: 64 x 256 tile, the four waves side by side (each 64 x 64) -- for outputs with at most 64 rows (dV: the rows are one
// head's channels), where half of a 128 x 128 tile's waves had nothing to multiply and the 208 keys took two workgroups
// that each staged (and split) the same dO panel: 1536 workgroups on 768 slots became 768.  The code operand then fills
// two LDS planes (columns 0-127 / 128-255).  No column-sum by-product in this form.
template <bool W4 = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void qgemm_bf16s_tn_kernel(QTnArgs p) {
  constexpr int BM = W4 ? 64 : 128, BN = W4 ? 256 : 128, NS = 3;
  constexpr int NA = W4 ? 2 : 4, NB = W4 ? 2 : 1, AKS = W4 ? 16 : 8;      // staging chunks per thread, k rows between A chunks
  constexpr int PLANE = QTN_BK * QTN_LD;
  __shared__ __attribute__((aligned(16))) unsigned char smem[(NS + NB) * PLANE];
  const int ntiles = p.tiles_m * p.tiles_n;
  // XCD-aware order: block b runs on XCD b % 8; give each XCD a contiguous run of logical ids so that the tiles which
  // share one dY panel (same split, same tm, all tn) hit the same L2 instead of re-fetching the panel per XCD
  int lid, gby;
  xcd_remap_grid(lid, gby);
  const int tile = lid % ntiles, sidx = lid / ntiles;
  const int tm = tile / p.tiles_n, tn = tile % p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = W4 ? 0 : wid >> 1, wn = W4 ? wid : wid & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  const int b0 = gby / p.nb1, b1 = gby % p.nb1;
  const bool direct = p.C != nullptr;

  const int nkt = (p.Ktok + QTN_BK - 1) / QTN_BK;
  const int tps = (nkt + p.split - 1) / p.split;
  const int t_begin = sidx * tps, t_end = min(nkt, t_begin + tps);

  // staging maps
  const int a_k = W4 ? tid >> 4 : tid >> 5, a_t = W4 ? (tid & 15) * 4 : (tid & 31) * 4;          // + AKS*i rows
  const int b_k = tid >> 3, b_c = (tid & 7) * 16;
  const bool a_ok = (m0 + a_t) < p.M;                      // M % 4 == 0 (host check)
  bool b_ok[NB];                                           // N % 16 == 0
#pragma unroll
  for (int c = 0; c < NB; ++c) b_ok[c] = (n0 + b_c + 128 * c) < p.N;
  const float* Ap = p.A + b0 * p.sA0 + b1 * p.sA1 + (a_ok ? m0 + a_t : 0);
  const int8_t* Bp = p.B + b0 * p.sB0 + b1 * p.sB1;          // (+ the chunk's column, or column 0 for a chunk past N: never read past a row)
  // gload only issues the loads; masks, the effective step, the column sums and the split happen at the LDS store of
  // the next iteration, behind the MFMAs of this one (a value touched inside gload is waited for in front of them)
  f32x4v ra[NA];
  float rs[NA];
  i32x4 rb[NB];
  bool rok[NA], rbok[NB];
#pragma unroll
  for (int c = 0; c < NB; ++c) rbok[c] = false;
  float4 csacc = make_float4(0.f, 0.f, 0.f, 0.f);   // column sums of the raw dY (bias gradient), tn == 0 tiles only
  const bool do_csum = !W4 && (direct ? (p.baft != nullptr) : (p.csum != nullptr && tn == 0));
  // token index modulo S, kept incrementally (gload runs on consecutive k-steps): the integer modulo is ~22 VALU
  // instructions, four of them per k-step were a third of this kernel's staging work
  const bool kmod_inc = p.S >= QTN_BK;
  int kmod[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) kmod[i] = (t_begin * QTN_BK + a_k + AKS * i) % p.S;
  auto gload = [&](int kt_) {
    const bool live = kt_ < t_end;                  // past the end: repeat the last tile, masked out of the column sums
    const int k0 = min(kt_, t_end - 1) * QTN_BK;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int k = k0 + a_k + AKS * i;
      const int kc = min(k, p.Ktok - 1);
      ra[i] = *reinterpret_cast<const f32x4v*>(Ap + (int64_t)kc * p.lda);
      rs[i] = p.s[kmod_inc ? kmod[i] : kc % p.S];
      kmod[i] += QTN_BK;
      kmod[i] -= (kmod[i] >= p.S) ? p.S : 0;
      rok[i] = a_ok && k < p.Ktok && live;
    }
    const int k = k0 + b_k;
#pragma unroll
    for (int c = 0; c < NB; ++c) {
      rb[c] = *reinterpret_cast<const i32x4*>(Bp + (int64_t)min(k, p.Ktok - 1) * p.ldb + (b_ok[c] ? n0 + b_c + 128 * c : 0));
      rbok[c] = b_ok[c] && k < p.Ktok;
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < NA; ++i) asm volatile("" : "+v"(ra[i]), "+v"(rs[i]));
#pragma unroll
    for (int c = 0; c < NB; ++c) asm volatile("" : "+v"(rb[c]));
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const unsigned msk = rok[i] ? 0xffffffffu : 0u;
      float4 v;
      v.x = __uint_as_float(__float_as_uint(ra[i][0]) & msk);
      v.y = __uint_as_float(__float_as_uint(ra[i][1]) & msk);
      v.z = __uint_as_float(__float_as_uint(ra[i][2]) & msk);
      v.w = __uint_as_float(__float_as_uint(ra[i][3]) & msk);
      csacc.x += v.x; csacc.y += v.y; csacc.z += v.z; csacc.w += v.w;
      const float sc = ofq_lsq_eff_scale(rs[i], p.gscale);
      const f32x2v v01 = {v.x, v.y}, v23 = {v.z, v.w};
      unsigned lo[NS], hi[NS];
      split_pair_bf16<NS>(v01 * sc, lo);
      split_pair_bf16<NS>(v23 * sc, hi);
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        uint2 w;
        w.x = lo[q];
        w.y = hi[q];
        *reinterpret_cast<uint2*>(&smem[q * PLANE + (a_k + AKS * i) * QTN_LD + a_t * 2]) = w;
      }
    }
    // 16 int8 codes -> 16 bf16
#pragma unroll
    for (int c = 0; c < NB; ++c) {
      const i32x4 rbm = rb[c] & (rbok[c] ? -1 : 0);
      unsigned w[8];
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const int word = rbm[d];
        w[2 * d] = i8x2_to_bf16x2((int)(signed char)(word & 0xff), (int)(signed char)((word >> 8) & 0xff));
        w[2 * d + 1] = i8x2_to_bf16x2((int)(signed char)((word >> 16) & 0xff), (int)(signed char)((word >> 24) & 0xff));
      }
      unsigned char* dst = &smem[(NS + c) * PLANE + b_k * QTN_LD + b_c * 2];
      *reinterpret_cast<uint4*>(dst) = make_uint4(w[0], w[1], w[2], w[3]);
      *reinterpret_cast<uint4*>(dst + 16) = make_uint4(w[4], w[5], w[6], w[7]);
    }
  };

  f32x16q acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // per-lane transpose-read address inside a [k][t] plane (k-step 0, fragment column block 0)
  const int p16 = lane & 15;
  const int fr_off = (8 * lh + (p16 >> 2)) * QTN_LD + (16 * ((lane >> 4) & 1) + 4 * (p16 & 3)) * 2;

  // a wave whose 64 x 64 quarter lies outside the matrix (dV: 64 rows; plain dk / dq: 64 columns) only helps with staging
  const bool wave_on = (m0 + wm * 64 < p.M) && (n0 + wn * 64 < p.N);
  if (t_begin < t_end) {
    gload(t_begin);
    for (int kt = t_begin; kt < t_end; ++kt) {
      lstore();
      __syncthreads();
      gload(kt + 1);                       // unconditional (clamped, masked past the end), pinned ahead of the MFMAs
      __builtin_amdgcn_sched_barrier(0);
      if (wave_on)
#pragma unroll
      for (int ks = 0; ks < QTN_BK / 16; ++ks) {
        bf16x8 bv[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
          bv[j] = tr_frag(&smem[(NS + (W4 ? wn >> 1 : 0)) * PLANE + ks * 16 * QTN_LD + fr_off + ((W4 ? wn & 1 : wn) * 64 + j * 32) * 2]);
#pragma unroll
        for (int q = 0; q < NS; ++q) {
          bf16x8 av[2];
#pragma unroll
          for (int i = 0; i < 2; ++i) av[i] = tr_frag(&smem[q * PLANE + ks * 16 * QTN_LD + fr_off + (wm * 64 + i * 32) * 2]);
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
      }
      __syncthreads();
    }
  }
  if (!direct) {
    float* W = p.ws + (int64_t)sidx * p.M * p.N;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn * 64 + j * 32 + l31;
      if (n >= p.N) continue;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
          if (m < p.M) W[(int64_t)m * p.N + n] = acc[i][j][e];
        }
    }
  }
  float* red1 = reinterpret_cast<float*>(smem) + 8 * 32 * 4;      // 128 finished column sums live behind the partials
  if (do_csum) {     // reduce the 8 row-groups that share a column quad, one writer per quad
    float4* red = reinterpret_cast<float4*>(smem);
    red[a_k * 32 + (tid & 31)] = csacc;
    __syncthreads();
    if (a_k == 0) {
      float4 t = red[tid & 31];
#pragma unroll
      for (int g = 1; g < 8; ++g) {
        const float4 u = red[g * 32 + (tid & 31)];
        t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
      }
      if (!direct) {
        if (a_ok) *reinterpret_cast<float4*>(p.csum + (int64_t)sidx * p.M + m0 + a_t) = t;
      } else {
        *reinterpret_cast<float4*>(red1 + a_t) = t;
      }
    }
    __syncthreads();
  }
  if (direct) {
    float* Cb = p.C + b0 * p.sC0 + b1 * p.sC1;
    // transposed output (dV: C[n][m], m = the head's channels): a lane holds four consecutive m per accumulator quad, so
    // the row piece goes out as one 16-byte store instead of four dword stores that each touch 64 different lines
    const bool quad_ok = p.trans_out && (p.Mstore & 3) == 0 && (p.ldc & 3) == 0 && ((p.sC0 | p.sC1) & 3) == 0 && al16_dev(p.C);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn * 64 + j * 32 + l31;
      if (n >= p.Nstore) continue;
      const float bf = p.baft ? p.baft[n + b1 * p.sBf1] : 0.f;
      if (quad_ok) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int ml = wm * 64 + i * 32 + 8 * k + 4 * lh;
            const int m = m0 + ml;
            if (m < p.Mstore) {
              float4 v = make_float4(acc[i][j][4 * k], acc[i][j][4 * k + 1], acc[i][j][4 * k + 2], acc[i][j][4 * k + 3]);
              if (p.baft) { v.x += red1[ml] * bf; v.y += red1[ml + 1] * bf; v.z += red1[ml + 2] * bf; v.w += red1[ml + 3] * bf; }
              *reinterpret_cast<float4*>(&Cb[(int64_t)n * p.ldc + m]) = v;
            }
          }
        continue;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int ml = wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
          const int m = m0 + ml;
          if (m < p.Mstore) {
            float v = acc[i][j][e];
            if (p.baft) v += red1[ml] * bf;
            if (p.trans_out) Cb[(int64_t)n * p.ldc + m] = v;
            else Cb[(int64_t)m * p.ldc + n] = v;
          }
        }
    }
  }
}


// Wide variant for the linear layers (split-K mode only): 8 waves own a 128 x (128*NJ) tile, so one split of a dY
// panel feeds NJ times more MFMA work.  tools/probe/overlap_probe.hip shows that on gfx950 the VALU stream of one wave
// does NOT overlap the MFMA stream of its SIMD partner (233 us together vs 103 + 135 us alone), so every split/convert
// instruction is paid in full: the lever is fewer VALU instructions per MFMA, which the wide tile gives.  LDS is
// double buffered with ONE barrier per k-step (LDS-only barrier: global prefetches stay in flight across it).
template <int LD>
__device__ __forceinline__ bf16x8 tr_frag_ld(const unsigned char* base) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 4 * LD));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// ---- window-sized direct GEMM: one WAVE per (batch, column chunk) ------------------------------------------------------
// Swin's 49-token windows give C[m][n] = sum_k (sc[k] * A[k][m]) * B[k][n] with M <= 64 and K <= 64 per (window, head):
// on the 128 x 128 (x 384) tiles above a workgroup lives for one load round trip + one short k-loop, three (one) of them
// per CU, and the launch is bound by that latency chain, not by HBM or the MFMAs.  Here every wave is its own tile: the
// fp32 operand goes from global memory straight into MFMA fragment layout (lane = column m, eight consecutive k: dword
// loads, 128 B per row and half-wave), is split into its three bf16 planes once, in registers, and reused for every 64
// columns of the int8 operand, which is staged through a wave-private LDS slice ([k][64] bf16, transpose reads).  No
// workgroup barrier: the four waves of a workgroup are independent, eight waves per CU stay in flight.
// TRANS = trans_out: the operands swap MFMA roles so that the lanes run along the contiguous output dimension.
#define QTW_LDB 192                     // bytes per LDS row: 64 bf16 + 64 B pad (4 consecutive k rows on disjoint bank slots)
#define QTW_SLICE (64 * QTW_LDB + 2 * 64 * 4)
template <int MB, bool TRANS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MB == 2 ? 2 : 3, MB == 2 ? 2 : 3)))
void qgemm_bf16s_tn_win_kernel(QTnArgs p, int chunks, int cpc, int ntasks) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * QTW_SLICE];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int bx, by_unused;
  xcd_remap_grid(bx, by_unused);                       // heads of one window share B: keep them on one XCD
  const int task = bx * 4 + wid;
  if (task >= ntasks) return;                          // wave-uniform
  const int pair = task / chunks, chunk = task - pair * chunks;
  const int b0 = pair / p.nb1, b1 = pair - b0 * p.nb1;
  unsigned char* sb = smem + wid * QTW_SLICE;
  float* ssc = reinterpret_cast<float*>(sb + 64 * QTW_LDB);
  float* scs = ssc + 64;
  const int l31 = lane & 31, lh = lane >> 5;

  ssc[lane] = lane < p.Ktok ? ofq_lsq_eff_scale(p.s[lane % p.S], p.gscale) : 0.f;
  asm volatile("" ::: "memory");

  // A: fragments straight from global memory, split once
  const float* Ap = p.A + b0 * p.sA0 + b1 * p.sA1;
  unsigned av[3][4][MB][4];
  float csum[MB];
#pragma unroll
  for (int i = 0; i < MB; ++i) {
    const int m = 32 * i + l31;
    const bool mok = m < p.M;
    const float* Am = Ap + (mok ? m : 0);
    float cs = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float raw[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = 16 * ks + 8 * lh + e;
        raw[e] = Am[(unsigned)(min(k, p.Ktok - 1) * (int)p.lda)];
      }
      const float4 s0 = *reinterpret_cast<const float4*>(ssc + 16 * ks + 8 * lh);
      const float4 s1 = *reinterpret_cast<const float4*>(ssc + 16 * ks + 8 * lh + 4);
      const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        const int k = 16 * ks + 8 * lh + e;
        const float v0 = (mok && k < p.Ktok) ? raw[e] : 0.f;
        const float v1 = (mok && k + 1 < p.Ktok) ? raw[e + 1] : 0.f;
        cs += v0;
        cs += v1;
        const f32x2v x = {v0 * sc[e], v1 * sc[e + 1]};
        unsigned pl[3];
        split_pair_bf16<3>(x, pl);
#pragma unroll
        for (int q = 0; q < 3; ++q) av[q][ks][i][e >> 1] = pl[q];
      }
    }
    csum[i] = cs + __shfl_xor(cs, 32, 64);
  }
  if (p.baft) {
#pragma unroll
    for (int i = 0; i < MB; ++i)
      if (lh == 0) scs[32 * i + l31] = csum[i];
    asm volatile("" ::: "memory");
  }

  const int8_t* Bp = p.B + b0 * p.sB0 + b1 * p.sB1;
  float* Cb = p.C + b0 * p.sC0 + b1 * p.sC1;
  const int n_end = min(p.N, (chunk + 1) * cpc);
  const int st_row = lane >> 2, st_col = (lane & 3) * 16;
  const int p16 = lane & 15;
  const unsigned char* frb = sb + (8 * lh + (p16 >> 2)) * QTW_LDB + (16 * ((lane >> 4) & 1) + 4 * (p16 & 3)) * 2;
  auto bload = [&](int n0, i32x4 (&rb)[4]) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 16 + st_row, col = n0 + st_col;
      const bool ok = row < p.Ktok && col < p.N;
      const i32x4 v = *reinterpret_cast<const i32x4*>(Bp + (unsigned)(min(row, p.Ktok - 1) * (int)p.ldb) + (col < p.N ? col : 0));
      rb[it] = v & (ok ? -1 : 0);
    }
  };
  i32x4 rb[4];
  bload(chunk * cpc, rb);
  for (int n0 = chunk * cpc; n0 < n_end; n0 += 64) {
    // stage: 16 codes -> 16 bf16 per lane and row group
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      unsigned w[8];
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const int word = rb[it][d];
        w[2 * d] = i8x2_to_bf16x2((int)(signed char)(word & 0xff), (int)(signed char)((word >> 8) & 0xff));
        w[2 * d + 1] = i8x2_to_bf16x2((int)(signed char)((word >> 16) & 0xff), (int)(signed char)((word >> 24) & 0xff));
      }
      unsigned char* dst = sb + (it * 16 + st_row) * QTW_LDB + st_col * 2;
      *reinterpret_cast<uint4*>(dst) = make_uint4(w[0], w[1], w[2], w[3]);
      *reinterpret_cast<uint4*>(dst + 16) = make_uint4(w[4], w[5], w[6], w[7]);
    }
    asm volatile("" ::: "memory");                     // one wave: its LDS operations execute in program order
    if (n0 + 64 < n_end) bload(n0 + 64, rb);           // next block's codes fly behind the MFMAs
    const bool two = n0 + 32 < n_end;                  // a 32-column tail needs one column block only
#pragma unroll 1
    for (int j = 0; j < 2; ++j) {
      if (j == 1 && !two) break;
      f32x16q acc[MB];
#pragma unroll
      for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 bv = tr_frag_ld<QTW_LDB>(frb + ks * 16 * QTW_LDB + j * 64);
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
          for (int i = 0; i < MB; ++i) {
            typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
            const u32x4v au = {av[q][ks][i][0], av[q][ks][i][1], av[q][ks][i][2], av[q][ks][i][3]};
            const bf16x8 a = __builtin_bit_cast(bf16x8, au);
            if (TRANS) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv, a, acc[i], 0, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bv, acc[i], 0, 0, 0);
          }
      }
      // epilogue of the 32-column block
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        if (!TRANS) {
          const int n = n0 + 32 * j + l31;
          const bool nok = n < p.Nstore;
          const float bf = (p.baft && nok) ? p.baft[n + b1 * p.sBf1] : 0.f;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int m = 32 * i + (e & 3) + 8 * (e >> 2) + 4 * lh;
            float v = acc[i][e];
            if (p.baft) v += scs[m] * bf;
            if (nok && m < p.Mstore) Cb[(int64_t)m * p.ldc + n] = v;
          }
        } else {
          const int m = 32 * i + l31;
          const bool mok = m < p.Mstore;
          const float cm = p.baft ? scs[m] : 0.f;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int n = n0 + 32 * j + (e & 3) + 8 * (e >> 2) + 4 * lh;
            float v = acc[i][e];
            if (p.baft && n < p.Nstore) v += cm * p.baft[n + b1 * p.sBf1];
            if (mok && n < p.Nstore) Cb[(int64_t)n * p.ldc + m] = v;
          }
        }
      }
    }
    asm volatile("" ::: "memory");                     // the next block's LDS stores stay behind these fragment reads
  }
}

// direct-mode launch of the window kernel when the shape fits one wave tile; returns false when it does not apply
static bool tn_win_launch(const QTnArgs& a, int64_t batches, hipStream_t st) {
  static const bool off = getenv("OFQ_NO_WIN_TN") != nullptr;
  if (off || !a.C || a.split != 1 || (a.N & 15) || (a.ldb & 15) || (int64_t)a.Ktok * a.lda >= (1ll << 31) ||
      (int64_t)a.Ktok * a.ldb >= (1ll << 31))
    return false;
  // (a k-chunked form of this tile for the 198-token dV / plain dk was measured and dropped: every wave re-splits its A rows
  // per 64-column block and stalls on each chunk's loads -- 27.61 vs 27.58 ms/step, no gain over the workgroup tile)
  if (a.Ktok > 64 || a.M > 64 || a.S < a.Ktok) return false;
  // one wave walks up to 384 columns with the split planes of its A operand in registers
  const int cpc = a.N <= 384 ? (int)ceil_div(a.N, 64) * 64 : 384;
  const int chunks = (int)ceil_div(a.N, cpc);
  const int64_t ntasks = batches * chunks;
  if (ntasks >= (1ll << 31)) return false;
  const dim3 grid((unsigned)ceil_div(ntasks, 4)), block(256);
  const bool two = a.M > 32;
  if (a.trans_out) {
    if (two) hipLaunchKernelGGL((qgemm_bf16s_tn_win_kernel<2, true>), grid, block, 0, st, a, chunks, cpc, (int)ntasks);
    else hipLaunchKernelGGL((qgemm_bf16s_tn_win_kernel<1, true>), grid, block, 0, st, a, chunks, cpc, (int)ntasks);
  } else {
    if (two) hipLaunchKernelGGL((qgemm_bf16s_tn_win_kernel<2, false>), grid, block, 0, st, a, chunks, cpc, (int)ntasks);
    else hipLaunchKernelGGL((qgemm_bf16s_tn_win_kernel<1, false>), grid, block, 0, st, a, chunks, cpc, (int)ntasks);
  }
  return true;
}

#ifdef TNW_TIMING
__device__ unsigned long long g_tnw_dbg[8][8];     // [wave][phase] cycles of block 0 (tools/probe/tn_probe.hip)
#define TNW_T(slot) do { const unsigned long long t_ = clock64(); tacc[slot] += t_ - tlast; tlast = t_; } while (0)
#else
#define TNW_T(slot) do {} while (0)
#endif

// The body of the wide dW kernel: workgroup `lid` of the problem `p` (tile = lid % ntiles, split index = lid / ntiles),
// batch entry `gby`.  Two entry points share it: one problem per launch (qgemm_bf16s_tn_wide_kernel) and several
// problems per launch (qgemm_bf16s_tn_wide_group_kernel, the deferred weight gradients of a transformer block).
// F16: two fp16 planes of dY * (token step * 2^E) against the codes widened to fp16 (see split2_f16): 8 NJ MFMAs per k-step.
template <int NJ, bool F16 = false>
__device__ __forceinline__ void tn_wide_body(const QTnArgs& p, const int lid, const int gby) {
  constexpr int BM = 128, BN = 128 * NJ, NS = F16 ? 2 : 3;
  constexpr int LDA = QTN_LD;                 // 320 B: 4 consecutive k rows land on disjoint 64-B bank slots
  constexpr int LDB = BN * 2 + 64;            // same residue (64) modulo the 256-B bank line
  constexpr int PLANE = QTN_BK * LDA;
  constexpr int STAGE = NS * PLANE + QTN_BK * LDB;
  constexpr int CPR = BN / 8;                 // 8-byte code chunks per k row
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];
  const int ntiles = p.tiles_m * p.tiles_n;
  const int tile = lid % ntiles, sidx = lid / ntiles;
  const int tm = tile / p.tiles_n, tn = tile % p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 2, wn = wid & 3;
  const int l31 = lane & 31, lh = lane >> 5;

  // F16: the launch's power of two from the maximum word of dY and the largest token step (effective steps are the raw
  // steps floored at 1e-5 and rounded once more: the margin covers it)
  float sE = 1.f, inv_sE = 1.f;
  const unsigned c64 = 0x64646464u;
  if constexpr (F16) {
    const float m = fmaxf(block512_absmax(p.s, p.S, reinterpret_cast<float*>(smem), tid), 1e-5f) * 1.0001f;
    const float am = ofq_amax_load(p.amax);
    f16_plane_scale(am == am ? am * m : am, sE, inv_sE);
  }
  const int nkt = (p.Ktok + QTN_BK - 1) / QTN_BK;
  const int tps = (nkt + p.split - 1) / p.split;
  const int t_begin = sidx * tps, t_end = min(nkt, t_begin + tps);

  const int b0 = gby / p.nb1, b1 = gby % p.nb1;
  const bool direct = p.C != nullptr;                       // batched, un-split: C written here (attention dqkx)
  const int a_k = tid >> 5, a_t = (tid & 31) * 4;          // rows a_k, a_k + 16
  const bool a_ok = (m0 + a_t) < p.M;
  const float* Ap = p.A + b0 * p.sA0 + b1 * p.sA1 + (a_ok ? m0 + a_t : 0);
  int b_row[NJ], b_col[NJ];
  bool b_ok[NJ];
  const int8_t* Bp[NJ];
#pragma unroll
  for (int i = 0; i < NJ; ++i) {
    const int f = tid + 512 * i;
    b_row[i] = f / CPR;
    b_col[i] = (f % CPR) * 8;
    b_ok[i] = (n0 + b_col[i]) < p.N;                       // N % 8 == 0 (host check)
    Bp[i] = p.B + b0 * p.sB0 + b1 * p.sB1 + (b_ok[i] ? n0 + b_col[i] : 0);
  }
  // two register prefetch slots: the loads of k-step t are issued two steps before their LDS store (the ~2 us HBM
  // latency is longer than one k-step)
  f32x4v ra[2][2];
  float rs[2][2];
  bool rok[2][2], rbok[2][NJ];
  u32x2v rb[2][NJ];
  float4 csacc = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool do_csum = direct ? (p.baft != nullptr) : (p.csum != nullptr && tn == 0);
  // all element offsets fit 32 bits (host check); loads are unconditional on clamped rows, masking happens at the
  // LDS store so nothing waits on a load inside gload
  const int ldA = (int)p.lda, ldB = (int)p.ldb;
  int kmod[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) kmod[i] = (t_begin * QTN_BK + a_k + 16 * i) % p.S;
#ifdef TNW_TIMING
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = clock64();
  const unsigned long long tstart = tlast;
#endif
  // gload / lstore run unconditionally on every k-step (tiles past t_end repeat tile t_end-1 with `live` false, so
  // their rows are masked out of the column sums and their stage is never read): with `if (kt + 3 < t_end)` guards the
  // wait-count pass merges the "loads not issued" path into the steady state and waits for *every* outstanding load
  // in lstore, i.e. the two-step prefetch degenerates to one
  auto gload = [&](int kt_, auto SLOT) {
    constexpr int sl = decltype(SLOT)::value;
    const bool live = kt_ < t_end;
    const int kt = min(kt_, t_end - 1);
    const int k0 = kt * QTN_BK;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int k = k0 + a_k + 16 * i;
      const int kc = min(k, p.Ktok - 1);
      rok[sl][i] = a_ok && k < p.Ktok && live;
      ra[sl][i] = *reinterpret_cast<const f32x4v*>(Ap + (unsigned)(kc * ldA));
      rs[sl][i] = p.s[kmod[i]];
      kmod[i] += QTN_BK;                                   // gload runs on consecutive k-steps: k mod S incrementally
      kmod[i] -= (kmod[i] >= p.S) ? p.S : 0;               // S >= QTN_BK (host check)
    }
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
      const int k = k0 + b_row[i];
      rbok[sl][i] = b_ok[i] && k < p.Ktok;
      rb[sl][i] = *reinterpret_cast<const u32x2v*>(Bp[i] + (unsigned)(min(k, p.Ktok - 1) * ldB));
    }
  };
  auto lstore = [&](unsigned char* sb, auto SLOT) {
    constexpr int sl = decltype(SLOT)::value;
    // pin the slot's registers here: the k-loop body is one basic block, and without an ordered use the selects / masks
    // on the loaded values are placed right behind the loads' issue (one k-step early), where they wait for them
#pragma unroll
    for (int i = 0; i < 2; ++i) asm volatile("" : "+v"(ra[sl][i]), "+v"(rs[sl][i]));
#pragma unroll
    for (int i = 0; i < NJ; ++i) asm volatile("" : "+v"(rb[sl][i]));
#ifdef TNW_TIMING
    asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    TNW_T(4);
#endif
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float sc = F16 ? ofq_lsq_eff_scale(rs[sl][i], p.gscale) * sE : ofq_lsq_eff_scale(rs[sl][i], p.gscale);
      const unsigned msk = rok[sl][i] ? 0xffffffffu : 0u;
      float4 v;
      v.x = __uint_as_float(__float_as_uint(ra[sl][i].x) & msk);
      v.y = __uint_as_float(__float_as_uint(ra[sl][i].y) & msk);
      v.z = __uint_as_float(__float_as_uint(ra[sl][i].z) & msk);
      v.w = __uint_as_float(__float_as_uint(ra[sl][i].w) & msk);
      csacc.x += v.x; csacc.y += v.y; csacc.z += v.z; csacc.w += v.w;
      const f32x2v v01 = {v.x, v.y}, v23 = {v.z, v.w};
      unsigned lo[NS], hi[NS];
      if constexpr (F16) {
        const f32x2v x01 = v01 * sc, x23 = v23 * sc;
        split2_f16(x01[0], x01[1], lo[0], lo[1]);
        split2_f16(x23[0], x23[1], hi[0], hi[1]);
      } else {
        split_pair_bf16<NS>(v01 * sc, lo);
        split_pair_bf16<NS>(v23 * sc, hi);
      }
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        uint2 w;
        w.x = lo[q];
        w.y = hi[q];
        *reinterpret_cast<uint2*>(&sb[q * PLANE + (a_k + 16 * i) * LDA + a_t * 2]) = w;
      }
      __builtin_amdgcn_sched_barrier(0);     // one row chunk at a time: interleaving them only costs registers
    }
#ifdef TNW_TIMING
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    TNW_T(5);
#endif
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
      const unsigned bm = rbok[sl][i] ? 0xffffffffu : 0u;
      const int w0 = (int)(rb[sl][i][0] & bm), w1 = (int)(rb[sl][i][1] & bm);
      uint4 w;
      if constexpr (F16) {
        valu_cvt4_i8_f16((unsigned)w0, c64, w.x, w.y);
        valu_cvt4_i8_f16((unsigned)w1, c64, w.z, w.w);
        if (!rbok[sl][i]) w = make_uint4(0u, 0u, 0u, 0u);        // (the masked code bytes are 0, i.e. code 0: already zero)
      } else {
      w.x = i8x2_to_bf16x2((int)(signed char)(w0 & 0xff), (int)(signed char)((w0 >> 8) & 0xff));
      w.y = i8x2_to_bf16x2((int)(signed char)((w0 >> 16) & 0xff), (int)(signed char)((w0 >> 24) & 0xff));
      w.z = i8x2_to_bf16x2((int)(signed char)(w1 & 0xff), (int)(signed char)((w1 >> 8) & 0xff));
      w.w = i8x2_to_bf16x2((int)(signed char)((w1 >> 16) & 0xff), (int)(signed char)((w1 >> 24) & 0xff));
      }
      *reinterpret_cast<uint4*>(&sb[NS * PLANE + b_row[i] * LDB + b_col[i] * 2]) = w;
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  f32x16q acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int p16 = lane & 15;
  const int fr_a = (8 * lh + (p16 >> 2)) * LDA + (16 * ((lane >> 4) & 1) + 4 * (p16 & 3)) * 2;
  const int fr_b = (8 * lh + (p16 >> 2)) * LDB + (16 * ((lane >> 4) & 1) + 4 * (p16 & 3)) * 2;
  // Fragment schedule of one k-step (two 16-deep MFMA steps): every fragment of the first step is requested before the
  // first MFMA; the fragments of the second step are requested inside the first step's MFMA sequence, each dY plane
  // into the registers of the plane that has just been used up (48 fragment VGPRs instead of 72 -- the kernel sits at
  // the 256-VGPR limit of two waves per SIMD).  The scheduling barriers pin that order.
  static_assert(QTN_BK == 32, "two MFMA steps per k-step");
  auto compute = [&](const unsigned char* sb) {
    bf16x8 av[NS][2], bv[2][NJ];
    const unsigned char* sa = &sb[fr_a + wm * 64 * 2];
    const unsigned char* sbb = &sb[NS * PLANE + fr_b + wn * 32 * NJ * 2];
#pragma unroll
    for (int j = 0; j < NJ; ++j) bv[0][j] = tr_frag_ld<LDB>(sbb + j * 64);
#pragma unroll
    for (int q = 0; q < NS; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i) av[q][i] = tr_frag_ld<LDA>(sa + q * PLANE + i * 64);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < NS; ++q) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = mfma_16b<F16>(av[q][i], bv[0][j], acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);
      if (q == 0) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) bv[1][j] = tr_frag_ld<LDB>(sbb + 16 * LDB + j * 64);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) av[q][i] = tr_frag_ld<LDA>(sa + q * PLANE + 16 * LDA + i * 64);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int q = 0; q < NS; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = mfma_16b<F16>(av[q][i], bv[1][j], acc[i][j]);
    __builtin_amdgcn_sched_barrier(0);       // the staging that follows waits on global loads: keep it behind the MFMAs
  };


  using Slot0 = std::integral_constant<int, 0>;
  using Slot1 = std::integral_constant<int, 1>;
  // one k-step: MFMA on `cur` (tile kt) and staging of tile kt+1 (register slot (kt+1)&1 -> `nxt`), then the loads of
  // tile kt+3 into the freed slot
#ifdef TNW_SERIAL_STAGING
  auto step = [&](int kt, const unsigned char* cur, unsigned char* nxt, auto SLOT) {
    compute(cur);
    TNW_T(0);
    lstore(nxt, SLOT);
    TNW_T(1);
    gload(kt + 3, SLOT);
    TNW_T(2);
    lds_barrier();
    TNW_T(3);
  };
#else
  // One k-step with the staging of tile kt+1 and the loads of tile kt+3 cut into small pieces behind the 12*NJ MFMAs of
  // tile kt (see the note at static_for).  Per dY row chunk i (18 pieces): [effective step of the row, masked], then per
  // element [x = v*sc, column sum, p0 = hi16(x)] [r1 = x - p0, p1 = hi16(r1)] [r2 = r1 - p1] with a pack piece after
  // each pair, then three LDS stores; per code chunk: two convert+pack pieces (no masks: tokens past Ktok are zeroed
  // through sc, columns past N are never written) and a store; then the seven loads in consumption order.  The
  // fragments of the second 16-deep MFMA step are read behind the MFMAs that used up their registers.
  // F16 piece list per dY row chunk (11 pieces): [step], per pair [x0, x1 = v * sc; column sums] [h = cvt_pk] [r0, r1] [l = cvt_pk],
  // two plane stores
  constexpr int NM = 4 * NS * NJ, NPA = F16 ? 11 : 18, NPB = 3, NP = 2 * NPA + NPB * NJ + 2 + NJ;
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
  unsigned offA[2] = {0u, 0u}, offB[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) offB[j] = 0u;
  const unsigned stepA = 4u * (unsigned)(QTN_BK * ldA), stepB = (unsigned)(QTN_BK * ldB);
  const unsigned maxA = 4u * (unsigned)((p.Ktok - 1) * ldA), maxB = (unsigned)((p.Ktok - 1) * ldB);
  auto step = [&](int kt, const unsigned char* cur, unsigned char* nxt, auto SLOT) {
    constexpr int sl = decltype(SLOT)::value;
    bf16x8 av[NS][2], bv[2][NJ];
    const unsigned char* sa = &cur[fr_a + wm * 64 * 2];
    const unsigned char* sbb = &cur[NS * PLANE + fr_b + wn * 32 * NJ * 2];
#pragma unroll
    for (int j = 0; j < NJ; ++j) bv[0][j] = tr_frag_ld<LDB>(sbb + j * 64);
#pragma unroll
    for (int q = 0; q < NS; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i) av[q][i] = tr_frag_ld<LDA>(sa + q * PLANE + i * 64);
    __builtin_amdgcn_sched_barrier(0);
    float sc = 0.f, okf = 0.f, x_ = 0.f, r1_ = 0.f, p0v[2], p1v[2], r2v[2];
    unsigned lo[NS], hi[NS], bw[4];
    // loads of tile kt+3 (clamped to the last tile of this split, masked out when past it)
    const bool live3 = kt + 3 < t_end;
    const int rows_left3 = p.Ktok - min(kt + 3, t_end - 1) * QTN_BK;       // token rows of that tile that exist
    const unsigned advA = live3 ? stepA : 0u, advB = live3 ? stepB : 0u;
    auto piece = [&](auto P_) {
      constexpr int P = decltype(P_)::value;
      if constexpr (P < 2 * NPA) {
        constexpr int i = P / NPA, r = P % NPA;
        if constexpr (r == 0) {
          asm volatile("" : "+v"(ra[sl][i]), "+v"(rs[sl][i]));       // first touch: the wait for the slot's loads lands here
          const float e = valu_eff_scale(rs[sl][i], p.gscale);
          sc = rok[sl][i] ? (F16 ? e * sE : e) : 0.f;
          okf = rok[sl][i] ? 1.f : 0.f;
        } else if constexpr (F16) {
          if constexpr (r < 9) {
            constexpr int pr = (r - 1) / 4, st = (r - 1) % 4, e = pr * 2;
            if constexpr (st == 0) {
              valu_mul2(ra[sl][i][e], sc, ra[sl][i][e + 1], sc, x_, r1_);
              cs[e] = valu_fma(ra[sl][i][e], okf, cs[e]);
              cs[e + 1] = valu_fma(ra[sl][i][e + 1], okf, cs[e + 1]);
            }
            if constexpr (st == 1) (pr == 0 ? lo : hi)[0] = valu_cvt_pk_f16(x_, r1_);
            if constexpr (st == 2) valu_resid2_f16((pr == 0 ? lo : hi)[0], x_, r1_, p0v[0], p0v[1]);
            if constexpr (st == 3) (pr == 0 ? lo : hi)[1] = valu_cvt_pk_f16(p0v[0], p0v[1]);
          } else {
            constexpr int q = r - 9;
            uint2 w;
            w.x = lo[q];
            w.y = hi[q];
            *reinterpret_cast<uint2*>(&nxt[q * PLANE + (a_k + 16 * i) * LDA + a_t * 2]) = w;
          }
        } else if constexpr (r < 15) {
          constexpr int pr = (r - 1) / 7, rr = (r - 1) % 7;
          if constexpr (rr < 6) {
            constexpr int el = rr / 3, st = rr % 3, e = pr * 2 + el;
            if constexpr (st == 0) {
              valu_mul_hi16(ra[sl][i][e], sc, x_, p0v[el]);
              cs[e] = valu_fma(ra[sl][i][e], okf, cs[e]);
            }
            if constexpr (st == 1) valu_sub_hi16(x_, p0v[el], r1_, p1v[el]);
            if constexpr (st == 2) { r2v[el] = valu_sub(r1_, p1v[el]); }
          } else {
            valu_pack3_hi16(p0v, p1v, r2v, pr == 0 ? lo : hi);
          }
        } else {
          constexpr int q = r - 15;
          uint2 w;
          w.x = lo[q];
          w.y = hi[q];
          *reinterpret_cast<uint2*>(&nxt[q * PLANE + (a_k + 16 * i) * LDA + a_t * 2]) = w;
        }
      } else if constexpr (P < 2 * NPA + NPB * NJ) {
        constexpr int j = (P - 2 * NPA) / NPB, r = (P - 2 * NPA) % NPB;
        if constexpr (r == 0) {
          asm volatile("" : "+v"(rb[sl][j]));
          if constexpr (F16) valu_cvt4_i8_f16(rb[sl][j][0], c64, bw[0], bw[1]);
          else valu_cvt4_i8_bf16(rb[sl][j][0], bw[0], bw[1]);
        } else if constexpr (r == 1) {
          if constexpr (F16) valu_cvt4_i8_f16(rb[sl][j][1], c64, bw[2], bw[3]);
          else valu_cvt4_i8_bf16(rb[sl][j][1], bw[2], bw[3]);
        } else {
          *reinterpret_cast<uint4*>(&nxt[NS * PLANE + b_row[j] * LDB + b_col[j] * 2]) = make_uint4(bw[0], bw[1], bw[2], bw[3]);
        }
      } else if constexpr (P < 2 * NPA + NPB * NJ + 2) {
        // byte offsets advance by one tile per k-step (no advance past the split's last tile) and are clamped to the
        // last token row: one add + one min per pointer instead of a 64-bit multiply-add chain per load
        constexpr int i = P - 2 * NPA - NPB * NJ;
        rok[sl][i] = a_ok && (a_k + 16 * i) < rows_left3 && live3;
        offA[i] += advA;
        ra[sl][i] = *reinterpret_cast<const f32x4v*>(reinterpret_cast<const char*>(Ap) + min(offA[i], maxA));
        rs[sl][i] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.s) + 4u * (unsigned)kmod[i]);
        kmod[i] += QTN_BK;
        kmod[i] -= (kmod[i] >= p.S) ? p.S : 0;
      } else {
        constexpr int j = P - 2 * NPA - NPB * NJ - 2;
        offB[j] += advB;
        rb[sl][j] = *reinterpret_cast<const u32x2v*>(reinterpret_cast<const char*>(Bp[j]) + min(offB[j], maxB));
      }
    };
    static_for<NM>([&](auto G_) {
      constexpr int G = decltype(G_)::value;
      constexpr int ks = G / (2 * NS * NJ), q = (G / (2 * NJ)) % NS, i = (G / NJ) % 2, j = G % NJ;
      acc[i][j] = mfma_16b<F16>(av[q][i], bv[ks][j], acc[i][j]);
      if constexpr (ks == 0) {                        // second-step fragments into the registers that have just been used up
        if constexpr (G < NJ) bv[1][G] = tr_frag_ld<LDB>(sbb + 16 * LDB + G * 64);
        if constexpr (j == NJ - 1) av[q][i] = tr_frag_ld<LDA>(sa + q * PLANE + 16 * LDA + i * 64);
      }
      constexpr int P0 = G * NP / NM, P1 = (G + 1) * NP / NM;
      static_for<P1 - P0>([&](auto D_) { piece(std::integral_constant<int, P0 + decltype(D_)::value>{}); });
      __builtin_amdgcn_sched_barrier(0);
    });
    lds_barrier();
  };
#endif

  if (t_begin < t_end) {
    gload(t_begin, Slot0());
    gload(t_begin + 1, Slot1());
    lstore(smem, Slot0());
    gload(t_begin + 2, Slot0());
    lds_barrier();
#ifndef TNW_SERIAL_STAGING
    cs[0] = csacc.x; cs[1] = csacc.y; cs[2] = csacc.z; cs[3] = csacc.w;      // the k-loop continues the same running sums
    {   // the prologue has loaded tiles t_begin .. t_begin+2 (clamped): the k-loop's first load is tile t_begin+3
      const int tl = min(t_begin + 2, t_end - 1);
#pragma unroll
      for (int i = 0; i < 2; ++i) offA[i] = 4u * (unsigned)((tl * QTN_BK + a_k + 16 * i) * ldA);
#pragma unroll
      for (int j = 0; j < NJ; ++j) offB[j] = (unsigned)((tl * QTN_BK + b_row[j]) * ldB);
    }
#endif
    int kt = t_begin;
    for (; kt + 1 < t_end; kt += 2) {
      step(kt, smem, smem + STAGE, Slot1());                               // tile kt+1 lives in slot 1
      step(kt + 1, smem + STAGE, smem, Slot0());                           // tile kt+2 in slot 0
    }
    if (kt < t_end) step(kt, smem, smem + STAGE, Slot1());
  }
#ifndef TNW_SERIAL_STAGING
  if (t_begin < t_end) csacc = make_float4(cs[0], cs[1], cs[2], cs[3]);
#endif
#ifdef TNW_TIMING
  if (blockIdx.x == 0 && lane == 0) {
    for (int q = 0; q < 6; ++q) g_tnw_dbg[wid][q] = tacc[q];
    g_tnw_dbg[wid][6] = clock64() - tstart;
    g_tnw_dbg[wid][7] = t_end - t_begin;
  }
#endif
  if (!direct) {
    float* W = p.ws + (int64_t)sidx * p.M * p.N;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int n = n0 + wn * 32 * NJ + j * 32 + l31;
      if (n >= p.N) continue;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
          if (m < p.M) W[(int64_t)m * p.N + n] = F16 ? acc[i][j][e] * inv_sE : acc[i][j][e];
        }
    }
  }
  float* red1 = reinterpret_cast<float*>(smem) + 16 * 32 * 4;     // 128 finished column sums behind the partials
  if (do_csum) {     // reduce the 16 row-groups that share a column quad, one writer per quad
    float4* red = reinterpret_cast<float4*>(smem);
    red[a_k * 32 + (tid & 31)] = csacc;
    __syncthreads();
    if (a_k == 0) {
      float4 t = red[tid & 31];
#pragma unroll
      for (int g = 1; g < 16; ++g) {
        const float4 u = red[g * 32 + (tid & 31)];
        t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
      }
      if (!direct) {
        if (a_ok) *reinterpret_cast<float4*>(p.csum + (int64_t)sidx * p.M + m0 + a_t) = t;
      } else {
        *reinterpret_cast<float4*>(red1 + a_t) = t;
      }
    }
    __syncthreads();
  }
  if (direct) {
    float* Cb = p.C + b0 * p.sC0 + b1 * p.sC1;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int n = n0 + wn * 32 * NJ + j * 32 + l31;
      if (n >= p.Nstore) continue;
      const float bf = p.baft ? p.baft[n + b1 * p.sBf1] : 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int ml = wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
          const int m = m0 + ml;
          if (m < p.Mstore) {
            float v = F16 ? acc[i][j][e] * inv_sE : acc[i][j][e];
            if (p.baft) v += red1[ml] * bf;
            Cb[(int64_t)m * p.ldc + n] = v;
          }
        }
    }
  }
}

template <int NJ, bool F16 = false>
__global__ __launch_bounds__(512) void qgemm_bf16s_tn_wide_kernel(QTnArgs p) {
  int lid, gby;
  xcd_remap_grid(lid, gby);
  tn_wide_body<NJ, F16>(p, lid, gby);
}

// ---- persistent direct-mode variant (attention dqkx: K = the 198 tokens of an image, 7 k-steps per tile) ---------------
// One workgroup walks `tpw` consecutive tiles of ONE outer batch entry (image): tile q -> inner batch b1 = q / (tiles_m *
// tiles_n) (the head), tile q % (...).  The k-steps of all its tiles form one continuous stream through the same
// two-stage LDS ring and two register prefetch slots as above: the loads of global step g+3 and the staging of step g+1
// run behind the MFMAs of step g whichever tiles those steps belong to, so a tile boundary costs its epilogue (the stores
// of the finished 128 x 384 tile) and nothing else -- no pipeline drain, no fresh memory round trip.  The one-tile-per-
// workgroup launch of the same problem (1536 workgroups of 7 k-steps on 256 CUs: six rounds of prologue + 7 steps +
// epilogue) took 178 us per DeiT-S block for 35 us of MFMA work.
template <int NJ, bool STK = false, bool F16 = false>
__device__ __forceinline__ void tn_wide_stream_body(const QTnArgs& p, const int chunk, const int b0, const int tpw) {
  constexpr int BM = 128, BN = 128 * NJ, NS = F16 ? 2 : 3;
  constexpr int LDA = QTN_LD;
  constexpr int LDB = BN * 2 + 64;
  constexpr int PLANE = QTN_BK * LDA;
  constexpr int STAGE = NS * PLANE + QTN_BK * LDB;
  constexpr int CPR = BN / 8;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];
  __shared__ __attribute__((aligned(16))) float sred[16 * 32 * 4 + BM + BN];   // column sums of a finished tile, its offsets
  __shared__ int srow[BM + 4];      // STK: element offset of every tile row inside the image's slab of C (-1: no such row), then
                                    // per 32-row block: 1 when the block is one head's 32 consecutive real rows
  const int tpi = p.tiles_m * p.tiles_n;                 // tiles per inner batch entry
  const int T = tpi * p.nb1;
  const int q0 = chunk * tpw, q1 = min(T, q0 + tpw);
  if (q0 >= q1) return;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 2, wn = wid & 3;
  const int l31 = lane & 31, lh = lane >> 5;
  const int nkt = (p.Ktok + QTN_BK - 1) / QTN_BK;
  const int a_k = tid >> 5, a_t = (tid & 31) * 4;
  const int ldA = (int)p.lda, ldB = (int)p.ldb;
  float sE = 1.f, inv_sE = 1.f;        // F16: the launch's power of two (see tn_wide_body)
  const unsigned c64 = 0x64646464u;
  if constexpr (F16) {
    const float m = fmaxf(block512_absmax(p.s, p.S, reinterpret_cast<float*>(smem), tid), 1e-5f) * 1.0001f;
    const float am = ofq_amax_load(p.amax);
    f16_plane_scale(am == am ? am * m : am, sE, inv_sE);
  }
  int b_row[NJ], b_col[NJ];
#pragma unroll
  for (int i = 0; i < NJ; ++i) {
    const int f = tid + 512 * i;
    b_row[i] = f / CPR;
    b_col[i] = (f % CPR) * 8;
  }
  unsigned rowA[2], rowB[NJ];                            // byte offsets of this lane's rows inside a k-step
#pragma unroll
  for (int i = 0; i < 2; ++i) rowA[i] = 4u * (unsigned)((a_k + 16 * i) * ldA);
#pragma unroll
  for (int j = 0; j < NJ; ++j) rowB[j] = (unsigned)(b_row[j] * ldB);
  const unsigned maxA = 4u * (unsigned)((p.Ktok - 1) * ldA), maxB = (unsigned)((p.Ktok - 1) * ldB);

  // ---- load cursor: the tile / k-step the NEXT global loads fetch --------------------------------------------------------
  // (tile bases are wave-uniform -> scalar registers; a lane adds one 32-bit offset: row part clamped to the last token,
  // column part zeroed for columns outside the matrix, whose products are never stored)
  int lq = q0, lk = 0;
  const char* LAs;                                       // cursor tile's A panel, column m0          (uniform)
  const char* LBs;                                       // cursor tile's B panel, column n0          (uniform)
  unsigned colA;                                         // this lane's column quad inside the tile, in bytes
  unsigned colB[NJ];
  bool la_ok;
  auto set_load_tile = [&](int q) {
    const int qc = min(q, q1 - 1);
    const int b1 = qc / tpi, t = qc - b1 * tpi;
    const int m0 = (t / p.tiles_n) * BM, n0 = (t % p.tiles_n) * BN;
    if constexpr (STK) {        // this lane's column quad = rows r .. r + 3 of the stacked output: head h, key m (stk_mp % 4 == 0)
      const int r = m0 + a_t;
      const int h = (int)(((unsigned)r * p.stk_magic) >> 20);
      const int m = r - h * p.stk_mp;
      la_ok = h < p.stk_h;
      LAs = reinterpret_cast<const char*>(p.A + b0 * p.sA0);
      colA = la_ok ? 4u * (unsigned)(h * (int)p.sA1 + m) : 0u;
    } else {
      la_ok = (m0 + a_t) < p.M;
      LAs = reinterpret_cast<const char*>(p.A + b0 * p.sA0 + b1 * p.sA1 + m0);
      colA = la_ok ? 4u * (unsigned)a_t : 0u;
    }
    LBs = reinterpret_cast<const char*>(p.B + b0 * p.sB0 + b1 * p.sB1 + n0);
#pragma unroll
    for (int j = 0; j < NJ; ++j) colB[j] = (n0 + b_col[j]) < p.N ? (unsigned)b_col[j] : 0u;
  };
  set_load_tile(lq);
  auto advance_cursor = [&]() {
    if (++lk == nkt) {
      lk = 0;
      ++lq;
      set_load_tile(lq);
    }
  };

  f32x4v ra[2][2];
  float rs[2][2];
  bool rok[2][2];
  u32x2v rb[2][NJ];
  // prologue-style (un-interleaved) load / stage, used for the first three steps of the stream only
  auto gload = [&](auto SLOT) {
    constexpr int sl = decltype(SLOT)::value;
    const bool live = lq < q1;
    const int k0 = lk * QTN_BK;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int k = k0 + a_k + 16 * i;
      rok[sl][i] = la_ok && k < p.Ktok && live;
      ra[sl][i] = *reinterpret_cast<const f32x4v*>(LAs + (min(4u * (unsigned)(k0 * ldA) + rowA[i], maxA) + colA));
      rs[sl][i] = p.s[min(k, p.S - 1)];
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      rb[sl][j] = *reinterpret_cast<const u32x2v*>(LBs + (min((unsigned)(k0 * ldB) + rowB[j], maxB) + colB[j]));
    advance_cursor();
  };
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
  auto lstore = [&](unsigned char* sb, auto SLOT) {
    constexpr int sl = decltype(SLOT)::value;
#pragma unroll
    for (int i = 0; i < 2; ++i) asm volatile("" : "+v"(ra[sl][i]), "+v"(rs[sl][i]));
#pragma unroll
    for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(rb[sl][j]));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float sc = rok[sl][i] ? (F16 ? ofq_lsq_eff_scale(rs[sl][i], p.gscale) * sE : ofq_lsq_eff_scale(rs[sl][i], p.gscale)) : 0.f;
      const float okf = rok[sl][i] ? 1.f : 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) cs[e] = valu_fma(ra[sl][i][e], okf, cs[e]);
      const f32x2v v01 = {ra[sl][i][0], ra[sl][i][1]}, v23 = {ra[sl][i][2], ra[sl][i][3]};
      unsigned lo[NS], hi[NS];
      if constexpr (F16) {
        const f32x2v x01 = v01 * sc, x23 = v23 * sc;
        split2_f16(x01[0], x01[1], lo[0], lo[1]);
        split2_f16(x23[0], x23[1], hi[0], hi[1]);
      } else {
        split_pair_bf16<NS>(v01 * sc, lo);
        split_pair_bf16<NS>(v23 * sc, hi);
      }
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        uint2 w;
        w.x = lo[q];
        w.y = hi[q];
        *reinterpret_cast<uint2*>(&sb[q * PLANE + (a_k + 16 * i) * LDA + a_t * 2]) = w;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      unsigned bw[4];
      if constexpr (F16) {
        valu_cvt4_i8_f16(rb[sl][j][0], c64, bw[0], bw[1]);
        valu_cvt4_i8_f16(rb[sl][j][1], c64, bw[2], bw[3]);
      } else {
        valu_cvt4_i8_bf16(rb[sl][j][0], bw[0], bw[1]);
        valu_cvt4_i8_bf16(rb[sl][j][1], bw[2], bw[3]);
      }
      *reinterpret_cast<uint4*>(&sb[NS * PLANE + b_row[j] * LDB + b_col[j] * 2]) = make_uint4(bw[0], bw[1], bw[2], bw[3]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  f32x16q acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int p16 = lane & 15;
  const int fr_a = (8 * lh + (p16 >> 2)) * LDA + (16 * ((lane >> 4) & 1) + 4 * (p16 & 3)) * 2;
  const int fr_b = (8 * lh + (p16 >> 2)) * LDB + (16 * ((lane >> 4) & 1) + 4 * (p16 & 3)) * 2;
  static_assert(QTN_BK == 32, "two MFMA steps per k-step");

  using Slot0 = std::integral_constant<int, 0>;
  using Slot1 = std::integral_constant<int, 1>;
  constexpr int NM = 4 * NS * NJ, NPA = F16 ? 11 : 18, NPB = 3, NP = 2 * NPA + NPB * NJ + 2 + NJ;      // (piece lists: tn_wide_body)
  // one k-step of the stream: MFMA on `cur`, staging of the next step (register slot SLOT -> `nxt`), loads at the cursor
  // into the freed slot; the piece list is the one of tn_wide_body
  bool skip_i1 = false;           // this wave's second 32-row block lies outside the matrix in the tile being computed
  auto step = [&](const unsigned char* cur, unsigned char* nxt, auto SLOT) {
    constexpr int sl = decltype(SLOT)::value;
    bf16x8 av[NS][2], bv[2][NJ];
    const unsigned char* sa = &cur[fr_a + wm * 64 * 2];
    const unsigned char* sbb = &cur[NS * PLANE + fr_b + wn * 32 * NJ * 2];
#pragma unroll
    for (int j = 0; j < NJ; ++j) bv[0][j] = tr_frag_ld<LDB>(sbb + j * 64);
#pragma unroll
    for (int q = 0; q < NS; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i) av[q][i] = tr_frag_ld<LDA>(sa + q * PLANE + i * 64);
    __builtin_amdgcn_sched_barrier(0);
    float sc = 0.f, okf = 0.f, x_ = 0.f, r1_ = 0.f, p0v[2], p1v[2], r2v[2];
    unsigned lo[NS], hi[NS], bw[4];
    const bool live = lq < q1;
    const int rows_left = p.Ktok - lk * QTN_BK;
    const unsigned kofsA = 4u * (unsigned)(lk * QTN_BK * ldA), kofsB = (unsigned)(lk * QTN_BK * ldB);
    const int kbase = lk * QTN_BK;
    auto piece = [&](auto P_) {
      constexpr int P = decltype(P_)::value;
      if constexpr (P < 2 * NPA) {
        constexpr int i = P / NPA, r = P % NPA;
        if constexpr (r == 0) {
          asm volatile("" : "+v"(ra[sl][i]), "+v"(rs[sl][i]));
          const float e = valu_eff_scale(rs[sl][i], p.gscale);
          sc = rok[sl][i] ? (F16 ? e * sE : e) : 0.f;
          okf = rok[sl][i] ? 1.f : 0.f;
        } else if constexpr (F16) {
          if constexpr (r < 9) {
            constexpr int pr = (r - 1) / 4, st = (r - 1) % 4, e = pr * 2;
            if constexpr (st == 0) {
              valu_mul2(ra[sl][i][e], sc, ra[sl][i][e + 1], sc, x_, r1_);
              cs[e] = valu_fma(ra[sl][i][e], okf, cs[e]);
              cs[e + 1] = valu_fma(ra[sl][i][e + 1], okf, cs[e + 1]);
            }
            if constexpr (st == 1) (pr == 0 ? lo : hi)[0] = valu_cvt_pk_f16(x_, r1_);
            if constexpr (st == 2) valu_resid2_f16((pr == 0 ? lo : hi)[0], x_, r1_, p0v[0], p0v[1]);
            if constexpr (st == 3) (pr == 0 ? lo : hi)[1] = valu_cvt_pk_f16(p0v[0], p0v[1]);
          } else {
            constexpr int q = r - 9;
            uint2 w;
            w.x = lo[q];
            w.y = hi[q];
            *reinterpret_cast<uint2*>(&nxt[q * PLANE + (a_k + 16 * i) * LDA + a_t * 2]) = w;
          }
        } else if constexpr (r < 15) {
          constexpr int pr = (r - 1) / 7, rr = (r - 1) % 7;
          if constexpr (rr < 6) {
            constexpr int el = rr / 3, st = rr % 3, e = pr * 2 + el;
            if constexpr (st == 0) {
              valu_mul_hi16(ra[sl][i][e], sc, x_, p0v[el]);
              cs[e] = valu_fma(ra[sl][i][e], okf, cs[e]);
            }
            if constexpr (st == 1) valu_sub_hi16(x_, p0v[el], r1_, p1v[el]);
            if constexpr (st == 2) { r2v[el] = valu_sub(r1_, p1v[el]); }
          } else {
            valu_pack3_hi16(p0v, p1v, r2v, pr == 0 ? lo : hi);
          }
        } else {
          constexpr int q = r - 15;
          uint2 w;
          w.x = lo[q];
          w.y = hi[q];
          *reinterpret_cast<uint2*>(&nxt[q * PLANE + (a_k + 16 * i) * LDA + a_t * 2]) = w;
        }
      } else if constexpr (P < 2 * NPA + NPB * NJ) {
        constexpr int j = (P - 2 * NPA) / NPB, r = (P - 2 * NPA) % NPB;
        if constexpr (r == 0) {
          asm volatile("" : "+v"(rb[sl][j]));
          if constexpr (F16) valu_cvt4_i8_f16(rb[sl][j][0], c64, bw[0], bw[1]);
          else valu_cvt4_i8_bf16(rb[sl][j][0], bw[0], bw[1]);
        } else if constexpr (r == 1) {
          if constexpr (F16) valu_cvt4_i8_f16(rb[sl][j][1], c64, bw[2], bw[3]);
          else valu_cvt4_i8_bf16(rb[sl][j][1], bw[2], bw[3]);
        } else {
          *reinterpret_cast<uint4*>(&nxt[NS * PLANE + b_row[j] * LDB + b_col[j] * 2]) = make_uint4(bw[0], bw[1], bw[2], bw[3]);
        }
      } else if constexpr (P < 2 * NPA + NPB * NJ + 2) {
        constexpr int i = P - 2 * NPA - NPB * NJ;
        rok[sl][i] = la_ok && (a_k + 16 * i) < rows_left && live;
        ra[sl][i] = *reinterpret_cast<const f32x4v*>(LAs + (min(kofsA + rowA[i], maxA) + colA));
        rs[sl][i] = p.s[min(kbase + a_k + 16 * i, p.S - 1)];
      } else {
        constexpr int j = P - 2 * NPA - NPB * NJ - 2;
        rb[sl][j] = *reinterpret_cast<const u32x2v*>(LBs + (min(kofsB + rowB[j], maxB) + colB[j]));
      }
    };
    static_for<NM>([&](auto G_) {
      constexpr int G = decltype(G_)::value;
      constexpr int ks = G / (2 * NS * NJ), q = (G / (2 * NJ)) % NS, i = (G / NJ) % 2, j = G % NJ;
#ifdef TNS_SKIP_PAD_BLOCKS
      if (i == 0 || !skip_i1)
#endif
        acc[i][j] = mfma_16b<F16>(av[q][i], bv[ks][j], acc[i][j]);
      if constexpr (ks == 0) {
        if constexpr (G < NJ) bv[1][G] = tr_frag_ld<LDB>(sbb + 16 * LDB + G * 64);
        if constexpr (j == NJ - 1) av[q][i] = tr_frag_ld<LDA>(sa + q * PLANE + 16 * LDA + i * 64);
      }
      constexpr int P0 = G * NP / NM, P1 = (G + 1) * NP / NM;
      static_for<P1 - P0>([&](auto D_) { piece(std::integral_constant<int, P0 + decltype(D_)::value>{}); });
      __builtin_amdgcn_sched_barrier(0);
    });
    lds_barrier();
    advance_cursor();
  };
  auto set_compute_tile = [&](int q) {
    const int t = q % tpi;
    skip_i1 = ((t / p.tiles_n) * BM + wm * 64 + 32) >= p.Mstore;
  };

  // the finished tile q: C (+ colsum_k(A)[m] * baft[n]) to global memory, accumulators back to zero.  Nothing here may
  // wait on the vector-memory counter: the prefetch loads of the next steps are in flight (the offset vector comes from
  // LDS, where bf_reg -- loaded at the tile's first step -- was parked before the tile's last barrier).
  auto aval = [&](float v) -> float { return F16 ? v * inv_sE : v; };      // an accumulator in the units of the product
  auto epilogue = [&](int q) {
    const int b1 = q / tpi, t = q - b1 * tpi;
    const int m0 = (t / p.tiles_n) * BM, n0 = (t % p.tiles_n) * BN;
    float* red1 = sred + 16 * 32 * 4;
    float* sbf = red1 + BM;
    if (p.baft) {
      float4* red = reinterpret_cast<float4*>(sred);       // the tile's partial column sums: stored before its last step
      if (a_k == 0) {
        float4 tt = red[tid & 31];
#pragma unroll
        for (int g = 1; g < 16; ++g) {
          const float4 u = red[g * 32 + (tid & 31)];
          tt.x += u.x; tt.y += u.y; tt.z += u.z; tt.w += u.w;
        }
        *reinterpret_cast<float4*>(red1 + a_t) = tt;
      }
      lds_barrier();
    }
    // uniform tile base + one 32-bit lane offset per store (an image's slab of C is far below 2^31 bytes: host check).
    // The lane ids pass through an empty volatile asm: otherwise the 32 row offsets are loop-invariant, get hoisted out
    // of the k-step stream and cost 60+ live VGPRs there (the kernel spilled 170 registers)
    int l31e = l31, lhe = lh;
    asm volatile("" : "+v"(l31e), "+v"(lhe));
    float* Cs = STK ? p.C + b0 * p.sC0 + n0 : p.C + b0 * p.sC0 + b1 * p.sC1 + (int64_t)m0 * p.ldc + n0;
    const int ldc = (int)p.ldc;
    const bool full_n = (n0 + BN) <= p.Nstore;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int rv = p.Mstore - (m0 + wm * 64 + i * 32);         // valid rows of this wave's 32-row block (wave-uniform)
      int blk_base = 0;
      if constexpr (STK) {      // srow was filled before the tile's last barrier (pre()): whole block of one head -> its base
        const int whole = __builtin_amdgcn_readfirstlane(srow[BM + wm * 2 + i]);
        blk_base = __builtin_amdgcn_readfirstlane(srow[wm * 64 + i * 32]);
        rv = whole ? 32 : 1;                               // (1: take the row-by-row path below)
      }
      if (rv <= 0) continue;                               // all padding: its MFMAs were skipped, acc stayed zero
      float rsum[16];
#pragma unroll
      for (int e = 0; e < 16; ++e)
        rsum[e] = p.baft ? red1[wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhe] : 0.f;
      if (STK && full_n && rv >= 32) {                     // one head's 32 consecutive rows: uniform base, straight stores
        float* Cb = Cs + blk_base;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int nl = wn * 32 * NJ + j * 32 + l31e;
          const float bf = p.baft ? sbf[nl] : 0.f;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int mr = (e & 3) + 8 * (e >> 2) + 4 * lhe;
            Cb[mr * ldc + nl] = p.baft ? aval(acc[i][j][e]) + rsum[e] * bf : aval(acc[i][j][e]);
            acc[i][j][e] = 0.f;
          }
        }
      } else if (STK) {                                    // a block that straddles two heads or holds pad rows
        int off[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) off[e] = srow[wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhe];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int nl = wn * 32 * NJ + j * 32 + l31e;
          const bool nok = (n0 + nl) < p.Nstore;
          const float bf = p.baft ? sbf[nl] : 0.f;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            if (nok && off[e] >= 0) Cs[off[e] + nl] = p.baft ? aval(acc[i][j][e]) + rsum[e] * bf : aval(acc[i][j][e]);
            acc[i][j][e] = 0.f;
          }
        }
      } else if (full_n && rv >= 32) {                     // interior block: straight stores
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int nl = wn * 32 * NJ + j * 32 + l31e;
          const float bf = p.baft ? sbf[nl] : 0.f;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int ml = wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhe;
            Cs[ml * ldc + nl] = p.baft ? aval(acc[i][j][e]) + rsum[e] * bf : aval(acc[i][j][e]);
            acc[i][j][e] = 0.f;
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int nl = wn * 32 * NJ + j * 32 + l31e;
          const bool nok = (n0 + nl) < p.Nstore;
          const float bf = p.baft ? sbf[nl] : 0.f;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int mr = (e & 3) + 8 * (e >> 2) + 4 * lhe;
            const int ml = wm * 64 + i * 32 + mr;
            if (nok && mr < rv) Cs[ml * ldc + nl] = p.baft ? aval(acc[i][j][e]) + rsum[e] * bf : aval(acc[i][j][e]);
            acc[i][j][e] = 0.f;
          }
        }
      }
    }
    if (p.baft || STK) lds_barrier();      // sred / red1 / sbf / srow are rewritten during the next tile (at its last k-step: its first one when K <= 32)
  };

  gload(Slot0());
  gload(Slot1());
  lstore(smem, Slot0());
  gload(Slot0());
  lds_barrier();
  int cq = q0, ck = 0;
  const int G = (q1 - q0) * nkt;
  float bf_reg = 0.f;             // this thread's element of the tile's offset vector baft[n0 .. n0 + BN)
  auto load_bf = [&](int q) {
    if (p.baft && tid < BN) {
      const int b1 = q / tpi, t = q - b1 * tpi;
      const int n = (t % p.tiles_n) * BN + tid;
      bf_reg = p.baft[min(n, p.Nstore - 1) + b1 * p.sBf1];
    }
  };
  auto pre = [&]() {
    if (ck == 0) load_bf(cq);
    if (ck == nkt - 1) {          // every k-step of tile cq has been staged: the staging of this step feeds the next tile
      // (sred is free: the previous tile's epilogue ended with an LDS barrier; this step's barrier publishes the stores)
      reinterpret_cast<float4*>(sred)[a_k * 32 + (tid & 31)] = make_float4(cs[0], cs[1], cs[2], cs[3]);
      if (p.baft && tid < BN) sred[16 * 32 * 4 + BM + tid] = bf_reg;
      if constexpr (STK) {
        if (tid < BM) {
          const int m0c = ((cq % tpi) / p.tiles_n) * BM;
          const int r = m0c + tid;
          const int h = (int)(((unsigned)r * p.stk_magic) >> 20), m = r - h * p.stk_mp;
          srow[tid] = (h < p.stk_h && m < p.stk_valid) ? h * (int)p.sC1 + m * (int)p.ldc : -1;
          if ((tid & 31) == 0) {
            const int r1 = r + 31;
            const int h1 = (int)(((unsigned)r1 * p.stk_magic) >> 20), m1 = r1 - h1 * p.stk_mp;
            srow[BM + (tid >> 5)] = (h1 == h && h < p.stk_h && m1 < p.stk_valid) ? 1 : 0;
          }
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) cs[e] = 0.f;
    }
  };
  auto post = [&]() {
    if (++ck == nkt) {
      epilogue(cq);
      ck = 0;
      ++cq;
      set_compute_tile(min(cq, q1 - 1));
    }
  };
  set_compute_tile(q0);
  int g = 0;
  for (; g + 1 < G; g += 2) {
    pre();
    step(smem, smem + STAGE, Slot1());
    post();
    pre();
    step(smem + STAGE, smem, Slot0());
    post();
  }
  if (g < G) {
    pre();
    step(smem, smem + STAGE, Slot1());
    post();
  }
}

template <int NJ, bool STK = false, bool F16 = false>
__global__ __launch_bounds__(512) void qgemm_bf16s_tn_wide_stream_kernel(QTnArgs p, int tpw, int stagger) {
  int chunk, b0;
  xcd_remap_grid(chunk, b0);
  // All workgroups start together and every tile takes the same time, so without this they would all reach their
  // epilogues (196 KB of stores each, 50 MB chip-wide) in the same microsecond, six times per launch -- and on gfx9 the
  // vector-memory counter is shared by loads and stores: a wave cannot consume a prefetched load that it issued after its
  // epilogue stores before those stores have retired, so the store burst stalls the k-step stream (measured: 15 us per
  // tile boundary).  Phase-shifting the workgroups by a fraction of a tile period spreads the stores over the launch.
  if (stagger > 0) {
    const int phase = (blockIdx.y * gridDim.x + blockIdx.x) % stagger;
    for (int i = 0; i < phase; ++i) __builtin_amdgcn_s_sleep(70);           // ~2.2 us each (64 x 70 clocks)
  }
  tn_wide_stream_body<NJ, STK, F16>(p, chunk, b0, tpw);
}

// Several split-K problems in one launch.  The weight-gradient GEMMs of the linear layers have no consumer before the
// optimiser step (or the gradient bucket's all-reduce), so the host defers them (functional.queue_dw) and launches the
// ones of a whole transformer block together: 45-48 tiles x split 5 instead of five launches of 3-18 tiles x split
// 14-85.  A workgroup then owns ~160 k-steps instead of 9-57 (prologue, epilogue and the first memory round trip are paid
// once), and the partials of a block shrink from ~250 MB to ~47 MB (256 workgroups x 196 KB per LAUNCH, whatever the
// problem: fewer launches, fewer partials).
#define QTN_GROUP_MAX 8
struct QTnGroup {
  QTnArgs job[QTN_GROUP_MAX];
  int wg_start[QTN_GROUP_MAX + 1];      // first workgroup of job j in the launch order (after the XCD remap)
  int njobs;
};
template <int NJ, bool F16 = false>
__global__ __launch_bounds__(512) void qgemm_bf16s_tn_wide_group_kernel(QTnGroup g) {
  int L, gby;
  xcd_remap_grid(L, gby);
  int j = 0;
#pragma unroll
  for (int q = 1; q < QTN_GROUP_MAX; ++q) j += (q < g.njobs && L >= g.wg_start[q]) ? 1 : 0;
  tn_wide_body<NJ, F16>(g.job[j], L - g.wg_start[j], 0);
}

// Split-K reduce, latency-parallel version (N % 4 == 0, N >= 256): the row-per-block kernel below walks the `split`
// partials of an element four at a time, i.e. split/4 dependent memory round trips in a launch of only M blocks (384 rows =
// 1.5 blocks per CU) -- 86 % of its wave time is parked.  Here a block owns 64 float4 chunks of the output and its four
// waves each sum every fourth partial, four loads in flight (split/16 round trips), combined through LDS in a fixed order
// (deterministic).  db[o] = sum_s csum[s][o] is computed by the block(s) touching row o (written by the one holding
// the row's first chunk); dW[o][c] += db[o] * baft[c] as before.
__device__ __forceinline__ void tn_reduce4_body(const int bx, const float* __restrict__ ws, float* __restrict__ C,
                                                const float* __restrict__ csum, float* __restrict__ db,
                                                const float* __restrict__ baft, int M, int N, int split) {
  __shared__ float4 red[3][64];
  __shared__ float dbs[2];
  const int tx = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int N4 = N >> 2;
  const int64_t MN = (int64_t)M * N;
  const int g0 = bx * 64;
  const int total = M * N4;
  const int o_first = g0 / N4, o_last = min(g0 + 63, total - 1) / N4;      // N4 >= 64: at most two rows per block
  if (part < 2) {
    const int o = part == 0 ? o_first : o_last;
    float v = 0.f;
    if (csum) {
      for (int s = tx; s < split; s += 64) v += csum[(int64_t)s * M + o];
      v = ofq_wave_sum(v);
      const int gfirst = o * N4;                                             // the block holding chunk (o, 0) publishes db[o]
      if (tx == 0 && gfirst >= g0 && gfirst < g0 + 64 && (part == 0 || o_last != o_first)) db[o] = v;
    } else if (db) {
      v = db[o];
    }
    if (tx == 0) dbs[part] = v;
  }
  const int g = min(g0 + tx, total - 1);
  const int o = g / N4, c4 = g - o * N4;
  const float* p = ws + (int64_t)o * N + 4 * c4;
  float4 a[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) a[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int s = part; s < split; s += 16) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int su = s + 4 * u;
      const float4 v = *reinterpret_cast<const float4*>(p + (int64_t)min(su, split - 1) * MN);     // unconditional load
      const float m = su < split ? 1.f : 0.f;
      a[u].x += v.x * m; a[u].y += v.y * m; a[u].z += v.z * m; a[u].w += v.w * m;
    }
  }
  float4 t;
  t.x = (a[0].x + a[1].x) + (a[2].x + a[3].x);
  t.y = (a[0].y + a[1].y) + (a[2].y + a[3].y);
  t.z = (a[0].z + a[1].z) + (a[2].z + a[3].z);
  t.w = (a[0].w + a[1].w) + (a[2].w + a[3].w);
  if (part > 0) red[part - 1][tx] = t;
  __syncthreads();
  if (part == 0 && g0 + tx < total) {
    const float4 r1 = red[0][tx], r2 = red[1][tx], r3 = red[2][tx];
    t.x = (t.x + r1.x) + (r2.x + r3.x);
    t.y = (t.y + r1.y) + (r2.y + r3.y);
    t.z = (t.z + r1.z) + (r2.z + r3.z);
    t.w = (t.w + r1.w) + (r2.w + r3.w);
    if (baft && (db || csum)) {
      const float dbo = dbs[o == o_first ? 0 : 1];
      const float4 bf = *reinterpret_cast<const float4*>(baft + 4 * c4);
      t.x += dbo * bf.x; t.y += dbo * bf.y; t.z += dbo * bf.z; t.w += dbo * bf.w;
    }
    *reinterpret_cast<float4*>(C + (int64_t)o * N + 4 * c4) = t;
  }
}

__global__ __launch_bounds__(256) void qgemm_tn_reduce4_kernel(const float* __restrict__ ws, float* __restrict__ C,
                                                               const float* __restrict__ csum, float* __restrict__ db,
                                                               const float* __restrict__ baft, int M, int N, int split) {
  tn_reduce4_body(blockIdx.x, ws, C, csum, db, baft, M, N, split);
}

// the reduces of a grouped launch (qgemm_bf16s_tn_wide_group_kernel) in one launch
struct QTnRedJob { const float* ws; float* C; const float* csum; float* db; const float* baft; int M, N, split, blk_start; };
struct QTnRedGroup { QTnRedJob job[QTN_GROUP_MAX]; int njobs; };
__global__ __launch_bounds__(256) void qgemm_tn_reduce4_group_kernel(QTnRedGroup g) {
  const int bx = blockIdx.x;
  int j = 0;
#pragma unroll
  for (int q = 1; q < QTN_GROUP_MAX; ++q) j += (q < g.njobs && bx >= g.job[q].blk_start) ? 1 : 0;
  const QTnRedJob& r = g.job[j];
  tn_reduce4_body(bx - r.blk_start, r.ws, r.C, r.csum, r.db, r.baft, r.M, r.N, r.split);
}

// one block per output row o:  db[o] = sum_s csum[s][o] (when the GEMM produced column sums), then
// dW[o][c] = sum_s ws[s][o][c] + db[o] * baft[c]  -- fixed order, four partial sums in flight per thread
__global__ __launch_bounds__(256) void qgemm_tn_reduce_kernel(const float* __restrict__ ws, float* __restrict__ C,
                                                              const float* __restrict__ csum, float* __restrict__ db,
                                                              const float* __restrict__ baft, int M, int N, int split) {
  __shared__ float dbs;
  const int o = blockIdx.x;
  const int64_t MN = (int64_t)M * N;
  if (threadIdx.x < 64) {
    float v = 0.f;
    if (csum) {
      for (int s = threadIdx.x; s < split; s += 64) v += csum[(int64_t)s * M + o];
      v = ofq_wave_sum(v);
      if (threadIdx.x == 0) db[o] = v;
    } else if (db) {
      v = db[o];
    }
    if (threadIdx.x == 0) dbs = v;
  }
  __syncthreads();
  const float dbo = dbs;
  for (int c = threadIdx.x; c < N; c += 256) {
    const float* p = ws + (int64_t)o * N + c;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int s = 0;
    for (; s + 3 < split; s += 4) {
      a0 += p[(int64_t)s * MN];
      a1 += p[(int64_t)(s + 1) * MN];
      a2 += p[(int64_t)(s + 2) * MN];
      a3 += p[(int64_t)(s + 3) * MN];
    }
    for (; s < split; ++s) a0 += p[(int64_t)s * MN];
    float acc = (a0 + a1) + (a2 + a3);
    if (baft && (db || csum)) acc += dbo * baft[c];
    C[(int64_t)o * N + c] = acc;
  }
}

extern "C" size_t ofq_qgemm_bf16s_tn_ws_bytes(int64_t M, int64_t N, int split) {
  return (size_t)split * M * (N + 1) * sizeof(float);       // partial products + partial column sums
}

extern "C" int ofq_qgemm_bf16s_tn(const float* dY, const int8_t* codes, float* dW, const float* lsq_s, int64_t S,
                                  float gscale, float* db, int compute_db, const float* baft, int64_t Ktok, int64_t M,
                                  int64_t N, int64_t lda, int64_t ldb, int split, void* ws, size_t ws_bytes, const void* amax,
                                  ofq_stream_t stream) {
  if (!dY || !codes || !dW || !lsq_s || !ws || Ktok <= 0 || M <= 0 || N <= 0 || S <= 0 || split < 1) return OFQ_EINVAL;
  if ((M & 3) || (N & 15) || (lda & 3) || (ldb & 15) || !al16(dY) || !al16(codes) || Ktok >= (1ll << 30)) return OFQ_EINVAL;
  if (ws_bytes < ofq_qgemm_bf16s_tn_ws_bytes(M, N, split)) return OFQ_ENOWS;
  QTnArgs a = {};
  a.A = dY; a.B = codes; a.ws = (float*)ws; a.s = lsq_s; a.lda = lda; a.ldb = ldb;
  a.M = (int)M; a.N = (int)N; a.Ktok = (int)Ktok; a.S = (int)S; a.split = split;
  a.tiles_m = (int)ceil_div(M, 128); a.tiles_n = (int)ceil_div(N, 128); a.gscale = gscale; a.nb1 = 1;
  a.amax = (const unsigned*)amax;        // (used by the wide kernels; the narrow one keeps its three bf16 planes)
  if (compute_db && !db) return OFQ_EINVAL;
  a.csum = compute_db ? (float*)ws + (size_t)split * M * N : nullptr;
  hipStream_t st = (hipStream_t)stream;
  static const bool narrow_only = getenv("OFQ_TN_NARROW") != nullptr;      // A/B switch for tools/tn_bench.py
  if (N > 128 && (N & 7) == 0 && !narrow_only && S >= QTN_BK && Ktok * lda < (1ll << 31) && Ktok * ldb < (1ll << 31)) {
    // wide tile: one dY split feeds three (two when N is not a multiple of 384) 128-column blocks
    if (N % 384 == 0) {
      a.tiles_n = (int)(N / 384);
      const dim3 grid((unsigned)(a.tiles_m * a.tiles_n * split));
      if (amax) hipLaunchKernelGGL((qgemm_bf16s_tn_wide_kernel<3, true>), grid, dim3(512), 0, st, a);
      else hipLaunchKernelGGL((qgemm_bf16s_tn_wide_kernel<3, false>), grid, dim3(512), 0, st, a);
    } else {
      a.tiles_n = (int)ceil_div(N, 256);
      const dim3 grid((unsigned)(a.tiles_m * a.tiles_n * split));
      if (amax) hipLaunchKernelGGL((qgemm_bf16s_tn_wide_kernel<2, true>), grid, dim3(512), 0, st, a);
      else hipLaunchKernelGGL((qgemm_bf16s_tn_wide_kernel<2, false>), grid, dim3(512), 0, st, a);
    }
  } else {
    hipLaunchKernelGGL(qgemm_bf16s_tn_kernel<false>, dim3((unsigned)(a.tiles_m * a.tiles_n * split)), dim3(256), 0, st, a);
  }
  OFQ_LAUNCH_CHECK();
  static const bool red_rows = getenv("OFQ_TN_REDUCE_ROWS") != nullptr;      // A/B switch (tools/)
  if ((N & 3) == 0 && N >= 256 && !red_rows)
    hipLaunchKernelGGL(qgemm_tn_reduce4_kernel, dim3((unsigned)ceil_div(M * (N / 4), 64)), dim3(256), 0, st, (const float*)ws, dW,
                       compute_db ? (const float*)a.csum : (const float*)nullptr, db, baft, (int)M, (int)N, split);
  else
    hipLaunchKernelGGL(qgemm_tn_reduce_kernel, dim3((unsigned)M), dim3(256), 0, st, (const float*)ws, dW,
                       compute_db ? (const float*)a.csum : (const float*)nullptr, db, baft, (int)M, (int)N, split);
  OFQ_LAUNCH_CHECK();
  return 0;
}

static bool tn_wide_ok(int64_t Ktok, int64_t N, int64_t S, int64_t lda, int64_t ldb) {
  return N > 128 && (N & 7) == 0 && S >= QTN_BK && Ktok * lda < (1ll << 31) && Ktok * ldb < (1ll << 31);
}

extern "C" size_t ofq_qgemm_bf16s_tn_group_ws_bytes(const ofq_tn_job* jobs, int njobs, int split) {
  size_t b = 0;
  for (int j = 0; jobs && j < njobs; ++j) b += ofq_qgemm_bf16s_tn_ws_bytes(jobs[j].M, jobs[j].N, split);
  return b;
}

// Several weight-gradient GEMMs (same semantics as ofq_qgemm_bf16s_tn, one ofq_tn_job each) in one GEMM launch and one
// reduce launch: see qgemm_bf16s_tn_wide_group_kernel.  Every job must be wide-tile eligible and all of one tile class
// (N % 384 == 0 for all, or for none); the common `split` is the caller's choice (about 256 / total tiles).
extern "C" int ofq_qgemm_bf16s_tn_group(const ofq_tn_job* jobs, int njobs, int split, void* ws, size_t ws_bytes,
                                        ofq_stream_t stream) {
  if (!jobs || njobs < 1 || njobs > QTN_GROUP_MAX || split < 1 || !ws) return OFQ_EINVAL;
  if (ws_bytes < ofq_qgemm_bf16s_tn_group_ws_bytes(jobs, njobs, split)) return OFQ_ENOWS;
  QTnGroup g = {};
  QTnRedGroup r = {};
  const bool three = jobs[0].N % 384 == 0;
  float* wsf = (float*)ws;
  int wg = 0, blk = 0;
  for (int j = 0; j < njobs; ++j) {
    const ofq_tn_job& q = jobs[j];
    if (!q.dY || !q.codes || !q.dW || !q.lsq_s || q.Ktok <= 0 || q.M <= 0 || q.N <= 0 || q.S <= 0) return OFQ_EINVAL;
    if ((q.M & 3) || (q.N & 15) || (q.lda & 3) || (q.ldb & 15) || !al16(q.dY) || !al16(q.codes) || q.Ktok >= (1ll << 30))
      return OFQ_EINVAL;
    if (!tn_wide_ok(q.Ktok, q.N, q.S, q.lda, q.ldb) || (q.N % 384 == 0) != three || q.N < 256) return OFQ_EINVAL;
    if (q.compute_db && !q.db) return OFQ_EINVAL;
    QTnArgs& a = g.job[j];
    a.A = q.dY; a.B = q.codes; a.ws = wsf; a.s = q.lsq_s; a.lda = q.lda; a.ldb = q.ldb;
    a.M = (int)q.M; a.N = (int)q.N; a.Ktok = (int)q.Ktok; a.S = (int)q.S; a.split = split;
    a.tiles_m = (int)ceil_div(q.M, 128); a.tiles_n = three ? (int)(q.N / 384) : (int)ceil_div(q.N, 256);
    a.gscale = q.gscale; a.nb1 = 1;
    a.amax = (const unsigned*)q.amax;
    if ((q.amax != nullptr) != (jobs[0].amax != nullptr)) return OFQ_EINVAL;      // one operand form per launch
    a.csum = q.compute_db ? wsf + (size_t)split * q.M * q.N : nullptr;
    g.wg_start[j] = wg;
    wg += a.tiles_m * a.tiles_n * split;
    QTnRedJob& rj = r.job[j];
    rj.ws = wsf; rj.C = q.dW; rj.csum = a.csum; rj.db = q.db; rj.baft = q.baft;
    rj.M = a.M; rj.N = a.N; rj.split = split; rj.blk_start = blk;
    blk += (int)ceil_div(q.M * (q.N / 4), 64);
    wsf += (size_t)split * q.M * (q.N + 1);
  }
  g.wg_start[njobs] = wg;
  g.njobs = r.njobs = njobs;
  hipStream_t st = (hipStream_t)stream;
  const bool f16 = jobs[0].amax != nullptr;
  if (three) {
    if (f16) hipLaunchKernelGGL((qgemm_bf16s_tn_wide_group_kernel<3, true>), dim3((unsigned)wg), dim3(512), 0, st, g);
    else hipLaunchKernelGGL((qgemm_bf16s_tn_wide_group_kernel<3, false>), dim3((unsigned)wg), dim3(512), 0, st, g);
  } else {
    if (f16) hipLaunchKernelGGL((qgemm_bf16s_tn_wide_group_kernel<2, true>), dim3((unsigned)wg), dim3(512), 0, st, g);
    else hipLaunchKernelGGL((qgemm_bf16s_tn_wide_group_kernel<2, false>), dim3((unsigned)wg), dim3(512), 0, st, g);
  }
  OFQ_LAUNCH_CHECK();
  hipLaunchKernelGGL(qgemm_tn_reduce4_group_kernel, dim3((unsigned)blk), dim3(256), 0, st, r);
  OFQ_LAUNCH_CHECK();
  return 0;
}

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

// ------------------------------------------------------------------------------------------------ helpers
// codes^T as bf16: in int8 [R][Cc] -> out bf16 [Cc][R]   (weights only: a few MB per step)
__global__ __launch_bounds__(256) void codes_transpose_bf16_kernel(const int8_t* __restrict__ in, unsigned short* __restrict__ out,
                                                                   int R, int Cc, int f16) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < R && c < Cc) ? (float)in[(int64_t)r * Cc + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (c < Cc && r < R) {
      const _Float16 h = (_Float16)tile[tx][i];
      out[(int64_t)c * R + r] = f16 ? __builtin_bit_cast(unsigned short, h) : (unsigned short)(__float_as_uint(tile[tx][i]) >> 16);
    }
  }
}

// r[n] = sum_k vec[k] * codes[n][k]   (the post-quantiser offset's contribution to every output column)
__global__ __launch_bounds__(256) void rowdot_i8_kernel(const int8_t* __restrict__ codes, const float* __restrict__ vec,
                                                        float* __restrict__ out, int N, int K) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float acc = 0.f;
  for (int k = lane; k < K; k += 64) acc += vec[k] * (float)codes[(int64_t)n * K + k];
  acc = ofq_wave_sum(acc);
  if (lane == 0) out[n] = acc;
}

// same, 16 codes per load: a row is owned by 16 lanes (K % 16 == 0, 16-byte aligned rows)
#define RD16_RPG 4                        // rows per 16-lane group: 64 rows per workgroup, six 16-byte loads in flight per lane
__device__ __forceinline__ void rowdot_i8_v16_body(int bx, const int8_t* __restrict__ codes, const float* __restrict__ vec,
                                                   float* __restrict__ out, int N, int K) {
  const int l16 = threadIdx.x & 15;
  const int nb = bx * (16 * RD16_RPG) + (threadIdx.x >> 4);
  const int k0 = l16 * 16;
  float acc[RD16_RPG];
#pragma unroll
  for (int j = 0; j < RD16_RPG; ++j) acc[j] = 0.f;
  auto fma16 = [&](float& a, int k, const i32x4& c, const float4 (&v)[4]) {
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int word = c[w];
      a += v[w].x * (float)(signed char)(word & 0xff) + v[w].y * (float)(signed char)((word >> 8) & 0xff) +
           v[w].z * (float)(signed char)((word >> 16) & 0xff) + v[w].w * (float)(word >> 24);
    }
  };
  if (K <= 512) {
    // K = 384 (the attention prep): all rows' chunks are requested before the arithmetic, the vector chunk is read once
    // for the four rows of the group
    const bool in0 = k0 < K, in1 = k0 + 256 < K;
    i32x4 c0[RD16_RPG], c1[RD16_RPG];
#pragma unroll
    for (int j = 0; j < RD16_RPG; ++j) {
      const int n = min(nb + 16 * j, N - 1);
      const int8_t* row = codes + (int64_t)n * K;
      c0[j] = *reinterpret_cast<const i32x4*>(row + (in0 ? k0 : 0));
      c1[j] = *reinterpret_cast<const i32x4*>(row + (in1 ? k0 + 256 : 0));
    }
    if (in0) {
      float4 v[4];
#pragma unroll
      for (int w = 0; w < 4; ++w) v[w] = *reinterpret_cast<const float4*>(vec + k0 + 4 * w);
#pragma unroll
      for (int j = 0; j < RD16_RPG; ++j) fma16(acc[j], k0, c0[j], v);
    }
    if (in1) {
      float4 v[4];
#pragma unroll
      for (int w = 0; w < 4; ++w) v[w] = *reinterpret_cast<const float4*>(vec + k0 + 256 + 4 * w);
#pragma unroll
      for (int j = 0; j < RD16_RPG; ++j) fma16(acc[j], k0 + 256, c1[j], v);
    }
  } else {
#pragma unroll
    for (int j = 0; j < RD16_RPG; ++j) {
      const int n = nb + 16 * j;
      if (n >= N) continue;
      const int8_t* row = codes + (int64_t)n * K;
      for (int k = k0; k < K; k += 256) {
        float4 v[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) v[w] = *reinterpret_cast<const float4*>(vec + k + 4 * w);
        fma16(acc[j], k, *reinterpret_cast<const i32x4*>(row + k), v);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < RD16_RPG; ++j) {
    float a = acc[j];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    const int n = nb + 16 * j;
    if (n < N && l16 == 0) out[n] = a;
  }
}
__global__ __launch_bounds__(256) void rowdot_i8_v16_kernel(const int8_t* __restrict__ codes, const float* __restrict__ vec,
                                                            float* __restrict__ out, int N, int K) {
  rowdot_i8_v16_body(blockIdx.x, codes, vec, out, N, K);
}

extern "C" int ofq_codes_transpose_bf16(const int8_t* codes, void* out_bf16, int64_t rows, int64_t cols, ofq_stream_t stream) {
  if (!codes || !out_bf16 || rows <= 0 || cols <= 0) return OFQ_EINVAL;
  hipLaunchKernelGGL(codes_transpose_bf16_kernel, dim3((unsigned)ceil_div(cols, 32), (unsigned)ceil_div(rows, 32)), dim3(256),
                     0, (hipStream_t)stream, codes, (unsigned short*)out_bf16, (int)rows, (int)cols, 0);
  OFQ_LAUNCH_CHECK();
  return 0;
}
// the same, codes as FP16 (B operand of the two-plane form of the backward GEMMs)
extern "C" int ofq_codes_transpose_f16(const int8_t* codes, void* out_bf16, int64_t rows, int64_t cols, ofq_stream_t stream) {
  if (!codes || !out_bf16 || rows <= 0 || cols <= 0) return OFQ_EINVAL;
  hipLaunchKernelGGL(codes_transpose_bf16_kernel, dim3((unsigned)ceil_div(cols, 32), (unsigned)ceil_div(rows, 32)), dim3(256),
                     0, (hipStream_t)stream, codes, (unsigned short*)out_bf16, (int)rows, (int)cols, 1);
  OFQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int ofq_rowdot_i8(const int8_t* codes, const float* vec, float* out, int64_t rows, int64_t cols,
                             ofq_stream_t stream) {
  if (!codes || !vec || !out || rows <= 0 || cols <= 0) return OFQ_EINVAL;
  if ((cols & 15) == 0 && al16(codes) && al16(vec))
    hipLaunchKernelGGL(rowdot_i8_v16_kernel, dim3((unsigned)ceil_div(rows, 16 * RD16_RPG)), dim3(256), 0, (hipStream_t)stream, codes, vec,
                       out, (int)rows, (int)cols);
  else
    hipLaunchKernelGGL(rowdot_i8_kernel, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream, codes, vec, out,
                       (int)rows, (int)cols);
  OFQ_LAUNCH_CHECK();
  return 0;
}

static int qgemm_i8_linear(const int8_t* A, const int8_t* B, float* C, const float* bias, const float* col_scale,
                           float col_mult, const float* r, const float* lsq_s, int64_t S, float gscale, int64_t M, int64_t N,
                           int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int8_t* qout, int64_t ldq, const float* q_s,
                           int64_t q_S, float q_gscale, const float* q_b4, int q_lo, int q_hi, int q_gelu, int q_rowmul,
                           int64_t q_coldiv, int q_colmode, ofq_stream_t stream) {
  if (!A || !B || (!C && !qout) || !col_scale || !lsq_s || M <= 0 || N <= 0 || K <= 0 || S <= 0) return OFQ_EINVAL;
  if ((K & 15) || (lda & 15) || (ldb & 15) || !al16(A) || !al16(B) || M >= (1ll << 30) || N >= (1ll << 30)) return OFQ_EINVAL;
  if (qout && (!q_s || q_S <= 0 || (N & 15) || (ldq & 15) || ldq < N || !al16(qout) || q_lo < -128 || q_hi > 255 || q_rowmul < 1 ||
               q_coldiv < 1 || (q_rowmul > 1 && (q_coldiv % 128 || q_rowmul * q_coldiv != N)) || (q_colmode && q_S != N)))
    return OFQ_EINVAL;
  QGemmArgs a = {};
  a.A = A; a.B = B; a.C = C; a.bias = bias; a.cs = col_scale; a.r = r; a.s = lsq_s;
  a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.M = (int)M; a.N = (int)N; a.K = (int)K; a.S = (int)S;
  a.tiles_m = (int)ceil_div(M, 128); a.tiles_n = (int)ceil_div(N, 128); a.gscale = gscale; a.alpha = col_mult; a.nb1 = 1;
  a.qout = qout; a.ldq = ldq; a.qs = q_s; a.qS = (int)q_S; a.qgscale = q_gscale; a.qb4 = q_b4;
  a.qlo = (float)q_lo; a.qhi = (float)q_hi; a.qgelu = q_gelu; a.qrowmul = q_rowmul;
  a.qcoldiv = (int)(q_coldiv > (1ll << 30) ? (1ll << 30) : q_coldiv); a.qcolmode = q_colmode;
  hipLaunchKernelGGL((qgemm_i8_nt_kernel<0>), dim3((unsigned)(a.tiles_m * a.tiles_n)), dim3(256), 0, (hipStream_t)stream, a);
  OFQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int ofq_qgemm_i8_nt(const int8_t* A, const int8_t* B, float* C, const float* bias, const float* col_scale,
                               float col_mult, const float* r, const float* lsq_s, int64_t S, float gscale, int64_t M,
                               int64_t N, int64_t K,
                               int64_t lda, int64_t ldb, int64_t ldc, ofq_stream_t stream) {
  return qgemm_i8_linear(A, B, C, bias, col_scale, col_mult, r, lsq_s, S, gscale, M, N, K, lda, ldb, ldc, nullptr, 0, nullptr, 0,
                         0.f, nullptr, 0, 0, 0, 1, 1, 0, stream);
}

extern "C" int ofq_qgemm_i8_nt_q(const int8_t* A, const int8_t* B, float* C, const float* bias, const float* col_scale,
                                 float col_mult, const float* r, const float* lsq_s, int64_t S, float gscale, int64_t M,
                                 int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int8_t* qcodes, int64_t ldq,
                                 const float* q_s, int64_t q_S, float q_gscale, const float* q_b4, int q_lo, int q_hi,
                                 int q_gelu, int q_rowmul, int64_t q_coldiv, int q_colmode, ofq_stream_t stream) {
  if (!qcodes) return OFQ_EINVAL;
  return qgemm_i8_linear(A, B, C, bias, col_scale, col_mult, r, lsq_s, S, gscale, M, N, K, lda, ldb, ldc, qcodes, ldq, q_s, q_S,
                         q_gscale, q_b4, q_lo, q_hi, q_gelu, q_rowmul, q_coldiv, q_colmode, stream);
}

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
  static_assert(!(LSQ && F16), "the fused LSQ epilogue exists for the three-plane form only");
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
            const float ge = ok ? acc[i][j][e] * p.alpha : 0.f;
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
  static const bool narrow_only = getenv("OFQ_NT_NARROW") != nullptr;      // A/B switch for tools/tn_bench.py
  const int nj = N > 256 ? 3 : 2;
  // few rows (late Swin stages): 128-column tiles of the 4-wave kernel give 2-3x more workgroups, which matters more than
  // the shared split (measured: 185 vs 207 us at M=6272, N=768, K=3072; an 8-wave 128x128 variant lost to it as well)
  const bool too_few = (int64_t)a.tiles_m * ceil_div(N, 128 * nj) < 160 && (int64_t)a.tiles_m * a.tiles_n >= 192 && !col_scale && !col_bias;
  if ((nsplit == 3 || amax) && N > 128 && !narrow_only && !too_few) {      // wide tiles: the dY panel is split once per 384 (256) columns
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
  static const bool narrow_only = getenv("OFQ_PLANE_GEMM_NARROW") != nullptr;       // A/B switch
  // measured at 25 344 rows (tools/plane_gemm_bench.py, nine products, wide / narrow us): N = 384, K = 384: 80 / 108;
  // N = 384, K = 1536: 238 / 346; N = 1152, K = 384: 234 / 239; N = 1536, K = 384: 300 / 281 -- the wide tile wins where a
  // workgroup's k-loop is long or the narrow grid (594 workgroups on 512 slots) quantises badly
  static const bool wide_always = getenv("OFQ_PLANE_GEMM_WIDE") != nullptr;
  if (!narrow_only && N >= 256 && 128 * ldc < (1ll << 28) && (K & 15) == 0 && (wide_always || K >= 1024 || N <= 384)) {
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

// dX GEMM + backward of the layer's input quantiser in one kernel (QLinear: x -> move_b4 -> LSQ -> move_aft -> linear,
// qlinear.py:66-69; backward = dX_hat = dY @ W_hat, then lsq.py:571-602's autograd on (x, s, b4, baft)).
static void nt_lsq_tiles(int64_t M, int64_t N, int64_t* tm, int64_t* tn) {
  *tm = ceil_div(M, 128);
  *tn = ceil_div(N, N > 256 ? 384 : 256);
}

extern "C" size_t ofq_qgemm_bf16s_nt_lsq_ws_bytes(int64_t M, int64_t N) {
  int64_t tm, tn;
  nt_lsq_tiles(M, N, &tm, &tn);
  return (size_t)(M * tn + tm * 2 * N) * sizeof(float);
}

extern "C" int ofq_qgemm_bf16s_nt_lsq(const float* dY, const void* B_bf16, const float* k_scale, float alpha, const float* x,
                                      const float* lsq_s, int64_t S, float gscale, const float* b4, int lo, int hi, int gelu,
                                      float* dx, float* ds, float* db4, float* dbaft, int64_t M, int64_t N, int64_t K,
                                      int64_t lda, int64_t ldb, int64_t ldx, void* ws, size_t ws_bytes, ofq_stream_t stream) {
  if (!dY || !B_bf16 || !x || !lsq_s || !dx || !ws || M <= 0 || N <= 128 || K <= 0 || S <= 0) return OFQ_EINVAL;
  if ((K & 7) || (lda & 3) || (ldb & 7) || ldx < N || !al16(dY) || !al16(B_bf16) || (k_scale && !al16(k_scale)) ||
      M >= (1ll << 30) || N >= (1ll << 30))
    return OFQ_EINVAL;
  if (ws_bytes < ofq_qgemm_bf16s_nt_lsq_ws_bytes(M, N)) return OFQ_ENOWS;
  int64_t tm, tn;
  nt_lsq_tiles(M, N, &tm, &tn);
  QGemmArgs a = {};
  a.A = dY; a.B = B_bf16; a.C = dx; a.s = k_scale;
  a.lda = lda; a.ldb = ldb; a.ldc = ldx; a.M = (int)M; a.N = (int)N; a.K = (int)K;
  a.tiles_m = (int)tm; a.tiles_n = (int)tn; a.alpha = alpha; a.nb1 = 1;
  a.lx = x; a.ldlx = ldx; a.ls = lsq_s; a.lS = (int)S; a.lgscale = gscale; a.lb4 = b4; a.llo = (float)lo; a.lhi = (float)hi;
  a.lgelu = gelu; a.lrow = (float*)ws; a.lcol = (float*)ws + (size_t)M * tn;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)(tm * tn));
  if (N > 256) hipLaunchKernelGGL((qgemm_bf16s_nt_wide_kernel<3, true>), grid, dim3(512), 0, st, a);
  else hipLaunchKernelGGL((qgemm_bf16s_nt_wide_kernel<2, true>), grid, dim3(512), 0, st, a);
  OFQ_LAUNCH_CHECK();
  SumJobs jobs = {};
  int64_t maxc = 0;
  if (ds) {
    if (M % S) return OFQ_EINVAL;
    jobs.j[0] = {a.lrow, ds, S, M / S, S * tn, (int)tn, gscale, 0, 0};
    maxc = S;
  }
  if (db4) { jobs.j[1] = {a.lcol, db4, N, tm, 2 * N, 1, 1.0f, 0, 0}; if (N > maxc) maxc = N; }
  if (dbaft) { jobs.j[2] = {a.lcol + N, dbaft, N, tm, 2 * N, 1, 1.0f, 0, 0}; if (N > maxc) maxc = N; }
  if (maxc > 0) {
    strided_sum_launch(jobs, maxc, 3, st);
    OFQ_LAUNCH_CHECK();
  }
  return 0;
}

// out[r][v] = sum_k vecs[v][k] * codes[r][k]          (several offset vectors at once; V <= 32)
__global__ __launch_bounds__(256) void rowdot_i8_multi_kernel(const int8_t* __restrict__ codes, const float* __restrict__ vecs,
                                                              float* __restrict__ out, int R, int K, int V) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  for (int v = 0; v < V; ++v) {
    float acc = 0.f;
    for (int k = lane; k < K; k += 64) acc += vecs[(int64_t)v * K + k] * (float)codes[(int64_t)r * K + k];
    acc = ofq_wave_sum(acc);
    if (lane == 0) out[(int64_t)r * V + v] = acc;
  }
}

// same, 16 lanes per row and 16 codes per load; the row's codes stay in registers for all V vectors (K <= 512)
#define RDM_LDS_FLOATS 6144              // the V vectors staged in LDS when they fit (24 KB: 12 heads x 512)
__device__ __forceinline__ void rowdot_i8_multi_v16_body(int bx, const int8_t* __restrict__ codes, const float* __restrict__ vecs,
                                                         float* __restrict__ out, int R, int K, int V, float* vlds) {
  // every 16-lane row group reads all V vectors: from LDS (one cooperative copy per workgroup) instead of 8 * V float4
  // loads per lane through the texture path (which bounded this kernel: 21 us for 10 MB of codes)
  if (V * K <= RDM_LDS_FLOATS) {
    for (int i = threadIdx.x * 4; i < V * K; i += 1024) *reinterpret_cast<float4*>(vlds + i) = *reinterpret_cast<const float4*>(vecs + i);
    __syncthreads();
    vecs = vlds;
  }
  const int l16 = threadIdx.x & 15;
  const int r = bx * 16 + (threadIdx.x >> 4);
  const bool rok = r < R;
  i32x4 c[2];
  bool cok[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int k = (l16 + 16 * u) * 16;
    cok[u] = rok && k < K;
    c[u] = cok[u] ? *reinterpret_cast<const i32x4*>(codes + (int64_t)r * K + k) : i32x4{0, 0, 0, 0};
  }
  for (int v = 0; v < V; ++v) {
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (!cok[u]) continue;
      const float* vp = vecs + (int64_t)v * K + (l16 + 16 * u) * 16;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const float4 f = *reinterpret_cast<const float4*>(vp + 4 * w);
        const int word = c[u][w];
        acc += f.x * (float)(signed char)(word & 0xff) + f.y * (float)(signed char)((word >> 8) & 0xff) +
               f.z * (float)(signed char)((word >> 16) & 0xff) + f.w * (float)(word >> 24);
      }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (rok && l16 == 0) out[(int64_t)r * V + v] = acc;
  }
}
__global__ __launch_bounds__(256) void rowdot_i8_multi_v16_kernel(const int8_t* __restrict__ codes, const float* __restrict__ vecs,
                                                                  float* __restrict__ out, int R, int K, int V) {
  __shared__ __attribute__((aligned(16))) float vlds[RDM_LDS_FLOATS];
  rowdot_i8_multi_v16_body(blockIdx.x, codes, vecs, out, R, K, V, vlds);
}

// out[r][h] = sum_{c<d} x[r][h*d + c] * vec[h*d + c]      (per-head dot of an fp32 row with an offset vector)
__global__ __launch_bounds__(256) void rowdot_f32_seg_kernel(const float* __restrict__ x, const float* __restrict__ vec,
                                                             float* __restrict__ out, int R, int H, int d, int64_t ld) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  for (int h = 0; h < H; ++h) {
    float acc = 0.f;
    for (int c = lane; c < d; c += 64) acc += x[(int64_t)r * ld + h * d + c] * vec[h * d + c];
    acc = ofq_wave_sum(acc);
    if (lane == 0) out[(int64_t)r * H + h] = acc;
  }
}
// d % 4 == 0, 16-byte aligned rows: 16 lanes per row and float4 loads (one per head and 64 channels), all of a lane's
// loads requested before the arithmetic (the dword-per-lane form above ran at 2.3 TB/s)
template <int HMAX>
__global__ __launch_bounds__(256) void rowdot_f32_seg_v4_kernel(const float* __restrict__ x, const float* __restrict__ vec,
                                                                float* __restrict__ out, int R, int H, int d, int64_t ld) {
  const int l16 = threadIdx.x & 15;
  const int r = blockIdx.x * 16 + (threadIdx.x >> 4);
  const bool rok = r < R;
  const float* row = x + (int64_t)(rok ? r : 0) * ld;
  const int d4 = d >> 2;                               // <= 16 (host check)
  const bool act = l16 < d4;
  float4 xv[HMAX];
#pragma unroll
  for (int h = 0; h < HMAX; ++h)
    if (h < H) xv[h] = *reinterpret_cast<const float4*>(row + h * d + (act ? l16 : 0) * 4);
#pragma unroll
  for (int h = 0; h < HMAX; ++h) {
    if (h < H) {
      const float4 v = *reinterpret_cast<const float4*>(vec + h * d + (act ? l16 : 0) * 4);
      float acc = act ? (xv[h].x * v.x + xv[h].y * v.y) + (xv[h].z * v.z + xv[h].w * v.w) : 0.f;
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
      if (rok && l16 == 0) out[(int64_t)r * H + h] = acc;
    }
  }
}

// batched int8 transpose with zero padding: in [B][R][Cc] -> out [B][Cc][Rp]  (V codes for the P*V product)
__device__ __forceinline__ void codes_transpose_i8_body(int bx, int by, int bz, int8_t (*tile)[36], const int8_t* __restrict__ in,
                                                        int8_t* __restrict__ out, int R, int Cc, int Rp) {
  // 32 x 32 bytes per tile: one dword load per thread (four consecutive columns of one row), its bytes scattered into the
  // transposed LDS tile, one dword store per thread (four consecutive rows of one column).  Cc % 4 == 0 and Rp % 4 == 0
  // (host check), so a dword is all inside or all outside the matrices.
  const int r0 = by * 32, c0 = bx * 32;
  const int8_t* ib = in + (int64_t)bz * R * Cc;
  int8_t* ob = out + (int64_t)bz * Cc * Rp;
  {
    const int r = r0 + (threadIdx.x >> 3), cq = (threadIdx.x & 7) * 4, c = c0 + cq;
    int word = 0;
    if (r < R && c < Cc) word = *reinterpret_cast<const int*>(ib + (int64_t)r * Cc + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) tile[cq + j][threadIdx.x >> 3] = (int8_t)((word >> (8 * j)) & 0xff);
  }
  __syncthreads();
  {
    const int cc = threadIdx.x >> 3, rq = (threadIdx.x & 7) * 4;
    const int c = c0 + cc, r = r0 + rq;
    if (c < Cc && r < Rp) *reinterpret_cast<int*>(ob + (int64_t)c * Rp + r) = *reinterpret_cast<const int*>(&tile[cc][rq]);
  }
}
// any size / alignment, one byte per access
__global__ __launch_bounds__(256) void codes_transpose_i8_bytes_kernel(const int8_t* __restrict__ in, int8_t* __restrict__ out,
                                                                       int R, int Cc, int Rp) {
  __shared__ int8_t tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int8_t* ib = in + (int64_t)blockIdx.z * R * Cc;
  int8_t* ob = out + (int64_t)blockIdx.z * Cc * Rp;
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < R && c < Cc) ? ib[(int64_t)r * Cc + c] : (int8_t)0;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (c < Cc && r < Rp) ob[(int64_t)c * Rp + r] = tile[tx][i];
  }
}
__global__ __launch_bounds__(256) void codes_transpose_i8_kernel(const int8_t* __restrict__ in, int8_t* __restrict__ out, int R,
                                                                 int Cc, int Rp) {
  __shared__ __attribute__((aligned(16))) int8_t tile[32][36];
  codes_transpose_i8_body(blockIdx.x, blockIdx.y, blockIdx.z, tile, in, out, R, Cc, Rp);
}

// The three operand-preparation jobs of the QKR attention core in ONE launch: u[b,n,h] = x codes . baq[h] (row dots with H
// vectors), tq[b,m,h] = qkx codes . bax, and the per-image transpose of the v codes to [C][Np] (B operand of the int8 P.V
// GEMM).  Each is a small, occupancy-bound kernel (~20 us for 10-60 MB); as ranges of one grid they run side by side.
struct AttnPrepArgs {
  const int8_t* xcodes; const float* baq; float* u;          // job 0: [R0 = B*N][C] . [H][C] -> [R0][H]
  const int8_t* qcodes; const float* bax; float* tq;         // job 1: [R1 = B*N*H][C] . [C] -> [R1]
  const int8_t* vcodes; int8_t* vT;                          // job 2: [B][N][C] -> [B][C][Np]
  float* z;                                                  // job 3 (one block, optional): z[h] = baq[h][:] . bax
  int R0, R1, C, H, N, Np, nb0, nb1, tx2, ty2, nb2;          // nb0 / nb1 / nb2: blocks of jobs 0 / 1 / 2; job 2 grid: tx2 x ty2 x B
};
__global__ __launch_bounds__(256) void qattn_prep_kernel(AttnPrepArgs a) {
  __shared__ __attribute__((aligned(16))) int8_t tile[32][36];
  __shared__ __attribute__((aligned(16))) float vlds[RDM_LDS_FLOATS];
  int b = blockIdx.x;
  if (b < a.nb0) { rowdot_i8_multi_v16_body(b, a.xcodes, a.baq, a.u, a.R0, a.C, a.H, vlds); return; }
  b -= a.nb0;
  if (b < a.nb1) { rowdot_i8_v16_body(b, a.qcodes, a.bax, a.tq, a.R1, a.C); return; }
  b -= a.nb1;
  if (b >= a.nb2) {          // the offset-offset term of the scores (attention.py:206-210: move_qkx_aft . move_aft of x), one wave per head
    const int lane = threadIdx.x & 63;
    for (int h = threadIdx.x >> 6; h < a.H; h += 4) {
      float acc = 0.f;
      for (int c = lane; c < a.C; c += 64) acc += a.baq[h * a.C + c] * a.bax[c];
      acc = ofq_wave_sum(acc);
      if (lane == 0) a.z[h] = acc;
    }
    return;
  }
  const int bx = b % a.tx2, by = (b / a.tx2) % a.ty2, bz = b / (a.tx2 * a.ty2);
  codes_transpose_i8_body(bx, by, bz, tile, a.vcodes, a.vT, a.N, a.C, a.Np);
}
extern "C" int ofq_qattn_prep(const int8_t* xcodes, const float* baq, float* u, const int8_t* qcodes, const float* bax, float* tq,
                              const int8_t* vcodes, int8_t* vT, float* z, int64_t B, int64_t H, int64_t N, int64_t C, int64_t Np,
                              ofq_stream_t stream) {
  if (!xcodes || !baq || !u || !qcodes || !bax || !tq || !vcodes || !vT || B <= 0 || H <= 0 || N <= 0 || Np < N) return OFQ_EINVAL;
  if ((C & 15) || (Np & 3) || C > 512 || !al16(xcodes) || !al16(qcodes) || !al16(baq) || !al16(bax) || !al16(vcodes) || !al16(vT) ||
      B * N * H >= (1ll << 31))
    return OFQ_EINVAL;
  AttnPrepArgs a = {};
  a.xcodes = xcodes; a.baq = baq; a.u = u; a.qcodes = qcodes; a.bax = bax; a.tq = tq; a.vcodes = vcodes; a.vT = vT;
  a.R0 = (int)(B * N); a.R1 = (int)(B * N * H); a.C = (int)C; a.H = (int)H; a.N = (int)N; a.Np = (int)Np;
  a.nb0 = (int)ceil_div(B * N, 16); a.nb1 = (int)ceil_div(B * N * H, 16 * RD16_RPG);
  a.tx2 = (int)ceil_div(C, 32); a.ty2 = (int)ceil_div(Np, 32);
  a.z = z;
  if ((int64_t)a.tx2 * a.ty2 * B >= (1ll << 31)) return OFQ_EINVAL;
  a.nb2 = (int)((int64_t)a.tx2 * a.ty2 * B);
  const int64_t total = (int64_t)a.nb0 + a.nb1 + a.nb2 + (z ? 1 : 0);
  if (total >= (1ll << 31)) return OFQ_EINVAL;
  hipLaunchKernelGGL(qattn_prep_kernel, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, a);
  OFQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int ofq_rowdot_i8_multi(const int8_t* codes, const float* vecs, float* out, int64_t rows, int64_t cols, int nvec,
                                   ofq_stream_t stream) {
  if (!codes || !vecs || !out || rows <= 0 || cols <= 0 || nvec <= 0) return OFQ_EINVAL;
  if ((cols & 15) == 0 && cols <= 512 && al16(codes) && al16(vecs))
    hipLaunchKernelGGL(rowdot_i8_multi_v16_kernel, dim3((unsigned)ceil_div(rows, 16)), dim3(256), 0, (hipStream_t)stream, codes,
                       vecs, out, (int)rows, (int)cols, nvec);
  else
    hipLaunchKernelGGL(rowdot_i8_multi_kernel, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream, codes, vecs,
                       out, (int)rows, (int)cols, nvec);
  OFQ_LAUNCH_CHECK();
  return 0;
}
extern "C" int ofq_rowdot_f32_seg(const float* x, const float* vec, float* out, int64_t rows, int heads, int head_dim, int64_t ld,
                                  ofq_stream_t stream) {
  if (!x || !vec || !out || rows <= 0 || heads <= 0 || head_dim <= 0) return OFQ_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if ((head_dim & 3) == 0 && head_dim <= 64 && heads <= 24 && (ld & 3) == 0 && al16(x) && al16(vec)) {
    const dim3 grid((unsigned)ceil_div(rows, 16)), block(256);
    if (heads <= 6) hipLaunchKernelGGL(rowdot_f32_seg_v4_kernel<6>, grid, block, 0, st, x, vec, out, (int)rows, heads, head_dim, ld);
    else if (heads <= 12) hipLaunchKernelGGL(rowdot_f32_seg_v4_kernel<12>, grid, block, 0, st, x, vec, out, (int)rows, heads, head_dim, ld);
    else hipLaunchKernelGGL(rowdot_f32_seg_v4_kernel<24>, grid, block, 0, st, x, vec, out, (int)rows, heads, head_dim, ld);
  } else {
    hipLaunchKernelGGL(rowdot_f32_seg_kernel, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, st, x, vec, out, (int)rows, heads,
                       head_dim, ld);
  }
  OFQ_LAUNCH_CHECK();
  return 0;
}extern "C" int ofq_codes_transpose_i8(const int8_t* in, int8_t* out, int64_t batches, int64_t rows, int64_t cols, int64_t rows_padded,
                                      ofq_stream_t stream) {
  if (!in || !out || batches <= 0 || rows <= 0 || cols <= 0 || rows_padded < rows) return OFQ_EINVAL;
  const dim3 grid((unsigned)ceil_div(cols, 32), (unsigned)ceil_div(rows_padded, 32), (unsigned)batches);
  if ((cols & 3) == 0 && (rows_padded & 3) == 0 && (((uintptr_t)in | (uintptr_t)out) & 3) == 0)
    hipLaunchKernelGGL(codes_transpose_i8_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, out, (int)rows, (int)cols,
                       (int)rows_padded);
  else
    hipLaunchKernelGGL(codes_transpose_i8_bytes_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, out, (int)rows, (int)cols,
                       (int)rows_padded);
  OFQ_LAUNCH_CHECK();
  return 0;
}

// Batched attention GEMMs on int8 codes: 64 x 64 tiles when the whole (tokens x tokens) / (tokens x head_dim) matrix of a
// batch entry fits in one (the 49-token Swin windows: a 128 x 128 tile would be 85 % padding), 128 x 128 tiles otherwise.
template <int EPI>
static void i8_attn_launch(QGemmArgs& a, int64_t M, int64_t N, int64_t batches, hipStream_t st) {
  static const bool no_small = getenv("OFQ_NO_SMALL_TILES") != nullptr;          // A/B switch (tools/)
  static const bool no_tall = getenv("OFQ_NO_TALL_TILES") != nullptr;            // A/B switch
  if (M <= 64 && N <= 64 && !no_small) {
    a.tiles_m = a.tiles_n = 1;
    hipLaunchKernelGGL((qgemm_i8_nt_kernel<EPI, 1>), dim3(1u, (unsigned)batches), dim3(256), 0, st, a);
  } else if (EPI == 2 && N <= 64 && M > 128 && !no_tall) {     // P.V of a long sequence: 256 x 64 tiles, four waves stacked
    a.tiles_m = (int)ceil_div(M, 256); a.tiles_n = 1;
    hipLaunchKernelGGL((qgemm_i8_nt_kernel<EPI, 3>), dim3((unsigned)a.tiles_m, (unsigned)batches), dim3(256), 0, st, a);
  } else {
    a.tiles_m = (int)ceil_div(M, 128); a.tiles_n = (int)ceil_div(N, 128);
    hipLaunchKernelGGL((qgemm_i8_nt_kernel<EPI, 2>), dim3((unsigned)(a.tiles_m * a.tiles_n), (unsigned)batches), dim3(256), 0, st, a);
  }
}

// ---- attention products on the codes -------------------------------------------------------------------------
// scores: S[b,h,n,m] = ax[n]*(aq[m,h]*(qx[b,n,:].qq[b,m,h,:]) + u[b,n,h]) + aq[m,h]*tq[b,m,h] + z[h]
extern "C" int ofq_qattn_scores_i8(const int8_t* xcodes, const int8_t* qcodes, float* S, const float* sx, float gscale_x,
                                   const float* sq, float gscale_q, const float* u, const float* tq, const float* z, int64_t B,
                                   int64_t H, int64_t N, int64_t C, int64_t ldS, ofq_stream_t stream) {
  if (!xcodes || !qcodes || !S || !sx || !sq || !u || !tq || !z || B <= 0 || H <= 0 || N <= 0 || (C & 15) || ldS < N)
    return OFQ_EINVAL;
  QGemmArgs a = {};
  a.A = xcodes; a.B = qcodes; a.C = S; a.s = sx; a.s2 = sq; a.u = u; a.tq = tq; a.z = z;
  a.lda = C; a.ldb = H * C; a.ldc = ldS;
  a.sA0 = N * C; a.sA1 = 0; a.sB0 = N * H * C; a.sB1 = C; a.sC0 = H * N * ldS; a.sC1 = N * ldS;
  a.M = (int)N; a.N = (int)N; a.K = (int)C; a.S = (int)N; a.nb1 = (int)H; a.s2s0 = (int)H; a.s2s1 = 1;
  a.gscale = gscale_x; a.gscale2 = gscale_q;
  i8_attn_launch<1>(a, N, N, B * H, (hipStream_t)stream);
  OFQ_LAUNCH_CHECK();
  return 0;
}
// Plain attention (attention.py:92-96) on the codes: S[b,h,n,m] = q_hat[b,n,hd:hd+d] . k_hat[b,m,hd:hd+d] with
// q_hat = aq[n]*qq + bq[c], k_hat = ak[m]*qk + bk[c] (per-token steps, per-channel offsets):
//   S = aq[n]*(ak[m]*I + u[b,n,h]) + ak[m]*tq[b,m,h] + z[h],  u = qq . bk|head, tq = bq|head . qk, z = bq|head . bk|head
extern "C" int ofq_qattn_scores_plain_i8(const int8_t* qcodes, const int8_t* kcodes, float* S, const float* sq, float gscale_q,
                                         const float* sk, float gscale_k, const float* u, const float* tq, const float* z,
                                         int64_t B, int64_t H, int64_t N, int64_t d, int64_t ldS, ofq_stream_t stream) {
  if (!qcodes || !kcodes || !S || !sq || !sk || !u || !tq || !z || B <= 0 || H <= 0 || N <= 0 || (d & 15) || ldS < N)
    return OFQ_EINVAL;
  QGemmArgs a = {};
  const int64_t C = H * d;
  a.A = qcodes; a.B = kcodes; a.C = S; a.s = sq; a.s2 = sk; a.u = u; a.tq = tq; a.z = z;
  a.lda = C; a.ldb = C; a.ldc = ldS;
  a.sA0 = N * C; a.sA1 = d; a.sB0 = N * C; a.sB1 = d; a.sC0 = H * N * ldS; a.sC1 = N * ldS;
  a.M = (int)N; a.N = (int)N; a.K = (int)d; a.S = (int)N; a.nb1 = (int)H; a.s2s0 = 1; a.s2s1 = 0;
  a.gscale = gscale_q; a.gscale2 = gscale_k;
  i8_attn_launch<1>(a, N, N, B * H, (hipStream_t)stream);
  OFQ_LAUNCH_CHECK();
  return 0;
}
// P*V: O[b,n,h*d+c] = ap[n]*(av[h*d+c]*(qp[b,h,n,:].qvT[b,h*d+c,:]) + bav[h*d+c]*rp[b,h,n])
extern "C" int ofq_qattn_pv_i8(const int8_t* pcodes, const int8_t* vcodesT, float* O, const float* sp, float gscale_p,
                               const float* sv, float gscale_v, const float* bav, const float* rp, int64_t B, int64_t H,
                               int64_t N, int64_t d, int64_t Np, ofq_stream_t stream) {
  if (!pcodes || !vcodesT || !O || !sp || !sv || !rp || B <= 0 || H <= 0 || N <= 0 || d <= 0 || (Np & 15) || Np < N) return OFQ_EINVAL;
  QGemmArgs a = {};
  const int64_t C = H * d;
  a.A = pcodes; a.B = vcodesT; a.C = O; a.s = sp; a.s2 = sv; a.z = bav; a.rp = rp;
  a.lda = Np; a.ldb = Np; a.ldc = C;
  a.sA0 = H * N * Np; a.sA1 = N * Np; a.sB0 = C * Np; a.sB1 = d * Np; a.sC0 = N * C; a.sC1 = d;
  a.M = (int)N; a.N = (int)d; a.K = (int)Np; a.S = (int)N; a.nb1 = (int)H;
  a.gscale = gscale_p; a.gscale2 = gscale_v;
  i8_attn_launch<2>(a, N, d, B * H, (hipStream_t)stream);
  OFQ_LAUNCH_CHECK();
  return 0;
}
// ---- dP on wave tiles ---------------------------------------------------------------------------------------------------
// The contraction of dP is the head dimension (d = 32 / 64): two or four MFMA steps.  For Swin's 49-token windows the
// 128 x 128 workgroup tile is 85 % padding and the launch one load round trip + a barrier-paced k-step per workgroup.
// Here one WAVE owns 64 rows of one (window, head): its dO rows go from global memory straight into MFMA-fragment layout (each lane reads eight
// consecutive floats of its row) and are scaled and split into the three bf16 planes once, in registers; then the wave
// walks the key tokens 32 at a time, reading the eight consecutive int8 codes of a fragment lane directly from the code
// matrix (no LDS, no barrier at all) and storing 32-token row segments.  KS = d / 16.
template <int KS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void qgemm_bf16s_nt_win_kernel(QGemmArgs p, int mchunks, int ntasks) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int bx, by_unused;
  xcd_remap_grid(bx, by_unused);
  const int task = bx * 4 + wid;
  if (task >= ntasks) return;                          // wave-uniform
  const int pair = task / mchunks, mc = task - pair * mchunks;
  const int b0 = pair / p.nb1, b1 = pair - b0 * p.nb1;
  const int m0 = mc * 64;
  const int l31 = lane & 31, lh = lane >> 5;
  const float* Ab = reinterpret_cast<const float*>(p.A) + b0 * p.sA0 + b1 * p.sA1;
  const int8_t* Bb = reinterpret_cast<const int8_t*>(p.B) + b0 * p.sB0 + b1 * p.sB1;
  float* Cb = p.C + b0 * p.sC0 + b1 * p.sC1;
  const float* ksp = p.s ? p.s + b1 * p.sK1 : nullptr;

  unsigned av[3][KS][2][4];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int kb = 16 * ks + 8 * lh;                   // K % 16 == 0 (host check): a fragment lane is all in
    float sc[8];
    if (ksp) {
      const float4 s0 = *reinterpret_cast<const float4*>(ksp + kb), s1 = *reinterpret_cast<const float4*>(ksp + kb + 4);
      sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w; sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) sc[e] = 1.f;
    }
    if (p.gscale2 > 0.f) {
#pragma unroll
      for (int e = 0; e < 8; ++e) sc[e] = ofq_lsq_eff_scale(sc[e], p.gscale2);      // raw LSQ step -> effective value
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = m0 + 32 * i + l31;
      const float z = row < p.M ? 1.f : 0.f;
      const float* ar = Ab + (int64_t)min(row, p.M - 1) * p.lda + kb;
      const float4 a0 = *reinterpret_cast<const float4*>(ar), a1 = *reinterpret_cast<const float4*>(ar + 4);
      const float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        const f32x2v kk = {sc[e] * z, sc[e + 1] * z};
        const f32x2v aa = {v[e], v[e + 1]};
        unsigned pl[3];
        split_pair_bf16<3>(aa * kk, pl);
#pragma unroll
        for (int q = 0; q < 3; ++q) av[q][ks][i][e >> 1] = pl[q];
      }
    }
  }
  // per-row addend of this lane's 2 x 16 accumulator rows
  float uu[2][16];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = m0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * lh;
      uu[i][e] = p.u ? p.u[((int64_t)b0 * p.M + min(row, p.M - 1)) * p.nb1 + b1] : 0.f;
    }

  auto bload = [&](int n0, u32x2v (&rb)[KS]) {
    const int col = min(n0 + l31, p.N - 1);
    const int8_t* br = Bb + (int64_t)col * p.ldb + 8 * lh;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) rb[ks] = *reinterpret_cast<const u32x2v*>(br + 16 * ks);
  };
  u32x2v rb[KS];
  bload(0, rb);
  for (int n0 = 0; n0 < p.N; n0 += 32) {
    const bool okc = n0 + l31 < p.N;
    const unsigned msk = okc ? 0xffffffffu : 0u;
    bf16x8 bv[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int w0 = (int)(rb[ks][0] & msk), w1 = (int)(rb[ks][1] & msk);
      typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
      u32x4v w;
      w[0] = i8x2_to_bf16x2((int)(signed char)(w0 & 0xff), (int)(signed char)((w0 >> 8) & 0xff));
      w[1] = i8x2_to_bf16x2((int)(signed char)((w0 >> 16) & 0xff), (int)(signed char)((w0 >> 24) & 0xff));
      w[2] = i8x2_to_bf16x2((int)(signed char)(w1 & 0xff), (int)(signed char)((w1 >> 8) & 0xff));
      w[3] = i8x2_to_bf16x2((int)(signed char)((w1 >> 16) & 0xff), (int)(signed char)((w1 >> 24) & 0xff));
      bv[ks] = __builtin_bit_cast(bf16x8, w);
    }
    if (n0 + 32 < p.N) bload(n0 + 32, rb);               // the next 32 key tokens' codes fly behind the MFMAs
    f32x16q acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
          const u32x4v au = {av[q][ks][i][0], av[q][ks][i][1], av[q][ks][i][2], av[q][ks][i][3]};
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, au), bv[ks], acc[i], 0, 0, 0);
        }
    if (okc) {
      const int col = n0 + l31;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = m0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * lh;
          if (row < p.M) Cb[(int64_t)row * p.ldc + col] = acc[i][e] * p.alpha + uu[i][e];
        }
    }
  }
}

// dP[b,h,n,m] = sum_c (dO[b,n,h*d+c]*av[h*d+c]) * qv[b,m,h*d+c] + w[b,n,h]
extern "C" int ofq_qattn_dp_bf16s(const float* dO, const int8_t* vcodes, float* dP, const float* sv, float gscale_v,
                                  const float* w, int64_t B, int64_t H, int64_t N, int64_t d, int64_t ldP, ofq_stream_t stream) {
  if (!dO || !vcodes || !dP || !sv || B <= 0 || H <= 0 || N <= 0 || (d & 7) || ldP < N) return OFQ_EINVAL;
  QGemmArgs a = {};
  const int64_t C = H * d;
  a.A = dO; a.B = vcodes; a.C = dP; a.s = sv; a.gscale2 = gscale_v; a.u = w; a.b_is_i8 = 1;
  a.lda = C; a.ldb = C; a.ldc = ldP;
  a.sA0 = N * C; a.sA1 = d; a.sB0 = N * C; a.sB1 = d; a.sC0 = H * N * ldP; a.sC1 = N * ldP; a.sK1 = d;
  a.M = (int)N; a.N = (int)N; a.K = (int)d; a.nb1 = (int)H; a.alpha = 1.f;
  a.tiles_m = (int)ceil_div(N, 128); a.tiles_n = (int)ceil_div(N, 128);
  static const bool no_win = getenv("OFQ_NO_WIN_NT") != nullptr;
  const int mchunks = (int)ceil_div(N, 64);
  const int64_t ntasks = B * H * mchunks;
  // window-sized token counts only: at N = 198 (four row chunks, seven key blocks per wave) the workgroup tile wins
  // (measured: 27.73 vs 27.87 ms/step for DeiT-S; Swin-T 50.43 -> 49.86 ms with the wave tile)
  if (!no_win && N <= 64 && (d == 64 || d == 32 || d == 16 || d == 48) && al16(dO) && al16(sv) && (C & 3) == 0 &&
      ntasks < (1ll << 31)) {
    const dim3 grid((unsigned)ceil_div(ntasks, 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
    switch (d / 16) {
      case 1: hipLaunchKernelGGL((qgemm_bf16s_nt_win_kernel<1>), grid, block, 0, st, a, mchunks, (int)ntasks); break;
      case 2: hipLaunchKernelGGL((qgemm_bf16s_nt_win_kernel<2>), grid, block, 0, st, a, mchunks, (int)ntasks); break;
      case 3: hipLaunchKernelGGL((qgemm_bf16s_nt_win_kernel<3>), grid, block, 0, st, a, mchunks, (int)ntasks); break;
      default: hipLaunchKernelGGL((qgemm_bf16s_nt_win_kernel<4>), grid, block, 0, st, a, mchunks, (int)ntasks); break;
    }
    OFQ_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL((qgemm_bf16s_nt_kernel<3, true>), dim3((unsigned)(a.tiles_m * a.tiles_n), (unsigned)(B * H)), dim3(256), 0,
                     (hipStream_t)stream, a);
  OFQ_LAUNCH_CHECK();
  return 0;
}
// dV[b,m,h*d+c] = sum_n qp[b,h,n,m] * (ap[n] * dO[b,n,h*d+c])
extern "C" int ofq_qattn_dv_bf16s(const float* dO, const int8_t* pcodes, float* dV, const float* sp, float gscale_p, int64_t B,
                                  int64_t H, int64_t N, int64_t d, int64_t Np, ofq_stream_t stream) {
  if (!dO || !pcodes || !dV || !sp || B <= 0 || H <= 0 || N <= 0 || (d & 3) || (Np & 15) || Np < N) return OFQ_EINVAL;
  QTnArgs a = {};
  const int64_t C = H * d;
  a.A = dO; a.B = pcodes; a.C = dV; a.s = sp; a.lda = C; a.ldb = Np; a.ldc = C;
  a.sA0 = N * C; a.sA1 = d; a.sB0 = H * N * Np; a.sB1 = N * Np; a.sC0 = N * C; a.sC1 = d;
  a.M = (int)d; a.N = (int)Np; a.Ktok = (int)N; a.S = (int)N; a.split = 1; a.nb1 = (int)H;
  a.Mstore = (int)d; a.Nstore = (int)N; a.trans_out = 1; a.gscale = gscale_p;
  a.tiles_m = (int)ceil_div(d, 128); a.tiles_n = (int)ceil_div(Np, 128);
  if (tn_win_launch(a, B * H, (hipStream_t)stream)) {     // Swin windows: one wave per (window, head)
    OFQ_LAUNCH_CHECK();
    return 0;
  }
  static const bool no_w4 = getenv("OFQ_DV_NO_W4") != nullptr;       // A/B switch
  if (d <= 64 && Np > 128 && !no_w4) {       // one head's channels x all keys: 64 x 256 tiles, four waves side by side
    a.tiles_m = 1; a.tiles_n = (int)ceil_div(Np, 256);
    hipLaunchKernelGGL(qgemm_bf16s_tn_kernel<true>, dim3((unsigned)a.tiles_n, (unsigned)(B * H)), dim3(256), 0, (hipStream_t)stream, a);
  } else {
    hipLaunchKernelGGL(qgemm_bf16s_tn_kernel<false>, dim3((unsigned)(a.tiles_m * a.tiles_n), (unsigned)(B * H)), dim3(256), 0,
                       (hipStream_t)stream, a);
  }
  OFQ_LAUNCH_CHECK();
  return 0;
}
// dqkx[b,m,h,c] = sum_n dS[b,h,n,m] * (ax[n]*qx[b,n,c] + bax[c])
extern "C" int ofq_qattn_dqkx_bf16s(const float* dS, const int8_t* xcodes, float* dqkx, const float* sx, float gscale_x,
                                    const float* bax, int64_t B, int64_t H, int64_t N, int64_t C, int64_t ldS, const void* amax,
                                    ofq_stream_t stream) {
  if (!dS || !xcodes || !dqkx || !sx || B <= 0 || H <= 0 || N <= 0 || (C & 15) || (ldS & 3) || ldS < N) return OFQ_EINVAL;
  QTnArgs a = {};
  a.amax = (const unsigned*)amax;      // two-plane fp16 form of the wide kernels (C % 384 == 0); elsewhere three bf16 planes
  a.A = dS; a.B = xcodes; a.C = dqkx; a.s = sx; a.baft = bax; a.lda = ldS; a.ldb = C; a.ldc = H * C;
  a.sA0 = H * N * ldS; a.sA1 = N * ldS; a.sB0 = N * C; a.sB1 = 0; a.sC0 = N * H * C; a.sC1 = C;
  a.M = (int)ldS; a.N = (int)C; a.Ktok = (int)N; a.S = (int)N; a.split = 1; a.nb1 = (int)H;
  a.Mstore = (int)N; a.Nstore = (int)C; a.trans_out = 0; a.gscale = gscale_x;
  a.tiles_m = (int)ceil_div(ldS, 128); a.tiles_n = (int)ceil_div(C, 128);
  if (tn_win_launch(a, B * H, (hipStream_t)stream)) {
    OFQ_LAUNCH_CHECK();
    return 0;
  }
  static const bool narrow_only = getenv("OFQ_TN_NARROW") != nullptr;
  if (C % 384 == 0 && N >= QTN_BK && !narrow_only && N * ldS < (1ll << 31) && N * C < (1ll << 31)) {
    a.tiles_n = (int)(C / 384);       // one split of a dS panel feeds all 384 columns
    static const bool no_stream = getenv("OFQ_TN_NO_STREAM") != nullptr;      // A/B switch
    static const bool no_stack = getenv("OFQ_TN_NO_STACK") != nullptr;        // A/B switch
    int64_t T = (int64_t)a.tiles_m * a.tiles_n * H;                           // tiles per image
    // Stacked form: the H heads share the B operand (x_hat of the image), so their ldS-row outputs are tiled as one
    // (H ldS)-row matrix: 10 tiles of 128 rows per DeiT-S image instead of 6 x 2 whose second one holds 69 rows.
    bool stacked = false;
    if (!no_stream && !no_stack && a.tiles_n == 1 && H > 1 && ceil_div(H * ldS, 128) < (int64_t)a.tiles_m * H && H * ldS < 2048 &&
        H * N * ldS * 4 < (1ll << 31) && (N * H + H) * C < (1ll << 29)) {
      const unsigned magic = (unsigned)(((1u << 20) + ldS - 1) / ldS);
      bool ok = true;
      for (int64_t r = 0; r < ceil_div(H * ldS, 128) * 128 + 32 && ok; ++r) ok = (int64_t)(((unsigned)r * magic) >> 20) == r / ldS;
      if (ok) {
        stacked = true;
        a.stk_mp = (int)ldS; a.stk_h = (int)H; a.stk_valid = (int)N; a.stk_magic = magic;
        a.M = (int)(H * ldS); a.Mstore = a.M; a.nb1 = 1; a.tiles_m = (int)ceil_div(H * ldS, 128);
        T = a.tiles_m;
      }
    }
    if (!no_stream && T >= 2) {
      // persistent workgroups: each walks `tpw` tiles of one image (about one workgroup per CU in total)
      int64_t tpw = (B * T) / 256;
      if (const char* e = getenv("OFQ_TN_STREAM_TPW")) tpw = atoi(e);       // test hook: tiles per workgroup
      tpw = tpw < 1 ? 1 : (tpw > T ? T : tpw);
      while (T % tpw) --tpw;                                                  // equal chunks
      int stagger = 0;                                                      // measured: 121 us without, 149 us with a 6-phase shift
      if (const char* e = getenv("OFQ_TN_STREAM_STAGGER")) stagger = atoi(e);  // A/B switch
      const dim3 grid((unsigned)(T / tpw), (unsigned)B);
      hipStream_t st = (hipStream_t)stream;
      if (stacked) {
        if (amax) hipLaunchKernelGGL((qgemm_bf16s_tn_wide_stream_kernel<3, true, true>), grid, dim3(512), 0, st, a, (int)tpw, stagger);
        else hipLaunchKernelGGL((qgemm_bf16s_tn_wide_stream_kernel<3, true, false>), grid, dim3(512), 0, st, a, (int)tpw, stagger);
      } else {
        if (amax) hipLaunchKernelGGL((qgemm_bf16s_tn_wide_stream_kernel<3, false, true>), grid, dim3(512), 0, st, a, (int)tpw, stagger);
        else hipLaunchKernelGGL((qgemm_bf16s_tn_wide_stream_kernel<3, false, false>), grid, dim3(512), 0, st, a, (int)tpw, stagger);
      }
      OFQ_LAUNCH_CHECK();
      return 0;
    }
    if (amax)
      hipLaunchKernelGGL((qgemm_bf16s_tn_wide_kernel<3, true>), dim3((unsigned)(a.tiles_m * a.tiles_n), (unsigned)(B * H)), dim3(512), 0,
                         (hipStream_t)stream, a);
    else
    hipLaunchKernelGGL(qgemm_bf16s_tn_wide_kernel<3>, dim3((unsigned)(a.tiles_m * a.tiles_n), (unsigned)(B * H)), dim3(512), 0,
                       (hipStream_t)stream, a);
    OFQ_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(qgemm_bf16s_tn_kernel<false>, dim3((unsigned)(a.tiles_m * a.tiles_n), (unsigned)(B * H)), dim3(256), 0,
                     (hipStream_t)stream, a);
  OFQ_LAUNCH_CHECK();
  return 0;
}
// Plain attention, backward of the scores (autograd of attention.py:96):
//   dq_hat[b,n,hd+c] = sum_m (dS[b,h,n,m] * ak[m]) * qk[b,m,hd+c]                     (+ bk[c] * rowsum(dS): rows of a
//                      softmax backward sum to zero, see functional.KEEP_ZERO_ROWSUM_TERM)
//   dk_hat[b,m,hd+c] = sum_n dS[b,h,n,m] * (aq[n] * qq[b,n,hd+c] + bq[hd+c])
extern "C" int ofq_qattn_dq_plain_bf16s(const float* dS, const int8_t* kcodes, float* dq, const float* sk, float gscale_k,
                                        int64_t B, int64_t H, int64_t N, int64_t d, int64_t ldS, ofq_stream_t stream) {
  if (!dS || !kcodes || !dq || !sk || B <= 0 || H <= 0 || N <= 0 || (d & 15) || (ldS & 7) || ldS < N) return OFQ_EINVAL;
  QNnArgs a = {};
  const int64_t C = H * d;
  a.A = dS; a.B = kcodes; a.C = dq; a.s = sk; a.lda = ldS; a.ldb = C; a.ldc = C;
  a.sA0 = H * N * ldS; a.sA1 = N * ldS; a.sB0 = N * C; a.sB1 = d; a.sC0 = N * C; a.sC1 = d; a.nb1 = (int)H;
  a.M = (int)N; a.N = (int)d; a.K = (int)N; a.nkb = 1; a.ks_stride = 1; a.accumulate = 0; a.gscale = gscale_k;
  a.tiles_m = (int)ceil_div(N, 128); a.tiles_n = (int)ceil_div(d, 128);
  hipLaunchKernelGGL(qgemm_bf16s_nn_kernel, dim3((unsigned)(a.tiles_m * a.tiles_n), (unsigned)(B * H)), dim3(256), 0,
                     (hipStream_t)stream, a);
  OFQ_LAUNCH_CHECK();
  return 0;
}
extern "C" int ofq_qattn_dk_plain_bf16s(const float* dS, const int8_t* qcodes, float* dk, const float* sq, float gscale_q,
                                        const float* bq, int64_t B, int64_t H, int64_t N, int64_t d, int64_t ldS,
                                        ofq_stream_t stream) {
  if (!dS || !qcodes || !dk || !sq || B <= 0 || H <= 0 || N <= 0 || (d & 15) || (ldS & 3) || ldS < N) return OFQ_EINVAL;
  QTnArgs a = {};
  const int64_t C = H * d;
  a.A = dS; a.B = qcodes; a.C = dk; a.s = sq; a.baft = bq; a.sBf1 = d; a.lda = ldS; a.ldb = C; a.ldc = C;
  a.sA0 = H * N * ldS; a.sA1 = N * ldS; a.sB0 = N * C; a.sB1 = d; a.sC0 = N * C; a.sC1 = d;
  a.M = (int)ldS; a.N = (int)d; a.Ktok = (int)N; a.S = (int)N; a.split = 1; a.nb1 = (int)H;
  a.Mstore = (int)N; a.Nstore = (int)d; a.trans_out = 0; a.gscale = gscale_q;
  a.tiles_m = (int)ceil_div(ldS, 128); a.tiles_n = (int)ceil_div(d, 128);
  if (!tn_win_launch(a, B * H, (hipStream_t)stream))
    hipLaunchKernelGGL(qgemm_bf16s_tn_kernel<false>, dim3((unsigned)(a.tiles_m * a.tiles_n), (unsigned)(B * H)), dim3(256), 0,
                       (hipStream_t)stream, a);
  OFQ_LAUNCH_CHECK();
  return 0;
}
// dxq[b,n,c] (+)= sum_h sum_m (dS[b,h,n,m]*aq[m,h]) * qq[b,m,h,c]
// Wide dxq kernel (C % 384 == 0): 8 waves own a 128 x 384 tile of dx_hat[b], i.e. all channels of 128 tokens, so the
// dS panel is split into its bf16 planes once instead of three times; the A side is the wide dX kernel's ([row][k]
// planes, ds_read_b128 fragments), the B side the wide dW kernel's (int8 codes -> [k][c] bf16, transpose reads).  The
// k index runs over (head, key token); double-buffered LDS, two register slots, staging interleaved into the MFMA
// stream (see static_for).  k past the token count is zeroed through the per-k step (v_mul_legacy_f32: 0 * x = 0 even
// for the uninitialised pad columns of dS) and through zero codes.
template <bool F16>
__global__ __launch_bounds__(512) void qgemm_bf16s_nn_wide_kernel(QNnArgs p) {
  constexpr int BM = 128, NJ = 3, BN = 128 * NJ, NS = F16 ? 2 : 3;
  constexpr int PLANE = BM * QBS_LD;                // [row][k] bf16, 80 B rows
  constexpr int LDB = BN * 2 + 64;                  // [k][c] bf16
  constexpr int STAGE = NS * PLANE + QBS_BK * LDB;
  constexpr int CPR = BN / 8;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];
  int tm, b0;
  xcd_remap_grid(tm, b0);
  const int m0 = tm * BM;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 2, wn = wid & 3;
  const int l31 = lane & 31, lh = lane >> 5;
  const float* Ab = p.A + b0 * p.sA0;
  const int8_t* Bb = p.B + b0 * p.sB0;
  const int K = p.K;
  const int nkt = (K + QBS_BK - 1) / QBS_BK;
  const int T = nkt * p.nkb;

  const int kqa = (tid & 7) * 4;
  float sE = 1.f, inv_sE = 1.f;        // F16: the launch's power of two (see tn_wide_body); steps of every (key, head)
  const unsigned c64 = 0x64646464u;
  if constexpr (F16) {
    const float m = fmaxf(block512_absmax(p.s, K * p.ks_stride, reinterpret_cast<float*>(smem), tid), 1e-5f) * 1.0001f;
    const float am = ofq_amax_load(p.amax);
    f16_plane_scale(am == am ? am * m : am, sE, inv_sE);
  }
  unsigned rowoff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) rowoff[i] = (unsigned)(min(m0 + ((tid + 512 * i) >> 3), p.M - 1) * (int)p.lda);
  int b_row[NJ], b_col[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int f = tid + 512 * j;
    b_row[j] = f / CPR;
    b_col[j] = (f % CPR) * 8;
  }
  f32x4v ra[2][2];
  float rsv[2][4];
  u32x2v rb[2][NJ];
  int rkn[2];                                        // valid k of the slot's 4-chunk (0..4)
  bool rbk[2][NJ];
  // the load stream walks (head, k-tile) one tile per call; past the last tile it repeats it (never consumed)
  int lkb = 0, lkt = 0, lt = 0;
  auto gload = [&](auto SLOT) {
    constexpr int sl = decltype(SLOT)::value;
    const int k0 = lkt * QBS_BK;
    const int nv = min(max(K - (k0 + kqa), 0), 4);
    rkn[sl] = nv;
    const int ka = nv > 0 ? k0 + kqa : 0;            // an all-out chunk reads the row start (valid memory), then is zeroed
    const float* At = Ab + lkb * p.sAk + ka;
#pragma unroll
    for (int i = 0; i < 2; ++i) ra[sl][i] = *reinterpret_cast<const f32x4v*>(At + rowoff[i]);
    const float* sp = p.s + lkb;
#pragma unroll
    for (int e = 0; e < 4; ++e) rsv[sl][e] = sp[(int64_t)min(ka + e, K - 1) * p.ks_stride];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int k = k0 + b_row[j];
      rbk[sl][j] = k < K;
      rb[sl][j] = *reinterpret_cast<const u32x2v*>(Bb + lkb * p.sBk + (int64_t)min(k, K - 1) * p.ldb + b_col[j]);
    }
    const int adv = (lt + 1 < T) ? 1 : 0;             // scalar selects, no branch inside the MFMA stream
    lt += adv;
    lkt += adv;
    const int wrap = (lkt == nkt) ? 1 : 0;
    lkt = wrap ? 0 : lkt;
    lkb += wrap;
  };
  // pieces of the staging of one slot (shared by the prologue, which runs them back to back, and the k-step)
  float ksv[4], x_ = 0.f, r1_ = 0.f, p0v[2], p1v[2], r2v[2];
  unsigned lo[NS], hi[NS], bw[4];
  constexpr int NPA = F16 ? 10 : 17, NPB = 3, NPS = 2 * NPA + NPB * NJ;      // staging pieces; the k-step appends one load piece
  auto stage_piece = [&](unsigned char* nxt, auto SLOT, auto P_) {
    constexpr int sl = decltype(SLOT)::value;
    constexpr int P = decltype(P_)::value;
    if constexpr (P < 2 * NPA) {
      constexpr int i = P / NPA, r = P % NPA;
      if constexpr (r == 0 && i == 0) {
        asm volatile("" : "+v"(ra[sl][0]), "+v"(ra[sl][1]), "+v"(rsv[sl][0]), "+v"(rsv[sl][1]), "+v"(rsv[sl][2]), "+v"(rsv[sl][3]));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float t = valu_eff_scale(rsv[sl][e], p.gscale);
          ksv[e] = e < rkn[sl] ? (F16 ? t * sE : t) : 0.f;
        }
      }
      if constexpr (F16) {
        if constexpr (r < 8) {
          constexpr int pr = r / 4, st = r % 4, e = pr * 2;
          if constexpr (st == 0)      // (legacy multiply: 0 * x = 0 also for the never-written pad columns of dS)
            asm("v_mul_legacy_f32 %0, %2, %3\n\tv_mul_legacy_f32 %1, %4, %5" : "=&v"(x_), "=v"(r1_)
                : "v"(ra[sl][i][e]), "v"(ksv[e]), "v"(ra[sl][i][e + 1]), "v"(ksv[e + 1]));
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
          if constexpr (st == 0)
            asm("v_mul_legacy_f32 %0, %2, %3\n\tv_and_b32 %1, 0xffff0000, %0" : "=&v"(x_), "=v"(p0v[el]) : "v"(ra[sl][i][e]), "v"(ksv[e]));
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
    } else {
      constexpr int j = (P - 2 * NPA) / NPB, r = (P - 2 * NPA) % NPB;
      if constexpr (r == 0) {
        asm volatile("" : "+v"(rb[sl][j]));
        if constexpr (F16) valu_cvt4_i8_f16(rbk[sl][j] ? rb[sl][j][0] : 0u, c64, bw[0], bw[1]);
        else valu_cvt4_i8_bf16(rbk[sl][j] ? rb[sl][j][0] : 0u, bw[0], bw[1]);
      } else if constexpr (r == 1) {
        if constexpr (F16) valu_cvt4_i8_f16(rbk[sl][j] ? rb[sl][j][1] : 0u, c64, bw[2], bw[3]);
        else valu_cvt4_i8_bf16(rbk[sl][j] ? rb[sl][j][1] : 0u, bw[2], bw[3]);
      } else {
        *reinterpret_cast<uint4*>(&nxt[NS * PLANE + b_row[j] * LDB + b_col[j] * 2]) = make_uint4(bw[0], bw[1], bw[2], bw[3]);
      }
    }
  };

  f32x16q acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int p16 = lane & 15;
  const int fr_b = (8 * lh + (p16 >> 2)) * LDB + (16 * ((lane >> 4) & 1) + 4 * (p16 & 3)) * 2;
  using Slot0 = std::integral_constant<int, 0>;
  using Slot1 = std::integral_constant<int, 1>;
  constexpr int NM = 4 * NS * NJ, NP = NPS + 1;
  auto step = [&](const unsigned char* cur, unsigned char* nxt, auto SLOT) {
    const unsigned char* a = &cur[(wm * 64 + l31) * QBS_LD + lh * 16];
    const unsigned char* sbb = &cur[NS * PLANE + fr_b + wn * 32 * NJ * 2];
    bf16x8 av[NS][2], bv[2][NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) bv[0][j] = tr_frag_ld<LDB>(sbb + j * 64);
#pragma unroll
    for (int q = 0; q < NS; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i) av[q][i] = *reinterpret_cast<const bf16x8*>(a + q * PLANE + i * 32 * QBS_LD);
    __builtin_amdgcn_sched_barrier(0);
    static_for<NM>([&](auto G_) {
      constexpr int G = decltype(G_)::value;
      constexpr int ks = G / (2 * NS * NJ), q = (G / (2 * NJ)) % NS, i = (G / NJ) % 2, j = G % NJ;
      acc[i][j] = mfma_16b<F16>(av[q][i], bv[ks][j], acc[i][j]);
      if constexpr (ks == 0) {
        if constexpr (G < NJ) bv[1][G] = tr_frag_ld<LDB>(sbb + 16 * LDB + G * 64);
        if constexpr (j == NJ - 1) av[q][i] = *reinterpret_cast<const bf16x8*>(a + q * PLANE + i * 32 * QBS_LD + 32);
      }
      constexpr int P0 = G * NP / NM, P1 = (G + 1) * NP / NM;
      static_for<P1 - P0>([&](auto D_) {
        constexpr int P = P0 + decltype(D_)::value;
        if constexpr (P < NPS) stage_piece(nxt, SLOT, std::integral_constant<int, P>{});
        else gload(SLOT);                            // the slot is free again: loads of the tile three steps ahead
      });
      __builtin_amdgcn_sched_barrier(0);
    });
    lds_barrier();
  };
  gload(Slot0());
  gload(Slot1());
  static_for<NPS>([&](auto P_) { stage_piece(smem, Slot0(), P_); });
  gload(Slot0());
  lds_barrier();
  {
    int t = 0;
    for (; t + 1 < T; t += 2) {
      step(smem, smem + STAGE, Slot1());
      step(smem + STAGE, smem, Slot0());
    }
    if (t < T) step(smem, smem + STAGE, Slot1());
  }

  float* Cb = p.C + b0 * p.sC0;
  int ncol[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) ncol[j] = wn * 32 * NJ + j * 32 + l31;          // N == 384: every column exists
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int eb = 0; eb < 4; ++eb) {
      float old[4][NJ];
      if (p.accumulate) {      // old values fetched unconditionally on clamped rows, a quad of rows at a time
#pragma unroll
        for (int ee = 0; ee < 4; ++ee) {
          const int mc = min(m0 + wm * 64 + i * 32 + ee + 8 * eb + 4 * lh, p.M - 1);
#pragma unroll
          for (int j = 0; j < NJ; ++j) old[ee][j] = Cb[(int64_t)mc * p.ldc + ncol[j]];
        }
      }
#pragma unroll
      for (int ee = 0; ee < 4; ++ee) {
        const int m = m0 + wm * 64 + i * 32 + ee + 8 * eb + 4 * lh;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          if (m < p.M) {
            const float v = F16 ? acc[i][j][eb * 4 + ee] * inv_sE : acc[i][j][eb * 4 + ee];
            Cb[(int64_t)m * p.ldc + ncol[j]] = p.accumulate ? v + old[ee][j] : v;
          }
      }
    }
}

extern "C" int ofq_qattn_dxq_bf16s(const float* dS, const int8_t* qcodes, float* dxq, const float* sq, float gscale_q,
                                   int accumulate, int64_t B, int64_t H, int64_t N, int64_t C, int64_t ldS, const void* amax,
                                   ofq_stream_t stream) {
  if (!dS || !qcodes || !dxq || !sq || B <= 0 || H <= 0 || N <= 0 || (C & 15) || (ldS & 7) || ldS < N) return OFQ_EINVAL;
  QNnArgs a = {};
  a.amax = (const unsigned*)amax;      // two-plane fp16 form of the wide kernel (C == 384); elsewhere three bf16 planes
  a.A = dS; a.B = qcodes; a.C = dxq; a.s = sq; a.lda = ldS; a.ldb = H * C; a.ldc = C;
  a.sA0 = H * N * ldS; a.sB0 = N * H * C; a.sC0 = N * C; a.sAk = N * ldS; a.sBk = C;
  a.M = (int)N; a.N = (int)C; a.K = (int)N; a.nkb = (int)H; a.ks_stride = (int)H; a.accumulate = accumulate; a.gscale = gscale_q;
  a.tiles_m = (int)ceil_div(N, 128); a.tiles_n = (int)ceil_div(C, 128);
  static const bool nn_narrow = getenv("OFQ_NN_NARROW") != nullptr;           // A/B switch (tools/)
  if (C == 384 && !nn_narrow && N * ldS < (1ll << 31)) {
    if (amax) hipLaunchKernelGGL(qgemm_bf16s_nn_wide_kernel<true>, dim3((unsigned)a.tiles_m, (unsigned)B), dim3(512), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(qgemm_bf16s_nn_wide_kernel<false>, dim3((unsigned)a.tiles_m, (unsigned)B), dim3(512), 0, (hipStream_t)stream, a);
  } else
    hipLaunchKernelGGL(qgemm_bf16s_nn_kernel, dim3((unsigned)(a.tiles_m * a.tiles_n), (unsigned)B), dim3(256), 0,
                       (hipStream_t)stream, a);
  OFQ_LAUNCH_CHECK();
  return 0;
}
