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
  if (!a.C || a.split != 1 || (a.N & 15) || (a.ldb & 15) || (int64_t)a.Ktok * a.lda >= (1ll << 31) ||
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
  const int kstep = QTN_BK % p.S;        // any S >= 1: kmod < S and kstep < S, so one conditional subtraction per k-step suffices
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
      kmod[i] += kstep;                                    // gload runs on consecutive k-steps: k mod S incrementally
      kmod[i] -= (kmod[i] >= p.S) ? p.S : 0;
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
        kmod[i] += kstep;
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
  __shared__ float dbs[3];
  const int tx = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int N4 = N >> 2;
  const int64_t MN = (int64_t)M * N;
  const int g0 = bx * 64;
  const int total = M * N4;
  const int o_first = g0 / N4, o_last = min(g0 + 63, total - 1) / N4;      // N4 >= 32 (N >= 128): at most three rows per block
  if (part < 3) {
    const int o = min(o_first + part, o_last);
    float v = 0.f;
    if (csum) {
      for (int s = tx; s < split; s += 64) v += csum[(int64_t)s * M + o];
      v = ofq_wave_sum(v);
      const int gfirst = o * N4;                                             // the block holding chunk (o, 0) publishes db[o]
      if (tx == 0 && gfirst >= g0 && gfirst < g0 + 64 && o_first + part <= o_last) db[o] = v;
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
      const float dbo = dbs[o - o_first];
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
  static const bool narrow_small_s = getenv("OFQ_TN_NARROW_SMALL_S") != nullptr;      // A/B hook: round 5's routing of S < 32
  if (N > 128 && (N & 7) == 0 && Ktok * lda < (1ll << 31) && Ktok * ldb < (1ll << 31) && !(narrow_small_s && S < QTN_BK)) {
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
  if ((N & 3) == 0 && N >= 128)
    hipLaunchKernelGGL(qgemm_tn_reduce4_kernel, dim3((unsigned)ceil_div(M * (N / 4), 64)), dim3(256), 0, st, (const float*)ws, dW,
                       compute_db ? (const float*)a.csum : (const float*)nullptr, db, baft, (int)M, (int)N, split);
  else
    hipLaunchKernelGGL(qgemm_tn_reduce_kernel, dim3((unsigned)M), dim3(256), 0, st, (const float*)ws, dW,
                       compute_db ? (const float*)a.csum : (const float*)nullptr, db, baft, (int)M, (int)N, split);
  OFQ_LAUNCH_CHECK();
  return 0;
}

static bool tn_wide_ok(int64_t Ktok, int64_t N, int64_t S, int64_t lda, int64_t ldb) {
  (void)S;                         // (any step-vector length: Swin's 4-D MLP quantisers have S = 7 / 14 / 28)
  return N > 128 && (N & 7) == 0 && Ktok * lda < (1ll << 31) && Ktok * ldb < (1ll << 31);
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
    if (!tn_wide_ok(q.Ktok, q.N, q.S, q.lda, q.ldb) || (q.N % 384 == 0) != three) return OFQ_EINVAL;      // (N > 128: the reduce's bound)
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
