#pragma once
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
