// K6-K9, K11  fp32 GEMM on the matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, one rounding per product,
// bit-identical to an fmaf chain) for every dense product of the fake-quant path: F.linear (qlinear.py:69),
// the two QKR einsums (attention.py:200, :210), W_q^T W_k (attention.py:193), P*V (attention.py:219), q k^T
// (attention.py:96) and all of their autograd backward products.
//
// Tiling for gfx950: 256-thread workgroup = 4 waves in a 2x2 grid, workgroup tile BM x BN (128x128 or
// 128x64), BK = 32.  Operands are staged global -> registers -> LDS with a two-deep LDS ring (one barrier per
// k-tile; the next tile's global loads are in flight during the current tile's 16 k-pairs of MFMA).
//   K-contiguous operand  (A of NN/NT, B of NT):  LDS [rows][BK+1]  -> ds_read_b32, lane stride 33: no conflicts
//   M/N-contiguous operand (A of TN, B of NN/TN): LDS [BK][cols]    -> ds_read_b32, consecutive lanes
// The k order inside an MFMA pair is (k, k+1) for lanes 0-31 / 32-63, the native 32x32x2 layout.
// Workgroup ids are remapped so that consecutive tiles (sharing an A row-panel) sit on one XCD's L2.
// Roofline: MFMA, 157.3 TFLOP/s fp32 peak; 2*M*N*K flops per call.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
  const float* A; const float* B; float* C; const float* bias; float* ws;
  int64_t M, N, K, lda, ldb, ldc;
  int64_t sA0, sA1, sB0, sB1, sC0, sC1, sAk, sBk;
  int nb1, nkb, split, tiles_m, tiles_n, vecA, vecB, accumulate;
  float alpha;
};

#define GEMM_BK 32

__device__ __forceinline__ float4 ld4_guard(const float* p, bool vec, int valid) {
  // valid = number of in-range elements starting at p (<=0: none)
  if (valid >= 4 && vec) return *reinterpret_cast<const float4*>(p);
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
  if (valid > 0) r.x = p[0];
  if (valid > 1) r.y = p[1];
  if (valid > 2) r.z = p[2];
  if (valid > 3) r.w = p[3];
  return r;
}

template <int BM, int BN, bool AKC, bool BKC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs p) {
  constexpr int BK = GEMM_BK;
  constexpr int LDA_S = AKC ? (BK + 1) : BM;
  constexpr int LDB_S = BKC ? (BK + 1) : BN;
  constexpr int A_ELEMS = AKC ? BM * (BK + 1) : BK * BM;
  constexpr int B_ELEMS = BKC ? BN * (BK + 1) : BK * BN;
  constexpr int NLA = BM / 32;  // float4 loads per thread for A
  constexpr int NLB = BN / 32;
  constexpr int RM = BM / 64, RN = BN / 64;
  __shared__ __attribute__((aligned(16))) float smem[2 * (A_ELEMS + B_ELEMS)];
  constexpr int STAGE = A_ELEMS + B_ELEMS;   // stage s: A at smem + s*STAGE, B right behind it

  // ---- XCD-aware bijective remap of the tile id (hardware places block b on XCD b % 8) ----
  const int ntiles = p.tiles_m * p.tiles_n;
  int tile = blockIdx.x;
  {
    const int q = ntiles >> 3, r = ntiles & 7;
    const int xcd = tile & 7, loc = tile >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int tm = tile / p.tiles_n, tn = tile % p.tiles_n;
  const int sidx = blockIdx.y % p.split, bidx = blockIdx.y / p.split;
  const int b0 = bidx / p.nb1, b1 = bidx % p.nb1;
  const float* Ab = p.A + b0 * p.sA0 + b1 * p.sA1;
  const float* Bb = p.B + b0 * p.sB0 + b1 * p.sB1;
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;

  const int nkt = (int)((p.K + BK - 1) / BK);
  const int T = p.nkb * nkt;
  const int tps = (T + p.split - 1) / p.split;
  const int t_begin = sidx * tps;
  const int t_end = min(T, t_begin + tps);

  f32x16 acc[RM][RN];
#pragma unroll
  for (int i = 0; i < RM; ++i)
#pragma unroll
    for (int j = 0; j < RN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  float4 ra[NLA], rb[NLB];

  auto gload = [&](int t) {
    const int kb = t / nkt, kt = t % nkt;
    const int64_t k0 = (int64_t)kt * BK;
    const float* At = Ab + kb * p.sAk;
    const float* Bt = Bb + kb * p.sBk;
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
      const int f = tid + 256 * i;
      if (AKC) {
        const int row = f >> 3, kq = f & 7;
        const int64_t m = m0 + row, k = k0 + kq * 4;
        ra[i] = ld4_guard(At + m * p.lda + k, p.vecA, (m < p.M) ? (int)min((int64_t)4, p.K - k) : 0);
      } else {
        const int k = f / (BM / 4), m4 = f % (BM / 4);
        const int64_t m = m0 + m4 * 4, kk = k0 + k;
        ra[i] = ld4_guard(At + kk * p.lda + m, p.vecA, (kk < p.K) ? (int)min((int64_t)4, p.M - m) : 0);
      }
    }
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
      const int f = tid + 256 * i;
      if (BKC) {
        const int row = f >> 3, kq = f & 7;
        const int64_t n = n0 + row, k = k0 + kq * 4;
        rb[i] = ld4_guard(Bt + n * p.ldb + k, p.vecB, (n < p.N) ? (int)min((int64_t)4, p.K - k) : 0);
      } else {
        const int k = f / (BN / 4), n4 = f % (BN / 4);
        const int64_t n = n0 + n4 * 4, kk = k0 + k;
        rb[i] = ld4_guard(Bt + kk * p.ldb + n, p.vecB, (kk < p.K) ? (int)min((int64_t)4, p.N - n) : 0);
      }
    }
  };
  auto lstore = [&](int buf) {
    float* a = smem + buf * STAGE;
    float* b = a + A_ELEMS;
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
      const int f = tid + 256 * i;
      if (AKC) {
        float* d = a + (f >> 3) * LDA_S + (f & 7) * 4;
        d[0] = ra[i].x; d[1] = ra[i].y; d[2] = ra[i].z; d[3] = ra[i].w;
      } else {
        *reinterpret_cast<float4*>(a + (f / (BM / 4)) * LDA_S + (f % (BM / 4)) * 4) = ra[i];
      }
    }
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
      const int f = tid + 256 * i;
      if (BKC) {
        float* d = b + (f >> 3) * LDB_S + (f & 7) * 4;
        d[0] = rb[i].x; d[1] = rb[i].y; d[2] = rb[i].z; d[3] = rb[i].w;
      } else {
        *reinterpret_cast<float4*>(b + (f / (BN / 4)) * LDB_S + (f % (BN / 4)) * 4) = rb[i];
      }
    }
  };

  if (t_begin < t_end) {
    gload(t_begin);
    lstore(0);
    __syncthreads();
    int buf = 0;
    for (int t = t_begin; t < t_end; ++t) {
      const bool more = (t + 1) < t_end;
      if (more) gload(t + 1);
      const float* a = smem + buf * STAGE;
      const float* b = a + A_ELEMS;
      const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
      for (int kk = 0; kk < BK / 2; ++kk) {
        const int kidx = 2 * kk + lh;
        float av[RM], bv[RN];
#pragma unroll
        for (int i = 0; i < RM; ++i) {
          const int m = wm * (BM / 2) + i * 32 + l31;
          av[i] = AKC ? a[m * LDA_S + kidx] : a[kidx * LDA_S + m];
        }
#pragma unroll
        for (int j = 0; j < RN; ++j) {
          const int n = wn * (BN / 2) + j * 32 + l31;
          bv[j] = BKC ? b[n * LDB_S + kidx] : b[kidx * LDB_S + n];
        }
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
          for (int j = 0; j < RN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
      }
      if (more) lstore(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
  }

  // ---- epilogue: C/D layout of 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5) ----
  const int l31 = lane & 31, lh = lane >> 5;
  if (p.split == 1) {
    float* Cb = p.C + b0 * p.sC0 + b1 * p.sC1;
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
      for (int j = 0; j < RN; ++j) {
        const int64_t n = n0 + wn * (BN / 2) + j * 32 + l31;
        if (n >= p.N) continue;
        const float bz = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int64_t m = m0 + wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
          if (m < p.M) {
            float v = __fadd_rn(__fmul_rn(acc[i][j][e], p.alpha), bz);
            float* dst = Cb + m * p.ldc + n;
            if (p.accumulate) v += *dst;
            *dst = v;
          }
        }
      }
  } else {
    float* Wb = p.ws + ((int64_t)bidx * p.split + sidx) * p.M * p.N;
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
      for (int j = 0; j < RN; ++j) {
        const int64_t n = n0 + wn * (BN / 2) + j * 32 + l31;
        if (n >= p.N) continue;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int64_t m = m0 + wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
          if (m < p.M) Wb[m * p.N + n] = acc[i][j][e];
        }
      }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Fast path: branch-free staging.  Preconditions (checked on the host): 16-byte aligned bases, every leading
// dimension / batch stride a multiple of 4 floats, and ld >= roundup4(contiguous extent), so a float4 read that
// starts inside a row never leaves the row's allocation.  Out-of-range rows are read from a clamped address and
// zeroed with selects (no divergent control flow around the loads: the compiler can batch all global loads of a
// k-tile and wait once), the tail of a contiguous extent that is not a multiple of 4 is zeroed per element.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 zero_tail(float4 v, int rem) {
  v.x = rem > 0 ? v.x : 0.f;
  v.y = rem > 1 ? v.y : 0.f;
  v.z = rem > 2 ? v.z : 0.f;
  v.w = rem > 3 ? v.w : 0.f;
  return v;
}

template <int BM, int BN, bool AKC, bool BKC>
__global__ __launch_bounds__(256) void gemm_f32_fast_kernel(GemmArgs p) {
  constexpr int BK = GEMM_BK;
  constexpr int LDA_S = AKC ? (BK + 1) : BM;
  constexpr int LDB_S = BKC ? (BK + 1) : BN;
  constexpr int A_ELEMS = AKC ? BM * (BK + 1) : BK * BM;
  constexpr int B_ELEMS = BKC ? BN * (BK + 1) : BK * BN;
  constexpr int STAGE = A_ELEMS + B_ELEMS;
  constexpr int NLA = BM / 32, NLB = BN / 32;
  constexpr int RM = BM / 64, RN = BN / 64;
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

  const int ntiles = p.tiles_m * p.tiles_n;
  int tile = blockIdx.x;
  {
    const int q = ntiles >> 3, r = ntiles & 7;
    const int xcd = tile & 7, loc = tile >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int tm = tile / p.tiles_n, tn = tile % p.tiles_n;
  const int sidx = blockIdx.y % p.split, bidx = blockIdx.y / p.split;
  const int b0 = bidx / p.nb1, b1 = bidx % p.nb1;
  const float* Ab = p.A + b0 * p.sA0 + b1 * p.sA1;
  const float* Bb = p.B + b0 * p.sB0 + b1 * p.sB1;
  const int m0 = tm * BM, n0 = tn * BN;
  const int M = (int)p.M, N = (int)p.N, K = (int)p.K;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;

  const int nkt = (K + BK - 1) / BK;
  const int T = p.nkb * nkt;
  const int tps = (T + p.split - 1) / p.split;
  const int t_begin = sidx * tps;
  const int t_end = min(T, t_begin + tps);

  // per-thread staging descriptors: element offset at k-tile 0, and what is needed to mask
  int64_t offA[NLA], offB[NLB];
  int remA[NLA], remB[NLB];      // KC: row valid ? 4 : 0 (k tail handled per tile);  MC: valid count along the contiguous dim
  int kkA[NLA], kkB[NLB];        // KC: k offset inside the tile (kq*4);               MC: k row inside the tile
#pragma unroll
  for (int i = 0; i < NLA; ++i) {
    const int f = tid + 256 * i;
    if (AKC) {
      const int row = f >> 3, kq = f & 7;
      const int m = m0 + row;
      offA[i] = (int64_t)min(m, M - 1) * p.lda + kq * 4;
      remA[i] = m < M ? 4 : 0;
      kkA[i] = kq * 4;
    } else {
      const int k = f / (BM / 4), m = m0 + (f % (BM / 4)) * 4;
      offA[i] = (int64_t)k * p.lda + (m < M ? m : 0);
      remA[i] = min(4, M - m);
      kkA[i] = k;
    }
  }
#pragma unroll
  for (int i = 0; i < NLB; ++i) {
    const int f = tid + 256 * i;
    if (BKC) {
      const int row = f >> 3, kq = f & 7;
      const int n = n0 + row;
      offB[i] = (int64_t)min(n, N - 1) * p.ldb + kq * 4;
      remB[i] = n < N ? 4 : 0;
      kkB[i] = kq * 4;
    } else {
      const int k = f / (BN / 4), n = n0 + (f % (BN / 4)) * 4;
      offB[i] = (int64_t)k * p.ldb + (n < N ? n : 0);
      remB[i] = min(4, N - n);
      kkB[i] = k;
    }
  }

  f32x16 acc[RM][RN];
#pragma unroll
  for (int i = 0; i < RM; ++i)
#pragma unroll
    for (int j = 0; j < RN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  float4 ra[NLA], rb[NLB];

  auto gload = [&](int t) {
    const int kb = t / nkt, kt = t - kb * nkt;
    const int k0 = kt * BK;
    const float* At = Ab + kb * p.sAk;
    const float* Bt = Bb + kb * p.sBk;
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
      if (AKC) {
        const int k = k0 + kkA[i];
        const float4 v = *reinterpret_cast<const float4*>(At + offA[i] + (k < K ? k0 : -kkA[i]));
        ra[i] = zero_tail(v, min(remA[i], K - k));
      } else {
        const int k = k0 + kkA[i];
        const float4 v = *reinterpret_cast<const float4*>(At + offA[i] + (k < K ? (int64_t)k0 * p.lda : -(int64_t)kkA[i] * p.lda));
        ra[i] = zero_tail(v, k < K ? remA[i] : 0);
      }
    }
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
      if (BKC) {
        const int k = k0 + kkB[i];
        const float4 v = *reinterpret_cast<const float4*>(Bt + offB[i] + (k < K ? k0 : -kkB[i]));
        rb[i] = zero_tail(v, min(remB[i], K - k));
      } else {
        const int k = k0 + kkB[i];
        const float4 v = *reinterpret_cast<const float4*>(Bt + offB[i] + (k < K ? (int64_t)k0 * p.ldb : -(int64_t)kkB[i] * p.ldb));
        rb[i] = zero_tail(v, k < K ? remB[i] : 0);
      }
    }
  };
  auto lstore = [&](int buf) {
    float* a = smem + buf * STAGE;
    float* b = a + A_ELEMS;
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
      const int f = tid + 256 * i;
      if (AKC) {
        float* d = a + (f >> 3) * LDA_S + (f & 7) * 4;
        d[0] = ra[i].x; d[1] = ra[i].y; d[2] = ra[i].z; d[3] = ra[i].w;
      } else {
        *reinterpret_cast<float4*>(a + (f / (BM / 4)) * LDA_S + (f % (BM / 4)) * 4) = ra[i];
      }
    }
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
      const int f = tid + 256 * i;
      if (BKC) {
        float* d = b + (f >> 3) * LDB_S + (f & 7) * 4;
        d[0] = rb[i].x; d[1] = rb[i].y; d[2] = rb[i].z; d[3] = rb[i].w;
      } else {
        *reinterpret_cast<float4*>(b + (f / (BN / 4)) * LDB_S + (f % (BN / 4)) * 4) = rb[i];
      }
    }
  };

  const int l31 = lane & 31, lh = lane >> 5;
  if (t_begin < t_end) {
    gload(t_begin);
    lstore(0);
    __syncthreads();
    int buf = 0;
    for (int t = t_begin; t < t_end; ++t) {
      const bool more = (t + 1) < t_end;
      if (more) gload(t + 1);
      const float* a = smem + buf * STAGE + (AKC ? (wm * (BM / 2) + l31) * LDA_S + lh : lh * LDA_S + wm * (BM / 2) + l31);
      const float* b = smem + buf * STAGE + A_ELEMS +
                       (BKC ? (wn * (BN / 2) + l31) * LDB_S + lh : lh * LDB_S + wn * (BN / 2) + l31);
#pragma unroll
      for (int kk = 0; kk < BK / 2; ++kk) {
        float av[RM], bv[RN];
#pragma unroll
        for (int i = 0; i < RM; ++i) av[i] = AKC ? a[i * 32 * LDA_S + 2 * kk] : a[2 * kk * LDA_S + i * 32];
#pragma unroll
        for (int j = 0; j < RN; ++j) bv[j] = BKC ? b[j * 32 * LDB_S + 2 * kk] : b[2 * kk * LDB_S + j * 32];
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
          for (int j = 0; j < RN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
      }
      if (more) lstore(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
  }

  if (p.split == 1) {
    float* Cb = p.C + b0 * p.sC0 + b1 * p.sC1;
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
      for (int j = 0; j < RN; ++j) {
        const int n = n0 + wn * (BN / 2) + j * 32 + l31;
        if (n >= N) continue;
        const float bz = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = m0 + wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
          if (m < M) {
            float v = __fadd_rn(__fmul_rn(acc[i][j][e], p.alpha), bz);
            float* dst = Cb + (int64_t)m * p.ldc + n;
            if (p.accumulate) v += *dst;
            *dst = v;
          }
        }
      }
  } else {
    float* Wb = p.ws + ((int64_t)bidx * p.split + sidx) * p.M * p.N;
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
      for (int j = 0; j < RN; ++j) {
        const int n = n0 + wn * (BN / 2) + j * 32 + l31;
        if (n >= N) continue;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = m0 + wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
          if (m < M) Wb[(int64_t)m * p.N + n] = acc[i][j][e];
        }
      }
  }
}

// split-K second stage: C = alpha * sum_s ws[s] + bias (+ C), fixed order
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(GemmArgs p) {
  const int64_t MN = p.M * p.N;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int bidx = blockIdx.y;
  if (i >= MN) return;
  const float* w = p.ws + (int64_t)bidx * p.split * MN + i;
  float acc = 0.f;
  for (int s = 0; s < p.split; ++s) acc += w[(int64_t)s * MN];
  const int64_t m = i / p.N, n = i % p.N;
  const int b0 = bidx / p.nb1, b1 = bidx % p.nb1;
  float* dst = p.C + b0 * p.sC0 + b1 * p.sC1 + m * p.ldc + n;
  float v = __fadd_rn(__fmul_rn(acc, p.alpha), p.bias ? p.bias[n] : 0.f);
  if (p.accumulate) v += *dst;
  *dst = v;
}

static bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

extern "C" size_t ofq_gemm_ws_bytes(const ofq_gemm_desc* d) {
  if (!d || d->split_k <= 1) return 0;
  return (size_t)d->nb0 * d->nb1 * d->split_k * d->M * d->N * sizeof(float);
}

template <int BM, int BN>
static int gemm_launch(const GemmArgs& a, const ofq_gemm_desc* d, int nbatch, bool fast, hipStream_t st) {
  dim3 grid((unsigned)(a.tiles_m * a.tiles_n), (unsigned)(nbatch * a.split));
  const bool akc = d->transA == 0, bkc = d->transB != 0;
  if (fast) {
    if (akc && bkc) hipLaunchKernelGGL((gemm_f32_fast_kernel<BM, BN, true, true>), grid, dim3(256), 0, st, a);
    else if (akc && !bkc) hipLaunchKernelGGL((gemm_f32_fast_kernel<BM, BN, true, false>), grid, dim3(256), 0, st, a);
    else if (!akc && !bkc) hipLaunchKernelGGL((gemm_f32_fast_kernel<BM, BN, false, false>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((gemm_f32_fast_kernel<BM, BN, false, true>), grid, dim3(256), 0, st, a);
  } else {
    if (akc && bkc) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, true, true>), grid, dim3(256), 0, st, a);
    else if (akc && !bkc) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, true, false>), grid, dim3(256), 0, st, a);
    else if (!akc && !bkc) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, false, false>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, false, true>), grid, dim3(256), 0, st, a);
  }
  OFQ_LAUNCH_CHECK();
  return 0;
}

static inline int64_t round4(int64_t x) { return (x + 3) & ~(int64_t)3; }

extern "C" int ofq_gemm_f32(const ofq_gemm_desc* d, void* ws, size_t ws_bytes, ofq_stream_t stream) {
  if (!d || !d->A || !d->B || !d->C || d->M <= 0 || d->N <= 0 || d->K <= 0) return OFQ_EINVAL;
  if (d->nb0 < 1 || d->nb1 < 1 || d->nkb < 1 || d->split_k < 1) return OFQ_EINVAL;
  if (d->split_k > 1 && (!ws || ws_bytes < ofq_gemm_ws_bytes(d))) return OFQ_ENOWS;
  GemmArgs a = {};
  a.A = d->A; a.B = d->B; a.C = d->C; a.bias = d->bias; a.ws = (float*)ws;
  a.M = d->M; a.N = d->N; a.K = d->K; a.lda = d->lda; a.ldb = d->ldb; a.ldc = d->ldc;
  a.sA0 = d->sA0; a.sA1 = d->sA1; a.sB0 = d->sB0; a.sB1 = d->sB1; a.sC0 = d->sC0; a.sC1 = d->sC1;
  a.sAk = d->sAk; a.sBk = d->sBk;
  a.nb1 = d->nb1; a.nkb = d->nkb; a.split = d->split_k; a.accumulate = d->accumulate; a.alpha = d->alpha;
  a.vecA = aligned16(d->A) && !(d->lda & 3) && !(d->sA0 & 3) && !(d->sA1 & 3) && !(d->sAk & 3);
  a.vecB = aligned16(d->B) && !(d->ldb & 3) && !(d->sB0 & 3) && !(d->sB1 & 3) && !(d->sBk & 3);
  const int nbatch = d->nb0 * d->nb1;
  hipStream_t st = (hipStream_t)stream;
  // fast (branch-free) staging needs float4-safe geometry; anything else takes the generic kernel
  const int64_t contigA = d->transA ? d->M : d->K, contigB = d->transB ? d->K : d->N;
  const bool fast = a.vecA && a.vecB && d->lda >= round4(contigA) && d->ldb >= round4(contigB) &&
                    d->M < (1ll << 30) && d->N < (1ll << 30) && d->K < (1ll << 30);
  // tile shape: 128x64 for narrow outputs; 64x64 when 128x128 tiles would leave the last round of the
  // 256 CUs mostly idle (e.g. M=25344, N=384: 594 tiles = 2.3 rounds)
  int rc;
  const int64_t t128 = ceil_div(d->M, 128) * ceil_div(d->N, 128) * nbatch * d->split_k;
  const int64_t t64 = ceil_div(d->M, 64) * ceil_div(d->N, 64) * nbatch * d->split_k;
  const double eff128 = (double)t128 / (double)(ceil_div(t128, 256) * 256);
  const double eff64 = (double)t64 / (double)(ceil_div(t64, 256) * 256);
  const int force = d->tile_hint;
  if (force == 64 || (force == 0 && d->N > 64 && fast && eff128 < 0.85 && eff64 > eff128 + 0.08)) {
    a.tiles_m = (int)ceil_div(d->M, 64); a.tiles_n = (int)ceil_div(d->N, 64);
    rc = gemm_launch<64, 64>(a, d, nbatch, fast, st);
  } else if (d->N <= 64 && force != 128) {
    a.tiles_m = (int)ceil_div(d->M, 128); a.tiles_n = (int)ceil_div(d->N, 64);
    rc = gemm_launch<128, 64>(a, d, nbatch, fast, st);
  } else {
    a.tiles_m = (int)ceil_div(d->M, 128); a.tiles_n = (int)ceil_div(d->N, 128);
    rc = gemm_launch<128, 128>(a, d, nbatch, fast, st);
  }
  if (rc) return rc;
  if (d->split_k > 1) {
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)ceil_div(d->M * d->N, 256), (unsigned)nbatch),
                       dim3(256), 0, st, a);
    OFQ_LAUNCH_CHECK();
  }
  return 0;
}
