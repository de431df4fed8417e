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
    // for the four rows of the group.  Row offsets are 32-bit (rows x K < 2^32, host check) on the uniform base pointer, the
    // loads unconditional on clamped offsets, every offset register live until its data has been consumed (the empty asm
    // statements below).  WHY this form: round 6 traced the one non-reproducible result of a training step on a SHARED GPU to
    // round 5's form of this routine (loads on 64-bit row pointers under `if (k0 < K)`): tq of rows (row % 64) in {35, 39, 43, 47}
    // -- lanes 48..63 of the THIRD row load, `global_load_dwordx4 v[2:5], v[2:3], off` in that build -- wrong in 1-4 of 2376 rows
    // about once in 400 calls, with the inputs in memory verified before and after the launch, and only while another process
    // had work on the GPU (tools/two_rank_trace.py DUMP_OP=qattn_prep).  Round 5's form, that form with its row
    // pointers kept live (another register allocation of the same loads), and this form without the keep-alive were compared under
    // -DRD16_VARIANT=3 / 2 / 1 in commit a29d053 (tools/gpu/r06_variants.sh, profiles/r06_shared_gpu_rowdot_variants.txt): 32
    // differing repetitions of 1398 for round 5's form, 0 of 1398 for each of the others.  The mechanism inside the failing build
    // is NOT established (its waits are the right ones; the aliased destination is still there in the passing keep-alive build).
    const bool in0 = k0 < K, in1 = k0 + 256 < K;
    i32x4 c0[RD16_RPG], c1[RD16_RPG];
    unsigned off0[RD16_RPG], off1[RD16_RPG];
#pragma unroll
    for (int j = 0; j < RD16_RPG; ++j) {
      const unsigned n = (unsigned)min(nb + 16 * j, N - 1);
      off0[j] = n * (unsigned)K + (unsigned)(in0 ? k0 : 0);
      off1[j] = n * (unsigned)K + (unsigned)(in1 ? k0 + 256 : 0);
    }
#pragma unroll
    for (int j = 0; j < RD16_RPG; ++j) {
      c0[j] = *reinterpret_cast<const i32x4*>(codes + off0[j]);
      c1[j] = *reinterpret_cast<const i32x4*>(codes + off1[j]);
    }
    float4 v0[4], v1[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      v0[w] = *reinterpret_cast<const float4*>(vec + (in0 ? k0 : 0) + 4 * w);
      v1[w] = *reinterpret_cast<const float4*>(vec + (in1 ? k0 + 256 : 0) + 4 * w);
    }
#pragma unroll
    for (int j = 0; j < RD16_RPG; ++j) {
      float t = 0.f;                                         // (the same association as ever: 0 + chunk 0, then + chunk 1)
      fma16(t, k0, c0[j], v0);
      const float a = in0 ? t : 0.f;
      t = a;
      fma16(t, k0 + 256, c1[j], v1);
      asm volatile("" :: "v"(off0[j]), "v"(off1[j]));       // (the offsets outlive the loads' results)
      acc[j] = in1 ? t : a;
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
  if ((cols & 15) == 0 && al16(codes) && al16(vec) && rows * cols < (1ll << 32))        // (32-bit row offsets)
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
