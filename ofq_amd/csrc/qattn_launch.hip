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
      B * N * H >= (1ll << 31) || B * N * H * C >= (1ll << 32))
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
  if (M <= 64 && N <= 64) {
    a.tiles_m = a.tiles_n = 1;
    hipLaunchKernelGGL((qgemm_i8_nt_kernel<EPI, 1>), dim3(1u, (unsigned)batches), dim3(256), 0, st, a);
  } else if (EPI == 2 && N <= 64 && M > 128) {     // P.V of a long sequence: 256 x 64 tiles, four waves stacked
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
  const int mchunks = (int)ceil_div(N, 64);
  const int64_t ntasks = B * H * mchunks;
  // window-sized token counts only: at N = 198 (four row chunks, seven key blocks per wave) the workgroup tile wins
  // (measured: 27.73 vs 27.87 ms/step for DeiT-S; Swin-T 50.43 -> 49.86 ms with the wave tile)
  if (N <= 64 && (d == 64 || d == 32 || d == 16 || d == 48) && al16(dO) && al16(sv) && (C & 3) == 0 &&
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
  if (d <= 64 && Np > 128) {       // one head's channels x all keys: 64 x 256 tiles, four waves side by side
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
  if (C % 384 == 0 && N >= QTN_BK && N * ldS < (1ll << 31) && N * C < (1ll << 31)) {
    a.tiles_n = (int)(C / 384);       // one split of a dS panel feeds all 384 columns
    int64_t T = (int64_t)a.tiles_m * a.tiles_n * H;                           // tiles per image
    // Stacked form: the H heads share the B operand (x_hat of the image), so their ldS-row outputs are tiled as one
    // (H ldS)-row matrix: 10 tiles of 128 rows per DeiT-S image instead of 6 x 2 whose second one holds 69 rows.
    bool stacked = false;
    if (a.tiles_n == 1 && H > 1 && ceil_div(H * ldS, 128) < (int64_t)a.tiles_m * H && H * ldS < 2048 &&
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
    if (T >= 2) {
      // persistent workgroups: each walks `tpw` tiles of one image (about one workgroup per CU in total)
      int64_t tpw = (B * T) / 256;
      if (const char* e = getenv("OFQ_TN_STREAM_TPW")) tpw = atoi(e);       // test hook: tiles per workgroup
      tpw = tpw < 1 ? 1 : (tpw > T ? T : tpw);
      while (T % tpw) --tpw;                                                  // equal chunks
      int stagger = 0;                                                      // measured: 121 us without, 149 us with a 6-phase shift
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
  if (C == 384 && N * ldS < (1ll << 31)) {
    if (amax) hipLaunchKernelGGL(qgemm_bf16s_nn_wide_kernel<true>, dim3((unsigned)a.tiles_m, (unsigned)B), dim3(512), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(qgemm_bf16s_nn_wide_kernel<false>, dim3((unsigned)a.tiles_m, (unsigned)B), dim3(512), 0, (hipStream_t)stream, a);
  } else
    hipLaunchKernelGGL(qgemm_bf16s_nn_kernel, dim3((unsigned)(a.tiles_m * a.tiles_n), (unsigned)B), dim3(256), 0,
                       (hipStream_t)stream, a);
  OFQ_LAUNCH_CHECK();
  return 0;
}
