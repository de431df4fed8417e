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
                                      int64_t lda, int64_t ldb, int64_t ldx, void* ws, size_t ws_bytes, const void* amax,
                                      ofq_stream_t stream) {
  if (!dY || !B_bf16 || !x || !lsq_s || !dx || !ws || M <= 0 || N <= 128 || K <= 0 || S <= 0) return OFQ_EINVAL;
  if ((K & 7) || (lda & 3) || (ldb & 7) || ldx < N || !al16(dY) || !al16(B_bf16) || (k_scale && !al16(k_scale)) ||
      M >= (1ll << 30) || N >= (1ll << 30))
    return OFQ_EINVAL;
  if (ws_bytes < ofq_qgemm_bf16s_nt_lsq_ws_bytes(M, N)) return OFQ_ENOWS;
  int64_t tm, tn;
  nt_lsq_tiles(M, N, &tm, &tn);
  QGemmArgs a = {};
  a.A = dY; a.B = B_bf16; a.C = dx; a.s = k_scale; a.amax = (const unsigned*)amax;      // amax: fp16 codes, two planes of dY
  a.lda = lda; a.ldb = ldb; a.ldc = ldx; a.M = (int)M; a.N = (int)N; a.K = (int)K;
  a.tiles_m = (int)tm; a.tiles_n = (int)tn; a.alpha = alpha; a.nb1 = 1;
  a.lx = x; a.ldlx = ldx; a.ls = lsq_s; a.lS = (int)S; a.lgscale = gscale; a.lb4 = b4; a.llo = (float)lo; a.lhi = (float)hi;
  a.lgelu = gelu; a.lrow = (float*)ws; a.lcol = (float*)ws + (size_t)M * tn;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)(tm * tn));
  if (amax) {
    if (N > 256) hipLaunchKernelGGL((qgemm_bf16s_nt_wide_kernel<3, true, true>), grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((qgemm_bf16s_nt_wide_kernel<2, true, true>), grid, dim3(512), 0, st, a);
  } else if (N > 256) hipLaunchKernelGGL((qgemm_bf16s_nt_wide_kernel<3, true>), grid, dim3(512), 0, st, a);
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
