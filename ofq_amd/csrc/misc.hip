// Column sums (bias gradients) and the CGA step hooks.
//  - ofq_colsum: db = sum over rows of dY, the autograd of `out += bias` (qlinear.py:71).  HBM-bound, 4 B/elt.
//  - ofq_cga_*: freeze_outside_boundary_weight_idx (cga.py:450-469) and the grad-mask / restore hooks around
//    optimizer.step() (cga.py:962-964, :994-997).  The reference needs two .cpu() syncs per tensor for the
//    loop bounds (cga.py:465); here the global min/max level stays on the device.
#include "common.h"

#define CS_GY 64

__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, float* __restrict__ part,
                                                             int64_t rows, int64_t cols, int64_t ld) {
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int64_t c = (int64_t)blockIdx.x * 64 + cx;
  const int64_t chunk = ceil_div(rows, (int64_t)gridDim.y);
  const int64_t r0 = (int64_t)blockIdx.y * chunk, r1 = min(rows, r0 + chunk);
  float acc = 0.f;
  if (c < cols)
    for (int64_t r = r0 + ry; r < r1; r += 4) acc += x[r * ld + c];
  __shared__ float sh[4][64];
  sh[ry][cx] = acc;
  __syncthreads();
  if (ry == 0 && c < cols) part[(int64_t)blockIdx.y * cols + c] = (sh[0][cx] + sh[1][cx]) + (sh[2][cx] + sh[3][cx]);
}

extern "C" size_t ofq_colsum_ws_bytes(int64_t rows, int64_t cols) { return (size_t)CS_GY * cols * sizeof(float); }

extern "C" int ofq_colsum(const float* x, float* out, int64_t rows, int64_t cols, int64_t ld, void* ws, size_t ws_bytes,
                          ofq_stream_t stream) {
  if (!x || !out || !ws || rows <= 0 || cols <= 0 || ld < cols) return OFQ_EINVAL;
  if (ws_bytes < ofq_colsum_ws_bytes(rows, cols)) return OFQ_ENOWS;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)ceil_div(cols, 64), CS_GY), dim3(256), 0, st, x, (float*)ws,
                     rows, cols, ld);
  OFQ_LAUNCH_CHECK();
  SumJobs jobs = {};
  jobs.j[0] = {(const float*)ws, out, cols, CS_GY, cols, 1, 1.0f, 0, 0};
  strided_sum_launch(jobs, cols, 1, st);
  OFQ_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------- CGA
__device__ __forceinline__ float cga_row_scale(const float* w, int64_t cols, int lane) {
  double acc = 0.0;
  for (int64_t i = lane; i < cols; i += 64) acc += (double)fabsf(w[i]);
  acc = ofq_wave_sum(acc);
  return 2.0f * ofq_div((float)acc, (float)cols);                       // cga.py:462
}
__device__ __forceinline__ float cga_b4(float wv, float s, float n) {
  float c = fminf(fmaxf(ofq_div(wv, s), -1.0f), 1.0f - 1e-6f);          // cga.py:455-456
  return __fsub_rn(__fmul_rn(c, n), 0.5f);                              // cga.py:458
}

__global__ void cga_range_init_kernel(int32_t* range) { range[0] = 0x7fffffff; range[1] = -0x7fffffff; }

__global__ __launch_bounds__(256) void cga_range_kernel(const float* __restrict__ W, int64_t rows, int64_t cols, float n,
                                                        int32_t* range) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* w = W + row * cols;
  const float s = cga_row_scale(w, cols, lane);
  int lo = 0x7fffffff, hi = -0x7fffffff;
  for (int64_t i = lane; i < cols; i += 64) {
    int L = (int)rintf(cga_b4(w[i], s, n));
    lo = min(lo, L); hi = max(hi, L);
  }
  for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o, 64)); hi = max(hi, __shfl_xor(hi, o, 64)); }
  if (lane == 0) { atomicMin(&range[0], lo); atomicMax(&range[1], hi); }
}

__global__ __launch_bounds__(256) void cga_mask_kernel(const float* __restrict__ W, int64_t rows, int64_t cols, float n,
                                                       float th_hi, float th_lo, const int32_t* __restrict__ range,
                                                       float* __restrict__ frozen) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* w = W + row * cols;
  const float s = cga_row_scale(w, cols, lane);
  const int imin = range[0], imax = range[1];
  for (int64_t i = lane; i < cols; i += 64) {
    const float b4 = cga_b4(w[i], s, n);
    float notfrozen = 0.f;
    for (int k = imin; k < imax; ++k) {                                 // np.arange(min, max)  cga.py:465
      const float d = __fsub_rn(b4, (float)k);
      notfrozen += (d <= th_hi && d >= th_lo) ? 1.f : 0.f;              // cga.py:466-467
    }
    frozen[row * cols + i] = 1.0f - notfrozen;                          // cga.py:469
  }
}

extern "C" int ofq_cga_freeze_mask(const float* W, int64_t rows, int64_t cols, int bits, float boundary_range,
                                   float* frozen, int32_t* range_ws, ofq_stream_t stream) {
  if (!W || !frozen || !range_ws || rows <= 0 || cols <= 0 || bits < 1 || bits > 8) return OFQ_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const float n = (float)(1 << (bits - 1));
  hipLaunchKernelGGL(cga_range_init_kernel, dim3(1), dim3(1), 0, st, range_ws);
  hipLaunchKernelGGL(cga_range_kernel, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, st, W, rows, cols, n, range_ws);
  // python-float thresholds compared against an fp32 tensor are rounded to fp32 first
  const float th_hi = (float)(0.5 + (double)boundary_range), th_lo = (float)(0.5 - (double)boundary_range);
  hipLaunchKernelGGL(cga_mask_kernel, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, st, W, rows, cols, n, th_hi,
                     th_lo, range_ws, frozen);
  OFQ_LAUNCH_CHECK();
  return 0;
}

// ---- all CGA tensors of the model in three launches (48 tensors for DeiT QKR: was 3 launches per tensor)
struct CgaTensor { const float* W; float* frozen; int32_t* range; int64_t rows, cols; };
#define CGA_PACK 40
struct CgaPack {
  CgaTensor t[CGA_PACK];
  int32_t first_block[CGA_PACK + 1];      // prefix sums of ceil(rows / 4)
  int32_t n;
};

__device__ __forceinline__ int cga_find(const CgaPack& pk, int block) {
  int ti = 0;
  while (ti + 1 < pk.n && block >= pk.first_block[ti + 1]) ++ti;
  return ti;
}

__global__ void cga_range_init_multi_kernel(CgaPack pk) {
  const int i = threadIdx.x;
  if (i < pk.n) { pk.t[i].range[0] = 0x7fffffff; pk.t[i].range[1] = -0x7fffffff; }
}

__global__ __launch_bounds__(256) void cga_range_multi_kernel(CgaPack pk, float n) {
  const int ti = cga_find(pk, blockIdx.x);
  const CgaTensor t = pk.t[ti];
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)(blockIdx.x - pk.first_block[ti]) * 4 + (threadIdx.x >> 6);
  if (row >= t.rows) return;
  const float* w = t.W + row * t.cols;
  const float s = cga_row_scale(w, t.cols, lane);
  int lo = 0x7fffffff, hi = -0x7fffffff;
  for (int64_t i = lane; i < t.cols; i += 64) {
    int L = (int)rintf(cga_b4(w[i], s, n));
    lo = min(lo, L); hi = max(hi, L);
  }
  for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o, 64)); hi = max(hi, __shfl_xor(hi, o, 64)); }
  if (lane == 0) { atomicMin(&t.range[0], lo); atomicMax(&t.range[1], hi); }
}

__global__ __launch_bounds__(256) void cga_mask_multi_kernel(CgaPack pk, float n, float th_hi, float th_lo) {
  const int ti = cga_find(pk, blockIdx.x);
  const CgaTensor t = pk.t[ti];
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)(blockIdx.x - pk.first_block[ti]) * 4 + (threadIdx.x >> 6);
  if (row >= t.rows) return;
  const float* w = t.W + row * t.cols;
  const float s = cga_row_scale(w, t.cols, lane);
  const int imin = t.range[0], imax = t.range[1];
  for (int64_t i = lane; i < t.cols; i += 64) {
    const float b4 = cga_b4(w[i], s, n);
    float notfrozen = 0.f;
    for (int k = imin; k < imax; ++k) {                                 // np.arange(min, max)  cga.py:465
      const float d = __fsub_rn(b4, (float)k);
      notfrozen += (d <= th_hi && d >= th_lo) ? 1.f : 0.f;              // cga.py:466-467
    }
    t.frozen[row * t.cols + i] = 1.0f - notfrozen;                      // cga.py:469
  }
}

extern "C" int64_t ofq_cga_tensor_entry_bytes(void) { return (int64_t)sizeof(CgaTensor); }

// tensors: HOST array of {const float* W; float* frozen; int32_t* range (2 ints of scratch); int64 rows; int64 cols}
extern "C" int ofq_cga_freeze_mask_multi(const void* tensors, int64_t n_tensors, int bits, float boundary_range,
                                         ofq_stream_t stream) {
  if (!tensors || n_tensors <= 0 || bits < 1 || bits > 8) return OFQ_EINVAL;
  const CgaTensor* ts = (const CgaTensor*)tensors;
  hipStream_t st = (hipStream_t)stream;
  const float n = (float)(1 << (bits - 1));
  const float th_hi = (float)(0.5 + (double)boundary_range), th_lo = (float)(0.5 - (double)boundary_range);
  for (int64_t base = 0; base < n_tensors; base += CGA_PACK) {
    CgaPack pk = {};
    pk.n = (int32_t)((n_tensors - base < CGA_PACK) ? (n_tensors - base) : CGA_PACK);
    int64_t blocks = 0;
    for (int i = 0; i < pk.n; ++i) {
      pk.t[i] = ts[base + i];
      if (!pk.t[i].W || !pk.t[i].frozen || !pk.t[i].range || pk.t[i].rows <= 0 || pk.t[i].cols <= 0) return OFQ_EINVAL;
      pk.first_block[i] = (int32_t)blocks;
      blocks += ceil_div(pk.t[i].rows, 4);
      if (blocks >= (1ll << 31)) return OFQ_EINVAL;
    }
    pk.first_block[pk.n] = (int32_t)blocks;
    hipLaunchKernelGGL(cga_range_init_multi_kernel, dim3(1), dim3(64), 0, st, pk);
    hipLaunchKernelGGL(cga_range_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, st, pk, n);
    hipLaunchKernelGGL(cga_mask_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, st, pk, n, th_hi, th_lo);
    OFQ_LAUNCH_CHECK();
  }
  return 0;
}

__global__ void cga_mask_grad_save_kernel(float* __restrict__ grad, const float* __restrict__ W,
                                          const float* __restrict__ frozen, float* __restrict__ saved, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float f = frozen[i], g = grad[i];
    grad[i] = __fadd_rn(__fmul_rn(__fmul_rn(g, f), 0.0f), __fmul_rn(g, __fsub_rn(1.0f, f)));   // cga.py:962
    saved[i] = __fmul_rn(W[i], f);                                                             // cga.py:964
  }
}
__global__ void cga_restore_kernel(float* __restrict__ W, const float* __restrict__ frozen,
                                   const float* __restrict__ saved, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    W[i] = __fadd_rn(__fmul_rn(W[i], __fsub_rn(1.0f, frozen[i])), saved[i]);                   // cga.py:994-997
}

extern "C" int ofq_cga_mask_grad_save(float* grad, const float* W, const float* frozen, float* saved, int64_t n,
                                      ofq_stream_t stream) {
  if (!grad || !W || !frozen || !saved || n <= 0) return OFQ_EINVAL;
  hipLaunchKernelGGL(cga_mask_grad_save_kernel, dim3((unsigned)min((int64_t)2048, ceil_div(n, 256))), dim3(256), 0,
                     (hipStream_t)stream, grad, W, frozen, saved, n);
  OFQ_LAUNCH_CHECK();
  return 0;
}
extern "C" int ofq_cga_restore(float* W, const float* frozen, const float* saved, int64_t n, ofq_stream_t stream) {
  if (!W || !frozen || !saved || n <= 0) return OFQ_EINVAL;
  hipLaunchKernelGGL(cga_restore_kernel, dim3((unsigned)min((int64_t)2048, ceil_div(n, 256))), dim3(256), 0,
                     (hipStream_t)stream, W, frozen, saved, n);
  OFQ_LAUNCH_CHECK();
  return 0;
}


// exact (erf) GELU, elementwise: the fp32 KD teacher's MLP activation (deit_vision_transformer.py:61, nn.GELU()); the
// quantised student folds its GELU into fc2's quantiser kernels instead
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n,
                                                       unsigned* __restrict__ amax) {
  const int64_t i4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  float m = 0.f;
  if (i4 + 3 < n && ((((uintptr_t)x | (uintptr_t)y) & 15) == 0)) {
    const float4 v = *reinterpret_cast<const float4*>(x + i4);
    const float4 o = make_float4(ofq_gelu(v.x), ofq_gelu(v.y), ofq_gelu(v.z), ofq_gelu(v.w));
    *reinterpret_cast<float4*>(y + i4) = o;
    m = ofq_absmax4(m, o.x, o.y, o.z, o.w);
  } else {
    for (int64_t i = i4; i < n && i < i4 + 4; ++i) { y[i] = ofq_gelu(x[i]); m = fmaxf(m, fabsf(y[i])); }
  }
  if (amax) ofq_amax_publish(amax, m);           // max |y| for a consumer on fp16 planes (the teacher's fc2)
}
extern "C" int ofq_gelu_fwd(const float* x, float* y, int64_t n, void* amax_out, ofq_stream_t stream) {
  if (!x || !y || n <= 0) return OFQ_EINVAL;
  hipLaunchKernelGGL(gelu_fwd_kernel, dim3((unsigned)ceil_div(n, 1024)), dim3(256), 0, (hipStream_t)stream, x, y, n, (unsigned*)amax_out);
  OFQ_LAUNCH_CHECK();
  return 0;
}

// ---- max |x| of a gradient tensor, as the bits of a float in a group of OFQ_AMAX_WORDS device words whose maximum is the result
// (atomic maxima over the bit patterns of |x|: they order like the values, and a maximum does not depend on the order of its
// updates -- deterministic; see ofq_amax_publish).  The two-plane fp16 form of the
// backward GEMMs (csrc/qgemm_planes.hip, split2_f16) takes its power-of-two scale from such a word; the backward kernels that PRODUCE a
// gradient tensor write it as a by-product (their amax_out argument), this kernel serves producers that do not.
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, int64_t rows, int cols4, int64_t ld,
                                                     unsigned* __restrict__ amax) {
  float m = 0.f;
  unsigned nanbits = 0u;
  const int64_t total = rows * cols4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / cols4;
    const int c = (int)(i - r * cols4);
    const float4 v = *reinterpret_cast<const float4*>(x + r * ld + 4 * c);
    m = fmaxf(fmaxf(m, fabsf(v.x)), fmaxf(fabsf(v.y), fmaxf(fabsf(v.z), fabsf(v.w))));
    if (!(v.x == v.x) || !(v.y == v.y) || !(v.z == v.z) || !(v.w == v.w)) nanbits = 0x7fc00000u;      // fmaxf drops NaNs
  }
  unsigned b = __float_as_uint(m) | nanbits;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned other = __shfl_xor(b, o, 64);
    b = other > b ? other : b;
  }
  if ((threadIdx.x & 63) == 0 && b != 0u)
    __hip_atomic_fetch_max(amax + ((blockIdx.x * 4u + (threadIdx.x >> 6)) & (OFQ_AMAX_WORDS - 1)) * OFQ_AMAX_STRIDE, b, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
}
extern "C" int ofq_absmax_f32(const float* x, int64_t rows, int64_t cols, int64_t ld, void* amax, ofq_stream_t stream) {
  if (!x || !amax || rows <= 0 || cols <= 0 || (cols & 3) || (ld & 3) || ld < cols || ((uintptr_t)x & 15) || cols >= (1ll << 31))
    return OFQ_EINVAL;
  const int64_t total = rows * (cols / 4);
  int64_t blocks = ceil_div(total, 256 * 8);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, rows, (int)(cols / 4), ld, (unsigned*)amax);
  OFQ_LAUNCH_CHECK();
  return 0;
}

// ---- effective LSQ steps as a vector: out[i * repeat + r] = (a - a g) + a g, a = max(s[i], 1e-5)  (lsq.py:6-18: the VALUE the
// quantiser kernels divide and multiply by).  The W8A8 patch embedding hands its weight steps (as column scale / k-scale) and its
// per-channel image steps (one per im2col column: repeat = kh * kw) to the code GEMMs; one launch instead of six ATen ones.
__global__ __launch_bounds__(256) void lsq_eff_scale_vec_kernel(const float* __restrict__ s, float gscale, float* __restrict__ out,
                                                                int64_t n, int repeat) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n * repeat) out[i] = ofq_lsq_eff_scale(s[i / repeat], gscale);
}
extern "C" int ofq_lsq_eff_scale_vec(const float* s, float gscale, float* out, int64_t n, int64_t repeat, ofq_stream_t stream) {
  if (!s || !out || n <= 0 || repeat <= 0 || repeat >= (1ll << 30) || n * repeat >= (1ll << 40)) return OFQ_EINVAL;
  hipLaunchKernelGGL(lsq_eff_scale_vec_kernel, dim3((unsigned)ceil_div(n * repeat, 256)), dim3(256), 0, (hipStream_t)stream, s, gscale, out,
                     n, (int)repeat);
  OFQ_LAUNCH_CHECK();
  return 0;
}

// ---- KD loss of the shipped recipes, forward and gradients in one pass ------------------------------------------------------------
// KDLossSoftandHard (src/quantization/utils.py:59-77, `--kd_hard_and_soft 1`):
//   loss = mean_b( -sum_k softmax(t_b)[k] * log_softmax(d_b)[k] ) + mean_b( -log_softmax(c_b)[y_b] )
// with c / d the student's class / distillation logits, t the teacher's logits, y the labels.  One wave per row: three
// (max, sum exp) pairs, the two row losses, and the gradients of the MEAN loss with respect to c and d as autograd forms them:
//   dc[k] = (softmax(c)[k] - [k == y]) / B          dd[k] = (softmax(d)[k] * sum_j p_t[j] - p_t[k]) / B.
// The row losses go to `rows` [2][B]; kd_loss_reduce_kernel adds them in index order (one thread: B is the batch) -- fixed order,
// no atomics.  Replaces ~20 ATen launches per step (two log_softmax, softmax, nll, three divisions, sums, their backwards).
// Labels follow nn.CrossEntropyLoss (the reference's hard-label term, utils.py:70): a row with target == -100 (ignore_index)
// contributes neither loss nor gradient and the hard term is the mean over the OTHER rows (rows[2B] = B / count rescales dc
// in the backward; count == 0 gives NaN, as the stock op does); any other label outside [0, K) -- where the stock op traps
// with a device assert -- makes the loss NaN.
__global__ __launch_bounds__(256) void kd_loss_rows_kernel(const float* __restrict__ c, const float* __restrict__ d,
                                                           const float* __restrict__ t, const int64_t* __restrict__ y,
                                                           float* __restrict__ dc, float* __restrict__ dd, float* __restrict__ rows,
                                                           int B, int K, int64_t ldc, int64_t ldd, int64_t ldt) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const float* cr = c + (int64_t)b * ldc;
  const float* dr = d + (int64_t)b * ldd;
  const float* tr = t + (int64_t)b * ldt;
  float mc = -INFINITY, md = -INFINITY, mt = -INFINITY;
  for (int k = lane; k < K; k += 64) {
    mc = fmaxf(mc, cr[k]);
    md = fmaxf(md, dr[k]);
    mt = fmaxf(mt, tr[k]);
  }
  mc = ofq_wave_max(mc); md = ofq_wave_max(md); mt = ofq_wave_max(mt);
  float zc = 0.f, zd = 0.f, zt = 0.f;
  for (int k = lane; k < K; k += 64) {
    zc += expf(cr[k] - mc);
    zd += expf(dr[k] - md);
    zt += expf(tr[k] - mt);
  }
  zc = ofq_wave_sum(zc); zd = ofq_wave_sum(zd); zt = ofq_wave_sum(zt);
  const float lzc = logf(zc), lzd = logf(zd);
  const int yb = (int)y[b];
  const bool ignored = y[b] == -100;              // nn.CrossEntropyLoss's ignore_index
  float soft = 0.f, psum = 0.f;
  for (int k = lane; k < K; k += 64) {
    const float p = expf(tr[k] - mt) / zt;
    soft += p * ((dr[k] - md) - lzd);
    psum += p;
  }
  soft = ofq_wave_sum(soft);
  psum = ofq_wave_sum(psum);
  const float invB = 1.f / (float)B;
  for (int k = lane; k < K; k += 64) {
    const float p = expf(tr[k] - mt) / zt;
    dd[(int64_t)b * K + k] = (expf((dr[k] - md) - lzd) * psum - p) * invB;
    dc[(int64_t)b * K + k] = ignored ? 0.f : (expf((cr[k] - mc) - lzc) - (k == yb ? 1.f : 0.f)) * invB;
  }
  if (lane == 0) {
    rows[b] = -soft;
    rows[B + b] = ignored ? 0.f : ((yb >= 0 && yb < K) ? -((cr[yb] - mc) - lzc) : NAN);
  }
}
__global__ void kd_loss_reduce_kernel(float* __restrict__ rows, const int64_t* __restrict__ y, float* __restrict__ loss, int B) {
  float s = 0.f, h = 0.f;
  int count = 0;
  for (int b = 0; b < B; ++b) {
    s += rows[b];
    h += rows[B + b];
    count += y[b] != -100 ? 1 : 0;
  }
  loss[0] = s / (float)B + h / (float)count;
  rows[2 * B] = (float)B / (float)count;          // 1.0 exactly when no row is ignored
}
// g (device scalar: the gradient arriving at the loss) times the two saved gradients; cls_scale[0] = B / (rows not ignored)
__global__ __launch_bounds__(256) void kd_loss_scale_kernel(const float* __restrict__ g, const float* __restrict__ dc,
                                                            const float* __restrict__ dd, const float* __restrict__ cls_scale,
                                                            float* __restrict__ oc, float* __restrict__ od, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const float gv = g[0];
    const float v = dc[i] * gv;
    oc[i] = cls_scale ? v * cls_scale[0] : v;
    od[i] = dd[i] * gv;
  }
}
extern "C" int ofq_kd_loss_fwd(const float* cls_logits, const float* dist_logits, const float* teacher_logits, const int64_t* target,
                               float* loss, float* dcls, float* ddist, float* row_ws, int64_t B, int64_t K, int64_t ld_cls,
                               int64_t ld_dist, int64_t ld_teacher, ofq_stream_t stream) {
  if (!cls_logits || !dist_logits || !teacher_logits || !target || !loss || !dcls || !ddist || !row_ws || B <= 0 || K <= 0 ||
      B >= (1ll << 24) || K >= (1ll << 24))
    return OFQ_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(kd_loss_rows_kernel, dim3((unsigned)ceil_div(B, 4)), dim3(256), 0, st, cls_logits, dist_logits, teacher_logits,
                     target, dcls, ddist, row_ws, (int)B, (int)K, ld_cls, ld_dist, ld_teacher);
  OFQ_LAUNCH_CHECK();
  hipLaunchKernelGGL(kd_loss_reduce_kernel, dim3(1), dim3(1), 0, st, row_ws, target, loss, (int)B);
  OFQ_LAUNCH_CHECK();
  return 0;
}
extern "C" int ofq_kd_loss_bwd(const float* grad_loss, const float* dcls, const float* ddist, const float* cls_scale, float* out_cls,
                               float* out_dist, int64_t n, ofq_stream_t stream) {
  if (!grad_loss || !dcls || !ddist || !out_cls || !out_dist || n <= 0) return OFQ_EINVAL;
  hipLaunchKernelGGL(kd_loss_scale_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, grad_loss, dcls, ddist,
                     cls_scale, out_cls, out_dist, n);
  OFQ_LAUNCH_CHECK();
  return 0;
}

// ---- token assembly of the distilled ViT (deit.py:32-44 / deit_vision_transformer.py: cat(cls, dist, patches) + pos_embed) --------
// out[b][t][:] = (t == 0 ? cls : t == 1 && dist ? dist : patches[b][t - ntok][:]) + pos[t][:]     one pass instead of a cat and an add;
// the backward is a view (patches), and ONE column sum over the batch for pos_embed whose first rows are the class /
// distillation tokens' gradients (ofq_colsum on the (B, T*C) view).
__global__ __launch_bounds__(256) void assemble_tokens_kernel(const float* __restrict__ patches, const float* __restrict__ tok0,
                                                              const float* __restrict__ tok1, const float* __restrict__ pos,
                                                              float* __restrict__ out, int64_t total4, int T, int C4, int ntok) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const int c4 = (int)(i % C4);
  const int64_t bt = i / C4;
  const int t = (int)(bt % T);
  const int64_t b = bt / T;
  const float4 p = reinterpret_cast<const float4*>(pos)[(int64_t)t * C4 + c4];
  const float4 v = t < ntok ? reinterpret_cast<const float4*>(t == 0 ? tok0 : tok1)[c4]
                            : reinterpret_cast<const float4*>(patches)[(b * (T - ntok) + (t - ntok)) * C4 + c4];
  reinterpret_cast<float4*>(out)[i] = make_float4(v.x + p.x, v.y + p.y, v.z + p.z, v.w + p.w);
}
extern "C" int ofq_assemble_tokens(const float* patches, const float* cls_token, const float* dist_token, const float* pos, float* out,
                                   int64_t B, int64_t T, int64_t C, ofq_stream_t stream) {
  const int ntok = dist_token ? 2 : 1;
  if (!patches || !cls_token || !pos || !out || B <= 0 || T <= ntok || C <= 0 || (C & 3) ||
      ((((uintptr_t)patches | (uintptr_t)cls_token | (uintptr_t)dist_token | (uintptr_t)pos | (uintptr_t)out)) & 15))
    return OFQ_EINVAL;
  const int64_t total4 = B * T * (C / 4);
  hipLaunchKernelGGL(assemble_tokens_kernel, dim3((unsigned)ceil_div(total4, 256)), dim3(256), 0, (hipStream_t)stream, patches, cls_token,
                     dist_token, pos, out, total4, (int)T, (int)(C / 4), ntok);
  OFQ_LAUNCH_CHECK();
  return 0;
}

// y[b][i][:] = x[b][idx[i]][:]: the token permutation of Swin's shifted-window partition / reverse (swin.py:103-131
// pad-free case: roll + view + permute + reshape collapse into one row gather; the backward is the gather with the inverse
// permutation).  HBM-bound, 8 B/elt; one float4 per thread, C % 4 == 0.
__global__ __launch_bounds__(256) void permute_tokens_kernel(const float* __restrict__ x, const int32_t* __restrict__ idx,
                                                             float* __restrict__ y, int64_t total4, int N, int C4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const int64_t row = i / C4;
  const int c = (int)(i - row * C4);
  const int64_t b = row / N;
  const int t = (int)(row - b * N);
  const float4 v = *reinterpret_cast<const float4*>(x + ((b * N + idx[t]) * C4 + c) * 4);
  *reinterpret_cast<float4*>(y + i * 4) = v;
}
extern "C" int ofq_permute_tokens(const float* x, const int32_t* idx, float* y, int64_t B, int64_t N, int64_t C,
                                  ofq_stream_t stream) {
  if (!x || !idx || !y || x == y || B <= 0 || N <= 0 || C <= 0 || (C & 3) || N >= (1ll << 31) || C >= (1ll << 31) ||
      !al16(x) || !al16(y))
    return OFQ_EINVAL;
  const int64_t total4 = B * N * (C / 4);
  if (ceil_div(total4, 256) >= (1ll << 31)) return OFQ_EINVAL;
  hipLaunchKernelGGL(permute_tokens_kernel, dim3((unsigned)ceil_div(total4, 256)), dim3(256), 0, (hipStream_t)stream, x, idx,
                     y, total4, (int)N, (int)(C / 4));
  OFQ_LAUNCH_CHECK();
  return 0;
}

// Deferred second-stage sums (common.h: SumDeferState).  ofq_sum_defer(1) ... ofq_sum_defer(0) brackets the calls whose
// second stage may wait; ofq_sum_flush launches everything queued so far, up to OFQ_SUM_MULTI jobs per launch and lane
// layout.  Host state is per process (one process per GPU, launches serialised on one stream).
extern "C" void ofq_sum_defer(int on) {
  SumDeferState& ds = sum_defer_state();
  if (on < 0) {            // a backward pass that failed: forget what was queued (its outputs may be gone already)
    ds.pend.clear();
    ds.on = false;
    return;
  }
  ds.on = on != 0;
}
extern "C" int ofq_sum_pending(void) { return (int)sum_defer_state().pend.size(); }
extern "C" int ofq_sum_flush(ofq_stream_t stream) {
  SumDeferState& ds = sum_defer_state();
  hipStream_t st = (hipStream_t)stream;
  const int cpbs[3] = {4, 64, 16};
  for (int v = 0; v < 3; ++v) {
    SumJobsMulti m = {};
    int n = 0;
    int64_t tiles = 0;
    auto launch = [&]() {
      if (n == 0) return;
      const dim3 grid((unsigned)tiles, (unsigned)n), block(1024);
      if (cpbs[v] == 4) hipLaunchKernelGGL(strided_sum_multi_kernel_t<4>, grid, block, 0, st, m);
      else if (cpbs[v] == 64) hipLaunchKernelGGL(strided_sum_multi_kernel_t<64>, grid, block, 0, st, m);
      else hipLaunchKernelGGL(strided_sum_multi_kernel_t<16>, grid, block, 0, st, m);
      n = 0;
      tiles = 0;
    };
    for (const SumPending& p : ds.pend) {
      if (p.cpb != cpbs[v]) continue;
      m.j[n++] = p.job;
      const int64_t t = ceil_div(p.job.ncols, cpbs[v]);
      if (t > tiles) tiles = t;
      if (n == OFQ_SUM_MULTI) launch();
    }
    launch();
  }
  ds.pend.clear();
  OFQ_LAUNCH_CHECK();
  return 0;
}
