// K10  scale + softmax + unsigned LSQ over attention rows (reference attention.py:96-99, :213-216 and
// lsq.py:571-602 with all_positive=True).  One wave per row (n <= 256: four strided elements per lane,
// consecutive lanes on consecutive addresses), 4 rows per workgroup; max / sum / dot products are wave
// butterflies.  fwd: 4 B read + 8 B written per score (prob is kept for backward); bwd: 8 B read + 4 B
// written.  Pad columns [n, ld) are written as zeros so the P*V GEMM may read whole float4 groups.
#include "common.h"

#define SM_MAXE 4

__global__ __launch_bounds__(256) void softmax_lsq_fwd_kernel(const float* __restrict__ sc, const float* __restrict__ s,
                                                              float* __restrict__ prob, float* __restrict__ y,
                                                              int64_t rows, int n, int64_t ld, int64_t S, float alpha,
                                                              float hi, float gscale, unsigned char* __restrict__ codes,
                                                              float* __restrict__ code_rowsum, const float* __restrict__ addend,
                                                              int64_t add_period) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* src = sc + r * ld;
  float t[SM_MAXE];
  float m = -INFINITY;
#pragma unroll
  for (int e = 0; e < SM_MAXE; ++e) {
    int c = lane + 64 * e;
    t[e] = (c < n) ? __fmul_rn(src[c], alpha) : -INFINITY;
    // Swin: + relative-position bias (+ shift mask), one [n][ld] slab per (window, head), period = windows*heads
    if (addend && c < n) t[e] = __fadd_rn(t[e], addend[(((r / S) % add_period) * S + (r % S)) * ld + c]);
    m = fmaxf(m, t[e]);
  }
  m = ofq_wave_max(m);
  float sum = 0.f;
#pragma unroll
  for (int e = 0; e < SM_MAXE; ++e) {
    int c = lane + 64 * e;
    t[e] = (c < n) ? expf(t[e] - m) : 0.f;
    sum += t[e];
  }
  sum = ofq_wave_sum(sum);
  const float a = ofq_lsq_eff_scale(s[r % S], gscale);
  float qsum = 0.f;
#pragma unroll
  for (int e = 0; e < SM_MAXE; ++e) {
    int c = lane + 64 * e;
    if (c < ld) {
      float p = 0.f, out = 0.f, q = 0.f;
      if (c < n) {
        p = ofq_div(t[e], sum);
        float v;
        float yi = ofq_lsq_quant(p, a, 0.f, hi, q, v);
        out = __fmul_rn(yi, a);
      }
      prob[r * ld + c] = p;
      if (y) y[r * ld + c] = out;
      if (codes) codes[r * ld + c] = (unsigned char)(int)q;
      qsum += q;
    }
  }
  if (code_rowsum) {
    qsum = ofq_wave_sum(qsum);
    if (lane == 0) code_rowsum[r] = qsum;
  }
}

__global__ __launch_bounds__(256) void softmax_lsq_bwd_kernel(const float* __restrict__ g, const float* __restrict__ prob,
                                                              const float* __restrict__ s, float* __restrict__ dsc,
                                                              float* __restrict__ rowpart, int64_t rows, int n,
                                                              int64_t ld, int64_t S, float alpha, float hi,
                                                              float gscale, float* __restrict__ ds_rowsum,
                                                              unsigned* __restrict__ amax) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float omax = 0.f;
  const float a = ofq_lsq_eff_scale(s[r % S], gscale);
  float p[SM_MAXE], dq[SM_MAXE];
  float rowds = 0.f, dot = 0.f;
#pragma unroll
  for (int e = 0; e < SM_MAXE; ++e) {
    int c = lane + 64 * e;
    p[e] = 0.f; dq[e] = 0.f;
    if (c < n) {
      p[e] = prob[r * ld + c];
      float ge = g[r * ld + c];
      float q, v;
      ofq_lsq_quant(p[e], a, 0.f, hi, q, v);
      bool inr = (v >= 0.f) && (v <= hi);
      dq[e] = inr ? ofq_div(__fmul_rn(ge, a), a) : 0.f;
      rowds += ge * (inr ? (q - v) : q);
      dot += dq[e] * p[e];
    }
  }
  rowds = ofq_wave_sum(rowds);
  dot = ofq_wave_sum(dot);
  if (lane == 0) rowpart[r] = rowds;
  float rsum = 0.f;
#pragma unroll
  for (int e = 0; e < SM_MAXE; ++e) {
    int c = lane + 64 * e;
    const float o = (c < n) ? (dq[e] - dot) * p[e] * alpha : 0.f;
    if (c < ld) dsc[r * ld + c] = o;
    rsum += o;
    omax = fmaxf(omax, fabsf(o));
  }
  if (amax) ofq_amax_publish(amax, omax);
  if (ds_rowsum) {       // ~0 in exact arithmetic; kept so that the offset term of dx_hat matches the reference's fp32 value
    rsum = ofq_wave_sum(rsum);
    if (lane == 0) ds_rowsum[r] = rsum;
  }
}

// ---- float4 variants (ld % 4 == 0, ld <= 256): lane l owns columns 4l..4l+3, so a row is one 16-byte load / store per
// lane and tensor instead of four strided dwords; same arithmetic per element, same wave reductions
#define SM_RPW 4      // row slots per lane group: their loads are all issued before the first row is processed

template <int W>
__device__ __forceinline__ float sm_group_max(float v) {
#pragma unroll
  for (int o = W / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// LPR lanes own one row (4 columns per lane): 64 for attention rows up to 256 wide (DeiT: 198), 16 for rows up to 64
// wide (Swin's 7x7 windows: 49), where a whole wave per row would leave 3/4 of the lanes idle.
template <int LPR>
__global__ __launch_bounds__(256) void softmax_lsq_fwd_v4_kernel(const float* __restrict__ sc, const float* __restrict__ s,
                                                                 float* __restrict__ prob, float* __restrict__ y,
                                                                 int64_t rows, int n, int64_t ld, int64_t S, float alpha,
                                                                 float hi, float gscale, unsigned char* __restrict__ codes,
                                                                 float* __restrict__ code_rowsum, const float* __restrict__ addend,
                                                                 int64_t add_period) {
  constexpr int GPB = 256 / LPR;                      // lane groups per workgroup
  const int lane = threadIdx.x % LPR;
  const int64_t r0 = ((int64_t)blockIdx.x * GPB + threadIdx.x / LPR) * SM_RPW;
  const int c0 = lane * 4;
  const bool act = c0 < ld;
  float4 vin[SM_RPW], ain[SM_RPW];
#pragma unroll
  for (int i = 0; i < SM_RPW; ++i) {
    const int64_t r = min(r0 + i, rows - 1);
    vin[i] = ain[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (act) {
      vin[i] = *reinterpret_cast<const float4*>(sc + r * ld + c0);
      if (addend) ain[i] = *reinterpret_cast<const float4*>(addend + (((r / S) % add_period) * S + (r % S)) * ld + c0);
    }
  }
#pragma unroll
  for (int i = 0; i < SM_RPW; ++i) {
    const int64_t r = r0 + i;
    const bool rok = r < rows;                        // (no early exit: the group shuffles need every lane)
    float t[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    const float vv[4] = {vin[i].x, vin[i].y, vin[i].z, vin[i].w}, aa[4] = {ain[i].x, ain[i].y, ain[i].z, ain[i].w};
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (c0 + e < n) {
        t[e] = __fmul_rn(vv[e], alpha);
        if (addend) t[e] = __fadd_rn(t[e], aa[e]);
      }
    float m = fmaxf(fmaxf(t[0], t[1]), fmaxf(t[2], t[3]));
    m = sm_group_max<LPR>(m);
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      t[e] = (c0 + e < n) ? expf(t[e] - m) : 0.f;
      sum += t[e];
    }
    sum = ofq_group_sum<LPR>(sum);
    const float a = ofq_lsq_eff_scale(s[min(r, rows - 1) % S], gscale);
    float qsum = 0.f;
    if (act && rok) {
      float p[4] = {0.f, 0.f, 0.f, 0.f}, out[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (c0 + e < n) {
          p[e] = ofq_div(t[e], sum);
          float v;
          const float yi = ofq_lsq_quant(p[e], a, 0.f, hi, q[e], v);
          out[e] = __fmul_rn(yi, a);
          qsum += q[e];
        }
      *reinterpret_cast<float4*>(prob + r * ld + c0) = make_float4(p[0], p[1], p[2], p[3]);
      if (y) *reinterpret_cast<float4*>(y + r * ld + c0) = make_float4(out[0], out[1], out[2], out[3]);
      if (codes)
        *reinterpret_cast<uchar4*>(codes + r * ld + c0) =
            make_uchar4((unsigned char)(int)q[0], (unsigned char)(int)q[1], (unsigned char)(int)q[2], (unsigned char)(int)q[3]);
    }
    if (code_rowsum) {
      qsum = ofq_group_sum<LPR>(qsum);
      if (lane == 0 && rok) code_rowsum[r] = qsum;
    }
  }
}

template <int LPR>
__global__ __launch_bounds__(256) void softmax_lsq_bwd_v4_kernel(const float* __restrict__ g, const float* __restrict__ prob,
                                                                 const float* __restrict__ s, float* __restrict__ dsc,
                                                                 float* __restrict__ rowpart, int64_t rows, int n,
                                                                 int64_t ld, int64_t S, float alpha, float hi,
                                                                 float gscale, float* __restrict__ ds_rowsum,
                                                                 unsigned* __restrict__ amax) {
  constexpr int GPB = 256 / LPR;
  float omax = 0.f;
  const int lane = threadIdx.x % LPR;
  const int64_t r0 = ((int64_t)blockIdx.x * GPB + threadIdx.x / LPR) * SM_RPW;
  const int c0 = lane * 4;
  const bool act = c0 < ld;
  float4 pin[SM_RPW], gin[SM_RPW];
#pragma unroll
  for (int i = 0; i < SM_RPW; ++i) {
    const int64_t r = min(r0 + i, rows - 1);
    pin[i] = gin[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (act) {
      pin[i] = *reinterpret_cast<const float4*>(prob + r * ld + c0);
      gin[i] = *reinterpret_cast<const float4*>(g + r * ld + c0);
    }
  }
#pragma unroll
  for (int i = 0; i < SM_RPW; ++i) {
    const int64_t r = r0 + i;
    const bool rok = r < rows;
    const float a = ofq_lsq_eff_scale(s[min(r, rows - 1) % S], gscale);
    float p[4] = {0.f, 0.f, 0.f, 0.f}, dq[4] = {0.f, 0.f, 0.f, 0.f};
    float rowds = 0.f, dot = 0.f;
    const float pp[4] = {pin[i].x, pin[i].y, pin[i].z, pin[i].w}, gg[4] = {gin[i].x, gin[i].y, gin[i].z, gin[i].w};
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (act && c0 + e < n) {
        p[e] = pp[e];
        float q, v;
        ofq_lsq_quant(p[e], a, 0.f, hi, q, v);
        const bool inr = (v >= 0.f) && (v <= hi);
        dq[e] = inr ? ofq_div(__fmul_rn(gg[e], a), a) : 0.f;
        rowds += gg[e] * (inr ? (q - v) : q);
        dot += dq[e] * p[e];
      }
    rowds = ofq_group_sum<LPR>(rowds);
    dot = ofq_group_sum<LPR>(dot);
    if (lane == 0 && rok) rowpart[r] = rowds;
    float rsum = 0.f;
    if (act && rok) {
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = (c0 + e < n) ? (dq[e] - dot) * p[e] * alpha : 0.f;
        rsum += o[e];
      }
      *reinterpret_cast<float4*>(dsc + r * ld + c0) = make_float4(o[0], o[1], o[2], o[3]);
      omax = ofq_absmax4(omax, o[0], o[1], o[2], o[3]);
    }
    if (ds_rowsum) {
      rsum = ofq_group_sum<LPR>(rsum);
      if (lane == 0 && rok) ds_rowsum[r] = rsum;
    }
  }
  if (amax) ofq_amax_publish(amax, omax);
}

extern "C" int ofq_softmax_lsq_fwd(const float* scores, const float* s, float* prob, float* y, int64_t rows, int64_t n,
                                   int64_t ld, int64_t S, float alpha, int hi, float gscale, uint8_t* codes,
                                   float* code_rowsum, const float* addend, int64_t add_period, ofq_stream_t stream) {
  // (y and codes may both be NULL: the probabilities alone -- the fp32 teacher's softmax, ofq_amd/teacher.py)
  if (!scores || !s || !prob || rows <= 0 || n <= 0 || n > 64 * SM_MAXE || ld < n || ld > 64 * SM_MAXE || S <= 0)
    return OFQ_EINVAL;
  const bool v4 = (ld & 3) == 0 && ((uintptr_t)scores & 15) == 0 && ((uintptr_t)prob & 15) == 0 && (!y || ((uintptr_t)y & 15) == 0) &&
                  (!codes || ((uintptr_t)codes & 3) == 0) && (!addend || ((uintptr_t)addend & 15) == 0);
  if (v4)
  {
    if (ld <= 64)
      hipLaunchKernelGGL(softmax_lsq_fwd_v4_kernel<16>, dim3((unsigned)ceil_div(rows, 16 * SM_RPW)), dim3(256), 0,
                         (hipStream_t)stream, scores, s, prob, y, rows, (int)n, ld, S, alpha, (float)hi, gscale, codes,
                         code_rowsum, addend, add_period > 0 ? add_period : 1);
    else
      hipLaunchKernelGGL(softmax_lsq_fwd_v4_kernel<64>, dim3((unsigned)ceil_div(rows, 4 * SM_RPW)), dim3(256), 0,
                         (hipStream_t)stream, scores, s, prob, y, rows, (int)n, ld, S, alpha, (float)hi, gscale, codes,
                         code_rowsum, addend, add_period > 0 ? add_period : 1);
  }
  else
    hipLaunchKernelGGL(softmax_lsq_fwd_kernel, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                       scores, s, prob, y, rows, (int)n, ld, S, alpha, (float)hi, gscale, codes, code_rowsum, addend,
                       add_period > 0 ? add_period : 1);
  OFQ_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t ofq_softmax_lsq_bwd_ws_bytes(int64_t rows) { return (size_t)rows * sizeof(float) + 256; }

extern "C" int ofq_softmax_lsq_bwd(const float* g, const float* prob, const float* s, float* dscores, float* ds,
                                   int64_t rows, int64_t n, int64_t ld, int64_t S, float alpha, int hi, float gscale,
                                   float* ds_rowsum, void* ws, size_t ws_bytes, void* amax_out, ofq_stream_t stream) {
  if (!g || !prob || !s || !dscores || !ws || rows <= 0 || n <= 0 || n > 64 * SM_MAXE || ld < n || ld > 64 * SM_MAXE ||
      S <= 0 || rows % S)
    return OFQ_EINVAL;
  if (ws_bytes < (size_t)rows * sizeof(float)) return OFQ_ENOWS;
  hipStream_t st = (hipStream_t)stream;
  if ((ld & 3) == 0 && (((uintptr_t)g | (uintptr_t)prob | (uintptr_t)dscores) & 15) == 0)
  {
    if (ld <= 64)
      hipLaunchKernelGGL(softmax_lsq_bwd_v4_kernel<16>, dim3((unsigned)ceil_div(rows, 16 * SM_RPW)), dim3(256), 0, st, g, prob, s,
                         dscores, (float*)ws, rows, (int)n, ld, S, alpha, (float)hi, gscale, ds_rowsum, (unsigned*)amax_out);
    else
      hipLaunchKernelGGL(softmax_lsq_bwd_v4_kernel<64>, dim3((unsigned)ceil_div(rows, 4 * SM_RPW)), dim3(256), 0, st, g, prob, s,
                         dscores, (float*)ws, rows, (int)n, ld, S, alpha, (float)hi, gscale, ds_rowsum, (unsigned*)amax_out);
  }
  else
    hipLaunchKernelGGL(softmax_lsq_bwd_kernel, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, st, g, prob, s, dscores,
                       (float*)ws, rows, (int)n, ld, S, alpha, (float)hi, gscale, ds_rowsum, (unsigned*)amax_out);
  OFQ_LAUNCH_CHECK();
  if (ds) {
    SumJobs jobs = {};
    jobs.j[0] = {(const float*)ws, ds, S, rows / S, S, 1, gscale, 0, 0};
    strided_sum_launch(jobs, S, 1, st);
    OFQ_LAUNCH_CHECK();
  }
  return 0;
}
