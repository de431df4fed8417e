// Shared device helpers for the OFQ gfx950 kernels.  Wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "../../include/ofq_hip.h"
#include <vector>

#define OFQ_WAVE 64

#define OFQ_LAUNCH_CHECK()                      \
  do {                                          \
    hipError_t e__ = hipGetLastError();         \
    if (e__ != hipSuccess) return (int)e__;     \
  } while (0)

// fp32 ops that must keep the reference's rounding sequence are written with the _rn intrinsics so
// that no fma contraction can merge them (the library is also built with -ffp-contract=off).
#ifdef OFQ_EXPERIMENT_APPROX_DIV      // tools/probe only: is the kernel bound by the IEEE division sequence?
__device__ __forceinline__ float ofq_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
#else
__device__ __forceinline__ float ofq_div(float a, float b) { return __fdiv_rn(a, b); }
#endif

// Effective LSQ scale value: clip(s,1e-5) then grad_scale():  (a - a*g) + a*g   (lsq.py:6-18, :593)
__device__ __forceinline__ float ofq_lsq_eff_scale(float s, float g) {
  float a = (s > 1e-5f) ? s : 1e-5f;
  float t = __fmul_rn(a, g);
  return __fadd_rn(__fsub_rn(a, t), t);
}

// One LSQ element.  Returns y_int = (q - u) + u (round_pass value, lsq.py:11-14); q is the level.
__device__ __forceinline__ float ofq_lsq_quant(float xin, float a, float lo, float hi, float& q, float& v) {
  v = ofq_div(xin, a);
  float u = fminf(fmaxf(v, lo), hi);
  q = rintf(u);  // RNE == torch.round
  return __fadd_rn(__fsub_rn(q, u), u);
}

// The level q = rint(clamp(x / a, lo, hi)) of ofq_lsq_quant without the IEEE division sequence in the common case.
// `ra` is the correctly rounded 1 / a, so x * ra is within 1.5 ulp of the correctly rounded x / a, and the two round to
// the same level unless the clamped product lies within `tol` = 8 ulp(max(|lo|, |hi|) + 1) of a half-integer.  Callers
// accumulate the distance to the nearest integer over a group of elements and redo the group with ofq_lsq_level_exact
// when any lane of the wave comes that close (rare: ~1e-5 per element at 2-4 bits), so the levels stay bit-identical at
// ~6 instead of ~16 VALU each.
__device__ __forceinline__ float ofq_lsq_level_tol(float lo, float hi) {
  const float m = fmaxf(fabsf(lo), fabsf(hi)) + 1.f;
  return 8.f * (m * 1.1920929e-7f);                   // 8 * ulp-ish (m * 2^-23 >= ulp(m))
}
// The group's flag is a running maximum of |u - q| (two VALU instructions per element; a bool OR-ed per element costs
// five: compare, select, shift, or, bit-op).  The caller tests !(dmax < half_m_tol) once per group.  A NaN takes the
// fast level, which is the exact form's too (v_med3_f32 and fmaxf both fall to `lo`).
__device__ __forceinline__ float ofq_lsq_level_rcp_d(float x, float ra, float lo, float hi, float& dmax) {
  const float u = __builtin_amdgcn_fmed3f(__fmul_rn(x, ra), lo, hi);
  const float q = rintf(u);
  dmax = fmaxf(dmax, fabsf(__fsub_rn(u, q)));
  return q;
}
__device__ __forceinline__ float ofq_lsq_level_exact(float x, float a, float lo, float hi) {
  return rintf(fminf(fmaxf(ofq_div(x, a), lo), hi));
}

// Correctly rounded x / d from the correctly rounded reciprocal rd = fl(1 / d): q0 = fl(x * rd), then two residual
// corrections q <- fma(fma(-d, q, x), rd, q) -- the tail of the hardware's own division expansion with an exact reciprocal
// in place of its refined estimate.  Valid while nothing under- or overflows (callers' operands are O(1e-30 .. 1e30)).
__device__ __forceinline__ float ofq_div_by_rcp(float x, float d, float rd) {
  float q = __fmul_rn(x, rd);
  q = __fmaf_rn(__fmaf_rn(-d, q, x), rd, q);
  q = __fmaf_rn(__fmaf_rn(-d, q, x), rd, q);
  return q;
}

// One element of the LSQ backward (lsq.py:593-601 under autograd): v = xin / a, q = rint(clamp(v)),
// dq = in_range ? (g * a) / a : 0 (the reference's own operation order), dsc = g * (in_range ? q - v : q).
__device__ __forceinline__ void ofq_lsq_bwd_exact(float xin, float g, float a, float lo, float hi, float& dq, float& dsc) {
  float q, v;
  ofq_lsq_quant(xin, a, lo, hi, q, v);
  const bool inr = (v >= lo) && (v <= hi);
  dq = inr ? ofq_div(__fmul_rn(g, a), a) : 0.f;
  dsc = g * (inr ? (q - v) : q);
}
// The same without the two IEEE division sequences (~11 instructions each).  `ra` is the correctly rounded 1 / a.
//  * v' = xin * ra is within 1.5 ulp of fl(xin / a): the level q and the range decision are the exact ones unless the
//    clamped product lies within `tol` of a half-integer / v' within `tol` of lo or hi -> `risky`, the caller redoes the
//    group with ofq_lsq_bwd_exact.  (q - v') differs from (q - v) by <= 1.5 ulp(v): it only enters the step gradient, a
//    sum over >= 10^4 elements compared at 1e-5.)
//  * (g * a) / a: t = fl(g * a); q0 = fl(t * ra); two residual corrections q <- fma(fma(-a, q, t), ra, q) -- the tail of
//    the hardware's own division expansion with an exact reciprocal in place of its refined estimate -- give the correctly
//    rounded quotient whenever nothing under- or overflows; |t| outside [2^-100, 2^100] is flagged risky as well.
// The exactness conditions are accumulated over a group as extrema (OfqLsqFlags, ~9 VALU per element, no per-element
// booleans) and tested once per group with ofq_lsq_flags_risky.
struct OfqLsqFlags {
  float dmax = 0.f;                       // max |u - rint(u)|
  float emin = 3.0e38f;                   // min distance of v' to lo / hi
  unsigned umin = 0xffffffffu;            // min over non-zero |t| of bits(|t|) - 1   (zero wraps to the top)
  unsigned umax = 0u;                     // max bits(|t|)   (inf / NaN on top)
};
__device__ __forceinline__ bool ofq_lsq_flags_risky(const OfqLsqFlags& f, float half_m_tol, float tol) {
  // |t| must lie strictly inside (2^-100, 2^100) unless it is zero
  return !(f.dmax < half_m_tol) | (f.emin < tol) | (f.umin < 0x0D800000u) | (f.umax >= 0x71800000u);
}
__device__ __forceinline__ void ofq_lsq_bwd_fast(float xin, float g, float a, float ra, float lo, float hi, OfqLsqFlags& f,
                                                 float& dq, float& dsc) {
  const float v = __fmul_rn(xin, ra);
  const float u = __builtin_amdgcn_fmed3f(v, lo, hi);
  const float q = rintf(u);
  const bool inr = (v >= lo) && (v <= hi);
  f.dmax = fmaxf(f.dmax, fabsf(__fsub_rn(u, q)));
  f.emin = fminf(f.emin, fminf(fabsf(__fsub_rn(v, lo)), fabsf(__fsub_rn(v, hi))));
  const float t = __fmul_rn(g, a);
  float qq = __fmul_rn(t, ra);
  qq = __fmaf_rn(__fmaf_rn(-a, qq, t), ra, qq);
  qq = __fmaf_rn(__fmaf_rn(-a, qq, t), ra, qq);
  const unsigned tb = __float_as_uint(t) & 0x7fffffffu;
  f.umin = min(f.umin, tb - 1u);
  f.umax = max(f.umax, tb);
  dq = inr ? qq : 0.f;
  dsc = g * (inr ? (q - v) : q);
}

__device__ __forceinline__ float ofq_gelu(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
// gelu(x) to within OFQ_GELU_FAST_EPS (absolute) of ofq_gelu(x), for deciding a quantisation LEVEL only: Abramowitz-Stegun
// 7.1.26 (|erf error| <= 1.5e-7) on v_rcp_f32 / v_exp_f32, ~17 VALU instead of the ~50 of erff's two divergent branches.
// Measured against the fp32 and fp64 GELU over [-12, 12]: <= 1.5e-6 (1 ulp more from each hardware transcendental); beyond
// |x| = 6 both forms give x or (-)0.  A caller adds EPS / step to the half-integer margin of ofq_lsq_level_rcp_d and
// redoes a flagged group with ofq_gelu + the exact division, so the codes stay bit-identical.
#define OFQ_GELU_FAST_EPS 4e-6f
__device__ __forceinline__ float ofq_gelu_fast(float y) {
  const float x = y * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(fabsf(x), 0.3275911f, 1.0f));
  float pl = __builtin_fmaf(t, 1.061405429f, -1.453152027f);
  pl = __builtin_fmaf(pl, t, 1.421413741f);
  pl = __builtin_fmaf(pl, t, -0.284496736f);
  pl = __builtin_fmaf(pl, t, 0.254829592f);
  const float pe = (pl * t) * __builtin_amdgcn_exp2f((x * x) * -1.4426950408889634f);      // 1 - erf(|x|)
  return (0.5f * y) * (x < 0.f ? pe : 2.0f - pe);
}
__device__ __forceinline__ float ofq_gelu_grad(float x) {
  // d/dx [x * Phi(x)] = Phi(x) + x * phi(x)
  float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

template <typename T>
__device__ __forceinline__ T ofq_wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// sum inside aligned groups of W lanes (W power of two <= 64)
template <int W>
__device__ __forceinline__ float ofq_group_sum(float v) {
#pragma unroll
  for (int o = W / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float ofq_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// By-product of the backward kernels that WRITE a gradient tensor: max |element| as the bits of a float, into a group of
// OFQ_AMAX_WORDS slots, OFQ_AMAX_STRIDE words apart (8 KB, zeroed by the caller), from which the two-plane fp16 GEMMs that consume the tensor take their
// power-of-two scale (csrc/qgemm_planes.hip, split2_f16).  m = this lane's running fmaxf(|x|) (>= 0).  One wave-wide reduction, then
// one fire-and-forget atomic maximum per wave on the word picked by the wave's position (64 words: the ~10^4 waves of a launch
// do not queue up behind one address, and nobody waits for a returned value).  The maximum of the group is the tensor's
// maximum; maxima commute, so the group does not depend on the order of the updates.
#define OFQ_AMAX_WORDS 64
#define OFQ_AMAX_STRIDE 32          // words between the slots of a group: every slot on its own 128-byte line (atomic units work per line)
__device__ __forceinline__ void ofq_amax_publish(unsigned* amax, float m) {
  m = ofq_wave_max(m);
  const unsigned b = __float_as_uint(m);
  if ((threadIdx.x & 63) == 0 && b != 0u) {
    const unsigned slot = (blockIdx.x * 5u + blockIdx.y * 3u + (threadIdx.x >> 6)) & (OFQ_AMAX_WORDS - 1);
    __hip_atomic_fetch_max(amax + slot * OFQ_AMAX_STRIDE, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// the maximum of a word group, by every wave that asks (all 64 lanes active): the value as a float (NaN pattern included)
__device__ __forceinline__ float ofq_amax_load(const unsigned* amax) {
  unsigned b = amax[(threadIdx.x & (OFQ_AMAX_WORDS - 1)) * OFQ_AMAX_STRIDE];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned t = __shfl_xor(b, o, 64);
    b = t > b ? t : b;
  }
  return __uint_as_float(b);
}
__device__ __forceinline__ float ofq_absmax4(float m, float a, float b, float c, float d) {
  return fmaxf(fmaxf(m, fabsf(a)), fmaxf(fabsf(b), fmaxf(fabsf(c), fabsf(d))));
}

__host__ __device__ static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// second stage: dst[c] = scale * sum_{r<nrows} sum_{t<cnt} src[r*row_stride + c*cnt + t], fixed order.
// Up to three independent jobs (ds, db4, dbaft) in one launch: blockIdx.y selects the job.
// source offset of column c: (c / col_div) * col_mul + (c % col_div) * cnt   (col_div = 0: plain c * cnt)
struct SumJob { const float* src; float* dst; int64_t ncols, nrows, row_stride; int cnt; float scale; int64_t col_div, col_mul; };
struct SumJobs { SumJob j[5]; };

// 1024 threads = CPB columns x (1024 / CPB) row lanes, four independent accumulators per lane: the partials are a few
// hundred rows deep, so the kernel is a chain of dependent L2 round trips unless many loads are in flight.  CPB = 64
// (one 256-B line per row) for wide jobs, 16 when the job has few columns, so that enough workgroups exist.
#define OFQ_SUM_COLS 64
// sum of `cnt` consecutive partials of one (row, column): eight loads in flight (the image quantiser's step gradient has
// 98 of them per row: as a chain of dependent loads that one job took 131 - 250 us per step)
__device__ __forceinline__ float ofq_sum_run(const float* __restrict__ p, int cnt) {
  if (cnt == 1) return p[0];
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int t = 0;
  for (; t + 7 < cnt; t += 8) {
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] += p[t + e];
  }
  for (; t < cnt; ++t) s[0] += p[t];
  return ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}
template <int CPB>
__device__ __forceinline__ void strided_sum_body(const SumJob& jb) {
  constexpr int RL = 1024 / CPB;
  __shared__ float part[RL][CPB + 1];
  const int cx = threadIdx.x % CPB, py = threadIdx.x / CPB;
  const int64_t c = (int64_t)blockIdx.x * CPB + cx;
  if ((int64_t)blockIdx.x * CPB >= jb.ncols) return;          // (a merged launch is as wide as its widest job)
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (jb.dst && c < jb.ncols) {
    const int64_t co = jb.col_div ? (c / jb.col_div) * jb.col_mul + (c % jb.col_div) * jb.cnt : c * jb.cnt;
    const float* base = jb.src + co;
    int64_t r = py;
    for (; r + 7 * RL < jb.nrows; r += 8 * RL) {         // thousands of partial rows (Swin windows): eight loads in flight
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] += ofq_sum_run(base + (r + RL * u) * jb.row_stride, jb.cnt);
    }
    for (; r + 3 * RL < jb.nrows; r += 4 * RL) {
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] += ofq_sum_run(base + (r + RL * u) * jb.row_stride, jb.cnt);
    }
    for (; r < jb.nrows; r += RL) acc[0] += ofq_sum_run(base + r * jb.row_stride, jb.cnt);
  }
  part[py][cx] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (py == 0 && jb.dst && c < jb.ncols) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < RL; ++t) s += part[t][cx];
    jb.dst[c] = s * jb.scale;
  }
}
template <int CPB>
__global__ __launch_bounds__(1024) void strided_sum_kernel_t(SumJobs jobs) {
  strided_sum_body<CPB>(jobs.j[blockIdx.y]);
}
// many jobs in one launch (ofq_sum_flush): the deferred second stages of several kernels
#define OFQ_SUM_MULTI 40
struct SumJobsMulti { SumJob j[OFQ_SUM_MULTI]; };
template <int CPB>
__global__ __launch_bounds__(1024) void strided_sum_multi_kernel_t(SumJobsMulti jobs) {
  strided_sum_body<CPB>(jobs.j[blockIdx.y]);
}

// Deferral (ofq_sum_defer / ofq_sum_flush, include/ofq_hip.h): while it is on, second stages are queued instead of
// launched -- their results are parameter gradients that nobody reads before the optimiser step (or the gradient
// all-reduce), and a step has ~95 of these 5-15 us launches.  Each job keeps the lane layout (CPB) its own launch would
// have used, so a flushed result equals the immediate one bit for bit.  The caller keeps the partial buffers alive
// and unmodified until the flush.
struct SumPending { SumJob job; int cpb; };
struct SumDeferState { bool on = false; std::vector<SumPending> pend; };
static inline SumDeferState& sum_defer_state() { static SumDeferState s; return s; }

static inline void strided_sum_launch(const SumJobs& jobs, int64_t maxcols, int njobs, hipStream_t st) {
  // jobs with a handful of columns over thousands of partial rows (Swin: 49 window-token scales over every window of
  // the batch) get their own launch with 256 row lanes per column; the rest go together as before
  SumJobs deep = jobs, rest = jobs;
  int64_t deep_cols = 0, rest_cols = 0;
  for (int i = 0; i < njobs; ++i) {
    if (!jobs.j[i].dst) continue;
    const bool d = jobs.j[i].ncols <= 64 && jobs.j[i].nrows >= 2048;
    (d ? rest : deep).j[i].dst = nullptr;
    int64_t& mc = d ? deep_cols : rest_cols;
    if (jobs.j[i].ncols > mc) mc = jobs.j[i].ncols;
  }
  (void)maxcols;
  // 64 columns per workgroup (one 256-B line per partial row) only when that still gives every CU a workgroup: the
  // qkx recompute-backward's three jobs (2304 columns x 197 partial rows) ran as 108 workgroups, 16 us for 3.6 MB
  const bool wide = rest_cols > 2048 && ceil_div(rest_cols, 64) * njobs >= 256;
  SumDeferState& ds = sum_defer_state();
  if (ds.on) {
    for (int i = 0; i < njobs; ++i) {
      if (deep.j[i].dst) ds.pend.push_back({deep.j[i], 4});
      if (rest.j[i].dst) ds.pend.push_back({rest.j[i], wide ? 64 : 16});
    }
    return;
  }
  if (deep_cols > 0)
    hipLaunchKernelGGL(strided_sum_kernel_t<4>, dim3((unsigned)ceil_div(deep_cols, 4), njobs), dim3(1024), 0, st, deep);
  if (rest_cols <= 0) return;
  if (wide)
    hipLaunchKernelGGL(strided_sum_kernel_t<64>, dim3((unsigned)ceil_div(rest_cols, 64), njobs), dim3(1024), 0, st, rest);
  else
    hipLaunchKernelGGL(strided_sum_kernel_t<16>, dim3((unsigned)ceil_div(rest_cols, 16), njobs), dim3(1024), 0, st, rest);
}
