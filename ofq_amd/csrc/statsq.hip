// K1  StatsQ weight quantiser (reference: src/quantization/quantizer/statsq.py:133-150).
// One wave per weight row, 4 rows per workgroup: float4 coalesced loads, fp64 wave reduction of
// sum|w| (so the fp32 scale is the correctly rounded row mean, independent of lane order), then a
// second pass over the (L1/L2-resident) row that quantises and writes W_hat, the scale and int8 levels.
// HBM-bound: 4 B read + 4 B written per weight (+1 B for the optional level).
#include "common.h"

__device__ __forceinline__ void statsq_row(const float* __restrict__ W, int64_t rows, int64_t cols, float n,
                                           float* __restrict__ out, float* __restrict__ scale, int8_t* __restrict__ levels,
                                           int scale_given, int odd_codes, unsigned short* __restrict__ codesT,
                                           const float* __restrict__ rvec, float* __restrict__ rout, int64_t row, int lane) {
  if (row >= rows) return;
  const float* w = W + row * cols;
  const bool vec = ((cols & 3) == 0) && ((((uintptr_t)W) & 15) == 0);
  float s;
  if (scale_given) {
    s = scale[row];
  } else {
    double acc = 0.0;
    if (vec) {
      const float4* w4 = reinterpret_cast<const float4*>(w);
      for (int64_t i = lane; i < cols / 4; i += 64) {
        float4 t = w4[i];
        acc += (double)fabsf(t.x) + (double)fabsf(t.y) + (double)fabsf(t.z) + (double)fabsf(t.w);
      }
    } else {
      for (int64_t i = lane; i < cols; i += 64) acc += (double)fabsf(w[i]);
    }
    acc = ofq_wave_sum(acc);
    float sum = (float)acc;                         // torch.mean: fp32 sum ...
    float mean = ofq_div(sum, (float)cols);         // ... divided by the count          statsq.py:138
    s = 2.0f * mean;
    if (lane == 0) scale[row] = s;
  }
  const float cmax = 1.0f - 1e-6f;                  // (clip_val/2) - 1e-6 in fp32      statsq.py:145
  float* o = out ? out + row * cols : nullptr;
  int8_t* lv = levels ? levels + row * cols : nullptr;
  float racc = 0.f;                                 // sum_k rvec[k] * code[row][k]  (offset term of the int8 GEMM)
  auto side = [&](int64_t k, float code) {          // by-products of the code path: transposed bf16 codes, row dot
    if (codesT) codesT[k * rows + row] = (unsigned short)(__float_as_uint(code) >> 16);   // small integers: exact in bf16
    if (rvec) racc += rvec[k] * code;
  };
  auto q1 = [&](float wv, float& L) -> float {
    float v = ofq_div(wv, s);                       // statsq.py:144
    float c = fminf(fmaxf(v, -1.0f), cmax);         // :145
    L = rintf(__fsub_rn(__fmul_rn(c, n), 0.5f));    // :147 round(c*n - 0.5), RNE
    float wq = __fmul_rn(s, ofq_div(__fadd_rn(L, 0.5f), n));
    return __fadd_rn(__fsub_rn(wq, wv), wv);        // :148 value of Wq.detach() - W.detach() + W
  };
  if (vec) {
    const float4* w4 = reinterpret_cast<const float4*>(w);
    float4* o4 = reinterpret_cast<float4*>(o);
    for (int64_t i = lane; i < cols / 4; i += 64) {
      float4 t = w4[i], r;
      float L0, L1, L2, L3;
      r.x = q1(t.x, L0); r.y = q1(t.y, L1); r.z = q1(t.z, L2); r.w = q1(t.w, L3);
      if (o) o4[i] = r;
      if (lv) {
        if (odd_codes) { L0 = 2.f * L0 + 1.f; L1 = 2.f * L1 + 1.f; L2 = 2.f * L2 + 1.f; L3 = 2.f * L3 + 1.f; }
        char4 c4 = make_char4((signed char)L0, (signed char)L1, (signed char)L2, (signed char)L3);
        reinterpret_cast<char4*>(lv)[i] = c4;
        side(4 * i, L0); side(4 * i + 1, L1); side(4 * i + 2, L2); side(4 * i + 3, L3);
      }
    }
  } else {
    for (int64_t i = lane; i < cols; i += 64) {
      float L;
      const float wq = q1(w[i], L);
      if (o) o[i] = wq;
      if (lv) {
        const float code = odd_codes ? 2.f * L + 1.f : L;
        lv[i] = (int8_t)code;
        side(i, code);
      }
    }
  }
  if (rout) {
    racc = ofq_wave_sum(racc);
    if (lane == 0) rout[row] = racc;
  }
}

__global__ __launch_bounds__(256) void statsq_fwd_kernel(const float* __restrict__ W, int64_t rows, int64_t cols,
                                                         float n, float* __restrict__ out,
                                                         float* __restrict__ scale, int8_t* __restrict__ levels,
                                                         int scale_given, int odd_codes,
                                                         unsigned short* __restrict__ codesT, const float* __restrict__ rvec,
                                                         float* __restrict__ rout) {
  statsq_row(W, rows, cols, n, out, scale, levels, scale_given, odd_codes, codesT, rvec, rout,
             (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), threadIdx.x & 63);
}

// Multi-tensor code path: the weights of a model only change at the optimizer step, so the scale / int8 codes / transposed
// bf16 codes / offset row-dots of ALL quantised linear layers are produced by one or two launches at the start of a
// training step (descriptors in the kernel arguments, like ofq_adamw_multi) instead of one 10-us, mostly latency-bound
// launch per layer inside its forward.  Row for row the arithmetic is statsq_row's.
#define STATSQ_MAX_TENSORS 40
struct StatsqEntry {          // host layout of ofq_statsq_codes_multi's table (ofq_statsq_tensor_entry_bytes() bytes each)
  const float* W; float* scale; int8_t* codes; unsigned short* codesT; const float* rvec; float* rout;
  int64_t rows, cols, bits;
};
struct StatsqPack {
  StatsqEntry e[STATSQ_MAX_TENSORS];
  int first_block[STATSQ_MAX_TENSORS + 1];
  int n;
};
__global__ __launch_bounds__(256) void statsq_multi_kernel(StatsqPack pk) {
  int t = 0;
  while (t + 1 < pk.n && (int)blockIdx.x >= pk.first_block[t + 1]) ++t;          // uniform scan over <= 40 entries
  const StatsqEntry& e = pk.e[t];
  statsq_row(e.W, e.rows, e.cols, (float)(1 << (e.bits - 1)), nullptr, e.scale, e.codes, 0, 1, e.codesT, e.rvec, e.rout,
             (int64_t)((int)blockIdx.x - pk.first_block[t]) * 4 + (threadIdx.x >> 6), threadIdx.x & 63);
}

extern "C" int64_t ofq_statsq_tensor_entry_bytes(void) { return (int64_t)sizeof(StatsqEntry); }

extern "C" int ofq_statsq_codes_multi(const void* host_entries, int64_t n, ofq_stream_t stream) {
  if (!host_entries || n <= 0) return OFQ_EINVAL;
  const StatsqEntry* he = (const StatsqEntry*)host_entries;
  for (int64_t base = 0; base < n; base += STATSQ_MAX_TENSORS) {
    StatsqPack pk;
    pk.n = (int)((n - base) < STATSQ_MAX_TENSORS ? (n - base) : STATSQ_MAX_TENSORS);
    int blocks = 0;
    for (int i = 0; i < pk.n; ++i) {
      const StatsqEntry& e = he[base + i];
      if (!e.W || !e.scale || !e.codes || e.rows <= 0 || e.cols <= 0 || e.bits < 1 || e.bits > 7 || (e.rvec && !e.rout))
        return OFQ_EINVAL;
      pk.e[i] = e;
      pk.first_block[i] = blocks;
      blocks += (int)((e.rows + 3) / 4);
    }
    pk.first_block[pk.n] = blocks;
    hipLaunchKernelGGL(statsq_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pk);
    OFQ_LAUNCH_CHECK();
  }
  return 0;
}

static int statsq_launch(const float* W, int64_t rows, int64_t cols, int bits, float* out, float* scale, int8_t* levels,
                         int scale_given, int odd_codes, void* codesT, const float* rvec, float* rout, ofq_stream_t stream) {
  if (!W || !scale || rows <= 0 || cols <= 0 || bits < 1 || bits > 8) return OFQ_EINVAL;
  if (odd_codes && bits > 7) return OFQ_EINVAL;       // 2L+1 must fit int8
  float n = (float)(1 << (bits - 1));
  dim3 grid((unsigned)((rows + 3) / 4));
  hipLaunchKernelGGL(statsq_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, W, rows, cols, n, out, scale,
                     levels, scale_given, odd_codes, (unsigned short*)codesT, rvec, rout);
  OFQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int ofq_statsq_fwd(const float* W, int64_t rows, int64_t cols, int bits, float* out, float* scale,
                              int8_t* levels, int scale_given, int odd_codes, ofq_stream_t stream) {
  if (!out) return OFQ_EINVAL;
  return statsq_launch(W, rows, cols, bits, out, scale, levels, scale_given, odd_codes, nullptr, nullptr, nullptr, stream);
}

extern "C" int ofq_statsq_codes_fwd(const float* W, int64_t rows, int64_t cols, int bits, float* out, float* scale,
                                    int8_t* codes, void* codesT_bf16, const float* rvec, float* rout, ofq_stream_t stream) {
  if (!codes || (rvec && !rout)) return OFQ_EINVAL;
  return statsq_launch(W, rows, cols, bits, out, scale, codes, 0, 1, codesT_bf16, rvec, rout, stream);
}
