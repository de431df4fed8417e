// K1  StatsQ weight quantiser (reference: src/quantization/quantizer/statsq.py:133-150).
// One wave per weight row, 4 rows per workgroup: float4 coalesced loads, fp64 wave reduction of
// sum|w| (so the fp32 scale is the correctly rounded row mean, independent of lane order), then a
// second pass over the (L1/L2-resident) row that quantises and writes W_hat, the scale and int8 levels.
// HBM-bound: 4 B read + 4 B written per weight (+1 B for the optional level).
#include "common.h"

// a small integer code as a 16-bit float: bf16 (the three-plane backward GEMMs) or fp16 (the two-plane ones), both exact
__device__ __forceinline__ unsigned short statsq_code16(float code, int f16) {
  if (f16) { const _Float16 h = (_Float16)code; return __builtin_bit_cast(unsigned short, h); }
  return (unsigned short)(__float_as_uint(code) >> 16);
}

__device__ __forceinline__ void statsq_row(const float* __restrict__ W, int64_t rows, int64_t cols, float n,
                                           float* __restrict__ out, float* __restrict__ scale, int8_t* __restrict__ levels,
                                           int scale_given, int odd_codes, unsigned short* __restrict__ codesT,
                                           const float* __restrict__ rvec, float* __restrict__ rout, int64_t row, int lane,
                                           int t_f16 = 0) {
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
    if (codesT) codesT[k * rows + row] = statsq_code16(code, t_f16);   // small integers: exact in bf16 / fp16
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
    // the offset row-dot is accumulated in four interleaved partial sums (float4 column c goes to partial (c / 64) % 4),
    // each reduced over the wave, then (p0 + p1) + (p2 + p3): the order of statsq_multi_kernel's eight-row form, where
    // the four waves of a workgroup hold those partials -- the two launch forms stay bit-identical
    const float4* w4 = reinterpret_cast<const float4*>(w);
    float4* o4 = reinterpret_cast<float4*>(o);
    float rp[4] = {0.f, 0.f, 0.f, 0.f};
    for (int64_t i0 = lane; i0 < cols / 4; i0 += 256) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t i = i0 + 64 * u;
        if (i >= cols / 4) break;
        float4 t = w4[i], r;
        float L0, L1, L2, L3;
        r.x = q1(t.x, L0); r.y = q1(t.y, L1); r.z = q1(t.z, L2); r.w = q1(t.w, L3);
        if (o) o4[i] = r;
        if (lv) {
          if (odd_codes) { L0 = 2.f * L0 + 1.f; L1 = 2.f * L1 + 1.f; L2 = 2.f * L2 + 1.f; L3 = 2.f * L3 + 1.f; }
          char4 c4 = make_char4((signed char)L0, (signed char)L1, (signed char)L2, (signed char)L3);
          reinterpret_cast<char4*>(lv)[i] = c4;
          if (codesT) {
            codesT[(4 * i) * rows + row] = statsq_code16(L0, t_f16);
            codesT[(4 * i + 1) * rows + row] = statsq_code16(L1, t_f16);
            codesT[(4 * i + 2) * rows + row] = statsq_code16(L2, t_f16);
            codesT[(4 * i + 3) * rows + row] = statsq_code16(L3, t_f16);
          }
          if (rvec) {
            const float4 rv = make_float4(rvec[4 * i], rvec[4 * i + 1], rvec[4 * i + 2], rvec[4 * i + 3]);
            rp[u] += (rv.x * L0 + rv.y * L1) + (rv.z * L2 + rv.w * L3);
          }
        }
      }
    }
    if (rout) {
      const float p0 = ofq_wave_sum(rp[0]), p1 = ofq_wave_sum(rp[1]), p2 = ofq_wave_sum(rp[2]), p3 = ofq_wave_sum(rp[3]);
      if (lane == 0) rout[row] = (p0 + p1) + (p2 + p3);
    }
    return;
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
                                                         float* __restrict__ rout, int t_f16) {
  statsq_row(W, rows, cols, n, out, scale, levels, scale_given, odd_codes, codesT, rvec, rout,
             (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), threadIdx.x & 63, t_f16);
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
// Eight rows per workgroup.  Phase 1: each wave takes two rows and forms their scales exactly as statsq_row does (fp64 sum of
// |w| over the 64 lanes).  Phase 2: a thread owns four consecutive columns of ALL eight rows, so the transposed bf16 codes
// leave as one 16-byte store per column (eight consecutive rows) instead of eight scattered 2-byte stores -- those stores
// bounded the per-row form (111 us for the 17.7 M leaf weights of DeiT-S).  Element for element the arithmetic is
// statsq_row's; the offset row-dot is summed in a different order (fp32, compared at 1e-6).
#define STATSQ_RPB 8
__global__ __launch_bounds__(256) void statsq_multi_kernel(StatsqPack pk) {
  int t = 0;
  while (t + 1 < pk.n && (int)blockIdx.x >= pk.first_block[t + 1]) ++t;          // uniform scan over <= 40 entries
  const StatsqEntry& e = pk.e[t];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t row0 = (int64_t)((int)blockIdx.x - pk.first_block[t]) * STATSQ_RPB;
  const int t_f16 = (int)((e.bits >> 8) & 1);          // bits | 0x100: the transposed codes as fp16 instead of bf16
  const int bits = (int)(e.bits & 0xff);
  const float n = (float)(1 << (bits - 1));
  const bool tiled = ((e.cols & 3) == 0) && ((e.rows & 7) == 0) && ((((uintptr_t)e.W) & 15) == 0) && e.codesT &&
                     ((((uintptr_t)e.codesT) & 15) == 0) && ((((uintptr_t)e.codes) & 3) == 0) && ((((uintptr_t)e.rvec) & 15) == 0);
  if (!tiled) {                                         // odd geometry: the per-row form, two rows per wave
    for (int i = 0; i < 2; ++i)
      statsq_row(e.W, e.rows, e.cols, n, nullptr, e.scale, e.codes, 0, 1, e.codesT, e.rvec, e.rout, row0 + 2 * wid + i, lane, t_f16);
    return;
  }
  __shared__ float s_sh[STATSQ_RPB];
  __shared__ float r_sh[4][STATSQ_RPB];
  for (int i = 0; i < 2; ++i) {
    const int64_t row = row0 + 2 * wid + i;             // rows % 8 == 0: every row of the group exists
    const float4* w4 = reinterpret_cast<const float4*>(e.W + row * e.cols);
    double acc = 0.0;
    for (int64_t c = lane; c < e.cols / 4; c += 64) {
      const float4 v = w4[c];
      acc += (double)fabsf(v.x) + (double)fabsf(v.y) + (double)fabsf(v.z) + (double)fabsf(v.w);
    }
    acc = ofq_wave_sum(acc);
    const float s = 2.0f * ofq_div((float)acc, (float)e.cols);                     // statsq.py:138
    if (lane == 0) { s_sh[2 * wid + i] = s; e.scale[row] = s; }
  }
  __syncthreads();
  float sc[STATSQ_RPB], racc[STATSQ_RPB];
#pragma unroll
  for (int r = 0; r < STATSQ_RPB; ++r) { sc[r] = s_sh[r]; racc[r] = 0.f; }
  const float cmax = 1.0f - 1e-6f;
  for (int64_t c = threadIdx.x; c < e.cols / 4; c += 256) {
    float4 wv[STATSQ_RPB];
#pragma unroll
    for (int r = 0; r < STATSQ_RPB; ++r) wv[r] = reinterpret_cast<const float4*>(e.W + (row0 + r) * e.cols)[c];
    const float4 rv = e.rvec ? *reinterpret_cast<const float4*>(e.rvec + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned short ct[4][STATSQ_RPB];
#pragma unroll
    for (int r = 0; r < STATSQ_RPB; ++r) {
      const float in[4] = {wv[r].x, wv[r].y, wv[r].z, wv[r].w};
      float code[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float v = ofq_div(in[k], sc[r]);                                     // statsq.py:144
        const float cl = fminf(fmaxf(v, -1.0f), cmax);                             // :145
        const float L = rintf(__fsub_rn(__fmul_rn(cl, n), 0.5f));                  // :147
        code[k] = 2.f * L + 1.f;                                                    // odd codes 2L+1
        ct[k][r] = statsq_code16(code[k], t_f16);                                  // small integers: exact in bf16 / fp16
      }
      reinterpret_cast<char4*>(e.codes + (row0 + r) * e.cols)[c] =
          make_char4((signed char)code[0], (signed char)code[1], (signed char)code[2], (signed char)code[3]);
      racc[r] += (rv.x * code[0] + rv.y * code[1]) + (rv.z * code[2] + rv.w * code[3]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      uint4 pk8;
      pk8.x = ct[k][0] | ((unsigned)ct[k][1] << 16); pk8.y = ct[k][2] | ((unsigned)ct[k][3] << 16);
      pk8.z = ct[k][4] | ((unsigned)ct[k][5] << 16); pk8.w = ct[k][6] | ((unsigned)ct[k][7] << 16);
      *reinterpret_cast<uint4*>(e.codesT + (4 * c + k) * e.rows + row0) = pk8;
    }
  }
  if (e.rout) {
#pragma unroll
    for (int r = 0; r < STATSQ_RPB; ++r) {
      const float v = ofq_wave_sum(racc[r]);
      if (lane == 0) r_sh[wid][r] = v;
    }
    __syncthreads();
    if (threadIdx.x < STATSQ_RPB)
      e.rout[row0 + threadIdx.x] = (r_sh[0][threadIdx.x] + r_sh[1][threadIdx.x]) + (r_sh[2][threadIdx.x] + r_sh[3][threadIdx.x]);
  }
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
      if (!e.W || !e.scale || !e.codes || e.rows <= 0 || e.cols <= 0 || (e.bits & 0xff) < 1 || (e.bits & 0xff) > 7 || (e.bits & ~0x1ffll) ||
          (e.rvec && !e.rout))
        return OFQ_EINVAL;
      pk.e[i] = e;
      pk.first_block[i] = blocks;
      blocks += (int)((e.rows + STATSQ_RPB - 1) / STATSQ_RPB);
    }
    pk.first_block[pk.n] = blocks;
    hipLaunchKernelGGL(statsq_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pk);
    OFQ_LAUNCH_CHECK();
  }
  return 0;
}

static int statsq_launch(const float* W, int64_t rows, int64_t cols, int bits, float* out, float* scale, int8_t* levels,
                         int scale_given, int odd_codes, void* codesT, const float* rvec, float* rout, ofq_stream_t stream,
                         int t_f16 = 0) {
  if (!W || !scale || rows <= 0 || cols <= 0 || bits < 1 || bits > 8) return OFQ_EINVAL;
  if (odd_codes && bits > 7) return OFQ_EINVAL;       // 2L+1 must fit int8
  float n = (float)(1 << (bits - 1));
  dim3 grid((unsigned)((rows + 3) / 4));
  hipLaunchKernelGGL(statsq_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, W, rows, cols, n, out, scale,
                     levels, scale_given, odd_codes, (unsigned short*)codesT, rvec, rout, t_f16);
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
  return statsq_launch(W, rows, cols, bits & 0xff, out, scale, codes, 0, 1, codesT_bf16, rvec, rout, stream, (bits >> 8) & 1);
}
