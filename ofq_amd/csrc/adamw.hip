// AdamW step over many tensors in one launch (timm create_optimizer_v2 -> torch.optim.AdamW,
// train.py:662, 933), with the CGA weight freeze folded in (cga.py:953-1013): where frozen[i] != 0 the gradient is
// masked before the moment updates (cga.py:962) and the weight keeps its value (save :964 + restore :994-997 = no write).
// HBM-bound: 16 B read + 12 B written per parameter (p, g, m, v -> p, m, v).
//
// torch.optim.AdamW (single-tensor form, amsgrad = maximize = False), per element:
//   p <- p * (1 - lr * wd);  m <- m + (g - m) * (1 - b1);  v <- b2 * v + (1 - b2) * g * g;
//   p <- p - (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps),   bc1 = 1 - b1^t, bc2 = 1 - b2^t
#include "common.h"

struct AdamWTensor {       // one entry per parameter tensor (host array handed to ofq_adamw_multi)
  float* p; const float* g; float* m; float* v; const float* frozen;
  int64_t n;
};

#define ADAMW_CHUNK 16384          // elements per workgroup
#define ADAMW_PACK 40              // tensors per launch: their descriptors travel in the kernel arguments (no device table,
                                   // no host-to-device copy, nothing to keep alive)
struct AdamWPack {
  AdamWTensor t[ADAMW_PACK];
  int32_t first_chunk[ADAMW_PACK + 1];     // prefix sums of the chunk counts
  int32_t n;
};

// the eight step scalars: by value (ofq_adamw_multi) or read from device memory (ofq_adamw_multi_dev: a launch captured
// in a hipGraph must not carry the per-step lr and bias corrections in its arguments)
struct AdamWHyper { float lr, omb1, b2, omb2, eps, wd, bc1, sqrt_bc2; };

// guard: a device word (NULL: none) that, when non-zero, makes the launch leave p, m and v untouched -- the step guard of
// ofq_step_guard (a stream-K hand-off timed out somewhere in this step, on this or on another rank: the gradients are invalid).
// The word is compared as bits without a float's sign bit, so an int 1, a float 0.25 (an averaged flag) and a NaN all count.
template <bool DEV>
__global__ __launch_bounds__(256) void adamw_multi_kernel(AdamWPack pk, AdamWHyper hv, const AdamWHyper* __restrict__ hd,
                                                          const unsigned* __restrict__ guard) {
  if (guard != nullptr && (__hip_atomic_load(guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0x7fffffffu) != 0u) return;
  const AdamWHyper h = DEV ? *hd : hv;
  const float lr = h.lr, omb1 = h.omb1, b2 = h.b2, omb2 = h.omb2, eps = h.eps, wd = h.wd, bc1 = h.bc1, sqrt_bc2 = h.sqrt_bc2;
  int ti = 0;
  while (ti + 1 < pk.n && (int)blockIdx.x >= pk.first_chunk[ti + 1]) ++ti;
  const AdamWTensor t = pk.t[ti];
  const int64_t start = (int64_t)((int)blockIdx.x - pk.first_chunk[ti]) * ADAMW_CHUNK;
  const int64_t end = min(t.n, start + (int64_t)ADAMW_CHUNK);
  const float decay = 1.0f - lr * wd;
  const float step = lr / bc1;
  // omb1 = 1 - beta1, omb2 = 1 - beta2 are formed in double on the host (1.0f - 0.999f is off by 5e-5 relative)
  const bool vec = (((uintptr_t)t.p | (uintptr_t)t.g | (uintptr_t)t.m | (uintptr_t)t.v | (uintptr_t)t.frozen) & 15) == 0;
  auto one = [&](float& p, float g, float& m, float& v, float f) {
    const bool frz = f != 0.f;
    if (frz) g = 0.f;                                    // grad * (1 - frozen), frozen in {0, 1}
    m = m + (g - m) * omb1;
    v = b2 * v + omb2 * g * g;
    if (!frz) {
      const float pd = p * decay;
      p = pd - step * (m / (sqrtf(v) / sqrt_bc2 + eps));
    }
  };
  if (vec) {
    const int64_t n4 = (end - start) / 4;
    float4* p4 = reinterpret_cast<float4*>(t.p + start);
    const float4* g4 = reinterpret_cast<const float4*>(t.g + start);
    float4* m4 = reinterpret_cast<float4*>(t.m + start);
    float4* v4 = reinterpret_cast<float4*>(t.v + start);
    const float4* f4 = t.frozen ? reinterpret_cast<const float4*>(t.frozen + start) : nullptr;
    for (int64_t i = threadIdx.x; i < n4; i += 256) {
      float4 p = p4[i], g = g4[i], m = m4[i], v = v4[i];
      const float4 f = f4 ? f4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      one(p.x, g.x, m.x, v.x, f.x); one(p.y, g.y, m.y, v.y, f.y); one(p.z, g.z, m.z, v.z, f.z); one(p.w, g.w, m.w, v.w, f.w);
      p4[i] = p; m4[i] = m; v4[i] = v;
    }
    for (int64_t i = start + n4 * 4 + threadIdx.x; i < end; i += 256) {
      float p = t.p[i], m = t.m[i], v = t.v[i];
      one(p, t.g[i], m, v, t.frozen ? t.frozen[i] : 0.f);
      t.p[i] = p; t.m[i] = m; t.v[i] = v;
    }
  } else {
    for (int64_t i = start + threadIdx.x; i < end; i += 256) {
      float p = t.p[i], m = t.m[i], v = t.v[i];
      one(p, t.g[i], m, v, t.frozen ? t.frozen[i] : 0.f);
      t.p[i] = p; t.m[i] = m; t.v[i] = v;
    }
  }
}

extern "C" int64_t ofq_adamw_tensor_entry_bytes(void) { return (int64_t)sizeof(AdamWTensor); }

// host image of the eight scalars exactly as ofq_adamw_multi forms them (1 - beta in double, sqrt(bc2) in double)
extern "C" int ofq_adamw_hyper_pack(float* host8, float lr, double beta1, double beta2, float eps, float weight_decay,
                                    double bias_correction1, double bias_correction2) {
  if (!host8 || bias_correction1 <= 0.0 || bias_correction2 <= 0.0) return OFQ_EINVAL;
  const AdamWHyper h = {lr, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), eps, weight_decay,
                        (float)bias_correction1, (float)sqrt(bias_correction2)};
  memcpy(host8, &h, sizeof(h));
  return 0;
}

// dst[i] = vals[i], i < n <= 32: the values travel in the kernel arguments (no staging buffer to keep alive, legal next to
// a graph replay on the same stream)
struct StoreF32Args { float v[32]; };
__global__ void store_f32_kernel(float* dst, StoreF32Args a, int n) {
  if ((int)threadIdx.x < n) dst[threadIdx.x] = a.v[threadIdx.x];
}
extern "C" int ofq_store_f32(float* dst_dev, const float* host_vals, int n, ofq_stream_t stream) {
  if (!dst_dev || !host_vals || n <= 0 || n > 32) return OFQ_EINVAL;
  StoreF32Args a = {};
  for (int i = 0; i < n; ++i) a.v[i] = host_vals[i];
  hipLaunchKernelGGL(store_f32_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, dst_dev, a, n);
  OFQ_LAUNCH_CHECK();
  return 0;
}

static int adamw_launch(const void* tensors, int64_t n_tensors, const AdamWHyper& hv, const AdamWHyper* hd, const void* guard,
                        hipStream_t stream) {
  const unsigned* gd = (const unsigned*)guard;
  const AdamWTensor* ts = (const AdamWTensor*)tensors;
  for (int64_t base = 0; base < n_tensors; base += ADAMW_PACK) {
    AdamWPack pk = {};
    pk.n = (int32_t)((n_tensors - base < ADAMW_PACK) ? (n_tensors - base) : ADAMW_PACK);
    int64_t chunks = 0;
    for (int i = 0; i < pk.n; ++i) {
      pk.t[i] = ts[base + i];
      if (!pk.t[i].p || !pk.t[i].g || !pk.t[i].m || !pk.t[i].v || pk.t[i].n < 0) return OFQ_EINVAL;
      pk.first_chunk[i] = (int32_t)chunks;
      chunks += ceil_div(pk.t[i].n, ADAMW_CHUNK);
      if (chunks >= (1ll << 31)) return OFQ_EINVAL;
    }
    pk.first_chunk[pk.n] = (int32_t)chunks;
    if (chunks == 0) continue;
    if (hd) hipLaunchKernelGGL(adamw_multi_kernel<true>, dim3((unsigned)chunks), dim3(256), 0, stream, pk, hv, hd, gd);
    else hipLaunchKernelGGL(adamw_multi_kernel<false>, dim3((unsigned)chunks), dim3(256), 0, stream, pk, hv, hd, gd);
    OFQ_LAUNCH_CHECK();
  }
  return 0;
}

// tensors: HOST array of n_tensors entries {p, g, m, v, frozen (or NULL), n} with device pointers inside
extern "C" int ofq_adamw_multi(const void* tensors, int64_t n_tensors, float lr, double beta1, double beta2, float eps,
                               float weight_decay, double bias_correction1, double bias_correction2, ofq_stream_t stream) {
  if (!tensors || n_tensors <= 0 || bias_correction1 <= 0.0 || bias_correction2 <= 0.0) return OFQ_EINVAL;
  AdamWHyper h;
  ofq_adamw_hyper_pack((float*)&h, lr, beta1, beta2, eps, weight_decay, bias_correction1, bias_correction2);
  return adamw_launch(tensors, n_tensors, h, nullptr, nullptr, (hipStream_t)stream);
}
// the same with a step guard (see adamw_multi_kernel): guard = a 4-byte device word or NULL
extern "C" int ofq_adamw_multi_g(const void* tensors, int64_t n_tensors, float lr, double beta1, double beta2, float eps,
                                 float weight_decay, double bias_correction1, double bias_correction2, const void* guard,
                                 ofq_stream_t stream) {
  if (!tensors || n_tensors <= 0 || bias_correction1 <= 0.0 || bias_correction2 <= 0.0) return OFQ_EINVAL;
  AdamWHyper h;
  ofq_adamw_hyper_pack((float*)&h, lr, beta1, beta2, eps, weight_decay, bias_correction1, bias_correction2);
  return adamw_launch(tensors, n_tensors, h, nullptr, guard, (hipStream_t)stream);
}

// the same step with the eight scalars read from DEVICE memory (hyper_dev: 8 floats in ofq_adamw_hyper_pack's order)
extern "C" int ofq_adamw_multi_dev(const void* tensors, int64_t n_tensors, const float* hyper_dev, ofq_stream_t stream) {
  if (!tensors || n_tensors <= 0 || !hyper_dev) return OFQ_EINVAL;
  return adamw_launch(tensors, n_tensors, AdamWHyper{}, (const AdamWHyper*)hyper_dev, nullptr, (hipStream_t)stream);
}
extern "C" int ofq_adamw_multi_dev_g(const void* tensors, int64_t n_tensors, const float* hyper_dev, const void* guard,
                                     ofq_stream_t stream) {
  if (!tensors || n_tensors <= 0 || !hyper_dev) return OFQ_EINVAL;
  return adamw_launch(tensors, n_tensors, AdamWHyper{}, (const AdamWHyper*)hyper_dev, guard, (hipStream_t)stream);
}

// The step guard.  words[0..n): device words that are non-zero when something made this step's gradients invalid -- the error
// words of the stream-K workspaces (ofq_qgemm_bf16s_nt_sk), the flag elements the data-parallel wrapper appends to its gradient
// buckets (every rank's flag, averaged by the bucket's all-reduce: the same value on every rank).  If any is set (bits without
// the sign bit non-zero): loss[0] <- NaN, guard_u32[0] <- 1, flag_f32[0] <- 1.0; otherwise guard_u32[0] <- 0, flag_f32[0] <- 0.0
// and the loss stays.  Every output is optional.  One thread, no host sync, capturable.
struct StepGuardArgs { const unsigned* w[32]; int n; };
__global__ void step_guard_kernel(StepGuardArgs a, float* __restrict__ loss, unsigned* __restrict__ guard, float* __restrict__ flag) {
  unsigned any = 0u;
  for (int i = 0; i < a.n; ++i) any |= __hip_atomic_load(a.w[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0x7fffffffu;
  if (any != 0u && loss != nullptr) loss[0] = __builtin_nanf("");
  if (guard != nullptr) guard[0] = any != 0u ? 1u : 0u;
  if (flag != nullptr) flag[0] = any != 0u ? 1.f : 0.f;
}
extern "C" int ofq_step_guard(const void* const* words, int n_words, float* loss, void* guard_u32, float* flag_f32,
                              ofq_stream_t stream) {
  if (n_words < 0 || n_words > 32 || (n_words > 0 && !words)) return OFQ_EINVAL;
  StepGuardArgs a = {};
  a.n = n_words;
  for (int i = 0; i < n_words; ++i) {
    if (!words[i] || ((uintptr_t)words[i] & 3)) return OFQ_EINVAL;
    a.w[i] = (const unsigned*)words[i];
  }
  hipLaunchKernelGGL(step_guard_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, a, loss, (unsigned*)guard_u32, flag_f32);
  OFQ_LAUNCH_CHECK();
  return 0;
}
