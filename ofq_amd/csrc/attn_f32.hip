// Fused fp32-grade attention of the KD teacher (reference train.py:906-910 runs the frozen fp32 DistilledVisionTransformer in
// every step; its attention is timm's softmax(q k^T * scale) v, deit_vision_transformer.py:85-116 without the quantisers).
//
// One workgroup per (image, head): K and V of the head (N <= 224 keys, d = 64) live in LDS as two fp16 planes each
// (x 2^E = hi + lo, E from the tile's own maximum: nothing travels between kernels), every wave owns 32 queries whose Q rows
// go from global memory straight into MFMA operand layout.  The score tile is formed TRANSPOSED, S^T[key][query] = K . Q^T,
// so a lane holds every key of its one query (split over the two lane halves): the softmax is in-lane arithmetic plus one
// cross-half exchange, and the probabilities are already in the operand layout of the second product O^T[d][query] =
// V^T . P^T (the k order of the 16-key MFMA slabs is permuted identically on both operands: lane half lh holds keys
// 4 lh + {0..3} and 8 + 4 lh + {0..3} of a slab -- the accumulator layout of the first product -- and the V fragments are
// transpose-read from LDS at those rows).  Three plane products per algorithmic one (hi.hi + hi.lo + lo.hi: the dropped
// lo.lo term is 2^-22 of the product), accumulated in fp32: scores and probabilities never reach HBM (1 GB per block at 128
// images), against the three launches (strided fp32-MFMA scores, softmax, strided fp32-MFMA P.V) this replaces.
#define AF_LD 144                        // bytes per LDS row: 64 fp16 + 16 B pad
#define AF_KEYS 224                      // 7 blocks of 32 keys
struct AttnF32Args {
  const float* qkv;                      // [B N][3 H d]: q | k | v column thirds, head h at columns h d .. of its third
  float* out;                            // [B N][H d]
  int B, H, N;
  float scale;
};

__device__ __forceinline__ bf16x8 af_tr_frag(const unsigned char* base) {
  // the slab's keys 4 lh + {0..3} (first read) and 8 + 4 lh + {0..3} (second): see the header comment
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 8 * AF_LD));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

__global__ __launch_bounds__(512) void attn_f32_fwd_kernel(AttnF32Args p) {
  constexpr int D = 64, PLANE = AF_KEYS * AF_LD;
  __shared__ __attribute__((aligned(16))) unsigned char sm[4 * PLANE];        // K hi, K lo, V hi, V lo
  __shared__ float red[16];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int b = blockIdx.x / p.H, h = blockIdx.x - b * p.H;
  const int N = p.N, C = p.H * D;
  const int64_t ldq = 3 * (int64_t)C;
  const float* base = p.qkv + (int64_t)b * N * ldq + h * D;

  // ---- this wave's 32 queries, straight into B-operand layout: lane (query l31, half lh) holds d = 16 s + 8 lh + {0..7}
  const int q = wid * 32 + l31;
  const bool wave_on = wid * 32 < N;
  f32x4v qr[4][2];
  {
    const float* qp = base + (int64_t)min(q, N - 1) * ldq + 8 * lh;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int c = 0; c < 2; ++c) qr[s][c] = *reinterpret_cast<const f32x4v*>(qp + 16 * s + 4 * c);
  }
  // ---- K and V of the head: 7 float4 per thread and matrix, held until the tile maxima are known
  constexpr int NIT = (AF_KEYS * 16 + 511) / 512;
  f32x4v kr[NIT], vr[NIT];
  float kmax = 0.f, vmax = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int idx = tid + 512 * it;
    const int row = idx >> 4, c4 = idx & 15;
    const float* kp = base + (int64_t)min(row, N - 1) * ldq + C + 4 * c4;
    kr[it] = *reinterpret_cast<const f32x4v*>(kp);
    vr[it] = *reinterpret_cast<const f32x4v*>(kp + C);
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const bool ok = ((tid + 512 * it) >> 4) < N;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      kmax = fmaxf(kmax, ok ? fabsf(kr[it][e]) : 0.f);
      vmax = fmaxf(vmax, ok ? fabsf(vr[it][e]) : 0.f);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    kmax = fmaxf(kmax, __shfl_xor(kmax, o, 64));
    vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
  }
  if (lane == 0) { red[wid] = kmax; red[8 + wid] = vmax; }
  __syncthreads();
  kmax = red[0]; vmax = red[8];
#pragma unroll
  for (int i = 1; i < 8; ++i) { kmax = fmaxf(kmax, red[i]); vmax = fmaxf(vmax, red[8 + i]); }
  float sEk, iEk, sEv, iEv;
  f16_plane_scale(kmax, sEk, iEk);
  f16_plane_scale(vmax, sEv, iEv);
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int idx = tid + 512 * it;
    const int row = idx >> 4, c4 = idx & 15;
    if (row < AF_KEYS) {
      const float ks = row < N ? sEk : 0.f, vs = row < N ? sEv : 0.f;          // rows past the last key: zeros
      unsigned h0, l0, h1, l1;
      split2_f16(kr[it][0] * ks, kr[it][1] * ks, h0, l0);
      split2_f16(kr[it][2] * ks, kr[it][3] * ks, h1, l1);
      *reinterpret_cast<uint2*>(&sm[row * AF_LD + c4 * 8]) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(&sm[PLANE + row * AF_LD + c4 * 8]) = make_uint2(l0, l1);
      split2_f16(vr[it][0] * vs, vr[it][1] * vs, h0, l0);
      split2_f16(vr[it][2] * vs, vr[it][3] * vs, h1, l1);
      *reinterpret_cast<uint2*>(&sm[2 * PLANE + row * AF_LD + c4 * 8]) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(&sm[3 * PLANE + row * AF_LD + c4 * 8]) = make_uint2(l0, l1);
    }
  }
  // ---- Q planes (per wave: its own power of two)
  float qmax = 0.f;
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) qmax = fmaxf(qmax, fabsf(qr[s][c][e]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) qmax = fmaxf(qmax, __shfl_xor(qmax, o, 64));
  float sEq, iEq;
  f16_plane_scale(qmax, sEq, iEq);
  bf16x8 qh[4], ql[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    unsigned hw[4], lw[4];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      split2_f16(qr[s][c][0] * sEq, qr[s][c][1] * sEq, hw[2 * c], lw[2 * c]);
      split2_f16(qr[s][c][2] * sEq, qr[s][c][3] * sEq, hw[2 * c + 1], lw[2 * c + 1]);
    }
    const i32x4 hv = {(int)hw[0], (int)hw[1], (int)hw[2], (int)hw[3]}, lv = {(int)lw[0], (int)lw[1], (int)lw[2], (int)lw[3]};
    qh[s] = __builtin_bit_cast(bf16x8, hv);
    ql[s] = __builtin_bit_cast(bf16x8, lv);
  }
  __syncthreads();
  if (!wave_on) return;

  // ---- S^T[key][query] = K . Q^T: 7 key blocks x 4 slabs of d x 3 plane products
  constexpr int NKB = AF_KEYS / 32;
  f32x16q sacc[NKB];
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
    for (int e = 0; e < 16; ++e) sacc[kb][e] = 0.f;
    const unsigned char* kp = &sm[(kb * 32 + l31) * AF_LD + lh * 16];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bf16x8 kh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const i32x4*>(kp + s * 32));
      const bf16x8 kl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const i32x4*>(kp + PLANE + s * 32));
      sacc[kb] = mfma_16b<true>(kh, qh[s], sacc[kb]);
      sacc[kb] = mfma_16b<true>(kh, ql[s], sacc[kb]);
      sacc[kb] = mfma_16b<true>(kl, qh[s], sacc[kb]);
    }
  }
  // ---- softmax over the keys of this lane's query: t = scale * s * log2(e); p = 2^(t - max t)
  const float c1 = iEq * iEk * p.scale * 1.4426950408889634f;
  float mx = -INFINITY;
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
      const float t = key < N ? sacc[kb][e] * c1 : -INFINITY;
      sacc[kb][e] = t;
      mx = fmaxf(mx, t);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float lsum = 0.f;
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float pe = __builtin_amdgcn_exp2f(sacc[kb][e] - mx);
      sacc[kb][e] = pe;
      lsum += pe;
    }
  lsum += __shfl_xor(lsum, 32, 64);
  // ---- O^T[d][query] = V^T . P^T: the probabilities (x 2^14, two planes) are the B operand as they lie in the accumulators
  f32x16q oacc[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[db][e] = 0.f;
  const int p16 = lane & 15;
  const int vfr = (4 * lh + (p16 >> 2)) * AF_LD + (16 * ((lane >> 4) & 1) + 4 * (p16 & 3)) * 2;
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      unsigned hw[4], lw[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        split2_f16(sacc[kb][8 * s2 + 2 * j] * 16384.f, sacc[kb][8 * s2 + 2 * j + 1] * 16384.f, hw[j], lw[j]);
      const i32x4 hv = {(int)hw[0], (int)hw[1], (int)hw[2], (int)hw[3]}, lv = {(int)lw[0], (int)lw[1], (int)lw[2], (int)lw[3]};
      const bf16x8 ph = __builtin_bit_cast(bf16x8, hv), pl = __builtin_bit_cast(bf16x8, lv);
      const unsigned char* vp = &sm[2 * PLANE + (kb * 32 + 16 * s2) * AF_LD + vfr];
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        const bf16x8 vh = af_tr_frag(vp + db * 64), vl = af_tr_frag(vp + PLANE + db * 64);
        oacc[db] = mfma_16b<true>(vh, ph, oacc[db]);
        oacc[db] = mfma_16b<true>(vh, pl, oacc[db]);
        oacc[db] = mfma_16b<true>(vl, ph, oacc[db]);
      }
    }
  if (q < N) {
    const float c2 = __fdiv_rn(iEv * (1.f / 16384.f), lsum);
    float* op = p.out + ((int64_t)b * N + q) * C + h * D + 4 * lh;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        f32x4v o = {oacc[db][4 * r] * c2, oacc[db][4 * r + 1] * c2, oacc[db][4 * r + 2] * c2, oacc[db][4 * r + 3] * c2};
        *reinterpret_cast<f32x4v*>(op + 32 * db + 8 * r) = o;
      }
  }
}

// softmax(scale * q k^T) v per (image, head) for the teacher's fp32 activations: qkv [B N][3 H 64], out [B N][H 64]
extern "C" int ofq_attn_f32_fwd(const float* qkv, float* out, int64_t B, int64_t H, int64_t N, int64_t d, float scale,
                                ofq_stream_t stream) {
  if (!qkv || !out || B <= 0 || H <= 0 || N <= 0 || d != 64 || N > AF_KEYS || B * H >= (1ll << 31) || !al16(qkv) || !al16(out))
    return OFQ_EINVAL;
  AttnF32Args a = {qkv, out, (int)B, (int)H, (int)N, scale};
  hipLaunchKernelGGL(attn_f32_fwd_kernel, dim3((unsigned)(B * H)), dim3(512), 0, (hipStream_t)stream, a);
  OFQ_LAUNCH_CHECK();
  return 0;
}
