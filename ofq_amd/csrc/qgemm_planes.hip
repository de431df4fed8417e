// ------------------------------------------------------------------------------------------------ bf16-split backward
#define QBS_BK 32                      // k per stage
#define QBS_LD (QBS_BK * 2 + 16)       // padded LDS row in bytes (bf16)

__device__ __forceinline__ unsigned pack_hi16(float lo_elem, float hi_elem) {
  // two fp32 whose low 16 bits are zero -> one dword of two bf16 (element order: lo_elem first)
  return (__float_as_uint(lo_elem) >> 16) | (__float_as_uint(hi_elem) & 0xffff0000u);
}
__device__ __forceinline__ float trunc_bf16(float x) { return __uint_as_float(__float_as_uint(x) & 0xffff0000u); }

typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
// three-way bf16 split of two neighbouring fp32 (one packed dword per plane).  Written on explicit 2-vectors so that the
// pairs are the naturally aligned (x,y) / (z,w) halves of the loaded float4: left to the SLP vectoriser the pairing was
// (x,w) / (y,z), whose register shuffles (v_mov of freshly loaded registers) forced early s_waitcnt vmcnt on the prefetch
template <int NS>
__device__ __forceinline__ void split_pair_bf16(f32x2v x, unsigned (&out)[NS]) {
  f32x2v rem = x;
#pragma unroll
  for (int q = 0; q < NS; ++q) {
    const u32x2v hb = __builtin_bit_cast(u32x2v, rem) & 0xffff0000u;
    out[q] = (hb.x >> 16) | hb.y;
    rem = rem - __builtin_bit_cast(f32x2v, hb);
  }
}

// ---- staging work interleaved into the MFMA stream -------------------------------------------------------------------
// tools/probe/filler_probe.hip: single-issue VALU instructions that follow an MFMA in the SAME wave's program order hide
// in its shadow -- with two waves per SIMD the first two per MFMA are free and each further one costs ~2.2 cycles
// instead of 4 -- while one v_pk_mul_f32 there costs ~10 cycles.  So the k-loops below place a few scalar staging
// instructions behind every MFMA (pinned with sched_barrier) instead of running the staging as a block after the MFMAs,
// and the split is written with these one-instruction wrappers so that the SLP vectoriser cannot re-pack it.
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

__device__ __forceinline__ float valu_mul(float a, float b) { float d; asm("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float valu_sub(float a, float b) { float d; asm("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float valu_add(float a, float b) { float d; asm("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float valu_hi16(float a) { float d; asm("v_and_b32 %0, 0xffff0000, %1" : "=v"(d) : "v"(a)); return d; }
__device__ __forceinline__ float valu_max(float a, float b) { float d; asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float valu_fma(float a, float b, float c) { float d; asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
// byte BYTE of w, sign-extended, as fp32 (one SDWA convert)
template <int BYTE>
__device__ __forceinline__ float valu_cvt_i8(unsigned w) {
  float d;
  if constexpr (BYTE == 0) asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0" : "=v"(d) : "v"(w));
  if constexpr (BYTE == 1) asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "=v"(d) : "v"(w));
  if constexpr (BYTE == 2) asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2" : "=v"(d) : "v"(w));
  if constexpr (BYTE == 3) asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3" : "=v"(d) : "v"(w));
  return d;
}
// four int8 codes of one dword -> two packed bf16 pairs (exact), one asm statement (4 SDWA converts + 2 v_perm_b32)
__device__ __forceinline__ void valu_cvt4_i8_bf16(unsigned w, unsigned& d01, unsigned& d23) {
  float t0, t1, t2, t3;
  asm("v_cvt_f32_i32_sdwa %2, sext(%6) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0\n\t"
      "v_cvt_f32_i32_sdwa %3, sext(%6) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n\t"
      "v_cvt_f32_i32_sdwa %4, sext(%6) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n\t"
      "v_cvt_f32_i32_sdwa %5, sext(%6) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3\n\t"
      "v_perm_b32 %0, %3, %2, %7\n\t"
      "v_perm_b32 %1, %5, %4, %7"
      : "=&v"(d01), "=&v"(d23), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
      : "v"(w), "s"(0x07060302u));
}
// ofq_lsq_eff_scale, instruction for instruction: a = max(s, 1e-5); t = a * g; (a - t) + t
__device__ __forceinline__ float valu_eff_scale(float s, float g) {
  float a, t, d;
  asm("v_max_f32 %0, 0x3727c5ac, %3\n\tv_mul_f32 %1, %0, %4\n\tv_sub_f32 %2, %0, %1\n\tv_add_f32 %2, %2, %1"
      : "=&v"(a), "=&v"(t), "=&v"(d) : "v"(s), "v"(g));
  return d;
}
// two-instruction steps of the split as ONE asm statement: between separate asm statements the hazard recogniser pads
// with s_nop (it cannot see what they are), which costs an issue slot each
__device__ __forceinline__ void valu_mul_hi16(float a, float b, float& x, float& p0) {        // x = a*b; p0 = hi16(x)
  asm("v_mul_f32 %0, %2, %3\n\tv_and_b32 %1, 0xffff0000, %0" : "=&v"(x), "=v"(p0) : "v"(a), "v"(b));
}
__device__ __forceinline__ void valu_sub_hi16(float a, float b, float& r, float& p) {          // r = a-b; p = hi16(r)
  asm("v_sub_f32 %0, %2, %3\n\tv_and_b32 %1, 0xffff0000, %0" : "=&v"(r), "=v"(p) : "v"(a), "v"(b));
}
// (lo_elem >> 16) | (hi_elem & 0xffff0000): the bf16 pair of two fp32 (truncating), one v_perm_b32
__device__ __forceinline__ void valu_pack3_hi16(const float (&a)[2], const float (&b)[2], const float (&c)[2], unsigned* d) {
  asm("v_perm_b32 %0, %4, %3, %9\n\tv_perm_b32 %1, %6, %5, %9\n\tv_perm_b32 %2, %8, %7, %9"
      : "=&v"(d[0]), "=&v"(d[1]), "=v"(d[2])
      : "v"(a[0]), "v"(a[1]), "v"(b[0]), "v"(b[1]), "v"(c[0]), "v"(c[1]), "s"(0x07060302u));
}
__device__ __forceinline__ unsigned valu_pack_hi16(float lo_elem, float hi_elem) {
  unsigned d;
  asm("v_perm_b32 %0, %1, %2, %3" : "=v"(d) : "v"(hi_elem), "v"(lo_elem), "s"(0x07060302u));
  return d;
}

// ---- two fp16 planes instead of three bf16 planes (round 5) -----------------------------------------------------------
// x (fp32, pre-scaled by a power of two so that the largest |x| of the launch sits below 2^15) = hi + lo + err with
// hi = rne_f16(x), lo = rne_f16(x - hi): x - hi is exact in fp32 (<= 13 significant bits), so |err| <= 2^-24 |x| as long
// as lo stays a normal fp16 (|x| >= 2^-3 of the scaled range), and <= 2^-25 absolute below that (fp16 denormals are kept by
// v_cvt_pk_f16_f32, v_fma_mix_f32 and v_mfma_f32_32x32x16_f16 alike: tools/probe/f16_mfma_probe.hip).  With the launch
// maximum scaled to [2^14, 2^15) that is: fp32 precision for every element within 2^-17 of the maximum, an absolute
// error of 2^-39 of the maximum below -- fp32-grade against the tensor scale, at two MFMAs per k-step instead of three.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <bool F16>
__device__ __forceinline__ f32x16q mfma_16b(bf16x8 a, bf16x8 b, f32x16q c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void valu_mul2(float a0, float b0, float a1, float b1, float& x0, float& x1) {
  asm("v_mul_f32 %0, %2, %3\n\tv_mul_f32 %1, %4, %5" : "=&v"(x0), "=v"(x1) : "v"(a0), "v"(b0), "v"(a1), "v"(b1));
}
__device__ __forceinline__ unsigned valu_cvt_pk_f16(float x0, float x1) {                       // (lo half: x0)
  unsigned d;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(d) : "v"(x0), "v"(x1));
  return d;
}
// r0 = x0 - f32(h.lo), r1 = x1 - f32(h.hi): one mixed-precision fma each, exact
__device__ __forceinline__ void valu_resid2_f16(unsigned h, float x0, float x1, float& r0, float& r1) {
  asm("v_fma_mix_f32 %0, %2, -1.0, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mix_f32 %1, %2, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "=&v"(r0), "=v"(r1) : "v"(h), "v"(x0), "v"(x1));
}
__device__ __forceinline__ void split2_f16(float x0, float x1, unsigned& hi, unsigned& lo) {
  float r0, r1;
  hi = valu_cvt_pk_f16(x0, x1);
  valu_resid2_f16(hi, x0, x1, r0, r1);
  lo = valu_cvt_pk_f16(r0, r1);
}
// four int8 codes of one dword -> two packed fp16 pairs (exact): bytes ^ 0x80 are the codes + 128 as unsigned, 0x64xx is
// the fp16 1024 + xx, minus 1152 gives the code.  c64 = 0x64646464 (a VGPR: v_perm_b32 may read one scalar operand only)
__device__ __forceinline__ void valu_cvt4_i8_f16(unsigned w, unsigned c64, unsigned& d01, unsigned& d23) {
  unsigned t;
  asm("v_xor_b32 %2, 0x80808080, %3\n\t"
      "v_perm_b32 %0, %4, %2, %5\n\t"
      "v_perm_b32 %1, %4, %2, %6\n\t"
      "v_pk_add_f16 %0, %0, %7\n\t"
      "v_pk_add_f16 %1, %1, %7"
      : "=&v"(d01), "=&v"(d23), "=&v"(t) : "v"(w), "v"(c64), "s"(0x04010400u), "s"(0x04030402u), "v"(0xE480E480u));
}
// The power of two that brings t (an upper bound of max |x| of the launch) into [2^14, 2^15), and its inverse; 1 for t = 0,
// inf or nan (a non-finite gradient stays non-finite through the product, as it would in fp32)
__device__ __forceinline__ void f16_plane_scale(float t, float& sE, float& inv_sE) {
  const int ex = (int)((__float_as_uint(t) >> 23) & 0xffu) - 127;
  int E = 14 - ex;
  E = E < -100 ? -100 : (E > 100 ? 100 : E);
  if (!(t > 0.f) || ((__float_as_uint(t) >> 23) & 0xffu) == 0xffu) E = 0;
  sE = __uint_as_float((unsigned)(E + 127) << 23);
  inv_sE = __uint_as_float((unsigned)(127 - E) << 23);
}
// max |v[0 .. n)| over an NT-thread workgroup, through NT / 64 floats of LDS at `red` (two barriers; every thread gets the result)
template <int NT = 512>
__device__ __forceinline__ float block_absmax(const float* __restrict__ v, int n, float* red, int tid) {
  float m = 0.f;
  for (int k = tid; k < n; k += NT) m = fmaxf(m, fabsf(v[k]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  float r = red[0];
#pragma unroll
  for (int i = 1; i < NT / 64; ++i) r = fmaxf(r, red[i]);
  __syncthreads();
  return r;
}
__device__ __forceinline__ float block512_absmax(const float* __restrict__ v, int n, float* red, int tid) {
  return block_absmax<512>(v, n, red, tid);
}

__device__ __forceinline__ unsigned i8x2_to_bf16x2(int b0, int b1) {
  // two small signed integers -> packed bf16 pair (exact for |v| <= 256)
  return (__float_as_uint((float)b0) >> 16) | (__float_as_uint((float)b1) & 0xffff0000u);
}

// NB > 1 (B_I8 = false): B is an fp32 matrix given as NB bf16 planes B = B_0 + B_1 + B_2 (plane r at B + r * sBp); the
// products A_q . B_r with q + r < PMAX are accumulated: PMAX = 5 keeps all nine (the exact product of the two fp32 values
// up to fp32 accumulation), PMAX = 3 the six leading ones (the dropped terms are <= 2^-24 of the product).  This is the
// GEMM of the frozen fp32 KD teacher, whose weights are split once: 6 / 9 bf16 MFMAs per k-step amortise the split of the
// activations that bounds the three-product kernels, at 16x the fp32-MFMA rate per instruction.
template <int NSPLIT, bool B_I8, int NB = 1, int PMAX = 5, bool F16 = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NB == 1 ? 3 : 2, NB == 1 ? 3 : 2))) void qgemm_bf16s_nt_kernel(QGemmArgs p) {
  static_assert(NB == 1 || !B_I8, "plane-split B is an fp32 operand");
  static_assert(!F16 || (NSPLIT == 2 && !B_I8 && NB == 1), "two fp16 planes: the linear layers' dX (B = fp16 codes)");
  constexpr int BM = 128, BN = 128;
  constexpr int PLANE = BM * QBS_LD;                 // bytes per bf16 plane of A
  constexpr int PLANE_B = BN * QBS_LD;
  __shared__ __attribute__((aligned(16))) unsigned char smem[NSPLIT * PLANE + NB * PLANE_B];
  int tm, tn, gby;
  qgemm_tile_id(p, tm, tn, gby);
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  const int b0 = gby / p.nb1, b1 = gby % p.nb1;
  const float* A = (const float*)p.A + b0 * p.sA0 + b1 * p.sA1;
  const unsigned short* B = (const unsigned short*)p.B + (B_I8 ? 0 : b0 * p.sB0 + b1 * p.sB1);
  const signed char* B8 = (const signed char*)p.B + b0 * p.sB0 + b1 * p.sB1;
  const float* ksp = p.s ? p.s + b1 * p.sK1 : nullptr;
  const int K = p.K;
  const int nkt = (K + QBS_BK - 1) / QBS_BK;

  // A: 128 rows x 32 fp32 = 1024 float4 -> 4 per thread (row = f >> 3, kq = f & 7)
  // B: 128 rows x 32 bf16 = 512 x 16 B  -> 2 per thread (row = f >> 2, kq = f & 3)
  int64_t offA[4], offB[2];
  bool okA[4], okB[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int f = tid + 256 * i;
    const int row = f >> 3;
    okA[i] = (m0 + row) < p.M;
    offA[i] = (int64_t)min(m0 + row, p.M - 1) * p.lda + (f & 7) * 4;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int f = tid + 256 * i;
    const int row = f >> 2;
    okB[i] = (n0 + row) < p.N;
    offB[i] = (int64_t)min(n0 + row, p.N - 1) * p.ldb + (f & 3) * 8;
  }
  const int kqa = (tid & 7) * 4;       // same for all 4 chunks (256 % 8 == 0)
  const int kqb = (tid & 3) * 8;
  // gload only issues the loads; scaling, masking, the int8 -> bf16 conversion and the split happen at the LDS store of
  // the next iteration, behind the MFMAs of this one (a value touched inside gload is waited for in front of them)
  float sE = 1.f, inv_sE = 1.f;        // F16: the launch's power of two (see qgemm_bf16s_nt_wide_sk_kernel)
  if constexpr (F16) {
    const float m = ksp ? block_absmax<256>(ksp, K, reinterpret_cast<float*>(smem), tid) : 1.f;
    const float a = ofq_amax_load(p.amax);
    f16_plane_scale(a == a ? a * m : a, sE, inv_sE);
  }
  f32x4v ra[4], rks;
  i32x4 rb[NB][2];
  bool rkina = false, rkinb = false;
  auto gload = [&](int kt) {
    const int k0 = kt * QBS_BK;
    const bool kina = (k0 + kqa) < K, kinb = (k0 + kqb) < K;       // K % 8 == 0 (host check)
    rks = *reinterpret_cast<const f32x4v*>(ksp ? ksp + (kina ? k0 + kqa : 0) : A + offA[0]);      // no ksp: any valid address
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const f32x4v*>(A + offA[i] + (kina ? k0 : -kqa));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (!B_I8) {
#pragma unroll
        for (int r = 0; r < NB; ++r) rb[r][i] = *reinterpret_cast<const i32x4*>(B + r * p.sBp + offB[i] + (kinb ? k0 : -kqb));
      } else {
        const u32x2v v = *reinterpret_cast<const u32x2v*>(B8 + offB[i] + (kinb ? k0 : -kqb));
        rb[0][i].x = (int)v[0];
        rb[0][i].y = (int)v[1];
      }
    }
    rkina = kina;
    rkinb = kinb;
  };
  auto lstore = [&]() {
    asm volatile("" : "+v"(rks));
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(ra[i]));
    float ks[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t = ksp ? rks[e] : 1.f;
      if (B_I8 && p.gscale2 > 0.f) t = ofq_lsq_eff_scale(t, p.gscale2);      // raw LSQ step -> effective value
      if constexpr (F16) t *= sE;
      ks[e] = rkina ? t : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = tid + 256 * i;
      const int row = f >> 3;
      const float z = okA[i] ? 1.f : 0.f;
      const f32x2v k01 = {ks[0] * z, ks[1] * z}, k23 = {ks[2] * z, ks[3] * z};
      const f32x2v a01 = {ra[i][0], ra[i][1]}, a23 = {ra[i][2], ra[i][3]};
      unsigned lo[NSPLIT], hi[NSPLIT];
      if constexpr (F16) {
        const f32x2v x01 = a01 * k01, x23 = a23 * k23;
        split2_f16(x01[0], x01[1], lo[0], lo[1]);
        split2_f16(x23[0], x23[1], hi[0], hi[1]);
      } else {
        split_pair_bf16<NSPLIT>(a01 * k01, lo);
        split_pair_bf16<NSPLIT>(a23 * k23, hi);
      }
#pragma unroll
      for (int sidx = 0; sidx < NSPLIT; ++sidx) {
        uint2 w;
        w.x = lo[sidx];
        w.y = hi[sidx];
        *reinterpret_cast<uint2*>(&smem[sidx * PLANE + row * QBS_LD + kqa * 2]) = w;
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int f = tid + 256 * i;
      const int row = f >> 2;
      const int m = (okB[i] && rkinb) ? -1 : 0;
      i32x4 w;
      if (!B_I8) {
#pragma unroll
        for (int r = 1; r < NB; ++r) {
          asm volatile("" : "+v"(rb[r][i]));
          *reinterpret_cast<i32x4*>(&smem[NSPLIT * PLANE + r * PLANE_B + row * QBS_LD + kqb * 2]) = rb[r][i] & m;
        }
        asm volatile("" : "+v"(rb[0][i]));
        w = rb[0][i] & m;
      } else {   // 8 int8 codes -> 8 bf16
        int w0 = rb[0][i].x, w1 = rb[0][i].y;
        asm volatile("" : "+v"(w0), "+v"(w1));
        w0 &= m;
        w1 &= m;
        w.x = (int)i8x2_to_bf16x2((int)(signed char)(w0 & 0xff), (int)(signed char)((w0 >> 8) & 0xff));
        w.y = (int)i8x2_to_bf16x2((int)(signed char)((w0 >> 16) & 0xff), (int)(signed char)((w0 >> 24) & 0xff));
        w.z = (int)i8x2_to_bf16x2((int)(signed char)(w1 & 0xff), (int)(signed char)((w1 >> 8) & 0xff));
        w.w = (int)i8x2_to_bf16x2((int)(signed char)((w1 >> 16) & 0xff), (int)(signed char)((w1 >> 24) & 0xff));
      }
      *reinterpret_cast<i32x4*>(&smem[NSPLIT * PLANE + row * QBS_LD + kqb * 2]) = w;
    }
  };

  f32x16q acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  gload(0);
  for (int kt = 0; kt < nkt; ++kt) {
    lstore();
    __syncthreads();
    gload(min(kt + 1, nkt - 1));        // unconditional (the last one is never stored) and pinned ahead of the MFMAs
    __builtin_amdgcn_sched_barrier(0);
    const unsigned char* a = &smem[(wm * 64 + l31) * QBS_LD + lh * 16];
    const unsigned char* b = &smem[NSPLIT * PLANE + (wn * 64 + l31) * QBS_LD + lh * 16];
#pragma unroll
    for (int ks = 0; ks < QBS_BK / 16; ++ks) {
      bf16x8 bv[NB][2];
#pragma unroll
      for (int r = 0; r < NB; ++r)
#pragma unroll
        for (int j = 0; j < 2; ++j) bv[r][j] = *reinterpret_cast<const bf16x8*>(b + r * PLANE_B + j * 32 * QBS_LD + ks * 32);
#pragma unroll
      for (int sidx = 0; sidx < NSPLIT; ++sidx) {
        bf16x8 av[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
          av[i] = *reinterpret_cast<const bf16x8*>(a + sidx * PLANE + i * 32 * QBS_LD + ks * 32);
#pragma unroll
        for (int r = 0; r < NB; ++r) {
          if (sidx + r >= PMAX) continue;
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = mfma_16b<F16>(av[i], bv[r][j], acc[i][j]);
        }
      }
    }
    __syncthreads();
  }

  // the per-row addend goes through LDS once (the loop's last barrier released smem) instead of 32 conditional global
  // loads per lane; old values for C += ... are fetched unconditionally on clamped addresses, a row quad at a time
  float* row_u = reinterpret_cast<float*>(smem);
  const bool has_u = B_I8 && p.u != nullptr;
  if (has_u) {
    if (tid < BM) row_u[tid] = p.u[((int64_t)b0 * p.M + min(m0 + tid, p.M - 1)) * p.nb1 + b1];
    __syncthreads();
  }
  float* Cb = p.C + b0 * p.sC0 + b1 * p.sC1;
  int ncc[2];
  bool nok[2];
  float cbias[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn * 64 + j * 32 + l31;
    nok[j] = n < p.N;
    ncc[j] = min(n, p.N - 1);
    cbias[j] = (NB > 1 && p.bias) ? p.bias[ncc[j]] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int eb = 0; eb < 4; ++eb) {
      float old[4][2];
      if (p.accumulate) {
#pragma unroll
        for (int ee = 0; ee < 4; ++ee) {
          const int mc = min(m0 + wm * 64 + i * 32 + ee + 8 * eb + 4 * lh, p.M - 1);
#pragma unroll
          for (int j = 0; j < 2; ++j) old[ee][j] = Cb[(int64_t)mc * p.ldc + ncc[j]];
        }
      }
#pragma unroll
      for (int ee = 0; ee < 4; ++ee) {
        const int ml = wm * 64 + i * 32 + ee + 8 * eb + 4 * lh;
        const int m = m0 + ml;
        const float uu = has_u ? row_u[ml] : 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j)
          if (m < p.M && nok[j]) {
            float v = acc[i][j][eb * 4 + ee] * (F16 ? p.alpha * inv_sE : p.alpha) + uu;
            if (NB > 1) v += cbias[j];
            if (p.accumulate) v += old[ee][j];
            Cb[(int64_t)m * p.ldc + ncc[j]] = v;
          }
      }
    }
}
