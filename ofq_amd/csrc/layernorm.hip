// LayerNorm over the channel dimension of the token matrix (deit_vision_transformer.py: Block.norm1 / norm2, the final
// norm; timm Block = x + attn(norm1(x)), x + mlp(norm2(x))), forward and backward, optionally fused with the residual
// add that precedes it.  HBM-bound: forward 8 B/elt (12 with the add), backward 12 B/elt (16 with the residual
// gradient) -- the stock three-kernel backward moves 32 B/elt.
//
//   fwd :  [xs = x + res]   mu = mean(xs), rstd = 1/sqrt(var(xs) + eps),   y = (xs - mu) * rstd * gamma + beta
//   bwd :  xh = (xs - mu) * rstd, g = dy * gamma,
//          dx = rstd * (g - mean(g) - xh * mean(g * xh))  [+ dres],   dgamma = sum_rows dy * xh,   dbeta = sum_rows dy
//
// One row is owned by TX lanes (32 or 64) holding J float4 each, so the row reductions are DPP/shuffle only; the forward
// walks rows with a one-row software prefetch (the backward does not: its extra operand sets would cost a wave per
// SIMD); the column sums for dgamma/dbeta go through LDS once per workgroup and a fixed-order second stage
// (strided_sum_kernel_t), like the LSQ offsets.  With the template flag Q the per-token LSQ that consumes the LayerNorm
// output is applied in the same pass (forward: codes only; backward: LSQ backward in front of the LayerNorm backward).
#include "common.h"

struct LnArgs {
  const float* x; const float* res; const float* gamma; const float* beta;
  float* y; float* xs; float* mean; float* rstd;
  const float* dy; const float* dres; float* dx; float* colpart;      // colpart [gx][2][C]
  unsigned* amax;        // backward, optional: bits of max |dx| (see ofq_amax_publish)
  int64_t R, C, ldx, ldy;
  int TX, TY;
  float eps;
  // fused per-token LSQ of the LayerNorm output (template flag Q): n = LN(xs) is quantised in the same pass,
  //   codes = LSQ(n + qb4[c]; step qs[r % qS]); backward recomputes n from (xs, mean, rstd) and applies ofq_lsq_bwd's
  //   arithmetic before the LayerNorm backward, so neither n nor its gradient ever travels through HBM
  const float* qs; const float* qb4; int8_t* qcodes; float* rowpart;
  int64_t qS;
  float qgscale, qlo, qhi;
  // token permutations (Swin's shifted-window partition / reverse folded into this pass; the rows come in images of permN):
  //   qperm[t]: where the QUANTISED row of token t lives inside its image -- codes, their gradient dy, the step index and the
  //             step-gradient partial all use row b * permN + qperm[t] (the window-major order the attention works in);
  //   rperm[t]: where token t's row of `res` lives (forward), and where its row of dx is written a second time (dx2, backward:
  //             the gradient of `res`, for a producer that works in that order).  NULL: the identity.
  const int* qperm; const int* rperm; float* dx2;
  int permN;
};

__device__ __forceinline__ int64_t ln_perm_row(int64_t r, const int* perm, int n) {
  if (perm == nullptr) return r;
  const int64_t t = r % n;
  return r - t + perm[t];
}

template <int TXW>
__device__ __forceinline__ float ln_row_sum(float v) {
#pragma unroll
  for (int o = TXW / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int J, int TXW, bool BWD, bool Q>
__global__ __launch_bounds__(256) void layernorm_kernel(LnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float red[];
  constexpr int TY = 256 / TXW;
  const int tx = threadIdx.x % TXW, ty = threadIdx.x / TXW;
  const int64_t w4 = a.C / 4;
  const float invC = 1.0f / (float)a.C;
  float dxmax = 0.f;
  float4 gam[J], bet[J];
  bool cok[J];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int64_t c4 = tx + (int64_t)j * TXW;
    cok[j] = c4 < w4;
    gam[j] = make_float4(1.f, 1.f, 1.f, 1.f);
    bet[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cok[j]) {
      if (a.gamma) gam[j] = *reinterpret_cast<const float4*>(a.gamma + c4 * 4);
      if ((!BWD || Q) && a.beta) bet[j] = *reinterpret_cast<const float4*>(a.beta + c4 * 4);
    }
  }
  float4 qb4v[J];
  if (Q) {
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int64_t c4 = tx + (int64_t)j * TXW;
      qb4v[j] = (cok[j] && a.qb4) ? *reinterpret_cast<const float4*>(a.qb4 + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  float4 acc_g[J], acc_b[J], acc_a[J];       // column sums: dgamma, dbeta (= d move_b4 when Q), d move_aft (Q)
  if (BWD) {
#pragma unroll
    for (int j = 0; j < J; ++j) acc_g[j] = acc_b[j] = acc_a[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int64_t rstride = (int64_t)gridDim.x * TY;
  int64_t r = (int64_t)blockIdx.x * TY + ty;
  const float q_tol = Q ? ofq_lsq_level_tol(a.qlo, a.qhi) : 0.f;
  const float q_hmt = 0.5f - q_tol;
  float4 xn[J], sn[J];          // x (and res / dy) of the next row
  float4 dn[J];
  auto issue = [&](int64_t rr) {
    const int64_t rres = (!BWD && a.res) ? ln_perm_row(rr, a.rperm, a.permN) : rr;
    const int64_t rqd = (BWD && Q) ? ln_perm_row(rr, a.qperm, a.permN) : rr;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      if (cok[j]) {
        const int64_t col = (tx + (int64_t)j * TXW) * 4;
        xn[j] = *reinterpret_cast<const float4*>(a.x + rr * a.ldx + col);
        if (!BWD && a.res) sn[j] = *reinterpret_cast<const float4*>(a.res + rres * a.ldx + col);
        if (BWD) {
          sn[j] = *reinterpret_cast<const float4*>(a.dy + rqd * a.ldy + col);
          if (a.dres) dn[j] = *reinterpret_cast<const float4*>(a.dres + rr * a.ldx + col);
        }
      }
    }
  };
  // forward: one-row software prefetch.  backward: no prefetch copy -- it holds three more accumulator/operand sets per
  // column and the extra registers cost a wave per SIMD, which hides more latency than the prefetch does
  if (!BWD && r < a.R) issue(r);
  for (; r < a.R; r += rstride) {
    if (BWD) issue(r);
    float4 xv[J], sv[J], dv[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
      xv[j] = cok[j] ? xn[j] : make_float4(0.f, 0.f, 0.f, 0.f);
      sv[j] = cok[j] ? sn[j] : make_float4(0.f, 0.f, 0.f, 0.f);
      if (BWD) dv[j] = cok[j] ? dn[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float mu = 0.f, rs = 0.f;
    if (BWD) { mu = a.mean[r]; rs = a.rstd[r]; }
    if (!BWD && r + rstride < a.R) issue(r + rstride);
    if (!BWD) {
      if (a.res) {
#pragma unroll
        for (int j = 0; j < J; ++j) {
          xv[j].x += sv[j].x; xv[j].y += sv[j].y; xv[j].z += sv[j].z; xv[j].w += sv[j].w;
        }
      }
      float s1 = 0.f;
#pragma unroll
      for (int j = 0; j < J; ++j) s1 += (xv[j].x + xv[j].y) + (xv[j].z + xv[j].w);
      mu = ln_row_sum<TXW>(s1) * invC;
      float s2 = 0.f;
#pragma unroll
      for (int j = 0; j < J; ++j) {
        if (!cok[j]) continue;
        const float d0 = xv[j].x - mu, d1 = xv[j].y - mu, d2 = xv[j].z - mu, d3 = xv[j].w - mu;
        s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
      }
      const float var = ln_row_sum<TXW>(s2) * invC;
      rs = 1.0f / sqrtf(var + a.eps);
      float q_al = 1.f, q_ra = 1.f;
      const int64_t rq = Q ? ln_perm_row(r, a.qperm, a.permN) : r;
      if (Q) {
        q_al = ofq_lsq_eff_scale(a.qs[rq % a.qS], a.qgscale);
        q_ra = ofq_div(1.0f, q_al);                       // correctly rounded reciprocal of the row's step
      }
#pragma unroll
      for (int j = 0; j < J; ++j) {
        if (!cok[j]) continue;
        const int64_t col = (tx + (int64_t)j * TXW) * 4;
        float4 o;
        o.x = (xv[j].x - mu) * rs * gam[j].x + bet[j].x;
        o.y = (xv[j].y - mu) * rs * gam[j].y + bet[j].y;
        o.z = (xv[j].z - mu) * rs * gam[j].z + bet[j].z;
        o.w = (xv[j].w - mu) * rs * gam[j].w + bet[j].w;
        if (!Q || a.y) {
          *reinterpret_cast<float4*>(a.y + r * a.ldy + col) = o;
          dxmax = ofq_absmax4(dxmax, o.x, o.y, o.z, o.w);          // forward: max |y| for a consumer on fp16 planes (the teacher's GEMMs)
        }
        if (a.res) *reinterpret_cast<float4*>(a.xs + r * a.ldx + col) = xv[j];
        if (Q) {
          // only the level leaves this kernel: x * fl(1/al) decides it unless the product sits within a few ulp of a
          // rounding boundary (common.h: ofq_lsq_level_rcp) -- then the four elements are redone with the IEEE division,
          // so the codes stay bit-identical at ~5 instead of ~16 VALU per element
          const float xe[4] = {__fadd_rn(o.x, qb4v[j].x), __fadd_rn(o.y, qb4v[j].y), __fadd_rn(o.z, qb4v[j].z),
                               __fadd_rn(o.w, qb4v[j].w)};
          float dmax = 0.f;
          float q0 = ofq_lsq_level_rcp_d(xe[0], q_ra, a.qlo, a.qhi, dmax);
          float q1 = ofq_lsq_level_rcp_d(xe[1], q_ra, a.qlo, a.qhi, dmax);
          float q2 = ofq_lsq_level_rcp_d(xe[2], q_ra, a.qlo, a.qhi, dmax);
          float q3 = ofq_lsq_level_rcp_d(xe[3], q_ra, a.qlo, a.qhi, dmax);
          if (!(dmax < q_hmt)) {
            q0 = ofq_lsq_level_exact(xe[0], q_al, a.qlo, a.qhi);
            q1 = ofq_lsq_level_exact(xe[1], q_al, a.qlo, a.qhi);
            q2 = ofq_lsq_level_exact(xe[2], q_al, a.qlo, a.qhi);
            q3 = ofq_lsq_level_exact(xe[3], q_al, a.qlo, a.qhi);
          }
          *reinterpret_cast<char4*>(a.qcodes + rq * a.C + col) =
              make_char4((signed char)(int)q0, (signed char)(int)q1, (signed char)(int)q2, (signed char)(int)q3);
        }
      }
      if (tx == 0) { a.mean[r] = mu; a.rstd[r] = rs; }
    } else {
      float4 xh[J], g[J];
      float sa = 0.f, sb = 0.f;
      if (Q) {       // sv holds the gradient of the quantised tensor: turn it into the gradient of n = LN(xs)
        const int64_t rq = ln_perm_row(r, a.qperm, a.permN);
        const float al = ofq_lsq_eff_scale(a.qs[rq % a.qS], a.qgscale);
        const float ral = ofq_div(1.0f, al);
        float rds = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
          if (!cok[j]) continue;
          const float xx[4] = {xv[j].x, xv[j].y, xv[j].z, xv[j].w}, gg[4] = {sv[j].x, sv[j].y, sv[j].z, sv[j].w};
          const float gm[4] = {gam[j].x, gam[j].y, gam[j].z, gam[j].w}, bt[4] = {bet[j].x, bet[j].y, bet[j].z, bet[j].w};
          const float b4[4] = {qb4v[j].x, qb4v[j].y, qb4v[j].z, qb4v[j].w};
          float dq[4], dsc[4], xq[4];
          OfqLsqFlags fl;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float n = (xx[e] - mu) * rs * gm[e] + bt[e];            // same expression as the forward
            xq[e] = __fadd_rn(n, b4[e]);
            ofq_lsq_bwd_fast(xq[e], gg[e], al, ral, a.qlo, a.qhi, fl, dq[e], dsc[e]);
          }
          if (ofq_lsq_flags_risky(fl, q_hmt, q_tol)) {       // an element within a few ulp of a rounding / range boundary: the IEEE divisions decide
#pragma unroll
            for (int e = 0; e < 4; ++e) ofq_lsq_bwd_exact(xq[e], gg[e], al, a.qlo, a.qhi, dq[e], dsc[e]);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) rds += dsc[e];
          acc_a[j].x += gg[0]; acc_a[j].y += gg[1]; acc_a[j].z += gg[2]; acc_a[j].w += gg[3];
          sv[j] = make_float4(dq[0], dq[1], dq[2], dq[3]);
        }
        rds = ln_row_sum<TXW>(rds);
        if (tx == 0) a.rowpart[rq] = rds;
      }
#pragma unroll
      for (int j = 0; j < J; ++j) {
        xh[j] = make_float4((xv[j].x - mu) * rs, (xv[j].y - mu) * rs, (xv[j].z - mu) * rs, (xv[j].w - mu) * rs);
        g[j] = make_float4(sv[j].x * gam[j].x, sv[j].y * gam[j].y, sv[j].z * gam[j].z, sv[j].w * gam[j].w);
        if (!cok[j]) continue;
        sa += (g[j].x + g[j].y) + (g[j].z + g[j].w);
        sb += (g[j].x * xh[j].x + g[j].y * xh[j].y) + (g[j].z * xh[j].z + g[j].w * xh[j].w);
        acc_g[j].x += sv[j].x * xh[j].x; acc_g[j].y += sv[j].y * xh[j].y;
        acc_g[j].z += sv[j].z * xh[j].z; acc_g[j].w += sv[j].w * xh[j].w;
        acc_b[j].x += sv[j].x; acc_b[j].y += sv[j].y; acc_b[j].z += sv[j].z; acc_b[j].w += sv[j].w;
      }
      const float ma = ln_row_sum<TXW>(sa) * invC, mb = ln_row_sum<TXW>(sb) * invC;
#pragma unroll
      for (int j = 0; j < J; ++j) {
        if (!cok[j]) continue;
        const int64_t col = (tx + (int64_t)j * TXW) * 4;
        float4 o;
        o.x = rs * (g[j].x - ma - xh[j].x * mb);
        o.y = rs * (g[j].y - ma - xh[j].y * mb);
        o.z = rs * (g[j].z - ma - xh[j].z * mb);
        o.w = rs * (g[j].w - ma - xh[j].w * mb);
        if (a.dres) { o.x += dv[j].x; o.y += dv[j].y; o.z += dv[j].z; o.w += dv[j].w; }
        *reinterpret_cast<float4*>(a.dx + r * a.ldx + col) = o;
        if (a.dx2) *reinterpret_cast<float4*>(a.dx2 + ln_perm_row(r, a.rperm, a.permN) * a.ldx + col) = o;
        dxmax = ofq_absmax4(dxmax, o.x, o.y, o.z, o.w);
      }
    }
  }
  if (a.amax) ofq_amax_publish(a.amax, dxmax);
  if (!BWD) return;
  // column partials: [TY][NACC][ncol] through LDS, summed over the row lanes in a fixed order
  constexpr int NACC = Q ? 3 : 2;
  const int ncol = TXW * J * 4;
  float* base = red + (size_t)ty * NACC * ncol;
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int c = (tx + j * TXW) * 4;
    base[c] = acc_g[j].x; base[c + 1] = acc_g[j].y; base[c + 2] = acc_g[j].z; base[c + 3] = acc_g[j].w;
    base[ncol + c] = acc_b[j].x; base[ncol + c + 1] = acc_b[j].y; base[ncol + c + 2] = acc_b[j].z; base[ncol + c + 3] = acc_b[j].w;
    if (Q) {
      base[2 * ncol + c] = acc_a[j].x; base[2 * ncol + c + 1] = acc_a[j].y;
      base[2 * ncol + c + 2] = acc_a[j].z; base[2 * ncol + c + 3] = acc_a[j].w;
    }
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < NACC * ncol; idx += 256) {
    const int c = idx % ncol, ac = idx / ncol;
    if (c >= a.C) continue;
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < TY; ++t) s += red[((size_t)t * NACC + ac) * ncol + c];
    a.colpart[((int64_t)blockIdx.x * NACC + ac) * a.C + c] = s;
  }
}

struct LnGeom { int TX, J, gx; };

static int ln_geom(int64_t R, int64_t C, LnGeom* g) {
  if (R <= 0 || C <= 0 || (C & 3) || C > 2048) return OFQ_EINVAL;
  const int64_t w4 = C / 4;
  g->TX = w4 <= 128 ? 32 : 64;
  g->J = (int)ceil_div(w4, g->TX);               // <= 4 (TX 32) or <= 8 (TX 64)
  const int TY = 256 / g->TX;
  int64_t gx = ceil_div(R, TY);
  if (gx > 1024) gx = 1024;
  g->gx = (int)gx;
  return 0;
}

template <bool BWD, bool Q>
static int ln_launch(const LnGeom& g, const LnArgs& a, hipStream_t st) {
  const size_t lds = BWD ? (size_t)(256 / g.TX) * (Q ? 3 : 2) * g.TX * g.J * 4 * sizeof(float) : 0;
  dim3 grid(g.gx), block(256);
#define LN_CASE(JJ, TXW) hipLaunchKernelGGL((layernorm_kernel<JJ, TXW, BWD, Q>), grid, block, lds, st, a); break
  if (g.TX == 32) {
    switch (g.J) {
      case 1: LN_CASE(1, 32);
      case 2: LN_CASE(2, 32);
      case 3: LN_CASE(3, 32);
      default: LN_CASE(4, 32);
    }
  } else {
    switch (g.J) {
      case 3: LN_CASE(3, 64);
      case 4: LN_CASE(4, 64);
      case 5: LN_CASE(5, 64);
      case 6: LN_CASE(6, 64);
      case 7: LN_CASE(7, 64);
      default: LN_CASE(8, 64);
    }
  }
#undef LN_CASE
  OFQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int ofq_layernorm_fwd(const float* x, const float* res, const float* gamma, const float* beta, float* y,
                                 float* xsum, float* mean, float* rstd, int64_t R, int64_t C, int64_t ldx, int64_t ldy,
                                 float eps, void* amax_out, ofq_stream_t stream) {
  if (!x || !y || !mean || !rstd || (res && !xsum)) return OFQ_EINVAL;
  if (ldx < C || ldy < C || (ldx & 3) || (ldy & 3)) return OFQ_EINVAL;
  LnGeom g;
  int rc = ln_geom(R, C, &g);
  if (rc) return rc;
  LnArgs a = {};
  a.x = x; a.res = res; a.gamma = gamma; a.beta = beta; a.y = y; a.xs = xsum; a.mean = mean; a.rstd = rstd;
  a.R = R; a.C = C; a.ldx = ldx; a.ldy = ldy; a.TX = g.TX; a.TY = 256 / g.TX; a.eps = eps;
  a.amax = (unsigned*)amax_out;
  return ln_launch<false, false>(g, a, (hipStream_t)stream);
}

extern "C" size_t ofq_layernorm_bwd_ws_bytes(int64_t R, int64_t C) {
  LnGeom g;
  if (ln_geom(R, C, &g)) return 0;
  return (size_t)g.gx * 2 * C * sizeof(float);
}

extern "C" int ofq_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                                 const float* dres, float* dx, float* dgamma, float* dbeta, int64_t R, int64_t C,
                                 int64_t ldx, int64_t ldy, void* ws, size_t ws_bytes, void* amax_out, ofq_stream_t stream) {
  if (!dy || !x || !mean || !rstd || !dx || !ws) return OFQ_EINVAL;
  if (ldx < C || ldy < C || (ldx & 3) || (ldy & 3)) return OFQ_EINVAL;
  LnGeom g;
  int rc = ln_geom(R, C, &g);
  if (rc) return rc;
  if (ws_bytes < ofq_layernorm_bwd_ws_bytes(R, C)) return OFQ_ENOWS;
  LnArgs a = {};
  a.x = x; a.gamma = gamma; a.mean = (float*)mean; a.rstd = (float*)rstd; a.dy = dy; a.dres = dres; a.dx = dx;
  a.colpart = (float*)ws; a.amax = (unsigned*)amax_out;
  a.R = R; a.C = C; a.ldx = ldx; a.ldy = ldy; a.TX = g.TX; a.TY = 256 / g.TX;
  hipStream_t st = (hipStream_t)stream;
  rc = ln_launch<true, false>(g, a, st);
  if (rc) return rc;
  if (dgamma || dbeta) {
    SumJobs jobs = {};
    if (dgamma) jobs.j[0] = {a.colpart, dgamma, C, g.gx, 2 * C, 1, 1.0f, 0, 0};
    if (dbeta) jobs.j[1] = {a.colpart + C, dbeta, C, g.gx, 2 * C, 1, 1.0f, 0, 0};
    strided_sum_launch(jobs, C, 2, st);
    OFQ_LAUNCH_CHECK();
  }
  return 0;
}

// ---- LayerNorm fused with the per-token LSQ that consumes its output (Block.norm1 -> attention input quantiser,
// Block.norm2 -> fc1's input quantiser: deit_vision_transformer.py:154-164 + qlinear.py:66-68 / attention.py:177)
static int ln_lsq_fwd(const float* x, const float* res, const float* gamma, const float* beta, float* y,
                      float* xsum, float* mean, float* rstd, int8_t* codes, const float* lsq_s, int64_t S,
                      float gscale, const float* b4, int lo, int hi, int64_t R, int64_t C, int64_t ldx,
                      float eps, const int* q_perm, const int* res_perm, int64_t perm_n, ofq_stream_t stream) {
  if (!x || !mean || !rstd || !codes || !lsq_s || S <= 0 || (res && !xsum)) return OFQ_EINVAL;
  if (ldx < C || (ldx & 3)) return OFQ_EINVAL;
  if ((q_perm || res_perm) && (perm_n <= 0 || perm_n >= (1ll << 31) || R % perm_n)) return OFQ_EINVAL;
  if (q_perm && perm_n % S) return OFQ_EINVAL;             // the step index is (row within the image) % S
  LnGeom g;
  int rc = ln_geom(R, C, &g);
  if (rc) return rc;
  LnArgs a = {};
  a.x = x; a.res = res; a.gamma = gamma; a.beta = beta; a.y = y; a.xs = xsum; a.mean = mean; a.rstd = rstd;
  a.R = R; a.C = C; a.ldx = ldx; a.ldy = C; a.TX = g.TX; a.TY = 256 / g.TX; a.eps = eps;
  a.qs = lsq_s; a.qS = S; a.qgscale = gscale; a.qb4 = b4; a.qlo = (float)lo; a.qhi = (float)hi; a.qcodes = codes;
  a.qperm = q_perm; a.rperm = res ? res_perm : nullptr; a.permN = (int)perm_n;
  return ln_launch<false, true>(g, a, (hipStream_t)stream);
}
extern "C" int ofq_layernorm_lsq_fwd(const float* x, const float* res, const float* gamma, const float* beta, float* y,
                                     float* xsum, float* mean, float* rstd, int8_t* codes, const float* lsq_s, int64_t S,
                                     float gscale, const float* b4, int lo, int hi, int64_t R, int64_t C, int64_t ldx,
                                     float eps, ofq_stream_t stream) {
  return ln_lsq_fwd(x, res, gamma, beta, y, xsum, mean, rstd, codes, lsq_s, S, gscale, b4, lo, hi, R, C, ldx, eps, nullptr, nullptr,
                    0, stream);
}
// The same with token permutations folded in (Swin: LayerNorm -> shifted-window partition -> input quantiser, and the window
// reverse in front of the residual add, swin_attention_and_mlp.py:312-323 / swin.py:160-170).  The R rows are images of perm_n
// tokens; q_perm[t] = row (within its image) of token t's codes, the step being lsq_s[that row % S]; res_perm[t] = row of token
// t in `res`; either may be NULL (identity).  xsum / mean / rstd stay in token order.
extern "C" int ofq_layernorm_lsq_fwd_perm(const float* x, const float* res, const float* gamma, const float* beta, float* y,
                                          float* xsum, float* mean, float* rstd, int8_t* codes, const float* lsq_s, int64_t S,
                                          float gscale, const float* b4, int lo, int hi, int64_t R, int64_t C, int64_t ldx,
                                          float eps, const int* q_perm, const int* res_perm, int64_t perm_n, ofq_stream_t stream) {
  if (y && q_perm) return OFQ_EINVAL;         // (the un-quantised LayerNorm output is written in token order only)
  return ln_lsq_fwd(x, res, gamma, beta, y, xsum, mean, rstd, codes, lsq_s, S, gscale, b4, lo, hi, R, C, ldx, eps, q_perm, res_perm,
                    perm_n, stream);
}

extern "C" size_t ofq_layernorm_lsq_bwd_ws_bytes(int64_t R, int64_t C) {
  LnGeom g;
  if (ln_geom(R, C, &g)) return 0;
  return ((size_t)g.gx * 3 * C + (size_t)R) * sizeof(float);
}

static int ln_lsq_bwd(const float* gq, const float* x, const float* mean, const float* rstd, const float* gamma,
                      const float* beta, const float* dres, const float* lsq_s, int64_t S, float gscale,
                      const float* b4, int lo, int hi, float* dx, float* dgamma, float* dbeta, float* db4, float* ds,
                      float* dbaft, int64_t R, int64_t C, int64_t ldx, int64_t ldg, void* ws, size_t ws_bytes,
                      void* amax_out, const int* q_perm, const int* res_perm, int64_t perm_n, float* dres_out,
                      ofq_stream_t stream) {
  if (!gq || !x || !mean || !rstd || !dx || !ws || !lsq_s || S <= 0 || R % S) return OFQ_EINVAL;
  if (ldx < C || ldg < C || (ldx & 3) || (ldg & 3)) return OFQ_EINVAL;
  if ((q_perm || res_perm) && (perm_n <= 0 || perm_n >= (1ll << 31) || R % perm_n)) return OFQ_EINVAL;
  if (q_perm && perm_n % S) return OFQ_EINVAL;             // the step index is (row within the image) % S
  LnGeom g;
  int rc = ln_geom(R, C, &g);
  if (rc) return rc;
  if (ws_bytes < ofq_layernorm_lsq_bwd_ws_bytes(R, C)) return OFQ_ENOWS;
  LnArgs a = {};
  a.x = x; a.gamma = gamma; a.beta = beta; a.mean = (float*)mean; a.rstd = (float*)rstd; a.dy = gq; a.dres = dres; a.dx = dx;
  a.colpart = (float*)ws; a.rowpart = (float*)ws + (size_t)g.gx * 3 * C; a.amax = (unsigned*)amax_out;
  a.R = R; a.C = C; a.ldx = ldx; a.ldy = ldg; a.TX = g.TX; a.TY = 256 / g.TX;
  a.qs = lsq_s; a.qS = S; a.qgscale = gscale; a.qb4 = b4; a.qlo = (float)lo; a.qhi = (float)hi;
  a.qperm = q_perm; a.rperm = res_perm; a.dx2 = dres_out; a.permN = (int)perm_n;
  hipStream_t st = (hipStream_t)stream;
  rc = ln_launch<true, true>(g, a, st);
  if (rc) return rc;
  // every second-stage sum of this call in ONE launch (they are a few microseconds each and latency-bound): the column
  // sums, the step gradient, and db4 -- the same column sum as dbeta, written a second time so that the two parameters
  // get separate gradient tensors without a copy kernel
  SumJobs jobs = {};
  int nj = 0;
  if (dgamma) jobs.j[nj++] = {a.colpart, dgamma, C, g.gx, 3 * C, 1, 1.0f, 0, 0};
  if (dbeta) jobs.j[nj++] = {a.colpart + C, dbeta, C, g.gx, 3 * C, 1, 1.0f, 0, 0};
  if (db4) jobs.j[nj++] = {a.colpart + C, db4, C, g.gx, 3 * C, 1, 1.0f, 0, 0};
  if (dbaft) jobs.j[nj++] = {a.colpart + 2 * C, dbaft, C, g.gx, 3 * C, 1, 1.0f, 0, 0};
  if (ds) jobs.j[nj++] = {a.rowpart, ds, S, R / S, S, 1, gscale, 0, 0};
  if (nj) {
    strided_sum_launch(jobs, C > S ? C : S, nj, st);
    OFQ_LAUNCH_CHECK();
  }
  return 0;
}
extern "C" int ofq_layernorm_lsq_bwd(const float* gq, const float* x, const float* mean, const float* rstd, const float* gamma,
                                     const float* beta, const float* dres, const float* lsq_s, int64_t S, float gscale,
                                     const float* b4, int lo, int hi, float* dx, float* dgamma, float* dbeta, float* db4, float* ds,
                                     float* dbaft, int64_t R, int64_t C, int64_t ldx, int64_t ldg, void* ws, size_t ws_bytes,
                                     void* amax_out, ofq_stream_t stream) {
  return ln_lsq_bwd(gq, x, mean, rstd, gamma, beta, dres, lsq_s, S, gscale, b4, lo, hi, dx, dgamma, dbeta, db4, ds, dbaft, R, C, ldx,
                    ldg, ws, ws_bytes, amax_out, nullptr, nullptr, 0, nullptr, stream);
}
// Backward of ofq_layernorm_lsq_fwd_perm: gq (and the step-gradient partials) are read / indexed at row q_perm[t] of their image;
// dres_out (optional, with res_perm): the rows of dx once more, token t at row res_perm[t] -- the gradient of the forward's
// permuted `res` operand, in the order its producer works in.  dx itself, dres, mean, rstd, x: token order.
extern "C" int ofq_layernorm_lsq_bwd_perm(const float* gq, const float* x, const float* mean, const float* rstd, const float* gamma,
                                          const float* beta, const float* dres, const float* lsq_s, int64_t S, float gscale,
                                          const float* b4, int lo, int hi, float* dx, float* dgamma, float* dbeta, float* db4,
                                          float* ds, float* dbaft, int64_t R, int64_t C, int64_t ldx, int64_t ldg, void* ws,
                                          size_t ws_bytes, void* amax_out, const int* q_perm, const int* res_perm, int64_t perm_n,
                                          float* dres_out, ofq_stream_t stream) {
  if (dres_out && !res_perm) return OFQ_EINVAL;
  return ln_lsq_bwd(gq, x, mean, rstd, gamma, beta, dres, lsq_s, S, gscale, b4, lo, hi, dx, dgamma, dbeta, db4, ds, dbaft, R, C, ldx,
                    ldg, ws, ws_bytes, amax_out, q_perm, dres_out ? res_perm : nullptr, perm_n, dres_out, stream);
}
