// K3/K4/K5  LSQ activation quantiser fused with its LearnableBias sandwich (and the optional exact-GELU
// prologue of QMLP): reference lsq.py:571-602 / :757-792, qbias.py:9-13, qlinear.py:66-68, :127.
//
// Geometry.  x is [R = outer*S rows][inner]; a workgroup is TY row-groups of TX lanes, every lane owns J
// float4 column groups (TX*J*4 >= the column tile), so a lane's bias columns never change while it walks
// rows with stride gridDim.x*TY.  That makes the three backward reductions register-resident:
//   - per-scale ds (row mode): lane partial -> TX-lane shuffle reduction -> one float per (row, slot)
//   - db4 / dbaft (and ds in channel mode): J*4 accumulators per lane across all of its rows ->
//     one LDS pass per workgroup -> partial[gridDim.x][bias_len]
// and a tiny second kernel finishes them in a fixed order (deterministic, no atomics).
// HBM traffic: fwd 4 B read + 4 B write per element; bwd 8 B read + 4 B write per element.
#include "common.h"

struct LsqGeom {
  int TX, TY, J, gx, gy, k, nslot, widek;   // widek: more bias phases than row-groups per workgroup
};

static int lsq_geom(int64_t R, int64_t inner, int64_t bias_len, LsqGeom* g, bool bwd = false) {
  if (inner <= 0 || (inner & 3) || R <= 0) return OFQ_EINVAL;
  int64_t k = (bias_len > 0) ? bias_len / inner : 1;
  if (bias_len > 0 && k * inner != bias_len) return OFQ_EINVAL;
  int64_t w4 = inner / 4;
  // backward keeps three accumulator sets per float4 column: at more than two float4 per lane the kernel drops to two
  // waves per SIMD and loses the loads in flight it needs (tools/lsq_bench.py: 58 -> 38 us on 25216 x 384)
  const int maxj = bwd ? 2 : 4;
  int TX = 16;
  while (TX < 256 && ceil_div(w4, TX) > 4) TX <<= 1;
  int J = (int)ceil_div(w4, TX);
  int gy = 1;
  if (J > 4) {  // very wide rows (the 224x224 image planes): tile the columns
    J = 4;
    gy = (int)ceil_div(w4, (int64_t)TX * J);
  }
  if (J > maxj) {
    // fewer float4 per lane: the largest power-of-two lane count (<= 128) that tiles the row exactly, columns tiled by gy
    int tx = 128;
    while (tx > 16 && (w4 % tx)) tx >>= 1;
    if (w4 % tx == 0) {
      TX = tx;
      J = maxj;
      gy = (int)ceil_div(w4, (int64_t)TX * J);
    }
  }
  int TY = 256 / TX;
  int widek = 0;
  if (k > 1) {
    if (k > TY) widek = 1;                       // e.g. Swin stage 3/4: 12 / 24 heads but only 8 / 4 row-groups
    else TY = (TY / (int)k) * (int)k;
  }
  int64_t gx = ceil_div(R, TY);
  int64_t cap = 1024 / gy;                       // (the wide-k partials [gx*TY][nacc][inner] stay below ~50 MB)
  if (cap < 1) cap = 1;
  if (gx > cap) gx = cap;
  if (widek) {                                   // the row stride gx*TY must keep every lane on one bias phase
    int64_t a = k, b = TY;
    while (b) { int64_t t = a % b; a = b; b = t; }
    const int64_t mult = k / a;                  // k / gcd(k, TY)
    gx = ceil_div(gx, mult) * mult;
  }
  g->TX = TX; g->TY = TY; g->J = J; g->gx = (int)gx; g->gy = gy; g->k = (int)k; g->widek = widek;
  g->nslot = TX >= 64 ? TX / 64 : 1;
  return 0;
}

struct LsqArgs {
  const float* x; const float* g; const float* s; const float* b4; const float* baft;
  float* y; float* dx; int8_t* codes;
  unsigned* amax;   // backward, optional: bits of max |dx| (see ofq_amax_publish)
  float* rowpart;   // [R][gy*nslot]
  float* colpart;   // [gx][nacc][k*inner]
  int64_t R, S, inner, ldx, ldy;
  int k, TX, TY, colmode, prologue, nacc, widek;
  float lo, hi, gscale;
  // patch layout (round 6; the W8A8 stem, qlinear.py:166-174: image quantiser -> stride == kernel conv): rows are (image, channel),
  // inner = H * W pixels; the forward's y / codes and the backward's gy live in im2col order instead -- row (image, patch),
  // column (channel, pixel of the patch) -- so the conv's GEMM operands need no permute copy either way.  im_cin == 0: off.
  int im_cin, im_w, im_ph, im_pw, im_gw, im_gh;
};

// offset of pixel `col` (a multiple of 4; W and the patch width are multiples of 4 as well: the four pixels stay together) of row
// (image b, channel c) in the patch layout, split into the part that depends on the pixel and the part that depends on the row
__device__ __forceinline__ int64_t lsq_im_pix(const LsqArgs& a, int64_t col) {
  const int y = (int)(col / a.im_w), x = (int)(col - (int64_t)y * a.im_w);
  const int64_t K = (int64_t)a.im_cin * a.im_ph * a.im_pw;
  return ((int64_t)(y / a.im_ph) * a.im_gw + x / a.im_pw) * K + (y % a.im_ph) * a.im_pw + (x % a.im_pw);
}
__device__ __forceinline__ int64_t lsq_im_row(const LsqArgs& a, int64_t r) {
  const int64_t b = r / a.im_cin, c = r - b * a.im_cin;
  return b * ((int64_t)a.im_gh * a.im_gw * a.im_cin * a.im_ph * a.im_pw) + c * (a.im_ph * a.im_pw);
}

#ifndef LSQ_WAVES_PER_EU
#define LSQ_WAVES_PER_EU
#endif
template <int J, bool BWD, bool COLMODE, bool GELU>
__global__ __launch_bounds__(256) LSQ_WAVES_PER_EU void lsq_kernel(LsqArgs a) {
  extern __shared__ __attribute__((aligned(16))) float red[];
  const int TX = a.TX, TY = a.TY;
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  const int64_t w4 = a.inner / 4;
  const int64_t c4base = (int64_t)blockIdx.y * TX * J;
  const int ph = (a.k > 1) ? (int)(((int64_t)blockIdx.x * TY + ty) % a.k) : 0;
  const bool active_row_group = ty < TY;

  // per-lane column state
  float4 b4v[J], bav[J], sv[J];
  bool cok[J];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    int64_t c4 = c4base + tx + (int64_t)j * TX;
    cok[j] = c4 < w4;
    b4v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    bav[j] = b4v[j];
    sv[j] = make_float4(1.f, 1.f, 1.f, 1.f);
    if (cok[j]) {
      int64_t bo = (int64_t)ph * a.inner + c4 * 4;
      if (a.b4) b4v[j] = *reinterpret_cast<const float4*>(a.b4 + bo);
      if (!BWD && a.baft) bav[j] = *reinterpret_cast<const float4*>(a.baft + bo);
      if (COLMODE) {
        float4 t = *reinterpret_cast<const float4*>(a.s + c4 * 4);
        sv[j] = make_float4(ofq_lsq_eff_scale(t.x, a.gscale), ofq_lsq_eff_scale(t.y, a.gscale),
                            ofq_lsq_eff_scale(t.z, a.gscale), ofq_lsq_eff_scale(t.w, a.gscale));
      }
    }
  }
  float acc_b4[J][4], acc_ba[J][4], acc_ds[J][4];
  if (BWD) {
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc_b4[j][e] = acc_ba[j][e] = acc_ds[j][e] = 0.f;
  }
  const float lo = a.lo, hi = a.hi;
  const int nslot = TX >= 64 ? TX / 64 : 1;
  float dxmax = 0.f;

  if (active_row_group) {
    // software-pipelined row walk: the next row's loads are issued before the current row is processed, so a wave
    // always has 2 rows (J float4 per tensor each) in flight and the stores never sit between a load and its use
    // The loads of a row (x, g and the row's step) are issued together, unconditionally -- lanes past the width and
    // the row past the end read a clamped, valid address -- and consumed together one iteration later.  Guards around
    // them (`if (cok[j])`, `if (r + rstride < R)`) and the step loaded by itself inside the iteration made the
    // compiler wait for every outstanding load right after issuing the next row's (s_waitcnt vmcnt(0)).
    const int64_t rstride = (int64_t)gridDim.x * TY;
    int64_t r = (int64_t)blockIdx.x * TY + ty;
    float4 xn[J], gn[J];
    float sn = 1.f;
    int64_t ccol[J];
#pragma unroll
    for (int j = 0; j < J; ++j) ccol[j] = cok[j] ? (c4base + tx + (int64_t)j * TX) * 4 : 0;
    const bool im = a.im_cin != 0;
    int64_t impix[J];                                   // patch layout: the pixel's part of the offset (constant per lane)
#pragma unroll
    for (int j = 0; j < J; ++j) impix[j] = im ? lsq_im_pix(a, ccol[j]) : 0;
    const int smod_step = (int)(rstride % a.S);
    int smod = (int)(r % a.S);                         // r mod S, kept incrementally
    auto issue = [&](int64_t rr, int sm) {
      const int64_t grow = (BWD && im) ? lsq_im_row(a, rr) : 0;
#pragma unroll
      for (int j = 0; j < J; ++j) {
        xn[j] = *reinterpret_cast<const float4*>(a.x + rr * a.ldx + ccol[j]);
        if (BWD) gn[j] = *reinterpret_cast<const float4*>(a.g + (im ? grow + impix[j] : rr * a.ldy + ccol[j]));
      }
      if (!COLMODE) sn = a.s[sm];
    };
    if (r < a.R) issue(r, smod);
    for (; r < a.R; r += rstride) {
      float arow = 1.f;
      if (!COLMODE) arow = ofq_lsq_eff_scale(sn, a.gscale);
      float4 xv[J], gv[J];
#pragma unroll
      for (int j = 0; j < J; ++j) {
        xv[j] = xn[j];
        if (BWD) gv[j] = gn[j];
      }
      {
        int sm2 = smod + smod_step;
        sm2 -= (sm2 >= a.S) ? a.S : 0;
        smod = sm2;
        const int64_t rn = r + rstride;
        issue(rn < a.R ? rn : r, smod);
      }
      float rowds = 0.f;
#pragma unroll
      for (int j = 0; j < J; ++j) {
        if (!cok[j]) continue;
        const int64_t col = (c4base + tx + (int64_t)j * TX) * 4;
        float xin[4] = {xv[j].x, xv[j].y, xv[j].z, xv[j].w};
        float bb[4] = {b4v[j].x, b4v[j].y, b4v[j].z, b4v[j].w};
        float sc[4] = {sv[j].x, sv[j].y, sv[j].z, sv[j].w};
        if (!BWD) {
          float ba[4] = {bav[j].x, bav[j].y, bav[j].z, bav[j].w};
          float out[4];
          signed char cd[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float xe = GELU ? ofq_gelu(xin[e]) : xin[e];
            float al = COLMODE ? sc[e] : arow;
            float q, v;
            float yi = ofq_lsq_quant(__fadd_rn(xe, bb[e]), al, lo, hi, q, v);
            out[e] = __fadd_rn(__fmul_rn(yi, al), ba[e]);
            cd[e] = (signed char)(int)q;   // low 8 bits: int8 for signed ranges, uint8 for unsigned
          }
          const int64_t yo = im ? lsq_im_row(a, r) + impix[j] : r * a.ldy + col;
          const int64_t co = im ? yo : r * a.inner + col;
          if (a.y) *reinterpret_cast<float4*>(a.y + yo) = make_float4(out[0], out[1], out[2], out[3]);
          if (a.codes) *reinterpret_cast<char4*>(a.codes + co) = make_char4(cd[0], cd[1], cd[2], cd[3]);
        } else {
          float ge[4] = {gv[j].x, gv[j].y, gv[j].z, gv[j].w};
          float dxo[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float xe = GELU ? ofq_gelu(xin[e]) : xin[e];
            float al = COLMODE ? sc[e] : arow;
            float q, v;
            ofq_lsq_quant(__fadd_rn(xe, bb[e]), al, lo, hi, q, v);
            bool inr = (v >= lo) && (v <= hi);
            float dq = inr ? ofq_div(__fmul_rn(ge[e], al), al) : 0.f;   // autograd order: (g*a)/a
            float dsc = ge[e] * (inr ? (q - v) : q);                    // q == clamp(v) when out of range
            acc_b4[j][e] += dq;
            acc_ba[j][e] += ge[e];
            if (COLMODE) acc_ds[j][e] += dsc; else rowds += dsc;
            dxo[e] = GELU ? dq * ofq_gelu_grad(xin[e]) : dq;
          }
          *reinterpret_cast<float4*>(a.dx + r * a.ldx + col) = make_float4(dxo[0], dxo[1], dxo[2], dxo[3]);
          dxmax = ofq_absmax4(dxmax, dxo[0], dxo[1], dxo[2], dxo[3]);
        }
      }
      if (BWD && !COLMODE) {
        int w = TX < 64 ? TX : 64;
        for (int o = w / 2; o > 0; o >>= 1) rowds += __shfl_xor(rowds, o, 64);
        if ((tx & 63) == 0 || (TX < 64 && tx == 0))
          a.rowpart[(r * gridDim.y + blockIdx.y) * nslot + (tx >> 6)] = rowds;
      }
    }
  }
  if (!BWD) return;
  if (a.amax) ofq_amax_publish(a.amax, dxmax);

  const int nacc = a.nacc;
  if (a.widek) {
    // more phases than row-groups: every row-group writes its own partial, [gridDim.x*TY][nacc][inner]; the phase of
    // row-group G is G % k, resolved by the second stage
    if (active_row_group) {
      float* dst = a.colpart + ((int64_t)blockIdx.x * TY + ty) * nacc * a.inner;
#pragma unroll
      for (int j = 0; j < J; ++j) {
        if (!cok[j]) continue;
        const int64_t col = (c4base + tx + (int64_t)j * TX) * 4;
        *reinterpret_cast<float4*>(dst + col) = make_float4(acc_b4[j][0], acc_b4[j][1], acc_b4[j][2], acc_b4[j][3]);
        *reinterpret_cast<float4*>(dst + a.inner + col) = make_float4(acc_ba[j][0], acc_ba[j][1], acc_ba[j][2], acc_ba[j][3]);
        if (nacc == 3)
          *reinterpret_cast<float4*>(dst + 2 * a.inner + col) = make_float4(acc_ds[j][0], acc_ds[j][1], acc_ds[j][2], acc_ds[j][3]);
      }
    }
    return;
  }
  // ---- column partials: one LDS pass per workgroup, reduced over the row-groups of equal phase ----
  const int ncol = TX * J * 4;                 // columns of this column tile (incl. masked ones)
  float* base = red + ((size_t)ty * nacc) * ncol;
  if (active_row_group) {
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int c = (tx + j * TX) * 4 + e;
        base[c] = acc_b4[j][e];
        base[ncol + c] = acc_ba[j][e];
        if (nacc == 3) base[2 * ncol + c] = acc_ds[j][e];
      }
  }
  __syncthreads();
  const int kk = a.k;
  const int64_t blen = (int64_t)kk * a.inner;
  for (int idx = threadIdx.x; idx < kk * nacc * ncol; idx += blockDim.x) {
    int c = idx % ncol;
    int ac = (idx / ncol) % nacc;
    int p = idx / (ncol * nacc);
    int64_t col = c4base * 4 + c;
    if (col >= a.inner) continue;
    float sum = 0.f;
    for (int t = p; t < TY; t += kk) sum += red[((size_t)t * nacc + ac) * ncol + c];
    a.colpart[((int64_t)blockIdx.x * nacc + ac) * blen + (int64_t)p * a.inner + col] = sum;
  }
}

static size_t lsq_lds_bytes(const LsqGeom& g, int nacc) {
  return (size_t)g.TY * nacc * g.TX * g.J * 4 * sizeof(float);
}

template <bool BWD>
static int lsq_launch(const LsqGeom& g, const LsqArgs& a, hipStream_t st) {
  dim3 grid(g.gx, g.gy), block(g.TX * g.TY);
  size_t lds = BWD ? lsq_lds_bytes(g, a.nacc) : 0;
  const bool cm = a.colmode != 0, ge = a.prologue == 1;
#define LSQ_LAUNCH_J(JJ)                                                                                     \
  do {                                                                                                      \
    if (cm && ge) hipLaunchKernelGGL((lsq_kernel<JJ, BWD, true, true>), grid, block, lds, st, a);           \
    else if (cm) hipLaunchKernelGGL((lsq_kernel<JJ, BWD, true, false>), grid, block, lds, st, a);           \
    else if (ge) hipLaunchKernelGGL((lsq_kernel<JJ, BWD, false, true>), grid, block, lds, st, a);           \
    else hipLaunchKernelGGL((lsq_kernel<JJ, BWD, false, false>), grid, block, lds, st, a);                  \
  } while (0)
  switch (g.J) {
    case 1: LSQ_LAUNCH_J(1); break;
    case 2: LSQ_LAUNCH_J(2); break;
    case 3: LSQ_LAUNCH_J(3); break;
    default: LSQ_LAUNCH_J(4); break;
  }
#undef LSQ_LAUNCH_J
  OFQ_LAUNCH_CHECK();
  return 0;
}

struct LsqPatch { int width, ph, pw; };        // width 0: off

static int lsq_patch_fill(LsqArgs& a, const LsqPatch& pt, int64_t S, int64_t inner, int64_t ldx, int64_t ldy, int scale_mode) {
  if (pt.width == 0) return 0;
  if (pt.width <= 0 || pt.ph <= 0 || pt.pw <= 0 || (pt.width & 3) || (pt.pw & 3) || inner % pt.width || pt.width % pt.pw ||
      (inner / pt.width) % pt.ph || ldx != inner || ldy != inner || scale_mode != 0 || S >= (1 << 20))
    return OFQ_EINVAL;
  a.im_cin = (int)S; a.im_w = pt.width; a.im_ph = pt.ph; a.im_pw = pt.pw;
  a.im_gw = pt.width / pt.pw; a.im_gh = (int)(inner / pt.width) / pt.ph;
  return 0;
}

static int lsq_fwd_impl(const float* x, const float* s, const float* b4, const float* baft, float* y,
                        int8_t* codes, int64_t outer, int64_t S, int64_t inner, int64_t ldx, int64_t ldy,
                        int64_t bias_len, int scale_mode, int lo, int hi, float gscale, int prologue, LsqPatch pt,
                        ofq_stream_t stream) {
  if (!x || !s || (!y && !codes) || outer <= 0 || S <= 0) return OFQ_EINVAL;
  if (scale_mode == 1 && S != 1) return OFQ_EINVAL;
  if ((b4 || baft) && bias_len <= 0) return OFQ_EINVAL;
  if (ldx < inner || ldy < inner || (ldx & 3) || (ldy & 3)) return OFQ_EINVAL;
  LsqGeom g;
  int rc = lsq_geom(outer * S, inner, (b4 || baft) ? bias_len : 0, &g);
  if (rc) return rc;
  LsqArgs a = {};
  a.x = x; a.s = s; a.b4 = b4; a.baft = baft; a.y = y; a.codes = codes;
  a.R = outer * S; a.S = S; a.inner = inner; a.ldx = ldx; a.ldy = ldy; a.k = g.k; a.TX = g.TX; a.TY = g.TY;
  a.colmode = scale_mode; a.prologue = prologue; a.nacc = 0;
  a.lo = (float)lo; a.hi = (float)hi; a.gscale = gscale;
  rc = lsq_patch_fill(a, pt, S, inner, ldx, ldy, scale_mode);
  if (rc) return rc;
  return lsq_launch<false>(g, a, (hipStream_t)stream);
}
extern "C" int ofq_lsq_fwd(const float* x, const float* s, const float* b4, const float* baft, float* y,
                           int8_t* codes, int64_t outer, int64_t S, int64_t inner, int64_t ldx, int64_t ldy,
                           int64_t bias_len, int scale_mode, int lo, int hi, float gscale, int prologue,
                           ofq_stream_t stream) {
  return lsq_fwd_impl(x, s, b4, baft, y, codes, outer, S, inner, ldx, ldy, bias_len, scale_mode, lo, hi, gscale, prologue,
                      LsqPatch{0, 0, 0}, stream);
}
// The image quantiser of the W8A8 patch embedding with its output in patch (im2col) order: x is [images][channels][H * W]
// (outer images, S channels: one step per channel, per-pixel offsets of length H * W), y / codes are [images * gh * gw]
// [channels * ph * pw] -- the operand layout of the stride == kernel convolution's GEMM (qlinear.py:166-174).
extern "C" int ofq_lsq_fwd_patch(const float* x, const float* s, const float* b4, const float* baft, float* y, int8_t* codes,
                                 int64_t images, int64_t channels, int64_t pixels, int64_t bias_len, int lo, int hi, float gscale,
                                 int width, int ph, int pw, ofq_stream_t stream) {
  if (width <= 0) return OFQ_EINVAL;
  return lsq_fwd_impl(x, s, b4, baft, y, codes, images, channels, pixels, pixels, pixels, bias_len, 0, lo, hi, gscale, 0,
                      LsqPatch{width, ph, pw}, stream);
}

static void lsq_ws_layout(const LsqGeom& g, int64_t R, int64_t inner, int nacc, size_t* row_floats,
                          size_t* col_floats) {
  *row_floats = (size_t)R * g.gy * g.nslot;
  *col_floats = g.widek ? (size_t)g.gx * g.TY * nacc * inner : (size_t)g.gx * nacc * g.k * inner;
}

extern "C" size_t ofq_lsq_bwd_ws_bytes(int64_t outer, int64_t S, int64_t inner, int64_t bias_len, int scale_mode) {
  LsqGeom g;
  if (lsq_geom(outer * S, inner, bias_len, &g, true)) return 0;
  size_t rf, cf;
  lsq_ws_layout(g, outer * S, inner, scale_mode ? 3 : 2, &rf, &cf);
  return (rf + cf) * sizeof(float) + 256;
}

static int lsq_bwd_impl(const float* gy, const float* x, const float* s, const float* b4, float* dx, float* ds,
                        float* db4, float* dbaft, int64_t outer, int64_t S, int64_t inner, int64_t ldx,
                        int64_t ldy, int64_t bias_len, int scale_mode, int lo, int hi, float gscale, int prologue,
                        void* ws, size_t ws_bytes, void* amax_out, LsqPatch pt, ofq_stream_t stream) {
  if (!gy || !x || !s || !dx || !ws || outer <= 0 || S <= 0) return OFQ_EINVAL;
  if (scale_mode == 1 && S != 1) return OFQ_EINVAL;
  if (ldx < inner || ldy < inner || (ldx & 3) || (ldy & 3)) return OFQ_EINVAL;
  LsqGeom g;
  int rc = lsq_geom(outer * S, inner, bias_len, &g, true);
  if (rc) return rc;
  const int nacc = scale_mode ? 3 : 2;
  size_t rf, cf;
  lsq_ws_layout(g, outer * S, inner, nacc, &rf, &cf);
  if (ws_bytes < (rf + cf) * sizeof(float)) return OFQ_ENOWS;
  LsqArgs a = {};
  a.x = x; a.g = gy; a.s = s; a.b4 = b4; a.dx = dx; a.amax = (unsigned*)amax_out;
  a.rowpart = (float*)ws; a.colpart = (float*)ws + rf;
  a.R = outer * S; a.S = S; a.inner = inner; a.ldx = ldx; a.ldy = ldy; a.k = g.k; a.TX = g.TX; a.TY = g.TY;
  a.colmode = scale_mode; a.prologue = prologue; a.nacc = nacc; a.widek = g.widek;
  a.lo = (float)lo; a.hi = (float)hi; a.gscale = gscale;
  rc = lsq_patch_fill(a, pt, S, inner, ldx, ldy, scale_mode);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  rc = lsq_launch<true>(g, a, st);
  if (rc) return rc;
  const int64_t blen = (int64_t)g.k * inner;
  SumJobs jobs = {};
  int64_t maxc = 0;
  if (g.widek) {
    // partial row-groups [G][nacc][inner], G = gx*TY, phase(G) = G % k: view as [G/k][k][nacc][inner]
    const int64_t G = (int64_t)g.gx * g.TY, rs = (int64_t)g.k * nacc * inner, cm = (int64_t)nacc * inner;
    if (ds) {
      if (scale_mode) return OFQ_EINVAL;
      jobs.j[0] = {a.rowpart, ds, S, outer, S * (int64_t)g.gy * g.nslot, g.gy * g.nslot, gscale, 0, 0};
      maxc = S;
    }
    if (db4) { jobs.j[1] = {a.colpart, db4, blen, G / g.k, rs, 1, 1.0f, inner, cm}; if (blen > maxc) maxc = blen; }
    if (dbaft) { jobs.j[2] = {a.colpart + inner, dbaft, blen, G / g.k, rs, 1, 1.0f, inner, cm}; if (blen > maxc) maxc = blen; }
  } else {
    if (ds) {
      if (scale_mode) jobs.j[0] = {a.colpart + 2 * blen, ds, inner, g.gx, (int64_t)nacc * blen, 1, gscale, 0, 0};
      else jobs.j[0] = {a.rowpart, ds, S, outer, S * (int64_t)g.gy * g.nslot, g.gy * g.nslot, gscale, 0, 0};
      maxc = jobs.j[0].ncols;
    }
    if (db4) { jobs.j[1] = {a.colpart, db4, blen, g.gx, (int64_t)nacc * blen, 1, 1.0f, 0, 0}; if (blen > maxc) maxc = blen; }
    if (dbaft) { jobs.j[2] = {a.colpart + blen, dbaft, blen, g.gx, (int64_t)nacc * blen, 1, 1.0f, 0, 0}; if (blen > maxc) maxc = blen; }
  }
  if (maxc > 0) {
    strided_sum_launch(jobs, maxc, 3, st);
    OFQ_LAUNCH_CHECK();
  }
  return 0;
}
extern "C" int ofq_lsq_bwd(const float* gy, const float* x, const float* s, const float* b4, float* dx, float* ds,
                           float* db4, float* dbaft, int64_t outer, int64_t S, int64_t inner, int64_t ldx,
                           int64_t ldy, int64_t bias_len, int scale_mode, int lo, int hi, float gscale, int prologue,
                           void* ws, size_t ws_bytes, void* amax_out, ofq_stream_t stream) {
  return lsq_bwd_impl(gy, x, s, b4, dx, ds, db4, dbaft, outer, S, inner, ldx, ldy, bias_len, scale_mode, lo, hi, gscale, prologue,
                      ws, ws_bytes, amax_out, LsqPatch{0, 0, 0}, stream);
}
// Backward of ofq_lsq_fwd_patch: gy in patch order ([images * gh * gw][channels * ph * pw]: the input gradient of the convolution's
// GEMM as it leaves that GEMM), x / dx in image order.  Workspace: ofq_lsq_bwd_ws_bytes(images, channels, pixels, bias_len, 0).
extern "C" int ofq_lsq_bwd_patch(const float* gy, const float* x, const float* s, const float* b4, float* dx, float* ds, float* db4,
                                 float* dbaft, int64_t images, int64_t channels, int64_t pixels, int64_t bias_len, int lo, int hi,
                                 float gscale, int width, int ph, int pw, void* ws, size_t ws_bytes, void* amax_out,
                                 ofq_stream_t stream) {
  if (width <= 0) return OFQ_EINVAL;
  return lsq_bwd_impl(gy, x, s, b4, dx, ds, db4, dbaft, images, channels, pixels, pixels, pixels, bias_len, 0, lo, hi, gscale, 0, ws,
                      ws_bytes, amax_out, LsqPatch{width, ph, pw}, stream);
}
