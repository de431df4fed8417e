/* ofq_hip.h — C ABI of libofq_hip.so, the MI355X (gfx950) kernels behind the OFQ QAT hot path.
 *
 * The reference (nbasyl/OFQ) has no FFI layer: its boundary is the Python nn.Module API
 * (SURVEY.md §8b).  Each entry point below replaces a chain of eager ATen ops inside one reference
 * module method; the citation gives that method (file:line under /root/reference).  The Python
 * host (ofq_amd/) binds these with ctypes and wraps them in torch.autograd.Function objects; see
 * INTEGRATION.md for the stub a reference maintainer would add.
 *
 * Conventions
 *  - all pointers are DEVICE pointers to contiguous fp32 unless stated; sizes are element counts
 *  - `stream` is a hipStream_t passed as void*; every call only enqueues work on it
 *  - no allocation and no host synchronisation  => hipGraph-capturable.  Host state, all of it: (1) the queue of deferred
 *    second-stage reductions between ofq_sum_defer(1) and ofq_sum_flush() (a capture records the queue at its flush); (2) the
 *    OFQ_* environment switches that some launchers read ONCE at their first call -- A/B and test hooks that choose between
 *    kernels giving the same results (tools/README.md), never needed in production; (3) nothing else: no caches, no streams,
 *    no device memory owned by the library (workspaces, flags and tables are the caller's)
 *  - return value: 0 on success, a hipError_t (>0) from the launch, or a negative OFQ_E* code
 *  - workspaces are caller-provided; the *_ws_bytes() functions are pure host arithmetic
 *
 * Entries that are NOT on the default training step (parity-tested, kept for the configurations / comparisons named):
 *    ofq_qattn_dqkx_lsq_bwd     the fused qkx backward: bit-identical to the pair the step runs, 278 vs 257 us (round 5)
 *    ofq_qgemm_bf16s_nt_lsq     dX GEMM + LSQ backward in one kernel: slower than the pair (OFQ_LSQ_BWD_FUSE=1, three-plane mode)
 *    ofq_qattn_scores_i8, ofq_qattn_dp_bf16s, ofq_softmax_lsq_fwd / _bwd
 *                               the unfused forms of the two fused attention kernels: what those are tested against, and the path
 *                               of token counts the fused kernels do not take (N > 256)
 *    ofq_gemm_bf16x3x3_nt       the KD teacher's nine- / six-product GEMM (teacher gemm="bf16x9" / "bf16x6"; the default is f16x4)
 *    ofq_cga_mask_grad_save, ofq_cga_restore
 *                               the reference's three-kernel CGA sequence for optimisers other than ofq_amd.optim.FusedAdamW
 */
#ifndef OFQ_HIP_H
#define OFQ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OFQ_ABI_VERSION 2
#define OFQ_EINVAL (-1)  /* bad argument (shape, alignment, null pointer) */
#define OFQ_ENOWS  (-2)  /* workspace too small */
#define OFQ_AMAX_WORDS 64  /* slots per absolute-maximum group (ofq_absmax_f32, amax / amax_out arguments) ... */
#define OFQ_AMAX_STRIDE 32 /* ... one every 32 words (its own 128-byte line): a group is 64 * 32 * 4 = 8192 bytes */

typedef void* ofq_stream_t;

int ofq_abi_version(void);
/* "OFQ_SOURCE_HASH=<16 hex>": content hash of the kernel sources + this header the library was built from */
const char* ofq_source_hash(void);

/* ---- K1  StatsQ weight quantiser: StatsQuantizer.forward, src/quantization/quantizer/statsq.py:133-150
 *  s_r = 2*mean_c|W_rc| ; L = rne(clamp(W/s,-1,1-1e-6)*n - 0.5) ; Wq = s*(L+0.5)/n ; out = (Wq - W) + W
 *  scale_given != 0: `scale` is an INPUT (used by tests to check levels bit-exactly for a given s).
 *  levels (int8, optional): L in [-n, n-1], or with odd_codes != 0 the odd integer 2L+1 (W_hat = s*(2L+1)/(2n)),
 *  the exact integer operand of ofq_qgemm_i8_nt.  Backward is the identity (STE, statsq.py:148): no kernel. */
int ofq_statsq_fwd(const float* W, int64_t rows, int64_t cols, int bits, float* out, float* scale,
                   int8_t* levels, int scale_given, int odd_codes, ofq_stream_t stream);
/*  The code path of the quantised linear layers in one launch: scale, the odd int8 codes 2L+1 [rows][cols], optionally
 *  the fake-quant values (out may be NULL), the codes transposed as bf16 [cols][rows] (operand of the dX GEMM) and
 *  rout[row] = sum_k rvec[k] * code[row][k] (the offset term of ofq_qgemm_i8_nt: rvec = the layer's move_aft).
 *  bits | 0x100 (here and in ofq_statsq_tensor's bits): the transposed codes are written as FP16 instead of bf16 -- the operand
 *  of the two-plane fp16 form of the dX GEMM (ofq_qgemm_bf16s_nt with amax). */
int ofq_statsq_codes_fwd(const float* W, int64_t rows, int64_t cols, int bits, float* out, float* scale, int8_t* codes,
                         void* codesT_bf16, const float* rvec, float* rout, ofq_stream_t stream);
/*  The same code path for MANY weight tensors in one or two launches (weights only change at the optimizer step, so a
 *  training step can refresh every layer's operands up front instead of one small launch inside each forward):
 *  host_entries = n records of ofq_statsq_tensor_entry_bytes() bytes, each { const float* W; float* scale; int8_t* codes;
 *  uint16_t* codesT_bf16 ( may be NULL ); const float* rvec ( may be NULL ); float* rout; int64 rows, cols, bits; }.
 *  Row for row identical to ofq_statsq_codes_fwd (no fake-quant values are written). */
int64_t ofq_statsq_tensor_entry_bytes(void);
int ofq_statsq_codes_multi(const void* host_entries, int64_t n, ofq_stream_t stream);

/* ---- K3/K5  LSQ activation quantiser with its LearnableBias sandwich:
 *  LsqQuantizer.forward lsq.py:571-602 (+ :72-101, :336-373, :419-437, :489-505), LsqQuantizer4v.forward
 *  lsq.py:757-792, LearnableBias.forward qbias.py:9-13, QLinear.forward qlinear.py:66-68,
 *  QMLP.forward qlinear.py:127 (exact-erf GELU prologue).
 *  x is viewed as [outer][S][inner] (row r = o*S + j, inner contiguous).
 *    scale_mode 0: scale index = r % S  (per "token": the reference's x.shape[-2] axis), s has S entries
 *    scale_mode 1: scale index = column (per channel, LsqQuantizer4v), s has `inner` entries, S must be 1
 *  b4 / baft (optional, may be NULL) have bias_len = k*inner entries, bias index = (r % k)*inner + c.
 *  y = ((rne(u) - u) + u) * a_eff + baft,  u = clamp((pre(x) + b4) / a_eff, lo, hi),
 *  a_eff = (a - a*g) + a*g with a = max(s, 1e-5), g = gscale  (fp32, lsq.py:6-18, :593).
 *  prologue: 0 none, 1 exact GELU.   y may be NULL when only the codes are wanted.   codes (optional, contiguous, 1 byte/elt) receives rne(u): int8 for signed
 *  ranges, uint8 for unsigned ones.
 *  ldx / ldy: row strides of x (and dx) / of y (and g), so column slices of a wider matrix (the q,k,v
 *  thirds of the qkv projection, attention.py:72-75) are quantised in place.  inner, ldx, ldy % 4 == 0. */
int ofq_lsq_fwd(const float* x, const float* s, const float* b4, const float* baft, float* y, int8_t* codes,
                int64_t outer, int64_t S, int64_t inner, int64_t ldx, int64_t ldy, int64_t bias_len, int scale_mode,
                int lo, int hi, float gscale, int prologue, ofq_stream_t stream);

/* ---- K4  LSQ backward (closed form of the autograd graph of the ops above):
 *  dx_q = (g*a_eff)/a_eff * 1[lo<=v<=hi];  dx = dx_q * pre'(x);  db4 = sum dx_q;  dbaft = sum g;
 *  ds_j = gscale * sum g * (rne(v)-v if in range else clamp(v)).
 *  ds/db4/dbaft are OVERWRITTEN (any may be NULL).  ws: ofq_lsq_bwd_ws_bytes() bytes of scratch. */
size_t ofq_lsq_bwd_ws_bytes(int64_t outer, int64_t S, int64_t inner, int64_t bias_len, int scale_mode);
int ofq_lsq_bwd(const float* g, const float* x, const float* s, const float* b4, float* dx, float* ds,
                float* db4, float* dbaft, int64_t outer, int64_t S, int64_t inner, int64_t ldx, int64_t ldy,
                int64_t bias_len, int scale_mode, int lo, int hi, float gscale, int prologue, void* ws,
                size_t ws_bytes, void* amax_out, ofq_stream_t stream);
/*  Round 6: the image quantiser of the W8A8 patch embedding (qlinear.py:166-174: per-channel step, per-pixel offsets, then a
 *  stride == kernel convolution) with its output in PATCH (im2col) order -- y / codes are [images * gh * gw][channels * ph * pw], the
 *  operand layout of the convolution's GEMM -- and its backward reading gy in that order: the reference's unfold / permute copies
 *  (three per step) are addressing.  x / dx: [images][channels][pixels = H * width]; width, pw multiples of 4. */
int ofq_lsq_fwd_patch(const float* x, const float* s, const float* b4, const float* baft, float* y, int8_t* codes, int64_t images,
                      int64_t channels, int64_t pixels, int64_t bias_len, int lo, int hi, float gscale, int width, int ph, int pw,
                      ofq_stream_t stream);
int ofq_lsq_bwd_patch(const float* gy, const float* x, const float* s, const float* b4, float* dx, float* ds, float* db4,
                      float* dbaft, int64_t images, int64_t channels, int64_t pixels, int64_t bias_len, int lo, int hi, float gscale,
                      int width, int ph, int pw, void* ws, size_t ws_bytes, void* amax_out, ofq_stream_t stream);

/* ---- K10  scale + softmax + unsigned LSQ on attention scores:
 *  QAttention*.forward attention.py:96-99 / :213-216.  scores [rows][ld] (ld >= n, row r belongs to
 *  query token r % S); prob (saved for backward) and y are written with the same ld, pad columns = 0.
 *  y = LSQ_unsigned(softmax(scores * alpha)); s has S entries. */
int ofq_softmax_lsq_fwd(const float* scores, const float* s, float* prob, float* y, int64_t rows, int64_t n,
                        int64_t ld, int64_t S, float alpha, int hi, float gscale, uint8_t* codes, float* code_rowsum,
                        const float* addend, int64_t add_period, ofq_stream_t stream);
                        /* codes (optional, [rows][ld] uint8) and their row sums feed ofq_qattn_pv_i8.
                         * addend (optional, [add_period][S][ld]): added to scores*alpha before the softmax, slab index
                         * (r / S) % add_period — Swin's relative-position bias + shift mask, one slab per (window, head)
                         * (swin_attention_and_mlp.py:201-221). */
size_t ofq_softmax_lsq_bwd_ws_bytes(int64_t rows);
/*  backward: g = dL/dy -> dscores (may alias g), ds[S] overwritten. */
int ofq_softmax_lsq_bwd(const float* g, const float* prob, const float* s, float* dscores, float* ds,
                        int64_t rows, int64_t n, int64_t ld, int64_t S, float alpha, int hi, float gscale,
                        float* ds_rowsum, void* ws, size_t ws_bytes, void* amax_out, ofq_stream_t stream);
                        /* ds_rowsum (optional, [rows]): sum_m dscores[r][m], used by ofq_qattn_dxq's offset term */

/* ---- K6-K9, K11  fp32 MFMA GEMM (v_mfma_f32_32x32x2_f32, exact fp32 fmaf chain):
 *  F.linear qlinear.py:69, torch.einsum attention.py:200/:210, `@` attention.py:96/:102/:193/:219 and
 *  their autograd backward.   C[b0,b1] = alpha * sum_kb opA(A[b0,b1,kb]) * opB(B[b0,b1,kb]) + bias[n]
 *    transA 0: A is [M][K] (lda),  1: A is [K][M];   transB 0: B is [K][N] (ldb),  1: B is [N][K]
 *  split_k > 1 needs ws of ofq_gemm_ws_bytes(); the partials are reduced deterministically. */
typedef struct ofq_gemm_desc {
  const float* A;
  const float* B;
  float* C;
  const float* bias; /* optional, N entries, added to every row */
  int64_t M, N, K;
  int64_t lda, ldb, ldc;
  int32_t transA, transB;
  int32_t nb0, nb1;  /* batch grid (>=1) */
  int64_t sA0, sA1, sB0, sB1, sC0, sC1;
  int32_t nkb;       /* extra contraction batches accumulated into one C (>=1) */
  int64_t sAk, sBk;
  int32_t split_k;   /* >=1 */
  float alpha;
  int32_t accumulate; /* C += result (beta = 1) when non-zero */
  int32_t tile_hint;  /* 0 = choose; 64 / 128 force the 64x64 / 128x128 workgroup tile (benchmarking) */
} ofq_gemm_desc;
size_t ofq_gemm_ws_bytes(const ofq_gemm_desc* d);
int ofq_gemm_f32(const ofq_gemm_desc* d, void* ws, size_t ws_bytes, ofq_stream_t stream);

/* ---- exact integer-code GEMMs for the fake-quantised linear layers (same maths as F.linear on the fake-quant
 *  values, qlinear.py:69, with the scales factored out of the contraction; see ofq_amd/csrc/qgemm_args.h)
 *  forward:  y[m,n] = col_mult*col_scale[n] * (a_eff[m % S] * sum_k A[m,k]*B[n,k] + r[n]) + bias[n]
 *            A = LSQ codes of the input [M][K] int8, B = weight codes [N][K] int8, a_eff from lsq_s/gscale as in
 *            ofq_lsq_fwd, r[n] = sum_k baft[k]*B[n,k] (ofq_rowdot_i8).  K, lda, ldb % 16 == 0.  i8 MFMA, exact. */
int ofq_qgemm_i8_nt(const int8_t* A, const int8_t* B, float* C, const float* bias, const float* col_scale,
                    float col_mult, const float* r, const float* lsq_s, int64_t S, float gscale, int64_t M, int64_t N,
                    int64_t K, int64_t lda, int64_t ldb, int64_t ldc, ofq_stream_t stream);
/*  Same GEMM with a by-product: the int8 codes of the NEXT layer's input quantiser applied to this output,
 *  qcodes[m][n] = LSQ([gelu](y[m][n]) + q_b4[n]; step q_s[m % q_S], range [q_lo, q_hi]) -- bit for bit what ofq_lsq_fwd
 *  computes from the stored y (qlinear.py:123-136: fc1 -> GELU -> fc2's offset + LSQ; attention.py:201-206: qkx, one
 *  step per (token, head); :184 v, one step per channel), so the consumer does not read y again.  N % 16 == 0.
 *  q_colmode 0: step index (m * q_rowmul + n / q_coldiv) % q_S  (q_rowmul = quantiser rows per output row, q_coldiv =
 *  their width, a multiple of 128 when q_rowmul > 1);  q_colmode 1: step index n (q_S == N). */
int ofq_qgemm_i8_nt_q(const int8_t* A, const int8_t* B, float* C, const float* bias, const float* col_scale,
                      float col_mult, const float* r, const float* lsq_s, int64_t S, float gscale, int64_t M, int64_t N,
                      int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int8_t* qcodes, int64_t ldq, const float* q_s,
                      int64_t q_S, float q_gscale, const float* q_b4, int q_lo, int q_hi, int q_gelu, int q_rowmul,
                      int64_t q_coldiv, int q_colmode, ofq_stream_t stream);

/* ---- backward of "linear layer -> input quantiser of its only consumer" WITHOUT the saved activation (qlinear.py:58-73
 *  followed by lsq.py:571-602 of the next module): ofq_qgemm_i8_nt_q accepts C = NULL and then writes only the consumer's
 *  codes; this entry recomputes the fp32 layer output y[M][N] from the int8 operands (same k-loop, same epilogue
 *  expression: bit-identical to the value the forward would have stored) and applies ofq_lsq_bwd's arithmetic to it in
 *  registers.  gy: gradient w.r.t. the consumer quantiser's output [M][ldg]; dy: gradient w.r.t. y [M][ldd] (may alias gy);
 *  ds / db4 / dbaft: gradients of the consumer's step / pre-offset / post-offset (any may be NULL).  Quantiser geometry as
 *  in ofq_qgemm_i8_nt_q (q_rowmul, q_coldiv, q_colmode); row mode needs M % (q_S / q_rowmul) == 0. */
size_t ofq_qgemm_i8_lsq_bwd_ws_bytes(int64_t M, int64_t N, int q_colmode);
int ofq_qgemm_i8_lsq_bwd(const int8_t* A, const int8_t* B, const float* bias, const float* col_scale, float col_mult,
                         const float* r, const float* lsq_s, int64_t S, float gscale, int64_t M, int64_t N, int64_t K,
                         int64_t lda, int64_t ldb, const float* gy, int64_t ldg, float* dy, int64_t ldd,
                         const float* q_s, int64_t q_S, float q_gscale, const float* q_b4, int q_lo, int q_hi,
                         int q_gelu, int q_rowmul, int64_t q_coldiv, int q_colmode, float* ds, float* db4, float* dbaft,
                         void* ws, size_t ws_bytes, void* amax_out, ofq_stream_t stream);
/*  QKR attention (attention.py:200-210 + lsq.py:571-602): the backward of [scores <- qkx quantiser <- qkx = x_hat . W_qk^T]
 *  from dS in ONE launch.  The incoming gradient of the quantiser, dqkx[(b, m), (h, c)] = sum_n dS[b, h, n, m] *
 *  (a_eff[n] * xcodes[b, n, c] + bax[c]), is formed inside the recompute kernel (two fp16 planes of dS on the scale of
 *  `amax`, as ofq_qattn_dqkx_bf16s forms it) instead of being written by one kernel and read by the next: the result
 *  equals ofq_qattn_dqkx_bf16s(amax) -> ofq_qgemm_i8_lsq_bwd bit for bit, 8 B per qkx element less HBM traffic.
 *  xcodes [B Ntok][lda] int8 (C used columns), wcodes [H C][ldb] int8; sx [Ntok] steps of x_hat (per token); dS
 *  [B][H][Ntok][ldS]; q_s [q_S = Ntok H] steps of the qkx quantiser, (token, head); dy [B Ntok][ldd] (H C columns).
 *  Needs Ntok >= 128 and even, C % 128 == 0, ldS even; workspace as ofq_qgemm_i8_lsq_bwd_ws_bytes(B Ntok, H C, 0).
 *  MEASURED (round 5, tools/dqkx_fused_bench.py, one DeiT-S block at 128 images): 278 us against 257 us for the pair --
 *  1.68 GB -> 0.75 GB of traffic, but the dS panel is split once per 128-column tile (three times per head) and a
 *  one-tile workgroup does not hide the panel's load latency behind the quantiser arithmetic; the training step
 *  therefore keeps the pair (functional.ScoresSoftmaxCodesFn.backward), this entry is parity-tested and not on its path. */
int ofq_qattn_dqkx_lsq_bwd(const int8_t* xcodes, const int8_t* wcodes, const float* bias, const float* col_scale,
                           float col_mult, const float* r, const float* sx, float gscale_x, const float* bax,
                           const float* dS, int64_t ldS, const void* amax, int64_t B, int64_t H, int64_t Ntok, int64_t C,
                           int64_t lda, int64_t ldb, float* dy, int64_t ldd, const float* q_s, int64_t q_S,
                           float q_gscale, const float* q_b4, int q_lo, int q_hi, float* ds, float* db4, float* dbaft,
                           void* ws, size_t ws_bytes, void* amax_out, ofq_stream_t stream);

/*  backward: C[m,n] (+)= alpha * sum_k (A[m,k]*k_scale[k]) * B[n,k]   A fp32 [M][K] (e.g. dY), B bf16 codes [N][K]
 *            (the transposed weight codes), A*k_scale split into nsplit (2|3) bf16 pieces; 3 = exact fp32 product.
 *            K % 8 == 0, lda % 4 == 0, ldb % 8 == 0.
 *            amax != NULL (with nsplit = 2): the TWO-PLANE FP16 form (round 5).  *amax = the bits of an upper bound of max |A|
 *            (a device word: what ofq_absmax_f32 or the producing backward kernel's amax_out wrote), B = the codes as FP16.
 *            The launch scales A*k_scale by the power of two that puts its largest magnitude into [2^14, 2^15), splits it into
 *            hi = rne_f16(x), lo = rne_f16(x - hi) -- |x - hi - lo| <= 2^-24 |x| for elements within 2^-17 of the maximum,
 *            <= 2^-39 of the maximum below -- and un-scales in the epilogue: fp32-grade on the scale of the tensor with two
 *            matrix-core products per element instead of three (reference op: autograd of F.linear, qlinear.py:69).
 *            col_scale / col_bias (optional, N > 128): C[m,n] = col_scale[n] * alpha * (...) + col_bias[n] -- the FORWARD of a
 *            layer whose weight is step[n] * code[n,k] on fp32 activations (the W8A8 patch embedding, qlinear.py:166-177). */
int ofq_qgemm_bf16s_nt(const float* A, const void* B_bf16, float* C, const float* k_scale, float alpha, int accumulate,
                       int nsplit, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, const void* amax,
                       const float* col_scale, const float* col_bias, ofq_stream_t stream);
/*  stream-K form of the same product, for shapes whose 128 x 384 tiles do not fill the chip (M = 25 344 tokens: 198 row
 *            tiles on 256 CUs): `num_wgs` workgroups (pass the CU count) share the tiles x k-steps of the launch evenly, each walking its share as one continuous k-step stream;
 *            a tile cut by a share boundary is finished by the workgroup holding its k = 0 piece, which adds the other
 *            holders' fp32 partials (published through `ws`) in workgroup order -- bit-identical from launch to launch,
 *            a different association of the same sums than ofq_qgemm_bf16s_nt.  Up to two K-segments:
 *              C[m,n] (+)= sum_seg alpha_seg * sum_k (A_seg[m,k] * k_scale_seg[k]) * B_seg[n,k]
 *            -- the input gradients that two layers send to one tensor (attention.py:180 and :200: v and W_qk both read
 *            x_hat) as ONE GEMM over the concatenated contraction; with two segments alpha is applied to the k-scale
 *            (callers pass powers of two: same bits as applying it last).  Every K % 32 == 0 AND the sum of K over the segments % 64 == 0
 *            (the workgroups share PAIRS of k-steps; EINVAL otherwise), lda >= K, ldb >= K, ldc >= N, N > 128, M * lda * 4 < 2^32.
 *            The launch uses the largest divisor of the tile count between 3/4 num_wgs and num_wgs as its workgroup count
 *            when there is one (whole tiles only: then every bit equals ofq_qgemm_bf16s_nt's), num_wgs workgroups with cut
 *            tiles otherwise; num_wgs < 0 forces exactly -num_wgs workgroups.
 *            `ws`: ofq_qgemm_bf16s_nt_sk_ws_bytes(|num_wgs|) bytes, ZEROED once by the caller before its first use and then
 *            owned by this entry point (calls sharing it must be ordered on one stream).  ofq_qgemm_bf16s_nt_sk_pays: 1 when
 *            the stream-K launch is expected to beat the one-tile-per-workgroup launch for the shape.
 *            Hand-off errors: every wait of an owner for a partial is bounded (~0.6 s); when one runs out the owner raises the
 *            STICKY error word of the workspace (int 4096 of `ws`), leaves the late publisher's flag alone and stores a wrong
 *            tile -- this and every later launch on the workspace are suspect until ofq_qgemm_bf16s_nt_sk_reset has re-zeroed the
 *            flag area.  ofq_qgemm_bf16s_nt_sk_check makes the word visible INSIDE a step without a host sync: loss[0] = NaN
 *            when it is set (one thread; capturable).  (int 4097 of `ws`: fault injection for tests -- the workgroup with that
 *            index + 1 never publishes; zero in real runs.) */
typedef struct ofq_nt_seg {
  const float* A; const void* B_bf16; const float* k_scale;
  int64_t K, lda, ldb;
  float alpha;
  const void* amax;     /* NULL: three bf16 planes, B_bf16 = bf16 codes.  Else (every segment of the launch): the device word
                           holding the bits of an upper bound of max |A| -- the two-plane fp16 form, B_bf16 = FP16 codes */
  int32_t hi_only;      /* segment 1 of a two-plane launch only: A's LEADING plane alone multiplies this segment's B (the
                           trailing plane's product is skipped) -- with B = the trailing plane of a split weight this makes
                           the forward below a three-product one (x_hi W_hi + x_lo W_hi + x_hi W_lo); K of segment 0 % 64 == 0 */
} ofq_nt_seg;
size_t ofq_qgemm_bf16s_nt_sk_ws_bytes(int num_wgs);
int ofq_qgemm_bf16s_nt_sk_pays(int64_t M, int64_t N, int64_t K, int num_wgs);
int ofq_qgemm_bf16s_nt_sk(const ofq_nt_seg* segs, int nseg, float* C, int accumulate, int64_t M, int64_t N, int64_t ldc,
                          int num_wgs, void* ws, size_t ws_bytes, const float* col_bias, ofq_stream_t stream);
/*            col_bias (optional): + col_bias[n] on every finished element.  With it the two-segment form is also the FORWARD of an
 *            fp32 linear layer on the fp16 matrix cores (the frozen KD teacher, train.py:428-442, :906-910): the weight split ONCE
 *            into W 2^Ew = Wh + Wl (two fp16 planes), the launch computes [x | x] . [Wh | Wl]^T with x split in the kernel --
 *            four plane products instead of the nine of ofq_gemm_bf16x3x3_nt (three with hi_only on the second segment: the
 *            dropped x_lo W_lo term is 2^-22 of a product), alpha_seg = 2^-Ew (ofq_amd/teacher.py "f16x4"). */
int ofq_qgemm_bf16s_nt_sk_check(const void* ws, float* loss, ofq_stream_t stream);
int ofq_qgemm_bf16s_nt_sk_reset(void* ws, ofq_stream_t stream);
/*  dX GEMM fused with the backward of the layer's own input quantiser (qlinear.py:66-69: x -> move_b4 -> LSQ -> move_aft
 *  -> F.linear): dX_hat = alpha * (dY * k_scale) @ B never leaves the kernel; its epilogue applies ofq_lsq_bwd's
 *  arithmetic (per-token step lsq_s[m % S], offset b4[n], optional GELU prologue) and writes dx[M][N] (ld ldx), and
 *  through a fixed-order second stage ds[S], db4[N], dbaft[N] (each optional).  x is the quantiser's fp32 input
 *  [M][N] (ld ldx).  N > 128, M % S == 0. */
size_t ofq_qgemm_bf16s_nt_lsq_ws_bytes(int64_t M, int64_t N);
int ofq_qgemm_bf16s_nt_lsq(const float* dY, const void* B_bf16, const float* k_scale, float alpha, const float* x,
                           const float* lsq_s, int64_t S, float gscale, const float* b4, int lo, int hi, int gelu,
                           float* dx, float* ds, float* db4, float* dbaft, int64_t M, int64_t N, int64_t K, int64_t lda,
                           int64_t ldb, int64_t ldx, void* ws, size_t ws_bytes, const void* amax, ofq_stream_t stream);
/*  (amax, round 6: NULL = B holds bf16 codes and dY is split into three bf16 planes; else the maximum word of dY, B holds fp16
 *  codes and dY is split into two fp16 planes, as in ofq_qgemm_bf16s_nt.) */
/*  weight gradient: dW[o,c] = sum_m (dY[m,o] * a_eff[m % S]) * codes[m,c] + db[o]*baft[c]  (= dY^T @ X_hat with
 *            X_hat = a_eff*codes + baft).  dY fp32 [Ktok][M], codes int8 [Ktok][N]; three bf16 pieces of dY*a_eff,
 *            LDS transpose reads, split-K over tokens with a deterministic reduction.  M % 4 == 0, N % 16 == 0.
 *            compute_db != 0: db[o] = sum_m dY[m,o] (the bias gradient) is produced by the same pass over dY and
 *            written to `db`; otherwise `db` (may be NULL) is an input. */
size_t ofq_qgemm_bf16s_tn_ws_bytes(int64_t M, int64_t N, int split);
int ofq_qgemm_bf16s_tn(const float* dY, const int8_t* codes, float* dW, const float* lsq_s, int64_t S, float gscale,
                       float* db, int compute_db, const float* baft, int64_t Ktok, int64_t M, int64_t N, int64_t lda,
                       int64_t ldb, int split, void* ws, size_t ws_bytes, const void* amax, ofq_stream_t stream);
/*  several weight gradients in one GEMM launch + one reduce launch (each job = one ofq_qgemm_bf16s_tn call, same
 *            results bit for bit).  The weight gradients of F.linear (qlinear.py:69) have no consumer before the
 *            optimiser step / gradient all-reduce (train.py:927-933), so a caller may collect the ones of a whole
 *            transformer block (Block.forward deit_vision_transformer.py:154-164: v, W_qk, proj, fc1, fc2) and run them
 *            here: each workgroup then owns a long token range, and the split-K partials shrink fivefold.
 *            Every job: N > 128, N % 16 == 0, any S >= 1 (round 6: N < 256 and step vectors shorter than a 32-token k-tile too); all jobs of one tile class (N % 384 == 0 for all or none);
 *            njobs <= 8; `split` (>= 1) is common, about 256 / (sum of ceil(M/128) * ceil(N/384) over the jobs). */
typedef struct ofq_tn_job {
  const float* dY; const int8_t* codes; float* dW; const float* lsq_s; float* db; const float* baft;
  int64_t S, Ktok, M, N, lda, ldb;
  float gscale; int32_t compute_db;
  const void* amax;     /* NULL: three bf16 planes of dY.  Else (every job of the launch): the device word with the bits of an upper
                           bound of max |dY| -- two fp16 planes of dY * step * 2^E (see ofq_qgemm_bf16s_nt) */
} ofq_tn_job;
size_t ofq_qgemm_bf16s_tn_group_ws_bytes(const ofq_tn_job* jobs, int njobs, int split);
int ofq_qgemm_bf16s_tn_group(const ofq_tn_job* jobs, int njobs, int split, void* ws, size_t ws_bytes, ofq_stream_t stream);
/*  int8 codes [rows][cols] -> bf16 [cols][rows];   out[r] = sum_k vec[k]*codes[r][k] */
int ofq_codes_transpose_bf16(const int8_t* codes, void* out_bf16, int64_t rows, int64_t cols, ofq_stream_t stream);
/*  the same with the codes as FP16 (B operand of the two-plane form: ofq_qgemm_bf16s_nt / _nt_sk with amax) */
int ofq_codes_transpose_f16(const int8_t* codes, void* out_f16, int64_t rows, int64_t cols, ofq_stream_t stream);
/*  amax: a GROUP of OFQ_AMAX_WORDS (64) slots OFQ_AMAX_STRIDE words apart (8 KB), zeroed by the caller; afterwards max over the slots = the bits of
 *            max |x[r][c]| over x fp32 [rows][cols] (ld) (atomic maxima on the bit patterns, spread over the group so that the
 *            waves of a launch do not queue up behind one address; order-independent: deterministic; a NaN in x gives the NaN
 *            pattern).  Every `amax` / `amax_out` argument of this header is such a group.  Serves the two-plane fp16 backward GEMMs
 *            (autograd of F.linear, qlinear.py:69) for gradient tensors whose producer did not write the word itself.
 *            cols % 4 == 0, ld % 4 == 0, x 16-byte aligned. */
int ofq_absmax_f32(const float* x, int64_t rows, int64_t cols, int64_t ld, void* amax, ofq_stream_t stream);
/*  out[i * repeat + r] = the effective LSQ step (a - a g) + a g, a = max(s[i], 1e-5) (lsq.py:6-18, grad_scale's value form) for
 *            i < n, r < repeat: the step vectors the W8A8 patch embedding (qlinear.py:166-177) hands to the code GEMMs */
int ofq_lsq_eff_scale_vec(const float* s, float gscale, float* out, int64_t n, int64_t repeat, ofq_stream_t stream);
int ofq_rowdot_i8(const int8_t* codes, const float* vec, float* out, int64_t rows, int64_t cols, ofq_stream_t stream);

/* ---- attention products on the integer codes (QAttention_qkreparam.forward attention.py:200-219 and autograd).
 *  Layouts: xcodes [B][N][C], qcodes (qkx) [B][N][H][C], vcodes [B][N][C], vcodesT [B][C][Np] (zero padded),
 *  pcodes [B][H][N][Np] (uint8 softmax codes, zero padded), S/dS/dP fp32 [B][H][N][ldS];  C = H*d, Np % 16 == 0.
 *  sx/sq/sv/sp are the LSQ step vectors of x (N), qkx (N*H), v (C), softmax (N) with their gradient scales. */
int ofq_qattn_scores_i8(const int8_t* xcodes, const int8_t* qcodes, float* S, const float* sx, float gscale_x,
                        const float* sq, float gscale_q, const float* u, const float* tq, const float* z, int64_t B,
                        int64_t H, int64_t N, int64_t C, int64_t ldS, ofq_stream_t stream);

/*  Plain (non-reparameterised) attention on the codes, attention.py:67-105: q_hat = aq[n]*qq + bq[c], k_hat = ak[m]*qk + bk[c]
 *  (per-token steps sq / sk [N], per-channel offsets), S[b][h][n][m] = q_hat[b,n,hd:hd+d] . k_hat[b,m,hd:hd+d] =
 *  aq[n]*(ak[m]*I + u[b,n,h]) + ak[m]*tq[b,m,h] + z[h]; qcodes / kcodes [B][N][H*d], u / tq [B][N][H], z [H].
 *  Backward of the scores: dq[b,n,hd+c] = sum_m (dS*ak[m])*qk, dk[b,m,hd+c] = sum_n dS*(aq[n]*qq + bq[hd+c]). */
int ofq_qattn_scores_plain_i8(const int8_t* qcodes, const int8_t* kcodes, float* S, const float* sq, float gscale_q,
                              const float* sk, float gscale_k, const float* u, const float* tq, const float* z,
                              int64_t B, int64_t H, int64_t N, int64_t d, int64_t ldS, ofq_stream_t stream);

/*  Scores GEMM + softmax + unsigned LSQ of the probabilities in one kernel (attention.py:96-99 / :207-216): the score
 *  matrix never goes to HBM.  acodes / bcodes and the epilogue terms as in ofq_qattn_scores_i8 (plain = 0: x codes [B][N][CK],
 *  qkx codes [B][N][H][CK], sb one step per (token, head)) or ofq_qattn_scores_plain_i8 (plain = 1: q / k codes [B][N][H*CK],
 *  CK = head dim, sb per token); softmax arguments as in ofq_softmax_lsq_fwd.  Outputs: prob fp32 [B][H][N][ld] (kept for
 *  the backward), codes uint8 [B][H][N][ld], rowsum [B][H][N] (row sums of the codes).  N <= 256, ld <= 256, ld % 4 == 0. */
int ofq_qattn_scores_softmax_i8(const int8_t* acodes, const int8_t* bcodes, const float* sa, float gscale_a, const float* sb,
                                float gscale_b, const float* u, const float* tq, const float* z, int plain, const float* sm_s,
                                float sm_gscale, float alpha, int hi, const float* addend, int64_t add_period, float* prob,
                                uint8_t* codes, float* rowsum, int64_t B, int64_t H, int64_t N, int64_t CK, int64_t ld,
                                ofq_stream_t stream);

int ofq_qattn_dq_plain_bf16s(const float* dS, const int8_t* kcodes, float* dq, const float* sk, float gscale_k,
                             int64_t B, int64_t H, int64_t N, int64_t d, int64_t ldS, ofq_stream_t stream);
int ofq_qattn_dk_plain_bf16s(const float* dS, const int8_t* qcodes, float* dk, const float* sq, float gscale_q,
                             const float* bq, int64_t B, int64_t H, int64_t N, int64_t d, int64_t ldS, ofq_stream_t stream);

int ofq_qattn_pv_i8(const int8_t* pcodes, const int8_t* vcodesT, float* O, const float* sp, float gscale_p,
                    const float* sv, float gscale_v, const float* bav, const float* rp, int64_t B, int64_t H, int64_t N,
                    int64_t d, int64_t Np, ofq_stream_t stream);
int ofq_qattn_dp_bf16s(const float* dO, const int8_t* vcodes, float* dP, const float* sv, float gscale_v, const float* w,
                       int64_t B, int64_t H, int64_t N, int64_t d, int64_t ldP, ofq_stream_t stream);
/*  dP GEMM + backward of P_hat = LSQ_unsigned(softmax(alpha * S)) in one kernel (autograd of attention.py:213-219):
 *  dS[b,h,n,:] from dO[b,n,h*d:(h+1)*d], the v codes / step sv (+ offset bav: the row constant dO . bav), the saved
 *  probabilities `prob` and the softmax quantiser's steps sm_s; dP never reaches memory.  ds[N] (optional) = the step
 *  gradient, ds_rowsum (optional) = row sums of dS.  N, ld <= 256, ld % 4 == 0, d % 16 == 0. */
size_t ofq_qattn_dp_softmax_bwd_ws_bytes(int64_t B, int64_t H, int64_t N);
int ofq_qattn_dp_softmax_bwd(const float* dO, const int8_t* vcodes, const float* sv, float gscale_v, const float* bav,
                             const float* prob, const float* sm_s, float sm_gscale, float alpha, int hi, float* dS,
                             float* ds, float* ds_rowsum, int64_t B, int64_t H, int64_t N, int64_t d, int64_t ld,
                             void* ws, size_t ws_bytes, void* amax_out, ofq_stream_t stream);
int ofq_qattn_dv_bf16s(const float* dO, const int8_t* pcodes, float* dV, const float* sp, float gscale_p, int64_t B,
                       int64_t H, int64_t N, int64_t d, int64_t Np, ofq_stream_t stream);
/*  (dqkx, dxq) amax: NULL = three bf16 planes of dS; else the device word with the bits of an upper bound of max |dS| over the
 *  N real columns of its rows (the pad columns up to ldS are never read and may hold anything): the two-plane fp16 form of the
 *  wide kernels, see ofq_qgemm_bf16s_nt */
int ofq_qattn_dqkx_bf16s(const float* dS, const int8_t* xcodes, float* dqkx, const float* sx, float gscale_x,
                         const float* bax, int64_t B, int64_t H, int64_t N, int64_t C, int64_t ldS, const void* amax,
                         ofq_stream_t stream);
int ofq_qattn_dxq_bf16s(const float* dS, const int8_t* qcodes, float* dxq, const float* sq, float gscale_q, int accumulate,
                        int64_t B, int64_t H, int64_t N, int64_t C, int64_t ldS, const void* amax, ofq_stream_t stream);
/*  helpers: out[r][v] = sum_k vecs[v][k]*codes[r][k];  out[r][h] = sum_{c<d} x[r][h*d+c]*vec[h*d+c];
 *  batched int8 transpose with zero padding in [B][R][C] -> out [B][C][Rp] */
int ofq_rowdot_i8_multi(const int8_t* codes, const float* vecs, float* out, int64_t rows, int64_t cols, int nvec,
                        ofq_stream_t stream);
int ofq_rowdot_f32_seg(const float* x, const float* vec, float* out, int64_t rows, int heads, int head_dim, int64_t ld,
                       ofq_stream_t stream);
int ofq_codes_transpose_i8(const int8_t* in, int8_t* out, int64_t batches, int64_t rows, int64_t cols, int64_t rows_padded,
                           ofq_stream_t stream);

/*  The three operand-preparation jobs of the QKR attention core (attention.py:207-219 on the codes) in one launch:
 *  u[b][n][h] = xcodes[b][n][:] . baq[h][:], tq[b][m][h] = qcodes[b][m][h][:] . bax[:], vT[b][c][Np] = transpose of
 *  vcodes[b][N][c] (zero-padded to Np).  Same values as ofq_rowdot_i8_multi / ofq_rowdot_i8 / ofq_codes_transpose_i8.
 *  z (optional, [H]): z[h] = baq[h][:] . bax[:], the offset-offset term of the scores, by one extra workgroup.
 *  C % 16 == 0, C <= 512, Np % 4 == 0, 16-byte aligned operands. */
int ofq_qattn_prep(const int8_t* xcodes, const float* baq, float* u, const int8_t* qcodes, const float* bax, float* tq,
                   const int8_t* vcodes, int8_t* vT, float* z, int64_t B, int64_t H, int64_t N, int64_t C, int64_t Np,
                   ofq_stream_t stream);


/* ---- column sum (bias gradients of F.linear: autograd of qlinear.py:71):  out[c] = sum_r x[r][c] */
size_t ofq_colsum_ws_bytes(int64_t rows, int64_t cols);
int ofq_colsum(const float* x, float* out, int64_t rows, int64_t cols, int64_t ld, void* ws, size_t ws_bytes,
               ofq_stream_t stream);

/* ---- LayerNorm over the channel dimension (deit_vision_transformer.py:91-102 Block.norm1/norm2 = nn.LayerNorm(dim,
 *  eps 1e-6), :138 the final norm; swin: torchvision LayerNorm eps 1e-5), optionally fused with the residual add that
 *  feeds it:  xsum = x + res (written when res != NULL),  y = (xsum - mean) * rstd * gamma + beta.  mean / rstd [rows]
 *  are kept for the backward.  Backward: dx = rstd*(g - mean(g) - xh*mean(g*xh)) [+ dres], g = dy*gamma,
 *  dgamma = sum_r dy*xh, dbeta = sum_r dy (fixed-order two-stage sums).  cols % 4 == 0, cols <= 2048. */
int ofq_layernorm_fwd(const float* x, const float* res, const float* gamma, const float* beta, float* y, float* xsum,
                      float* mean, float* rstd, int64_t rows, int64_t cols, int64_t ldx, int64_t ldy, float eps,
                      void* amax_out, ofq_stream_t stream);      /* amax_out: optional word group, max |y| */
size_t ofq_layernorm_bwd_ws_bytes(int64_t rows, int64_t cols);
int ofq_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                      const float* dres, float* dx, float* dgamma, float* dbeta, int64_t rows, int64_t cols, int64_t ldx,
                      int64_t ldy, void* ws, size_t ws_bytes, void* amax_out, ofq_stream_t stream);

/*  LayerNorm fused with the per-token LSQ that consumes its output (Block.norm1 -> attention input quantiser
 *  attention.py:177, Block.norm2 -> fc1's input quantiser qlinear.py:66-68): codes[r][c] = LSQ(LN(x [+ res])[r][c] +
 *  b4[c]; step lsq_s[r % S], range [lo, hi]) -- the same codes ofq_lsq_fwd produces from the LayerNorm output, which
 *  is only written when y != NULL.  The backward takes the gradient of the quantised tensor, recomputes the
 *  LayerNorm output from (x, mean, rstd), applies ofq_lsq_bwd's arithmetic and the LayerNorm backward in one pass:
 *  dx [+ dres], dgamma, dbeta, db4 (optional: the same column sums as dbeta, for move_b4), ds[S], dbaft[cols]. */
int ofq_layernorm_lsq_fwd(const float* x, const float* res, const float* gamma, const float* beta, float* y, float* xsum,
                          float* mean, float* rstd, int8_t* codes, const float* lsq_s, int64_t S, float gscale,
                          const float* b4, int lo, int hi, int64_t rows, int64_t cols, int64_t ldx, float eps,
                          ofq_stream_t stream);
size_t ofq_layernorm_lsq_bwd_ws_bytes(int64_t rows, int64_t cols);
int ofq_layernorm_lsq_bwd(const float* gq, const float* x, const float* mean, const float* rstd, const float* gamma,
                          const float* beta, const float* dres, const float* lsq_s, int64_t S, float gscale, const float* b4,
                          int lo, int hi, float* dx, float* dgamma, float* dbeta, float* db4, float* ds, float* dbaft,
                          int64_t rows, int64_t cols, int64_t ldx, int64_t ldg, void* ws, size_t ws_bytes, void* amax_out, ofq_stream_t stream);
/*  Round 6: the same pair with token permutations folded in -- Swin's LayerNorm -> shifted-window partition -> input quantiser
 *  (swin_attention_and_mlp.py:312-323, swin.py:103-131) and the window reverse in front of the residual add (swin.py:160-170),
 *  which the reference runs as roll / view / permute copies.  The R rows are images of perm_n tokens (R % perm_n == 0,
 *  perm_n % S == 0).  q_perm[t]: the row, inside its image, of token t's QUANTISED form -- its codes, the gradient gq of the
 *  quantised values, the step index (row % S) and the step-gradient partial all live there (window-major order); res_perm[t]:
 *  the row of token t in the forward's `res` operand and in the backward's dres_out (a second copy of dx's rows: the gradient of
 *  `res` in its producer's order).  Either permutation may be NULL (identity); x, xsum, mean, rstd, dres, dx: token order. */
int ofq_layernorm_lsq_fwd_perm(const float* x, const float* res, const float* gamma, const float* beta, float* y, float* xsum,
                               float* mean, float* rstd, int8_t* codes, const float* lsq_s, int64_t S, float gscale,
                               const float* b4, int lo, int hi, int64_t rows, int64_t cols, int64_t ldx, float eps,
                               const int* q_perm, const int* res_perm, int64_t perm_n, ofq_stream_t stream);
int ofq_layernorm_lsq_bwd_perm(const float* gq, const float* x, const float* mean, const float* rstd, const float* gamma,
                               const float* beta, const float* dres, const float* lsq_s, int64_t S, float gscale, const float* b4,
                               int lo, int hi, float* dx, float* dgamma, float* dbeta, float* db4, float* ds, float* dbaft,
                               int64_t rows, int64_t cols, int64_t ldx, int64_t ldg, void* ws, size_t ws_bytes, void* amax_out,
                               const int* q_perm, const int* res_perm, int64_t perm_n, float* dres_out, ofq_stream_t stream);

/* ---- AdamW over many tensors in one launch (train.py:662, :933: timm create_optimizer_v2 -> torch.optim.AdamW), with
 *  the CGA freeze folded in (cga.py:962-964, :994-997): where frozen[i] != 0 the gradient is masked before the moment
 *  updates and the weight is left untouched.  `tensors` is a HOST array of n_tensors entries
 *  {float* p; const float* g; float* m; float* v; const float* frozen_or_NULL; int64_t n} (device pointers inside); the
 *  descriptors travel in kernel arguments, 40 tensors per launch.  Per element:
 *  p *= 1 - lr*wd; m += (g - m)(1 - b1); v = b2 v + (1 - b2) g^2; p -= (lr / bc1) m / (sqrt(v)/sqrt(bc2) + eps). */
int64_t ofq_adamw_tensor_entry_bytes(void);
int ofq_adamw_multi(const void* tensors, int64_t n_tensors, float lr, double beta1, double beta2, float eps, float weight_decay,
                    double bias_correction1, double bias_correction2, ofq_stream_t stream);
/*  hipGraph form of the same step: a captured launch must not carry the per-step lr / bias corrections in its arguments,
 *  so the eight scalars {lr, 1-b1, b2, 1-b2, eps, wd, bc1, sqrt(bc2)} live in DEVICE memory.  ofq_adamw_hyper_pack forms
 *  their HOST image exactly as ofq_adamw_multi does (bit-identical updates); ofq_store_f32 writes up to 32 floats to device
 *  memory with the values travelling in the kernel arguments (legal beside a graph replay, nothing to keep alive). */
int ofq_adamw_hyper_pack(float* host8, float lr, double beta1, double beta2, float eps, float weight_decay,
                         double bias_correction1, double bias_correction2);
int ofq_adamw_multi_dev(const void* tensors, int64_t n_tensors, const float* hyper_dev, ofq_stream_t stream);
int ofq_store_f32(float* dst_dev, const float* host_vals, int n, ofq_stream_t stream);
/*  Step guard (round 6).  ofq_step_guard ORs up to 32 device words -- the error words of the stream-K workspaces (byte offset
 *  16384 of a workspace of ofq_qgemm_bf16s_nt_sk), the flag elements a data-parallel wrapper appends to its gradient buckets
 *  (averaged over the ranks by the bucket's own all-reduce, so every rank reads the same value) -- compared as bits without a
 *  float's sign bit; if any is set: loss[0] <- NaN, guard_u32[0] <- 1, flag_f32[0] <- 1.0, else guard_u32[0] <- 0,
 *  flag_f32[0] <- 0.0 (each output optional; one thread, no host sync, capturable).  The _g forms of the AdamW step take such
 *  a word: while it is non-zero the launch leaves p, m and v untouched, so a step whose gradients are invalid on ANY rank
 *  updates nothing on EVERY rank (the reference has no counterpart: torch's GEMMs cannot time out). */
int ofq_step_guard(const void* const* words, int n_words, float* loss, void* guard_u32, float* flag_f32, ofq_stream_t stream);
int ofq_adamw_multi_g(const void* tensors, int64_t n_tensors, float lr, double beta1, double beta2, float eps, float weight_decay,
                      double bias_correction1, double bias_correction2, const void* guard, ofq_stream_t stream);
int ofq_adamw_multi_dev_g(const void* tensors, int64_t n_tensors, const float* hyper_dev, const void* guard, ofq_stream_t stream);

/* ---- K16  CGA: freeze_outside_boundary_weight_idx cga.py:450-469 and the step hooks cga.py:962-964,
 *  :994-997.  frozen[r][c] in {0,1}; range_ws: 2 ints of scratch (global min / max level). */
int ofq_cga_freeze_mask(const float* W, int64_t rows, int64_t cols, int bits, float boundary_range,
                        float* frozen, int32_t* range_ws, ofq_stream_t stream);
/*  The same mask for every CGA tensor of the model in three launches per 40 tensors: `tensors` is a HOST array of
 *  {const float* W; float* frozen; int32_t* range_ws; int64_t rows; int64_t cols} (device pointers inside). */
int64_t ofq_cga_tensor_entry_bytes(void);
int ofq_cga_freeze_mask_multi(const void* tensors, int64_t n_tensors, int bits, float boundary_range, ofq_stream_t stream);
/*  grad *= (1-frozen); saved = W*frozen   (before optimizer.step) */
int ofq_cga_mask_grad_save(float* grad, const float* W, const float* frozen, float* saved, int64_t n,
                           ofq_stream_t stream);
/*  W = W*(1-frozen) + saved               (after optimizer.step) */
int ofq_cga_restore(float* W, const float* frozen, const float* saved, int64_t n, ofq_stream_t stream);

/* ---- amax_out of the backward kernels that WRITE a gradient tensor (ofq_lsq_bwd: dx, ofq_layernorm_bwd / _lsq_bwd: dx,
 *  ofq_softmax_lsq_bwd / ofq_qattn_dp_softmax_bwd: dscores, ofq_qgemm_i8_lsq_bwd: dy): optional device word, zeroed by the
 *  caller; the kernel raises it to the bits of max |element written| (atomic maxima on the bit patterns: order-independent,
 *  deterministic).  The two-plane fp16 GEMMs that consume the tensor (ofq_qgemm_bf16s_nt / _nt_sk / _tn / _tn_group,
 *  ofq_qattn_dqkx_bf16s / _dxq_bf16s with `amax`) take their power-of-two scale from it; ofq_absmax_f32 computes the same word
 *  for a tensor whose producer did not. */
/* ---- deferred second-stage sums.  The backward kernels of the quantisers (ofq_lsq_bwd, ofq_layernorm_lsq_bwd,
 *  ofq_layernorm_bwd, ofq_qgemm_i8_lsq_bwd, ofq_softmax_lsq_bwd, ofq_qattn_dp_softmax_bwd) finish with a small fixed-order
 *  reduction of their per-workgroup partials into d(step) / d(offset) / d(gamma) / d(beta) -- parameter gradients
 *  (lsq.py:593-601, torch.nn.LayerNorm under autograd) that nobody reads before the optimiser step or the gradient
 *  all-reduce, ~95 launches of 5-15 us per DeiT-S step.  Between ofq_sum_defer(1) and ofq_sum_defer(0) these reductions are
 *  queued on the host instead of launched; ofq_sum_flush launches everything queued, up to 40 reductions per launch, each
 *  with the lane layout its own launch would have used (the result is the immediate one bit for bit).  The caller keeps
 *  the workspaces of the queued calls alive and untouched until the flush, and nothing may read their ds / db outputs
 *  before it.  ofq_sum_pending: number of queued reductions; ofq_sum_defer(-1) discards the queue without launching
 *  (a backward pass that raised).  Host state is per process (one process per GPU). */
void ofq_sum_defer(int on);
int ofq_sum_pending(void);
int ofq_sum_flush(ofq_stream_t stream);

/* ---- exact (erf) GELU, y = gelu(x) elementwise (x may alias y): activation of the fp32 KD teacher's MLP
 *  (train.py:428-442, :906-910; deit_vision_transformer.py:44-62), whose forward otherwise runs on ofq_gemm_f32,
 *  ofq_layernorm_fwd and ofq_softmax_lsq_fwd's probabilities (ofq_amd/teacher.py). */
/* ---- KD teacher (train.py:906-910: the frozen fp32 DistilledVisionTransformer runs in every step), its attention in ONE
 *  launch: out[(b, n), h d + c] = sum_m softmax_m(scale * q[b,n,h,:] . k[b,m,h,:]) v[b,m,h,c] on the fp32 activations
 *  qkv [B N][3 H d] (q | k | v column thirds).  d == 64, N <= 224.  Both products run as three fp16-plane products per
 *  algorithmic one on tile-local power-of-two scales (fp32-grade: the dropped lo.lo term is 2^-22 of a product), the scores
 *  and probabilities stay in registers. */
int ofq_attn_f32_fwd(const float* qkv, float* out, int64_t B, int64_t H, int64_t N, int64_t d, float scale, ofq_stream_t stream);
int ofq_gelu_fwd(const float* x, float* y, int64_t n, void* amax_out, ofq_stream_t stream);      /* amax_out: optional word group, max |y| */

/*  KD loss of the shipped recipes (KDLossSoftandHard, src/quantization/utils.py:59-77, train.py:906-913), value and gradients:
 *            loss = mean_b(-sum_k softmax(teacher_b)[k] log_softmax(dist_b)[k]) + mean_b(-log_softmax(cls_b)[target_b]);
 *            dcls / ddist [B][K] = d loss / d logits (contiguous), row_ws: 2 B + 1 floats.  Labels as nn.CrossEntropyLoss takes them
 *            (utils.py:70): target == -100 (ignore_index) rows contribute nothing and the hard term is the mean over the other rows
 *            (row_ws[2 B] = B / their count, to be passed as cls_scale below; no such row: exactly 1); any other label outside
 *            [0, K), where the stock op traps, makes the loss NaN.  ofq_kd_loss_bwd: out_dist = grad_loss[0] * ddist,
 *            out_cls = grad_loss[0] * dcls * cls_scale[0] (cls_scale NULL: 1). */
int ofq_kd_loss_fwd(const float* cls_logits, const float* dist_logits, const float* teacher_logits, const int64_t* target,
                    float* loss, float* dcls, float* ddist, float* row_ws, int64_t B, int64_t K, int64_t ld_cls, int64_t ld_dist,
                    int64_t ld_teacher, ofq_stream_t stream);
int ofq_kd_loss_bwd(const float* grad_loss, const float* dcls, const float* ddist, const float* cls_scale, float* out_cls,
                    float* out_dist, int64_t n, ofq_stream_t stream);
/*  token assembly of the (distilled) ViT (deit.py:32-44): out[b] = cat(cls, [dist,] patches[b]) + pos; patches [B][T - ntok][C]
 *            (ntok = 2 with a distillation token, 1 with dist_token NULL), cls / dist [C], pos [T][C], out [B][T][C]; C % 4 == 0. */
int ofq_assemble_tokens(const float* patches, const float* cls_token, const float* dist_token, const float* pos, float* out,
                        int64_t B, int64_t T, int64_t C, ofq_stream_t stream);
/* ---- token permutation y[b][i][:] = x[b][idx[i]][:] (x, y: [B][N][C] fp32, C % 4 == 0, x != y; idx: int32[N], a
 *  permutation of 0..N-1).  Swin's shifted-window partition and its inverse when nothing is padded
 *  (src/swin.py:103-131 pad -> roll -> view -> permute -> reshape, :160-170 the way back): ONE row gather each way; the
 *  backward of either is the same call with the inverse permutation. */
int ofq_permute_tokens(const float* x, const int32_t* idx, float* y, int64_t B, int64_t N, int64_t C, ofq_stream_t stream);
/*  GEMM of the frozen fp32 teacher on the bf16 matrix cores: C[M][N] = A[M][K] . B[N][K]^T + bias[N], A fp32 (split into
 *  three bf16 planes inside the kernel), B pre-split by ofq_split_f32_bf16x3 into planes[3][plane_stride] (bf16, B = sum of
 *  the planes exactly).  products = 9: every plane pair, i.e. the exact product of the fp32 values up to fp32 accumulation;
 *  6: the six leading pairs (dropped terms <= 2^-24 of a product).  K % 8 == 0. */
int ofq_split_f32_bf16x3(const float* x, void* planes, int64_t n, int64_t plane_stride, ofq_stream_t stream);
int ofq_gemm_bf16x3x3_nt(const float* A, const void* B_planes, float* C, const float* bias, int products, int64_t M, int64_t N,
                         int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int64_t plane_stride, ofq_stream_t stream);

/* ---- on-device input pipeline (train.py:579-629 -> timm 0.5.4 FastCollateMixup + PrefetchLoader + RandomErasing): one
 *  pass over a decoded uint8 batch in [B][C][H][W]: mixup / cutmix with the mirrored sample B-1-b in uint8 space
 *  (use_mix; use_cutmix selects the box copy, otherwise rint(lam * x + one_minus_lam * x'), both scalars as float32), per-channel normalisation
 *  (x - mean255[c]) / std255[c] in fp32 (mean255 / std255: HOST arrays of C floats), and random erasing: rects = DEVICE
 *  int32 [B][4] {top, left, h, w} (h = 0: sample not erased) or NULL, the rectangle is overwritten with noise[b][c][y][x]
 *  (DEVICE fp32, same shape as out).  The random decisions are taken on the host (ofq_amd/data.py).  W % 4 == 0. */
int ofq_input_pipeline_u8(const uint8_t* in, float* out, int64_t B, int64_t C, int64_t H, int64_t W, const float* mean255,
                          const float* std255, int use_mix, int use_cutmix, float lam, float one_minus_lam, int yl, int yh, int xl,
                          int xh, const int32_t* rects, const float* noise, ofq_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* OFQ_HIP_H */
