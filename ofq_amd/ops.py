"""Thin Python launchers over the C ABI (include/ofq_hip.h).  No arithmetic happens here: each function
checks shapes, allocates outputs through torch's caching allocator, and enqueues HIP kernels on
torch's current stream.  CPU tensors are rejected: the product path has no CPU fallback.
"""
import math
import os
import ctypes as C

import torch

from . import _lib
from ._lib import GemmDesc

_lib_handle = None
_ws = {}


class KernelTimer:
    """Optional HIP-event instrumentation of one kernel class (used by bench.py for the roofline of the
    dominant kernel).  Events are recorded on torch's current stream, which is the stream the C ABI launches on."""

    def __init__(self):
        self.records = []            # (start_event, end_event, work_units)

    def bracket(self, units):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.records.append((s, e, units))
        return s, e

    def summary(self):
        torch.cuda.synchronize()
        ms = [s.elapsed_time(e) for s, e, _ in self.records]
        units = [u for _, _, u in self.records]
        n = len(ms)
        return {"launches": n, "total_ms": sum(ms), "avg_ms": sum(ms) / max(n, 1), "total_units": float(sum(units)),
                "avg_units": float(sum(units)) / max(n, 1)}


GEMM_TIMER = None   # set to a KernelTimer to time every ofq_gemm_f32 launch
TIMERS = None       # set to {} to time every matrix-core kernel class separately (bench.py roofline)


class _Timed:
    """with _Timed("class", flops): launch   -> HIP events around the launch when TIMERS is enabled."""
    __slots__ = ("ev",)

    def __init__(self, name, units):
        t = TIMERS
        if t is None:
            self.ev = None
        else:
            kt = t.get(name)
            if kt is None:
                kt = t[name] = KernelTimer()
            self.ev = kt.bracket(units)

    def __enter__(self):
        if self.ev is not None:
            self.ev[0].record()

    def __exit__(self, *a):
        if self.ev is not None:
            self.ev[1].record()
        return False


def lib():
    global _lib_handle
    if _lib_handle is None:
        _lib_handle = _lib.load()
    return _lib_handle


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_get_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """Raw hipStream_t of torch's current stream.  torch.cuda.current_stream() builds a Stream object per call (8 us, 3 ms
    of host time per training step); the raw getter is the same value without the wrapper."""
    if _raw_stream is not None and _get_device is not None:
        return _raw_stream(_get_device())
    return torch.cuda.current_stream().cuda_stream


LAUNCHES = [0]          # ctypes launches so far (engine.GraphedTrainStep: "has this capture segment recorded anything yet?")


def _chk(rc, what):
    LAUNCHES[0] += 1
    if rc != 0:
        raise RuntimeError("ofq_amd: %s failed with code %d" % (what, rc))


def _dev(t, name="tensor"):
    if not t.is_cuda:
        raise RuntimeError("ofq_amd: %s must live on a HIP device (got %s); the MI355X path has no CPU fallback"
                           % (name, t.device))
    if t.dtype != torch.float32:
        raise RuntimeError("ofq_amd: %s must be float32 (got %s)" % (name, t.dtype))
    return t


def _p(t):
    return 0 if t is None else t.data_ptr()


# ---- deferred second-stage sums (include/ofq_hip.h: ofq_sum_defer / ofq_sum_flush) ------------------------------------
_SUM_DEFER = [False]
_WS_POISON = os.environ.get("OFQ_WS_POISON") is not None     # test hook (tests/test_graph_gpu.py)
_SUM_KEEP = []           # workspaces (per-workgroup partials) of the calls whose second stage is queued


class deferred_sums:
    """with ops.deferred_sums(): the backward kernels called inside queue their second-stage reductions (d step, d offset,
    d gamma, d beta: parameter gradients) instead of launching them; ops.sum_flush() launches the queue, forty reductions
    per launch.  Inside the block every ops.workspace() call gets a private buffer that lives until the flush.  The
    caller guarantees that nothing reads those gradient outputs before the flush (functional._sums_deferrable)."""

    def __enter__(self):
        _SUM_DEFER[0] = True
        lib().ofq_sum_defer(1)
        return self

    def __exit__(self, *exc):
        lib().ofq_sum_defer(0)
        _SUM_DEFER[0] = False
        return False


def sum_flush():
    """Launch every queued second-stage reduction (no-op when nothing is queued) and release the kept workspaces."""
    try:
        if _SUM_KEEP or lib().ofq_sum_pending():
            _chk(lib().ofq_sum_flush(_stream()), "ofq_sum_flush")
    finally:
        del _SUM_KEEP[:]


def sum_pending():
    """Queued second-stage reductions / kept workspaces (host-side counters, no device sync)."""
    return len(_SUM_KEEP) + int(lib().ofq_sum_pending())


def sum_drop():
    """Forget the queued reductions without launching them (a backward pass that raised: the tensors they would write may
    have been released already)."""
    lib().ofq_sum_defer(-1)
    _SUM_DEFER[0] = False
    del _SUM_KEEP[:]


def workspace(nbytes, device):
    """Per-device scratch, grown on demand.  Kernels that use it are serialised on one stream."""
    if _SUM_DEFER[0]:
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        if _WS_POISON:
            buf.fill_(0xFF)               # test hook: every float of the partial buffers starts as a NaN
        _SUM_KEEP.append(buf)
        return buf
    key = (device.index, _stream())
    buf = _ws.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws[key] = buf
    return buf


# ------------------------------------------------------------------------------------------------ StatsQ
def statsq_fwd(W, bits, want_levels=False, scale=None, odd_codes=False):
    _dev(W, "weight")
    W = W.contiguous()
    rows, cols = W.shape
    out = torch.empty_like(W)
    given = scale is not None
    s = scale.contiguous() if given else torch.empty(rows, dtype=torch.float32, device=W.device)
    lv = torch.empty((rows, cols), dtype=torch.int8, device=W.device) if want_levels else None
    _chk(lib().ofq_statsq_fwd(W.data_ptr(), rows, cols, bits, out.data_ptr(), s.data_ptr(), _p(lv), int(given),
                              int(odd_codes), _stream()), "ofq_statsq_fwd")
    return out, s, lv


def statsq_codes_fwd(W, bits, rvec=None, need_values=True, want_T=True):
    """One launch for the code path of a quantised linear layer: returns (Wq or placeholder, scale, codes int8 [out][in],
    codesT bf16 [in][out] or None, r[out] = codes @ rvec or None)."""
    _dev(W, "weight")
    W = W.contiguous()
    rows, cols = W.shape
    dev = W.device
    out = torch.empty_like(W) if need_values else None
    s = torch.empty(rows, dtype=torch.float32, device=dev)
    codes = torch.empty((rows, cols), dtype=torch.int8, device=dev)
    codesT = torch.empty((cols, rows), dtype=torch.float16 if (GRAD_PLANES == 2 and "dx" in _DBG_F16) else torch.bfloat16, device=dev) if want_T else None
    r = torch.empty(rows, dtype=torch.float32, device=dev) if rvec is not None else None
    _chk(lib().ofq_statsq_codes_fwd(W.data_ptr(), rows, cols, bits | (0x100 if (GRAD_PLANES == 2 and "dx" in _DBG_F16) else 0), _p(out), s.data_ptr(), codes.data_ptr(), _p(codesT),
                                    _p(rvec), _p(r), _stream()), "ofq_statsq_codes_fwd")
    if out is None:
        out = placeholder((rows, cols), dev)
    return out, s, codes, codesT, r


def statsq_codes_multi(items):
    """items: list of (W, bits, rvec or None, want_T).  One or two launches for all of them; returns a list of
    (scale, codes int8 [out][in], codesT bf16 [in][out] or None, r or None), values identical to statsq_codes_fwd."""
    import ctypes as C
    import struct
    esz = int(lib().ofq_statsq_tensor_entry_bytes())
    assert esz == 72
    outs, blob, keep = [], bytearray(), []
    for (W, bits, rvec, want_T) in items:
        _dev(W, "weight")
        Wc = W.contiguous()
        rows, cols = Wc.shape
        dev = Wc.device
        s = torch.empty(rows, dtype=torch.float32, device=dev)
        codes = torch.empty((rows, cols), dtype=torch.int8, device=dev)
        codesT = torch.empty((cols, rows), dtype=torch.float16 if (GRAD_PLANES == 2 and "dx" in _DBG_F16) else torch.bfloat16, device=dev) if want_T else None
        r = torch.empty(rows, dtype=torch.float32, device=dev) if rvec is not None else None
        blob += struct.pack("<6Q3q", Wc.data_ptr(), s.data_ptr(), codes.data_ptr(), _p(codesT) or 0, _p(rvec) or 0, _p(r) or 0,
                            rows, cols, int(bits) | (0x100 if (GRAD_PLANES == 2 and "dx" in _DBG_F16) else 0))
        keep.append(Wc)
        outs.append((s, codes, codesT, r))
    buf = (C.c_char * len(blob)).from_buffer(blob)
    _chk(lib().ofq_statsq_codes_multi(C.addressof(buf), len(items), _stream()), "ofq_statsq_codes_multi")
    return outs


# ------------------------------------------------------------------------------------------------ LSQ
class LsqGeom:
    """How a tensor maps onto the kernel's [outer][S][inner] view (include/ofq_hip.h, ofq_lsq_fwd)."""
    __slots__ = ("outer", "S", "inner", "ldx", "ldy", "bias_len", "mode", "lo", "hi", "gscale", "prologue", "patch")

    def __init__(self, outer, S, inner, bias_len, mode, lo, hi, M, prologue=0, ldx=None, ldy=None):
        self.outer, self.S, self.inner = int(outer), int(S), int(inner)
        self.ldx = int(ldx if ldx is not None else inner)
        self.ldy = int(ldy if ldy is not None else inner)
        self.bias_len = int(bias_len)
        self.mode, self.lo, self.hi = int(mode), int(lo), int(hi)
        self.gscale = 1.0 / math.sqrt(hi * M)          # lsq.py:582-591: 1/sqrt(thd_pos * M), python double
        self.prologue = int(prologue)
        self.patch = None                              # (width, ph, pw): the image quantiser's output / gradient in im2col order


def placeholder(shape, device):
    """Zero-stride tensor of the given shape: an autograd edge carrier for values that only exist as integer codes."""
    return torch.empty(1, dtype=torch.float32, device=device).expand(*shape)


def lsq_fwd(x, s, b4, baft, g, y=None, want_codes=False, need_values=True):
    _dev(x, "x")
    yptr = 0
    if not need_values:
        assert want_codes
        y = placeholder((g.outer * g.S, g.ldy), x.device)
    else:
        if y is None:
            y = torch.empty((g.outer * g.S, g.ldy), dtype=torch.float32, device=x.device)
        yptr = y.data_ptr()
    codes = torch.empty((g.outer * g.S, g.inner), dtype=torch.int8, device=x.device) if want_codes else None
    if g.patch is not None:
        # ofq_lsq_fwd_patch: the same elements, written in the convolution's im2col order (the callers view y / codes as
        # [images * gh * gw][channels * ph * pw])
        if g.mode != 0 or g.prologue != 0 or g.ldx != g.inner or g.ldy != g.inner:
            raise RuntimeError("ofq_amd: lsq_fwd: the patch layout is the image quantiser's (per-row step, dense rows)")
        _chk(lib().ofq_lsq_fwd_patch(x.data_ptr(), s.data_ptr(), _p(b4), _p(baft), yptr, _p(codes), g.outer, g.S, g.inner,
                                     g.bias_len, g.lo, g.hi, g.gscale, g.patch[0], g.patch[1], g.patch[2], _stream()),
             "ofq_lsq_fwd_patch")
        return y, codes
    _chk(lib().ofq_lsq_fwd(x.data_ptr(), s.data_ptr(), _p(b4), _p(baft), yptr, _p(codes), g.outer, g.S,
                           g.inner, g.ldx, g.ldy, g.bias_len, g.mode, g.lo, g.hi, g.gscale, g.prologue, _stream()),
         "ofq_lsq_fwd")
    return y, codes


def lsq_bwd(gy, x, s, b4, g, dx=None, want_bias_grads=True, amax_word=None, db4_out=None):
    """db4_out: write the offset gradient there (a contiguous slice of a larger gradient tensor) instead of a fresh tensor.
    amax_word: raise THIS maximum word instead of a fresh one (the kernels raise it with an atomic max, so several launches that
    write column slices of one tensor can share the word of the whole tensor; the caller tags the tensor itself)."""
    _dev(gy, "grad")
    dev = x.device
    if dx is None:
        dx = torch.empty((g.outer * g.S, g.ldx), dtype=torch.float32, device=dev)
    ds = torch.empty_like(s)
    has_bias = g.bias_len > 0 and want_bias_grads
    db4 = (db4_out if db4_out is not None else torch.empty(g.bias_len, dtype=torch.float32, device=dev)) if has_bias else None
    dbaft = torch.empty(g.bias_len, dtype=torch.float32, device=dev) if has_bias else None
    nbytes = lib().ofq_lsq_bwd_ws_bytes(g.outer, g.S, g.inner, g.bias_len, g.mode)
    ws = workspace(nbytes, dev)
    am = amax_word if amax_word is not None else amax_out(dev)
    if g.patch is not None:
        _chk(lib().ofq_lsq_bwd_patch(gy.data_ptr(), x.data_ptr(), s.data_ptr(), _p(b4), dx.data_ptr(), ds.data_ptr(), _p(db4),
                                     _p(dbaft), g.outer, g.S, g.inner, g.bias_len, g.lo, g.hi, g.gscale, g.patch[0], g.patch[1],
                                     g.patch[2], ws.data_ptr(), ws.numel(), _p(am), _stream()), "ofq_lsq_bwd_patch")
        if am is not None and amax_word is None:
            tag_amax(dx, am)
        return dx, ds, db4, dbaft
    _chk(lib().ofq_lsq_bwd(gy.data_ptr(), x.data_ptr(), s.data_ptr(), _p(b4), dx.data_ptr(), ds.data_ptr(), _p(db4),
                           _p(dbaft), g.outer, g.S, g.inner, g.ldx, g.ldy, g.bias_len, g.mode, g.lo, g.hi, g.gscale,
                           g.prologue, ws.data_ptr(), ws.numel(), _p(am), _stream()), "ofq_lsq_bwd")
    if am is not None and amax_word is None:
        tag_amax(dx, am)
    return dx, ds, db4, dbaft


# ------------------------------------------------------------------------------------------------ softmax + LSQ
def softmax_lsq_fwd(scores, s, rows, n, ld, S, alpha, hi, M, want_codes=False, need_values=True, addend=None):
    prob = torch.empty_like(scores)
    y = torch.empty_like(scores) if need_values else placeholder(scores.shape, scores.device)
    gscale = 1.0 / math.sqrt(hi * M)
    codes = torch.empty(scores.shape, dtype=torch.uint8, device=scores.device) if want_codes else None
    rsum = torch.empty(rows, dtype=torch.float32, device=scores.device) if want_codes else None
    _chk(lib().ofq_softmax_lsq_fwd(scores.data_ptr(), s.data_ptr(), prob.data_ptr(), y.data_ptr() if need_values else 0,
                                   rows, n, ld, S, alpha, hi, gscale, _p(codes), _p(rsum), _p(addend),
                                   addend.shape[0] if addend is not None else 1, _stream()),
         "ofq_softmax_lsq_fwd")
    if want_codes:
        return prob, y, codes, rsum
    return prob, y


def softmax_lsq_bwd(g, prob, s, rows, n, ld, S, alpha, hi, M, inplace=True, want_rowsum=False):
    dsc = g if inplace else torch.empty_like(g)
    ds = torch.empty_like(s)
    gscale = 1.0 / math.sqrt(hi * M)
    rs = torch.empty(rows, dtype=torch.float32, device=g.device) if want_rowsum else None
    ws = workspace(lib().ofq_softmax_lsq_bwd_ws_bytes(rows), g.device)
    am = amax_out(g.device)
    _chk(lib().ofq_softmax_lsq_bwd(g.data_ptr(), prob.data_ptr(), s.data_ptr(), dsc.data_ptr(), ds.data_ptr(), rows, n,
                                   ld, S, alpha, hi, gscale, _p(rs), ws.data_ptr(), ws.numel(), _p(am), _stream()),
         "ofq_softmax_lsq_bwd")
    if am is not None:
        tag_amax(dsc, am)
    if want_rowsum:
        return dsc, ds, rs
    return dsc, ds


# ------------------------------------------------------------------------------------------------ GEMM
def gemm(A, B, Cout, M, N, K, lda, ldb, ldc, transA=False, transB=False, bias=None, nb0=1, nb1=1,
         sA=(0, 0), sB=(0, 0), sC=(0, 0), nkb=1, sAk=0, sBk=0, split_k=1, alpha=1.0, accumulate=False,
         offA=0, offB=0, offC=0, tile_hint=0):
    """Raw launcher; A/B/Cout are base tensors, off* are element offsets into them."""
    d = GemmDesc()
    d.A = A.data_ptr() + 4 * offA
    d.B = B.data_ptr() + 4 * offB
    d.C = Cout.data_ptr() + 4 * offC
    d.bias = _p(bias)
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc = lda, ldb, ldc
    d.transA, d.transB = int(transA), int(transB)
    d.nb0, d.nb1 = nb0, nb1
    d.sA0, d.sA1 = sA
    d.sB0, d.sB1 = sB
    d.sC0, d.sC1 = sC
    d.nkb, d.sAk, d.sBk = nkb, sAk, sBk
    d.split_k, d.alpha, d.accumulate = split_k, alpha, int(accumulate)
    d.tile_hint = tile_hint
    wsb = lib().ofq_gemm_ws_bytes(C.byref(d))
    ws = workspace(wsb, A.device) if wsb else None
    with _Timed("gemm_f32 (v_mfma_f32_32x32x2_f32)", 2.0 * M * N * K * nb0 * nb1 * nkb):
        _chk(lib().ofq_gemm_f32(C.byref(d), _p(ws), ws.numel() if ws is not None else 0, _stream()), "ofq_gemm_f32")
    return Cout


def _pick_split(M, N, K, target_wgs=512, bk=32):
    tiles = ((M + 127) // 128) * ((N + 127) // 128 if N > 64 else 1)
    split = max(1, min(target_wgs // max(tiles, 1), (K + bk - 1) // bk // 4))
    return int(split)


def linear_fwd(x2d, W, bias=None):
    """y[M,N] = x2d[M,K] @ W[N,K]^T + bias   (F.linear, qlinear.py:69-71)"""
    M, K = x2d.shape
    N = W.shape[0]
    y = torch.empty((M, N), dtype=torch.float32, device=x2d.device)
    return gemm(x2d, W, y, M, N, K, x2d.stride(0), W.stride(0), N, transB=True, bias=bias)


def linear_bwd_input(dy, W, out=None, accumulate=False):
    """dx[M,K] = dy[M,N] @ W[N,K]"""
    M, N = dy.shape
    K = W.shape[1]
    if out is None:
        out = torch.empty((M, K), dtype=torch.float32, device=dy.device)
    # few output tiles and a long contraction (classifier heads: 128 x 384 over K = 1000 ran 110 us on three workgroups)
    split = 1 if accumulate else _pick_split(M, K, N)
    return gemm(dy, W, out, M, K, N, dy.stride(0), W.stride(0), out.stride(0), accumulate=accumulate, split_k=split)


def linear_bwd_weight(dy, x2d):
    """dW[N,K] = dy[M,N]^T @ x2d[M,K]   (split-K over the token dimension)"""
    M, N = dy.shape
    K = x2d.shape[1]
    dW = torch.empty((N, K), dtype=torch.float32, device=dy.device)
    return gemm(dy, x2d, dW, N, K, M, dy.stride(0), x2d.stride(0), K, transA=True, split_k=_pick_split(N, K, M))


# ------------------------------------------------------------------------------------------------ code GEMMs
# Planes of the fp32 operand (dY) in the backward code GEMMs.  2 (default since round 5): two fp16 planes of the tensor scaled
# by a power of two taken from its absolute maximum -- fp32-grade on the scale of the tensor (include/ofq_hip.h,
# ofq_qgemm_bf16s_nt), two matrix-core products per element; 3: three bf16 planes, the exact fp32 product (rounds 1-4).
# It decides the format of the TRANSPOSED weight codes the quantisers keep (fp16 / bf16); the GEMM wrappers below follow
# the dtype of the codes they are handed, so tests can drive either form explicitly.
GRAD_PLANES = 3 if os.environ.get("OFQ_GRAD_PLANES") == "3" else 2
_DBG_F16 = {"dx", "dw", "dqkx", "dxq"}      # which GEMM families take the two-plane form (tools/two_rank_determinism.py narrows it)
# Test hook (tests/test_planes_fullsize_gpu.py): PLANE_PROBE(kind, A2d, k_scale or None, scale_axis, ncols) is called with the fp32
# gradient operand of every backward code GEMM that is about to be split into planes -- the operand as the kernel scales it is
# A * k_scale along `scale_axis` (1: per column = contraction index of dX; 0: per row, period len(k_scale), = the tokens of dW).
PLANE_PROBE = None


def _probe(kind, A, k_scale, axis, ncols=None):
    if PLANE_PROBE is not None:
        PLANE_PROBE(kind, A, k_scale, axis, ncols)


def codes_transpose_bf16(codes):
    rows, cols = codes.shape
    out = torch.empty((cols, rows), dtype=torch.bfloat16, device=codes.device)
    _chk(lib().ofq_codes_transpose_bf16(codes.data_ptr(), out.data_ptr(), rows, cols, _stream()), "ofq_codes_transpose_bf16")
    return out


def codes_transpose_f16(codes):
    rows, cols = codes.shape
    out = torch.empty((cols, rows), dtype=torch.float16, device=codes.device)
    _chk(lib().ofq_codes_transpose_f16(codes.data_ptr(), out.data_ptr(), rows, cols, _stream()), "ofq_codes_transpose_f16")
    return out


def codes_transpose_16(codes):
    """The transposed codes in the format of the configured backward form (GRAD_PLANES)."""
    return codes_transpose_f16(codes) if (GRAD_PLANES == 2 and "dx" in _DBG_F16) else codes_transpose_bf16(codes)


# ---- absolute maxima of gradient tensors (the power-of-two scale of the two-plane form) --------------------------------------
# A word per gradient tensor, holding the bits of max |x|: written by the backward kernel that produces the tensor (its amax_out
# argument) or by ofq_absmax_f32, read by the GEMMs that consume the tensor.  It travels as an attribute of the tensor object
# (autograd hands a Function's output to the next Function as the same object, and a reshape in between keeps it as `._base`);
# a tensor without one (an ATen op in between, an accumulated gradient) gets its maximum computed on the spot.
# Inside engine's step the words come from a pool that is zeroed once per backward pass (one fill instead of one per tensor, and
# fixed addresses for a captured step).
_AMAX_POOL = {}
_AMAX_STATE = {"active": False, "next": 0}
AMAX_GROUP = 64 * 32            # device words per tensor: OFQ_AMAX_WORDS slots, OFQ_AMAX_STRIDE words apart (one 128-byte line each)
AMAX_POOL_WORDS = 1024 * AMAX_GROUP     # 8 MB, zeroed once per backward pass


def amax_begin(device):
    """engine._step_compute, right before loss.backward(): zero the pool, hand its words out from the start again."""
    pool = _AMAX_POOL.get(device.index)
    if pool is None:
        pool = _AMAX_POOL[device.index] = torch.zeros(AMAX_POOL_WORDS, dtype=torch.int32, device=device)
    else:
        pool.zero_()
    _AMAX_STATE["active"], _AMAX_STATE["next"] = True, 0


def amax_end():
    _AMAX_STATE["active"] = False


def amax_word(device):
    """A zeroed group of AMAX_GROUP device words for one tensor's maximum."""
    if _AMAX_STATE["active"] and _AMAX_STATE["next"] + AMAX_GROUP <= AMAX_POOL_WORDS and device.index in _AMAX_POOL:
        i = _AMAX_STATE["next"]
        _AMAX_STATE["next"] = i + AMAX_GROUP
        return _AMAX_POOL[device.index][i:i + AMAX_GROUP]
    return torch.zeros(AMAX_GROUP, dtype=torch.int32, device=device)


def tag_amax(t, word):
    """Attach a maximum word to the tensor object, together with the version counter and address it was valid for: autograd's
    input buffers add a second gradient IN PLACE (`old.add_(new)`) when they hold the only reference, and the Python object -- with
    its attribute -- survives that; the in-place add bumps the version counter, and amax_of() then ignores the stale word."""
    t._ofq_amax = (word, t._version, t.data_ptr())
    return t


def _amax_valid(t):
    tag = getattr(t, "_ofq_amax", None)
    if tag is None:
        return None
    word, version, ptr = tag
    if t._version != version or t.data_ptr() != ptr:
        return None                   # written by somebody else since the producer's kernel: the word is no bound any more
    return word


def amax_of(t):
    """The maximum word a producer attached to `t` (or to the tensor `t` is a view of: an upper bound is all that is needed), or
    None when there is none or the tensor has been modified in place since (views share their base's version counter)."""
    w = _amax_valid(t)
    if w is None and t._base is not None:
        w = _amax_valid(t._base)
    return w


def absmax(t2d):
    """max |t2d| into a fresh word (ofq_absmax_f32); the word is attached to the tensor and returned."""
    w = amax_word(t2d.device)
    rows, cols = t2d.shape
    if cols % 4 or t2d.stride(1) != 1 or t2d.stride(0) % 4 or t2d.data_ptr() % 16:
        w[:1].copy_(t2d.detach().abs().max().reshape(1).view(torch.int32))       # odd geometry: the stock reduction
    else:
        _chk(lib().ofq_absmax_f32(t2d.data_ptr(), rows, cols, t2d.stride(0), w.data_ptr(), _stream()), "ofq_absmax_f32")
    tag_amax(t2d, w)
    return w


def amax_for(t2d):
    w = amax_of(t2d)
    return w if w is not None else absmax(t2d)


def amax_out(device):
    """The word a gradient-producing backward kernel is to raise (its amax_out argument), or None in three-plane mode."""
    return amax_word(device) if GRAD_PLANES == 2 else None


def rowdot_i8(codes, vec):
    rows, cols = codes.shape
    out = torch.empty(rows, dtype=torch.float32, device=codes.device)
    _chk(lib().ofq_rowdot_i8(codes.data_ptr(), vec.data_ptr(), out.data_ptr(), rows, cols, _stream()), "ofq_rowdot_i8")
    return out


def qgemm_i8_nt(xcodes, wcodes, bias, col_scale, col_mult, r, lsq_s, S, gscale, fuse=None, store_y=True):
    """y = col_mult*col_scale[n]*(a_eff[m % S]*(xcodes @ wcodes^T) + r[n]) + bias[n]
    fuse (optional): LsqGeom-like description of the next layer's input quantiser {s, S, gscale, b4, lo, hi, gelu};
    the kernel then also emits that quantiser's int8 codes of y into fuse["codes_out"].
    store_y=False (needs fuse): only the codes are written; returns None (the backward recomputes y, qgemm_i8_lsq_bwd)."""
    M, K = xcodes.shape
    N = wcodes.shape[0]
    if not store_y and fuse is None:
        raise RuntimeError("ofq_amd: qgemm_i8_nt(store_y=False) needs a fused consumer quantiser")
    y = torch.empty((M, N), dtype=torch.float32, device=xcodes.device) if store_y else None
    with _Timed('qgemm_i8_nt (v_mfma_i32_32x32x32_i8)', 2.0 * M * N * K):
        if fuse is None:
            _chk(lib().ofq_qgemm_i8_nt(xcodes.data_ptr(), wcodes.data_ptr(), y.data_ptr(), _p(bias), col_scale.data_ptr(),
                                       col_mult, _p(r), lsq_s.data_ptr(), S, gscale, M, N, K, xcodes.stride(0),
                                       wcodes.stride(0), N, _stream()), "ofq_qgemm_i8_nt")
        else:
            qc = torch.empty((M, N), dtype=torch.int8, device=xcodes.device)
            _chk(lib().ofq_qgemm_i8_nt_q(xcodes.data_ptr(), wcodes.data_ptr(), _p(y), _p(bias), col_scale.data_ptr(),
                                         col_mult, _p(r), lsq_s.data_ptr(), S, gscale, M, N, K, xcodes.stride(0),
                                         wcodes.stride(0), N, qc.data_ptr(), N, fuse["s"].data_ptr(), fuse["S"],
                                         fuse["gscale"], _p(fuse["b4"]), fuse["lo"], fuse["hi"], int(fuse["gelu"]),
                                         int(fuse.get("rowmul", 1)), int(fuse.get("coldiv", N)), int(fuse.get("colmode", 0)),
                                         _stream()), "ofq_qgemm_i8_nt_q")
            fuse["codes_out"] = qc
    return y


def qgemm_i8_lsq_bwd(gy2d, prod, q, want_bias_grads=True):
    """Backward of [linear layer -> its consumer's input quantiser] with the layer output recomputed from the codes.
    prod: the producing GEMM's operands {xcodes (M,K) int8, wcodes (N,K) int8, bias, w_scale, w_mult, r, act_s, act_S,
    act_gscale}; q: the consumer quantiser as handed to qgemm_i8_nt's `fuse` {s, S, gscale, b4, lo, hi, gelu, rowmul,
    coldiv, colmode}.  Returns (dy (M,N): gradient w.r.t. the layer output, ds, db4, dbaft) like lsq_bwd."""
    xc, wc = prod["xcodes"], prod["wcodes"]
    M, K = xc.shape
    N = wc.shape[0]
    dev = gy2d.device
    if gy2d.shape != (M, N) or gy2d.stride(1) != 1:
        raise RuntimeError("ofq_amd: qgemm_i8_lsq_bwd: gradient of shape %s for a %dx%d layer output" % (tuple(gy2d.shape), M, N))
    dy = torch.empty((M, N), dtype=torch.float32, device=dev)
    ds = torch.empty_like(q["s"])
    has_bias = q["b4"] is not None and want_bias_grads
    db4 = torch.empty(N, dtype=torch.float32, device=dev) if has_bias else None
    dbaft = torch.empty(N, dtype=torch.float32, device=dev) if has_bias else None
    colmode = int(q.get("colmode", 0))
    ws = workspace(lib().ofq_qgemm_i8_lsq_bwd_ws_bytes(M, N, colmode), dev)
    am = amax_out(dev)
    with _Timed('qgemm_i8_lsqbwd (int8 recompute + LSQ backward epilogue)', 2.0 * M * N * K):
        _chk(lib().ofq_qgemm_i8_lsq_bwd(xc.data_ptr(), wc.data_ptr(), _p(prod["bias"]), prod["w_scale"].data_ptr(),
                                        prod["w_mult"], _p(prod["r"]), prod["act_s"].data_ptr(), prod["act_S"],
                                        prod["act_gscale"], M, N, K, xc.stride(0), wc.stride(0), gy2d.data_ptr(),
                                        gy2d.stride(0), dy.data_ptr(), N, q["s"].data_ptr(), q["S"], q["gscale"], _p(q["b4"]),
                                        q["lo"], q["hi"], int(q["gelu"]), int(q.get("rowmul", 1)), int(q.get("coldiv", N)),
                                        colmode, ds.data_ptr(), _p(db4), _p(dbaft), ws.data_ptr(), ws.numel(), _p(am), _stream()),
             "ofq_qgemm_i8_lsq_bwd")
    if am is not None:
        tag_amax(dy, am)
    return dy, ds, db4, dbaft


def attn_f32_ok(N, d):
    return d == 64 and 0 < N <= 224


def attn_f32_fwd(qkv2d, B, H, N, d, scale):
    """softmax(scale q k^T) v per (image, head) on the fp32 teacher's qkv projection (ofq_attn_f32_fwd): (B N, H d)"""
    if qkv2d.shape != (B * N, 3 * H * d) or not qkv2d.is_contiguous():
        raise RuntimeError("ofq_amd: attn_f32_fwd: qkv of shape %s for (%d, %d, %d, %d)" % (tuple(qkv2d.shape), B, H, N, d))
    out = torch.empty((B * N, H * d), dtype=torch.float32, device=qkv2d.device)
    with _Timed('attn_f32_fwd (teacher attention, 3x v_mfma_f32_32x32x16_f16)', 4.0 * B * H * N * N * d):
        _chk(lib().ofq_attn_f32_fwd(qkv2d.data_ptr(), out.data_ptr(), B, H, N, d, float(scale), _stream()), "ofq_attn_f32_fwd")
    return out


def dqkx_lsq_fusable(prod, q, sx, gx, N, C, ldS):
    """May the qkx quantiser's backward form its incoming gradient itself (ofq_qattn_dqkx_lsq_bwd)?  The producer of qkx must
    be the GEMM on THESE x codes / steps, the quantiser the per-(token, head) one, the shapes what the kernel tiles."""
    return (GRAD_PLANES == 2 and "dqkx" in _DBG_F16 and prod is not None and N >= 128 and N % 2 == 0 and C % 128 == 0
            and ldS % 2 == 0 and not q.get("colmode", 0) and not q["gelu"] and prod["act_S"] == N
            and prod["act_s"].data_ptr() == sx.data_ptr() and float(prod["act_gscale"]) == float(gx)
            and prod["xcodes"].shape[1] == C and prod["xcodes"].stride(0) % 16 == 0)


def qattn_dqkx_lsq_bwd(dS, prod, q, bax, B, H, N, C, ldS, want_bias_grads=True, planes=None):
    """ofq_qattn_dqkx_lsq_bwd: (dy, ds, db4, dbaft) of qattn_dqkx -> qgemm_i8_lsq_bwd without the dqkx tensor."""
    xc, wc = prod["xcodes"], prod["wcodes"]
    M, Nout = B * N, H * C
    if xc.shape[0] != M or wc.shape[0] != Nout or q["S"] != N * H or int(q.get("rowmul", 1)) != H:
        raise RuntimeError("ofq_amd: qattn_dqkx_lsq_bwd: operands do not describe a (%d, %d, %d, %d) QKR attention" % (B, H, N, C))
    dev = dS.device
    dy = torch.empty((M, Nout), dtype=torch.float32, device=dev)
    ds = torch.empty_like(q["s"])
    has_bias = q["b4"] is not None and want_bias_grads
    db4 = torch.empty(Nout, dtype=torch.float32, device=dev) if has_bias else None
    dbaft = torch.empty(Nout, dtype=torch.float32, device=dev) if has_bias else None
    ws = workspace(lib().ofq_qgemm_i8_lsq_bwd_ws_bytes(M, Nout, 0), dev)
    amax = scores_amax(dS, N, planes)
    am = amax_out(dev)
    with _Timed('qattn_dqkx_lsqbwd (dqkx GEMM on 2 fp16 planes + int8 recompute + LSQ backward epilogue)',
                2.0 * B * H * N * N * C + 2.0 * M * Nout * C):
        _chk(lib().ofq_qattn_dqkx_lsq_bwd(xc.data_ptr(), wc.data_ptr(), _p(prod["bias"]), prod["w_scale"].data_ptr(),
                                          prod["w_mult"], _p(prod["r"]), prod["act_s"].data_ptr(), prod["act_gscale"], _p(bax),
                                          dS.data_ptr(), ldS, amax.data_ptr(), B, H, N, C, xc.stride(0), wc.stride(0),
                                          dy.data_ptr(), Nout, q["s"].data_ptr(), q["S"], q["gscale"], _p(q["b4"]), q["lo"],
                                          q["hi"], ds.data_ptr(), _p(db4), _p(dbaft), ws.data_ptr(), ws.numel(), _p(am),
                                          _stream()), "ofq_qattn_dqkx_lsq_bwd")
    if am is not None:
        tag_amax(dy, am)
    return dy, ds, db4, dbaft


# ---- stream-K input-gradient GEMM (ofq_qgemm_bf16s_nt_sk) ----------------------------------------------------------------
def _pl(two):
    """What a backward code GEMM issues per algorithmic multiply-add: two fp16 products (round 5) or three bf16 ones."""
    return "2x v_mfma_f32_32x32x16_f16" if two else "3x v_mfma_f32_32x32x16_bf16"


def nt_class(two):
    return 'qgemm_bf16s_nt_wide (linear dX, %s)' % _pl(two)      # bench.py's timer class of the dX GEMMs
NT_SK = os.environ.get("OFQ_NT_SK", "1") != "0"          # A/B switch: "0" = always the one-tile-per-workgroup kernel
_NT_SK_FORCE = os.environ.get("OFQ_NT_SK") == "force"   # test hook: stream-K for every shape it accepts
_sk_ws = {}
_cus = {}


def num_cus(device):
    n = _cus.get(device.index)
    if n is None:
        n = _cus[device.index] = int(torch.cuda.get_device_properties(device).multi_processor_count)
    return n


def _sk_workspace(device):
    """Partial-tile slots + flags of the stream-K kernel: one per (device, stream), zeroed once (the kernel leaves the
    flags zero), never shared with ops.workspace() users."""
    key = (device.index, _stream())
    buf = _sk_ws.get(key)
    if buf is None:
        buf = _sk_ws[key] = torch.zeros(int(lib().ofq_qgemm_bf16s_nt_sk_ws_bytes(num_cus(device))), dtype=torch.uint8, device=device)
    return buf


def nt_sk_error(device):
    """The stream-K kernels' error word (a bounded spin ran out: the owner of a cut tile gave up waiting for a partial and
    stored a wrong tile), over every workspace of the device: 0 = never.  A host sync: called outside timed regions
    (bench.py after the timed loop, train.py at its log points, the tests)."""
    err = 0
    for (idx, _st), buf in _sk_ws.items():
        if idx == device.index:
            err |= int(buf[:32768].view(torch.int32)[4096].item())
    return err


def nt_sk_poison(loss):
    """loss <- NaN when a stream-K hand-off on this device / stream has timed out (one tiny launch, no host sync, capturable):
    engine's step calls it right after the backward pass, so a corrupt gradient is announced by the loss of the same step."""
    buf = _sk_ws.get((loss.device.index, _stream()))
    if buf is not None:
        _chk(lib().ofq_qgemm_bf16s_nt_sk_check(buf.data_ptr(), loss.data_ptr(), _stream()), "ofq_qgemm_bf16s_nt_sk_check")


_guard_words = {}


def step_guard_word(device):
    """The device's step-guard word (int32, 0 = the step is good): written by step_guard() once per step, read by the AdamW launches
    of FusedAdamW (a non-zero word makes them leave p, m, v untouched) and by nt_sk_poll on the host.  Allocated by the first EAGER
    step (never inside a capture: engine's warm-up steps come first)."""
    w = _guard_words.get(device.index)
    if w is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("ofq_amd: the step-guard word must exist before a stream capture (run an eager step first)")
        w = _guard_words[device.index] = torch.zeros(1, dtype=torch.int32, device=device)
    return w


def step_guard_ptr(device):
    """Address of the guard word for the AdamW launches: 0 (no guard) while no step of engine has created it."""
    w = _guard_words.get(device.index)
    return 0 if w is None else w.data_ptr()


def sk_error_word_ptrs(device):
    """Addresses of the error words of the device's stream-K workspaces (int32 word 4096 of each flag area)."""
    return [buf.data_ptr() + 4 * 4096 for (idx, _st), buf in _sk_ws.items() if idx == device.index]


def step_guard(device, loss=None, extra_words=(), flag_out=None, set_guard=True):
    """One tiny launch (no host sync, capturable): OR the stream-K error words of the device and `extra_words` (device addresses:
    the flag elements of the data-parallel buckets, the same on every rank once reduced); if any is set, loss <- NaN, the guard word
    <- 1 (AdamW then updates nothing) and flag_out[0] <- 1.0; else guard word <- 0, flag_out[0] <- 0.0.  engine calls it between
    the gradient collectives and the optimiser (set_guard) and DataParallel when it packs a bucket (flag_out = the bucket's flag)."""
    words = list(sk_error_word_ptrs(device)) + [int(w) for w in extra_words]
    if len(words) > 32:
        words = words[:len(sk_error_word_ptrs(device))] + words[-(32 - len(sk_error_word_ptrs(device))):]     # the LATEST buckets' flags
    arr = (_lib.vp * max(1, len(words)))(*words)
    g = step_guard_word(device) if set_guard else None
    _chk(lib().ofq_step_guard(arr, len(words), _p(loss), _p(g), _p(flag_out), _stream()), "ofq_step_guard")


_sk_poll = {}


def nt_sk_poll(device, wait=False):
    """Host side of the same check without a synchronisation: every call queues an asynchronous copy of the error words to
    pinned memory behind the work enqueued so far and looks at the copy queued by the PREVIOUS call if it has arrived.
    Raises once an error word is seen (after re-zeroing the flag areas, so that training can be resumed from a checkpoint
    without restarting the process); engine.train_step / GraphedTrainStep call it once per step."""
    bufs = [buf[:32768].view(torch.int32)[4096:4097] for (idx, _st), buf in _sk_ws.items() if idx == device.index]
    g = _guard_words.get(device.index)
    if g is not None:
        # the guard word: with several ranks it carries EVERY rank's error (reduced with the buckets) and is the ONLY word looked
        # at -- this rank's own error words are ahead of it by up to a step, and a rank that raised on them alone would leave its
        # peers waiting in a collective
        bufs = [g] if wait else bufs + [g]
    elif wait:
        bufs = []                 # (before the first step of a several-rank run: nothing symmetric to look at yet)
    st = _sk_poll.get(device.index)
    if st is not None and wait:
        st[1].synchronize()       # several ranks: all of them must look at the SAME step's word, so that they raise together
    if st is not None and st[1].query():
        bad = bool(st[0][:st[2]].any())
        _sk_poll.pop(device.index)
        st = None
        if bad:
            nt_sk_reset(device)
            raise RuntimeError("ofq_amd: a stream-K hand-off timed out in an earlier step, on this rank or on another (a workgroup "
                               "of ofq_qgemm_bf16s_nt_sk waited ~0.6 s for a partial tile): the gradients since then are invalid "
                               "and the optimiser has skipped those steps on every rank; the flag areas have been re-zeroed -- "
                               "restore the last checkpoint")
    if st is None and bufs:
        host = torch.empty(len(bufs), dtype=torch.int32).pin_memory()
        for i, buf in enumerate(bufs):
            host[i:i + 1].copy_(buf, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        _sk_poll[device.index] = (host, ev, len(bufs))


def nt_sk_reset(device):
    """Re-zero flags, error word and test words of every stream-K workspace of the device (after nt_sk_error() != 0)."""
    _sk_poll.pop(device.index, None)
    for (idx, _st), buf in _sk_ws.items():
        if idx == device.index:
            _chk(lib().ofq_qgemm_bf16s_nt_sk_reset(buf.data_ptr(), _stream()), "ofq_qgemm_bf16s_nt_sk_reset")
    g = _guard_words.get(device.index)
    if g is not None:
        g.zero_()


def nt_sk_inject_fault(device, wg):
    """Test hook: workgroup `wg` of the next stream-K launches on the current stream never publishes its partial (wg < 0: off)."""
    buf = _sk_workspace(device)
    buf[:32768].view(torch.int32)[4097] = wg + 1 if wg >= 0 else 0


def qgemm_bf16s_nt_sk(segs, out, accumulate=False, wgs=None, col_bias=None, hi_only_last=False):
    """out[m,n] (+)= sum over segs of alpha * sum_k (A[m,k]*k_scale[k]) * B[n,k]; segs = [(A, B_bf16, k_scale, alpha), ...] (1 or 2)
    hi_only_last (two fp16-code segments): the second segment's B meets A's leading plane only (ofq_nt_seg.hi_only)."""
    M, N = out.shape
    dev = out.device
    arr = (_lib.NtSeg * len(segs))()
    K = 0
    f16 = segs[0][1].dtype == torch.float16           # fp16 codes: the two-plane form, every segment with its maximum word
    keep = []
    for i, (A, B, ks, alpha) in enumerate(segs):
        if (B.dtype == torch.float16) != f16:
            raise RuntimeError("ofq_amd: qgemm_bf16s_nt_sk: the segments' codes must share one format (fp16 or bf16)")
        arr[i].A, arr[i].B_bf16, arr[i].k_scale = A.data_ptr(), B.data_ptr(), _p(ks)
        arr[i].K, arr[i].lda, arr[i].ldb, arr[i].alpha = A.shape[1], A.stride(0), B.stride(0), alpha
        if f16:
            w = amax_for(A)
            keep.append(w)
            arr[i].amax = w.data_ptr()
        _probe("dx", A, ks, 1)
        K += A.shape[1]
    if hi_only_last:
        if len(segs) != 2 or not f16 or segs[0][0].shape[1] % 64:
            raise RuntimeError("ofq_amd: qgemm_bf16s_nt_sk: hi_only_last needs two fp16-code segments, the first with K % 64 == 0")
        arr[1].hi_only = 1
    ws = _sk_workspace(dev)
    g = num_cus(dev) if wgs is None else -int(wgs)        # wgs: exactly that many workgroups (<= the CU count; tests)
    with _Timed(nt_class(f16), 2.0 * M * N * K):
        _chk(lib().ofq_qgemm_bf16s_nt_sk(arr, len(segs), out.data_ptr(), int(accumulate), M, N, out.stride(0), g, ws.data_ptr(),
                                         ws.numel(), _p(col_bias), _stream()), "ofq_qgemm_bf16s_nt_sk")
    return out


NT_CONCAT = os.environ.get("OFQ_NO_NT_CONCAT") is None     # A/B switch: "v + W_qk input gradients as one GEMM"


def nt_concat_ok(A, B_bf16, parked):
    """May the input-gradient GEMM of (A, B) run as a further K-segment of the parked ones?  (same M and N, whole pairs of
    32-wide k-steps, two segments at most, 32-bit row offsets)"""
    if not (NT_SK and NT_CONCAT):
        return False
    M, K = A.shape
    N = B_bf16.shape[0]
    if N <= 128 or K % 32 or M * A.stride(0) * 4 >= 2 ** 32 or N * B_bf16.stride(0) * 2 >= 2 ** 32:
        return False
    ktot = K
    for (a, b, _, _) in (parked or ()):
        if a.shape[0] != M or b.shape[0] != N or b.dtype != B_bf16.dtype:
            return False
        ktot += a.shape[1]
    return len(parked or ()) <= 1 and (not parked or ktot % 64 == 0)


def nt_sk_pays(M, N, K, device):
    if not NT_SK:
        return False
    if _NT_SK_FORCE:
        return N > 128 and K % 64 == 0
    return bool(lib().ofq_qgemm_bf16s_nt_sk_pays(M, N, K, num_cus(device)))


def qgemm_bf16s_nt(A, B_bf16, k_scale, alpha, out=None, accumulate=False, nsplit=3, sk=None, col_scale=None, col_bias=None, amax=None):
    """out[m,n] (+)= alpha * sum_k (A[m,k]*k_scale[k]) * B[n,k],  A fp32, B bf16 integer codes
    sk: None = stream-K where it pays (ofq_qgemm_bf16s_nt_sk_pays), False / True = never / always"""
    M, K = A.shape
    N = B_bf16.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    # (the stream-K launch shares PAIRS of 32-wide k-steps and addresses rows with 32-bit offsets: other shapes take the
    # one-tile-per-workgroup kernel also when sk=True)
    sk_able = N > 128 and K % 64 == 0 and M * A.stride(0) * 4 < 2 ** 32 and N * B_bf16.stride(0) * 2 < 2 ** 32
    f16 = B_bf16.dtype == torch.float16                # fp16 codes: the two-plane form (ops.GRAD_PLANES == 2 hands these out)
    if col_scale is not None or col_bias is not None:
        sk = False                                   # (the per-column epilogue lives in the one-tile-per-workgroup kernel)
    if (nsplit == 3 or f16) and sk_able and (sk if sk is not None else nt_sk_pays(M, N, K, A.device)):
        return qgemm_bf16s_nt_sk([(A, B_bf16, k_scale, alpha)], out, accumulate)
    if f16 and amax is None:
        amax = amax_for(A)
    if not f16:
        amax = None
    _probe("dx", A, k_scale, 1)
    with _Timed(nt_class(f16), 2.0 * M * N * K):
        _chk(lib().ofq_qgemm_bf16s_nt(A.data_ptr(), B_bf16.data_ptr(), out.data_ptr(), _p(k_scale), alpha, int(accumulate),
                                      2 if f16 else nsplit, M, N, K, A.stride(0), B_bf16.stride(0), out.stride(0), _p(amax),
                                      _p(col_scale), _p(col_bias), _stream()),
             "ofq_qgemm_bf16s_nt")
    return out


def qgemm_bf16s_nt_lsq(dy2d, B_bf16, k_scale, alpha, x2d, s, b4, g, want_bias_grads=True):
    """dX GEMM + LSQ backward of the layer's input quantiser in one kernel; returns (dx, ds, db4, dbaft) like lsq_bwd.
    g: the quantiser's LsqGeom (per-token step, one offset phase)."""
    M, K = dy2d.shape
    N = B_bf16.shape[0]
    dev = dy2d.device
    dx = torch.empty((M, N), dtype=torch.float32, device=dev)
    ds = torch.empty_like(s)
    has_bias = b4 is not None and want_bias_grads
    db4 = torch.empty(N, dtype=torch.float32, device=dev) if has_bias else None
    dbaft = torch.empty(N, dtype=torch.float32, device=dev) if has_bias else None
    ws = workspace(lib().ofq_qgemm_bf16s_nt_lsq_ws_bytes(M, N), dev)
    amax = amax_for(dy2d) if B_bf16.dtype == torch.float16 else None         # fp16 codes: the two-plane form
    with _Timed('qgemm_bf16s_nt_wide_lsq (linear dX + LSQ backward epilogue)', 2.0 * M * N * K):
        _chk(lib().ofq_qgemm_bf16s_nt_lsq(dy2d.data_ptr(), B_bf16.data_ptr(), _p(k_scale), alpha, x2d.data_ptr(), s.data_ptr(),
                                          g.S, g.gscale, _p(b4), g.lo, g.hi, int(g.prologue == 1), dx.data_ptr(), ds.data_ptr(),
                                          _p(db4), _p(dbaft), M, N, K, dy2d.stride(0), B_bf16.stride(0), x2d.stride(0),
                                          ws.data_ptr(), ws.numel(), _p(amax), _stream()), "ofq_qgemm_bf16s_nt_lsq")
    return dx, ds, db4, dbaft


def _planes_amax(t2d, planes):
    """The maximum word of `t2d` when the two-plane form is asked for (planes None: ops.GRAD_PLANES), else None."""
    return amax_for(t2d) if ((GRAD_PLANES if planes is None else planes) == 2 and "dw" in _DBG_F16) else None


def qgemm_bf16s_tn(dy2d, xcodes2d, lsq_s, S, gscale, db, baft, split=None, compute_db=False, out=None, planes=None):
    """dW[o,c] = sum_m (dy[m,o]*a_eff[m % S]) * codes[m,c] + db[o]*baft[c];  compute_db: also returns db = colsum(dy).
    out: write dW there (a contiguous (o, c) fp32 tensor, e.g. the weight's slice of a gradient bucket).
    planes: 2 / 3 planes of dy (None: ops.GRAD_PLANES; the wide kernels only, the narrow one always takes three)."""
    Ktok, M = dy2d.shape
    N = xcodes2d.shape[1]
    if split is None:
        if N > 128 and N % 8 == 0:       # wide 128x384 (128x256) tiles, one 512-thread workgroup per CU
            tiles = ((M + 127) // 128) * (N // 384 if N % 384 == 0 else (N + 255) // 256)
            split = max(1, min(256 // tiles, (Ktok + 31) // 32 // 4))
        else:
            tiles = ((M + 127) // 128) * ((N + 127) // 128)
            split = max(1, min(512 // tiles, (Ktok + 31) // 32 // 4))
    if out is not None and (tuple(out.shape) != (M, N) or not out.is_contiguous() or out.dtype != torch.float32
                            or out.device != dy2d.device):
        raise ValueError("qgemm_bf16s_tn: out must be a contiguous fp32 (%d, %d) tensor on the operands' device" % (M, N))
    dW = out if out is not None else torch.empty((M, N), dtype=torch.float32, device=dy2d.device)
    if compute_db:
        db = torch.empty(M, dtype=torch.float32, device=dy2d.device)
    ws = workspace(lib().ofq_qgemm_bf16s_tn_ws_bytes(M, N, split), dy2d.device)
    amax = _planes_amax(dy2d, planes)
    _probe("dw", dy2d, lsq_eff_scale(lsq_s, gscale) if PLANE_PROBE is not None else None, 0)
    with _Timed('qgemm_bf16s_tn_wide (linear dW, %s)' % _pl(amax is not None), 2.0 * Ktok * M * N):
        _chk(lib().ofq_qgemm_bf16s_tn(dy2d.data_ptr(), xcodes2d.data_ptr(), dW.data_ptr(), lsq_s.data_ptr(), S, gscale,
                                      _p(db), int(compute_db), _p(baft), Ktok, M, N, dy2d.stride(0), xcodes2d.stride(0),
                                      split, ws.data_ptr(), ws.numel(), _p(amax), _stream()), "ofq_qgemm_bf16s_tn")
    return (dW, db) if compute_db else dW


TN_GROUP_MAX = 8
# Which dW problems join a grouped launch.  The entry takes any N > 128 and any S >= 1 (round 6); measured on one box
# (profiles/r06_dw_routing_ab.txt): grouping the N = 192 layers of DeiT-T / Swin's stage 2 changes nothing (17.27 -> 17.27 ms,
# 42.07 -> 42.16), so the rule for N stays round 5's; S < 32 (Swin's 4-D MLP quantisers) joins.  Environment: A/B hooks.
TN_GROUP_MIN_N = int(os.environ.get("OFQ_TN_GROUP_MIN_N", "256"))
TN_GROUP_MIN_S = int(os.environ.get("OFQ_TN_GROUP_MIN_S", "1"))


def tn_tiles(M, N):
    """Workgroup tiles of one wide dW problem (128 x 384, or 128 x 256 when N is not a multiple of 384)."""
    return ((M + 127) // 128) * (N // 384 if N % 384 == 0 else (N + 255) // 256)


def tn_groupable(Ktok, M, N, S, lda, ldb):
    """May this dW problem join a grouped launch (ofq_qgemm_bf16s_tn_group)?"""
    return (N >= TN_GROUP_MIN_N and N % 16 == 0 and M % 4 == 0 and S >= TN_GROUP_MIN_S and lda % 4 == 0 and ldb % 16 == 0
            and Ktok * lda < (1 << 31) and Ktok * ldb < (1 << 31))


def qgemm_bf16s_tn_group(jobs, split=None, planes=None):
    """jobs: list of dicts (dy2d, xcodes2d, lsq_s, S, gscale, baft, dW, db[, amax]) -- dW (M, N) / db (M) are OUTPUT tensors the
    caller has allocated; every job computes what qgemm_bf16s_tn(compute_db=db is not None) would.  One GEMM launch and
    one reduce launch for all of them.  planes: as qgemm_bf16s_tn (a job's "amax" word, if present, is used as it is)."""
    n = len(jobs)
    assert 1 <= n <= TN_GROUP_MAX
    arr = (_lib.TnJob * n)()
    keep = []
    tiles = 0
    flops = 0.0
    nkt = 1
    for i, j in enumerate(jobs):
        dy, xc = j["dy2d"], j["xcodes2d"]
        Ktok, M = dy.shape
        N = xc.shape[1]
        tiles += tn_tiles(M, N)
        flops += 2.0 * Ktok * M * N
        nkt = max(nkt, (Ktok + 31) // 32)
        a = arr[i]
        # dW / db may be given as raw device addresses: a caller that has handed the tensors to autograd must not keep a
        # second reference (AccumulateGrad adopts a gradient only when it holds the last one, and clones it otherwise)
        pw, pb = j["dW"], j["db"]
        a.dY, a.codes, a.lsq_s = dy.data_ptr(), xc.data_ptr(), j["lsq_s"].data_ptr()
        a.dW = pw if isinstance(pw, int) else pw.data_ptr()
        a.db = pb if isinstance(pb, int) else _p(pb)
        a.baft = _p(j["baft"])
        a.S, a.Ktok, a.M, a.N, a.lda, a.ldb = j["S"], Ktok, M, N, dy.stride(0), xc.stride(0)
        a.gscale, a.compute_db = j["gscale"], int(pb is not None)
        w = j.get("amax")
        if w is None:
            w = _planes_amax(dy, planes)
        keep.append(w)
        a.amax = _p(w)
        _probe("dw", dy, lsq_eff_scale(j["lsq_s"], j["gscale"]) if PLANE_PROBE is not None else None, 0)
    if split is None:
        split = max(1, min(256 // tiles, nkt // 4))
    dev = jobs[0]["dy2d"].device
    ws = workspace(lib().ofq_qgemm_bf16s_tn_group_ws_bytes(arr, n, split), dev)
    with _Timed('qgemm_bf16s_tn_wide_group (linear dW of a block, %s)' % _pl(keep[0] is not None), flops):
        _chk(lib().ofq_qgemm_bf16s_tn_group(arr, n, split, ws.data_ptr(), ws.numel(), _stream()), "ofq_qgemm_bf16s_tn_group")


# ------------------------------------------------------------------------------------------------ attention on codes
def lsq_eff_scale(s, gscale):
    """(a - a*g) + a*g with a = max(s, 1e-5): the scale VALUE the LSQ kernels divide by (fp32, lsq.py:6-18)."""
    a = torch.where(s > 1e-5, s, torch.full_like(s, 1e-5))
    t = a * gscale
    return (a - t) + t


def lsq_eff_scale_vec(s, gscale, repeat=1):
    """lsq_eff_scale as ONE launch, each value `repeat` times in a row (ofq_lsq_eff_scale_vec)."""
    out = torch.empty(s.numel() * repeat, dtype=torch.float32, device=s.device)
    _chk(lib().ofq_lsq_eff_scale_vec(s.data_ptr(), float(gscale), out.data_ptr(), s.numel(), int(repeat), _stream()), "ofq_lsq_eff_scale_vec")
    return out


def rowdot_i8_multi(codes2d, vecs2d):
    R, K = codes2d.shape
    V = vecs2d.shape[0]
    out = torch.empty((R, V), dtype=torch.float32, device=codes2d.device)
    _chk(lib().ofq_rowdot_i8_multi(codes2d.data_ptr(), vecs2d.data_ptr(), out.data_ptr(), R, K, V, _stream()),
         "ofq_rowdot_i8_multi")
    return out


def rowdot_f32_seg(x2d, vec, H, d):
    R = x2d.shape[0]
    out = torch.empty((R, H), dtype=torch.float32, device=x2d.device)
    _chk(lib().ofq_rowdot_f32_seg(x2d.data_ptr(), vec.data_ptr(), out.data_ptr(), R, H, d, x2d.stride(0), _stream()),
         "ofq_rowdot_f32_seg")
    return out


def codes_transpose_i8(codes3d, rows_padded):
    Bn, R, Cc = codes3d.shape
    out = torch.empty((Bn, Cc, rows_padded), dtype=torch.int8, device=codes3d.device)
    _chk(lib().ofq_codes_transpose_i8(codes3d.data_ptr(), out.data_ptr(), Bn, R, Cc, rows_padded, _stream()),
         "ofq_codes_transpose_i8")
    return out


def qattn_prep(xcodes, baq2, qcodes, bax, vcodes, B, H, N, C, Np):
    """u, tq and the transposed v codes of the QKR attention core in one launch (same values as rowdot_i8_multi /
    rowdot_i8 / codes_transpose_i8)."""
    dev = xcodes.device
    u = torch.empty((B * N, H), dtype=torch.float32, device=dev)
    tq = torch.empty(B * N * H, dtype=torch.float32, device=dev)
    vT = torch.empty((B, C, Np), dtype=torch.int8, device=dev)
    z = torch.empty(H, dtype=torch.float32, device=dev)              # z[h] = baq[h] . bax (was a torch.mv per block)
    _chk(lib().ofq_qattn_prep(xcodes.data_ptr(), baq2.data_ptr(), u.data_ptr(), qcodes.data_ptr(), bax.data_ptr(), tq.data_ptr(),
                              vcodes.data_ptr(), vT.data_ptr(), z.data_ptr(), B, H, N, C, Np, _stream()), "ofq_qattn_prep")
    return u, tq, vT, z


def qattn_scores(xcodes, qcodes, sx, gx, sq, gq, u, tq, z, B, H, N, C, ldS):
    S = torch.empty((B, H, N, ldS), dtype=torch.float32, device=xcodes.device)
    with _Timed('qgemm_i8_nt (v_mfma_i32_32x32x32_i8)', 2.0 * B * H * N * N * C):
        _chk(lib().ofq_qattn_scores_i8(xcodes.data_ptr(), qcodes.data_ptr(), S.data_ptr(), sx.data_ptr(), gx, sq.data_ptr(),
                                       gq, u.data_ptr(), tq.data_ptr(), z.data_ptr(), B, H, N, C, ldS, _stream()),
             "ofq_qattn_scores_i8")
    return S


def qattn_scores_plain(qcodes, kcodes, sq, gq, sk, gk, u, tq, z, B, H, N, d, ldS):
    S = torch.empty((B, H, N, ldS), dtype=torch.float32, device=qcodes.device)
    with _Timed('qgemm_i8_nt (v_mfma_i32_32x32x32_i8)', 2.0 * B * H * N * N * d):
        _chk(lib().ofq_qattn_scores_plain_i8(qcodes.data_ptr(), kcodes.data_ptr(), S.data_ptr(), sq.data_ptr(), gq, sk.data_ptr(),
                                             gk, u.data_ptr(), tq.data_ptr(), z.data_ptr(), B, H, N, d, ldS, _stream()),
             "ofq_qattn_scores_plain_i8")
    return S


def qattn_scores_softmax(acodes, bcodes, sa, ga, sb, gb, u, tq, z, plain, sm_s, alpha, hi, B, H, N, CK, ld, addend=None):
    """Scores GEMM + softmax + LSQ of the probabilities in one kernel; returns (prob fp32, codes uint8, code row sums)."""
    dev = acodes.device
    prob = torch.empty((B, H, N, ld), dtype=torch.float32, device=dev)
    codes = torch.empty((B, H, N, ld), dtype=torch.uint8, device=dev)
    rsum = torch.empty(B * H * N, dtype=torch.float32, device=dev)
    gscale = 1.0 / math.sqrt(hi * (B * H * N))
    with _Timed('qattn_scores_softmax (int8 scores + softmax + LSQ)', 2.0 * B * H * N * N * CK):
        _chk(lib().ofq_qattn_scores_softmax_i8(acodes.data_ptr(), bcodes.data_ptr(), sa.data_ptr(), ga, sb.data_ptr(), gb,
                                               u.data_ptr(), tq.data_ptr(), z.data_ptr(), int(plain), sm_s.data_ptr(), gscale,
                                               alpha, int(hi), _p(addend), addend.shape[0] if addend is not None else 1,
                                               prob.data_ptr(), codes.data_ptr(), rsum.data_ptr(), B, H, N, CK, ld, _stream()),
             "ofq_qattn_scores_softmax_i8")
    return prob, codes, rsum


def qattn_dq_plain(dS, kcodes, sk, gk, B, H, N, d, ldS):
    dq = torch.empty((B, N, H * d), dtype=torch.float32, device=dS.device)
    with _Timed('qgemm_bf16s_nn (3x v_mfma_f32_32x32x16_bf16)', 2.0 * B * H * N * N * d):
        _chk(lib().ofq_qattn_dq_plain_bf16s(dS.data_ptr(), kcodes.data_ptr(), dq.data_ptr(), sk.data_ptr(), gk, B, H, N, d, ldS,
                                            _stream()), "ofq_qattn_dq_plain_bf16s")
    return dq


def qattn_dk_plain(dS, qcodes, sq, gq, bq, B, H, N, d, ldS):
    dk = torch.empty((B, N, H * d), dtype=torch.float32, device=dS.device)
    with _Timed('qgemm_bf16s_tn (3x v_mfma_f32_32x32x16_bf16)', 2.0 * B * H * N * N * d):
        _chk(lib().ofq_qattn_dk_plain_bf16s(dS.data_ptr(), qcodes.data_ptr(), dk.data_ptr(), sq.data_ptr(), gq, _p(bq), B, H, N,
                                            d, ldS, _stream()), "ofq_qattn_dk_plain_bf16s")
    return dk


def qattn_pv(pcodes, vcodesT, sp, gp, sv, gv, bav, rp, B, H, N, d, Np):
    O = torch.empty((B, N, H * d), dtype=torch.float32, device=pcodes.device)
    with _Timed('qgemm_i8_nt (v_mfma_i32_32x32x32_i8)', 2.0 * B * H * N * N * d):
        _chk(lib().ofq_qattn_pv_i8(pcodes.data_ptr(), vcodesT.data_ptr(), O.data_ptr(), sp.data_ptr(), gp, sv.data_ptr(), gv,
                                   _p(bav), rp.data_ptr(), B, H, N, d, Np, _stream()), "ofq_qattn_pv_i8")
    return O


def qattn_dp(dO, vcodes, sv, gv, w, B, H, N, d, ldP):
    """dP[b,h,n,m] = sum_j dO[b,n,hd+j] * (av_eff[hd+j] * qv[b,m,hd+j]) + w[b,n,h]; av_eff = effective value of the step sv
    (taken inside the kernel when gv > 0; gv = 0: sv is used as is)"""
    dP = torch.empty((B, H, N, ldP), dtype=torch.float32, device=dO.device)
    with _Timed('qgemm_bf16s_nt (attention dP, 3x v_mfma_f32_32x32x16_bf16)', 2.0 * B * H * N * N * d):
        _chk(lib().ofq_qattn_dp_bf16s(dO.data_ptr(), vcodes.data_ptr(), dP.data_ptr(), sv.data_ptr(), float(gv), _p(w), B, H, N,
                                      d, ldP, _stream()), "ofq_qattn_dp_bf16s")
    return dP


def qattn_dp_softmax_bwd(dO, vcodes, sv, gv, bav, prob, sm_s, alpha, hi, B, H, N, d, ld, want_rowsum=False):
    """dS, ds (softmax quantiser step gradient), [row sums of dS] from dO: the dP GEMM and ofq_softmax_lsq_bwd in one kernel
    (same values as qattn_dp + softmax_lsq_bwd up to the summation order inside the K = d products)."""
    dS = torch.empty((B, H, N, ld), dtype=torch.float32, device=dO.device)
    ds = torch.empty_like(sm_s)
    rs = torch.empty(B * H * N, dtype=torch.float32, device=dO.device) if want_rowsum else None
    gscale = 1.0 / math.sqrt(hi * (B * H * N))
    ws = workspace(lib().ofq_qattn_dp_softmax_bwd_ws_bytes(B, H, N), dO.device)
    am = amax_out(dO.device)
    with _Timed('qattn_dp_softmax_bwd (dP GEMM + softmax-LSQ backward)', 2.0 * B * H * N * N * d):
        _chk(lib().ofq_qattn_dp_softmax_bwd(dO.data_ptr(), vcodes.data_ptr(), sv.data_ptr(), float(gv), _p(bav), prob.data_ptr(),
                                            sm_s.data_ptr(), gscale, float(alpha), int(hi), dS.data_ptr(), ds.data_ptr(), _p(rs),
                                            B, H, N, d, ld, ws.data_ptr(), ws.numel(), _p(am), _stream()), "ofq_qattn_dp_softmax_bwd")
    if am is not None:
        tag_amax(dS, am)
    return dS, ds, rs


def qattn_dv(dO, pcodes, sp, gp, B, H, N, d, Np):
    dV = torch.empty((B, N, H * d), dtype=torch.float32, device=dO.device)
    with _Timed('qgemm_bf16s_tn (attention dV, 3x v_mfma_f32_32x32x16_bf16)', 2.0 * B * H * N * N * d):
        _chk(lib().ofq_qattn_dv_bf16s(dO.data_ptr(), pcodes.data_ptr(), dV.data_ptr(), sp.data_ptr(), gp, B, H, N, d, Np,
                                      _stream()), "ofq_qattn_dv_bf16s")
    return dV


def scores_amax(dS, N, planes=None):
    """The maximum word of a score-gradient tensor (B, H, N, ld) over its N REAL columns (the pad columns up to ld are never
    read and may hold anything): the producer's word when it attached one, else a reduction over the strided view."""
    if (GRAD_PLANES if planes is None else planes) != 2:
        return None
    w = amax_of(dS)
    if w is None:
        w = absmax(dS.view(-1, dS.shape[-1])[:, :N])
        tag_amax(dS, w)
    return w


def qattn_dqkx(dS, xcodes, sx, gx, bax, B, H, N, C, ldS, planes=None):
    dq = torch.empty((B, N, H, C), dtype=torch.float32, device=dS.device)
    amax = scores_amax(dS, N, planes) if "dqkx" in _DBG_F16 else None
    _probe("dqkx", dS.view(-1, dS.shape[-1]), None, 1, N)
    with _Timed('qgemm_bf16s_tn_wide_stream (attention dqkx, %s)' % _pl(amax is not None), 2.0 * B * H * N * N * C):
        _chk(lib().ofq_qattn_dqkx_bf16s(dS.data_ptr(), xcodes.data_ptr(), dq.data_ptr(), sx.data_ptr(), gx, _p(bax), B, H, N, C,
                                        ldS, _p(amax), _stream()), "ofq_qattn_dqkx_bf16s")
    return dq


def qattn_dxq(dS, qcodes, sq, gq, B, H, N, C, ldS, out=None, accumulate=False, planes=None):
    if out is None:
        out = torch.empty((B, N, C), dtype=torch.float32, device=dS.device)
    amax = scores_amax(dS, N, planes) if "dxq" in _DBG_F16 else None
    _probe("dxq", dS.view(-1, dS.shape[-1]), None, 1, N)
    with _Timed('qgemm_bf16s_nn_wide (attention dxq, %s)' % _pl(amax is not None), 2.0 * B * H * N * N * C):
        _chk(lib().ofq_qattn_dxq_bf16s(dS.data_ptr(), qcodes.data_ptr(), out.data_ptr(), sq.data_ptr(), gq, int(accumulate), B,
                                       H, N, C, ldS, _p(amax), _stream()), "ofq_qattn_dxq_bf16s")
    return out


def kd_loss_fwd(cls_logits, dist_logits, teacher_logits, target):
    """(loss, dcls, ddist) of KDLossSoftandHard in two launches (ofq_kd_loss_fwd); logits [B][K] fp32 with unit inner stride."""
    B, K = cls_logits.shape
    dev = cls_logits.device
    loss = torch.empty((), dtype=torch.float32, device=dev)
    dcls = torch.empty((B, K), dtype=torch.float32, device=dev)
    ddist = torch.empty((B, K), dtype=torch.float32, device=dev)
    rows = torch.empty(2 * B + 1, dtype=torch.float32, device=dev)
    _chk(lib().ofq_kd_loss_fwd(cls_logits.data_ptr(), dist_logits.data_ptr(), teacher_logits.data_ptr(), target.data_ptr(),
                               loss.data_ptr(), dcls.data_ptr(), ddist.data_ptr(), rows.data_ptr(), B, K, cls_logits.stride(0),
                               dist_logits.stride(0), teacher_logits.stride(0), _stream()), "ofq_kd_loss_fwd")
    return loss, dcls, ddist, rows[2 * B:]          # (the last one: B / number of rows whose label is not ignore_index)


def kd_loss_bwd(g, dcls, ddist, cls_scale=None):
    oc, od = torch.empty_like(dcls), torch.empty_like(ddist)
    _chk(lib().ofq_kd_loss_bwd(g.data_ptr(), dcls.data_ptr(), ddist.data_ptr(), 0 if cls_scale is None else cls_scale.data_ptr(),
                               oc.data_ptr(), od.data_ptr(), dcls.numel(), _stream()), "ofq_kd_loss_bwd")
    return oc, od


def assemble_tokens(patches, cls_token, dist_token, pos):
    """cat(cls, [dist,] patches) + pos in one pass; patches (B, P, C), tokens (1, 1, C), pos (1, P + ntok, C)"""
    B, P, C = patches.shape
    T = pos.shape[1]
    out = torch.empty((B, T, C), dtype=torch.float32, device=patches.device)
    _chk(lib().ofq_assemble_tokens(patches.data_ptr(), cls_token.data_ptr(), _p(dist_token), pos.data_ptr(), out.data_ptr(), B, T, C,
                                   _stream()), "ofq_assemble_tokens")
    return out


def colsum(x2d):
    rows, cols = x2d.shape
    out = torch.empty(cols, dtype=torch.float32, device=x2d.device)
    ws = workspace(lib().ofq_colsum_ws_bytes(rows, cols), x2d.device)
    _chk(lib().ofq_colsum(x2d.data_ptr(), out.data_ptr(), rows, cols, x2d.stride(0), ws.data_ptr(), ws.numel(),
                          _stream()), "ofq_colsum")
    return out


# ------------------------------------------------------------------------------------------------ LayerNorm
def layernorm_fwd(x2d, gamma, beta, eps, res2d=None, want_amax=False):
    """y = LN(x [+ res]); returns (y, xsum or None, mean, rstd); want_amax: y carries its maximum word (ops.amax_of)"""
    _dev(x2d, "x")
    rows, cols = x2d.shape
    dev = x2d.device
    y = torch.empty((rows, cols), dtype=torch.float32, device=dev)
    xsum = torch.empty((rows, cols), dtype=torch.float32, device=dev) if res2d is not None else None
    mean = torch.empty(rows, dtype=torch.float32, device=dev)
    rstd = torch.empty(rows, dtype=torch.float32, device=dev)
    am = amax_word(dev) if want_amax else None
    _chk(lib().ofq_layernorm_fwd(x2d.data_ptr(), _p(res2d), _p(gamma), _p(beta), y.data_ptr(), _p(xsum), mean.data_ptr(),
                                 rstd.data_ptr(), rows, cols, x2d.stride(0), cols, float(eps), _p(am), _stream()),
         "ofq_layernorm_fwd")
    if am is not None:
        tag_amax(y, am)
    return y, xsum, mean, rstd


def layernorm_bwd(dy2d, x2d, mean, rstd, gamma, dres2d=None, want_affine_grads=True):
    """returns (dx [+ dres], dgamma, dbeta)"""
    rows, cols = x2d.shape
    dev = x2d.device
    dx = torch.empty((rows, cols), dtype=torch.float32, device=dev)
    dg = torch.empty(cols, dtype=torch.float32, device=dev) if want_affine_grads else None
    db = torch.empty(cols, dtype=torch.float32, device=dev) if want_affine_grads else None
    ws = workspace(lib().ofq_layernorm_bwd_ws_bytes(rows, cols), dev)
    am = amax_out(dev)
    _chk(lib().ofq_layernorm_bwd(dy2d.data_ptr(), x2d.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _p(gamma), _p(dres2d),
                                 dx.data_ptr(), _p(dg), _p(db), rows, cols, x2d.stride(0), dy2d.stride(0), ws.data_ptr(),
                                 ws.numel(), _p(am), _stream()), "ofq_layernorm_bwd")
    if am is not None:
        tag_amax(dx, am)
    return dx, dg, db


def layernorm_lsq_fwd(x2d, gamma, beta, eps, s, b4, g, res2d=None, q_perm=None, res_perm=None):
    """codes = LSQ(LN(x [+ res]) + b4) without materialising the LayerNorm output; returns (codes, xsum|None, mean, rstd).
    q_perm / res_perm (int32, one image's tokens): ofq_layernorm_lsq_fwd_perm -- token t's codes go to row q_perm[t] of its image
    (the quantiser's step is indexed there), token t's `res` row is read from row res_perm[t]."""
    _dev(x2d, "x")
    rows, cols = x2d.shape
    dev = x2d.device
    xsum = torch.empty((rows, cols), dtype=torch.float32, device=dev) if res2d is not None else None
    mean = torch.empty(rows, dtype=torch.float32, device=dev)
    rstd = torch.empty(rows, dtype=torch.float32, device=dev)
    codes = torch.empty((rows, cols), dtype=torch.int8, device=dev)
    if q_perm is not None or res_perm is not None:
        n = (q_perm if q_perm is not None else res_perm).numel()
        _chk(lib().ofq_layernorm_lsq_fwd_perm(x2d.data_ptr(), _p(res2d), _p(gamma), _p(beta), 0, _p(xsum), mean.data_ptr(),
                                              rstd.data_ptr(), codes.data_ptr(), s.data_ptr(), g.S, g.gscale, _p(b4), g.lo, g.hi,
                                              rows, cols, x2d.stride(0), float(eps), _p(q_perm), _p(res_perm), n, _stream()),
             "ofq_layernorm_lsq_fwd_perm")
        return codes, xsum, mean, rstd
    _chk(lib().ofq_layernorm_lsq_fwd(x2d.data_ptr(), _p(res2d), _p(gamma), _p(beta), 0, _p(xsum), mean.data_ptr(),
                                     rstd.data_ptr(), codes.data_ptr(), s.data_ptr(), g.S, g.gscale, _p(b4), g.lo, g.hi, rows,
                                     cols, x2d.stride(0), float(eps), _stream()), "ofq_layernorm_lsq_fwd")
    return codes, xsum, mean, rstd


def layernorm_lsq_bwd(gq2d, x2d, mean, rstd, gamma, beta, s, b4, g, dres2d=None, q_perm=None, res_perm=None):
    """returns (dx [+ dres], dgamma, dbeta, db4 (same values as dbeta, its own tensor; None without b4), ds, dbaft)
    q_perm: gq2d's rows (and the step index) are in the permuted order of layernorm_lsq_fwd(q_perm=...); res_perm: a seventh
    result, dx's rows once more in the order of the forward's permuted `res` operand (its gradient)."""
    rows, cols = x2d.shape
    dev = x2d.device
    dx = torch.empty((rows, cols), dtype=torch.float32, device=dev)
    dg = torch.empty(cols, dtype=torch.float32, device=dev) if gamma is not None else None
    db = torch.empty(cols, dtype=torch.float32, device=dev)
    db4 = torch.empty(cols, dtype=torch.float32, device=dev) if b4 is not None else None
    dba = torch.empty(cols, dtype=torch.float32, device=dev)
    ds = torch.empty_like(s)
    ws = workspace(lib().ofq_layernorm_lsq_bwd_ws_bytes(rows, cols), dev)
    am = amax_out(dev)
    if q_perm is not None or res_perm is not None:
        n = (q_perm if q_perm is not None else res_perm).numel()
        dx2 = torch.empty((rows, cols), dtype=torch.float32, device=dev) if res_perm is not None else None
        _chk(lib().ofq_layernorm_lsq_bwd_perm(gq2d.data_ptr(), x2d.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _p(gamma),
                                              _p(beta), _p(dres2d), s.data_ptr(), g.S, g.gscale, _p(b4), g.lo, g.hi, dx.data_ptr(),
                                              _p(dg), db.data_ptr(), _p(db4), ds.data_ptr(), dba.data_ptr(), rows, cols,
                                              x2d.stride(0), gq2d.stride(0), ws.data_ptr(), ws.numel(), _p(am), _p(q_perm),
                                              _p(res_perm), n, _p(dx2), _stream()), "ofq_layernorm_lsq_bwd_perm")
        if am is not None:
            tag_amax(dx, am)
            if dx2 is not None:
                tag_amax(dx2, am)          # (the same rows in another order: the same maximum)
        return dx, dg, db, db4, ds, dba, dx2
    _chk(lib().ofq_layernorm_lsq_bwd(gq2d.data_ptr(), x2d.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _p(gamma), _p(beta),
                                     _p(dres2d), s.data_ptr(), g.S, g.gscale, _p(b4), g.lo, g.hi, dx.data_ptr(), _p(dg),
                                     db.data_ptr(), _p(db4), ds.data_ptr(), dba.data_ptr(), rows, cols, x2d.stride(0),
                                     gq2d.stride(0), ws.data_ptr(), ws.numel(), _p(am), _stream()), "ofq_layernorm_lsq_bwd")
    if am is not None:
        tag_amax(dx, am)
    return dx, dg, db, db4, ds, dba


# ------------------------------------------------------------------------------------------------ CGA
def cga_freeze_mask(W, bits, boundary_range):
    _dev(W, "weight")
    W = W.contiguous()
    rows, cols = W.shape
    frozen = torch.empty_like(W)
    rng = torch.empty(2, dtype=torch.int32, device=W.device)
    _chk(lib().ofq_cga_freeze_mask(W.data_ptr(), rows, cols, bits, float(boundary_range), frozen.data_ptr(),
                                   rng.data_ptr(), _stream()), "ofq_cga_freeze_mask")
    return frozen


def cga_freeze_mask_multi(weights, bits, boundary_range, frozen=None, ranges=None):
    """freeze masks of many weight tensors in three launches per 40 tensors; returns (list of masks, range scratch)."""
    import numpy as np
    dev = weights[0].device
    if frozen is None:
        frozen = [torch.empty_like(w) for w in weights]
    if ranges is None:
        ranges = torch.empty((len(weights), 2), dtype=torch.int32, device=dev)
    assert lib().ofq_cga_tensor_entry_bytes() == 40
    tab = np.empty((len(weights), 5), dtype=np.int64)
    for i, (w, f) in enumerate(zip(weights, frozen)):
        if not w.is_cuda or w.dim() != 2 or not w.is_contiguous():
            raise RuntimeError("ofq_amd: CGA weights must be contiguous 2-D tensors on a HIP device")
        tab[i] = (w.data_ptr(), f.data_ptr(), ranges[i].data_ptr(), w.shape[0], w.shape[1])
    _chk(lib().ofq_cga_freeze_mask_multi(tab.ctypes.data, len(weights), bits, float(boundary_range), _stream()),
         "ofq_cga_freeze_mask_multi")
    return frozen, ranges


def cga_mask_grad_save(grad, W, frozen):
    saved = torch.empty_like(W)
    _chk(lib().ofq_cga_mask_grad_save(grad.data_ptr(), W.data_ptr(), frozen.data_ptr(), saved.data_ptr(), W.numel(),
                                      _stream()), "ofq_cga_mask_grad_save")
    return saved


def cga_restore(W, frozen, saved):
    _chk(lib().ofq_cga_restore(W.data_ptr(), frozen.data_ptr(), saved.data_ptr(), W.numel(), _stream()),
         "ofq_cga_restore")


def permute_tokens(x, idx32):
    """y[b, i, :] = x[b, idx32[i], :] for x (B, N, C) fp32 contiguous, idx32 an int32 permutation of range(N)."""
    _dev(x, "x")
    if x.dim() != 3 or not x.is_contiguous() or x.dtype != torch.float32 or idx32.dtype != torch.int32 \
            or idx32.numel() != x.shape[1]:
        raise ValueError("permute_tokens: (B, N, C) contiguous fp32 and an int32 index of length N")
    y = torch.empty_like(x)
    _chk(lib().ofq_permute_tokens(x.data_ptr(), idx32.data_ptr(), y.data_ptr(), x.shape[0], x.shape[1], x.shape[2],
                                  _stream()), "ofq_permute_tokens")
    return y


# ------------------------------------------------------------------------------------------------ fp32 teacher helpers
def gelu_(x, want_amax=False):
    """exact GELU in place; want_amax: x carries its maximum word afterwards (ops.amax_of)"""
    _dev(x, "x")
    am = amax_word(x.device) if want_amax else None
    _chk(lib().ofq_gelu_fwd(x.data_ptr(), x.data_ptr(), x.numel(), _p(am), _stream()), "ofq_gelu_fwd")
    if am is not None:
        tag_amax(x, am)
    return x


def split_f32_bf16x3(W):
    """fp32 matrix -> (3, rows, cols) bf16 planes with W = p0 + p1 + p2 exactly (one-off split of a frozen weight)."""
    _dev(W, "weight")
    W = W.contiguous()
    planes = torch.empty((3,) + tuple(W.shape), dtype=torch.bfloat16, device=W.device)
    _chk(lib().ofq_split_f32_bf16x3(W.data_ptr(), planes.data_ptr(), W.numel(), W.numel(), _stream()), "ofq_split_f32_bf16x3")
    return planes


def split_f32_f16x2(W):
    """A frozen fp32 weight as two fp16 planes of W * 2^E (E puts max |W| into [2^14, 2^15)): returns (hi, lo, 2^-E) with
    W = (hi + lo) * 2^-E up to 2^-24 |W| for entries within 2^-17 of the maximum (torch ops: a one-off at load time)."""
    _dev(W, "weight")
    import math
    amax = float(W.detach().abs().max())
    E = 14 - math.frexp(amax)[1] + 1 if amax > 0 else 0            # frexp: amax = m * 2^e, m in [0.5, 1)
    Ws = W.detach().double() * (2.0 ** E)
    hi = Ws.to(torch.float16)
    lo = (Ws - hi.double()).to(torch.float16)
    return hi.contiguous(), lo.contiguous(), 2.0 ** (-E)


F16X4_PRODUCTS = 3      # plane products of linear_f16x4: 3 = x_hi W_hi + x_lo W_hi + x_hi W_lo (the dropped x_lo W_lo is 2^-22 of a product)


def linear_f16x4(x2d, planes, bias=None, products=None):
    """y = x2d @ W^T + bias with W as split_f32_f16x2's planes: [x | x] . [hi | lo]^T as ONE two-segment code GEMM
    (ofq_qgemm_bf16s_nt_sk; x split into two fp16 planes inside the kernel, fp32 accumulation): four plane products, or three
    (the trailing planes' product skipped: half of the second segment's MFMAs) when K allows it."""
    hi, lo, inv = planes
    y = torch.empty((x2d.shape[0], hi.shape[0]), dtype=torch.float32, device=x2d.device)
    three = (F16X4_PRODUCTS if products is None else products) == 3 and x2d.shape[1] % 64 == 0
    return qgemm_bf16s_nt_sk([(x2d, hi, None, inv), (x2d, lo, None, inv)], y, col_bias=bias, hi_only_last=three)


def gemm_bf16x3x3_nt(x2d, planes, bias=None, products=9):
    """y[M,N] = x2d[M,K] @ W[N,K]^T + bias with W given as its three bf16 planes (split_f32_bf16x3): fp32-grade product on
    the bf16 matrix cores (products = 9: exact plane pairs; 6: the leading ones)."""
    _dev(x2d, "x")
    M, K = x2d.shape
    N = planes.shape[1]
    y = torch.empty((M, N), dtype=torch.float32, device=x2d.device)
    with _Timed('qgemm_bf16s_nt (3x v_mfma_f32_32x32x16_bf16)', 2.0 * M * N * K):
        _chk(lib().ofq_gemm_bf16x3x3_nt(x2d.data_ptr(), planes.data_ptr(), y.data_ptr(), _p(bias), int(products), M, N, K,
                                        x2d.stride(0), planes.stride(1), N, planes.stride(0), _stream()), "ofq_gemm_bf16x3x3_nt")
    return y
