"""Module-granularity autograd Functions over the HIP kernels (ofq_amd/ops.py).

The reference's autograd graph has ~18 000 ATen nodes per step; here a transformer block is
about twenty Function nodes, each one or two kernel launches, with hand-derived backward products.
Shapes: B batch, N tokens, C channels, H heads, d = C/H, Np = N rounded up to a multiple of 4 (row stride
of the attention matrices, so that their rows stay 16-byte aligned for float4 loads).
"""
import os

import torch

from . import ops
from . import parallel as _parallel


def pad4(n):
    return (n + 3) // 4 * 4


class LinearFn(torch.autograd.Function):
    """y = x_hat @ W_hat^T + bias (qlinear.py:69-71) on already fake-quantised operands."""

    @staticmethod
    def forward(ctx, xq, Wq, bias):
        shp = xq.shape
        x2d = xq.reshape(-1, shp[-1])
        y = ops.linear_fwd(x2d, Wq, bias)
        ctx.save_for_backward(x2d, Wq)
        ctx.has_bias = bias is not None
        ctx.in_shape = shp
        return y.view(*shp[:-1], Wq.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2d, Wq = ctx.saved_tensors
        dy2d = dy.reshape(-1, dy.shape[-1])
        if not dy2d.is_contiguous():
            dy2d = dy2d.contiguous()
        dx = ops.linear_bwd_input(dy2d, Wq) if ctx.needs_input_grad[0] else None
        dW = ops.linear_bwd_weight(dy2d, x2d) if ctx.needs_input_grad[1] else None
        db = ops.colsum(dy2d) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return (dx.view(ctx.in_shape) if dx is not None else None), dW, db


class CodeWeightLinearFn(torch.autograd.Function):
    """LinearFn for a layer whose fake-quantised weight is step[n] * integer code (the W8A8 patch embedding, qlinear.py:166-177:
    LSQ weights; 25 088 x 768 x 384 for DeiT: 173 / 198 / 178 us as fp32 GEMMs).  Forward and input gradient run as code GEMMs
    of the linear layers' kind, and with `xaux` (the image quantiser's int8 codes) the weight gradient as well,
        y[m, n]  = step[n] * sum_k x[m, k] * code[n, k] + bias[n]  (ofq_qgemm_bf16s_nt with col_scale / col_bias: x split)
        dx[m, k] = sum_n (dy[m, n] * step[n]) * code[n, k]         (ofq_qgemm_bf16s_nt: dy split, k-scale = steps)
    -- the products the fp32 GEMMs form with W_hat[n, k] = step[n] * code[n, k], up to fp32 rounding of that factorisation."""

    @staticmethod
    def forward(ctx, xq, Wq, bias, codes, steps, xaux=None):
        x2d = xq.reshape(-1, xq.shape[-1])
        N, K = codes.shape
        ctx.xaux = xaux
        if N > 128 and K % 8 == 0 and x2d.is_contiguous():
            # y = steps[n] * (x . code[n, :]) + bias[n]: the activations as THREE bf16 planes (the exact fp32 product) against the
            # codes, whatever ops.GRAD_PLANES says: the two-plane trade of DESIGN 4b is a backward-only trade, the forward pass --
            # values and integer levels -- is the same bits in both modes (round 5 ran this one forward GEMM on two fp16 planes
            # under GRAD_PLANES = 2; tests/test_planes_fullsize_gpu.py compares the losses of the two modes and found it)
            c16 = codes.to(torch.bfloat16)
            y = ops.qgemm_bf16s_nt(x2d, c16, None, 1.0, col_scale=steps, col_bias=bias)
        else:
            y = ops.linear_fwd(x2d, Wq, bias)
        ctx.save_for_backward(x2d, codes, steps)
        ctx.has_bias = bias is not None
        ctx.in_shape = xq.shape
        return y.view(*xq.shape[:-1], Wq.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2d, codes, steps = ctx.saved_tensors
        am = ops.amax_of(dy)                       # (the producer's maximum word survives the copy below)
        dy2d = dy.reshape(-1, dy.shape[-1])
        if not dy2d.is_contiguous():
            dy2d = dy2d.contiguous()
        if am is not None and ops.amax_of(dy2d) is None:
            ops.tag_amax(dy2d, am)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.qgemm_bf16s_nt(dy2d, ops.codes_transpose_16(codes), steps, 1.0).view(ctx.in_shape)
        dW = db = None
        xa = ctx.xaux
        if ctx.needs_input_grad[1] and xa is not None:
            # x_hat[m, k] = ax[k] * qx[m, k] + boff[m % P, k] (P = rows sharing one offset pattern: the patches of an image):
            #   dW[n, k] = ax[k] * sum_m dy[m, n] qx[m, k]  +  sum_p (sum_b dy[b P + p, n]) * boff[p, k]
            # the first sum is the code GEMM of the linear layers' weight gradients (its column sums of dy are the bias gradient),
            # the second a (N x P) . (P x K) product of the batch-summed gradient
            ones = xa["ones"]
            dWc, db = ops.qgemm_bf16s_tn(dy2d, xa["qx"], ones, ones.numel(), 0.0, None, None, compute_db=True)
            dW = dWc.mul_(xa["ax"])
            P = xa["boff"].shape[0]
            dys = ops.colsum(dy2d.view(-1, P * dy2d.shape[1])).view(P, dy2d.shape[1])
            ops.gemm(dys, xa["boff"], dW, dW.shape[0], dW.shape[1], P, dys.stride(0), xa["boff"].stride(0), dW.stride(0), transA=True,
                     accumulate=True)
            if not (ctx.has_bias and ctx.needs_input_grad[2]):
                db = None
        else:
            dW = ops.linear_bwd_weight(dy2d, x2d) if ctx.needs_input_grad[1] else None
            db = ops.colsum(dy2d) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return dx, dW, db, None, None, None


def code_weight_linear_ok(out_features, in_features, quant):
    """codes in int8 and shapes the code GEMM takes (K = out_features % 8, N = in_features)"""
    return out_features % 8 == 0 and in_features % 4 == 0 and quant.thd_neg >= -128 and quant.thd_pos <= 127


# attention scores backward: also add the (mathematically zero) row-sum term of dx_hat, as the reference's autograd does
KEEP_ZERO_ROWSUM_TERM = os.environ.get("OFQ_KEEP_ZERO_ROWSUM_TERM", "0") == "1"


def _shared_grad(acc, shape, device):
    """Gradient buffer of a tensor with several consumers (x_hat of the QKR attention feeds the v GEMM, the W_qk GEMM and
    the scores): the first backward to run allocates it and returns it to autograd, the later ones accumulate into it
    in place (their GEMMs have an accumulate flag) and return None, so autograd never launches an add kernel.
    Returns (buffer, accumulate, value_to_return).  acc is a per-forward dict or None."""
    if acc is None:
        buf = torch.empty(shape, dtype=torch.float32, device=device)
        return buf, False, buf
    buf = acc.get("buf")
    if buf is None:
        buf = torch.empty(shape, dtype=torch.float32, device=device)
        acc["buf"] = buf
        return buf, False, buf
    return buf, True, None


# accumulators (see CodesLinearFn.backward) that hold a parked input-gradient GEMM: must be empty when a backward pass ends
_DX_PENDING = []


def assert_no_pending_dx():
    """A parked input-gradient GEMM whose partner never arrived would leave its gradient unwritten: fail loudly."""
    if _DX_PENDING:
        n = len(_DX_PENDING)
        del _DX_PENDING[:]
        raise RuntimeError("ofq_amd: %d parked input-gradient GEMM(s) were never launched (a consumer of a shared input "
                           "did not run its backward)" % n)


_END_CHECK_QUEUED = [False]


def _check_parked_at_end_of_backward():
    if _END_CHECK_QUEUED[0]:
        return
    _END_CHECK_QUEUED[0] = True

    def _cb():
        _END_CHECK_QUEUED[0] = False
        assert_no_pending_dx()
    torch.autograd.Variable._execution_engine.queue_callback(_cb)


def assert_step_queues_empty():
    """End of a training step's backward pass (engine._step_body, after flush_dw): no parked input-gradient GEMM, no queued
    weight-gradient GEMM, no deferred second-stage reduction -- each of them stands for gradient memory that autograd already
    handed on unwritten."""
    assert_no_pending_dx()
    # a queued kernel wrote to the address of the tensor autograd was given: the leaf's .grad must BE that tensor (adopted),
    # not a clone AccumulateGrad made of it while it was still unwritten
    # (leaves of a synchronised DataParallel are exempt: its bucket launch copies their gradients into the bucket slices)
    bad = [1 for leaf, addr in _ADOPT_CHECK
           if leaf.grad is not None and leaf.grad.data_ptr() != addr and id(leaf) not in _parallel._GRAD_SLOTS]
    del _ADOPT_CHECK[:]
    if bad:
        raise RuntimeError("ofq_amd: %d deferred gradient(s) were cloned by autograd instead of adopted" % len(bad))
    if _DW_QUEUE or _DW_TILES[0]:
        drop_dw()
        raise RuntimeError("ofq_amd: weight-gradient GEMMs were still queued after flush_dw()")
    if ops.sum_pending():
        ops.sum_drop()
        raise RuntimeError("ofq_amd: second-stage reductions were still deferred after flush_dw()")


# ---- deferred weight gradients -----------------------------------------------------------------------------------------
# dW of a linear layer has no consumer inside the backward pass (StatsQ's backward is the identity, so it goes straight to
# the parameter's .grad, or to AllWqkFn at the very end).  Inside engine's training step (DW_DEFER set around
# loss.backward()) CodesLinearFn.backward therefore only *queues* its dW GEMM -- output tensors allocated, returned to
# autograd, not yet written -- and the queue is launched as ONE grouped GEMM (ops.qgemm_bf16s_tn_group) once it holds about
# a transformer block's worth of tiles: long token ranges per workgroup instead of 9-57 k-steps, a fifth of the split-K
# partials.  Everything that READS a gradient flushes first: the bucket all-reduce (parallel.DataParallel._launch),
# AllWqkFn.backward, and engine._step_body right after loss.backward().  Outside the training step nothing is deferred.
DW_DEFER = False
DW_GROUP = os.environ.get("OFQ_NO_DW_GROUP") is None       # A/B switch
DW_FLUSH_TILES = int(os.environ.get("OFQ_DW_FLUSH_TILES", "46"))   # a DeiT-S QKR block queues 12 + 12 + 3 + 18 + 3 = 48
_DW_QUEUE = []
_DW_TILES = [0]


# The same holds for the second-stage reductions of the quantiser / LayerNorm backward kernels (d step, d offset, d gamma,
# d beta): inside the training step they are queued in the library (ops.deferred_sums) and launched together with the
# block's dW GEMMs, forty per launch instead of one launch each (~95 launches of 5-15 us per DeiT-S step).
SUM_DEFER = os.environ.get("OFQ_NO_SUM_DEFER") is None     # A/B switch


# Leaves that own a queued (= handed to autograd, not yet written) gradient.  A second gradient for such a leaf (module
# called twice, tied weights, shared quantiser) is ADDED to the first one by autograd the moment its node returns -- so the
# first one has to be written by then: whoever is about to produce a gradient for a leaf in this set flushes the queues
# first (_settle) and takes the immediate path.  Emptied by every flush.
_QUEUED_LEAVES = set()
# ... and leaves that have been given a queued gradient at any time in the running backward pass.  Autograd keeps the
# gradients of a leaf with several producers in the input buffer of its AccumulateGrad node until the last one has arrived
# (`leaf.grad` stays None meanwhile) and ADDS each newcomer to the buffer at once: a second gradient must therefore never be
# queued, also when the first one has been flushed in between.  Lives for one backward pass (begin_backward).
_DEFERRED_ONCE = set()
_ADOPT_CHECK = []          # (leaf, address the queued kernel wrote): verified at the end of the step


def _forget_parked_dx():
    """Drop input-gradient GEMMs parked by a backward pass that never finished (their dY / weight operands stay pinned otherwise,
    and the next step would fail in assert_no_pending_dx for something that step did not do)."""
    for acc in _DX_PENDING:
        acc.pop("lin_pending", None)
    del _DX_PENDING[:]
    _END_CHECK_QUEUED[0] = False


def begin_backward():
    """engine._step_body, right before loss.backward()."""
    _forget_parked_dx()
    _QUEUED_LEAVES.clear()
    _DEFERRED_ONCE.clear()
    del _ADOPT_CHECK[:]


def _settle(*leaves):
    """Before an immediate gradient for `leaves` goes back to autograd: launch the queues if one of them still waits there."""
    if any(t is not None and id(t) in _QUEUED_LEAVES for t in leaves):
        flush_dw()


def _claim(*leaves):
    """True (and the leaves marked) when their gradients may be queued; False after settling when one of them already owns a
    queued gradient, or the same leaf appears twice."""
    ids = [id(t) for t in leaves if t is not None]
    if any(i in _DEFERRED_ONCE for i in ids) or len(set(ids)) != len(ids):
        _settle(*leaves)
        return False
    _QUEUED_LEAVES.update(ids)
    _DEFERRED_ONCE.update(ids)
    return True


def _sums_deferrable(*leaves):
    """True when the parameter gradients a backward kernel is about to produce may be written later: inside the training
    step, and every receiving tensor is a leaf whose .grad is empty -- autograd then adopts the returned tensor as the
    gradient without reading it (an existing .grad would be accumulated into, i.e. read, right away)."""
    if not (DW_DEFER and SUM_DEFER):
        return False
    for t in leaves:
        if t is None:
            continue
        # (a leaf that does not require a gradient: autograd drops the returned tensor at once, and the queued kernel
        # would write into memory that has been handed to somebody else by then)
        if not t.is_leaf or not t.requires_grad:
            return False
        if t.grad is not None:       # the new gradient will be accumulated into .grad: that one must not be waiting in a queue
            _settle(*leaves)
            return False
    return _claim(*leaves)


class _Immediate:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def sum_scope(*leaves):
    """Context for a backward kernel call: ops.deferred_sums() when its reductions may wait (see _sums_deferrable)."""
    return ops.deferred_sums() if _sums_deferrable(*leaves) else _Immediate()


def has_queued_work():
    """Anything for flush_dw() to launch?  (host-side counters only: never loads the library, parallel.DataParallel asks
    from its gradient hooks, also on the CPU / gloo path)"""
    return bool(_DW_QUEUE) or bool(ops._SUM_KEEP) or (ops._lib_handle is not None and ops._lib_handle.ofq_sum_pending() != 0)


def flush_dw():
    """Launch every queued weight-gradient GEMM and second-stage reduction (no-op when the queues are empty)."""
    q = _DW_QUEUE
    ops.sum_flush()
    try:
        while q:
            three = q[0]["xcodes2d"].shape[1] % 384 == 0
            n = 1
            while n < len(q) and n < ops.TN_GROUP_MAX and (q[n]["xcodes2d"].shape[1] % 384 == 0) == three:
                n += 1
            ops.qgemm_bf16s_tn_group(q[:n])
            del q[:n]
    except BaseException:
        drop_dw()                 # never carry raw output addresses of a failed step into the next one
        raise
    _DW_TILES[0] = 0
    _QUEUED_LEAVES.clear()        # everything that was queued is on the stream now


def drop_dw():
    """Forget the queue (a backward pass that raised)."""
    _forget_parked_dx()
    del _DW_QUEUE[:]
    _DW_TILES[0] = 0
    _QUEUED_LEAVES.clear()
    ops.sum_drop()


def queue_dw(dy2d, xcodes2d, lsq_s, S, gscale, baft, out, db_to_autograd=True):
    """Queue dW = dy2d^T @ (a_eff * codes + baft) and db = colsum(dy2d); returns the (not yet written) output tensors.
    db_to_autograd False: the caller will NOT hand db to autograd (a layer without a bias, or whose bias takes no gradient).
    The kernel still writes it (the column sums feed the offset term of dW), so the queue keeps the tensor alive until the
    flush -- dropped by the caller it would be freed at once, its memory handed to the next small tensor of the backward
    pass (at 128 images: the value quantiser's three gradients) and overwritten there when the queue is flushed."""
    M, N = dy2d.shape[1], xcodes2d.shape[1]
    dW = out if out is not None else torch.empty((M, N), dtype=torch.float32, device=dy2d.device)
    db = torch.empty(M, dtype=torch.float32, device=dy2d.device)
    q = _DW_QUEUE
    if q and ((q[-1]["xcodes2d"].shape[1] % 384 == 0) != (N % 384 == 0) or len(q) >= ops.TN_GROUP_MAX):
        flush_dw()
    # the outputs are queued as raw addresses: autograd must hold the only reference to dW / db, or AccumulateGrad would
    # clone them (still unwritten) instead of adopting them as .grad; they stay alive as .grad / in autograd's input buffers
    q.append({"dy2d": dy2d, "xcodes2d": xcodes2d, "lsq_s": lsq_s, "S": S, "gscale": gscale, "baft": baft,
              "dW": dW.data_ptr(), "db": db.data_ptr(), "db_keep": None if db_to_autograd else db,
              # (the maximum word travels as an attribute of the gradient tensor: looked up now, while that object is at hand)
              "amax": ops._planes_amax(dy2d, None)})
    _DW_TILES[0] += ops.tn_tiles(M, N)
    if _DW_TILES[0] >= DW_FLUSH_TILES:
        flush_dw()
    return dW, (db if db_to_autograd else None)


class CodesLinearFn(torch.autograd.Function):
    """Same function as LinearFn, computed on the integer codes: forward is one exact int8-MFMA GEMM with the
    scales applied in the epilogue (ofq_qgemm_i8_nt), dX is the 3-way bf16-split GEMM against the transposed
    weight codes (ofq_qgemm_bf16s_nt, fp32-exact), dW stays a split-K fp32-MFMA GEMM on the fake-quant input.
    `xq` / `Wq` are the fp32 fake-quant tensors (they carry the autograd edges to the quantisers); `aux` holds
    the non-differentiable codes and scales."""

    @staticmethod
    def forward(ctx, xq, Wq, bias, aux):
        shp = xq.shape
        K = shp[-1]
        x2d = xq.reshape(-1, K)
        r = aux.get("r")
        if r is None and aux["baft"] is not None:
            r = ops.rowdot_i8(aux["wcodes"], aux["baft"])
        fuse = aux.get("fuse")
        # the consumer of this output is a quantiser applied right here (fuse) that asked not to keep the fp32 values
        # (store_y False): only its codes are written; its backward recomputes the output from the codes
        store_y = not (fuse is not None and fuse.get("store_y") is False)
        y = ops.qgemm_i8_nt(aux["xcodes"].view(-1, K), aux["wcodes"], bias, aux["w_scale"], aux["w_mult"], r,
                            aux["act_s"], aux["act_S"], aux["act_gscale"], fuse=fuse, store_y=store_y)
        if not store_y:
            fuse["producer"] = {"xcodes": aux["xcodes"].view(-1, K), "wcodes": aux["wcodes"],
                                "bias": None if bias is None else bias.detach(), "w_scale": aux["w_scale"],
                                "w_mult": aux["w_mult"], "r": r, "act_s": aux["act_s"], "act_S": aux["act_S"],
                                "act_gscale": aux["act_gscale"]}
            y = ops.placeholder((x2d.shape[0], Wq.shape[0]), xq.device)
        ctx.codes_only = xq.stride(-1) == 0           # x_hat exists only as codes (placeholder carrier tensor)
        ctx.save_for_backward(*(() if ctx.codes_only else (x2d,)))
        ctx.aux = aux
        ctx.has_bias = bias is not None
        ctx.bias_leaf = bias if (bias is not None and bias.is_leaf) else None
        ctx.in_shape = shp
        return y.view(*shp[:-1], Wq.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2d = None if ctx.codes_only else ctx.saved_tensors[0]
        aux = ctx.aux
        dy2d = dy.reshape(-1, dy.shape[-1])
        if not dy2d.is_contiguous():
            dy2d = dy2d.contiguous()
        dx = None
        link = aux.get("lsq_link")
        if ctx.needs_input_grad[0] and link and "geom" in link and ctx.codes_only:
            # the input quantiser's backward runs in the dX GEMM's epilogue; its four gradients wait in the link for
            # _LsqFn.backward, the carrier tensor gets a zero-stride dummy gradient
            x_in = link["x"]
            x2 = x_in.reshape(-1, ctx.in_shape[-1])
            res = ops.qgemm_bf16s_nt_lsq(dy2d, aux["wcodesT"], aux["w_scale"], aux["w_mult"], x2, link["s"], link["b4"],
                                         link["geom"])
            link["done"] = (res[0].view(x_in.shape), res[1], res[2], res[3])
            dx = ops.placeholder(ctx.in_shape, dy2d.device)
        elif ctx.needs_input_grad[0]:
            K_in0 = ctx.in_shape[-1]
            acc = aux.get("xgrad_acc")
            buf, accumulate, dx = _shared_grad(acc, ctx.in_shape, dy2d.device)
            seg = (dy2d, aux["wcodesT"], aux["w_scale"], aux["w_mult"])
            # Several linear layers read this input (v and W_qk of the QKR attention, attention.py:180, :200): their input
            # gradients are ONE GEMM over the concatenated contraction, [dY_v | dY_qkx] . [W_v ; W_qk].  Every layer but the
            # last to arrive leaves its operands in `acc` (the tuple keeps dY alive); the last one launches.  A layer parks
            # only when the buffer already holds a gradient (accumulate): the first writer must write.
            npend = 0 if acc is None else len(acc.get("lin_pending", ()))
            total = 1 if acc is None else acc.get("lin_total", 1)
            if (accumulate and total > 1 and npend + 1 < total - acc.get("lin_done", 0)
                    and ops.nt_concat_ok(dy2d, aux["wcodesT"], acc.get("lin_pending"))):
                acc.setdefault("lin_pending", []).append(seg)
                _DX_PENDING.append(acc)
                if not DW_DEFER:
                    # outside engine's step (whose end runs assert_step_queues_empty) a partner that never arrives would go
                    # unnoticed: have the END of this backward pass check it (autograd's per-pass callback queue)
                    _check_parked_at_end_of_backward()
            elif npend:
                segs = acc.pop("lin_pending") + [seg]
                _DX_PENDING[:] = [a_ for a_ in _DX_PENDING if a_ is not acc]
                if ops.nt_concat_ok(dy2d, aux["wcodesT"], segs[:-1]):
                    ops.qgemm_bf16s_nt_sk(segs[::-1], buf.view(-1, K_in0), accumulate=accumulate)
                else:
                    for a_, b_, ks_, al_ in segs:
                        ops.qgemm_bf16s_nt(a_, b_, ks_, al_, out=buf.view(-1, K_in0), accumulate=accumulate)
                        accumulate = True
                acc["lin_done"] = acc.get("lin_done", 0) + len(segs)
            else:
                ops.qgemm_bf16s_nt(dy2d, aux["wcodesT"], aux["w_scale"], aux["w_mult"], out=buf.view(-1, K_in0),
                                   accumulate=accumulate)
                if acc is not None:
                    acc["lin_done"] = acc.get("lin_done", 0) + 1
        need_db = (ctx.has_bias and ctx.needs_input_grad[2]) or aux["baft"] is not None
        dW = db = None
        N_out, K_in = dy2d.shape[1], ctx.in_shape[-1]
        if ctx.needs_input_grad[1] and N_out % 4 == 0 and K_in % 16 == 0:
            # dY^T @ (a_eff*codes + baft): bf16-split TN GEMM on the codes + rank-1 offset term; the same pass over
            # dY also yields the bias gradient (column sums)
            xc2 = aux["xcodes"].view(-1, K_in)
            slot = _parallel.grad_slot(aux.get("w_leaf"))
            if slot is None and aux.get("w_grad_out") is not None and not aux["w_grad_out_used"][0]:
                slot = aux["w_grad_out"]           # (W_qk: its slice of AllWqkFn's gradient buffer; one writer per backward)
                aux["w_grad_out_used"][0] = True
            # a queued result is still unwritten when autograd accumulates it: that is only sound when accumulating means
            # adopting the tensor (no gradient there yet), never adding to one
            w_leaf, b_leaf = aux.get("w_leaf"), ctx.bias_leaf
            # (a weight that is not a leaf -- W_qk -- may wait only when the node that consumes its gradient flushes the
            # queue before reading it: WqkFn / AllWqkFn say so through aux["w_flushes"])
            adopt = ((w_leaf.grad is None if w_leaf is not None else bool(aux.get("w_flushes")))
                     and (b_leaf is None or b_leaf.grad is None) and (not ctx.has_bias or b_leaf is not None))
            if (DW_DEFER and DW_GROUP and adopt and ops.tn_groupable(dy2d.shape[0], N_out, K_in, aux["act_S"],
                                                                     dy2d.stride(0), xc2.stride(0))
                    and _claim(w_leaf, b_leaf)):
                dW, db = queue_dw(dy2d, xc2, aux["act_s"], aux["act_S"], aux["act_gscale"], aux["baft"], slot,
                                  db_to_autograd=bool(ctx.has_bias and ctx.needs_input_grad[2]))
                if w_leaf is not None:
                    _ADOPT_CHECK.append((w_leaf, dW.data_ptr()))
                if b_leaf is not None and db is not None:
                    _ADOPT_CHECK.append((b_leaf, db.data_ptr()))
            else:
                _settle(w_leaf, b_leaf)
                dW, db = ops.qgemm_bf16s_tn(dy2d, xc2, aux["act_s"], aux["act_S"], aux["act_gscale"], None, aux["baft"],
                                            compute_db=True, out=slot)
        else:
            _settle(aux.get("w_leaf"), ctx.bias_leaf)
            db = ops.colsum(dy2d) if need_db else None
            dW = ops.linear_bwd_weight(dy2d, x2d) if ctx.needs_input_grad[1] else None
        return dx, dW, (db if ctx.has_bias else None), None


def codes_only_ok(in_features, out_features):
    """x_hat may stay un-materialised (codes only) when the weight-gradient GEMM can run on the codes as well."""
    return in_features % 16 == 0 and out_features % 4 == 0


def codes_linear_ok(in_features, wquant, act_quant):
    """The code GEMMs need K % 16 == 0, StatsQ weights of <= 7 bits (2L+1 in int8) and activation codes in int8."""
    return (in_features % 16 == 0 and wquant.num_bits <= 7 and act_quant.thd_neg >= -128 and act_quant.thd_pos <= 127)


def codes_linear(xq, xcodes, geom, act_quant, baft, weight, wquant, bias, fuse=None, lsq_link=None, xgrad_acc=None):
    """y = xq @ StatsQ(weight)^T + bias on the integer codes.  xq/xcodes/geom come from LsqQuantizer.quant(want_codes=True).
    `fuse`: see ops.qgemm_i8_nt (the next layer's input codes as a by-product of this GEMM's epilogue)."""
    bvec = baft.detach() if baft is not None else None
    Wq = wquant(weight, want_codes=True, rvec=bvec, need_values=False)
    aux = {"xcodes": xcodes, "wcodes": wquant._codes, "w_scale": wquant._s_dev, "r": wquant._r,
           "wcodesT": wquant.codes_T() if torch.is_grad_enabled() else None,   # bf16 [in][out] for dX
           "w_mult": 1.0 / float(2 ** wquant.num_bits), "baft": baft.detach() if baft is not None else None,
           "act_s": act_quant.s.detach(), "act_S": geom.S, "act_gscale": geom.gscale, "fuse": fuse, "lsq_link": lsq_link, "xgrad_acc": xgrad_acc,
           # StatsQ's backward is the identity (statsq.py:148), so dW of a leaf weight IS its .grad: see parallel.grad_slot
           "w_leaf": weight if (weight.is_leaf and weight.requires_grad) else None,
           "w_flushes": bool(getattr(weight, "_ofq_flushes", False)),
           "w_grad_out": getattr(weight, "_ofq_grad_out", None), "w_grad_out_used": [False]}
    return CodesLinearFn.apply(xq, Wq, bias, aux)


class WqkFn(torch.autograd.Function):
    """W_qk[h] = W_q[h]^T @ W_k[h]  ->  (H*C, C)   (attention.py:190-194)."""

    @staticmethod
    def forward(ctx, Wq, Wk, H):
        C = Wq.shape[1]
        d = Wq.shape[0] // H
        out = torch.empty((H * C, C), dtype=torch.float32, device=Wq.device)
        ops.gemm(Wq, Wk, out, C, C, d, C, C, C, transA=True, nb0=H, sA=(d * C, 0), sB=(d * C, 0), sC=(C * C, 0),
                 tile_hint=64 if d <= 64 else 0)
        ctx.save_for_backward(Wq, Wk)
        ctx.H = H
        return out

    @staticmethod
    def backward(ctx, g):
        flush_dw()                         # (the promise behind wqk(): a queued W_qk gradient is written before it is read)
        Wq, Wk = ctx.saved_tensors
        H = ctx.H
        C = Wq.shape[1]
        d = Wq.shape[0] // H
        g = g.contiguous()
        dWq = torch.empty_like(Wq)
        dWk = torch.empty_like(Wk)
        # dWq[h][j,c] = sum_c' Wk[h][j,c'] g[h][c,c']      (NT)
        # d = 64 rows per head: 64x64 tiles (the 128x128 default leaves half of every tile empty: 43 -> 17 us each)
        ops.gemm(Wk, g, dWq, d, C, C, C, C, C, transB=True, nb0=H, sA=(d * C, 0), sB=(C * C, 0), sC=(d * C, 0),
                 tile_hint=64 if d <= 64 else 0)
        # dWk[h][j,c'] = sum_c Wq[h][j,c] g[h][c,c']       (NN)
        ops.gemm(Wq, g, dWk, d, C, C, C, C, C, nb0=H, sA=(d * C, 0), sB=(C * C, 0), sC=(d * C, 0),
                 tile_hint=64 if d <= 64 else 0)
        return dWq, dWk, None


class AllWqkFn(torch.autograd.Function):
    """WqkFn for all blocks of a model in one batched GEMM each way (batch = blocks x heads): W_qk only depends on
    parameters, and the per-block products are tiny (0.11 GFLOP, ~25-33 us each, latency-bound: three launches per block).
    Inputs: H, then q.weight, k.weight of every block; outputs: one (H*C, C) tensor per block (views of one buffer).
    The backward runs once, when the last block's gradient has arrived, i.e. at the end of the model's backward."""

    @staticmethod
    def forward(ctx, H, want_gbuf, *ws):
        L = len(ws) // 2
        Wq = torch.stack(ws[0::2])                                    # (L, H*d, C)
        Wk = torch.stack(ws[1::2])
        C = Wq.shape[2]
        d = Wq.shape[1] // H
        out = torch.empty((L, H * C, C), dtype=torch.float32, device=Wq.device)
        # (K = d = 64: 64 x 64 tiles -- the 128 x 128 default takes 185 us for the 72 products of DeiT-S, tools/wqk_bench.py)
        ops.gemm(Wq, Wk, out, C, C, d, C, C, C, transA=True, nb0=L * H, sA=(d * C, 0), sB=(d * C, 0), sC=(C * C, 0),
                 tile_hint=64 if d <= 64 else 0)
        ctx.save_for_backward(Wq, Wk)
        ctx.H = H
        ctx.set_materialize_grads(False)
        # one buffer for the L gradients that come back: the dW GEMMs of the W_qk layers write their slices of it directly
        # (all_wqk hands the slices out as `_ofq_grad_out`), and the backward below reads it without a 42 MB torch.stack.
        # Only when a backward pass can follow (want_gbuf: decided by the caller, grad mode is off inside a Function's forward);
        # it travels back as a non-differentiable extra output, not through a module global
        ctx.gbuf = torch.empty((L, H * C, C), dtype=torch.float32, device=Wq.device) if want_gbuf else None
        extra = ctx.gbuf if ctx.gbuf is not None else out.new_empty(0)
        ctx.mark_non_differentiable(extra)
        return tuple(out.unbind(0)) + (extra,)

    @staticmethod
    def backward(ctx, *gs):
        flush_dw()                         # the W_qk gradients of the last blocks may still be queued
        gs = gs[:-1]                       # (the last output is the gradient buffer itself)
        Wq, Wk = ctx.saved_tensors
        H = ctx.H
        L, _, C = Wq.shape
        d = Wq.shape[1] // H
        gb = getattr(ctx, "gbuf", None)
        if gb is not None and all(gi is not None and gi.data_ptr() == gb[l].data_ptr() and gi.is_contiguous() for l, gi in enumerate(gs)):
            g = gb                     # every block's dW GEMM wrote its slice in place
        else:
            g = torch.stack([gi if gi is not None else torch.zeros((H * C, C), dtype=Wq.dtype, device=Wq.device) for gi in gs])
        dWq = torch.empty_like(Wq)
        dWk = torch.empty_like(Wk)
        ops.gemm(Wk, g, dWq, d, C, C, C, C, C, transB=True, nb0=L * H, sA=(d * C, 0), sB=(C * C, 0), sC=(d * C, 0),
                 tile_hint=64 if d <= 64 else 0)
        ops.gemm(Wq, g, dWk, d, C, C, C, C, C, nb0=L * H, sA=(d * C, 0), sB=(C * C, 0), sC=(d * C, 0),
                 tile_hint=64 if d <= 64 else 0)
        grads = [None, None]
        for l in range(L):
            grads += [dWq[l], dWk[l]]
        return tuple(grads)


BULK_WQK = os.environ.get("OFQ_NO_BULK_WQK", "0") != "1"
STEP_CACHE_ACTIVE = False          # set by engine.train_step around the forward (see engine.refresh_weight_codes)


def all_wqk(attns):
    """Pre-compute W_qk (and, inside a training step, its StatsQ operands) for every QKR attention module of `attns` that
    shares one shape; each module picks its tensor up from `_wqk_pre` in its forward.  Returns the modules served."""
    if not BULK_WQK or len(attns) < 2:
        return []
    a0 = attns[0]
    ok = all(hasattr(a, "q") and hasattr(a, "k") and hasattr(a, "qk_quant") and a.q.weight.shape == a0.q.weight.shape
             and a.num_heads == a0.num_heads and a.q.weight.is_cuda for a in attns)
    if not ok:
        return []
    ws = []
    for a in attns:
        ws += [a.q.weight, a.k.weight]
    want_gbuf = torch.is_grad_enabled() and any(w.requires_grad for w in ws)
    outs = AllWqkFn.apply(a0.num_heads, want_gbuf, *ws)
    outs, gbuf = outs[:-1], (outs[-1] if want_gbuf else None)
    for l, (a, w) in enumerate(zip(attns, outs)):
        w._ofq_flushes = True              # AllWqkFn.backward flushes the dW queue before it reads this tensor's gradient
        if gbuf is not None:
            w._ofq_grad_out = gbuf[l]      # where this block's W_qk gradient is to be written (CodesLinearFn.backward)
        a._wqk_pre = w
    if STEP_CACHE_ACTIVE:
        # the StatsQ operands of the 12 W_qk in one launch (their per-block launches are as latency-bound as the GEMMs)
        todo = [(a, a.qk_quant._last_args) for a in attns if a.qk_quant._last_args is not None]
        if len(todo) == len(attns):
            res = ops.statsq_codes_multi([(a._wqk_pre.detach(), la[1], None if la[2] is None else la[2].detach(), la[3])
                                          for a, la in todo])
            for (a, la), r in zip(todo, res):
                a.qk_quant._pre = (a._wqk_pre, la[2], la[3], r)
    return attns


def all_plain_prep(attns):
    """The per-head offset operands of the plain attention's scores -- eye_h * bk, eye_h * bq (H rows of length C: the head's
    channels of the offset vector, zeros elsewhere) and z[h] = bq|h . bk|h (attention.py:96 with the offsets of :77-78 expanded) --
    for ALL blocks of a model in five small launches instead of four per block; they depend on parameters only.  Each module picks
    its slices up from `_plain_pre` (ScoresSoftmaxCodesFn / QKScoresCodesFn); returns the modules served."""
    if not BULK_WQK or len(attns) < 2 or os.environ.get("OFQ_NO_PLAIN_PREP") is not None:      # (A/B switch)
        return []
    a0 = attns[0]
    ok = all(hasattr(a, "move_q_aft") and hasattr(a, "move_k_aft") and not hasattr(a, "qk_quant") and a.num_heads == a0.num_heads
             and a.move_q_aft.bias.shape == a0.move_q_aft.bias.shape and a.move_q_aft.bias.is_cuda for a in attns)
    if not ok:
        return []
    H = a0.num_heads
    C = a0.move_q_aft.bias.numel()
    with torch.no_grad():
        bq = torch.stack([a.move_q_aft.bias.detach().reshape(C) for a in attns])            # (L, C)
        bk = torch.stack([a.move_k_aft.bias.detach().reshape(C) for a in attns])
        eye = _head_mask(H, C // H, bq.device)                                                 # (H, C)
        ebk, ebq = eye.unsqueeze(0) * bk.unsqueeze(1), eye.unsqueeze(0) * bq.unsqueeze(1)     # (L, H, C): same products as eye * b
        z = (bq * bk).view(len(attns), H, C // H).sum(2)                                       # (L, H): same sums as .view(H, d).sum(1)
    for l, a in enumerate(attns):
        a._plain_pre = (ebk[l], ebq[l], z[l])
    return attns


class QKRScoresFn(torch.autograd.Function):
    """S[b,h,n,m] = sum_c xq[b,n,c] * qkx[b,m,h,c]   (attention.py:207-210), S stored (B,H,N,Np)."""

    @staticmethod
    def forward(ctx, xq, qkx, H):
        B, N, C = xq.shape
        Np = pad4(N)
        S = torch.empty((B, H, N, Np), dtype=torch.float32, device=xq.device)
        ops.gemm(xq, qkx, S, N, N, C, C, H * C, Np, transB=True, nb0=B, nb1=H, sA=(N * C, 0), sB=(N * H * C, C),
                 sC=(H * N * Np, N * Np))
        ctx.save_for_backward(xq, qkx)
        ctx.H = H
        return S

    @staticmethod
    def backward(ctx, dS):
        xq, qkx = ctx.saved_tensors
        H = ctx.H
        B, N, C = xq.shape
        Np = pad4(N)
        dS = dS.contiguous()
        dxq = torch.empty_like(xq)
        dqkx = torch.empty_like(qkx)
        # dxq[b,n,c] = sum_h sum_m dS[b,h,n,m] qkx[b,m,h,c]
        ops.gemm(dS, qkx, dxq, N, C, N, Np, H * C, C, nb0=B, sA=(H * N * Np, 0), sB=(N * H * C, 0), sC=(N * C, 0),
                 nkb=H, sAk=N * Np, sBk=C)
        # dqkx[b,m,h,c] = sum_n dS[b,h,n,m] xq[b,n,c]
        ops.gemm(dS, xq, dqkx, N, C, N, Np, C, H * C, transA=True, nb0=B, nb1=H, sA=(H * N * Np, N * Np),
                 sB=(N * C, 0), sC=(N * H * C, C))
        return dxq, dqkx, None


class QKScoresFn(torch.autograd.Function):
    """Plain attention scores S[b,h,n,m] = sum_j q[b,n,h*d+j] k[b,m,h*d+j]  (attention.py:96, before *scale)."""

    @staticmethod
    def forward(ctx, q, k, H):
        B, N, C = q.shape
        d = C // H
        Np = pad4(N)
        S = torch.empty((B, H, N, Np), dtype=torch.float32, device=q.device)
        ops.gemm(q, k, S, N, N, d, C, C, Np, transB=True, nb0=B, nb1=H, sA=(N * C, d), sB=(N * C, d),
                 sC=(H * N * Np, N * Np))
        ctx.save_for_backward(q, k)
        ctx.H = H
        return S

    @staticmethod
    def backward(ctx, dS):
        q, k = ctx.saved_tensors
        H = ctx.H
        B, N, C = q.shape
        d = C // H
        Np = pad4(N)
        dS = dS.contiguous()
        dq = torch.empty_like(q)
        dk = torch.empty_like(k)
        # dq[b,n,hd+j] = sum_m dS[b,h,n,m] k[b,m,hd+j]           (NN)
        ops.gemm(dS, k, dq, N, d, N, Np, C, C, nb0=B, nb1=H, sA=(H * N * Np, N * Np), sB=(N * C, d), sC=(N * C, d))
        # dk[b,m,hd+j] = sum_n dS[b,h,n,m] q[b,n,hd+j]           (TN)
        ops.gemm(dS, q, dk, N, d, N, Np, C, C, transA=True, nb0=B, nb1=H, sA=(H * N * Np, N * Np), sB=(N * C, d),
                 sC=(N * C, d))
        return dq, dk, None


def _addend_grad(dS, addend, alpha):
    """d(addend)[p] = sum over the batch rows that used slab p of dS / alpha   (S' = alpha*S + addend)."""
    B, H, N, Np = dS.shape
    P = addend.shape[0]
    G = B * H // P
    if dS.is_cuda and dS.is_contiguous() and dS.dtype == torch.float32 and G > 1:
        # the library's two-stage column sum, not torch.sum: over 8192 windows (Swin-T's first block) ATen's multi-workgroup
        # reduction returned different last bits in 1 of ~150 replays of the captured step (tools/swin_relpos_trace.py; on its own
        # the same call is reproducible: tools/relpos_determinism_probe.py) -- the one gradient of the Swin step that did
        return ops.colsum(dS.view(G, P * N * Np)).view(P, N, Np) / alpha
    return dS.reshape(G, P, N, Np).sum(0) / alpha


class SoftmaxLsqFn(torch.autograd.Function):
    """P_hat = LSQ_unsigned(softmax(S * alpha [+ addend]))   (attention.py:96-99 / :213-216; Swin adds the
    relative-position bias and shift mask, swin_attention_and_mlp.py:201-224).  s per query token."""

    @staticmethod
    def forward(ctx, S, s, N, alpha, hi, addend=None):
        B, H = S.shape[0], S.shape[1]
        Np = S.shape[3]
        rows = B * H * N
        prob, y = ops.softmax_lsq_fwd(S, s, rows, N, Np, N, alpha, hi, B * H * N, addend=addend)
        ctx.save_for_backward(prob, s)
        ctx.addend = addend
        ctx.meta = (rows, N, Np, alpha, hi, B * H * N)
        ctx.sum_leaves = (s,)
        return y

    @staticmethod
    def backward(ctx, g):
        prob, s = ctx.saved_tensors
        rows, N, Np, alpha, hi, M = ctx.meta
        g = g.contiguous()
        with sum_scope(*ctx.sum_leaves):
            dS, ds = ops.softmax_lsq_bwd(g, prob, s, rows, N, Np, N, alpha, hi, M, inplace=True)
        dadd = _addend_grad(dS, ctx.addend, alpha) if (ctx.addend is not None and ctx.needs_input_grad[5]) else None
        return dS, ds, None, None, None, dadd


class PVFn(torch.autograd.Function):
    """O[b,n,h*d+j] = sum_m P[b,h,n,m] v[b,m,h*d+j]   ((attn @ v).transpose(1,2).reshape, attention.py:102/:219)."""

    @staticmethod
    def forward(ctx, P, v, N):
        B, H = P.shape[0], P.shape[1]
        Np = P.shape[3]
        C = v.shape[2]
        d = C // H
        O = torch.empty((B, N, C), dtype=torch.float32, device=v.device)
        ops.gemm(P, v, O, N, d, N, Np, C, C, nb0=B, nb1=H, sA=(H * N * Np, N * Np), sB=(N * C, d), sC=(N * C, d))
        ctx.save_for_backward(P, v)
        return O

    @staticmethod
    def backward(ctx, dO):
        P, v = ctx.saved_tensors
        B, H, N, Np = P.shape
        C = v.shape[2]
        d = C // H
        dO = dO.contiguous()
        dP = torch.empty_like(P)
        dv = torch.empty_like(v)
        # dP[b,h,n,m] = sum_j dO[b,n,hd+j] v[b,m,hd+j]            (NT); pad columns are never read downstream
        ops.gemm(dO, v, dP, N, N, d, C, C, Np, transB=True, nb0=B, nb1=H, sA=(N * C, d), sB=(N * C, d),
                 sC=(H * N * Np, N * Np))
        # dv[b,m,hd+j] = sum_n P[b,h,n,m] dO[b,n,hd+j]            (TN)
        ops.gemm(P, dO, dv, N, d, N, Np, C, C, transA=True, nb0=B, nb1=H, sA=(H * N * Np, N * Np), sB=(N * C, d),
                 sC=(N * C, d))
        return dP, dv, None


class QKVSplitLsqFn(torch.autograd.Function):
    """Plain path: qkv (+ move_qkv_b4) split into q,k,v thirds, q/k per-token LSQ, v per-channel LSQ, then the
    per-third post offsets (attention.py:71-90), done in place on column slices of the (B*N, 3C) projection."""

    @staticmethod
    def forward(ctx, qkv, b4, sq, sk, sv, baq, bak, bav, gq, gk, gv):
        B, N, C3 = qkv.shape
        C = C3 // 3
        qkv2 = qkv.reshape(B * N, C3)
        outs = []
        for i, (s, ba, g) in enumerate(((sq, baq, gq), (sk, bak, gk), (sv, bav, gv))):
            y, _ = ops.lsq_fwd(qkv2[:, i * C:], s, b4[i * C:(i + 1) * C], ba, g)
            outs.append(y.view(B, N, C))
        ctx.save_for_backward(qkv2, b4, sq, sk, sv)
        ctx.geoms = (gq, gk, gv)
        ctx.shape = (B, N, C)
        return tuple(outs)

    @staticmethod
    def backward(ctx, dq, dk, dv):
        qkv2, b4, sq, sk, sv = ctx.saved_tensors
        B, N, C = ctx.shape
        dqkv = torch.empty_like(qkv2)
        db4_all = torch.empty(3 * C, dtype=torch.float32, device=dqkv.device)       # the three launches write their slice (no cat)
        res = []
        am = ops.amax_out(dqkv.device)       # ONE maximum word for the three column slices: the qkv projection's backward GEMMs
        for i, (s, g, dy) in enumerate(((sq, ctx.geoms[0], dq), (sk, ctx.geoms[1], dk), (sv, ctx.geoms[2], dv))):     # read dqkv whole
            dy = dy.contiguous()
            _, ds, db4, dbaft = ops.lsq_bwd(dy, qkv2[:, i * C:], s, b4[i * C:(i + 1) * C], g, dx=dqkv[:, i * C:], amax_word=am,
                                            db4_out=db4_all[i * C:(i + 1) * C])
            res.append((ds, dbaft))
        if am is not None:
            ops.tag_amax(dqkv, am)           # (on the BASE: a consumer's reshape is a view of dqkv, and amax_of looks at t and t._base)
        out = dqkv.view(B, N, 3 * C)
        return (out, db4_all, res[0][0], res[1][0], res[2][0], res[0][1], res[1][1],
                res[2][1], None, None, None)


class QKVSplitLsqCodesFn(torch.autograd.Function):
    """QKVSplitLsqFn with the three outputs as integer codes only (operands of the code GEMMs of the plain attention
    core): q / k per-token steps, v per-channel step, on column slices of the (B*N, 3C) projection.  The fp32 q_hat, k_hat,
    v_hat are never written; zero-stride carriers take the autograd edges.  Backward = QKVSplitLsqFn's."""

    @staticmethod
    def forward(ctx, qkv, b4, sq, sk, sv, baq, bak, bav, gq, gk, gv):
        B, N, C3 = qkv.shape
        C = C3 // 3
        qkv2 = qkv.reshape(B * N, C3)
        codes = []
        for i, (s, g) in enumerate(((sq, gq), (sk, gk), (sv, gv))):
            _, cd = ops.lsq_fwd(qkv2[:, i * C:], s, b4[i * C:(i + 1) * C], None, g, want_codes=True, need_values=False)
            codes.append(cd.view(B, N, C))
        ctx.save_for_backward(qkv2, b4, sq, sk, sv)
        ctx.geoms = (gq, gk, gv)
        ctx.shape = (B, N, C)
        ctx.mark_non_differentiable(*codes)
        ctx.set_materialize_grads(False)
        ph = [ops.placeholder((B, N, C), qkv.device) for _ in range(3)]
        return ph[0], ph[1], ph[2], codes[0], codes[1], codes[2]

    @staticmethod
    def backward(ctx, dq, dk, dv, _c0, _c1, _c2):
        qkv2, b4, sq, sk, sv = ctx.saved_tensors
        B, N, C = ctx.shape
        dqkv = torch.empty_like(qkv2)
        db4_all = torch.empty(3 * C, dtype=torch.float32, device=dqkv.device)       # the three launches write their slice (no cat)
        res = []
        am = ops.amax_out(dqkv.device)       # ONE maximum word for the three column slices: the qkv projection's backward GEMMs
        for i, (s, g, dy) in enumerate(((sq, ctx.geoms[0], dq), (sk, ctx.geoms[1], dk), (sv, ctx.geoms[2], dv))):     # read dqkv whole
            dy = dy.contiguous()
            _, ds, db4, dbaft = ops.lsq_bwd(dy, qkv2[:, i * C:], s, b4[i * C:(i + 1) * C], g, dx=dqkv[:, i * C:], amax_word=am,
                                            db4_out=db4_all[i * C:(i + 1) * C])
            res.append((ds, dbaft))
        if am is not None:
            ops.tag_amax(dqkv, am)           # (on the BASE: a consumer's reshape is a view of dqkv, and amax_of looks at t and t._base)
        out = dqkv.view(B, N, 3 * C)
        return (out, db4_all, res[0][0], res[1][0], res[2][0], res[0][1], res[1][1],
                res[2][1], None, None, None)


_HEAD_MASKS = {}


def _head_mask(H, d, dev):
    """(H, H*d) 0/1 matrix, row h selects the channels of head h (a constant: built once per geometry and device)."""
    key = (H, d, str(dev))
    m = _HEAD_MASKS.get(key)
    if m is None:
        m = _HEAD_MASKS[key] = torch.eye(H, device=dev, dtype=torch.float32).repeat_interleave(d, dim=1)
    return m


class QKScoresCodesFn(torch.autograd.Function):
    """Plain attention scores on the codes: S[b,h,n,m] = q_hat[b,n,hd:hd+d] . k_hat[b,m,hd:hd+d], one exact int8 GEMM
    per (b, h) plus the offset terms in its epilogue; backward = two bf16-split GEMMs per (b, h)."""

    @staticmethod
    def forward(ctx, q, k, aux):
        B, N, C = q.shape
        H = aux["H"]
        d = C // H
        Np = pad16(N)
        dev = q.device
        # per-head offset vectors as H rows of length C with zeros outside the head's channels: the per-head dot products
        # of the codes with the other operand's offsets are plain row dots against them
        eye = _head_mask(H, d, dev)                                                           # (H, C), cached
        u = ops.rowdot_i8_multi(aux["qcodes"].view(B * N, C), eye * aux["bk"])                 # [B*N, H]: qq . bk|head
        tq = ops.rowdot_i8_multi(aux["kcodes"].view(B * N, C), eye * aux["bq"])                # [B*N, H]: bq|head . qk
        z = (aux["bq"] * aux["bk"]).view(H, d).sum(1)
        S = ops.qattn_scores_plain(aux["qcodes"], aux["kcodes"], aux["sq"], aux["gq"], aux["sk"], aux["gk"], u, tq, z,
                                   B, H, N, d, Np)
        ctx.aux = aux
        ctx.dims = (B, H, N, d, Np)
        return S

    @staticmethod
    def backward(ctx, dS):
        aux = ctx.aux
        B, H, N, d, Np = ctx.dims
        dS = dS.contiguous()
        dq = ops.qattn_dq_plain(dS, aux["kcodes"], aux["sk"], aux["gk"], B, H, N, d, Np)
        dk = ops.qattn_dk_plain(dS, aux["qcodes"], aux["sq"], aux["gq"], aux["bq"], B, H, N, d, Np)
        rs = aux["link"].pop("ds_rowsum", None)
        if KEEP_ZERO_ROWSUM_TERM:
            # + rowsum_m(dS)[b,h,n] * bk[hd+c]: rows of a softmax backward sum to zero (see QKRScoresCodesFn.backward)
            if rs is None:
                rs = dS[..., :N].sum(-1).reshape(-1)
            dq.view(B, N, H, d).add_(rs.view(B, H, N, 1).permute(0, 2, 1, 3) * aux["bk"].view(1, 1, H, d))
        return dq, dk, None


# =====================================================================================================
# QKR attention core on the integer codes (same functions as QKRScoresFn / SoftmaxLsqFn / PVFn above)
# =====================================================================================================
def pad16(n):
    return (n + 15) // 16 * 16


class QKRScoresCodesFn(torch.autograd.Function):
    """S[b,h,n,m] = x_hat[b,n,:] . qkx_hat[b,m,h,:] with x_hat = ax*qx + bax, qkx_hat = aq*qq + baq: one exact int8
    GEMM per (b,h) plus three small offset terms applied in its epilogue; backward = two bf16-split GEMMs."""

    @staticmethod
    def forward(ctx, xq, qkx, aux):
        B, N, C = xq.shape
        H = aux["H"]
        Np = pad16(N)
        baq2 = aux["baq"].view(H, C)
        u = ops.rowdot_i8_multi(aux["xcodes"].view(B * N, C), baq2)          # [B*N, H]
        tq = ops.rowdot_i8(aux["qcodes"].view(B * N * H, C), aux["bax"])      # [B*N*H]
        z = torch.mv(baq2, aux["bax"])                                        # [H]
        S = ops.qattn_scores(aux["xcodes"], aux["qcodes"], aux["sx"], aux["gx"], aux["sq"], aux["gq"], u, tq, z, B, H, N, C, Np)
        ctx.aux = aux
        ctx.dims = (B, H, N, C, Np)
        return S

    @staticmethod
    def backward(ctx, dS):
        aux = ctx.aux
        B, H, N, C, Np = ctx.dims
        dS = dS.contiguous()
        dqkx = ops.qattn_dqkx(dS, aux["xcodes"], aux["sx"], aux["gx"], aux["bax"], B, H, N, C, Np)
        dxq, accumulate, ret = _shared_grad(aux.get("xgrad_acc"), (B, N, C), dS.device)
        ops.qattn_dxq(dS, aux["qcodes"], aux["sq"], aux["gq"], B, H, N, C, Np, out=dxq, accumulate=accumulate)
        rs = aux["link"].pop("ds_rowsum", None)
        if KEEP_ZERO_ROWSUM_TERM:
            # + sum_h rowsum_m(dS)[b,h,n] * baq[h,c]: the offset move_qkx_aft enters x_hat . (q + baq)^T, so dx_hat gets
            # rowsum(dS) * baq.  dS is a softmax backward, whose rows sum to zero: the term is fp32 rounding residue
            # (~1e-7 of dx_hat, 1e-3 is the parity tolerance) and cost a K=6 GEMM over the 39 MB buffer per block.
            if rs is None:
                rs = dS[..., :N].sum(-1).reshape(-1)
            dxq.view(B * N, C).addmm_(rs.view(B, H, N).permute(0, 2, 1).reshape(B * N, H), aux["baq"].view(H, C))
        return ret, dqkx, None


class SoftmaxLsqCodesFn(torch.autograd.Function):
    """SoftmaxLsqFn that also emits the uint8 codes of P_hat and their row sums (operands of the int8 P.V GEMM)."""

    @staticmethod
    def forward(ctx, S, s, N, alpha, hi, link, addend=None):
        B, H = S.shape[0], S.shape[1]
        Np = S.shape[3]
        rows = B * H * N
        prob, y, codes, rsum = ops.softmax_lsq_fwd(S, s, rows, N, Np, N, alpha, hi, B * H * N, want_codes=True,
                                                   need_values=False, addend=addend)
        ctx.save_for_backward(prob, s)
        ctx.addend = addend
        ctx.meta = (rows, N, Np, alpha, hi, B * H * N)
        ctx.link = link
        ctx.mark_non_differentiable(codes, rsum)
        ctx.set_materialize_grads(False)          # no zero-filled "gradients" for the code outputs
        return y, codes, rsum

    @staticmethod
    def backward(ctx, g, _gc, _gr):
        if g is None:
            return None, None, None, None, None, None, None
        prob, s = ctx.saved_tensors
        rows, N, Np, alpha, hi, M = ctx.meta
        g = g.contiguous()
        dS, ds, rs = ops.softmax_lsq_bwd(g, prob, s, rows, N, Np, N, alpha, hi, M, inplace=True, want_rowsum=True)
        ctx.link["ds_rowsum"] = rs
        dadd = _addend_grad(dS, ctx.addend, alpha) if (ctx.addend is not None and ctx.needs_input_grad[6]) else None
        return dS, ds, None, None, None, None, dadd


FUSE_SCORES_SOFTMAX = os.environ.get("OFQ_NO_SCORES_SOFTMAX_FUSE") is None
FUSE_DP_SOFTMAX_BWD = os.environ.get("OFQ_NO_DP_SOFTMAX_FUSE") is None          # A/B switch
ATTN_PREP = os.environ.get("OFQ_NO_ATTN_PREP") is None          # u, tq, v^T in one launch (A/B switch)


class ScoresSoftmaxCodesFn(torch.autograd.Function):
    """QKRScoresCodesFn / QKScoresCodesFn followed by SoftmaxLsqCodesFn as ONE forward kernel (ofq_qattn_scores_softmax_i8):
    the score matrix stays in LDS between the int8 GEMM and the softmax, only prob (for the backward), the P codes and their
    row sums are written.  The backward is the unfused pair's: softmax-LSQ backward on the saved prob, then the two
    bf16-split GEMMs of the scores.  aux["plain"]: plain attention (q / k codes per head) instead of QKR (x / qkx codes)."""

    @staticmethod
    def forward(ctx, a_carrier, b_carrier, s, aux, addend=None):
        plain = aux["plain"]
        H = aux["H"]
        if plain:
            B, N, C = a_carrier.shape
            CK = C // H
            pre = aux.get("plain_pre")
            if pre is not None:                     # (all blocks' offset operands in one batch: all_plain_prep)
                ebk, ebq, z = pre
            else:
                eye = _head_mask(H, CK, a_carrier.device)
                ebk, ebq = eye * aux["bk"], eye * aux["bq"]
                z = (aux["bq"] * aux["bk"]).view(H, CK).sum(1)
            u = ops.rowdot_i8_multi(aux["qcodes"].view(B * N, C), ebk)
            tq = ops.rowdot_i8_multi(aux["kcodes"].view(B * N, C), ebq)
            ac, bc, sa, ga, sb, gb = aux["qcodes"], aux["kcodes"], aux["sq"], aux["gq"], aux["sk"], aux["gk"]
        else:
            B, N, C = a_carrier.shape
            CK = C
            baq2 = aux["baq"].view(H, C)
            vlink = aux.get("vlink")
            if ATTN_PREP and vlink is not None and C <= 512 and baq2.is_contiguous() and aux["bax"].is_contiguous():
                # u, tq and the transposed v codes (for the P.V GEMM that follows) in one launch
                u, tq, vlink["vT"], z = ops.qattn_prep(aux["xcodes"], baq2, aux["qcodes"], aux["bax"], vlink["vcodes"], B, H, N,
                                                       C, pad16(N))
            else:
                u = ops.rowdot_i8_multi(aux["xcodes"].view(B * N, C), baq2)
                tq = ops.rowdot_i8(aux["qcodes"].view(B * N * H, C), aux["bax"])
                z = torch.mv(baq2, aux["bax"])
            ac, bc, sa, ga, sb, gb = aux["xcodes"], aux["qcodes"], aux["sx"], aux["gx"], aux["sq"], aux["gq"]
        Np = pad16(N)
        prob, codes, rsum = ops.qattn_scores_softmax(ac, bc, sa, ga, sb, gb, u, tq, z, plain, s, aux["alpha"], aux["hi"],
                                                     B, H, N, CK, Np, addend=addend)
        ctx.save_for_backward(prob, s)
        ctx.addend = addend
        ctx.aux = aux
        ctx.sum_leaf = s
        ctx.dims = (B, H, N, C, Np)
        ctx.mark_non_differentiable(codes, rsum)
        ctx.set_materialize_grads(False)
        vl = aux.get("vlink")
        if vl is not None:
            # the P.V product that consumes these codes hands its dO over instead of a materialised dP (PVCodesFn.backward)
            vl["fuse_dp"] = FUSE_DP_SOFTMAX_BWD and 64 < N <= 256 and Np <= 256 and (C // H) % 16 == 0 and C % 4 == 0
        return ops.placeholder((B, H, N, Np), prob.device), codes, rsum

    @staticmethod
    def backward(ctx, g, _gc, _gr):
        if g is None:
            return None, None, None, None, None
        prob, s = ctx.saved_tensors
        aux = ctx.aux
        B, H, N, C, Np = ctx.dims
        rows, alpha, hi = B * H * N, aux["alpha"], aux["hi"]
        vl = aux.get("vlink")
        dO = vl.pop("dO_for_dp", None) if vl is not None else None
        with sum_scope(ctx.sum_leaf):
            if dO is not None:
                dS, ds, rs = ops.qattn_dp_softmax_bwd(dO, vl["vcodes"], vl["sv"], vl["gv"], vl["bav"], prob, s, alpha, hi, B, H, N,
                                                      dO.shape[-1] // H, Np, want_rowsum=KEEP_ZERO_ROWSUM_TERM)
            else:
                g = g.contiguous()
                dS, ds, rs = ops.softmax_lsq_bwd(g, prob, s, rows, N, Np, N, alpha, hi, rows, inplace=True, want_rowsum=True)
        dadd = _addend_grad(dS, ctx.addend, alpha) if (ctx.addend is not None and ctx.needs_input_grad[4]) else None
        if aux["plain"]:
            d = C // H
            da = ops.qattn_dq_plain(dS, aux["kcodes"], aux["sk"], aux["gk"], B, H, N, d, Np)
            db = ops.qattn_dk_plain(dS, aux["qcodes"], aux["sq"], aux["gq"], aux["bq"], B, H, N, d, Np)
            if KEEP_ZERO_ROWSUM_TERM:
                da.view(B, N, H, d).add_(rs.view(B, H, N, 1).permute(0, 2, 1, 3) * aux["bk"].view(1, 1, H, d))
            return da, db, ds, None, dadd
        db = ops.qattn_dqkx(dS, aux["xcodes"], aux["sx"], aux["gx"], aux["bax"], B, H, N, C, Np)
        dxq, accumulate, ret = _shared_grad(aux.get("xgrad_acc"), (B, N, C), dS.device)
        ops.qattn_dxq(dS, aux["qcodes"], aux["sq"], aux["gq"], B, H, N, C, Np, out=dxq, accumulate=accumulate)
        if KEEP_ZERO_ROWSUM_TERM:
            dxq.view(B * N, C).addmm_(rs.view(B, H, N).permute(0, 2, 1).reshape(B * N, H), aux["baq"].view(H, C))
        return ret, db, ds, None, dadd


def scores_softmax_fusable(N):
    """The fused kernel holds a 64-query x 256-key score panel (every key of a row must fit: N <= 256), or a 64 x 64 one
    for the 49-token Swin windows (one window and head per workgroup)."""
    return FUSE_SCORES_SOFTMAX and N <= 256 and pad16(N) <= 256 and (N > 64 or not NO_WINDOW_SCORES_SOFTMAX)


NO_WINDOW_SCORES_SOFTMAX = os.environ.get("OFQ_NO_WINDOW_SCORES_SOFTMAX") is not None      # A/B switch


class PVCodesFn(torch.autograd.Function):
    """O = P_hat @ V_hat per head on the codes: P_hat = ap*qp, V_hat = av*qv + bav."""

    @staticmethod
    def forward(ctx, P, v, aux):
        B, H, N, Np = P.shape
        C = v.shape[2]
        d = C // H
        vT = aux.pop("vT", None)                    # prepared together with u / tq (ScoresSoftmaxCodesFn) ...
        if vT is None:
            vT = ops.codes_transpose_i8(aux["vcodes"].view(B, N, C), Np)
        O = ops.qattn_pv(aux["pcodes"], vT, aux["sp"], aux["gp"], aux["sv"], aux["gv"], aux["bav"], aux["rp"], B, H, N, d, Np)
        ctx.aux = aux
        ctx.dims = (B, H, N, d, Np)
        return O

    @staticmethod
    def backward(ctx, dO):
        aux = ctx.aux
        B, H, N, d, Np = ctx.dims
        C = H * d
        dO = dO.contiguous()
        dV = ops.qattn_dv(dO, aux["pcodes"], aux["sp"], aux["gp"], B, H, N, d, Np)
        if aux.get("fuse_dp"):
            # dP is not materialised: the backward of the softmax quantiser (ScoresSoftmaxCodesFn.backward, the only
            # consumer of this gradient) computes it tile by tile from dO inside ofq_qattn_dp_softmax_bwd
            aux["dO_for_dp"] = dO
            return ops.placeholder((B, H, N, Np), dO.device), dV, None
        w = ops.rowdot_f32_seg(dO.view(B * N, C), aux["bav"], H, d)
        dP = ops.qattn_dp(dO, aux["vcodes"], aux["sv"], aux["gv"], w, B, H, N, d, Np)
        return dP, dV, None


# =====================================================================================================
# LayerNorm (Block.norm1 / norm2, deit_vision_transformer.py:91-102), optionally fused with the residual add before it
class LayerNormFn(torch.autograd.Function):
    """y = LayerNorm(x) over the last dimension (nn.LayerNorm semantics, biased variance)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        shp = x.shape
        x2d = x.reshape(-1, shp[-1])
        if x2d.stride(-1) != 1 or (x2d.stride(0) & 3):
            x2d = x2d.contiguous()
        y, _, mean, rstd = ops.layernorm_fwd(x2d, weight, bias, eps)
        ctx.save_for_backward(x2d, mean, rstd, weight)
        ctx.shape = shp
        ctx.affine = weight is not None
        ctx.sum_leaves = (weight, bias)
        return y.view(shp)

    @staticmethod
    def backward(ctx, dy):
        x2d, mean, rstd, weight = ctx.saved_tensors
        dy2d = dy.reshape(x2d.shape)
        if dy2d.stride(-1) != 1 or (dy2d.stride(0) & 3):
            dy2d = dy2d.contiguous()
        with sum_scope(*ctx.sum_leaves):
            dx, dg, db = ops.layernorm_bwd(dy2d, x2d, mean, rstd, weight, want_affine_grads=ctx.affine)
        return dx.view(ctx.shape), dg, db, None


class AddLayerNormFn(torch.autograd.Function):
    """(xs, y) = (x + res, LayerNorm(x + res)): the residual add of a block fused with the norm that follows it; the
    backward adds the gradient arriving on xs (the residual stream) to the LayerNorm input gradient in the same pass."""

    @staticmethod
    def forward(ctx, x, res, weight, bias, eps):
        shp = x.shape
        x2d = x.reshape(-1, shp[-1]).contiguous()
        r2d = res.reshape(-1, shp[-1]).contiguous()
        y, xs, mean, rstd = ops.layernorm_fwd(x2d, weight, bias, eps, res2d=r2d)
        ctx.save_for_backward(xs, mean, rstd, weight)
        ctx.shape = shp
        ctx.affine = weight is not None
        ctx.sum_leaves = (weight, bias)
        ctx.set_materialize_grads(False)
        return xs.view(shp), y.view(shp)

    @staticmethod
    def backward(ctx, dxs, dy):
        xs, mean, rstd, weight = ctx.saved_tensors
        if dy is None:
            return dxs, dxs, None, None, None
        dy2d = dy.reshape(xs.shape).contiguous()
        dres = None if dxs is None else dxs.reshape(xs.shape).contiguous()
        with sum_scope(*ctx.sum_leaves):
            dx, dg, db = ops.layernorm_bwd(dy2d, xs, mean, rstd, weight, dres2d=dres, want_affine_grads=ctx.affine)
        dx = dx.view(ctx.shape)
        return dx, dx, dg, db, None


def layer_norm(norm, x):
    """nn.LayerNorm module -> the HIP LayerNorm on device tensors (stock op for the CPU-side fp32 skeleton tests)."""
    if isinstance(norm, torch.nn.LayerNorm) and x.is_cuda and len(norm.normalized_shape) == 1 \
            and norm.normalized_shape[0] % 4 == 0 and norm.normalized_shape[0] <= 2048 and x.dtype == torch.float32:
        return LayerNormFn.apply(x, norm.weight, norm.bias, norm.eps)
    return norm(x)


def add_layer_norm(norm, x, res):
    """(x + res, norm(x + res)) in one pass."""
    if isinstance(norm, torch.nn.LayerNorm) and x.is_cuda and len(norm.normalized_shape) == 1 \
            and norm.normalized_shape[0] % 4 == 0 and norm.normalized_shape[0] <= 2048 and x.dtype == torch.float32:
        return AddLayerNormFn.apply(x, res, norm.weight, norm.bias, norm.eps)
    xs = x + res
    return xs, norm(xs)


class NormQuantFn(torch.autograd.Function):
    """(xs, x_hat carrier, codes) = LayerNorm(x [+ res]) followed by the per-token LSQ of its only consumer, in one kernel
    each way (csrc/layernorm.hip, template flag Q): the LayerNorm output is never written, the backward recomputes it
    from (xs, mean, rstd) and runs the quantiser's backward in front of the LayerNorm backward."""

    @staticmethod
    def forward(ctx, x, res, weight, bias, eps, s, b4, baft, geom, q_perm=None, res_perm=None, qshape=None):
        """q_perm / res_perm / qshape (round 6, Swin): the quantised form leaves in the token order q_perm describes (shape qshape:
        windows x window tokens x C), and `res` arrives in the order res_perm describes (its own shape) -- the shifted-window
        partition and reverse ride in this pass; x and the first output (x + res) stay in x's order."""
        shp = x.shape
        C = shp[-1]
        x2d = x.reshape(-1, C).contiguous()
        r2d = None if res is None else res.reshape(-1, C).contiguous()
        if res is None:
            res_perm = None
        codes, xs, mean, rstd = ops.layernorm_lsq_fwd(x2d, weight, bias, eps, s, b4, geom, res2d=r2d, q_perm=q_perm, res_perm=res_perm)
        xin = xs if xs is not None else x2d
        ctx.save_for_backward(xin, mean, rstd, weight, bias, s, b4, q_perm, res_perm)
        ctx.geom, ctx.shape, ctx.has_res = geom, shp, res is not None
        ctx.res_shape = None if res is None else res.shape
        ctx.sum_leaves = (weight, bias, s, b4, baft)
        ctx.mark_non_differentiable(codes)
        ctx.set_materialize_grads(False)
        first = xs.view(shp) if xs is not None else ops.placeholder(shp, x.device)
        return first, ops.placeholder(tuple(qshape) if qshape is not None else shp, x.device), codes

    @staticmethod
    def backward(ctx, dxs, dxq, _gc):
        xin, mean, rstd, weight, bias, s, b4, q_perm, res_perm = ctx.saved_tensors
        shp = ctx.shape
        C = shp[-1]
        if not ctx.has_res:
            dxs = None
        if dxq is None:
            if dxs is not None and res_perm is not None:
                raise RuntimeError("ofq_amd NormQuantFn: a permuted residual operand needs the quantised output's gradient")
            return dxs, dxs, None, None, None, None, None, None, None, None, None, None
        gq = dxq.reshape(-1, C).contiguous()
        dres = None if dxs is None else dxs.reshape(-1, C).contiguous()
        with sum_scope(*ctx.sum_leaves):
            if q_perm is None and res_perm is None:
                dx, dg, db, db4, ds, dba = ops.layernorm_lsq_bwd(gq, xin, mean, rstd, weight, bias, s, b4, ctx.geom, dres2d=dres)
                dx2 = dx
            else:
                dx, dg, db, db4, ds, dba, dx2 = ops.layernorm_lsq_bwd(gq, xin, mean, rstd, weight, bias, s, b4, ctx.geom,
                                                                      dres2d=dres, q_perm=q_perm,
                                                                      res_perm=res_perm if ctx.has_res else None)
                if dx2 is None:
                    dx2 = dx
        dx = dx.view(shp)
        dres_out = None
        if ctx.has_res:
            dres_out = dx if dx2 is dx or dx2.data_ptr() == dx.data_ptr() else dx2.view(ctx.res_shape)
        return (dx, dres_out, dg, (db if bias is not None else None), None, ds, db4, dba, None, None, None, None)


def norm_quant(norm, spec, x, res=None, q_perm=None, res_perm=None, qshape=None):
    """spec: {"quant": LsqQuantizer, "b4": Parameter, "baft": Parameter} of the only consumer of norm(x [+ res]).
    Returns (x [+ res], (x_hat carrier, codes, geom)) or None when the fused kernel does not apply.
    q_perm / qshape: the consumer sees the tokens of every image permuted (token t at row q_perm[t]) in the shape qshape, and its
    quantiser's geometry is qshape's; res_perm: `res` is given in such an order (NormQuantFn)."""
    qz = spec["quant"]
    C = x.shape[-1]
    if not (isinstance(norm, torch.nn.LayerNorm) and x.is_cuda and x.dtype == torch.float32 and x.dim() in (3, 4) and
            len(norm.normalized_shape) == 1 and C % 4 == 0 and C <= 2048 and qz.initialized_alpha and qz.s is not None):
        return None
    geom = qz._geom(tuple(qshape) if qshape is not None else tuple(x.shape), spec["b4"].numel(), 0, None, None)
    if geom.mode != 0 or geom.bias_len != geom.inner or geom.lo < -128 or geom.hi > 127:
        return None
    if q_perm is not None and q_perm.numel() % geom.S:
        return None
    xs, xq, codes = NormQuantFn.apply(x, res, norm.weight, norm.bias, norm.eps, qz.s, spec["b4"], spec["baft"], geom,
                                      q_perm, res_perm, qshape)
    return (xs if res is not None else x), (xq, codes, geom)


class KDLossFn(torch.autograd.Function):
    """KDLossSoftandHard (src/quantization/utils.py:59-77) on (cls logits, dist logits, teacher logits, labels): value and both
    gradients in one pass over the rows (ofq_kd_loss_fwd), the backward scales the saved gradients by the incoming one."""

    @staticmethod
    def forward(ctx, cls_out, dist_out, soft_target, hard_target):
        loss, dcls, ddist, cls_scale = ops.kd_loss_fwd(cls_out, dist_out, soft_target, hard_target)
        ctx.save_for_backward(dcls, ddist, cls_scale)
        return loss

    @staticmethod
    def backward(ctx, g):
        dcls, ddist, cls_scale = ctx.saved_tensors
        oc, od = ops.kd_loss_bwd(g.contiguous(), dcls, ddist, cls_scale)
        return oc, od, None, None


def kd_loss_fusable(cls_out, dist_out, soft_target, hard_target):
    ok = lambda t: torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1   # noqa: E731
    return (ok(cls_out) and ok(dist_out) and ok(soft_target) and cls_out.shape == dist_out.shape == soft_target.shape
            and torch.is_tensor(hard_target) and hard_target.is_cuda and hard_target.dtype == torch.int64 and hard_target.dim() == 1
            and hard_target.is_contiguous() and hard_target.shape[0] == cls_out.shape[0] and not soft_target.requires_grad)


class AssembleTokensFn(torch.autograd.Function):
    """cat(cls_token, [dist_token,] patches) + pos_embed (deit.py:32-44) in one launch.  Backward: the patches' gradient is a
    view, pos_embed's is ONE column sum over the batch, and its first rows are the class / distillation tokens' gradients."""

    @staticmethod
    def forward(ctx, patches, cls_token, dist_token, pos):
        ctx.ntok = 1 if dist_token is None else 2
        return ops.assemble_tokens(patches.contiguous(), cls_token, dist_token, pos)

    @staticmethod
    def backward(ctx, g):
        B, T, C = g.shape
        g = g.contiguous()
        dpos = ops.colsum(g.view(B, T * C)).view(1, T, C)
        dcls = dpos[:, 0:1]
        ddist = dpos[:, 1:2] if ctx.ntok == 2 else None
        return g[:, ctx.ntok:], dcls, ddist, dpos
