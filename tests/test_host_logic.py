"""CPU tests of the host side that mirrors the reference's module API: construction, parameter / state-dict names
(= checkpoint format), model surgery by name list, optimizer groups, CGA module filter, geometry
of the LSQ kernel launches, and the loud failure on CPU tensors (there is no CPU fallback)."""
from functools import partial
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.nn as nn

from util import load_golden, group, case_names


def _tiny(depth=2, dim=32, heads=2, qkr=True, wb=2, ab=2, ncls=10):
    from ofq_amd import engine
    from ofq_amd.deit import DistilledVisionTransformer
    model = DistilledVisionTransformer(img_size=224, patch_size=16, embed_dim=dim, depth=depth, num_heads=heads,
                                       mlp_ratio=2, qkv_bias=True, num_classes=ncls,
                                       norm_layer=partial(nn.LayerNorm, eps=1e-6), act_layer=nn.GELU)
    args = SimpleNamespace(qmodules=engine.default_qmodules(depth), wq_mode="statsq", wq_enable=True, wq_bitw=wb,
                           aq_enable=True, aq_mode="lsq", aq_bitw=ab, wq_per_channel=True, aq_per_channel=True,
                           model_type="deit", pretrained_initialized=True, qk_reparam=qkr, qk_reparam_type=0)
    return engine.get_qat_model(model, args)


def _fake_init_lsq(model, B=2, N=198):
    """Create every lazily-initialised LSQ `s` with the shape the first forward would give it (no GPU needed)."""
    from ofq_amd.quantization.quantizer import lsq as L
    from ofq_amd.quantization.modules.attention import QAttention, QAttention_qkreparam
    from ofq_amd.quantization.modules.qlinear import QLinear, LSQ_QConv2d, LSQ_QLinear4head, LSQ_input
    for name, m in model.named_modules():
        if isinstance(m, (QLinear, LSQ_input)):
            m.input_quant_fn.s = nn.Parameter(torch.ones(N))
        if isinstance(m, LSQ_QConv2d):
            m.input_quant_fn.s = nn.Parameter(torch.ones(3))
            m.lsqw_fn.s = nn.Parameter(torch.ones(m.out_channels))
        if isinstance(m, LSQ_QLinear4head):
            m.input_quant_fn.s = nn.Parameter(torch.ones(1))
            m.lsqw_fn.s = nn.Parameter(torch.ones(m.out_features))
        if isinstance(m, QAttention):
            C = m.proj.in_features
            m.quan_a_q_fn.s = nn.Parameter(torch.ones(N))
            m.quan_a_k_fn.s = nn.Parameter(torch.ones(N))
            m.quan_a_v_fn.s = nn.Parameter(torch.ones(C))
            m.quan_a_softmax_fn.s = nn.Parameter(torch.ones(N))
        if isinstance(m, QAttention_qkreparam):
            C = m.proj.in_features
            m.quan_a_qkx_fn.s = nn.Parameter(torch.ones(N * m.num_heads))
            m.quan_a_v_fn.s = nn.Parameter(torch.ones(C))
            m.quan_a_softmax_fn.s = nn.Parameter(torch.ones(N))


@pytest.mark.parametrize("case", ["plain_w4a4", "qkr_w2a2"])
def test_state_dict_matches_the_reference_checkpoint_format(case):
    g = group(load_golden("g7_tiny_deit"), case)
    B, depth, dim, heads, wb, ab, qkr, seed, ncls, mlp_ratio = [int(v) for v in g["meta"]]
    model = _tiny(depth, dim, heads, bool(qkr), wb, ab, ncls)
    _fake_init_lsq(model)
    ref = {k[2:]: v.shape for k, v in g.items() if k.startswith("p:")}
    mine = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert set(mine) == set(ref), (sorted(set(ref) - set(mine)), sorted(set(mine) - set(ref)))
    for k in ref:
        assert tuple(ref[k]) == mine[k], k
    sd = {k[2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("p:")}
    model.load_state_dict(sd, strict=True)


def test_module_state_dicts_match_reference_goldens():
    from ofq_amd.quantization.modules.qlinear import QLinear, QMLP, LSQ_QConv2d, LSQ_QLinear4head
    from ofq_amd.quantization.modules.attention import QAttention, QAttention_qkreparam, QAttention_qkreparam_4_cga
    from ofq_amd.deit_vision_transformer import Attention, Mlp
    d = load_golden("g4_attention")
    kinds = {"plain": QAttention, "qkr": QAttention_qkreparam, "qkrcga": QAttention_qkreparam_4_cga}
    for nme in case_names(d):
        g = group(d, nme)
        B, N, C, H, wb, ab, seed = [int(v) for v in g["meta"]]
        q = kinds[nme.split("_")[0]](m=Attention(dim=C, num_heads=H, qkv_bias=True), weight_bits=wb, input_bits=ab,
                                     pretrained_initialized=True)
        _fake_init_lsq(q, N=N)
        assert set(q.state_dict()) == {k[2:] for k in g if k.startswith("p:")}, nme
        assert not hasattr(q, "qkv") or nme.startswith("plain")
    d = load_golden("g5_qmlp")
    g = group(d, case_names(d)[0])
    q = QMLP(m=Mlp(in_features=16, hidden_features=64, act_layer=nn.GELU), weight_bits=2, input_bits=2,
             act_layer=nn.GELU, pretrained_initialized=True)
    _fake_init_lsq(q, N=5)
    assert set(q.state_dict()) == {k[2:] for k in g if k.startswith("p:")}
    assert q.fc2.input_quant_fn.all_positive and not q.fc1.input_quant_fn.all_positive      # qlinear.py:118-120


def test_surgery_replaces_by_name_and_forces_w8a8_on_stem_and_heads():
    from ofq_amd.quantization.modules.qlinear import LSQ_QConv2d, LSQ_QLinear4head, QMLP, QConv2d
    from ofq_amd.quantization.modules.attention import QAttention_qkreparam, QAttention
    m = _tiny(qkr=True)
    assert isinstance(m.patch_embed.proj, LSQ_QConv2d) and m.patch_embed.proj.input_bits == 8
    assert QConv2d is LSQ_QConv2d
    assert isinstance(m.head, LSQ_QLinear4head) and isinstance(m.head_dist, LSQ_QLinear4head)
    assert all(isinstance(b.attn, QAttention_qkreparam) and isinstance(b.mlp, QMLP) for b in m.blocks)
    assert isinstance(_tiny(qkr=False).blocks[0].attn, QAttention)
    assert m.blocks[0].mlp.fc1.weight_bits == 2 and m.blocks[0].attn.proj.input_bits == 2
    assert "act_bit=2" in m.blocks[0].mlp.fc1.extra_repr()


def test_error_behaviour_mirrors_the_reference():
    from ofq_amd.quantization.modules.qlinear import QLinear
    from ofq_amd.quantization.modules.attention import QAttention
    with pytest.raises(ValueError, match="Unknown quant_method"):
        QLinear(m=nn.Linear(8, 8), weight_quant_method="foo")                     # qlinear.py:53
    with pytest.raises(AssertionError):
        QAttention(m=nn.Linear(8, 8))                                             # attention.py:17
    q = QLinear(m=nn.Linear(16, 8), weight_bits=2, input_bits=2, pretrained_initialized=True)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        q(torch.zeros(2, 3, 16))


def test_optimizer_groups_follow_timm_no_weight_decay_rule():
    from ofq_amd import engine
    m = _tiny()
    _fake_init_lsq(m)
    groups = engine.param_groups_weight_decay(m, 0.05)
    no_decay = {id(p) for p in groups[0]["params"]}
    named = dict(m.named_parameters())
    for n in ("pos_embed", "cls_token", "dist_token", "blocks.0.mlp.fc1.bias", "blocks.0.mlp.fc1.input_quant_fn.s",
              "blocks.0.attn.move_qkx_b4.bias", "blocks.0.norm1.weight"):
        assert id(named[n]) in no_decay, n
    for n in ("blocks.0.mlp.fc1.weight", "blocks.0.attn.q.weight", "patch_embed.proj.weight", "head.weight"):
        assert id(named[n]) not in no_decay, n
    assert all(not p.requires_grad for n, p in named.items() if n.endswith("clip_val"))


def test_cga_module_filter_matches_cga_py():
    from ofq_amd import engine
    names = [k for k, _ in engine.cga_modules(_tiny(qkr=True), qk_reparam=True)]
    assert sorted(names) == sorted("blocks.%d.%s" % (i, s) for i in range(2) for s in ("attn.v", "attn.proj", "mlp.fc1", "mlp.fc2"))
    names = [k for k, _ in engine.cga_modules(_tiny(qkr=False), qk_reparam=False)]
    assert sorted(names) == sorted("blocks.%d.%s" % (i, s) for i in range(2) for s in ("attn.qkv", "attn.proj", "mlp.fc1", "mlp.fc2"))


def test_lsq_geometry_and_gradient_scale():
    from ofq_amd import ops
    from ofq_amd.quantization.quantizer.lsq import LsqQuantizer, LsqQuantizer4v, LsqQuantizer4img, LsqQuantizer4head_input
    q = LsqQuantizer(bit=2)
    g = q._geom((128, 198, 384), 384, 0, None, None)
    assert (g.outer, g.S, g.inner, g.mode, g.lo, g.hi) == (128, 198, 384, 0, -2, 1)
    assert abs(g.gscale - 1.0 / np.sqrt(1 * 128 * 384)) < 1e-12                  # lsq.py:584
    g = LsqQuantizer(bit=2, all_positive=True)._geom((128, 6, 198, 198), 0, 0, None, None)
    assert (g.outer, g.S, g.inner, g.lo, g.hi) == (768, 198, 198, 0, 3)
    assert abs(g.gscale - 1.0 / np.sqrt(3 * 128 * 6 * 198)) < 1e-12             # lsq.py:588
    g = LsqQuantizer(bit=2)._geom((128, 198 * 6, 384), 2304, 0, None, None)      # qkx: s per (token, head)
    assert (g.S, g.bias_len) == (1188, 2304)
    g = LsqQuantizer4v(bit=4)._geom((128, 198, 384), 384, 0, None, None)
    assert (g.outer, g.S, g.inner, g.mode) == (128 * 198, 1, 384, 1) and abs(g.gscale - 1 / np.sqrt(7 * 128 * 198)) < 1e-12
    g = LsqQuantizer4head_input(bit=8)._geom((128, 384), 384, 0, None, None)
    assert abs(g.gscale - 1 / np.sqrt(127 * 128 * 384)) < 1e-12                  # lsq.py:494
    qi = LsqQuantizer4img(bit=8)
    qi.thd_neg, qi.thd_pos = -128, 127
    g = qi._geom((128, 3, 224, 224), 50176, 0, None, None)
    assert (g.outer, g.S, g.inner) == (128, 3, 50176)


def test_kd_loss_matches_oracle():
    import ofq_oracle as O
    from ofq_amd.quantization.utils import KDLossSoftandHard
    torch.manual_seed(0)
    c, d, t = torch.randn(4, 10), torch.randn(4, 10), torch.randn(4, 10)
    y = torch.randint(0, 10, (4,))
    assert torch.allclose(KDLossSoftandHard()((c, d), y, t), O.kd_loss_soft_and_hard(c, d, y, t), rtol=1e-6)


def test_fp32_teacher_skeleton_runs_on_cpu_and_matches_shapes():
    from ofq_amd.deit import create_model
    m = create_model("deit_tiny_distilled_patch16_224", num_classes=10, depth=1)
    m.eval()
    with torch.no_grad():
        y, attn = m(torch.randn(1, 3, 224, 224))
    assert y.shape == (1, 10) and len(attn) == 1
    assert m.pos_embed.shape == (1, 198, 192)                                    # cls + dist + 196 (deit.py:23-25)


# ------------------------------------------------------------------------------------------------ input pipeline (host side)
def test_input_pipeline_oracle_on_hand_computed_cases():
    """oracle.input_pipeline restates timm 0.5.4's FastCollateMixup / PrefetchLoader / RandomErasing arithmetic; the
    reference holds no fixtures for it, so it is pinned on cases small enough to compute by hand."""
    import ofq_oracle as O
    x = np.zeros((2, 3, 2, 4), dtype=np.uint8)
    x[0] = 10
    x[1] = 201
    mean, std = (0.5, 0.5, 0.5), (0.25, 0.5, 1.0)
    # no mixing: (10 - 127.5) / 63.75, / 127.5, / 255
    y = O.input_pipeline(x, 1.0, False, (0, 0, 0, 0), None, None, mean, std)
    assert y.shape == (2, 3, 2, 4) and y.dtype == torch.float32
    assert torch.equal(y[0, :, 0, 0], torch.tensor([(10 - 127.5) / 63.75, (10 - 127.5) / 127.5, (10 - 127.5) / 255.0]).float())
    # mixup lam = 0.3: sample 0 <- rint(0.3 * 10 + 0.7 * 201) = rint(143.7) = 144, sample 1 <- rint(0.3 * 201 + 0.7 * 10) = 67
    y = O.input_pipeline(x, 0.3, False, (0, 0, 0, 0), None, None, mean, std)
    assert float(y[0, 2, 0, 0]) == pytest.approx((144 - 127.5) / 255.0) and float(y[1, 2, 1, 3]) == pytest.approx((67 - 127.5) / 255.0)
    # cutmix: the box rows [0,1) x cols [1,3) comes from the mirrored sample, the rest stays
    y = O.input_pipeline(x, 0.75, True, (0, 1, 1, 3), None, None, mean, std)
    got = (y[0, 2] * 255.0 + 127.5).round().int()
    assert got.tolist() == [[10, 201, 201, 10], [10, 10, 10, 10]]
    # erasing: rectangle of sample 1 replaced by the noise values, everything else normalised as before
    noise = np.full((2, 3, 2, 4), 7.0, dtype=np.float32)
    rects = np.array([[0, 0, 0, 0], [1, 2, 1, 2]], dtype=np.int32)
    y = O.input_pipeline(x, 1.0, False, (0, 0, 0, 0), rects, noise, mean, std)
    assert torch.equal(y[1, :, 1, 2:4], torch.full((3, 2), 7.0)) and float(y[1, 0, 0, 0]) == pytest.approx((201 - 127.5) / 63.75)
    assert not bool((y[0] == 7.0).any())
    # soft targets: label smoothing 0.1 over 4 classes, lam 0.3
    t = O.mixup_target(torch.tensor([1, 3]), 4, 0.3, 0.1)
    assert t.shape == (2, 4) and torch.allclose(t.sum(1), torch.ones(2))
    assert float(t[0, 1]) == pytest.approx(0.3 * 0.925 + 0.7 * 0.025) and float(t[0, 3]) == pytest.approx(0.3 * 0.025 + 0.7 * 0.925)


def test_input_pipeline_host_draws_follow_timm_formulas():
    """ofq_amd.data draws the augmentation parameters on the host in timm's order from timm's generators: a seeded run
    gives the values the published formulas give (Mixup._params_per_batch, rand_bbox, RandomErasing._erase)."""
    import random
    from ofq_amd.data import MixupParams, RandomErasingParams
    np.random.seed(7)
    mp = MixupParams(mixup_alpha=0.8, cutmix_alpha=1.0, prob=1.0, switch_prob=0.5, label_smoothing=0.1, num_classes=10)
    draws = [mp.draw(224, 224) for _ in range(8)]
    np.random.seed(7)
    for lam, use_cutmix, box in draws:                      # replay the generator by hand
        assert np.random.rand() < 1.0
        cm = np.random.rand() < 0.5
        lam_mix = np.random.beta(1.0, 1.0) if cm else np.random.beta(0.8, 0.8)
        assert cm == use_cutmix
        if cm:
            ratio = np.sqrt(1 - lam_mix)
            ch, cw = int(224 * ratio), int(224 * ratio)
            cy, cx = np.random.randint(0, 224), np.random.randint(0, 224)
            b = (int(np.clip(cy - ch // 2, 0, 224)), int(np.clip(cy + ch // 2, 0, 224)),
                 int(np.clip(cx - cw // 2, 0, 224)), int(np.clip(cx + cw // 2, 0, 224)))
            assert b == box and lam == pytest.approx(1.0 - (b[1] - b[0]) * (b[3] - b[2]) / (224.0 * 224.0))
        else:
            assert lam == float(lam_mix) and box == (0, 0, 0, 0)
    assert any(d[1] for d in draws) and not all(d[1] for d in draws)
    random.seed(11)
    rects = RandomErasingParams(probability=0.25).draw(256, 224, 224)
    erased = rects[rects[:, 2] > 0]
    assert 0.15 < len(erased) / 256 < 0.35                                   # re_prob 0.25
    area = erased[:, 2] * erased[:, 3] / (224.0 * 224.0)
    assert area.min() > 0.015 and area.max() < 0.36                          # min_area 0.02 .. max_area 1/3 (after rounding)
    assert (erased[:, 0] + erased[:, 2] <= 224).all() and (erased[:, 1] + erased[:, 3] <= 224).all()
    mp.mixup_enabled = False                                                 # train.py:865-869 (mixup_off_epoch)
    assert mp.draw(224, 224) == (1.0, False, (0, 0, 0, 0))


def test_bench_self_launch_ends_every_rank_when_one_fails():
    """`bench.py --gpus N` starts N child ranks itself.  Without a HIP device every child exits with an
    error: the parent must come back promptly with a non-zero code instead of waiting on a rank that will never answer."""
    import subprocess, sys, os, time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, torch; sys.argv=['bench.py','--gpus','2','--steps','1','--warmup','0'];"
            "torch.cuda.device_count=lambda: 2; sys.path.insert(0, %r); import bench; bench.main()" % root)
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, timeout=300,
                       env=dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))
    assert r.returncode != 0
    assert b"HIP device" in r.stderr
    assert b'"metric"' not in r.stdout
    assert time.time() - t0 < 240


def test_grad_slot_hands_out_bucket_slices_only_when_safe():
    """parallel.grad_slot: the bucket slice a dW kernel may write into.  None without a synchronised DataParallel, None
    when the parameter already holds a gradient (a second backward must accumulate), None for a parameter that merely
    reuses the id of a dead one; otherwise a NEW tensor object on the slice's memory (autograd adopts it without a copy)."""
    import torch
    from ofq_amd import parallel
    lin = torch.nn.Linear(4, 3)
    assert parallel.grad_slot(lin.weight) is None and parallel.grad_slot(None) is None
    dp = parallel.DataParallel(lin, broadcast=False)
    assert parallel.grad_slot(lin.weight) is None                       # world size 1, no force_sync: no slots registered
    dp.sync = True
    dp._build_buckets(1.0)
    b = dp._bucket_of[lin.weight]
    view = [v for v, p in zip(b.views, b.params) if p is lin.weight][0]
    a = parallel.grad_slot(lin.weight)
    assert a is not None and a is not view and a.data_ptr() == view.data_ptr() and a.shape == lin.weight.shape
    lin.weight.grad = torch.zeros_like(lin.weight)
    assert parallel.grad_slot(lin.weight) is None
    lin.weight.grad = None
    stale = parallel._GRAD_SLOTS[id(lin.weight)]
    other = torch.nn.Parameter(torch.zeros(3, 4))
    parallel._GRAD_SLOTS[id(other)] = stale                             # an id that now belongs to another parameter
    assert parallel.grad_slot(other) is None
    del parallel._GRAD_SLOTS[id(other)]
    for h in dp._hooks:
        h.remove()



def test_amax_word_is_dropped_when_the_tensor_is_modified_in_place():
    """ops.tag_amax / amax_of (the power-of-two scale of the two-plane backward GEMMs): autograd's input buffers add a second
    gradient in place (`old.add_(new)`) when they hold the only reference, and the Python object -- with the producer's maximum
    word on it -- survives; a too-small maximum would overflow the fp16 planes silently.  The tag carries the version
    counter and address it was valid for."""
    from ofq_amd import ops
    t = torch.ones(4, 8)
    word = torch.zeros(1, dtype=torch.int32)
    ops.tag_amax(t, word)
    assert ops.amax_of(t) is word
    assert ops.amax_of(t.view(2, 16)) is word            # a reshape in between keeps the word (views share the version counter)
    v = t.view(32)
    t.add_(torch.full((4, 8), 100.0))                    # what InputBuffer::add does to a gradient it owns
    assert ops.amax_of(t) is None and ops.amax_of(v) is None and ops.amax_of(t.view(2, 16)) is None
    ops.tag_amax(t, word)
    assert ops.amax_of(t) is word
    t.set_(torch.zeros(4, 8))                            # other memory under the same object
    assert ops.amax_of(t) is None
