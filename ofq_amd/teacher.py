"""Forward of the fp32 KD teacher through the HIP kernels (reference train.py:428-442 creates the
teacher with timm's create_model, train.py:906-910 runs `soft_target, _ = teacher(input)` in every step).

The teacher is the plain fp32 DistilledVisionTransformer of src/deit.py:19-67 / src/deit_vision_transformer.py:85-164: no
quantisers, no gradients needed (nothing optimises it; the reference leaves it trainable and back-propagates into it for
nothing).  `HipTeacher` runs that forward on the same kernels as the student's fp32 pieces: exact-fp32 MFMA GEMMs
(ofq_gemm_f32: linear layers with bias, strided-batched scores and P.V on column slices of the qkv projection, no head
permutes), the LayerNorm kernel with the residual add fused in, the softmax kernel's probabilities, a float4 erf-GELU.
Mode follows the wrapped module (the reference never calls .eval() on the teacher): training mode returns
((cls, dist), None), eval mode the averaged logits."""
import torch

from . import ops
from .functional import pad4


class HipTeacher:
    """gemm: "f16x4" (round 5) -- weights as two fp16 planes of W 2^E, activations as two fp16 planes of x 2^E' (E' from the
    maximum word of the kernel that produced x): y = [x | x] . [Wh | Wl]^T as one two-segment code GEMM, four plane products
    (error 2^-22-grade: tests hold it to the 1e-4 of the nine-product mode); "f32" -- the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32,
    157 TFLOP/s peak); "bf16x9" / "bf16x6" -- the four
    linear layers of every block on the bf16 matrix cores: the frozen weights are split ONCE into three bf16 planes
    (W = W0 + W1 + W2 exactly), the activations are split per tile inside the kernel, and all nine (six leading) plane
    products are accumulated in fp32 -- the exact product of the fp32 values (up to 2^-24), at 16x the fp32-MFMA
    instruction rate.  Call refresh() after loading new teacher weights."""

    def __init__(self, model, gemm="bf16x9"):
        self.m = model
        for p in model.parameters():
            p.requires_grad_(False)
        if getattr(model, "dist_token", None) is None:
            raise ValueError("HipTeacher: the KD recipes use the distilled DeiT (deit.py:19)")
        if gemm not in ("f32", "bf16x9", "bf16x6", "f16x4"):
            raise ValueError("HipTeacher: gemm must be 'f32', 'bf16x9', 'bf16x6' or 'f16x4'")
        self.gemm = gemm
        self._ones = None
        self._planes = {}

    def refresh(self):
        self._planes.clear()

    def _linear(self, x2d, lin):
        """x2d @ lin.weight^T + lin.bias"""
        K = lin.weight.shape[1]
        if self.gemm == "f32" or K % 8 or not lin.weight.is_cuda:
            return ops.linear_fwd(x2d, lin.weight, lin.bias)
        N = lin.weight.shape[0]
        if self.gemm == "f16x4" and N > 128 and K % 32 == 0 and x2d.shape[0] * x2d.stride(0) * 4 < 2 ** 32:
            # round 5: weight = two fp16 planes of W 2^E (split once), activations two fp16 planes of x 2^E' (split in the kernel,
            # E' from the maximum word the producing kernel left): four plane products instead of nine, one launch
            pl = self._planes.get((id(lin), 2))
            if pl is None:
                pl = self._planes[(id(lin), 2)] = ops.split_f32_f16x2(lin.weight)
            return ops.linear_f16x4(x2d, pl, lin.bias)
        pl = self._planes.get((id(lin), 3))
        if pl is None:
            pl = self._planes[(id(lin), 3)] = ops.split_f32_bf16x3(lin.weight.detach())
        return ops.gemm_bf16x3x3_nt(x2d, pl, lin.bias, products=6 if self.gemm == "bf16x6" else 9)

    def parameters(self):
        return self.m.parameters()

    def _ln(self, norm, x2d, res2d=None):
        y, xs, _, _ = ops.layernorm_fwd(x2d, norm.weight, norm.bias, norm.eps, res2d=res2d, want_amax=self.gemm == "f16x4")
        return y, xs

    def _attention(self, attn, n1, B, N, C):
        H = attn.num_heads
        d = C // H
        Np = pad4(N)
        qkv = self._linear(n1, attn.qkv)                                             # (B*N, 3C): q | k | v column thirds
        if self.gemm == "f16x4" and ops.attn_f32_ok(N, d) and qkv.is_contiguous():
            # round 5: scores, softmax and P.V in one launch on fp16 planes (scores / probabilities never reach HBM)
            return self._linear(ops.attn_f32_fwd(qkv, B, H, N, d, attn.scale), attn.proj)
        S = torch.empty((B, H, N, Np), dtype=torch.float32, device=n1.device)
        ops.gemm(qkv, qkv, S, N, N, d, 3 * C, 3 * C, Np, transB=True, nb0=B, nb1=H, sA=(N * 3 * C, d), sB=(N * 3 * C, d),
                 sC=(H * N * Np, N * Np), offB=C)                                     # q k^T per (image, head)
        if self._ones is None or self._ones.numel() != N:
            self._ones = torch.ones(N, dtype=torch.float32, device=n1.device)
        P, _ = ops.softmax_lsq_fwd(S, self._ones, B * H * N, N, Np, N, attn.scale, 1, 1, need_values=False)   # softmax(scale * S)
        O = torch.empty((B * N, C), dtype=torch.float32, device=n1.device)
        ops.gemm(P, qkv, O, N, d, N, Np, 3 * C, C, nb0=B, nb1=H, sA=(H * N * Np, N * Np), sB=(N * 3 * C, d), sC=(N * C, d),
                 offB=2 * C)                                                          # P v, written head-interleaved
        return self._linear(O, attn.proj)

    def _mlp(self, mlp, n2):
        h = self._linear(n2, mlp.fc1)
        ops.gelu_(h, want_amax=self.gemm == "f16x4")
        return self._linear(h, mlp.fc2)

    @torch.no_grad()
    def __call__(self, images):
        m = self.m
        if not images.is_cuda:
            raise RuntimeError("HipTeacher: input must be on a HIP device; there is no CPU fallback")
        B = images.shape[0]
        if self.gemm == "f16x4":
            ops.amax_begin(images.device)        # the maximum words of this forward from the pool (zeroed once), not one fill each
        try:
            return self._forward(images, m, B)
        finally:
            if self.gemm == "f16x4":
                ops.amax_end()

    def _forward(self, images, m, B):
        pe = m.patch_embed.proj
        kh, kw = pe.kernel_size
        gh, gw = images.shape[2] // kh, images.shape[3] // kw
        cols = images.view(B, images.shape[1], gh, kh, gw, kw).permute(0, 2, 4, 1, 3, 5).reshape(B * gh * gw, -1)
        tok = ops.linear_fwd(cols, pe.weight.view(pe.out_channels, -1), pe.bias).view(B, gh * gw, -1)
        x = torch.cat((m.cls_token.expand(B, -1, -1), m.dist_token.expand(B, -1, -1), tok), dim=1) + m.pos_embed
        N, C = x.shape[1], x.shape[2]
        x2d = x.reshape(B * N, C).contiguous()
        pending = None
        for blk in m.blocks:
            if pending is None:
                n1, _ = self._ln(blk.norm1, x2d)
                xin = x2d
            else:
                n1, xin = self._ln(blk.norm1, x2d, pending)           # x + previous MLP output, and its norm, in one pass
            a = self._attention(blk.attn, n1, B, N, C)
            n2, x2d = self._ln(blk.norm2, xin, a)
            pending = self._mlp(blk.mlp, n2)
        x2d = x2d + pending
        heads_in = x2d.view(B, N, C)[:, :2].reshape(B * 2, C).contiguous()      # only the class / distillation tokens are read
        hn, _ = self._ln(m.norm, heads_in)
        hn = hn.view(B, 2, C)
        cls = ops.linear_fwd(hn[:, 0].contiguous(), m.head.weight, m.head.bias)
        dist = ops.linear_fwd(hn[:, 1].contiguous(), m.head_dist.weight, m.head_dist.bias)
        if m.training:
            return (cls, dist), None
        return (cls + dist) / 2, None
