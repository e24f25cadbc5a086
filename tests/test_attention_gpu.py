"""HIP attention core vs fp32 softmax attention on the same bf16-rounded inputs (through the C ABI).
Tolerance: outputs are bf16 (8-bit mantissa), P is rounded to bf16 before P.V -> 2e-2 abs on O(1) outputs;
LSE (f32) 2e-3."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(q, k, v, scale):
    s = (q.float() @ k.float().transpose(-1, -2)) * scale
    lse2 = torch.logsumexp(s, dim=-1) / math.log(2.0)
    return torch.softmax(s, dim=-1) @ v.float(), lse2


@pytest.mark.parametrize("B,H,T,hd", [(2, 3, 257, 80), (2, 2, 197, 64), (1, 2, 33, 16), (1, 1, 160, 128), (3, 4, 65, 32),
                                      (1, 2, 5, 48), (2, 3, 197, 80), (3, 2, 37, 80), (2, 2, 37, 64), (2, 2, 256, 80),
                                      (1, 3, 224, 80), (2, 2, 225, 80)])
def test_attn_fwd_matches_reference(B, H, T, hd):
    from octic_vits_amd import ops
    g = torch.Generator().manual_seed(T * 131 + hd)
    q, k, v = (torch.randn(B, H, T, hd, generator=g).to(torch.bfloat16).cuda() for _ in range(3))
    scale = hd ** -0.5
    o, lse = ops.attn_fwd(q, k, v, scale)
    ro, rl = _ref(q, k, v, scale)
    assert torch.allclose(o.float(), ro, atol=2e-2, rtol=2e-2), (o.float() - ro).abs().max()
    assert torch.allclose(lse, rl, atol=2e-3, rtol=1e-4), (lse - rl).abs().max()


def test_attn_fwd_strided_qkv_views():
    """q,k,v as views of one fused [B,T,3,H,hd] tensor (the standard block's layout, deit/vit.py:38-39)."""
    from octic_vits_amd import ops
    B, T, H, hd = 2, 257, 4, 80
    qkv = torch.randn(B, T, 3, H, hd, generator=torch.Generator().manual_seed(1)).to(torch.bfloat16).cuda()
    q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    o, _ = ops.attn_fwd(q, k, v, hd ** -0.5)
    ro, _ = _ref(q, k, v, hd ** -0.5)
    assert torch.allclose(o.float(), ro, atol=2e-2, rtol=2e-2)


def test_attn_fwd_large_logits_are_stable():
    from octic_vits_amd import ops
    B, H, T, hd = 1, 2, 257, 80
    g = torch.Generator().manual_seed(3)
    q, k, v = (torch.randn(B, H, T, hd, generator=g).to(torch.bfloat16).cuda() for _ in range(3))
    q = q * 20.0   # sharp softmax: the running-max rescale path matters
    o, lse = ops.attn_fwd(q, k, v, hd ** -0.5)
    ro, rl = _ref(q, k, v, hd ** -0.5)
    assert torch.isfinite(o.float()).all()
    assert torch.allclose(o.float(), ro, atol=3e-2, rtol=3e-2)


def _ref_grads(q, k, v, do, scale):
    q, k, v = (t.float().detach().requires_grad_(True) for t in (q, k, v))
    o = torch.softmax((q @ k.transpose(-1, -2)) * scale, dim=-1) @ v
    o.backward(do.float())
    return q.grad, k.grad, v.grad


@pytest.mark.parametrize("B,H,T,hd", [(2, 3, 257, 80), (2, 2, 197, 64), (1, 2, 33, 16), (3, 4, 65, 32), (1, 2, 5, 48),
                                      (2, 3, 197, 80), (3, 2, 37, 80), (2, 2, 37, 64), (2, 2, 256, 80), (1, 3, 224, 80),
                                      (2, 2, 225, 80)])
def test_attn_bwd_matches_reference(B, H, T, hd):
    """Gradients vs autograd of the fp32 reference on the same bf16 inputs: P and dS are rounded to bf16 before the
    gradient MFMAs -> 3e-2 of the gradient scale."""
    from octic_vits_amd.functional import AttnFn
    g = torch.Generator().manual_seed(T * 17 + hd)
    q, k, v = (torch.randn(B, H, T, hd, generator=g).to(torch.bfloat16).cuda().requires_grad_(True) for _ in range(3))
    do = torch.randn(B, H, T, hd, generator=g).to(torch.bfloat16).cuda()
    o = AttnFn.apply(q, k, v, hd ** -0.5)
    o.backward(do)
    rq, rk, rv = _ref_grads(q, k, v, do, hd ** -0.5)
    for name, got, want in (("dq", q.grad, rq), ("dk", k.grad, rk), ("dv", v.grad, rv)):
        scale = max(1.0, float(want.abs().max()))
        err = float((got.float() - want).abs().max())
        assert err <= 3e-2 * scale, f"{name}: max err {err:.3e} (scale {scale:.3g})"


def test_attn_fused_qkv_function_matches_reference():
    from octic_vits_amd.functional import AttnFusedQKVFn
    B, T, H, hd = 2, 257, 4, 80
    g = torch.Generator().manual_seed(9)
    qkv = torch.randn(B, T, 3, H, hd, generator=g).to(torch.bfloat16).cuda().requires_grad_(True)
    do = torch.randn(B, T, H * hd, generator=g).to(torch.bfloat16).cuda()
    out = AttnFusedQKVFn.apply(qkv, hd ** -0.5)
    out.backward(do)
    ref = qkv.detach().float().requires_grad_(True)
    q, k, v = (ref[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    ro = (torch.softmax((q @ k.transpose(-1, -2)) * hd ** -0.5, dim=-1) @ v).transpose(1, 2).reshape(B, T, H * hd)
    ro.backward(do.float())
    assert torch.allclose(out.float(), ro, atol=2e-2, rtol=2e-2)
    scale = max(1.0, float(ref.grad.abs().max()))
    assert float((qkv.grad.float() - ref.grad).abs().max()) <= 3e-2 * scale


@pytest.mark.parametrize("B,T,H,w", [(2, 257, 16, 10), (3, 197, 16, 10), (1, 64, 4, 10), (2, 33, 8, 10), (20, 37, 16, 10),
                                     (3, 197, 16, 8), (2, 257, 16, 8), (1, 64, 4, 8)])
def test_packed_attention_matches_pack_attention_unpack(B, T, H, w):
    """octic_attn_{fwd,bwd}_packed (AttentionD8 between its two linears, reference d8_layers.py:631-656, on the packed rows)
    against the three-step path it replaces: pack kernels (bit-exact vs the oracle's pack_heads, test_kernels_gpu) ->
    attention kernels on [B,H,T,80] -> unpack.  Same arithmetic per head in a different element order inside the dot
    products: outputs and gradients agree to bf16 rounding (2e-2 of scale)."""
    from octic_vits_amd import functional as OF, ops
    c = w * H                                    # w = 10: head_dim 80 (ViT-H/14); w = 8: head_dim 64 (ViT-L/16), round 4
    hd = 8 * w
    torch.manual_seed(B * 1000 + T)
    qkv = (torch.randn(B, T, 3 * 8 * c, device="cuda") * 0.7).bfloat16().requires_grad_(True)
    do = torch.randn(B, T, 8 * c, device="cuda").bfloat16()
    assert ops.attn_packed_ok(T, c, H, qkv.dtype)
    o1 = OF.AttnPackedFn.apply(qkv, H, c, hd ** -0.5)
    (g1,) = torch.autograd.grad(o1, qkv, do)
    q, k, v = OF.PackHeadsFn.apply(qkv, H, c)
    o2 = OF.UnpackHeadsFn.apply(OF.AttnFn.apply(q, k, v, hd ** -0.5), c)
    (g2,) = torch.autograd.grad(o2, qkv, do)
    for a, b, name in ((o1, o2, "o"), (g1, g2, "dqkv")):
        a, b = a.float(), b.float()
        err = (a - b).abs().max().item()
        assert err <= 2e-2 * b.abs().max().item() + 1e-3, (name, err, b.abs().max().item())
        rel = ((a - b).norm() / b.norm()).item()
        assert rel < 1e-2, (name, rel)


def test_packed_attention_matches_fp64_reference():
    """Packed-row attention against the ORACLE's AttentionD8 core in float64: oracle.pack_heads (reference
    d8_layers.py:631-643) -> softmax attention -> oracle.unpack_heads (d8_layers.py:650-656).  No HIP kernel and no
    product code in the reference leg; checks the head-piece schedule of the kernels (HeadMap) against the reference's
    head split, forward and backward."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from oracle import octic_ref as R
    from octic_vits_amd import functional as OF
    B, T, H = 2, 257, 16
    c = 10 * H
    cv = 3 * c
    torch.manual_seed(7)
    qkv = (torch.randn(B, T, 3 * 8 * c, device="cuda") * 0.7).bfloat16().requires_grad_(True)
    do = torch.randn(B, T, 8 * c, device="cuda").bfloat16()
    o = OF.AttnPackedFn.apply(qkv, H, c, 80 ** -0.5)
    (g,) = torch.autograd.grad(o, qkv, do)

    x = qkv.detach().float().cpu().double().requires_grad_(True)
    tup = tuple(x[..., i * cv:(i + 1) * cv] for i in range(4)) + (x[..., 4 * cv:].reshape(B, T, 2, 2 * cv),)
    q, k, v = R.pack_heads(tup, H)
    p = torch.softmax(q @ k.transpose(-1, -2) * 80 ** -0.5, -1)
    out5 = R.unpack_heads(p @ v)
    ref = torch.cat(list(out5[:4]) + [out5[4].flatten(-2)], dim=-1)
    (gref,) = torch.autograd.grad(ref, x, do.float().cpu().double())
    for got, want, name in ((o, ref, "o"), (g, gref, "dqkv")):
        got, want = got.detach().float().cpu().double(), want.detach()
        err = (got - want).abs().max().item()
        assert err <= 2e-2 * want.abs().max().item(), (name, err)
        assert ((got - want).norm() / want.norm()).item() < 1e-2, name


def _bwd_raw(q, k, v, o, do, lse, scale, fused):
    from octic_vits_amd import ops
    dq, dk, dv = (torch.empty_like(q) for _ in range(3))
    old = ops.ATTN_BWD_FUSED
    ops.ATTN_BWD_FUSED = fused
    try:
        ops.attn_bwd(q, k, v, o, do, lse, scale, dq, dk, dv)
    finally:
        ops.ATTN_BWD_FUSED = old
    return dq, dk, dv


# (17, 16) and (40, 16): more (batch, head) units than the chip has CUs - the round-5 kernel is persistent over heads: chunks of
# 1-2 and 2-3 heads per workgroup (the next head's V behind the last tiles, its K image and first tile behind the row stores)
@pytest.mark.parametrize("T", [197, 193, 224, 225, 256, 37, 50, 64, 33, 32, 17, 1])
def test_single_pass_backward_below_257_tokens_matches_fp64_and_the_pair(T):
    """Round 5: the single-pass kernel for 193 .. 256 tokens (bwd_kernel<7 | 8, false>: every token inside a key tile, the
    last tile partial - its keys masked out of P and dS -, no extra-row machinery) and the resident-image kernel for <= 64
    tokens (bwd_small_kernel<1 | 2>: DINOv2's 37-token local crops): float64 reference, the dq + dkv pair's own error as the
    yardstick, bitwise repeatable, and the rows of the partial tile checked on their own."""
    from octic_vits_amd import ops
    B, H, hd = 3, 16, 80
    g = torch.Generator().manual_seed(T)
    q, k, v = (torch.randn(B, H, T, hd, generator=g).to(torch.bfloat16).cuda() for _ in range(3))
    do = torch.randn(B, H, T, hd, generator=g).to(torch.bfloat16).cuda()
    scale = hd ** -0.5
    o, lse = ops.attn_fwd(q, k, v, scale)
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    ro = torch.softmax((qd @ kd.transpose(-1, -2)) * scale, dim=-1) @ vd
    ro.backward(do.double())
    assert ops._attn_bwd_phases(T, hd)[0][0] == 3
    fused = _bwd_raw(q, k, v, o, do, lse, scale, True)
    pair = _bwd_raw(q, k, v, o, do, lse, scale, False)
    again = _bwd_raw(q, k, v, o, do, lse, scale, True)
    t0 = 32 * ((T - 1) // 32)                                  # first token of the last (partial) tile
    for name, f, p, a, want in zip(("dq", "dk", "dv"), fused, pair, again, (qd.grad, kd.grad, vd.grad)):
        assert torch.equal(f, a), f"{name}: two launches differ"
        sc = max(1.0, float(want.abs().max()))
        ef, ep = float((f.double() - want).abs().max()), float((p.double() - want).abs().max())
        assert ef <= 3e-2 * sc, f"{name}: max err {ef:.3e} (scale {sc:.3g})"
        assert ef <= 1.5 * ep + 1e-3 * sc, f"{name}: single-pass err {ef:.3e} vs two-kernel {ep:.3e}"
        assert float((f.double()[:, :, t0:] - want[:, :, t0:]).abs().max()) <= 3e-2 * sc, f"{name}: last tile"
        # (one token: dS = 0, so dq and dk are exactly zero in exact arithmetic - absolute bound there)
        assert float((f.double() - want).norm()) < 1.2e-2 * max(float(want.norm()), 1e-3), name


@pytest.mark.parametrize("B,H", [(2, 3), (5, 16), (17, 16), (40, 16)])
def test_single_pass_backward_matches_fp64_and_the_two_kernel_path(B, H):
    """csrc/attn80_bwd.hip (round 4: P and dS once per tile pair, dQ as a fixed-order sum of eight per-wave partials,
    key 256 on the vector unit) against (a) float64 autograd of softmax attention on the same bf16 inputs - 3e-2 of the
    gradient scale, the bound of the two-kernel path - with the error of every gradient ALSO required to be no worse than
    1.5 x the two-kernel path's own error; (b) bitwise repeatability; (c) the rows that take special paths: token 256 as
    key (dk / dv row 256) and as query (dq row 256) are checked separately at the same tolerance."""
    from octic_vits_amd import ops
    T, hd = 257, 80
    g = torch.Generator().manual_seed(31 + B)
    q, k, v = (torch.randn(B, H, T, hd, generator=g).to(torch.bfloat16).cuda() for _ in range(3))
    do = torch.randn(B, H, T, hd, generator=g).to(torch.bfloat16).cuda()
    scale = hd ** -0.5
    o, lse = ops.attn_fwd(q, k, v, scale)
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    ro = torch.softmax((qd @ kd.transpose(-1, -2)) * scale, dim=-1) @ vd
    ro.backward(do.double())
    fused = _bwd_raw(q, k, v, o, do, lse, scale, True)
    pair = _bwd_raw(q, k, v, o, do, lse, scale, False)
    again = _bwd_raw(q, k, v, o, do, lse, scale, True)
    for name, f, p, a, want in zip(("dq", "dk", "dv"), fused, pair, again, (qd.grad, kd.grad, vd.grad)):
        assert torch.equal(f, a), f"{name}: two launches differ"
        sc = max(1.0, float(want.abs().max()))
        ef, ep = float((f.double() - want).abs().max()), float((p.double() - want).abs().max())
        assert ef <= 3e-2 * sc, f"{name}: max err {ef:.3e} (scale {sc:.3g})"
        assert ef <= 1.5 * ep + 1e-3 * sc, f"{name}: single-pass err {ef:.3e} vs two-kernel {ep:.3e}"
        e256 = float((f.double()[:, :, 256] - want[:, :, 256]).abs().max())
        assert e256 <= 3e-2 * sc, f"{name}[256]: {e256:.3e}"
        rel = float((f.double() - want).norm() / want.norm())
        assert rel < 1.2e-2, (name, rel)


def test_single_pass_backward_on_strided_fused_qkv_rows_and_spiky_scores():
    """The standard half's layout (q, k, v = strided views of one [B,T,3,H,hd] tensor; gradients into one tensor of the same
    layout) and scores with a wide range (|x| up to ~25 after scaling): exp2(x - lse) spans the whole bf16 range of P."""
    from octic_vits_amd import ops
    B, H, T, hd = 3, 16, 257, 80
    g = torch.Generator().manual_seed(77)
    qkv = torch.randn(B, T, 3, H, hd, generator=g)
    qkv[:, :, 0] *= 3.0
    qkv[:, :, 1] *= 3.0
    qkv = qkv.to(torch.bfloat16).cuda()
    q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    do = torch.randn(B, H, T, hd, generator=g).to(torch.bfloat16).cuda()
    scale = hd ** -0.5
    o, lse = ops.attn_fwd(q, k, v, scale)
    dqkv = torch.empty_like(qkv)
    dq, dk, dv = (dqkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    ops.attn_bwd(q, k, v, o, do, lse, scale, dq, dk, dv)
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    ro = torch.softmax((qd @ kd.transpose(-1, -2)) * scale, dim=-1) @ vd
    ro.backward(do.double())
    for name, got, want in (("dq", dq, qd.grad), ("dk", dk, kd.grad), ("dv", dv, vd.grad)):
        sc = max(1.0, float(want.abs().max()))
        err = float((got.double() - want).abs().max())
        assert err <= 3e-2 * sc, f"{name}: max err {err:.3e} (scale {sc:.3g})"


def test_single_pass_backward_on_packed_rows_over_several_heads_per_workgroup():
    """Packed LinearD8 rows with more units than CUs (24 x 16 = 384 heads: one or two per workgroup): the persistent
    single-pass kernel against the dq + dkv pair (itself held to float64 above) on the same operands, and bitwise repeatable."""
    from octic_vits_amd import functional as OF
    from octic_vits_amd import ops
    B, T, H = 24, 257, 16
    c = 10 * H
    g = torch.Generator().manual_seed(3)
    qkv = (torch.randn(B, T, 3 * 8 * c, generator=g) * 0.7).to(torch.bfloat16).cuda().requires_grad_(True)
    do = torch.randn(B, T, 8 * c, generator=g).to(torch.bfloat16).cuda()
    o = OF.AttnPackedFn.apply(qkv, H, c, 80 ** -0.5)
    old = ops.ATTN_BWD_FUSED
    try:
        ops.ATTN_BWD_FUSED = True
        (g1,) = torch.autograd.grad(o, qkv, do, retain_graph=True)
        (g1b,) = torch.autograd.grad(o, qkv, do, retain_graph=True)
        ops.ATTN_BWD_FUSED = False
        (g2,) = torch.autograd.grad(o, qkv, do, retain_graph=True)
    finally:
        ops.ATTN_BWD_FUSED = old
    assert torch.equal(g1, g1b)
    sc = float(g2.float().abs().max())
    assert float((g1.float() - g2.float()).abs().max()) <= 2e-2 * sc
    assert float((g1.float() - g2.float()).norm() / g2.float().norm()) < 6e-3


@pytest.mark.parametrize("T,hd", [(197, 64), (197, 80), (37, 80), (37, 64), (257, 64)])
def test_attention_of_the_other_model_shapes_matches_fp64(T, hd):
    """The shapes beside ViT-H/14's (257, 80): ViT-L/16 (197, 64), the DINOv2 ViT-H/16 global crops (197, 80) and 96 x 96
    local crops (37, 80) - forward, log-sum-exp and all three gradients against float64 softmax attention on the same bf16
    inputs, on the standard half's strided fused-QKV layout, whatever kernel family the routing picks for the shape."""
    from octic_vits_amd import ops
    B, H = 5, 16
    g = torch.Generator().manual_seed(T + hd)
    qkv = torch.randn(B, T, 3, H, hd, generator=g).to(torch.bfloat16).cuda()
    q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    do = torch.randn(B, H, T, hd, generator=g).to(torch.bfloat16).cuda()
    scale = hd ** -0.5
    o, lse = ops.attn_fwd(q, k, v, scale)
    dqkv = torch.empty_like(qkv)
    dq, dk, dv = (dqkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    ops.attn_bwd(q, k, v, o, do, lse, scale, dq, dk, dv)
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    s = (qd @ kd.transpose(-1, -2)) * scale
    ro = torch.softmax(s, dim=-1) @ vd
    ro.backward(do.double())
    assert float((o.double() - ro).abs().max()) <= 2e-2 * max(1.0, float(ro.abs().max()))
    rl = torch.logsumexp(s.detach(), dim=-1) / math.log(2.0)
    assert float((lse.double() - rl).abs().max()) <= 2e-3
    for name, got, want in (("dq", dq, qd.grad), ("dk", dk, kd.grad), ("dv", dv, vd.grad)):
        sc = max(1.0, float(want.abs().max()))
        assert float((got.double() - want).abs().max()) <= 3e-2 * sc, name
        assert float((got.double() - want).norm() / want.norm()) < 1.2e-2, name
