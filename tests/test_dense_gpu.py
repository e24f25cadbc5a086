"""GPU parity of the standard-half row kernels (plain LayerNorm, layer-scale + stochastic-depth + residual tail)
through the C ABI, and of a whole fused standard block against the eager f32 oracle block.

Tolerances: f32 kernels vs fp64 torch math 2e-5; bf16 outputs 2^-8 relative (one rounding); column-sum
gradients (reductions over thousands of rows) 1e-4 relative to their scale.
"""
import copy

import pytest
import torch
import torch.nn.functional as F

from oracle import octic_ref as R

pytestmark = pytest.mark.gpu
DEV = "cuda"


def ops():
    from octic_vits_amd import ops as o
    return o


def gen(seed, *shape, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale)


@pytest.mark.parametrize("rows,d", [(1, 4), (37, 132), (257, 1280), (1030, 2048), (3000, 768), (16448, 1280)])
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
def test_dense_layernorm_fwd_bwd(rows, d, out_dtype):
    x = gen(1, rows, d) * 3 + 0.5
    w, b = gen(2, d) * 0.3 + 1, gen(3, d) * 0.2
    gy = gen(4, rows, d)
    dres = gen(5, rows, d)
    if out_dtype == torch.bfloat16:
        gy = gy.bfloat16().float()
    x64 = x.double().requires_grad_(True)
    w64, b64 = w.double().requires_grad_(True), b.double().requires_grad_(True)
    y64 = F.layer_norm(x64, (d,), w64, b64, 1e-6)
    y64.backward(gy.double())
    xc, wc, bc = x.to(DEV), w.to(DEV), b.to(DEV)
    y, stats = ops().dense_layernorm_fwd(xc, wc, bc, 1e-6, out_dtype)
    tol = 2e-5 if out_dtype == torch.float32 else 2 ** -7
    torch.testing.assert_close(y.float().cpu(), y64.detach().float(), rtol=tol, atol=tol)
    mean = x.double().mean(-1)
    torch.testing.assert_close(stats[:, 0].cpu().double(), mean, rtol=1e-5, atol=1e-5)
    dx, dw, db = ops().dense_layernorm_bwd(gy.to(DEV).to(out_dtype), xc, wc, stats, dres.to(DEV))
    ref_dx = x64.grad.float() + dres
    torch.testing.assert_close(dx.cpu(), ref_dx, rtol=1e-4, atol=1e-4)
    sc = float(rows) ** 0.5
    torch.testing.assert_close(dw.cpu(), w64.grad.float(), rtol=1e-4, atol=1e-4 * sc)
    torch.testing.assert_close(db.cpu(), b64.grad.float(), rtol=1e-4, atol=1e-4 * sc)
    # no residual cotangent, no parameter gradients
    dx2, dw2, db2 = ops().dense_layernorm_bwd(gy.to(DEV).to(out_dtype), xc, wc, stats, None, want_param_grads=False)
    torch.testing.assert_close(dx2.cpu(), x64.grad.float(), rtol=1e-4, atol=1e-4)
    assert dw2 is None and db2 is None


def test_dense_layernorm_without_affine_and_empty():
    x = gen(1, 10, 64).to(DEV)
    y, stats = ops().dense_layernorm_fwd(x, None, None, 1e-5, torch.float32)
    torch.testing.assert_close(y, F.layer_norm(x, (64,), None, None, 1e-5), rtol=2e-5, atol=2e-5)
    e = torch.empty(0, 64, device=DEV)
    y, stats = ops().dense_layernorm_fwd(e, None, None, 1e-5, torch.float32)
    assert y.shape == (0, 64)
    with pytest.raises(RuntimeError):
        ops().dense_layernorm_fwd(gen(1, 4, 6).to(DEV), None, None, 1e-5, torch.float32)      # d % 4 != 0


@pytest.mark.parametrize("B,T,d", [(1, 1, 8), (3, 17, 132), (8, 257, 1280), (64, 257, 1280)])
@pytest.mark.parametrize("ydt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("with_rs,with_gamma", [(True, True), (False, True), (True, False), (False, False)])
def test_scale_residual_fwd_bwd(B, T, d, ydt, with_rs, with_gamma):
    if B == 64 and not (with_rs and with_gamma):
        pytest.skip("full size only for the full form")
    x, y = gen(1, B, T, d), gen(2, B, T, d).to(ydt)
    gamma = (gen(3, d) * 0.5) if with_gamma else None
    rs = (torch.bernoulli(torch.full((B,), 0.6), generator=torch.Generator().manual_seed(4)) / 0.6) if with_rs else None
    gout = gen(5, B, T, d)
    dev = lambda t: None if t is None else t.to(DEV)
    out = ops().scale_residual_fwd(dev(x), dev(y), dev(gamma), dev(rs), T)
    s = torch.ones(B, 1, 1, dtype=torch.float64) if rs is None else rs.double().view(B, 1, 1)
    g = torch.ones(d, dtype=torch.float64) if gamma is None else gamma.double()
    ref = x.double() + s * g * y.double()
    torch.testing.assert_close(out.cpu().double(), ref, rtol=2e-6, atol=2e-6)
    gy, dgamma, colsum = ops().scale_residual_bwd(dev(gout), dev(y), dev(gamma), dev(rs), T)
    ref_gy = s * g * gout.double()
    tol = 1e-6 if ydt == torch.float32 else 2 ** -8
    torch.testing.assert_close(gy.cpu().double(), ref_gy, rtol=tol, atol=tol * 1e-2)
    sc = float(B * T) ** 0.5
    torch.testing.assert_close(dgamma.cpu().double(), (s * gout.double() * y.double()).sum((0, 1)), rtol=1e-4,
                               atol=1e-5 * sc)
    torch.testing.assert_close(colsum.cpu().double(), g * (s * gout.double()).sum((0, 1)), rtol=1e-4, atol=1e-5 * sc)


def _blocks(kind, dim, heads, drop_path):
    from octic_vits_amd import vit
    torch.manual_seed(0)
    if kind == "layer_scale":
        blk = vit.Layer_scale_init_Block(dim, heads, qkv_bias=True, drop_path=drop_path, init_values=0.5)
        ref = R.Layer_scale_init_Block(dim, heads, qkv_bias=True, drop_path=drop_path, init_values=0.5)
    else:
        blk = vit.Block(dim, heads, qkv_bias=True, drop_path=drop_path, init_values=None)
        ref = R.Block(dim, heads, qkv_bias=True, drop_path=drop_path, init_values=None)
    with torch.no_grad():
        for i, p in enumerate(blk.parameters()):
            p.copy_(gen(100 + i, *p.shape) * (0.08 if p.ndim > 1 else 0.3) + (1.0 if p.ndim == 1 and i % 2 == 0 else 0.0))
    ref.load_state_dict(blk.state_dict())
    return blk.to(DEV), ref


@pytest.mark.parametrize("kind", ["layer_scale", "timm"])
def test_fused_standard_block_matches_oracle(kind):
    """bf16-autocast fused block vs the f32 CPU oracle block: output within 1e-2 of the output scale (bf16 GEMM
    operands), every gradient within 3e-2 relative L2."""
    dim, heads, B, T = 128, 4, 4, 50
    blk, ref = _blocks(kind, dim, heads, 0.0)
    x = gen(7, B, T, dim)
    gout = gen(8, B, T, dim)
    xr = x.clone().requires_grad_(True)
    yr = ref(xr)
    yr.backward(gout)
    xg = x.to(DEV).requires_grad_(True)
    o = ops()
    o.KERNEL_TIMER.enable()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        yg = blk(xg)
    yg.backward(gout.to(DEV))
    names = set(o.KERNEL_TIMER.summary())
    o.KERNEL_TIMER.disable()
    assert any(n.startswith("dense_ln_fwd") for n in names) and any(n.startswith("scale_residual_bwd") for n in names)
    assert yg.dtype == torch.float32
    scale = float(yr.detach().abs().mean())
    assert float((yg.detach().cpu() - yr.detach()).abs().max()) < 2e-2 * max(scale, 1.0)
    rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-12))
    assert rel(xg.grad.cpu(), xr.grad) < 3e-2
    for (n, p), (_, q) in zip(blk.named_parameters(), ref.named_parameters()):
        assert rel(p.grad.cpu().float(), q.grad) < 3e-2, n


def test_fused_standard_block_drop_path_draws_like_eager():
    """Stochastic depth in the fused block consumes the RNG exactly like the eager block: same seed, same kept samples."""
    dim, heads, B, T = 64, 4, 16, 20
    blk, _ = _blocks("layer_scale", dim, heads, 0.5)
    blk.train()
    x = gen(9, B, T, dim).to(DEV)
    torch.manual_seed(123)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        fused = blk(x)
    torch.manual_seed(123)
    eager = blk(x.double().float())                       # f32, no autocast -> eager path
    kept_f = (fused - x).flatten(1).abs().amax(1) > 0
    assert 0 < int(kept_f.sum()) <= B
    torch.testing.assert_close(fused, eager, rtol=3e-2, atol=3e-2)


def test_fused_block_is_skipped_outside_its_regime():
    dim, heads = 64, 4
    blk, ref = _blocks("layer_scale", dim, heads, 0.0)
    x = gen(9, 2, 10, dim)
    o = ops()
    o.KERNEL_TIMER.enable()
    y = blk(x.to(DEV))                                       # f32 without autocast: eager ops, exact reference math
    names = set(o.KERNEL_TIMER.summary())
    o.KERNEL_TIMER.disable()
    assert not any(n.startswith("dense_") for n in names)
    torch.testing.assert_close(y.cpu(), ref(x), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("rows,d", [(1, 8), (37, 136), (514, 640), (16448, 5120)])
def test_dense_gelu_bwd_and_bias_grad(rows, d):
    """dh = gelu'(h) g (bf16 out: one rounding, 2^-8) and its column sums (reduction over rows: 1e-3 of scale)."""
    h = gen(1, rows, d).bfloat16()
    g = gen(2, rows, d).bfloat16()
    h64 = h.double().requires_grad_(True)
    F.gelu(h64).backward(g.double())
    dh, db = ops().dense_gelu_bwd(h.to(DEV), g.to(DEV))
    torch.testing.assert_close(dh.cpu().double(), h64.grad, rtol=2 ** -7, atol=2e-3)
    # the bias gradient sums exactly the bf16 values the downstream GEMMs consume
    torch.testing.assert_close(db.cpu().double(), dh.cpu().double().sum(0), rtol=1e-4, atol=1e-3 * float(rows) ** 0.5)
    dh2, db2 = ops().dense_gelu_bwd(h.to(DEV), g.to(DEV), want_colsum=False)
    assert db2 is None and torch.equal(dh2, dh)


def test_new_entry_points_reject_bad_arguments():
    """Negative OCTIC_E* codes, never a slower path: shapes the kernels do not cover, null tables."""
    import ctypes

    from octic_vits_amd import _lib
    L = _lib.lib()
    h = torch.zeros(4, 12, device=DEV, dtype=torch.bfloat16)          # d % 8 != 0
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    assert L.octic_dense_gelu_bwd(p(h), p(h), p(h), None, 4, 12, None) == -1          # OCTIC_ESHAPE
    assert L.octic_dense_gelu_bwd(None, p(h), p(h), None, 4, 16, None) == -4          # OCTIC_ENULL
    assert L.octic_linear_d8_prep_batch(None, 1, 1, 1, None) == -4
    assert L.octic_linear_d8_prep_batch(p(h), 0, 1, 1, None) == -1
    x = torch.zeros(4, 16, device=DEV)
    assert L.octic_scale_residual_fwd(p(x), p(x), 7, None, None, 1, p(x), 4, 16, None) == -3   # OCTIC_EDTYPE
    assert L.octic_linear_d8_prep_batch_blocks(160, 640) == 4 * 3 * 10 + 5 * 20


@pytest.mark.parametrize("rows,T,d,ydt", [(150, 50, 132, torch.bfloat16), (16448, 257, 1280, torch.bfloat16),
                                          (771, 257, 768, torch.float32), (5, 5, 4, torch.bfloat16)])
def test_resid_layernorm_fwd_equals_the_two_passes(rows, T, d, ydt):
    """octic_dense_resid_layernorm_fwd = octic_scale_residual_fwd followed by octic_dense_layernorm_fwd, bit for bit
    (stream, normalised rows and statistics)."""
    o = ops()
    x = gen(11, rows, d).to(DEV)
    yb = gen(12, rows, d).to(DEV).to(ydt)
    gamma = (gen(13, d) * 0.1).to(DEV)
    rs = (torch.arange(rows // T) % 3 != 0).float().div(0.66).to(DEV)
    w, b = (gen(14, d) * 0.3 + 1).to(DEV), (gen(15, d) * 0.2).to(DEV)
    for g_, r_ in ((gamma, rs), (None, None), (gamma, None)):
        out = o.scale_residual_fwd(x, yb, g_, r_, T)
        y, st = o.dense_layernorm_fwd(out, w, b, 1e-6, torch.bfloat16)
        out2, y2, st2 = o.dense_resid_layernorm_fwd(x, yb, g_, r_, T, w, b, 1e-6, torch.bfloat16)
        assert torch.equal(out2, out) and torch.equal(y2, y) and torch.equal(st2, st)


def test_next_norm_fusion_is_bitwise_the_unfused_block_chain():
    """Three chained standard blocks (vit.link_blocks): with the residual add + next LayerNorm as one row pass the output
    and every gradient equal the unfused chain bit for bit; the fused chain runs ONE stand-alone LayerNorm (the first
    block's norm1), the unfused one six."""
    from octic_vits_amd import functional as OF, vit
    dim, heads, B, T = 128, 4, 4, 50
    blocks = torch.nn.ModuleList([_blocks("layer_scale", dim, heads, 0.0)[0] for _ in range(3)])
    for i, blk in enumerate(blocks):
        for j, p in enumerate(blk.parameters()):
            torch.manual_seed(100 * i + j)
            p.data.add_(torch.randn_like(p) * 0.05)
    vit.link_blocks(blocks)
    x = gen(21, B, T, dim).to(DEV)
    gout = gen(22, B, T, dim).to(DEV)
    o = ops()
    res = {}
    saved, saved_tail = OF.NEXT_NORM_FUSED, OF.LN_TAIL_FUSED
    OF.LN_TAIL_FUSED = False          # (its slab sums run in another order: compared separately below)
    try:
        for mode in (True, False):
            OF.NEXT_NORM_FUSED = mode
            for p in blocks.parameters():
                p.grad = None
            xg = x.clone().requires_grad_(True)
            o.KERNEL_TIMER.enable()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                h = xg
                for blk in blocks:
                    h = blk(h)
            h.backward(gout)
            s = o.KERNEL_TIMER.summary()
            o.KERNEL_TIMER.disable()
            res[mode] = (h.detach().clone(), xg.grad.clone(), [p.grad.clone() for p in blocks.parameters()],
                         sum(v["launches"] for n, v in s.items() if n.startswith("dense_ln_fwd")),
                         sum(v["launches"] for n, v in s.items() if n.startswith("dense_resid_ln_fwd")))
    finally:
        OF.NEXT_NORM_FUSED, OF.LN_TAIL_FUSED = saved, saved_tail
    assert res[True][3] == 1 and res[True][4] == 5 and res[False][3] == 6 and res[False][4] == 0
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    for (n, _), a, b in zip(blocks.named_parameters(), res[True][2], res[False][2]):
        assert torch.equal(a, b), n


@pytest.mark.parametrize("rows,T,d", [(16448, 257, 1280), (771, 257, 768), (300, 50, 256)])
def test_layernorm_bwd_tail_equals_the_two_passes(rows, T, d):
    """octic_dense_layernorm_bwd_tail = octic_dense_layernorm_bwd then octic_scale_residual_bwd: dx and the bf16 branch
    cotangent bit for bit, the four parameter-gradient sums to 1e-5 of their scale (other slab order)."""
    o = ops()
    x = (gen(31, rows, d) * 2 + 0.3).to(DEV)
    w = (gen(32, d) * 0.3 + 1).to(DEV)
    gy = gen(33, rows, d).to(DEV).bfloat16()
    dres = gen(34, rows, d).to(DEV)
    yb = gen(35, rows, d).to(DEV).bfloat16()
    gamma = (gen(36, d) * 0.1).to(DEV)
    rs = (torch.arange(rows // T) % 3 != 0).float().div(0.66).to(DEV)
    _, stats = o.dense_layernorm_fwd(x, w, None, 1e-6, torch.bfloat16)
    for dr, g_, r_ in ((dres, gamma, rs), (None, None, None), (dres, gamma, None)):
        dx, dw, db = o.dense_layernorm_bwd(gy, x, w, stats, dr)
        gyb, dgamma, colsum = o.scale_residual_bwd(dx, yb, g_, r_, T)
        dx2, dw2, db2, gyb2, dgamma2, colsum2 = o.dense_layernorm_bwd_tail(gy, x, w, stats, dr, yb, g_, r_, T)
        assert torch.equal(dx2, dx) and torch.equal(gyb2, gyb)
        for a, b in ((dw2, dw), (db2, db), (dgamma2, dgamma), (colsum2, colsum)):
            assert float((a - b).abs().max()) <= 1e-5 * max(1.0, float(b.abs().max()))


def test_ln_tail_fusion_in_a_block_chain_matches_the_unfused_chain():
    """Three chained standard blocks of width 256 (whole-chunk rows: the fused LayerNorm-backward + residual-tail-backward
    pass and the unpredicated row paths are live): output and x-gradient bit for bit, parameter gradients to 1e-5 of
    their scale against the chain with the two backward passes separate (their slab sums run in another order)."""
    from octic_vits_amd import functional as OF, vit
    dim, heads, B, T = 256, 4, 4, 50
    blocks = torch.nn.ModuleList([_blocks("layer_scale", dim, heads, 0.0)[0] for _ in range(3)])
    for i, blk in enumerate(blocks):
        for j, p in enumerate(blk.parameters()):
            torch.manual_seed(100 * i + j)
            p.data.add_(torch.randn_like(p) * 0.05)
    vit.link_blocks(blocks)
    x = gen(41, B, T, dim).to(DEV)
    gout = gen(42, B, T, dim).to(DEV)
    o = ops()
    res = {}
    saved = OF.LN_TAIL_FUSED
    try:
        for mode in (True, False):
            OF.LN_TAIL_FUSED = mode
            for p in blocks.parameters():
                p.grad = None
            xg = x.clone().requires_grad_(True)
            o.KERNEL_TIMER.enable()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                h = xg
                for blk in blocks:
                    h = blk(h)
            h.backward(gout)
            s = o.KERNEL_TIMER.summary()
            o.KERNEL_TIMER.disable()
            res[mode] = (h.detach().clone(), xg.grad.clone(), [p.grad.clone() for p in blocks.parameters()],
                         sum(v["launches"] for n, v in s.items() if n.startswith("dense_ln_bwd_tail")))
    finally:
        OF.LN_TAIL_FUSED = saved
    assert res[True][3] == 5 and res[False][3] == 0
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    for (n, _), a, b in zip(blocks.named_parameters(), res[True][2], res[False][2]):
        assert float((a.float() - b.float()).abs().max()) <= 1e-5 * max(1.0, float(b.float().abs().max())), n


def test_dense_finish_batch_is_bitwise_the_single_launches():
    """octic_dense_finish_batch over 70 jobs of mixed sizes (two launches: 64 + 6) against 70 octic_dense_finish calls."""
    import ctypes
    from octic_vits_amd import ops as o, _lib
    L = _lib.lib()
    g = torch.Generator(device=DEV).manual_seed(3)
    jobs, want, keep = [], [], []
    for i in range(70):
        d = [1280, 640, 2560, 256, 12][i % 5]
        nblk = [256, 257, 16, 65, 3][i % 5]
        part = torch.randn(nblk, 2, d, device=DEV, generator=g)
        scale = torch.rand(d, device=DEV, generator=g) + 0.5 if i % 3 == 0 else None
        o0 = torch.full((d,), float("nan"), device=DEV) if i % 7 else None
        o1 = torch.full((d,), float("nan"), device=DEV) if i % 4 else None
        r0 = torch.full((d,), float("nan"), device=DEV) if o0 is not None else None
        r1 = torch.full((d,), float("nan"), device=DEV) if o1 is not None else None
        o.check(L.octic_dense_finish(o._p(part), nblk, d, o._p(r0), o._p(r1), o._p(scale), o._stream(part)))
        jobs.append((part, nblk, d, o0, o1, scale))
        want.append((r0, r1))
        keep.append((part, scale))
    arr = (o._FinishJob * len(jobs))()
    for i, (part, nblk, d, o0, o1, scale) in enumerate(jobs):
        arr[i].partials = part.data_ptr()
        arr[i].out0 = o0.data_ptr() if o0 is not None else None
        arr[i].out1 = o1.data_ptr() if o1 is not None else None
        arr[i].scale1 = scale.data_ptr() if scale is not None else None
        arr[i].nblocks, arr[i].d = nblk, d
    o.check(L.octic_dense_finish_batch(ctypes.cast(arr, ctypes.c_void_p), len(jobs), o._stream(jobs[0][0])))
    torch.cuda.synchronize()
    for (part, nblk, d, o0, o1, scale), (r0, r1) in zip(jobs, want):
        if o0 is not None:
            assert torch.equal(o0, r0)
        if o1 is not None:
            assert torch.equal(o1, r1)
    assert L.octic_dense_finish_batch(None, 1, None) == -4           # OCTIC_ENULL
    assert L.octic_dense_finish_batch(ctypes.cast(arr, ctypes.c_void_p), 0, None) == 0


def test_deferred_finishes_give_bitwise_the_same_parameter_gradients():
    """A standard block's backward with the parameter-gradient reductions postponed to the end of the pass (one batched
    launch, ops._DeferredFinishes) against the immediate launches: every gradient bit for bit; outside a backward pass and
    with the switch off nothing is postponed."""
    from octic_vits_amd import ops as o
    from octic_vits_amd.vit import Layer_scale_init_Block
    torch.manual_seed(0)
    # two DISTINCT blocks (chained fusions: next-norm, LayerNorm tail).  A parameter used twice in one pass would have its
    # second gradient added to the first before the postponed launch has produced either - the precondition in the docstring
    # of ops._DeferredFinishes ("nothing reads those gradients earlier") includes autograd's own accumulation.
    blk = torch.nn.Sequential(*[Layer_scale_init_Block(dim=256, num_heads=4, qkv_bias=True, init_values=0.5, drop_path=0.0)
                                for _ in range(2)]).cuda().train()
    x = torch.randn(6, 65, 256, device=DEV)
    cot = torch.randn(6, 65, 256, device=DEV)
    res = {}
    for mode in (False, True):
        blk.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = blk(xi)
        o.DEFERRED_FINISHES.enabled = mode
        try:
            y.backward(cot)
        finally:
            o.DEFERRED_FINISHES.enabled = False
        assert not o.DEFERRED_FINISHES.jobs and not o.DEFERRED_FINISHES.armed and not o.DEFERRED_FINISHES.pairs
        res[mode] = [xi.grad.clone()] + [p.grad.clone() for p in blk.parameters()]
    for a, b in zip(res[False], res[True]):
        assert torch.equal(a, b)


def test_deferred_octic_layernorm_finishes_are_bitwise_the_immediate_ones():
    """Two octic blocks: the LayerNormD8 parameter-gradient reductions postponed to the end of the backward pass
    (octic_layernorm_d8_bwd_finish_batch) against the immediate launches."""
    from octic_vits_amd import ops as o
    from octic_vits_amd.d8_layers import Layer_scale_init_BlockD8, Octic
    torch.manual_seed(0)
    blk = torch.nn.Sequential(*[Layer_scale_init_BlockD8(256, 4, qkv_bias=True, init_values=0.5) for _ in range(2)]).cuda().train()
    x = torch.randn(5, 33, 256, device=DEV)
    cot = torch.randn(5, 33, 256, device=DEV)
    res = {}
    for mode in (False, True):
        blk.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = blk(Octic(xi, 32))
        o.DEFERRED_FINISHES.enabled = mode
        try:
            y.packed.backward(cot)
        finally:
            o.DEFERRED_FINISHES.enabled = False
        assert not o.DEFERRED_FINISHES.ln_jobs and not o.DEFERRED_FINISHES.wg_jobs and not o.DEFERRED_FINISHES.jobs
        assert not o.DEFERRED_FINISHES.armed
        res[mode] = [xi.grad.clone()] + [p.grad.clone() for p in blk.parameters() if p.grad is not None]
    assert len(res[False]) == len(res[True]) > 20
    for a, b in zip(res[False], res[True]):
        assert torch.equal(a, b)
