"""GPU parity of every HIP kernel, called through the C ABI (ctypes), against the CPU oracle /
fp64 torch math on the same seeded inputs.

Tolerances (stated per check):
  f32 path  : kernels keep f32 end to end (exact-f32 MFMA) -> rtol/atol 2e-5 vs fp64-accurate oracle
              (GEMM rows are K<=1280 long fma chains: 1e-4 on the largest).
  bf16 path : operands rounded to bf16 (8 bit mantissa, rel 2^-9 = 2e-3 per element), f32 accumulate,
              bf16 outputs -> compare against the oracle evaluated on the SAME bf16-rounded inputs,
              atol/rtol 2e-2 of the output scale.
"""
import math

import numpy as np
import pytest
import torch

import cases
from oracle import octic_ref as R

pytestmark = pytest.mark.gpu

DEV = "cuda"


def ops():
    from octic_vits_amd import ops as o
    return o


def pack(xs):
    """reference 5-tuple -> packed [..., 8c]"""
    return torch.cat([xs[0], xs[1], xs[2], xs[3], xs[4].flatten(-2)], dim=-1).contiguous()


def unpack(t, c):
    return (t[..., :c], t[..., c:2 * c], t[..., 2 * c:3 * c], t[..., 3 * c:4 * c], t[..., 4 * c:].unflatten(-1, (2, 2 * c)))


def rand5(tag, B, T, c, dtype=torch.float32):
    return tuple(t.to(dtype) for t in cases.tuple5(tag, B, T, c))


def close(got, want, rtol, atol, msg=""):
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    scale = max(1.0, float(want.abs().max()))
    err = (got - want).abs().max().item()
    assert torch.allclose(got, want, rtol=rtol, atol=atol * scale), f"{msg}: max err {err:.3e} (scale {scale:.3g})"


SHAPES = [(2, 5, 8), (3, 17, 48), (2, 257, 160)]
TOLS = {torch.float32: (2e-5, 2e-5), torch.bfloat16: (2e-2, 2e-2)}


# ------------------------------------------------------------------------------------------- GELU
@pytest.mark.parametrize("B,T,c", SHAPES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("layout", ["packed", "tuple"])
def test_gelu_fwd_bwd(B, T, c, dtype, layout):
    o = ops()
    xs = rand5("gelu", B, T, c, dtype)
    gs = rand5("gelu.g", B, T, c, dtype)
    # oracle on the same (possibly bf16-rounded) inputs, evaluated in fp64
    xr = [t.double().requires_grad_(True) for t in xs]
    yr = R.TritonGeluD8()(tuple(xr))
    torch.autograd.backward(yr, [g.double() for g in gs])
    M = B * T
    if layout == "packed":
        x = pack(xs).to(DEV)
        g = pack(gs).to(DEV)
        y = torch.empty_like(x)
        gi = torch.empty_like(x)
        o.gelu_fwd(o.pview(x, c), o.pview(y, c), M, c, dtype, x)
        o.gelu_bwd(o.pview(g, c), o.pview(x, c), o.pview(gi, c), M, c, dtype, x)
        ys, gis = unpack(y, c), unpack(gi, c)
    else:
        xd = tuple(t.to(DEV) for t in xs)
        gd = tuple(t.to(DEV) for t in gs)
        ys = tuple(torch.empty_like(t) for t in xd)
        gis = tuple(torch.empty_like(t) for t in xd)
        xv, k1 = o.tview(xd, c)
        yv, k2 = o.tview(ys, c)
        gv, k3 = o.tview(gd, c)
        ov, k4 = o.tview(gis, c)
        o.gelu_fwd(xv, yv, M, c, dtype, xd[0])
        o.gelu_bwd(gv, xv, ov, M, c, dtype, xd[0])
    rtol, atol = TOLS[dtype]
    for i in range(5):
        close(ys[i], yr[i], rtol, atol, f"gelu fwd irrep {i}")
        close(gis[i], xr[i].grad, rtol, atol, f"gelu bwd irrep {i}")


# ------------------------------------------------------------------------------------- LayerNorm
@pytest.mark.parametrize("B,T,c", SHAPES + [(2, 9, 256)])
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
def test_layernorm_fwd_bwd(B, T, c, out_dtype):
    o = ops()
    xs = rand5("ln", B, T, c)
    ln = cases.fill_parameters(R.LayerNormD8(8 * c)).double()
    xr = [t.double().requires_grad_(True) for t in xs]
    yr = ln(tuple(xr))
    gs = rand5("ln.g", B, T, c, out_dtype)
    torch.autograd.backward(yr, [g.double() for g in gs])
    x = pack(xs).to(DEV)
    sc = ln.scaling
    alpha = [getattr(sc, "alpha_" + n).detach().float().to(DEV) for n in ("A1", "A2", "B1", "B2", "E")]
    beta = sc.beta.detach().float().to(DEV)
    y, stats = o.layernorm_fwd(x, alpha, beta, ln.eps, out_dtype, c)
    rtol, atol = TOLS[out_dtype]
    close(y, pack(yr), rtol, atol, "ln fwd")
    g = pack(gs).to(DEV)
    dres = pack(rand5("ln.res", B, T, c)).to(DEV)
    dx, dal, dbeta = o.layernorm_bwd(g, x, stats, alpha, dres, c)
    close(dx - dres, pack([t.grad for t in xr]), 1e-4, 1e-4, "ln dx")
    for i, n in enumerate(("A1", "A2", "B1", "B2", "E")):
        close(dal[i], getattr(sc, "alpha_" + n).grad, 1e-4, 1e-4, f"ln dalpha {n}")
    close(dbeta, sc.beta.grad, 1e-4, 1e-4, "ln dbeta")
    # no-affine, no-residual variant
    y2, st2 = o.layernorm_fwd(x, None, None, ln.eps, out_dtype, c)
    ln2 = R.LayerNormD8(8 * c, elementwise_affine=False).double()
    close(y2, pack(ln2(tuple(t.double() for t in xs))), rtol, atol, "ln fwd no affine")


@pytest.mark.parametrize("B,T,c", [(64, 257, 160), (3, 50, 32), (2, 9, 96)])
def test_layernorm_bwd_cast_equals_bwd_then_cast(B, T, c):
    """octic_layernorm_d8_bwd_cast = octic_layernorm_d8_bwd followed by octic_cast_rowscale of its dx, bit for bit."""
    o = ops()
    g0 = torch.Generator(device=DEV).manual_seed(77)
    x = torch.randn(B, T, 8 * c, generator=g0, device=DEV) * 2 + 0.3
    g = torch.randn(B, T, 8 * c, generator=g0, device=DEV).bfloat16()
    dres = torch.randn(B, T, 8 * c, generator=g0, device=DEV)
    alpha = [torch.rand(c if i < 4 else 2 * c, generator=g0, device=DEV) + 0.5 for i in range(5)]
    rs = ((torch.arange(B, device=DEV) % 3 != 0).float() / 0.66)
    _, stats = o.layernorm_fwd(x, alpha, None, 1e-5, torch.bfloat16, c)
    for dr, r_ in ((dres, rs), (None, None), (dres, None)):
        dx, dal, dbeta = o.layernorm_bwd(g, x, stats, alpha, dr, c)
        want = o.cast_rowscale(dx, r_, T, torch.bfloat16, c)
        dx2, dal2, dbeta2, gc = o.layernorm_bwd_cast(g, x, stats, alpha, dr, c, r_, T)
        assert torch.equal(dx2, dx) and torch.equal(gc, want) and torch.equal(dbeta2, dbeta)
        for a, b in zip(dal2, dal):
            assert torch.equal(a, b)


# ---------------------------------------------------------------------------------------- Linear
def _linear_ref(xs, W, bias, resid=None, rs=None, cs=None, T=None):
    """fp64 math of the fused linear (octic_hip.h)."""
    ys = []
    for i in range(5):
        y = xs[i].double() @ W[i].double().t()
        if i == 0 and bias is not None:
            y = y + bias.double()
        if cs is not None:
            y = y * cs[i].double()
        if rs is not None:
            shape = (-1,) + (1,) * (y.dim() - 1)
            y = y * rs.double().reshape(shape)
        if resid is not None:
            y = y + resid[i].double()
        ys.append(y)
    return ys


LIN_SHAPES = [(2, 5, 8, 16), (2, 17, 16, 8), (3, 50, 48, 144), (2, 257, 160, 480), (1, 130, 160, 160), (2, 60, 640, 160),
              (2, 197, 128, 384), (1, 197, 512, 128)]      # the last two: ViT-L/16 (c = 128) qkv and fc2 shapes


@pytest.mark.parametrize("B,T,cin,cout", LIN_SHAPES)
@pytest.mark.parametrize("dtype,out_dtype", [(torch.float32, torch.float32), (torch.bfloat16, torch.bfloat16),
                                             (torch.bfloat16, torch.float32)])
@pytest.mark.parametrize("fused", [False, True])
def test_linear_fwd(B, T, cin, cout, dtype, out_dtype, fused):
    o = ops()
    xs = rand5("lin.x", B, T, cin, dtype)
    lin = cases.fill_parameters(R.LinearD8(8 * cin, 8 * cout, bias=True))
    W = [getattr(lin, "lin_" + n).weight.detach().to(dtype) for n in ("A1", "A2", "B1", "B2", "E")]
    bias = lin.lin_A1.bias.detach().float()
    resid = rs = cs = None
    if fused:
        resid = rand5("lin.res", B, T, cout, out_dtype)
        rs = (torch.rand(B, generator=cases._gen("lin.rs")) > 0.3).float() / 0.7
        cs = [0.5 + 0.2 * cases.randn(f"lin.cs{i}", cout if i < 4 else 2 * cout) for i in range(5)]
    want = _linear_ref(xs, W, bias, resid, rs, cs)
    x = pack(xs).to(DEV)
    y = torch.empty((B, T, 8 * cout), dtype=out_dtype, device=DEV)
    Wd = [w.to(DEV).contiguous() for w in W]
    kw = {}
    keep = []
    if fused:
        r = pack(resid).to(DEV)
        keep = [r, rs.to(DEV), [t.to(DEV) for t in cs]]
        kw = dict(resid_v=o.pview(r, cout), rs=keep[1], rps=T, cs5=keep[2])
    o.linear_fwd(o.pview(x, cin), Wd, bias.to(DEV), o.pview(y, cout), B * T, cin, cout, dtype, out_dtype, x, **kw)
    rtol, atol = (1e-4, 1e-4) if dtype == torch.float32 else (2e-2, 2e-2)
    close(y, pack(want), rtol, atol, "linear fwd")


@pytest.mark.parametrize("B,T,cin,cout", LIN_SHAPES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_linear_fwd_tuple_layout_matches_packed(B, T, cin, cout, dtype):
    """The reference's 5 separate tensors go through the same kernel via strides."""
    o = ops()
    xs = tuple(t.to(DEV) for t in rand5("lin.x", B, T, cin, dtype))
    lin = cases.fill_parameters(R.LinearD8(8 * cin, 8 * cout, bias=False))
    Wd = [getattr(lin, "lin_" + n).weight.detach().to(dtype).to(DEV).contiguous() for n in ("A1", "A2", "B1", "B2", "E")]
    ys = tuple(torch.empty(B, T, cout, dtype=dtype, device=DEV) for _ in range(4)) + (
        torch.empty(B, T, 2, 2 * cout, dtype=dtype, device=DEV),)
    xv, k1 = o.tview(xs, cin)
    yv, k2 = o.tview(ys, cout)
    o.linear_fwd(xv, Wd, None, yv, B * T, cin, cout, dtype, dtype, xs[0])
    x = pack(xs)
    y = torch.empty((B, T, 8 * cout), dtype=dtype, device=DEV)
    o.linear_fwd(o.pview(x, cin), Wd, None, o.pview(y, cout), B * T, cin, cout, dtype, dtype, x)
    assert torch.equal(pack(ys), y)


@pytest.mark.parametrize("B,T,cin,cout", LIN_SHAPES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("scaled", [False, True])
def test_linear_wgrad(B, T, cin, cout, dtype, scaled):
    """dW (and the layer-scale chain rule dcs, dbias) vs autograd of the fp64 formula."""
    o = ops()
    xs = rand5("wg.x", B, T, cin, dtype)
    dys = rand5("wg.dy", B, T, cout, dtype)   # = rs * dL/dy, already in compute dtype
    lin = cases.fill_parameters(R.LinearD8(8 * cin, 8 * cout, bias=True))
    names = ("A1", "A2", "B1", "B2", "E")
    W = [getattr(lin, "lin_" + n).weight.detach().double().requires_grad_(True) for n in names]
    bias = lin.lin_A1.bias.detach().double().requires_grad_(True)
    cs = None
    if scaled:
        cs = [(0.5 + 0.2 * cases.randn(f"wg.cs{i}", cout if i < 4 else 2 * cout)).double().requires_grad_(True)
              for i in range(5)]
    ys = _linear_ref(xs, W, bias, None, None, cs)
    torch.autograd.backward(ys, [d.double() for d in dys])
    x, dy = pack(xs).to(DEV), pack(dys).to(DEV)
    M = B * T
    dysum = o.colsum_a1(o.pview(dy, cout), M, cout, dtype, dy)
    close(dysum, dys[0].double().sum((0, 1)), 1e-4, 1e-4, "colsum")
    kw = {}
    if scaled:
        kw = dict(w32=[w.detach().float().to(DEV).contiguous() for w in W], cs5=[t.detach().float().to(DEV) for t in cs],
                  bias=bias.detach().float().to(DEV))
    dw, dcs, dbias = o.linear_wgrad(o.pview(x, cin), o.pview(dy, cout), M, cin, cout, dtype, x, dysum=dysum,
                                    want_bias=True, **kw)
    tol = 1e-4 if dtype == torch.float32 else 1e-4  # inputs are exactly representable; accumulation is f32
    for i in range(5):
        close(dw[i], W[i].grad, tol, tol, f"dW {names[i]}")
        if scaled:
            close(dcs[i], cs[i].grad, tol, tol, f"dcs {names[i]}")
    close(dbias, bias.grad, tol, tol, "dbias")
    if o.wgrad_has_colsum(cin, cout, dtype):
        # the bf16 ring kernel leaves the column sums of dY behind the slabs: no explicit dysum needed
        dw2, dcs2, dbias2 = o.linear_wgrad(o.pview(x, cin), o.pview(dy, cout), M, cin, cout, dtype, x, dysum=None,
                                           want_bias=True, **kw)
        close(dbias2, bias.grad, tol, tol, "dbias from the wgrad kernel's own column sums")
        if scaled:
            close(dcs2[0], cs[0].grad, tol, tol, "dcs A1 from the wgrad kernel's own column sums")


def test_wgrad_is_bitwise_reproducible():
    o = ops()
    B, T, cin, cout = 2, 257, 160, 160
    x = pack(rand5("rep.x", B, T, cin, torch.bfloat16)).to(DEV)
    dy = pack(rand5("rep.dy", B, T, cout, torch.bfloat16)).to(DEV)
    a = o.linear_wgrad(o.pview(x, cin), o.pview(dy, cout), B * T, cin, cout, torch.bfloat16, x)[0]
    b = o.linear_wgrad(o.pview(x, cin), o.pview(dy, cout), B * T, cin, cout, torch.bfloat16, x)[0]
    assert all(torch.equal(p, q) for p, q in zip(a, b))


# --------------------------------------------------------------------------------- small kernels
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
def test_cast_rowscale(out_dtype):
    o = ops()
    B, T, c = 3, 17, 48
    x = pack(rand5("cast", B, T, c)).to(DEV)
    rs = torch.tensor([0.0, 2.0, 1.0], device=DEV)
    y = o.cast_rowscale(x, rs, T, out_dtype, c)
    want = (x * rs.view(B, 1, 1)).to(out_dtype)
    assert torch.equal(y, want)
    assert torch.equal(o.cast_rowscale(x, None, 1, out_dtype, c), x.to(out_dtype))


@pytest.mark.parametrize("B,T,H,c", [(2, 17, 2, 8), (2, 9, 4, 16), (2, 50, 6, 48), (2, 257, 16, 160), (1, 5, 8, 40)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_pack_unpack_heads(B, T, H, c, dtype):
    o = ops()
    qkvs = rand5("heads", B, T, 3 * c, dtype)
    q, k, v = R.pack_heads(qkvs, H)
    got = o.pack_heads(pack(qkvs).to(DEV), B, T, H, c, 3)
    for i, w in enumerate((q, k, v)):
        assert torch.equal(got[i].cpu(), w), f"pack s={i}"
    # unpack is the exact inverse (n_s = 3, three separately allocated tensors) and matches the oracle for n_s = 1
    back = o.unpack_heads([g.clone() for g in got], B, T, H, c)
    assert torch.equal(back.cpu(), pack(qkvs))
    out = o.unpack_heads([got[0]], B, T, H, c)
    assert torch.equal(out.cpu(), pack(R.unpack_heads(q)))


def test_linear_prep_matches_torch():
    o = ops()
    cin, cout = 48, 144
    lin = cases.fill_parameters(R.LinearD8(8 * cin, 8 * cout, bias=False))
    w = [getattr(lin, "lin_" + n).weight.detach().to(DEV).contiguous() for n in ("A1", "A2", "B1", "B2", "E")]
    cs = [(0.5 + 0.2 * cases.randn(f"prep.cs{i}", cout if i < 4 else 2 * cout)).to(DEV) for i in range(5)]
    for dtype in (torch.float32, torch.bfloat16):
        wb, wt = o.linear_prep(w, cs, cin, cout, dtype)
        for i in range(5):
            assert torch.equal(wb[i], w[i].to(dtype))
            assert torch.equal(wt[i], (w[i] * cs[i][:, None]).t().to(dtype))
        wb2, wt2 = o.linear_prep(w, None, cin, cout, dtype)
        assert all(torch.equal(a, b.t().to(dtype)) for a, b in zip(wt2, w))


@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
def test_handoff_and_power_spectrum(out_dtype):
    o = ops()
    B, T, c = 2, 17, 48
    xs = rand5("hand", B, T, c)
    x = pack(xs).to(DEV)
    want = torch.cat(R.convert_5tuple_to_8tuple(xs), dim=-1)
    got = o.handoff_cat_fwd(x, c, out_dtype)
    assert torch.equal(got.cpu(), want.to(out_dtype))
    dd = cases.randn("hand.dd", B, T, 8 * c)
    xr = [t.clone().requires_grad_(True) for t in xs]
    torch.cat(R.convert_5tuple_to_8tuple(tuple(xr)), dim=-1).backward(dd)
    assert torch.equal(o.handoff_cat_bwd(dd.to(DEV), c).cpu(), pack([t.grad for t in xr]))
    # power spectrum
    inv = R.PowerSpectrumInvariant(8 * c)
    xr = [t.clone().double().requires_grad_(True) for t in xs]
    yr = inv(tuple(xr))
    d6 = cases.randn("hand.d6", B, T, 6 * c)
    yr.backward(d6.double())
    rtol, atol = TOLS[out_dtype]
    close(o.power_spectrum_fwd(x, c, out_dtype), yr, rtol, atol, "power spectrum fwd")
    close(o.power_spectrum_bwd(d6.to(DEV), x, c), pack([t.grad for t in xr]), 2e-5, 2e-5, "power spectrum bwd")


# (14, 16, 1280) is the bench shape: ViT-H/14 at 224x224 (K = 588 padded to 592)
@pytest.mark.parametrize("p,G,D", [(4, 4, 64), (14, 2, 64), (16, 14, 384), (14, 16, 1280)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_lift_gemm_matches_patch_embed(p, G, D, dtype):
    """im2col + one GEMM against the symmetry-expanded kernels == the 8 strided convs of PatchEmbedD8."""
    o = ops()
    B, c = 2, D // 8
    pe = cases.fill_parameters(R.PatchEmbedD8(img_size=p * G, patch_size=p, embed_dim=D))
    img = cases.randn("lift.img", B, 3, p * G, p * G)
    want = pack(pe(img))  # [B, G*G, D]
    l8 = pe.lift8
    ws = [l8.conv_A1.expand_weight(), l8.conv_A2.expand_weight(), l8.conv_B1.expand_weight(), l8.conv_B2.expand_weight()]
    el, er = l8.conv_E_left.expand_weight(), l8.conv_E_right.expand_weight()
    rot = lambda k: k.rot90(k=1, dims=(-2, -1))
    wfull = torch.cat(ws + [el, er, rot(el), rot(er)], dim=0).flatten(1)  # packed channel order
    K = 3 * p * p
    Kpad = (K + 7) // 8 * 8
    wpad = torch.zeros(D, Kpad)
    wpad[:, :K] = wfull.detach()
    bias_full = torch.zeros(D)
    bias_full[:c] = l8.conv_A1.bias.detach()
    pos = cases.randn("lift.pos", G * G, D)
    patches = o.im2col(img.to(DEV), p, Kpad, dtype)
    tok0 = 1
    out = torch.full((B, tok0 + G * G, D), 7.0, device=DEV)
    o.lift_gemm(patches, wpad.to(dtype).to(DEV), bias_full.to(DEV), pos.to(DEV), out, B, G * G, tok0, Kpad, D)
    assert torch.all(out[:, 0] == 7.0)  # cls row untouched
    rtol, atol = (1e-4, 1e-4) if dtype == torch.float32 else (2e-2, 2e-2)
    close(out[:, 1:], want + pos, rtol, atol, "lift gemm")
    # weight gradient
    dout = cases.randn("lift.dout", B * G * G, D).to(dtype)
    dw = o.lift_wgrad(patches, dout.to(DEV), Kpad, D)
    ref = dout.double().t() @ patches.double().cpu()
    close(dw, ref, 1e-4, 1e-4, "lift wgrad")


def test_gelu_d8_is_a_dispatcher_op_that_torch_compile_traces_through():
    """Round 4 (review item 8): the reference's one custom-op boundary (TritonGeluD8Function, d8_gelu.py:456-478) as
    torch.library ops with schema, fake kernel and autograd formula (octic_vits_amd/custom_ops.py).  (a) torch.library.opcheck
    (schema, fake-tensor agreement, autograd registration, AOT dispatch) on GPU tensors; (b) torch.compile with
    fullgraph=True - i.e. ZERO graph breaks - of the module the reference compiles, on the 5-tensor and the packed form,
    forward and backward BITWISE equal to the eager op (backend aot_eager: the op itself is the HIP kernel either way)."""
    import torch
    from octic_vits_amd import custom_ops  # noqa: F401
    from octic_vits_amd.d8_layers import TritonGeluD8
    from octic_vits_amd.functional import Octic
    torch.manual_seed(3)
    c = 16
    for dt in (torch.float32, torch.bfloat16):
        x = torch.randn(3, 7, 8 * c, device=DEV).to(dt)
        torch.library.opcheck(torch.ops.octic.gelu_d8, (x.clone().requires_grad_(True), c))
        xs5 = [torch.randn(3, 7, c, device=DEV).to(dt).requires_grad_(True) for _ in range(4)] + \
              [torch.randn(3, 7, 2, 2 * c, device=DEV).to(dt).requires_grad_(True)]
        torch.library.opcheck(torch.ops.octic.gelu_d8_tuple, tuple(xs5))

        mod = TritonGeluD8()
        comp = torch.compile(mod, backend="aot_eager", fullgraph=True)      # fullgraph: any graph break raises
        gs = [torch.randn_like(t) for t in xs5]
        ye = mod(tuple(xs5))
        ge = torch.autograd.grad(ye, xs5, gs)
        yc = comp(tuple(xs5))
        gc = torch.autograd.grad(yc, xs5, gs)
        for a_, b_ in zip(list(ye) + list(ge), list(yc) + list(gc)):
            assert torch.equal(a_, b_)

        def packed(t):
            return torch.ops.octic.gelu_d8(t, c) * 2.0

        xp = x.clone().requires_grad_(True)
        ye = packed(xp)
        (ge,) = torch.autograd.grad(ye, xp, torch.ones_like(ye))
        yc = torch.compile(packed, backend="aot_eager", fullgraph=True)(xp)
        (gc,) = torch.autograd.grad(yc, xp, torch.ones_like(yc))
        assert torch.equal(ye, yc) and torch.equal(ge, gc)
        # and the module on the engine's native container runs through the same op
        y_o = mod(Octic(x, c))
        assert torch.equal(y_o.packed, torch.ops.octic.gelu_d8(x, c))


def test_sample_blocks_gather_and_scatter():
    """octic_sample_blocks against index_select / index_copy_ for whole samples (f32 and bf16 blocks, odd counts)."""
    from octic_vits_amd import ops as o
    g = torch.Generator(device="cuda").manual_seed(2)
    for dtype, shape in ((torch.float32, (13, 257, 1280)), (torch.bfloat16, (9, 33, 256)), (torch.float32, (5, 4))):
        full = torch.randn(shape, device="cuda", generator=g).to(dtype)
        idx = torch.randperm(shape[0], device="cuda", generator=g)[: max(1, shape[0] // 2 + 1)].sort().values
        got = o.gather_samples(full, idx)
        assert torch.equal(got, full.index_select(0, idx))
        comp = torch.randn((idx.numel(),) + shape[1:], device="cuda", generator=g).to(dtype)
        assert o.sample_blocks_ok(full, comp, idx)
        want = full.clone().index_copy_(0, idx, comp)
        assert torch.equal(o.scatter_samples_(full, idx, comp), want)
    assert not o.sample_blocks_ok(torch.zeros(4, 3, device="cuda"), torch.zeros(2, 3, device="cuda"),
                                  torch.tensor([0, 1], device="cuda"))            # 12-byte blocks: the ATen path
