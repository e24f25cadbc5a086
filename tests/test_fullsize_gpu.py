"""Parity at the BASELINE configs[1] size: B = 64, T = 257, c = 160 (M = 16 448 token rows, ~2000 workgroups per
launch, per-irrep wgrad row splits, the XCD remap with nwg % 8 != 0) — what bench.py executes, checked.

Every check has two legs:
  * the oracle (oracle/octic_ref.py, fp64 on the host) on a strided sample of batch elements / heads — rows of
    different samples never interact, so sample b of the full-size launch must equal the oracle on sample b alone;
  * a full-tensor comparison against a second formulation (plain torch ops on the device in f32/f64: matmul,
    softmax, index permutations) plus size-independent properties (pack/unpack round trip, bitwise-reproducible
    weight gradients, equivariance under the 8 group elements).

Tolerances: f32 path 1e-4 of the output scale (K <= 1280 fma chains); bf16 path 2e-2 of the output scale against
references evaluated on the SAME bf16-rounded operands (3e-2 for attention gradients, as in the small-shape tests).
"""
import math

import pytest
import torch

import cases
from oracle import octic_ref as R

pytestmark = pytest.mark.gpu

DEV = "cuda"
B, T, C, H = 64, 257, 160, 16
SAMPLE_B = [0, 21, 42, 63]          # batch elements checked against the host oracle
NAMES = ("A1", "A2", "B1", "B2", "E")


def ops():
    from octic_vits_amd import ops as o
    return o


def dev_randn(tag, *shape, dtype=torch.float32, scale=1.0):
    g = torch.Generator(device=DEV).manual_seed(cases._gen(tag).initial_seed())
    return (torch.randn(*shape, generator=g, device=DEV) * scale).to(dtype)


def unpack(t, c):
    return (t[..., :c], t[..., c:2 * c], t[..., 2 * c:3 * c], t[..., 3 * c:4 * c], t[..., 4 * c:].unflatten(-1, (2, 2 * c)))


def pack(xs):
    return torch.cat([xs[0], xs[1], xs[2], xs[3], xs[4].flatten(-2)], dim=-1).contiguous()


def close(got, want, tol, msg):
    got, want = got.detach().double(), want.detach().double().to(got.device)
    scale = max(1.0, float(want.abs().max()))
    err = float((got - want).abs().max())
    assert err <= tol * scale, f"{msg}: max err {err:.3e} > {tol:g} x scale {scale:.3g}"


def packed_tokens(tag, c, dtype, offset=True):
    """[B,T,8c] packed rows with a per-row offset (LayerNorm means are exercised, test_equivariance.py:124-127)."""
    x = dev_randn(tag, B, T, 8 * c)
    if offset:
        x = x + dev_randn(tag + ".off", B, T, 8, 1).expand(B, T, 8, c).reshape(B, T, 8 * c)
    return x.to(dtype).contiguous()


# ---------------------------------------------------------------------------------------- linear
def _weights(tag, cin, cout, dtype):
    lin = cases.fill_parameters(R.LinearD8(8 * cin, 8 * cout, bias=True), salt=tag)
    W = [getattr(lin, "lin_" + n).weight.detach().to(dtype).to(DEV).contiguous() for n in NAMES]
    return W, lin.lin_A1.bias.detach().float().to(DEV)


def _linear_second_formulation(x, W, bias, cin, resid=None, rs=None, cs=None):
    """Full-tensor device reference: five torch matmuls in fp64 on the (already rounded) operands."""
    xs = unpack(x.double(), cin)
    ys = []
    for i in range(5):
        y = xs[i] @ W[i].double().t()
        if i == 0 and bias is not None:
            y = y + bias.double()
        if cs is not None:
            y = y * cs[i].double()
        ys.append(y)
    y = pack(ys)
    if rs is not None:
        y = y * rs.double().view(-1, 1, 1)
    if resid is not None:
        y = y + resid.double()
    return y


FULL_LIN = [(160, 480), (160, 160), (160, 640), (640, 160), (480, 160)]   # qkv, proj, fc1, fc2 (= dgrad fc1), dgrad qkv


@pytest.mark.parametrize("cin,cout", FULL_LIN)
@pytest.mark.parametrize("dtype,out_dtype,fused", [(torch.bfloat16, torch.bfloat16, False),
                                                   (torch.bfloat16, torch.float32, True),
                                                   (torch.float32, torch.float32, True)])
def test_linear_fwd_full_size(cin, cout, dtype, out_dtype, fused):
    o = ops()
    x = packed_tokens(f"fs.lin.x{cin}", cin, dtype)
    W, bias = _weights(f"fs.lin{cin}x{cout}", cin, cout, dtype)
    resid = rs = cs = None
    kw = {}
    if fused:
        resid = packed_tokens(f"fs.lin.res{cout}", cout, out_dtype)
        rs = ((torch.rand(B, generator=cases._gen("fs.lin.rs")) > 0.5).float() / 0.5).to(DEV)
        cs = [(0.5 + 0.2 * cases.randn(f"fs.lin.cs{i}", cout if i < 4 else 2 * cout)).to(DEV) for i in range(5)]
        kw = dict(resid_v=o.pview(resid, cout), rs=rs, rps=T, cs5=cs)
    y = torch.empty((B, T, 8 * cout), dtype=out_dtype, device=DEV)
    o.linear_fwd(o.pview(x, cin), W, bias, o.pview(y, cout), B * T, cin, cout, dtype, out_dtype, x, **kw)
    tol = 1e-4 if dtype == torch.float32 else 2e-2
    # leg 1: host oracle (reference LinearD8 restatement in fp64) on the sampled batch elements
    lin = R.LinearD8(8 * cin, 8 * cout, bias=True).double()
    with torch.no_grad():
        for i, n in enumerate(NAMES):
            getattr(lin, "lin_" + n).weight.copy_(W[i].double().cpu())
        lin.lin_A1.bias.copy_(bias.double().cpu())
        for b in SAMPLE_B:
            ys = lin(tuple(t.cpu() for t in unpack(x[b].double(), cin)))
            want = pack(ys)
            if cs is not None:
                want = want * torch.cat([c.double().cpu() for c in cs] + [cs[4].double().cpu()])
            if rs is not None:
                want = want * float(rs[b])
            if resid is not None:
                want = want + resid[b].double().cpu()
            close(y[b].cpu(), want, tol, f"linear {cin}->{cout} sample {b} vs oracle")
    # leg 2: the whole tensor against torch matmuls on the device
    close(y, _linear_second_formulation(x, W, bias, cin, resid, rs, cs), tol, f"linear {cin}->{cout} full tensor")


@pytest.mark.parametrize("cin,cout", [(160, 480), (160, 160), (160, 640), (640, 160)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_linear_wgrad_full_size(cin, cout, dtype):
    """dW_i = dY_i^T X_i over all 16 448 (E: 32 896) rows, dbias = column sums; vs fp64 matmuls of the same operands
    (device) and the host oracle's autograd on the sampled batch elements (their contribution alone)."""
    o = ops()
    x = packed_tokens(f"fs.wg.x{cin}", cin, dtype)
    dy = packed_tokens(f"fs.wg.dy{cout}", cout, dtype, offset=False)
    M = B * T
    dw, _, dbias = o.linear_wgrad(o.pview(x, cin), o.pview(dy, cout), M, cin, cout, dtype, x,
                                  dysum=None if o.wgrad_has_colsum(cin, cout, dtype) else o.colsum_a1(o.pview(dy, cout), M, cout, dtype, dy),
                                  want_bias=True)
    xs, dys = unpack(x.double(), cin), unpack(dy.double(), cout)
    for i in range(5):
        want = dys[i].reshape(-1, dys[i].shape[-1]).t() @ xs[i].reshape(-1, xs[i].shape[-1])
        close(dw[i], want, 2e-4, f"dW {NAMES[i]} full")
    close(dbias, dys[0].sum((0, 1)), 2e-4, "dbias full")
    # bitwise reproducible at full size (deterministic slab order, no atomics)
    dw2, _, _ = o.linear_wgrad(o.pview(x, cin), o.pview(dy, cout), M, cin, cout, dtype, x, want_bias=False)
    assert all(torch.equal(a, b) for a, b in zip(dw, dw2)), "wgrad not bitwise reproducible at full size"
    # oracle leg: gradient of sum(dy * LinearD8(x)) on the sampled batch elements only, by linearity equal to the HIP
    # kernel run on exactly those rows
    sub_x = x[SAMPLE_B].contiguous()
    sub_dy = dy[SAMPLE_B].contiguous()
    dws, _, dbs = o.linear_wgrad(o.pview(sub_x, cin), o.pview(sub_dy, cout), len(SAMPLE_B) * T, cin, cout, dtype, sub_x,
                                 dysum=None if o.wgrad_has_colsum(cin, cout, dtype) else
                                 o.colsum_a1(o.pview(sub_dy, cout), len(SAMPLE_B) * T, cout, dtype, sub_dy), want_bias=True)
    lin = R.LinearD8(8 * cin, 8 * cout, bias=True).double()
    ys = lin(tuple(t.cpu() for t in unpack(sub_x.double(), cin)))
    torch.autograd.backward(ys, [t.cpu() for t in unpack(sub_dy.double(), cout)])
    for i, n in enumerate(NAMES):
        close(dws[i].cpu(), getattr(lin, "lin_" + n).weight.grad, 2e-4, f"dW {n} sampled rows vs oracle autograd")
    close(dbs.cpu(), lin.lin_A1.bias.grad, 2e-4, "dbias sampled rows vs oracle autograd")


# ------------------------------------------------------------------------------------- LayerNorm
@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float32])
def test_layernorm_full_size(out_dtype):
    o = ops()
    c = C
    x = packed_tokens("fs.ln.x", c, torch.float32)
    ln = cases.fill_parameters(R.LayerNormD8(8 * c)).double()
    sc = ln.scaling
    alpha = [getattr(sc, "alpha_" + n).detach().float().to(DEV) for n in NAMES]
    beta = sc.beta.detach().float().to(DEV)
    y, stats = o.layernorm_fwd(x, alpha, beta, ln.eps, out_dtype, c)
    g = packed_tokens("fs.ln.g", c, out_dtype, offset=False)
    dres = packed_tokens("fs.ln.res", c, torch.float32, offset=False)
    dx, dal, dbeta = o.layernorm_bwd(g, x, stats, alpha, dres, c)
    tol = 2e-5 if out_dtype == torch.float32 else 2e-2
    # leg 1: oracle forward/backward on the sampled batch elements (parameter gradients: their contribution is checked
    # through the second formulation below, which sums over every row)
    for b in SAMPLE_B:
        xr = [t.cpu().double().requires_grad_(True) for t in unpack(x[b], c)]
        yr = ln(tuple(xr))
        torch.autograd.backward(yr, [t.cpu().double() for t in unpack(g[b], c)])
        close(y[b].cpu(), pack(yr), tol, f"ln fwd sample {b}")
        close((dx[b] - dres[b]).cpu(), pack([t.grad for t in xr]), 1e-4, f"ln dx sample {b}")
    # leg 2: the math spec written with plain tensor ops on the device in fp64, whole tensor
    xd = x.double().view(B, T, 8, c)                       # octets: A1 A2 B1 B2 | E0a E0b | E1a E1b
    seg = torch.stack([xd[:, :, 0], xd[:, :, 1], xd[:, :, 2], xd[:, :, 3]], 2)         # [B,T,4,c]
    e = xd[:, :, 4:].reshape(B, T, 2, 2 * c)
    mu1, mue = seg.mean(-1, keepdim=True), e.mean(-1, keepdim=True)
    S = ((seg - mu1) ** 2).mean(-1).sum(-1) + 0.5 * ((e - mue) ** 2).mean(-1).sum(-1) + ln.eps
    std = (math.sqrt(2.0) / 4.0) * S.sqrt()
    xh = torch.cat([((seg - mu1) / std[..., None, None]).reshape(B, T, 4 * c),
                    ((e - mue) / std[..., None, None]).reshape(B, T, 4 * c)], -1)
    ap = torch.cat([a.double() for a in alpha] + [alpha[4].double()])
    want = xh * ap
    want[..., :c] += beta.double()
    close(y, want, tol, "ln fwd full tensor")
    gd = g.double()
    dal_want = (gd * xh).sum((0, 1))
    for i in range(4):
        close(dal[i], dal_want[i * c:(i + 1) * c], 2e-4, f"ln dalpha {NAMES[i]} full")
    close(dal[4], dal_want[4 * c:6 * c] + dal_want[6 * c:], 2e-4, "ln dalpha E full")
    close(dbeta, gd[..., :c].sum((0, 1)), 2e-4, "ln dbeta full")


# ------------------------------------------------------------------------------------------ GELU
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_gelu_full_size(dtype):
    """The MLP hidden width: c = 640 (84.2 M elements per tensor at B = 64)."""
    o = ops()
    c = 4 * C
    x = packed_tokens("fs.gelu.x", c, dtype, offset=False)
    g = packed_tokens("fs.gelu.g", c, dtype, offset=False)
    y, gi = torch.empty_like(x), torch.empty_like(x)
    M = B * T
    o.gelu_fwd(o.pview(x, c), o.pview(y, c), M, c, dtype, x)
    o.gelu_bwd(o.pview(g, c), o.pview(x, c), o.pview(gi, c), M, c, dtype, x)
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    for b in SAMPLE_B[:2]:
        xr = [t.cpu().double().requires_grad_(True) for t in unpack(x[b], c)]
        yr = R.TritonGeluD8()(tuple(xr))
        torch.autograd.backward(yr, [t.cpu().double() for t in unpack(g[b], c)])
        close(y[b].cpu(), pack(yr), tol, f"gelu fwd sample {b}")
        close(gi[b].cpu(), pack([t.grad for t in xr]), tol, f"gelu bwd sample {b}")
    # second formulation: the 8x8 orthonormal matrix F = (sqrt2/4) S+ as one matmul over the component axis (fp32 on
    # the device: 84 M elements x 8 components in fp64 would not fit comfortably next to the operands)
    Sp = torch.tensor([[1, 1, 1, 1, 1, 1, 1, 1], [1, 1, 1, 1, -1, -1, -1, -1], [1, -1, 1, -1, 1, -1, 1, -1],
                       [1, -1, 1, -1, -1, 1, -1, 1], [1, 1, -1, -1, -1, -1, 1, 1], [1, -1, -1, 1, 1, -1, -1, 1],
                       [1, -1, -1, 1, -1, 1, 1, -1], [-1, -1, 1, 1, -1, -1, 1, 1]], dtype=torch.float32, device=DEV)
    F = Sp * (math.sqrt(2.0) / 4.0)

    def comps(t):     # packed [n,T,8c] -> [n,T,8,c] in the oracle's component order x0..x7
        v = t.float().view(t.shape[0], T, 8, c)
        return torch.stack([v[:, :, 0], v[:, :, 1], v[:, :, 2], v[:, :, 3], v[:, :, 4], v[:, :, 6], v[:, :, 5], v[:, :, 7]], 2)

    for b0 in range(0, B, 16):           # in slabs of 16 samples to bound memory
        sl = slice(b0, b0 + 16)
        v = comps(x[sl])
        r = torch.einsum("ji,btic->btjc", F.t(), v)          # F^-1 = F^T
        yy = torch.einsum("ji,btic->btjc", F, torch.nn.functional.gelu(r))
        yy = torch.stack([yy[:, :, 0], yy[:, :, 1], yy[:, :, 2], yy[:, :, 3], yy[:, :, 4], yy[:, :, 6], yy[:, :, 5],
                          yy[:, :, 7]], 2).reshape(16, T, 8 * c)
        close(y[sl], yy, max(tol, 1e-5), f"gelu fwd full tensor slab {b0}")


# --------------------------------------------------------------------------------- head packing
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_pack_unpack_heads_full_size(dtype):
    o = ops()
    qkv = packed_tokens("fs.heads", 3 * C, dtype, offset=False)
    got = o.pack_heads(qkv, B, T, H, C, 3)
    # oracle on the sampled batch elements (bit exact: a permutation)
    for b in SAMPLE_B:
        q, k, v = R.pack_heads(tuple(t.cpu() for t in unpack(qkv[b:b + 1], 3 * C)), H)
        for i, w in enumerate((q, k, v)):
            assert torch.equal(got[i][b:b + 1].cpu(), w), f"pack s={i} sample {b}"
    # second formulation on the device, whole tensor: index arithmetic of SURVEY 10.4
    w1 = C // H
    xs = unpack(qkv, 3 * C)
    for s in range(3):
        parts = [xs[i][..., s * C:(s + 1) * C].reshape(B, T, H, w1) for i in range(4)]
        e = xs[4][..., s * 2 * C:(s + 1) * 2 * C].reshape(B, T, 2, H, 2 * w1)
        parts += [e[:, :, 0], e[:, :, 1]]
        want = torch.cat(parts, -1).permute(0, 2, 1, 3)
        assert torch.equal(got[s], want), f"pack s={s} full tensor"
    # round trip (size-independent property)
    assert torch.equal(o.unpack_heads([t.clone() for t in got], B, T, H, C), qkv)


# ------------------------------------------------------------------------------------- attention
def test_attention_full_size():
    """(B,H,T,hd) = (64,16,257,80): fp32 softmax attention on the device for every head; fp64 on the host for a
    sample of (batch, head) pairs."""
    from octic_vits_amd.functional import AttnFn
    hd = 8 * C // H
    q, k, v = (dev_randn(f"fs.attn.{n}", B, H, T, hd, dtype=torch.bfloat16).requires_grad_(True) for n in "qkv")
    do = dev_randn("fs.attn.do", B, H, T, hd, dtype=torch.bfloat16)
    scale = hd ** -0.5
    out = AttnFn.apply(q, k, v, scale)
    out.backward(do)
    qf, kf, vf = (t.detach().float().requires_grad_(True) for t in (q, k, v))
    ref = torch.softmax((qf @ kf.transpose(-1, -2)) * scale, dim=-1) @ vf
    ref.backward(do.float())
    close(out, ref, 2e-2, "attention fwd full tensor")
    for name, got, want in (("dq", q.grad, qf.grad), ("dk", k.grad, kf.grad), ("dv", v.grad, vf.grad)):
        close(got, want, 3e-2, f"attention {name} full tensor")
    for b, h in ((0, 0), (21, 5), (42, 11), (63, 15)):
        qd, kd, vd = (t[b, h].detach().double().cpu().requires_grad_(True) for t in (q, k, v))
        r = torch.softmax((qd @ kd.t()) * scale, dim=-1) @ vd
        r.backward(do[b, h].double().cpu())
        close(out[b, h].cpu(), r, 2e-2, f"attention fwd ({b},{h}) fp64")
        close(q.grad[b, h].cpu(), qd.grad, 3e-2, f"attention dq ({b},{h}) fp64")
        close(k.grad[b, h].cpu(), kd.grad, 3e-2, f"attention dk ({b},{h}) fp64")
        close(v.grad[b, h].cpu(), vd.grad, 3e-2, f"attention dv ({b},{h}) fp64")


# ----------------------------------------------------------------------------- block, full size
def _block(drop_path=0.0):
    from octic_vits_amd.d8_layers import Layer_scale_init_BlockD8
    blk = Layer_scale_init_BlockD8(8 * C, H, qkv_bias=True, drop_path=drop_path)
    ref = cases.fill_parameters(R.Layer_scale_init_BlockD8(8 * C, H, qkv_bias=True, drop_path=drop_path), salt="fs.blk")
    blk.load_state_dict(ref.state_dict(), strict=True)
    return blk.to(DEV), ref


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_block_full_size_sampled_vs_oracle(mode):
    """One DeiT-III octic ViT-H block, forward + backward at B = 64: the sampled batch elements must equal the oracle
    block run on those samples alone (f32: 1e-4; bf16 autocast: 3e-2 of scale vs the f32 oracle with the same weights)."""
    from octic_vits_amd.functional import Octic
    blk, ref = _block()
    x = packed_tokens("fs.blk.x", C, torch.float32).requires_grad_(True)
    cot = packed_tokens("fs.blk.cot", C, torch.float32, offset=False)
    if mode == "bf16":
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = blk(Octic(x, C))
    else:
        y = blk(Octic(x, C))
    y.packed.backward(cot)
    tol = 1e-4 if mode == "f32" else 3e-2
    for b in SAMPLE_B[:2]:
        xr = [t.detach().cpu().clone().requires_grad_(True) for t in unpack(x[b:b + 1], C)]
        yr = ref(tuple(xr))
        torch.autograd.backward(yr, [t.cpu() for t in unpack(cot[b:b + 1], C)])
        close(y.packed[b:b + 1].cpu(), pack(yr), tol, f"block fwd sample {b}")
        close(x.grad[b:b + 1].cpu(), pack([t.grad for t in xr]), tol, f"block dx sample {b}")


def test_block_full_size_equivariance_bf16():
    """Size-independent property at full size: the block commutes with the isotypic action of every group element
    (no cls token: T = 256; bf16 autocast, so the tolerance is bf16's, 3e-2 of scale; the f32 path is held to the
    reference's 1e-6-class tolerances in test_equivariance_gpu.py)."""
    from octic_vits_amd.functional import Octic
    blk, _ = _block()
    x = dev_randn("fs.eq.x", B, 256, 8 * C)
    xs = unpack(x, C)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        base = blk(Octic(x.contiguous(), C))
        for g in ("r", "m", "rr", "mr", "rrr", "mrr", "mrrr"):
            gx = R.spatial_and_isotypic_group_action(g, R.convert_5tuple_to_8tuple(xs))
            out = blk(R.convert_8tuple_to_5tuple(gx))
            want = R.convert_8tuple_to_5tuple(R.spatial_and_isotypic_group_action(g, R.convert_5tuple_to_8tuple(tuple(base))))
            close(pack(tuple(out)), pack(want), 3e-2, f"block equivariance under {g}")
