"""W-stationary LinearD8 kernel (csrc/gemm_wreg.hip: bf16 operands, k-chunk = cin <= 160) through the C ABI against an
fp64 restatement of LinearD8.forward (reference octic_vits/d8_layers.py:104-130: four one-dimensional irreps
x_g W_g^T (+ bias on A1), the two-dimensional irrep as [M, 2, 2cin] @ W_E^T) with the fused tail
resid + rs[sample] * cs * (.) of the residual blocks (d8_layers.py:484-498), on the same bf16-rounded operands.
Shapes: ragged row tails (M % 32 != 0), column tails (cout % 16 != 0), every swizzle class of the staged X rows
(cin / 8 = 4, 8, 12, 16, 20 chunks), both k-step paths (cin = 160: pipelined; cin < 160: guarded), ViT-H sizes.
Tolerances: f32 output 2e-5 of the output scale (f32 accumulation of <= 320 bf16 products), bf16 output 1e-2 (one
rounding to 8 mantissa bits).  The ring kernel (octic_route_override, OCTIC_ROUTE_LINEAR_RING) must agree with it to the same tolerances, and
exactly where no bias is folded into the accumulator."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
bf, f32 = torch.bfloat16, torch.float32


def _ref(x, w, bias, cin, cout, resid, cs, rs, rps):
    M = x.shape[0]
    xs = x.double()
    outs = []
    for g in range(4):
        y = xs[:, g * cin:(g + 1) * cin] @ w[g].double().t()
        if g == 0 and bias is not None:
            y = y + bias.double()
        if cs is not None:
            y = y * cs[g].double()
        outs.append(y)
    e = xs[:, 4 * cin:].reshape(M, 2, 2 * cin) @ w[4].double().t()
    if cs is not None:
        e = e * cs[4].double()
    outs.append(e.reshape(M, 4 * cout))
    y = torch.cat(outs, dim=1)
    if rs is not None:
        y = y * rs.double()[torch.arange(M, device=x.device) // rps][:, None]
    if resid is not None:
        y = y + resid.double()
    return y


def _case(M, cin, cout, out_dt, fused, bias_on, seed=0):
    from octic_vits_amd import _lib, ops
    g = torch.Generator(device=DEV).manual_seed(1000 * seed + M + cin + cout)
    rn = lambda *s: torch.randn(*s, generator=g, device=DEV)
    x = rn(M, 8 * cin).to(bf)
    w = [(rn(cout, cin) * cin ** -0.5).to(bf) for _ in range(4)] + [(rn(2 * cout, 2 * cin) * (2 * cin) ** -0.5).to(bf)]
    bias = rn(cout) if bias_on else None
    resid = cs = rs = None
    rps = 1
    if fused:
        resid = rn(M, 8 * cout).to(out_dt)
        cs = [torch.rand(cout, generator=g, device=DEV) + 0.5 for _ in range(4)] + [torch.rand(2 * cout, generator=g, device=DEV) + 0.5]
        rps = 37 if M % 37 == 0 else (257 if M % 257 == 0 else M)
        rs = torch.rand((M + rps - 1) // rps, generator=g, device=DEV) + 0.5
    want = _ref(x, w, bias, cin, cout, resid, cs, rs, rps)
    outs = {}
    L = _lib.lib()
    try:
        for off in (0, 1):
            _lib.route_override(_lib.ROUTE_LINEAR_RING, off)
            y = torch.full((M, 8 * cout), float("nan"), device=DEV, dtype=out_dt)
            ops.linear_fwd(ops.pview(x, cin), w, bias, ops.pview(y, cout), M, cin, cout, bf, out_dt, x,
                           resid_v=ops.pview(resid, cout) if fused else None, rs=rs, rps=rps, cs5=cs)
            torch.cuda.synchronize()
            outs[off] = y
    finally:
        _lib.route_override(_lib.ROUTE_LINEAR_RING, 0)
    return outs, want


SHAPES = [(96, 32, 32), (77, 64, 24), (500, 96, 40), (640, 128, 128), (333, 160, 160), (1001, 160, 480), (1028, 160, 640),
          (33, 32, 8), (16448, 160, 480), (16448, 160, 640), (16448, 160, 160)]


@pytest.mark.parametrize("M,cin,cout", SHAPES)
@pytest.mark.parametrize("out_dt,fused", [(bf, False), (f32, True), (bf, True), (f32, False)])
def test_wreg_matches_linear_d8(M, cin, cout, out_dt, fused):
    outs, want = _case(M, cin, cout, out_dt, fused, bias_on=True)
    scale = max(1.0, float(want.abs().max()))
    tol = (1e-2 if out_dt == bf else 2e-5) * scale
    for off, name in ((0, "W-stationary"), (1, "ring")):
        y = outs[off]
        assert not torch.isnan(y).any(), f"{name}: output rows / columns left unwritten"
        err = float((y.double() - want).abs().max())
        assert err <= tol, f"{name} kernel: max err {err:.3e} > {tol:.3e}"


@pytest.mark.parametrize("M,cin,cout", [(333, 160, 160), (16448, 160, 640), (500, 96, 40)])
def test_wreg_and_ring_agree_bitwise_without_bias(M, cin, cout):
    """Same MFMA instruction, same k order, f32 accumulators: without a bias (the W-stationary kernel starts its
    accumulators from the bias, the ring kernel adds it afterwards) the two kernels must produce identical bf16 rows."""
    outs, _ = _case(M, cin, cout, bf, False, bias_on=False, seed=3)
    assert torch.equal(outs[0], outs[1])


def test_wreg_is_the_routed_kernel_for_vith_shapes():
    """ops.linear_kernel_name mirrors dispatch_gemm: the ViT-H short-K problems are timed under the wreg name."""
    from octic_vits_amd import ops
    assert ops.linear_kernel_name(160, bf, bf, 0).startswith("linear_d8_wreg_kernel")
    assert ops.linear_kernel_name(640, bf, bf, 0).startswith("linear_d8_ring_kernel")
