"""Hand-written dense bf16 MFMA GEMM (csrc/dense_gemm.hip) through the C ABI vs torch fp32/fp64 matmul on the same
bf16-rounded operands.  Tolerance: bf16 output (8-bit mantissa) of an f32-accumulated sum -> 1e-2 of the output scale
(plus exact-path checks on integer-valued operands, where every product and partial sum is representable: bit exact)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"


def ops():
    from octic_vits_amd import ops as o
    return o


def rnd(shape, seed, scale=1.0, dtype=torch.bfloat16):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return (torch.randn(*shape, generator=g, device=DEV) * scale).to(dtype)


def close(got, want, tol, msg):
    got, want = got.double(), want.double()
    scale = max(1.0, float(want.abs().max()))
    err = float((got - want).abs().max())
    assert err <= tol * scale, f"{msg}: max err {err:.3e} > {tol:g} x {scale:.3g}"


# ViT-H standard block at B = 64 (M = 16448): forward (N,K) and input-gradient shapes; plus ragged edges
FULL = [(16448, 3840, 1280), (16448, 1280, 1280), (16448, 5120, 1280), (16448, 1280, 5120), (16448, 1280, 3840)]
SMALL = [(256, 256, 128), (300, 264, 256), (1000, 520, 384), (77, 1280, 1280), (4112, 1024, 1024), (513, 8, 128), (2000, 264, 192)]


@pytest.mark.parametrize("M,N,K", SMALL + FULL)
def test_dense_nt_plain(M, N, K):
    o = ops()
    a, b = rnd((M, K), 1), rnd((N, K), 2, K ** -0.5)
    bias = rnd((N,), 3, dtype=torch.float32)
    c = o.dense_gemm_nt(a, b, 0, bias=bias)
    want = a.double() @ b.double().t() + bias.double()
    close(c, want, 1e-2, f"plain {M}x{N}x{K}")
    c0 = o.dense_gemm_nt(a, b, 0)
    close(c0, a.double() @ b.double().t(), 1e-2, f"plain no bias {M}x{N}x{K}")


@pytest.mark.parametrize("M,N,K", [(300, 264, 256), (16448, 1280, 1280), (16448, 5120, 1280)])
def test_dense_nt_integer_operands_bit_exact(M, N, K):
    """Small-integer operands: products and f32 partial sums are exact, so any dropped / doubled K-slice, swapped
    fragment or mis-addressed row shows up as a wrong integer (asymmetric operands, cdna guide 'A=I check')."""
    o = ops()
    g = torch.Generator(device=DEV).manual_seed(5)
    a = torch.randint(-3, 4, (M, K), generator=g, device=DEV).to(torch.bfloat16)
    b = torch.randint(-2, 3, (N, K), generator=g, device=DEV).to(torch.bfloat16)
    c = o.dense_gemm_nt(a, b, 0)
    want = (a.float() @ b.float().t()).to(torch.bfloat16)
    assert torch.equal(c, want), f"{int((c != want).sum())} wrong elements"


def _force_tile(nt):
    """routing override of csrc/dense_gemm.hip: tile width of the plain mode (0 = cost model, 4 = 256-wide, 5 = 320-wide)"""
    from octic_vits_amd import _lib
    _lib.route_override(_lib.ROUTE_DENSE_TILE, nt)


# round 4: the 256 x 320 tile (one round of workgroups for the N = 1280 problems).  Shapes with N % 320 == 0: ViT-H's four
# N = 1280 uses, ragged M (1 .. 255 rows in the last panel, a panel of exactly 64 rows, fewer rows than one wave row),
# short K (ring never steady), more tiles than CUs (several rounds), a K-split tail that is not the 64-row panel
W320 = [(16448, 1280, 1280), (16448, 1280, 5120), (300, 320, 256), (77, 640, 128), (1000, 960, 384), (4112, 1280, 1024),
        (513, 320, 192), (66000, 1280, 256), (16448, 3840, 1280)]


@pytest.mark.parametrize("M,N,K", W320)
def test_dense_nt_320_wide_tile_integer_exact_and_equal_to_the_256_wide_tile(M, N, K):
    """Both tile widths forced in turn on the same operands.  Integer-valued operands make every product and partial sum
    exact: any dropped / doubled K-slice, swapped column set or mis-addressed DMA row of the 3 + 2 n-tile layout is a wrong
    integer.  On random operands the two widths differ only where the split-K tail sums slabs in another grouping, so they
    are held to the bf16 output tolerance against fp64 and to 2 ulp of each other."""
    o = ops()
    g = torch.Generator(device=DEV).manual_seed(11)
    a = torch.randint(-3, 4, (M, K), generator=g, device=DEV).to(torch.bfloat16)
    b = torch.randint(-2, 3, (N, K), generator=g, device=DEV).to(torch.bfloat16)
    bias = torch.randint(-4, 5, (N,), generator=g, device=DEV).float()
    want = (a.float() @ b.float().t() + bias).to(torch.bfloat16)
    try:
        for nt in (5, 4):
            _force_tile(nt)
            c = o.dense_gemm_nt(a, b, 0, bias=bias)
            assert torch.equal(c, want), f"tile {nt}: {int((c != want).sum())} wrong elements of {c.numel()}"
        ar, br = rnd((M, K), 1), rnd((N, K), 2, K ** -0.5)
        ref = ar.double() @ br.double().t()
        outs = []
        for nt in (5, 4):
            _force_tile(nt)
            c = o.dense_gemm_nt(ar, br, 0)
            close(c, ref, 1e-2, f"tile {nt} {M}x{N}x{K}")
            outs.append(c.float())
        scale = max(1.0, float(ref.abs().max()))
        assert float((outs[0] - outs[1]).abs().max()) <= 2 ** -6 * scale
    finally:
        _force_tile(0)


@pytest.mark.parametrize("M,N,K", [(16448, 1280, 1280), (16448, 1280, 5120), (66000, 1280, 256)])
def test_dense_nt_320_wide_tile_is_bitwise_repeatable(M, N, K):
    """The 64-row last panel goes through the split-K front of the grid (f32 slabs, ticket, last arriver sums in slab
    order): identical launches give identical bits, also right after a launch of another shape used the workspace."""
    o = ops()
    a, b = rnd((M, K), 3), rnd((N, K), 4, K ** -0.5)
    try:
        _force_tile(5)
        first = o.dense_gemm_nt(a, b, 0)
        o.dense_gemm_nt(rnd((3000, 1280), 5), rnd((1280, 1280), 6), 0)
        for _ in range(3):
            assert torch.equal(o.dense_gemm_nt(a, b, 0), first)
    finally:
        _force_tile(0)


@pytest.mark.parametrize("M,N,K", [(300, 264, 256), (16448, 5120, 1280)])
def test_dense_nt_gelu(M, N, K):
    o = ops()
    a, b = rnd((M, K), 11), rnd((N, K), 12, K ** -0.5)
    bias = rnd((N,), 13, dtype=torch.float32)
    h, y = o.dense_gemm_nt(a, b, 1, bias=bias)
    want = a.double() @ b.double().t() + bias.double()
    close(h, want, 1e-2, "gelu: pre-activation")
    assert torch.equal(y, torch.nn.functional.gelu(h.float()).to(torch.bfloat16)) or \
        float((y.float() - torch.nn.functional.gelu(h.float())).abs().max()) <= 1e-2


@pytest.mark.parametrize("M,N,K,T", [(3 * 50, 264, 256, 50), (16448, 1280, 5120, 257), (16448, 1280, 1280, 257)])
def test_dense_nt_resid(M, N, K, T):
    o = ops()
    a, b = rnd((M, K), 21), rnd((N, K), 22, K ** -0.5)
    bias = rnd((N,), 23, dtype=torch.float32)
    gamma = rnd((N,), 24, dtype=torch.float32) * 0.2 + 0.5
    rs = (torch.rand(M // T, generator=torch.Generator().manual_seed(3)) > 0.5).float().to(DEV) * 2.0
    x = rnd((M, N), 25, dtype=torch.float32)
    y, out = o.dense_gemm_nt(a, b, 2, bias=bias, gamma=gamma, rs=rs, rps=T, x=x)
    want_y = a.double() @ b.double().t() + bias.double()
    close(y, want_y, 1e-2, "resid: branch")
    want = x.double() + rs.double().repeat_interleave(T)[:, None] * gamma.double() * y.double()
    close(out, want, 1e-5, "resid: stream (from the rounded branch)")
    # no gamma / no rs variants
    y2, out2 = o.dense_gemm_nt(a, b, 2, bias=None, gamma=None, rs=None, x=x)
    close(out2, x.double() + y2.double(), 1e-5, "resid: plain residual")


@pytest.mark.parametrize("M,N,K", [(300, 264, 256), (16448, 5120, 1280)])
def test_dense_nt_dgelu(M, N, K):
    o = ops()
    a, b = rnd((M, K), 31), rnd((N, K), 32, K ** -0.5)
    h = rnd((M, N), 33)
    d = o.dense_gemm_nt(a, b, 3, h=h)
    g = (a.double() @ b.double().t()).to(torch.bfloat16)
    hf = h.double().requires_grad_(True)
    torch.nn.functional.gelu(hf).backward(g.double())
    close(d, hf.grad, 1e-2, "dgelu")
    # fc1's bias gradient from the same epilogue: the f32 column sums of the bf16 result it stored (fixed order -> the same
    # bits on every launch), split-K tail tiles included (16448 x 5120 x 1280: 20 tiles cut five ways)
    d2, cs = o.dense_gemm_nt(a, b, 3, h=h, want_colsum=True)
    assert torch.equal(d2, d)
    want = d.double().sum(0)
    assert float((cs.double() - want).abs().max()) <= 1e-5 * max(1.0, float(d.double().abs().sum(0).max()))
    assert torch.equal(o.dense_gemm_nt(a, b, 3, h=h, want_colsum=True)[1], cs)


@pytest.mark.parametrize("M,N,K", [(300, 264, 256), (12608, 5120, 1280), (16448, 5120, 1280)])
def test_dense_nt_gelu_only_equals_the_gelu_output_of_mode_1(M, N, K):
    """Mode 6 (round 5): gelu(acc + bias) alone for passes that never run a backward (inference, the DINOv2 teacher) -
    the tensor mode 1 returns beside its pre-activation, bit for bit (same epilogue arithmetic), split-K tail tiles included."""
    o = ops()
    a, b = rnd((M, K), 21), rnd((N, K), 22, K ** -0.5)
    bias = rnd((N,), 23, dtype=torch.float32)
    _, y1 = o.dense_gemm_nt(a, b, 1, bias=bias)
    y6 = o.dense_gemm_nt(a, b, 6, bias=bias)
    assert y6.dtype == torch.bfloat16 and torch.equal(y6, y1)
    assert torch.equal(o.dense_gemm_nt(a, b, 6), o.dense_gemm_nt(a, b, 1)[1])        # no bias


def test_standard_mlp_without_grad_takes_the_gelu_only_epilogue():
    """vit.Mlp.forward_fused under torch.no_grad (DenseMlpFn: no input needs a gradient -> mode 6) equals the training-mode
    forward of the same inputs bit for bit when the training pass keeps h (GELU_FACTOR off: mode 1), and to one bf16 ulp of the
    activation against the stored-factor form (mode 4)."""
    from octic_vits_amd import functional as OF, vit
    torch.manual_seed(0)
    mlp = vit.Mlp(1280, 5120).cuda()
    g = torch.Generator(device=DEV).manual_seed(2)
    y = torch.randn(2, 197, 1280, generator=g, device=DEV).to(torch.bfloat16)
    x = torch.randn(2, 197, 1280, generator=g, device=DEV)
    gamma = torch.rand(1280, generator=g, device=DEV)
    with torch.no_grad():
        out0 = mlp.forward_fused(y, x, gamma, None, torch.bfloat16)
    old = OF.GELU_FACTOR
    try:
        OF.GELU_FACTOR = False
        out1 = mlp.forward_fused(y.clone().requires_grad_(True), x.clone().requires_grad_(True), gamma, None, torch.bfloat16)
        OF.GELU_FACTOR = True
        out4 = mlp.forward_fused(y.clone().requires_grad_(True), x.clone().requires_grad_(True), gamma, None, torch.bfloat16)
    finally:
        OF.GELU_FACTOR = old
    assert torch.equal(out0, out1.detach())
    assert float((out0 - out4.detach()).abs().max()) <= 1e-2 * float(out0.abs().max())


@pytest.mark.parametrize("M,N,K", [(300, 264, 256), (16448, 5120, 1280)])
def test_dense_nt_gelu_factor_pair(M, N, K):
    """Modes 4 / 5 (round 5): fc1's epilogue leaves gelu'(h) in bf16 beside gelu(h), fc2's input gradient multiplies by the
    stored factor.  (a) mode 4's gelu output = mode 1's to one bf16 ulp, its factor = gelu' of the bf16-rounded pre-activation;
    (b) mode 5 = bf16(factor * acc) EXACTLY on integer operands, with the same column sums protocol as mode 3; (c) the chain
    4 -> 5 against the chain 1 -> 3: within the one extra bf16 rounding of the factor."""
    o = ops()
    a, b = rnd((M, K), 11), rnd((N, K), 12, K ** -0.5)
    bias = rnd((N,), 13, dtype=torch.float32)
    h, y1 = o.dense_gemm_nt(a, b, 1, bias=bias)
    f, y4 = o.dense_gemm_nt(a, b, 4, bias=bias)
    assert float((y4.float() - y1.float()).abs().max()) <= 2.0 ** -7 * max(1.0, float(y1.float().abs().max()))
    assert float(((y4.float() - y1.float()).abs() > 0).float().mean()) < 0.02          # (a last-bit difference is rare)
    hf = h.double().requires_grad_(True)
    torch.nn.functional.gelu(hf).sum().backward()
    assert float((f.double() - hf.grad).abs().max()) <= 2.0 ** -8 * 1.2                  # |gelu'| <= 1.13: half a bf16 ulp + erf error
    # (b) exact arithmetic of mode 5
    g = torch.Generator(device=DEV).manual_seed(5)
    ai = torch.randint(-3, 4, (M, K), generator=g, device=DEV).to(torch.bfloat16)
    bi = torch.randint(-2, 3, (N, K), generator=g, device=DEV).to(torch.bfloat16)
    d5, cs = o.dense_gemm_nt(ai, bi, 5, h=f, want_colsum=True)
    # (the accumulator is rounded to bf16 when it is staged for the row-wise pass, as in modes 1 - 3: two roundings)
    want = (f.float() * (ai.float() @ bi.float().t()).to(torch.bfloat16).float()).to(torch.bfloat16)
    assert torch.equal(d5, want)
    assert float((cs.double() - d5.double().sum(0)).abs().max()) <= 1e-5 * max(1.0, float(d5.double().abs().sum(0).max()))
    assert torch.equal(o.dense_gemm_nt(ai, bi, 5, h=f, want_colsum=True)[1], cs)
    # (c) the two chains
    up = rnd((M, N), 41, 0.5)            # stands for gbr W2: any bf16 cotangent reaching the epilogue through the GEMM
    eye = torch.eye(N, device=DEV, dtype=torch.bfloat16) if N <= 512 else None
    if eye is not None:                   # small case: drive the epilogues with an identity B so that acc == up exactly
        d3 = o.dense_gemm_nt(up[:, :256].contiguous(), eye[:N, :256].contiguous(), 3, h=h)
        d5b = o.dense_gemm_nt(up[:, :256].contiguous(), eye[:N, :256].contiguous(), 5, h=f)
        sc = max(1.0, float(d3.float().abs().max()))
        assert float((d3.float() - d5b.float()).abs().max()) <= 2.0 ** -7 * sc


@pytest.mark.parametrize("M,N,K", [(16448, 1280, 1280), (16448, 1280, 5120), (16448, 5120, 1280), (3000, 520, 4096),
                                   (2100, 264, 2048)])
def test_dense_nt_split_tail_is_bitwise_repeatable(M, N, K):
    """The last partial round of tiles is cut along K when that pays (long K, or a ragged last row panel); the last
    arriver sums the partial tiles in slab order, rows past M never travel: ten launches give the same bits."""
    o = ops()
    a, b = rnd((M, K), 41), rnd((N, K), 42, K ** -0.5)
    bias = rnd((N,), 43, dtype=torch.float32)
    first = o.dense_gemm_nt(a, b, 0, bias=bias)
    for _ in range(9):
        assert torch.equal(o.dense_gemm_nt(a, b, 0, bias=bias), first)
    close(first, a.double() @ b.double().t() + bias.double(), 1e-2, "split tail")


@pytest.mark.parametrize("rows,d,ld", [(16448, 3840, 3840), (1000, 264, 264), (257, 1280, 3840), (5, 8, 8)])
def test_dense_colsum_matches_f32_sum(rows, d, ld):
    """qkv bias gradient: f32 column sums of the bf16 cotangent (autograd's grad.sum(0), deit/vit.py:33)."""
    o = ops()
    g = rnd((rows, ld), 51)[:, :d]
    got = o.dense_colsum(g)
    want = g.double().sum(0)
    assert float((got.double() - want).abs().max()) <= 1e-5 * max(1.0, float(g.double().abs().sum(0).max()))
    assert torch.equal(o.dense_colsum(g), got)


def test_dense_prep_batch_matches_torch():
    import ctypes
    import numpy as np
    from octic_vits_amd import _lib
    L = _lib.lib()
    shapes = [(96, 128), (200, 72), (1280, 1280)]
    dt = np.dtype([("src", "<u8"), ("wb", "<u8"), ("wt", "<u8"), ("N", "<i4"), ("K", "<i4"), ("block_begin", "<i4"), ("pad", "<i4")])
    for src_dtype, code in ((torch.float32, _lib.F32), (torch.bfloat16, _lib.BF16)):
        ws = [rnd(s, 40 + i, dtype=src_dtype) for i, s in enumerate(shapes)]
        wbs = [torch.empty(s, dtype=torch.bfloat16, device=DEV) for s in shapes]
        wts = [torch.empty(s[::-1], dtype=torch.bfloat16, device=DEV) for s in shapes]
        tab = np.zeros(len(shapes), dtype=dt)
        blocks = 0
        for i, (N, K) in enumerate(shapes):
            tab[i]["src"], tab[i]["wb"], tab[i]["wt"] = ws[i].data_ptr(), wbs[i].data_ptr(), wts[i].data_ptr()
            tab[i]["N"], tab[i]["K"], tab[i]["block_begin"] = N, K, blocks
            blocks += L.octic_dense_prep_batch_blocks(N, K)
        items = torch.from_numpy(tab.view(np.uint8).copy()).to(DEV)
        _lib.check(L.octic_dense_prep_batch(ctypes.c_void_p(items.data_ptr()), len(shapes), blocks, code, None))
        torch.cuda.synchronize()
        for w, wb, wt in zip(ws, wbs, wts):
            assert torch.equal(wb, w.to(torch.bfloat16)) and torch.equal(wt, w.to(torch.bfloat16).t())


def test_standard_block_on_dense_hip_matches_library_path():
    """The standard block with its four projections on csrc/dense_gemm.hip (fused bias / GELU / layer-scale + drop-path +
    residual / GELU') against the same block on the BLAS library + separate row kernels: outputs and all gradients
    within bf16 tolerance (2e-2 of scale), same RNG draws."""
    from octic_vits_amd import functional as OF
    from octic_vits_amd.vit import Layer_scale_init_Block
    torch.manual_seed(0)
    blk = Layer_scale_init_Block(dim=256, num_heads=4, qkv_bias=True, init_values=0.5, drop_path=0.3).cuda().train()
    x = torch.randn(6, 65, 256, device=DEV)
    cot = torch.randn(6, 65, 256, device=DEV)
    res = {}
    saved = set(OF.DENSE_HIP)
    ALL = {"qkv", "dqkv", "proj", "dproj", "fc1", "dfc1", "fc2", "dfc2"}
    for mode in (True, False, "default"):
        OF.DENSE_HIP = ALL if mode is True else (set() if mode is False else saved)
        try:
            blk.zero_grad(set_to_none=True)
            xi = x.clone().requires_grad_(True)
            torch.manual_seed(123)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = blk(xi)
            y.backward(cot)
            res[mode] = [y.detach().clone(), xi.grad.clone()] + [p.grad.clone() for p in blk.parameters()]
        finally:
            OF.DENSE_HIP = saved
    names = ["out", "dx"] + [n for n, _ in blk.named_parameters()]
    for n, a, b, c in zip(names, res[True], res[False], res["default"]):
        close(a, b, 2e-2, f"block {n} (all GEMMs hand-written vs all on the library)")
        close(c, b, 2e-2, f"block {n} (default routing vs all on the library)")


# ------------------------------------------------------------------------------------------------ weight gradient (TN)
WG_SHAPES = [(200, 256, 256), (64, 512, 256), (1000, 256, 768), (16448, 1280, 1280), (16448, 3840, 1280),
             (16448, 5120, 1280), (16448, 1280, 5120)]


@pytest.mark.parametrize("M,N,K", WG_SHAPES)
def test_dense_wgrad_tn(M, N, K):
    """dW = dY^T X in f32 vs fp64 matmul of the same bf16 operands (f32 accumulation over M rows: 2e-4 of scale), and
    bitwise reproducibility (fixed-order reduction of the stream-K partial tiles)."""
    o = ops()
    dy, x = rnd((M, N), 51), rnd((M, K), 52)
    dw = o.dense_wgrad_tn(dy, x)
    want = dy.double().t() @ x.double()
    close(dw, want, 2e-4, f"wgrad {M}x{N}x{K}")
    dw2 = o.dense_wgrad_tn(dy, x)
    assert torch.equal(dw, dw2), "wgrad not bitwise reproducible"


@pytest.mark.parametrize("M,N,K", [(200, 256, 256), (16448, 1280, 1280), (16448, 5120, 1280)])
def test_dense_wgrad_tn_integer_operands_exact(M, N, K):
    o = ops()
    g = torch.Generator(device=DEV).manual_seed(7)
    dy = torch.randint(-2, 3, (M, N), generator=g, device=DEV).to(torch.bfloat16)
    x = torch.randint(-3, 4, (M, K), generator=g, device=DEV).to(torch.bfloat16)
    dw = o.dense_wgrad_tn(dy, x)
    want = dy.double().t() @ x.double()          # |sum| <= 6 * 16448 < 2^24: exact in f32
    assert torch.equal(dw.double(), want), f"{int((dw.double() != want).sum())} wrong elements"


# round 4: the 256 x 320 tile of the TN kernel (K % 320 == 0: 80 tiles x 3 slabs instead of 100 x 2 for the MLP weights)
def _force_tn_tile(width):
    """routing override of csrc/dense_wgrad.hip: tile width along K (0 = launch estimate, 256, 320)"""
    from octic_vits_amd import _lib
    _lib.route_override(_lib.ROUTE_WGRAD_TILE, width)


TN320_SHAPES = [(200, 256, 320), (64, 512, 640), (1, 256, 320), (130, 256, 960), (1000, 768, 1280), (16448, 1280, 1280),
                (16448, 3840, 1280), (16448, 5120, 1280), (16448, 1280, 5120), (12608, 1024, 1280)]


@pytest.mark.parametrize("M,N,K", TN320_SHAPES)
def test_dense_wgrad_tn_both_tile_widths_integer_exact(M, N, K):
    """Small-integer operands: every partial sum is an integer below 2^24, so both tile widths, every slab count and the
    fp64 product agree bit for bit (ragged M: rows past M read as zero through the buffer descriptor)."""
    o = ops()
    g = torch.Generator(device=DEV).manual_seed(11)
    dy = torch.randint(-2, 3, (M, N), generator=g, device=DEV).to(torch.bfloat16)
    x = torch.randint(-3, 4, (M, K), generator=g, device=DEV).to(torch.bfloat16)
    want = dy.double().t() @ x.double()
    try:
        for width in (320, 256, 0):
            _force_tn_tile(width)
            if width and K % width == 0:
                assert lib_tile(o, M, N, K) == width
            dw = o.dense_wgrad_tn(dy, x)
            assert torch.equal(dw.double(), want), f"width {width}: {int((dw.double() != want).sum())} wrong elements"
    finally:
        _force_tn_tile(0)


def lib_tile(o, M, N, K):
    from octic_vits_amd import _lib
    return int(_lib.lib().octic_dense_wgrad_tile(M, N, K))


@pytest.mark.parametrize("M,N,K", [(16448, 5120, 1280), (16448, 1280, 5120), (16448, 1280, 1280), (700, 512, 640)])
def test_dense_wgrad_tn_320_wide_tile_random_operands(M, N, K):
    """Random operands at the 320-wide tile vs fp64 (2e-4 of scale, the bound of test_dense_wgrad_tn), bitwise repeatable -
    also right after a launch of another shape and width went through the shared workspace - and within f32 summation
    noise of the 256-wide tile."""
    o = ops()
    dy, x = rnd((M, N), 61), rnd((M, K), 62)
    want = dy.double().t() @ x.double()
    try:
        _force_tn_tile(320)
        first = o.dense_wgrad_tn(dy, x)
        close(first, want, 2e-4, f"wgrad 320-wide {M}x{N}x{K}")
        _force_tn_tile(256)
        other = o.dense_wgrad_tn(dy, x)
        o.dense_wgrad_tn(rnd((900, 512), 63), rnd((900, 1280), 64))
        _force_tn_tile(320)
        o.dense_wgrad_tn(rnd((3000, 256), 65), rnd((3000, 640), 66))
        for _ in range(2):
            assert torch.equal(o.dense_wgrad_tn(dy, x), first), "320-wide wgrad not bitwise repeatable"
        scale = max(1.0, float(want.abs().max()))
        assert float((first - other).abs().max()) <= 1e-4 * scale
    finally:
        _force_tn_tile(0)


def test_dense_wgrad_tn_default_width():
    """256-wide tiles wherever K allows (the 320-wide tile measured no faster on the ViT-H shapes: csrc/dense_wgrad.hip
    dw_plan), 320-wide for K % 320 == 0 only."""
    o = ops()
    for N, K in ((3840, 1280), (1280, 1280), (5120, 1280), (1280, 5120), (1024, 1024)):
        assert lib_tile(o, 16448, N, K) == 256, (N, K)
    assert lib_tile(o, 16448, 1280, 960) == 320
    assert lib_tile(o, 16448, 1280, 200) == 0


# round 5: two weight gradients (same token rows, same K) as one launch - the qkv and proj weight gradients of a block
@pytest.mark.parametrize("M,N0,N1,K", [(16448, 3840, 1280, 1280), (390, 768, 256, 256), (12608, 3072, 1024, 1024), (1, 256, 256, 320),
                                       (700, 512, 1280, 640), (4112, 1280, 3840, 1280)])
def test_dense_wgrad_pair_integer_exact_and_equal_to_two_launches(M, N0, N1, K):
    """octic_dense_wgrad_tn_pair: the joint tile list [problem 0 | problem 1] with row slabs chosen for the sum.  Small-integer
    operands make every partial sum exact, so the pair, the two single launches and the fp64 products agree bit for bit
    whatever slab counts each launch picks; the second result lands in a tensor handed in by the caller."""
    o = ops()
    g = torch.Generator(device=DEV).manual_seed(M + N0)
    mk = lambda r, c, lo, hi: torch.randint(lo, hi, (r, c), generator=g, device=DEV).to(torch.bfloat16)
    dy0, x0, dy1, x1 = mk(M, N0, -2, 3), mk(M, K, -3, 4), mk(M, N1, -2, 3), mk(M, K, -3, 4)
    out1 = torch.full((N1, K), float("nan"), device=DEV)
    dw0, dw1 = o.dense_wgrad_tn_pair(dy0, x0, dy1, x1, dw1=out1)
    assert dw1.data_ptr() == out1.data_ptr()
    assert torch.equal(dw0.double(), dy0.double().t() @ x0.double())
    assert torch.equal(dw1.double(), dy1.double().t() @ x1.double())
    assert torch.equal(dw0, o.dense_wgrad_tn(dy0, x0)) and torch.equal(dw1, o.dense_wgrad_tn(dy1, x1))
    # operands that are column slices of wider tensors (row strides), as the fused-QKV gradient hands them over
    wide = mk(M, N0 + 64, -2, 3)
    dw0s, _ = o.dense_wgrad_tn_pair(wide[:, :N0], x0, dy1, x1)
    assert torch.equal(dw0s.double(), wide[:, :N0].double().t() @ x0.double())


def test_paired_qkv_proj_weight_gradients_in_a_block_backward():
    """functional.WGRAD_PAIRED: inside a backward pass that allows postponed parameter gradients (ops.DEFERRED_FINISHES, as
    train.Trainer switches it on) proj's weight gradient is parked and written by the qkv weight gradient's launch of the same
    block.  Against the unpaired backward: every other gradient bitwise, the two weight gradients to f32 summation order (and
    bitwise here: both launches pick the same slab count at this size); nothing is left parked."""
    from octic_vits_amd import functional as OF
    from octic_vits_amd import ops as o
    from octic_vits_amd.vit import Layer_scale_init_Block
    torch.manual_seed(0)
    blk = torch.nn.Sequential(*[Layer_scale_init_Block(dim=256, num_heads=4, qkv_bias=True, init_values=0.5, drop_path=0.0)
                                for _ in range(2)]).cuda().train()
    x = torch.randn(6, 65, 256, device=DEV)
    cot = torch.randn(6, 65, 256, device=DEV)
    res, launches = {}, {}
    for paired in (False, True):
        OF.WGRAD_PAIRED = paired
        blk.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = blk(xi)
        o.DEFERRED_FINISHES.enabled = True
        o.KERNEL_TIMER.enable()
        try:
            y.backward(cot)
        finally:
            o.DEFERRED_FINISHES.enabled = False
            OF.WGRAD_PAIRED = True
        torch.cuda.synchronize()
        launches[paired] = sorted(k for k in o.KERNEL_TIMER.summary() if k.startswith("dense_tn_kernel"))
        o.KERNEL_TIMER.disable()
        assert not o.DEFERRED_FINISHES.pairs and all(b.attn._wgpair.pending is None for b in blk)
        res[paired] = {n: p.grad.clone() for n, p in blk.named_parameters()}
        res[paired]["x"] = xi.grad.clone()
    assert any("+" in k for k in launches[True]) and not any("+" in k for k in launches[False]), launches
    for n in res[False]:
        a_, b_ = res[False][n], res[True][n]
        assert torch.isfinite(b_).all(), n
        if n.endswith(("qkv.weight", "proj.weight")):
            assert float((a_ - b_).norm()) <= 1e-5 * float(a_.norm()), n
        else:
            assert torch.equal(a_, b_), n


# ---- round 6: per-image row panels + the class-token kernel (octic_dense_gemm_nt_tokens, tokens = 257) ----------------
def _force_image(v):
    """routing override: 0 = launch model, 1 = per-image panels wherever legal, 2 = never"""
    from octic_vits_amd import _lib
    _lib.route_override(_lib.ROUTE_DENSE_IMAGE, v)


IMG = [(64, 1280, 1280), (64, 1280, 5120), (64, 5120, 1280), (64, 3840, 1280), (3, 512, 256), (17, 320, 768), (33, 1280, 256),
       (1, 256, 256), (130, 1280, 512)]


@pytest.mark.parametrize("B,N,K", IMG)
def test_dense_nt_per_image_panels_integer_exact_and_equal_to_classic_panels(B, N, K):
    """M = B x 257 token rows, the per-image route forced on and off.  Integer-valued operands: every product and partial sum
    is exact, so a mis-addressed panel row (the +1 / x 257 row map), a dropped class-token row or a wrong K slice of the
    eight-wave class-token contraction is a wrong integer - bit-exact against torch and against the classic panels.  Random
    operands: both routes within the bf16 output tolerance of fp64; patch rows bitwise equal wherever the classic panel was
    not a split-K tail (full K in one workgroup either way); two launches of the per-image route bitwise equal."""
    o = ops()
    M = B * 257
    g = torch.Generator(device=DEV).manual_seed(21)
    a = torch.randint(-3, 4, (M, K), generator=g, device=DEV).to(torch.bfloat16)
    b = torch.randint(-2, 3, (N, K), generator=g, device=DEV).to(torch.bfloat16)
    bias = torch.randint(-4, 5, (N,), generator=g, device=DEV).float()
    want = (a.float() @ b.float().t() + bias).to(torch.bfloat16)
    try:
        _force_image(1)
        assert o.dense_plan(M, N, K, 0, 257)[2] is True
        c1 = o.dense_gemm_nt(a, b, 0, bias=bias, tokens=257)
        _force_image(2)
        assert o.dense_plan(M, N, K, 0, 257)[2] is False
        c2 = o.dense_gemm_nt(a, b, 0, bias=bias, tokens=257)
        assert torch.equal(c1, want), f"per-image: {int((c1 != want).sum())} wrong elements"
        assert torch.equal(c2, want)
        a, b = rnd((M, K), 1), rnd((N, K), 2, K ** -0.5)
        bias = rnd((N,), 3, dtype=torch.float32)
        ref = a.double() @ b.double().t() + bias.double()
        _force_image(1)
        c1 = o.dense_gemm_nt(a, b, 0, bias=bias, tokens=257)
        c1b = o.dense_gemm_nt(a, b, 0, bias=bias, tokens=257)
        _force_image(2)
        c2 = o.dense_gemm_nt(a, b, 0, bias=bias, tokens=257)
        close(c1, ref, 1e-2, "per-image")
        close(c2, ref, 1e-2, "classic")
        assert torch.equal(c1, c1b)
        # without tokens the call is the classic launch, whatever the override says
        _force_image(1)
        assert torch.equal(o.dense_gemm_nt(a, b, 0, bias=bias), c2)
    finally:
        _force_image(0)


@pytest.mark.parametrize("B,N,K", [(64, 5120, 1280), (5, 512, 256), (20, 1024, 512)])
def test_dense_nt_per_image_fused_tails(B, N, K):
    """Modes 1 / 4 / 6 (GELU tails) and 3 / 5 (GELU' / stored factor times the product, + column sums) on per-image panels:
    against the classic panels (elementwise tails of identical pre-activations on the patch rows; class-token rows within
    one bf16 ulp of the differently ordered f32 sum) and the column sums against a float64 sum of the bf16 result."""
    o = ops()
    M = B * 257
    a, b = rnd((M, K), 31), rnd((N, K), 32, K ** -0.5)
    bias = rnd((N,), 33, dtype=torch.float32)
    hpre = rnd((M, N), 34)
    try:
        res = {}
        for force in (1, 2):
            _force_image(force)
            r = {}
            r["h1"], r["g1"] = o.dense_gemm_nt(a, b, 1, bias=bias, tokens=257)
            r["f4"], r["g4"] = o.dense_gemm_nt(a, b, 4, bias=bias, tokens=257)
            r["g6"] = o.dense_gemm_nt(a, b, 6, bias=bias, tokens=257)
            r["d3"], r["s3"] = o.dense_gemm_nt(a, b, 3, h=hpre, want_colsum=True, tokens=257)
            r["d5"], r["s5"] = o.dense_gemm_nt(a, b, 5, h=r["f4"], want_colsum=True, tokens=257)
            r["d3n"] = o.dense_gemm_nt(a, b, 3, h=hpre, tokens=257)
            res[force] = r
        img, cls = res[1], res[2]
        pre = (a.double() @ b.double().t() + bias.double())
        close(img["h1"], pre, 1e-2, "mode 1 pre-activation")
        assert torch.equal(img["g1"], img["g6"]) and torch.equal(img["g1"], img["g4"])      # one gelu, three modes
        rows = torch.arange(M, device=DEV)
        patch = rows % 257 != 0
        for k in ("h1", "g1", "f4", "g4", "g6", "d3", "d5", "d3n"):
            x, y = img[k].float(), cls[k].float()
            # patch rows of full panels run the same K order in both routes unless the classic panel was a split-K tail
            scale = max(1.0, float(y.abs().max()))
            assert float((x - y).abs().max()) <= 2 ** -7 * scale, k
            assert float((x[~patch] - y[~patch]).abs().max()) <= 2 ** -7 * scale, k + " (class-token rows)"
        assert torch.equal(img["d3"], img["d3n"])
        for k, d in (("s3", "d3"), ("s5", "d5")):
            want = img[d].double().sum(0)
            close(img[k], want, 2e-5, f"column sums {k}")
            close(cls[k], cls[d].double().sum(0), 2e-5, f"classic column sums {k}")
        # gelu / gelu' of the bf16-rounded pre-activation, as the row kernels see it
        hb = img["h1"].float()
        close(img["g1"], torch.nn.functional.gelu(hb.double()), 1e-2, "gelu")
        want_f = 0.5 * (1 + torch.erf(hb.double() / 2 ** 0.5)) + hb.double() * torch.exp(-0.5 * hb.double() ** 2) / (2 * torch.pi) ** 0.5
        close(img["f4"], want_f, 1e-2, "gelu'")
    finally:
        _force_image(0)


def test_dense_nt_launch_model_takes_per_image_panels_for_vit_huge():
    """At the bench's shape (64 images x 257 tokens) the launch model picks per-image panels where the saved split-K front of
    the grid is worth more than the class-token launch costs (N = 1280 with K = 1280 / 3840: proj and the input gradients of
    proj / qkv - exactly one round of 256 workgroups) and says what it does through octic_dense_gemm_plan."""
    o = ops()
    M = 64 * 257
    for N, K in ((1280, 1280), (1280, 3840), (1280, 5120)):
        tile, cs_rows, image, grid = o.dense_plan(M, N, K, 0, 257)
        assert image and tile == 320 and grid == 256, (N, K, tile, image, grid)
        assert cs_rows == 2 * 64 + 4
    assert o.dense_plan(M, 1280, 1280, 0, 0)[2] is False and o.dense_plan(M, 1280, 1280, 2, 257)[2] is False
    assert o.dense_plan(M, 5120, 1280, 5, 257)[2] is False            # measured: no shorter on per-image panels
    assert o.dense_plan(8 * 257, 1280, 1280, 0, 257)[2] is False      # one partial round either way: no extra launch


@pytest.mark.parametrize("B,N,K", [(64, 1280, 5120), (64, 1280, 3840), (130, 1280, 640), (5, 640, 960), (64, 320, 320), (17, 1280, 1280)])
def test_dense_nt_class_token_rows_as_two_launches(B, N, K):
    """The long-K class-token path (K cut over workgroups into f32 partial tiles + a summing launch; 64- or 80-column
    workgroups) forced wherever legal, on per-image panels: integer-valued operands bit-exact against torch and against the
    single-launch class-token kernel; random operands within the bf16 output tolerance, bitwise reproducible."""
    from octic_vits_amd import _lib
    o = ops()
    M = B * 257
    g = torch.Generator(device=DEV).manual_seed(23)
    a = torch.randint(-3, 4, (M, K), generator=g, device=DEV).to(torch.bfloat16)
    b = torch.randint(-2, 3, (N, K), generator=g, device=DEV).to(torch.bfloat16)
    bias = torch.randint(-4, 5, (N,), generator=g, device=DEV).float()
    want = (a.float() @ b.float().t() + bias).to(torch.bfloat16)
    try:
        _force_image(1)
        _lib.route_override(_lib.ROUTE_DENSE_CLS2, 2)
        c2 = o.dense_gemm_nt(a, b, 0, bias=bias, tokens=257)
        c2n = o.dense_gemm_nt(a, b, 0, tokens=257)
        _lib.route_override(_lib.ROUTE_DENSE_CLS2, 1)
        c1 = o.dense_gemm_nt(a, b, 0, bias=bias, tokens=257)
        assert torch.equal(c2, want), f"two launches: {int((c2 != want).sum())} wrong elements"
        assert torch.equal(c1, want)
        assert torch.equal(c2n, (a.float() @ b.float().t()).to(torch.bfloat16))
        a, b = rnd((M, K), 1), rnd((N, K), 2, K ** -0.5)
        bias = rnd((N,), 3, dtype=torch.float32)
        _lib.route_override(_lib.ROUTE_DENSE_CLS2, 2)
        x = o.dense_gemm_nt(a, b, 0, bias=bias, tokens=257)
        y = o.dense_gemm_nt(a, b, 0, bias=bias, tokens=257)
        close(x, a.double() @ b.double().t() + bias.double(), 1e-2, "two launches")
        assert torch.equal(x, y)
    finally:
        _lib.route_override(_lib.ROUTE_DENSE_CLS2, 0)
        _force_image(0)
