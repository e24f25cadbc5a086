"""Dispatcher-visible ops (torch.library) for the engine's entry points - what the north star calls "surfaced as PyTorch-ROCm
custom ops", beyond the one boundary the reference itself has (custom_ops.py: the D8-GELU of octic_vits/d8_gelu.py:456-478).

Why: the reference recipe trains under ``torch.compile`` (experiments/train_deit.py:51, deit/main.py:341-342).  The eager
product path is a set of ``torch.autograd.Function``s over ctypes with Python-side caches (prepared bf16 weights adopted from
the optimizer step, fused residual + next-norm links carried as tensor attributes): the fastest way to issue the step from
Python, but opaque to Dynamo - every ctypes call is a graph break.  The ops below are the SAME kernels behind schemas, fake
(meta) kernels and ``register_autograd`` formulas, free of Python state: a module that finds itself being traced
(``torch.compiler.is_compiling()``) calls them instead, and a block of either half traces into ONE graph.

    octic half   torch.ops.octic.layernorm_d8 / _bwd      LayerNormD8 + AffineD8        octic_vits/d8_layers.py:132-186
                 torch.ops.octic.linear_d8 / _bwd         LinearD8 (+ fused residual tail)   d8_layers.py:104-130, 698-707
                 torch.ops.octic.attn_packed / _bwd       AttentionD8's head split + SDPA + re-assembly   d8_layers.py:631-656
                 torch.ops.octic.gelu_d8 (custom_ops.py)  TritonGeluD8                  d8_gelu.py:456-482
                 torch.ops.octic.lift / _bwd              PatchEmbedD8 / LiftD8 (+ pos-embed, cls row)   d8_layers.py:284-497
                 torch.ops.octic.handoff_cat / _bwd       irreps -> regular features    model.py:196-200
                 torch.ops.octic.power_spectrum / _bwd    PowerSpectrumInvariant        d8_invariantization.py:49-64
    standard half torch.ops.octic.dense_layernorm / _bwd  nn.LayerNorm (f32 stream -> bf16)    deit/vit.py:131-134
                 torch.ops.octic.dense_linear / _bwd      nn.Linear (+ exact GELU) on csrc/dense_gemm.hip, dense_wgrad.hip   deit/vit.py:14-56
                 torch.ops.octic.dense_mlp / _bwd         timm Mlp: fc1 + GELU + fc2, gelu'(h) kept as a factor (round 6)   deit/vit.py:131-134
                 torch.ops.octic.attn_qkv / _bwd          SDPA on the fused [B,T,3,H,hd] projection   deit/vit.py:38-45
                 torch.ops.octic.scale_residual / _bwd    x + drop_path(gamma * y)      deit/vit.py:131-134

Weight preparation (bf16 casts, the transposed operand of the input gradient): a traced graph has no place for a cache keyed on
parameter versions, so by default it happens inside the ops, per call.  Round 6: where ``train.FusedLamb`` owns the compute-dtype
copies - it rewrites the SAME buffers in place after every step (DenseWeightCache.static_nt, WeightPrep.flat) - the modules hand
those buffers to the ops as plain tensor inputs (``wb`` / ``wt``, ``pwb`` / ``pwt``) and no preparation launch is left in the
traced step.  The fusions that need Python state are what the traced path still gives up against the eager one (numbers in
DESIGN.md); the arithmetic is identical kernel for kernel.
There is no CPU kernel: a CPU tensor raises, like every product op."""
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import custom_ops  # noqa: F401  (registers octic::gelu_d8*)
from . import ops

_lib = torch.library


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _f32(t):
    return None if t is None else _c(t.detach().float())


def _empty(ref):
    return ref.new_empty(0)


def _opt(t):
    """zero-size placeholder (custom ops cannot return None) -> None"""
    return None if (t is None or t.numel() == 0) else t


def is_tracing():
    return torch.compiler.is_compiling()


# ------------------------------------------------------------------------------------------------ LayerNormD8
@_lib.custom_op("octic::layernorm_d8", mutates_args=())
def layernorm_d8(x: Tensor, a1: Optional[Tensor], a2: Optional[Tensor], b1: Optional[Tensor], b2: Optional[Tensor],
                 ae: Optional[Tensor], beta: Optional[Tensor], eps: float, c: int, out_bf16: bool) -> Tuple[Tensor, Tensor]:
    x = _c(x.float())
    alpha = None if a1 is None else [_f32(t) for t in (a1, a2, b1, b2, ae)]
    y, stats = ops.layernorm_fwd(x, alpha, _f32(beta), eps, torch.bfloat16 if out_bf16 else torch.float32, c)
    return y, stats


@layernorm_d8.register_fake
def _(x, a1, a2, b1, b2, ae, beta, eps, c, out_bf16):
    if x.shape[-1] != 8 * c:
        raise ValueError(f"layernorm_d8: last dimension {x.shape[-1]} is not 8 c = {8 * c}")
    y = x.new_empty(x.shape, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    return y, x.new_empty((x.numel() // (8 * c), 8), dtype=torch.float32)


@_lib.custom_op("octic::layernorm_d8_bwd", mutates_args=())
def layernorm_d8_bwd(g: Tensor, x: Tensor, stats: Tensor, a1: Optional[Tensor], a2: Optional[Tensor], b1: Optional[Tensor],
                     b2: Optional[Tensor], ae: Optional[Tensor], c: int
                     ) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor]:
    x = _c(x.float())
    alpha = None if a1 is None else [_f32(t) for t in (a1, a2, b1, b2, ae)]
    dx, dal, dbeta = ops.layernorm_bwd(_c(g), x, stats, alpha, None, c, want_param_grads=alpha is not None)
    if dal is None:
        e = _empty(dx)
        return dx, e, e.clone(), e.clone(), e.clone(), e.clone(), e.clone()
    return (dx, *dal, dbeta)


@layernorm_d8_bwd.register_fake
def _(g, x, stats, a1, a2, b1, b2, ae, c):
    dx = x.new_empty(x.shape, dtype=torch.float32)
    if a1 is None:
        return (dx,) + tuple(dx.new_empty(0) for _ in range(6))
    return (dx,) + tuple(t.new_empty(t.shape, dtype=torch.float32) for t in (a1, a2, b1, b2, ae)) + (dx.new_empty(c),)


def _ln_setup(ctx, inputs, output):
    x, a1, a2, b1, b2, ae, beta, eps, c, out_bf16 = inputs
    ctx.save_for_backward(x, output[1], *([a1, a2, b1, b2, ae] if a1 is not None else []))
    ctx.c, ctx.affine, ctx.has_beta, ctx.xdtype = c, a1 is not None, beta is not None, x.dtype


def _ln_backward(ctx, g, _gstats):
    x, stats, *alpha = ctx.saved_tensors
    a = alpha if ctx.affine else [None] * 5
    dx, d1, d2, d3, d4, d5, dbeta = torch.ops.octic.layernorm_d8_bwd(g, x, stats, *a, ctx.c)
    dal = [d1, d2, d3, d4, d5] if ctx.affine else [None] * 5
    return (dx.to(ctx.xdtype), *dal, dbeta if ctx.has_beta else None, None, None, None)


layernorm_d8.register_autograd(_ln_backward, setup_context=_ln_setup)


# ------------------------------------------------------------------------------------------------ LinearD8
def _lin_dtype(x):
    return x.dtype if x.dtype in (torch.float32, torch.bfloat16) else torch.float32


@_lib.custom_op("octic::linear_d8", mutates_args=())
def linear_d8(x: Tensor, wA1: Tensor, wA2: Tensor, wB1: Tensor, wB2: Tensor, wE: Tensor, bias: Optional[Tensor],
              resid: Optional[Tensor], rs: Optional[Tensor], sA1: Optional[Tensor], sA2: Optional[Tensor],
              sB1: Optional[Tensor], sB2: Optional[Tensor], sE: Optional[Tensor], cin: int, cout: int, rps: int,
              pwb: Optional[Tensor] = None, pwt: Optional[Tensor] = None) -> Tensor:
    """y = resid + rs[row / rps] * cs * (x W^T + bias)   (resid, rs, cs optional); operands in x's dtype (f32 | bf16).
    pwb / pwt: the flat prepared bf16 copies of the weights (forward operand / transposed, layer-scale folded in) that
    functional.PrepBatch refreshes in place after every fused optimizer step - given, no preparation launches run here."""
    x = _c(x)
    dtype = _lin_dtype(x)
    cs32 = None if sA1 is None else [_f32(s) for s in (sA1, sA2, sB1, sB2, sE)]
    if pwb is not None and dtype == torch.bfloat16 and pwb.dtype == torch.bfloat16:
        wb = ops.prep_views(pwb, cin, cout, False)
    else:
        w32 = [_f32(w) for w in (wA1, wA2, wB1, wB2, wE)]
        wb, _wt = ops.linear_prep(w32, None, cin, cout, dtype, want_wb=(dtype != torch.float32))
        if wb is None:
            wb = w32
    fused = resid is not None
    out_dtype = resid.dtype if fused else dtype
    M = x.numel() // (8 * cin)
    y = torch.empty(x.shape[:-1] + (8 * cout,), dtype=out_dtype, device=x.device)
    ops.linear_fwd(ops.pview(x, cin), wb, _f32(bias), ops.pview(y, cout), M, cin, cout, dtype, out_dtype, x,
                   resid_v=ops.pview(_c(resid), cout) if fused else None, rs=_f32(rs), rps=rps, cs5=cs32)
    return y


@linear_d8.register_fake
def _(x, wA1, wA2, wB1, wB2, wE, bias, resid, rs, sA1, sA2, sB1, sB2, sE, cin, cout, rps, pwb=None, pwt=None):
    if x.shape[-1] != 8 * cin:
        raise ValueError(f"linear_d8: expected {8 * cin} channels, got {x.shape[-1]}")
    return x.new_empty(x.shape[:-1] + (8 * cout,), dtype=resid.dtype if resid is not None else x.dtype)


@_lib.custom_op("octic::linear_d8_bwd", mutates_args=())
def linear_d8_bwd(dy: Tensor, x: Tensor, wA1: Tensor, wA2: Tensor, wB1: Tensor, wB2: Tensor, wE: Tensor,
                  bias: Optional[Tensor], rs: Optional[Tensor], sA1: Optional[Tensor], sA2: Optional[Tensor],
                  sB1: Optional[Tensor], sB2: Optional[Tensor], sE: Optional[Tensor], cin: int, cout: int, rps: int,
                  fused: bool, need_dx: bool, pwt: Optional[Tensor] = None
                  ) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor]:
    """-> (dx, dW x 5, dbias, dcs x 5): the chain rule of linear_d8 for dy = dL/dy (f32 when the tail is fused)."""
    x = _c(x)
    dtype = _lin_dtype(x)
    w5 = (wA1, wA2, wB1, wB2, wE)
    w32 = [_f32(w) for w in w5]
    has_cs, has_bias = sA1 is not None, bias is not None
    cs32 = [_f32(s) for s in (sA1, sA2, sB1, sB2, sE)] if has_cs else None
    M = x.numel() // (8 * cin)
    dy = _c(dy)
    if fused or rs is not None or dy.dtype != dtype:
        g = ops.cast_rowscale(_c(dy.float()), _f32(rs), rps, dtype, cout)
    else:
        g = dy
    gv, xv = ops.pview(g, cout), ops.pview(x, cin)
    if need_dx:
        if pwt is not None and dtype == torch.bfloat16 and pwt.dtype == torch.bfloat16:
            wt = ops.prep_views(pwt, cin, cout, True)
        else:
            _wb, wt = ops.linear_prep(w32, cs32, cin, cout, dtype, want_wb=False)
        dx = torch.empty(x.shape, dtype=dtype, device=x.device)
        ops.linear_fwd(gv, wt, None, ops.pview(dx, cin), M, cout, cin, dtype, dtype, x)
    else:
        dx = _empty(x)
    b32 = _f32(bias)
    dysum = ops.colsum_a1(gv, M, cout, dtype, x) if (has_bias and not ops.wgrad_has_colsum(cin, cout, dtype)) else None
    dw, dcs, dbias = ops.linear_wgrad(xv, gv, M, cin, cout, dtype, x, w32=w32 if has_cs else None, cs5=cs32, bias=b32,
                                      dysum=dysum, want_bias=has_bias, may_defer=False)
    e = _empty(dw[0])
    dcs = list(dcs) if has_cs else [e.clone() for _ in range(5)]
    return (dx, *dw, dbias if has_bias else e.clone(), *dcs)


@linear_d8_bwd.register_fake
def _(dy, x, wA1, wA2, wB1, wB2, wE, bias, rs, sA1, sA2, sB1, sB2, sE, cin, cout, rps, fused, need_dx, pwt=None):
    f = lambda t: t.new_empty(t.shape, dtype=torch.float32)
    e = lambda: x.new_empty(0, dtype=torch.float32)
    dx = x.new_empty(x.shape) if need_dx else x.new_empty(0)
    dcs = [f(s) for s in (sA1, sA2, sB1, sB2, sE)] if sA1 is not None else [e() for _ in range(5)]
    return (dx, f(wA1), f(wA2), f(wB1), f(wB2), f(wE), f(bias) if bias is not None else e(), *dcs)


def _lin_setup(ctx, inputs, output):
    x, wA1, wA2, wB1, wB2, wE, bias, resid, rs, sA1, sA2, sB1, sB2, sE, cin, cout, rps, _pwb, pwt = inputs
    ctx.has = (bias is not None, resid is not None, rs is not None, sA1 is not None, pwt is not None)
    opt = [t for t in (bias, rs, sA1, sA2, sB1, sB2, sE, pwt) if t is not None]
    ctx.save_for_backward(x, wA1, wA2, wB1, wB2, wE, *opt)
    ctx.meta = (cin, cout, rps)
    ctx.wdtypes = tuple(w.dtype for w in (wA1, wA2, wB1, wB2, wE))


def _lin_backward(ctx, dy):
    x, wA1, wA2, wB1, wB2, wE, *opt = ctx.saved_tensors
    has_bias, fused, has_rs, has_cs, has_pwt = ctx.has
    opt = list(opt)
    bias = opt.pop(0) if has_bias else None
    rs = opt.pop(0) if has_rs else None
    pwt = opt.pop() if has_pwt else None
    cs = opt if has_cs else [None] * 5
    cin, cout, rps = ctx.meta
    outs = torch.ops.octic.linear_d8_bwd(dy, x, wA1, wA2, wB1, wB2, wE, bias, rs, *cs, cin, cout, rps, fused,
                                         ctx.needs_input_grad[0], pwt)
    dx = outs[0] if ctx.needs_input_grad[0] else None
    dw = [d.to(t) for d, t in zip(outs[1:6], ctx.wdtypes)]
    dcs = list(outs[7:12]) if has_cs else [None] * 5
    return (dx, *dw, outs[6] if has_bias else None, dy if fused else None, None, *dcs, None, None, None, None, None)


linear_d8.register_autograd(_lin_backward, setup_context=_lin_setup)


# ------------------------------------------------------------------------------------------------ AttentionD8 core
@_lib.custom_op("octic::attn_packed", mutates_args=())
def attn_packed(qkv: Tensor, H: int, c: int, scale: float) -> Tuple[Tensor, Tensor]:
    """qkv packed [B,T,3*8c] bf16 (LinearD8 output) -> (o packed [B,T,8c], lse [B,H,T])."""
    return ops.attn_fwd_packed(_c(qkv), H, c, scale)


@attn_packed.register_fake
def _(qkv, H, c, scale):
    B, T = qkv.shape[0], qkv.shape[1]
    if qkv.shape[-1] != 24 * c:
        raise ValueError("attn_packed: qkv must be [B, T, 3 * 8c]")
    return qkv.new_empty((B, T, 8 * c)), qkv.new_empty((B, H, T), dtype=torch.float32)


@_lib.custom_op("octic::attn_packed_bwd", mutates_args=())
def attn_packed_bwd(do: Tensor, qkv: Tensor, o: Tensor, lse: Tensor, H: int, c: int, scale: float) -> Tensor:
    return ops.attn_bwd_packed(_c(qkv), o, _c(do.to(qkv.dtype)), lse, H, c, scale)


@attn_packed_bwd.register_fake
def _(do, qkv, o, lse, H, c, scale):
    return torch.empty_like(qkv, memory_format=torch.contiguous_format)


def _ap_setup(ctx, inputs, output):
    qkv, H, c, scale = inputs
    ctx.save_for_backward(qkv, output[0], output[1])
    ctx.meta = (H, c, scale)


def _ap_backward(ctx, do, _glse):
    qkv, o, lse = ctx.saved_tensors
    return torch.ops.octic.attn_packed_bwd(do, qkv, o, lse, *ctx.meta), None, None, None


attn_packed.register_autograd(_ap_backward, setup_context=_ap_setup)


# ------------------------------------------------------------------------------------------------ lift (patch embedding)
@_lib.custom_op("octic::lift", mutates_args=())
def lift(img: Tensor, wfull: Tensor, bias_full: Optional[Tensor], pos: Optional[Tensor], cls_row: Optional[Tensor], p: int,
         bf16: bool) -> Tuple[Tensor, Tensor]:
    """-> (tokens [B, tok0 + G*G, D] f32, patches [B*G*G, Kpad]): im2col + one GEMM against the symmetry-expanded kernels,
    positional embedding and cls row fused."""
    dtype = torch.bfloat16 if bf16 else torch.float32
    B, Cin, Hh, Ww = img.shape
    D, K = wfull.shape
    Kpad = (K + 7) // 8 * 8
    n_p = (Hh // p) * (Ww // p)
    tok0 = 0 if cls_row is None else 1
    patches = ops.im2col(img, p, Kpad, dtype)
    wpad = torch.zeros((D, Kpad), dtype=dtype, device=img.device)
    wpad[:, :K] = wfull.detach()
    out = torch.empty((B, tok0 + n_p, D), dtype=torch.float32, device=img.device)
    ops.lift_gemm(patches, wpad, _f32(bias_full), _f32(pos), out, B, n_p, tok0, Kpad, D)
    if tok0:
        out[:, 0] = cls_row.detach().float()
    return out, patches


@lift.register_fake
def _(img, wfull, bias_full, pos, cls_row, p, bf16):
    B, Cin, Hh, Ww = img.shape
    D, K = wfull.shape
    n_p = (Hh // p) * (Ww // p)
    tok0 = 0 if cls_row is None else 1
    return (img.new_empty((B, tok0 + n_p, D), dtype=torch.float32),
            img.new_empty((B * n_p, (K + 7) // 8 * 8), dtype=torch.bfloat16 if bf16 else torch.float32))


@_lib.custom_op("octic::lift_wgrad", mutates_args=())
def lift_wgrad(patches: Tensor, dtok: Tensor, K: int) -> Tensor:
    """dW [D, K] f32 = dtok^T patches (dtok [B, G*G, D]: the cotangent of the patch tokens, cls rows removed)."""
    D = dtok.shape[-1]
    d2 = _c(dtok.to(patches.dtype)).reshape(patches.shape[0], D)
    return _c(ops.lift_wgrad(patches, d2, patches.shape[1], D)[:, :K])


@lift_wgrad.register_fake
def _(patches, dtok, K):
    return patches.new_empty((dtok.shape[-1], K), dtype=torch.float32)


def _lift_setup(ctx, inputs, output):
    img, wfull, bias_full, pos, cls_row, p, bf16 = inputs
    ctx.save_for_backward(output[1])
    ctx.meta = (wfull.shape[1], 0 if cls_row is None else 1, bias_full is not None, pos is not None)
    ctx.wdtype = wfull.dtype


def _lift_backward(ctx, dout, _gp):
    (patches,) = ctx.saved_tensors
    K, tok0, has_bias, has_pos = ctx.meta
    dtok = dout[:, tok0:]
    dw = torch.ops.octic.lift_wgrad(patches, dtok, K).to(ctx.wdtype)
    dpos = dtok.sum(0) if (has_pos or has_bias) else None
    dbias = dpos.sum(0) if has_bias else None
    dcls = dout[:, 0].sum(0) if tok0 else None
    return None, dw, dbias, dpos if has_pos else None, dcls, None, None


lift.register_autograd(_lift_backward, setup_context=_lift_setup)



# ------------------------------------------------------------------------------------------------ timm Mlp (fc1 + GELU + fc2)
def _mlp_hip_ok(rows, D, Hd):
    return (_hip_gemm_ok(rows, Hd, D) and _hip_gemm_ok(rows, D, Hd) and ops.dense_wgrad_ok(rows, Hd, D)
            and ops.dense_wgrad_ok(rows, D, Hd))


@_lib.custom_op("octic::dense_mlp", mutates_args=())
def dense_mlp(y: Tensor, w1: Tensor, b1: Optional[Tensor], w2: Tensor, b2: Optional[Tensor], w1b: Optional[Tensor] = None,
              w1t: Optional[Tensor] = None, w2b: Optional[Tensor] = None, w2t: Optional[Tensor] = None
              ) -> Tuple[Tensor, Tensor, Tensor]:
    """fc2(gelu(fc1(y))) of timm's Mlp (deit/vit.py:131-134 `self.mlp(self.norm2(x))`) on bf16 rows -> (out, factor, a):
    fc1's epilogue leaves a = gelu(h) and factor = gelu'(h) (one erf for both, mode 4 of octic_dense_gemm_nt); the backward
    multiplies by the stored factor inside fc2's input-gradient epilogue (mode 5) - no separate GELU-backward pass, as in the
    eager path (functional.DenseMlpFn).  w?b / w?t: the optimizer's static bf16 / transposed copies (see dense_linear)."""
    Hd, D = w1.shape
    y2 = _c(y).reshape(-1, D)
    if y2.dtype != torch.bfloat16 or not _mlp_hip_ok(y2.shape[0], D, Hd):
        raise RuntimeError("octic::dense_mlp runs bf16 rows of shapes csrc/dense_gemm.hip and dense_wgrad.hip take; "
                           "other shapes go through octic::dense_linear")
    tok = int(y.shape[-2]) if y.dim() >= 3 else 0
    if w1b is None or w1b.dtype != torch.bfloat16:
        w1b = w1.detach().to(torch.bfloat16)
    if w2b is None or w2b.dtype != torch.bfloat16:
        w2b = w2.detach().to(torch.bfloat16)
    f, a = ops.dense_gemm_nt(y2, w1b, 4, bias=_f32(b1), tokens=tok)
    out = ops.dense_gemm_nt(a, w2b, 0, bias=_f32(b2), tokens=tok)
    return out.view(y.shape), f, a


@dense_mlp.register_fake
def _(y, w1, b1, w2, b2, w1b=None, w1t=None, w2b=None, w2t=None):
    rows = y.numel() // y.shape[-1]
    return (torch.empty_like(y, memory_format=torch.contiguous_format), y.new_empty((rows, w1.shape[0])),
            y.new_empty((rows, w1.shape[0])))


@_lib.custom_op("octic::dense_mlp_bwd", mutates_args=())
def dense_mlp_bwd(dout: Tensor, y: Tensor, f: Tensor, a: Tensor, w1: Tensor, w2: Tensor, need_dy: bool,
                  w1t: Optional[Tensor] = None, w2t: Optional[Tensor] = None
                  ) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    """-> (dy, dW1, db1, dW2, db2), all parameter gradients f32."""
    Hd, D = w1.shape
    y2 = _c(y).reshape(-1, D)
    g2 = _c(dout.to(torch.bfloat16)).reshape(-1, D)
    tok = int(y.shape[-2]) if y.dim() >= 3 else 0
    if w2t is None or w2t.dtype != torch.bfloat16:
        w2t = w2.detach().to(torch.bfloat16).t().contiguous()
    db2 = ops.dense_colsum(g2)
    dh, db1 = ops.dense_gemm_nt(g2, w2t, 5, h=f, want_colsum=True, tokens=tok)       # factor * (dout W2), + fc1's bias gradient
    dw2 = ops.dense_wgrad_tn(g2, a)
    dw1 = ops.dense_wgrad_tn(dh, y2)
    if need_dy:
        if w1t is None or w1t.dtype != torch.bfloat16:
            w1t = w1.detach().to(torch.bfloat16).t().contiguous()
        dy = ops.dense_gemm_nt(dh, w1t, 0, tokens=tok).view(y.shape)
    else:
        dy = y.new_empty(0)
    return dy, dw1, db1, dw2, db2


@dense_mlp_bwd.register_fake
def _(dout, y, f, a, w1, w2, need_dy, w1t=None, w2t=None):
    f32 = lambda t: t.new_empty(t.shape, dtype=torch.float32)
    return ((torch.empty_like(y, memory_format=torch.contiguous_format) if need_dy else y.new_empty(0)), f32(w1),
            w1.new_empty(w1.shape[0], dtype=torch.float32), f32(w2), w2.new_empty(w2.shape[0], dtype=torch.float32))


def _mlp_setup(ctx, inputs, output):
    y, w1, b1, w2, b2, _w1b, w1t, _w2b, w2t = inputs
    ctx.has_t = (w1t is not None, w2t is not None)
    ctx.save_for_backward(y, output[1], output[2], w1, w2, *[t for t in (w1t, w2t) if t is not None])
    ctx.meta = (b1 is not None, b2 is not None, w1.dtype, w2.dtype, None if b1 is None else b1.dtype,
                None if b2 is None else b2.dtype)


def _mlp_backward(ctx, dout, _gf, _ga):
    y, f, a, w1, w2, *rest = ctx.saved_tensors
    rest = list(rest)
    w1t = rest.pop(0) if ctx.has_t[0] else None
    w2t = rest.pop(0) if ctx.has_t[1] else None
    has_b1, has_b2, w1d, w2d, b1d, b2d = ctx.meta
    dy, dw1, db1, dw2, db2 = torch.ops.octic.dense_mlp_bwd(dout, y, f, a, w1, w2, ctx.needs_input_grad[0], w1t, w2t)
    return ((dy if ctx.needs_input_grad[0] else None), dw1.to(w1d), (db1.to(b1d) if has_b1 else None), dw2.to(w2d),
            (db2.to(b2d) if has_b2 else None), None, None, None, None)


dense_mlp.register_autograd(_mlp_backward, setup_context=_mlp_setup)

# ------------------------------------------------------------------------------------------------ hand-off
@_lib.custom_op("octic::handoff_cat", mutates_args=())
def handoff_cat(x: Tensor, c: int, out_bf16: bool) -> Tensor:
    return ops.handoff_cat_fwd(_c(x.float()), c, torch.bfloat16 if out_bf16 else torch.float32)


@handoff_cat.register_fake
def _(x, c, out_bf16):
    return x.new_empty(x.shape, dtype=torch.bfloat16 if out_bf16 else torch.float32)


@_lib.custom_op("octic::handoff_cat_bwd", mutates_args=())
def handoff_cat_bwd(g: Tensor, c: int) -> Tensor:
    return ops.handoff_cat_bwd(g, c)


@handoff_cat_bwd.register_fake
def _(g, c):
    return g.new_empty(g.shape, dtype=torch.float32)


handoff_cat.register_autograd(lambda ctx, g: (torch.ops.octic.handoff_cat_bwd(g, ctx.c), None, None),
                              setup_context=lambda ctx, inputs, output: setattr(ctx, "c", inputs[1]))


@_lib.custom_op("octic::power_spectrum", mutates_args=())
def power_spectrum(x: Tensor, c: int, out_bf16: bool) -> Tensor:
    return ops.power_spectrum_fwd(_c(x.float()), c, torch.bfloat16 if out_bf16 else torch.float32)


@power_spectrum.register_fake
def _(x, c, out_bf16):
    return x.new_empty(x.shape[:-1] + (6 * c,), dtype=torch.bfloat16 if out_bf16 else torch.float32)


@_lib.custom_op("octic::power_spectrum_bwd", mutates_args=())
def power_spectrum_bwd(g: Tensor, x: Tensor, c: int) -> Tensor:
    return ops.power_spectrum_bwd(g, _c(x.float()), c)


@power_spectrum_bwd.register_fake
def _(g, x, c):
    return x.new_empty(x.shape, dtype=torch.float32)


def _ps_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0])
    ctx.c = inputs[1]


power_spectrum.register_autograd(lambda ctx, g: (torch.ops.octic.power_spectrum_bwd(g, ctx.saved_tensors[0], ctx.c), None, None),
                                 setup_context=_ps_setup)


# ================================================================================================ standard half
@_lib.custom_op("octic::dense_layernorm", mutates_args=())
def dense_layernorm(x: Tensor, w: Optional[Tensor], b: Optional[Tensor], eps: float, out_bf16: bool) -> Tuple[Tensor, Tensor]:
    """nn.LayerNorm over the last dimension of the f32 stream -> (y in the compute dtype, stats [rows, 2] = mean, rstd)."""
    return ops.dense_layernorm_fwd(_c(x.float()), _f32(w), _f32(b), eps, torch.bfloat16 if out_bf16 else torch.float32)


@dense_layernorm.register_fake
def _(x, w, b, eps, out_bf16):
    d = x.shape[-1]
    return (x.new_empty(x.shape, dtype=torch.bfloat16 if out_bf16 else torch.float32),
            x.new_empty((x.numel() // d, 2), dtype=torch.float32))


@_lib.custom_op("octic::dense_layernorm_bwd", mutates_args=())
def dense_layernorm_bwd(gy: Tensor, x: Tensor, w: Optional[Tensor], stats: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    dx, dw, db = ops.dense_layernorm_bwd(_c(gy), _c(x.float()), _f32(w), stats, None, want_param_grads=True)
    return dx, dw, db


@dense_layernorm_bwd.register_fake
def _(gy, x, w, stats):
    d = x.shape[-1]
    return x.new_empty(x.shape, dtype=torch.float32), x.new_empty(d, dtype=torch.float32), x.new_empty(d, dtype=torch.float32)


def _dln_setup(ctx, inputs, output):
    x, w, b, eps, out_bf16 = inputs
    ctx.save_for_backward(x, output[1], *([w] if w is not None else []))
    ctx.has = (w is not None, b is not None)
    ctx.xdtype = x.dtype


def _dln_backward(ctx, gy, _gs):
    x, stats, *w = ctx.saved_tensors
    dx, dw, db = torch.ops.octic.dense_layernorm_bwd(gy, x, w[0] if w else None, stats)
    return dx.to(ctx.xdtype), dw if ctx.has[0] else None, db if ctx.has[1] else None, None, None


dense_layernorm.register_autograd(_dln_backward, setup_context=_dln_setup)


def _hip_gemm_ok(rows, N, K):
    return K % 64 == 0 and K >= 128 and N % 64 == 0 and N >= 128 and rows * max(N, K) * 2 < 2 ** 31


@_lib.custom_op("octic::dense_linear", mutates_args=())
def dense_linear(x: Tensor, w: Tensor, b: Optional[Tensor], gelu: bool, wb: Optional[Tensor] = None,
                 wt: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """nn.Linear on bf16 rows: (x W^T + b, -) or, with gelu, (gelu(h), h) with h = x W^T + b from ONE epilogue
    (deit/vit.py Mlp: fc1 + act).  csrc/dense_gemm.hip where the shape allows, the BLAS library otherwise.
    wb / wt: the bf16 copy of w and its transposed copy that train.FusedLamb rewrites in place after every step
    (functional.DenseWeightCache.static_nt) - given, no cast / transpose launches run here or in the backward."""
    N, K = w.shape
    x2 = _c(x).reshape(-1, K)
    if wb is None or wb.dtype != torch.bfloat16:
        wb = w.detach().to(torch.bfloat16)
    lead = x.shape[:-1]
    tok = int(x.shape[-2]) if x.dim() >= 3 else 0
    if x2.dtype == torch.bfloat16 and _hip_gemm_ok(x2.shape[0], N, K) and N % 8 == 0:
        if gelu:
            h, y = ops.dense_gemm_nt(x2, wb, 1, bias=_f32(b), tokens=tok)
            return y.view(lead + (N,)), h.view(lead + (N,))
        y = ops.dense_gemm_nt(x2, wb, 0, bias=_f32(b), tokens=tok)
        return y.view(lead + (N,)), y.new_empty(0)
    h = torch.nn.functional.linear(x2, wb.to(x2.dtype), None if b is None else b.detach().to(x2.dtype))
    if gelu:
        return torch.nn.functional.gelu(h).view(lead + (N,)), h.view(lead + (N,))
    return h.view(lead + (N,)), h.new_empty(0)


@dense_linear.register_fake
def _(x, w, b, gelu, wb=None, wt=None):
    y = x.new_empty(x.shape[:-1] + (w.shape[0],))
    return y, (torch.empty_like(y) if gelu else x.new_empty(0))


@_lib.custom_op("octic::dense_linear_bwd", mutates_args=())
def dense_linear_bwd(dy: Tensor, x: Tensor, w: Tensor, h: Tensor, gelu: bool, need_dx: bool,
                     wt: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, Tensor]:
    """-> (dx, dW f32, db f32): input gradient on the NT kernel (GELU' in its epilogue when gelu), weight gradient on the TN
    kernel, bias gradient as the column sums of the cotangent."""
    N, K = w.shape
    x2 = _c(x).reshape(-1, K)
    g2 = _c(dy.to(x2.dtype)).reshape(-1, N)
    M = x2.shape[0]
    hip = x2.dtype == torch.bfloat16 and _hip_gemm_ok(M, N, K) and _hip_gemm_ok(M, K, N)
    if need_dx and (wt is None or wt.dtype != torch.bfloat16):
        wt = w.detach().to(torch.bfloat16).t().contiguous()        # [K, N]: the input gradient is an NT problem too
    if gelu:
        h2 = _c(h).reshape(-1, N)
        if x2.dtype == torch.bfloat16:
            dh, db = ops.dense_gelu_bwd(h2, g2, want_colsum=True)
        else:
            hh = h2.float().requires_grad_(True)
            with torch.enable_grad():
                (dh,) = torch.autograd.grad(torch.nn.functional.gelu(hh), hh, g2.float())
            dh = dh.to(x2.dtype)
            db = dh.float().sum(0)
    else:
        dh = g2
        db = ops.dense_colsum(g2) if (g2.dtype == torch.bfloat16 and N % 2 == 0) else g2.float().sum(0)
    if need_dx:
        dx = (ops.dense_gemm_nt(dh, wt, 0, tokens=int(x.shape[-2]) if x.dim() >= 3 else 0) if hip
              else (dh @ wt.t().to(dh.dtype)))
        dx = dx.view(x.shape)
    else:
        dx = x.new_empty(0)
    if dh.dtype == torch.bfloat16 and ops.dense_wgrad_ok(M, N, K):
        dw = ops.dense_wgrad_tn(dh, x2)
    else:
        dw = (dh.float().t() @ x2.float())
    return dx, dw, db


@dense_linear_bwd.register_fake
def _(dy, x, w, h, gelu, need_dx, wt=None):
    return ((torch.empty_like(x, memory_format=torch.contiguous_format) if need_dx else x.new_empty(0)),
            w.new_empty(w.shape, dtype=torch.float32), w.new_empty(w.shape[0], dtype=torch.float32))


def _dl_setup(ctx, inputs, output):
    x, w, b, gelu, _wb, wt = inputs
    ctx.save_for_backward(x, w, output[1], *(() if wt is None else (wt,)))
    ctx.meta = (gelu, b is not None, w.dtype, None if b is None else b.dtype)


def _dl_backward(ctx, dy, _gh):
    x, w, h, *rest = ctx.saved_tensors
    gelu, has_b, wdt, bdt = ctx.meta
    dx, dw, db = torch.ops.octic.dense_linear_bwd(dy, x, w, h, gelu, ctx.needs_input_grad[0], rest[0] if rest else None)
    return (dx if ctx.needs_input_grad[0] else None), dw.to(wdt), (db.to(bdt) if has_b else None), None, None, None


dense_linear.register_autograd(_dl_backward, setup_context=_dl_setup)


@_lib.custom_op("octic::attn_qkv", mutates_args=())
def attn_qkv(qkv: Tensor, scale: float) -> Tuple[Tensor, Tensor]:
    """softmax(q k^T scale) v on the fused projection output qkv [B,T,3,H,hd] (read through strides) -> (o [B,T,H*hd], lse)."""
    qkv = _c(qkv)
    B, T, _, H, hd = qkv.shape
    q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    o = torch.empty((B, T, H, hd), dtype=qkv.dtype, device=qkv.device)
    ov = o.permute(0, 2, 1, 3)
    lse = torch.empty((B, H, T), dtype=torch.float32, device=qkv.device)
    st = q.stride()
    ops.check(ops.lib().octic_attn_fwd(ops._p(q), ops._p(k), ops._p(v), ops._p(o), ops._p(lse), B, H, T, hd, st[0], st[1],
                                       st[2], ov.stride(0), ov.stride(1), ov.stride(2), float(scale), ops._stream(qkv)))
    return o.view(B, T, H * hd), lse


@attn_qkv.register_fake
def _(qkv, scale):
    B, T, three, H, hd = qkv.shape
    return qkv.new_empty((B, T, H * hd)), qkv.new_empty((B, H, T), dtype=torch.float32)


@_lib.custom_op("octic::attn_qkv_bwd", mutates_args=())
def attn_qkv_bwd(do: Tensor, qkv: Tensor, o: Tensor, lse: Tensor, scale: float) -> Tensor:
    qkv = _c(qkv)
    B, T, _, H, hd = qkv.shape
    do = _c(do.to(qkv.dtype)).view(B, T, H, hd)
    dqkv = torch.empty_like(qkv)
    q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    dq, dk, dv = (dqkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    ops.attn_bwd(q, k, v, _c(o).view(B, T, H, hd).permute(0, 2, 1, 3), do.permute(0, 2, 1, 3), lse, scale, dq, dk, dv)
    return dqkv


@attn_qkv_bwd.register_fake
def _(do, qkv, o, lse, scale):
    return torch.empty_like(qkv, memory_format=torch.contiguous_format)


def _aq_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], output[0], output[1])
    ctx.scale = inputs[1]


attn_qkv.register_autograd(
    lambda ctx, do, _gl: (torch.ops.octic.attn_qkv_bwd(do, *ctx.saved_tensors, ctx.scale), None),
    setup_context=_aq_setup)


@_lib.custom_op("octic::scale_residual", mutates_args=())
def scale_residual(x: Tensor, y: Tensor, gamma: Optional[Tensor], rs: Optional[Tensor], rps: int) -> Tensor:
    """x + rs[row / rps] * gamma * y   (x: f32 stream, y: branch output)."""
    return ops.scale_residual_fwd(_c(x.float()), _c(y), _f32(gamma), _f32(rs), rps)


@scale_residual.register_fake
def _(x, y, gamma, rs, rps):
    return x.new_empty(x.shape, dtype=torch.float32)


@_lib.custom_op("octic::scale_residual_bwd", mutates_args=())
def scale_residual_bwd(gout: Tensor, y: Tensor, gamma: Optional[Tensor], rs: Optional[Tensor], rps: int) -> Tuple[Tensor, Tensor]:
    gy, dgamma, _col = ops.scale_residual_bwd(_c(gout.float()), _c(y), _f32(gamma), _f32(rs), rps, want_gamma=gamma is not None,
                                              want_colsum=False)
    return gy, (dgamma if dgamma is not None else gy.new_empty(0, dtype=torch.float32))


@scale_residual_bwd.register_fake
def _(gout, y, gamma, rs, rps):
    return (torch.empty_like(y, memory_format=torch.contiguous_format),
            (gamma.new_empty(gamma.shape, dtype=torch.float32) if gamma is not None else y.new_empty(0, dtype=torch.float32)))


def _sr_setup(ctx, inputs, output):
    x, y, gamma, rs, rps = inputs
    ctx.save_for_backward(y, *([gamma] if gamma is not None else []), *([rs] if rs is not None else []))
    ctx.meta = (gamma is not None, rs is not None, rps, x.dtype, None if gamma is None else gamma.dtype)


def _sr_backward(ctx, gout):
    y, *rest = ctx.saved_tensors
    has_g, has_rs, rps, xdt, gdt = ctx.meta
    rest = list(rest)
    gamma = rest.pop(0) if has_g else None
    rs = rest.pop(0) if has_rs else None
    gy, dgamma = torch.ops.octic.scale_residual_bwd(gout, y, gamma, rs, rps)
    return gout.to(xdt), gy, (dgamma.to(gdt) if has_g else None), None, None


scale_residual.register_autograd(_sr_backward, setup_context=_sr_setup)

ALL_OPS = ("layernorm_d8", "linear_d8", "attn_packed", "lift", "handoff_cat", "power_spectrum", "dense_layernorm",
           "dense_linear", "attn_qkv", "scale_residual")
