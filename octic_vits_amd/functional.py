"""Autograd wrappers of the HIP engine (the PyTorch-ROCm custom-op surface).

Every hot op of the octic block is one ``torch.autograd.Function`` whose forward AND backward are
launches of hand-written gfx950 kernels through the C ABI (``ops.py`` → ``liboctic_hip.so``).
The engine's native tensor is the *packed* token row ``[B, T, 8c] = [A1|A2|B1|B2|E_row0|E_row1]``;
``Octic`` wraps it as the reference's 5-tuple (zero-copy views) so modules stay drop-in.

Precision follows the caller like the reference does: plain fp32 tensors run the exact-f32 MFMA
path (used for the reference's fp32 equivariance tolerances); under
``torch.autocast('cuda', dtype=torch.bfloat16)`` linears/GELU/attention run in bf16 with f32
accumulation while the residual stream, LayerNorm statistics and parameter gradients stay f32
(that is what autocast does to the reference).
"""
import torch

from . import ops


class Octic(tuple):
    """The reference's 5-tuple (A1,A2,B1,B2:[B,T,c]; E:[B,T,2,2c]) as views of one packed tensor."""

    def __new__(cls, packed: torch.Tensor, c: int):
        self = super().__new__(cls, ops.split_packed(packed, c))
        self.packed, self.c = packed, c
        return self


def as_packed(xs):
    """Packed [.., 8c] tensor of an Octic (free) or of any reference-style 5-tuple (one torch.cat)."""
    if isinstance(xs, Octic):
        return xs.packed, xs.c
    if len(xs) != 5:
        raise AssertionError("Input should be a 5-tuple")
    c = xs[0].shape[-1]
    if xs[4].shape[-2:] != (2, 2 * c):
        raise ValueError(f"E irrep must be [..., 2, {2 * c}], got {tuple(xs[4].shape)}")
    return torch.cat([xs[0], xs[1], xs[2], xs[3], xs[4].flatten(-2)], dim=-1), c


_FP16_NOTE = [False]


ATTN_PACKED = True     # AttentionD8: attention kernels read / write the packed irrep rows (no pack / unpack passes)
RAGGED = None          # ragged.Ragged of the [1, R, .] tensor the blocks are looking at (several crop sets as one row tensor:
                       # set by dinov2_models' block loop, consulted by the attention modules), or None


def compute_dtype(t: torch.Tensor):
    """Operand dtype of the octic kernels for `t` under the ambient autocast state.  bf16 autocast -> bf16 (the fast
    path, BASELINE dtype).  fp16 autocast (the reference's DeiT default, deit/engine.py:56) -> float32: the engine has no
    fp16 kernels, so the octic half runs on its exact-f32 MFMA path - a superset of fp16's precision, slower than bf16;
    the standard blocks run the reference's own eager fp16 ops.  Said once on stderr, never silently."""
    if torch.is_autocast_enabled("cuda"):
        dt = torch.get_autocast_dtype("cuda")
        if dt == torch.bfloat16:
            return dt
        if dt == torch.float16:
            if not _FP16_NOTE[0]:
                _FP16_NOTE[0] = True
                import sys
                print("octic engine: fp16 autocast -> the octic blocks compute in float32 (no fp16 kernels; use "
                      "dtype=torch.bfloat16 for the fast path)", file=sys.stderr)
            if t.dtype not in (torch.float32, torch.float16):
                raise TypeError(f"octic engine under fp16 autocast expects float32 / float16 inputs, got {t.dtype}")
            return torch.float32
        raise NotImplementedError(f"octic engine: autocast dtype {dt} is not supported (bfloat16 or float16)")
    if t.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError(f"octic engine supports float32 / bfloat16, got {t.dtype}")
    return t.dtype


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


# ------------------------------------------------------------------------------------------- GELU
class GeluD8PackedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, c):
        x = _c(x)
        y = torch.empty_like(x)
        M = x.numel() // (8 * c)
        ops.gelu_fwd(ops.pview(x, c), ops.pview(y, c), M, c, x.dtype, x)
        ctx.save_for_backward(x)
        ctx.c = c
        return y

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        c = ctx.c
        g = _c(g.to(x.dtype))
        gi = torch.empty_like(x)
        ops.gelu_bwd(ops.pview(g, c), ops.pview(x, c), ops.pview(gi, c), x.numel() // (8 * c), c, x.dtype, x)
        return gi, None


class GeluD8Function(torch.autograd.Function):
    """Drop-in for the reference's one custom-op boundary ``TritonGeluD8Function`` (d8_gelu.py:456-478):
    five separate tensors in, five out; strides are handed to the kernel, nothing is repacked."""

    @staticmethod
    def forward(ctx, x_A1, x_A2, x_B1, x_B2, x_2d):
        xs = tuple(_c(t) for t in (x_A1, x_A2, x_B1, x_B2, x_2d))
        c = xs[0].shape[-1]
        ys = tuple(torch.empty_like(t) for t in xs)
        M = xs[0].numel() // c
        xv, k1 = ops.tview(xs, c)
        yv, k2 = ops.tview(ys, c)
        ops.gelu_fwd(xv, yv, M, c, xs[0].dtype, xs[0])
        ctx.save_for_backward(*xs)
        return ys

    @staticmethod
    def backward(ctx, *gs):
        xs = ctx.saved_tensors
        c = xs[0].shape[-1]
        gs = tuple(_c(g.to(xs[0].dtype)) for g in gs)
        outs = tuple(torch.empty_like(t) for t in xs)
        gv, k1 = ops.tview(gs, c)
        xv, k2 = ops.tview(xs, c)
        ov, k3 = ops.tview(outs, c)
        ops.gelu_bwd(gv, xv, ov, xs[0].numel() // c, c, xs[0].dtype, xs[0])
        return outs


# ------------------------------------------------------------------------------------- LayerNorm
class LayerNormD8Fn(torch.autograd.Function):
    """Returns (y, x): the second output is the input stream itself.  A block that takes its residual connection
    from it gets the residual cotangent added inside the backward kernel (no separate accumulation pass)."""

    @staticmethod
    def forward(ctx, x, a1, a2, b1, b2, ae, beta, eps, c, out_dtype):
        xin = x
        x = _c(x.float())
        alpha = None if a1 is None else [_c(t.float()) for t in (a1, a2, b1, b2, ae)]
        y, stats = ops.layernorm_fwd(x, alpha, None if beta is None else _c(beta.float()), eps, out_dtype, c)
        ctx.save_for_backward(x, stats, *(alpha or []))
        ctx.c, ctx.has_affine, ctx.has_beta = c, alpha is not None, beta is not None
        ctx.set_materialize_grads(False)
        return y, xin.view_as(xin)

    @staticmethod
    def backward(ctx, g, gres):
        x, stats, *alpha = ctx.saved_tensors
        if g is None:
            return (gres,) + (None,) * 9
        alpha = alpha if ctx.has_affine else None
        dres = None if gres is None else _c(gres.float())
        dx, dal, dbeta = ops.layernorm_bwd(_c(g), x, stats, alpha, dres, ctx.c, want_param_grads=ctx.has_affine)
        if dal is None:
            dal = [None] * 5
        return (dx, *dal, dbeta if ctx.has_beta else None, None, None, None)


# ---------------------------------------------------------------------------------------- Linear
class WeightPrep:
    """Compute-dtype copies of a LinearD8's weights: ``wb`` for the forward GEMM and ``wt`` — transposed,
    with the layer-scale folded in — for the input-gradient GEMM (dX = dY·diag(cs)·W)."""

    def __init__(self):
        self.key = None
        self.wb = self.wt = None
        self.last = None            # (w5, cs5, dtype, cin, cout) of the latest call: what a PrepBatch needs
        self.flat = None            # (wb, wt) flat bf16 buffers a PrepBatch refreshes IN PLACE after every optimizer step: what a
                                    # traced graph may read as plain inputs (dispatch.py) - None while this object prepares itself

    @staticmethod
    def _key(w5, cs5, dtype):
        return (dtype, tuple((w.data_ptr(), w._version) for w in w5),
                None if cs5 is None else tuple((s.data_ptr(), s._version) for s in cs5))

    def get(self, w5, cs5, dtype, cin, cout):
        key = self._key(w5, cs5, dtype)
        if key != self.key:
            with torch.no_grad():
                w32 = [_c(w.detach().float()) for w in w5]
                cs32 = None if cs5 is None else [_c(s.detach().float()) for s in cs5]
                wb, self.wt = ops.linear_prep(w32, cs32, cin, cout, dtype, want_wb=(dtype != torch.float32))
                self.wb = wb if wb is not None else w32   # f32: the master weights are the forward operand
            self.key = key
            self.flat = None
        self.last = (w5, cs5, dtype, cin, cout)
        return self.wb, self.wt

    def adopt(self, wb, wt, flat=None):
        """Copies produced elsewhere (PrepBatch) are current for the parameters' present versions."""
        w5, cs5, dtype, _, _ = self.last
        self.wb, self.wt = wb, wt
        self.key = self._key(w5, cs5, dtype)
        self.flat = flat


class PrepBatch:
    """All bf16 LinearD8 weight preparations of a model in ONE launch (octic_linear_d8_prep_batch), run by the fused
    optimizer right after its update instead of 64 per-layer launches at the next forward."""

    def __init__(self, preps):
        import numpy as np
        self.preps = [p for p in preps if p.last is not None and p.last[2] == torch.bfloat16
                      and all(t.dtype == torch.float32 and t.is_contiguous() for t in p.last[0])
                      and (p.last[1] is None or all(t.dtype == torch.float32 and t.is_contiguous() for t in p.last[1]))]
        if not self.preps:
            self.items = None
            return
        dev = self.preps[0].last[0][0].device
        dt = np.dtype([("w", "<u8", 5), ("cs", "<u8", 5), ("wb", "<u8"), ("wt", "<u8"), ("cin", "<i4"), ("cout", "<i4"),
                       ("block_begin", "<i4"), ("block_count", "<i4")])
        tab = np.zeros(len(self.preps), dtype=dt)
        self.bufs, self.flats, blocks = [], [], 0
        for i, p in enumerate(self.preps):
            w5, cs5, dtype, cin, cout = p.last
            n = 8 * cin * cout
            wb = torch.empty(n, dtype=dtype, device=dev)
            wt = torch.empty(n, dtype=dtype, device=dev)
            self.bufs.append((ops.prep_views(wb, cin, cout, False), ops.prep_views(wt, cin, cout, True)))
            self.flats.append((wb, wt))
            cnt = ops.lib().octic_linear_d8_prep_batch_blocks(cin, cout)   # one 64 x 64 tile per workgroup
            tab[i]["w"] = [t.data_ptr() for t in w5]
            tab[i]["cs"] = [0] * 5 if cs5 is None else [t.data_ptr() for t in cs5]
            tab[i]["wb"], tab[i]["wt"] = wb.data_ptr(), wt.data_ptr()
            tab[i]["cin"], tab[i]["cout"] = cin, cout
            tab[i]["block_begin"], tab[i]["block_count"] = blocks, cnt
            blocks += cnt
        self.total_blocks = blocks
        self.items = torch.from_numpy(tab.view(np.uint8).copy()).to(dev)
        self._keep = [p.last for p in self.preps]        # the parameters whose addresses are in the table
        self._ptrs = self._current_ptrs()

    def _current_ptrs(self):
        return tuple(t.data_ptr() for p in self.preps for t in (tuple(p.last[0]) + tuple(p.last[1] or ())))

    def stale(self):
        """True when a parameter was re-allocated since the table was built (the table holds raw addresses)."""
        return self.items is not None and self._current_ptrs() != self._ptrs

    def run(self):
        if self.items is None:
            return
        ops.check(ops.lib().octic_linear_d8_prep_batch(ops._p(self.items), len(self.preps), self.total_blocks,
                                                       ops.dt_code(torch.bfloat16), ops._stream(self.items)))
        for p, (wb, wt), flat in zip(self.preps, self.bufs, self.flats):
            p.adopt(wb, wt, flat)


class LinearD8Fn(torch.autograd.Function):
    """y = resid + rs*cs*(x W^T + b)   (resid / rs / cs optional).  See octic_hip.h for the math."""

    @staticmethod
    def forward(ctx, x, wA1, wA2, wB1, wB2, wE, bias, resid, rs, sA1, sA2, sB1, sB2, sE, cin, cout, rps, dtype, prep):
        w5 = (wA1, wA2, wB1, wB2, wE)
        cs5 = None if sA1 is None else (sA1, sA2, sB1, sB2, sE)
        ops._require_cuda(x)
        fused = resid is not None
        out_dtype = resid.dtype if fused else dtype
        x_in_dtype = x.dtype
        if x.dtype != dtype:
            x = ops.cast_rowscale(_c(x.float()), None, 1, dtype, cin)
        x = _c(x)
        wb, wt = prep.get(w5, cs5, dtype, cin, cout)
        lead = x.shape[:-1]
        M = x.numel() // (8 * cin)
        y = torch.empty(lead + (8 * cout,), dtype=out_dtype, device=x.device)
        b32 = None if bias is None else _c(bias.detach().float())
        cs32 = None if cs5 is None else [_c(s.detach().float()) for s in cs5]
        rs32 = None if rs is None else _c(rs.float())
        rv = ops.pview(_c(resid), cout) if fused else None
        ops.linear_fwd(ops.pview(x, cin), wb, b32, ops.pview(y, cout), M, cin, cout, dtype, out_dtype, x,
                       resid_v=rv, rs=rs32, rps=rps, cs5=cs32)
        ctx.save_for_backward(x, rs32, b32, *w5, *(cs32 or []))
        ctx.meta = (cin, cout, rps, dtype, fused, cs5 is not None, bias is not None, x_in_dtype, wt)
        return y

    @staticmethod
    def backward(ctx, dy):
        cin, cout, rps, dtype, fused, has_cs, has_bias, x_in_dtype, wt = ctx.meta
        x, rs32, b32, *rest = ctx.saved_tensors
        w5, cs32 = rest[:5], (rest[5:] if has_cs else None)
        M = x.numel() // (8 * cin)
        dy = _c(dy)
        if fused or rs32 is not None or dy.dtype != dtype:
            g = ops.cast_rowscale(_c(dy.float()), rs32, rps, dtype, cout)  # cotangent of the branch
        else:
            g = dy
        gv, xv = ops.pview(g, cout), ops.pview(x, cin)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(x.shape, dtype=dtype, device=x.device)
            ops.linear_fwd(gv, wt, None, ops.pview(dx, cin), M, cout, cin, dtype, dtype, x)
            if dx.dtype != x_in_dtype:
                dx = dx.to(x_in_dtype)
        # bias gradient = column sums of the invariant block of g: the bf16 wgrad kernel produces them on the side
        # (one extra MFMA row); other paths run the column-sum kernel
        dysum = ops.colsum_a1(gv, M, cout, dtype, x) if (has_bias and not ops.wgrad_has_colsum(cin, cout, dtype)) else None
        w32 = [_c(w.detach().float()) for w in w5] if has_cs else None
        f32_masters = all(w.dtype == torch.float32 for w in w5)      # else the casts below read dw at once: no deferral
        dw, dcs, dbias = ops.linear_wgrad(xv, gv, M, cin, cout, dtype, x, w32=w32, cs5=cs32, bias=b32, dysum=dysum,
                                          want_bias=has_bias, may_defer=f32_masters, wparams=w5 if f32_masters else None)
        if not f32_masters:
            dw = [d.to(w.dtype) for d, w in zip(dw, w5)]
        dcs = dcs if has_cs else [None] * 5
        return (dx, *dw, dbias, dy if fused else None, None, *dcs, None, None, None, None, None)


class LinearD8NormFn(torch.autograd.Function):
    """LinearD8Fn with the fused residual tail (y = resid + rs*cs*(x W^T + b), the new f32 stream) followed by the
    LayerNormD8 that opens the NEXT branch: returns (y, LayerNorm(y)).  The forward is the two kernels one after the other
    (the GEMM epilogue owns column slices, not rows); the point is the backward: the norm's backward kernel, which
    produces the stream cotangent dx anyway, also stores bf16(rs * dx) - the cotangent of this layer's branch - so the
    cast_rowscale pass over dx disappears (functional.OCTIC_NEXT_NORM)."""

    @staticmethod
    def forward(ctx, x, wA1, wA2, wB1, wB2, wE, bias, resid, rs, sA1, sA2, sB1, sB2, sE, cin, cout, rps, dtype, prep,
                a1, a2, b1, b2, ae, beta, eps):
        w5 = (wA1, wA2, wB1, wB2, wE)
        cs5 = None if sA1 is None else (sA1, sA2, sB1, sB2, sE)
        ops._require_cuda(x)
        x_in_dtype = x.dtype
        if x.dtype != dtype:
            x = ops.cast_rowscale(_c(x.float()), None, 1, dtype, cin)
        x = _c(x)
        wb, wt = prep.get(w5, cs5, dtype, cin, cout)
        M = x.numel() // (8 * cin)
        y = torch.empty(x.shape[:-1] + (8 * cout,), dtype=resid.dtype, device=x.device)
        b32 = None if bias is None else _c(bias.detach().float())
        cs32 = None if cs5 is None else [_c(s.detach().float()) for s in cs5]
        rs32 = None if rs is None else _c(rs.float())
        ops.linear_fwd(ops.pview(x, cin), wb, b32, ops.pview(y, cout), M, cin, cout, dtype, resid.dtype, x,
                       resid_v=ops.pview(_c(resid), cout), rs=rs32, rps=rps, cs5=cs32)
        alpha = None if a1 is None else [_c(t.float()) for t in (a1, a2, b1, b2, ae)]
        yn, stats = ops.layernorm_fwd(y, alpha, None if beta is None else _c(beta.float()), eps, dtype, cout)
        ctx.save_for_backward(x, rs32, b32, y, stats, *w5, *(cs32 or []), *(alpha or []))
        ctx.meta = (cin, cout, rps, dtype, cs5 is not None, bias is not None, x_in_dtype, wt, alpha is not None,
                    beta is not None)
        ctx.set_materialize_grads(False)
        return y, yn

    @staticmethod
    def backward(ctx, dy, dyn):
        cin, cout, rps, dtype, has_cs, has_bias, x_in_dtype, wt, has_affine, has_beta = ctx.meta
        x, rs32, b32, y, stats, *rest = ctx.saved_tensors
        w5 = rest[:5]
        cs32 = rest[5:10] if has_cs else None
        alpha = rest[(10 if has_cs else 5):] if has_affine else None
        M = x.numel() // (8 * cin)
        dal, dbeta, g = [None] * 5, None, None
        if dyn is not None:
            dres = None if dy is None else _c(dy.float())
            gn = _c(dyn)
            if ops.layernorm_bwd_cast_ok(gn, y, cout) and dtype == torch.bfloat16:
                dy, dal_, dbeta, g = ops.layernorm_bwd_cast(gn, y, stats, alpha, dres, cout, rs32, rps,
                                                            want_param_grads=has_affine)
            else:
                dy, dal_, dbeta = ops.layernorm_bwd(gn, y, stats, alpha, dres, cout, want_param_grads=has_affine)
            dal = dal_ if dal_ is not None else dal
        dy = _c(dy)
        if g is None:
            g = ops.cast_rowscale(_c(dy.float()), rs32, rps, dtype, cout)  # cotangent of the branch
        gv, xv = ops.pview(g, cout), ops.pview(x, cin)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(x.shape, dtype=dtype, device=x.device)
            ops.linear_fwd(gv, wt, None, ops.pview(dx, cin), M, cout, cin, dtype, dtype, x)
            if dx.dtype != x_in_dtype:
                dx = dx.to(x_in_dtype)
        dysum = ops.colsum_a1(gv, M, cout, dtype, x) if (has_bias and not ops.wgrad_has_colsum(cin, cout, dtype)) else None
        w32 = [_c(w.detach().float()) for w in w5] if has_cs else None
        f32_masters = all(w.dtype == torch.float32 for w in w5)      # else the casts below read dw at once: no deferral
        dw, dcs, dbias = ops.linear_wgrad(xv, gv, M, cin, cout, dtype, x, w32=w32, cs5=cs32, bias=b32, dysum=dysum,
                                          want_bias=has_bias, may_defer=f32_masters, wparams=w5 if f32_masters else None)
        if not f32_masters:
            dw = [d.to(w.dtype) for d, w in zip(dw, w5)]
        dcs = dcs if has_cs else [None] * 5
        return (dx, *dw, dbias, dy, None, *dcs, None, None, None, None, None, *dal, dbeta if has_beta else None, None)


# --------------------------------------------------------------------------------- head packing
class PackHeadsFn(torch.autograd.Function):
    """packed qkv [B,T,3*8c] -> (q, k, v) each [B,H,T,8c/H].  Three separate outputs so autograd hands the three
    SDPA gradients straight back (no zero-fill + add of a stacked [3,...] tensor)."""

    @staticmethod
    def forward(ctx, qkv, H, c):
        B, T = qkv.shape[0], qkv.shape[1]
        ctx.meta = (B, T, H, c)
        q, k, v = ops.pack_heads(_c(qkv), B, T, H, c, 3)
        return q, k, v

    @staticmethod
    def backward(ctx, gq, gk, gv):
        B, T, H, c = ctx.meta
        return ops.unpack_heads([gq, gk, gv], B, T, H, c), None, None


class UnpackHeadsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, o, c):
        B, H, T, _ = o.shape
        ctx.meta = (B, T, H, c)
        return ops.unpack_heads([o], B, T, H, c)

    @staticmethod
    def backward(ctx, g):
        B, T, H, c = ctx.meta
        return ops.pack_heads(_c(g), B, T, H, c, 1)[0], None


# ------------------------------------------------------------------------------------- attention
class AttnFn(torch.autograd.Function):
    """softmax(q k^T / sqrt(hd)) v for separate q, k, v of shape [B,H,T,hd] (bf16) — HIP forward and backward."""

    @staticmethod
    def forward(ctx, q, k, v, scale):
        q, k, v = _c(q), _c(k), _c(v)
        o, lse = ops.attn_fwd(q, k, v, scale)
        ctx.save_for_backward(q, k, v, o, lse)
        ctx.scale = scale
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse = ctx.saved_tensors
        do = _c(do)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        ops.attn_bwd(q, k, v, o, do, lse, ctx.scale, dq, dk, dv)
        return dq, dk, dv, None


class AttnPackedFn(torch.autograd.Function):
    """AttentionD8's core on packed rows (reference d8_layers.py:631-656): qkv [B,T,3*8c] -> o [B,T,8c].  The head
    vectors are gathered from / scattered to the irrep pieces inside the attention kernels, so the four pack / unpack
    passes of a training step (and their 0.5 GB of traffic per block) do not exist."""

    @staticmethod
    def forward(ctx, qkv, H, c, scale):
        qkv = _c(qkv)
        o, lse = ops.attn_fwd_packed(qkv, H, c, scale)
        ctx.save_for_backward(qkv, o, lse)
        ctx.meta = (H, c, scale)
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, o, lse = ctx.saved_tensors
        H, c, scale = ctx.meta
        return ops.attn_bwd_packed(qkv, o, _c(do), lse, H, c, scale), None, None, None


class AttnFusedQKVFn(torch.autograd.Function):
    """Attention on a fused projection output qkv [B,T,3,H,hd] (the layout of the standard block, deit/vit.py:38-39):
    q/k/v are read through strides, the result is written as [B,T,H*hd] and the gradient comes back as ONE
    [B,T,3,H,hd] tensor (no permute copies, no zero-fill + add of three partial gradients)."""

    @staticmethod
    def forward(ctx, qkv, scale):
        qkv = _c(qkv)
        B, T, _, H, hd = qkv.shape
        q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
        o = torch.empty((B, T, H, hd), dtype=qkv.dtype, device=qkv.device)
        ov = o.permute(0, 2, 1, 3)
        lse = torch.empty((B, H, T), dtype=torch.float32, device=qkv.device)
        st = q.stride()
        t = ops.KERNEL_TIMER.start()
        ops.check(ops.lib().octic_attn_fwd(ops._p(q), ops._p(k), ops._p(v), ops._p(o), ops._p(lse), B, H, T, hd, st[0], st[1],
                                           st[2], ov.stride(0), ov.stride(1), ov.stride(2), float(scale), ops._stream(qkv)))
        ops.KERNEL_TIMER.stop(t, "attn_fwd_kernel", 4 * q.numel() * 2, 4.0 * B * H * T * T * hd)
        ctx.save_for_backward(qkv, o, lse)
        ctx.scale = scale
        return o.view(B, T, H * hd)

    @staticmethod
    def backward(ctx, do):
        qkv, o, lse = ctx.saved_tensors
        B, T, _, H, hd = qkv.shape
        do = _c(do).view(B, T, H, hd)
        dqkv = torch.empty_like(qkv)
        q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
        dq, dk, dv = (dqkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
        ops.attn_bwd(q, k, v, o.permute(0, 2, 1, 3), do.permute(0, 2, 1, 3), lse, ctx.scale, dq, dk, dv)
        return dqkv, None


def attention_core(q, k, v, dropout_p=0.0):
    """Drop-in for F.scaled_dot_product_attention(q, k, v) on [B,H,T,hd]: HIP kernels for the shapes they cover
    (bf16, short sequences, no dropout), torch SDPA otherwise (f32 exact path, odd head sizes)."""
    B, H, T, hd = q.shape
    if dropout_p == 0.0 and q.is_cuda and ops.attn_supported(T, hd, q.dtype):
        return AttnFn.apply(q, k, v, hd ** -0.5)
    o = torch.nn.functional.scaled_dot_product_attention(q, k, v, dropout_p=dropout_p)
    return o if o.dtype == q.dtype else o.to(q.dtype)       # fp16 autocast hands back fp16: the octic rows stay in q's dtype


# -------------------------------------------------------------------------------------- standard half
class DenseWeightCache:
    """Compute-dtype copy of an nn.Linear's weight and bias, refreshed when the master parameters change."""

    def __init__(self):
        self.key = None
        self.w = self.b = None
        self.wt = self.wt_key = None
        self.static = False         # the copies are refreshed IN PLACE by their producer after every optimizer step (adopt):
        self.static_wt = False      # a traced graph may read them as plain inputs (dispatch.py)

    def static_nt(self):
        """(bf16 W, bf16 W^T or None) when the fused optimizer keeps them current in place, else (None, None)."""
        # (reads nothing that changes from step to step - no keys, no version counters: torch.compile guards on what a traced
        # function reads, and a guard on a version counter would recompile the block after every optimizer step)
        if self.static and self.w is not None and self.w.dtype == torch.bfloat16:
            return self.w, (self.wt if self.static_wt else None)
        return None, None

    @staticmethod
    def _key(w, b, dtype):
        return (dtype, w.data_ptr(), w._version, None if b is None else (b.data_ptr(), b._version))

    def get(self, w, b, dtype):
        key = self._key(w, b, dtype)
        if key != self.key:
            with torch.no_grad():
                self.w = _c(w.detach().to(dtype))
                self.b = None if b is None else _c(b.detach().to(dtype))
            self.key = key
            self.wt, self.wt_key = None, None     # a transposed copy of the old cast is stale with it
            self.static = self.static_wt = False
        return self.w, self.b

    def adopt(self, w, b, w_copy, b_copy, dtype, wt_copy=None):
        """Take compute-dtype copies produced elsewhere (the fused optimizer writes them in its update pass) as the
        current cache content for the parameters' present versions."""
        self.w, self.b = w_copy, b_copy
        self.key = self._key(w, b, dtype)
        self.static = True
        self.static_wt = wt_copy is not None
        if wt_copy is not None:
            self.wt, self.wt_key = wt_copy, self.key

    def get_nt(self, w, b, need_wt=True):
        """(bf16 W [N,K], bf16 W^T [K,N] or None) for the hand-written dense GEMMs (forward / input-gradient operands);
        the transposed copy only exists for layers whose input gradient runs on csrc/dense_gemm.hip."""
        wb, _ = self.get(w, b, torch.bfloat16)
        if not need_wt:
            return wb, None
        if self.wt is None or self.wt_key != self.key:
            with torch.no_grad():
                self.wt = wb.t().contiguous()
            self.wt_key = self.key
        return wb, self.wt


class SoftCrossEntropyFn(torch.autograd.Function):
    """Per row of the student logits s [N, K]: -sum_k t_k log_softmax(s / temp)_k against the teacher probabilities
    tprob [Nt, K] (f32, no gradient; row r uses tprob[r % Nt]) -> [N] f32.  One read of s and tprob forward, one read of each
    and one write backward (csrc/ssl_loss.hip) - dinov2/loss/dino_clstoken_loss.py:78-92, ibot_patch_loss.py:26-34."""

    @staticmethod
    def forward(ctx, s, tprob, temp):
        tprob = _c(tprob.detach().float())
        loss, lse, tsum = ops.soft_ce_fwd(s, tprob, 1.0 / temp)
        ctx.save_for_backward(s, tprob, lse, tsum)
        ctx.temp = temp
        return loss

    @staticmethod
    def backward(ctx, g):
        s, tprob, lse, tsum = ctx.saved_tensors
        return ops.soft_ce_bwd(s, tprob, 1.0 / ctx.temp, g, lse, tsum), None, None


def loss_rows_ok(t):
    """The prototype-axis row kernels take this tensor (GPU, f32 / bf16, rows of K % 8 == 0 elements)."""
    return t.is_cuda and t.dtype in (torch.float32, torch.bfloat16) and t.shape[-1] % 8 == 0 and not torch.compiler.is_compiling()


class DenseLayerNormFn(torch.autograd.Function):
    """nn.LayerNorm over the last dim of an f32 residual stream, result in the compute dtype.  Returns (y, x): the
    second output is the stream itself, to be used for the residual connection, so that the residual cotangent
    comes back through this node and is added inside the backward kernel instead of by a separate pass."""

    @staticmethod
    def forward(ctx, x, w, b, eps, out_dtype):
        x = _c(x)
        w32 = None if w is None else _c(w.detach().float())
        b32 = None if b is None else _c(b.detach().float())
        y, stats = ops.dense_layernorm_fwd(x, w32, b32, eps, out_dtype)
        ctx.save_for_backward(x, stats, w32)
        ctx.has_w, ctx.has_b = w is not None, b is not None
        ctx.set_materialize_grads(False)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, gy, gres):
        x, stats, w32 = ctx.saved_tensors
        if gy is None:
            return gres, None, None, None, None
        want = ctx.has_w and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        dres = None if gres is None else _c(gres.float())
        dx, dw, db = ops.dense_layernorm_bwd(_c(gy), x, w32, stats, dres, want_param_grads=want)
        return dx, (dw if ctx.has_w else None), (db if ctx.has_b else None), None, None


class RowsTo:
    """Where the tail of a branch that saw only some rows of the residual stream puts its result: rows `rowmap[r]` of
    `stream` (int32 map, one entry per compact row), in place.  `link` carries the stream's cotangent from the tail's backward
    to the LayerNorm's (DenseLayerNormRowsFn), which edits it in place - DINOv2's batch-subset stochastic depth
    (dinov2/layers/block.py:113-140) without gather / scatter passes."""
    __slots__ = ("stream", "rowmap", "link")

    def __init__(self, stream, rowmap, link):
        self.stream, self.rowmap, self.link = stream, rowmap, link


class _Link:
    __slots__ = ("g",)

    def __init__(self):
        self.g = None


class DenseLayerNormRowsFn(torch.autograd.Function):
    """LayerNorm of rows rowmap[r] of the f32 stream -> (y compute dtype, xa f32): both compact [1, rows, d]; xa = the rows as
    read (the residual input of the branch's tail, and what the backward needs after the stream was edited in place).
    Backward: the stream's cotangent g arrives through `link` (set by the tail's backward, which returns no gradient for xa);
    g[rowmap[r]] = LN'(gy[r]) + g[rowmap[r]] in place, and g is returned as the gradient of the stream."""

    @staticmethod
    def forward(ctx, stream, rowmap, w, b, eps, out_dtype, link):
        if stream.dtype != torch.float32 or not stream.is_contiguous():
            raise ValueError("DenseLayerNormRowsFn: the stream must be a contiguous f32 tensor (it is edited in place)")
        w32 = None if w is None else _c(w.detach().float())
        b32 = None if b is None else _c(b.detach().float())
        y, stats, xa = ops.dense_layernorm_fwd_rows(stream, rowmap, w32, b32, eps, out_dtype)
        ctx.save_for_backward(xa, stats, w32, rowmap)
        ctx.has_w, ctx.has_b, ctx.link = w is not None, b is not None, link
        ctx.set_materialize_grads(False)
        return y, xa

    @staticmethod
    def backward(ctx, gy, gxa):
        xa, stats, w32, rowmap = ctx.saved_tensors
        g, ctx.link.g = ctx.link.g, None
        if g is None or gxa is not None or gy is None:
            raise RuntimeError("DenseLayerNormRowsFn: expects the stream cotangent through its link (RowsTo tail) and no direct "
                               "gradient for the compact rows")
        want = ctx.has_w and (ctx.needs_input_grad[2] or ctx.needs_input_grad[3])
        dw, db = ops.dense_layernorm_bwd_rows_(_c(gy), xa, w32, stats, g, rowmap, want_param_grads=want)
        return g, None, (dw if ctx.has_w else None), (db if ctx.has_b else None), None, None, None


class DenseLinearFn(torch.autograd.Function):
    """nn.Linear on the BLAS library with cached compute-dtype weights (no per-step cast kernels in the graph)."""

    @staticmethod
    def forward(ctx, x, w, b, dtype, cache):
        ops._require_cuda(x)
        xb = _c(x if x.dtype == dtype else x.to(dtype))
        wb, bb = cache.get(w, b, dtype)
        with torch.autocast("cuda", enabled=False):
            y = torch.nn.functional.linear(xb, wb, bb)
        ctx.save_for_backward(xb, wb)
        ctx.has_b, ctx.x_dtype = b is not None, x.dtype
        return y

    @staticmethod
    def backward(ctx, gy):
        xb, wb = ctx.saved_tensors
        g2, x2 = _c(gy).reshape(-1, wb.shape[0]), xb.reshape(-1, wb.shape[1])
        with torch.autocast("cuda", enabled=False):
            gx = (g2 @ wb).view(xb.shape).to(ctx.x_dtype) if ctx.needs_input_grad[0] else None
            gw = (g2.t() @ x2).float()
            gb = g2.sum(0, dtype=torch.float32) if ctx.has_b else None
        return gx, gw, gb, None, None


class DenseLinearGeluFn(torch.autograd.Function):
    """gelu(x W^T + b) of the standard MLP (bf16): library GEMM with bias epilogue + torch's exact GELU forward; the
    backward is one HIP pass that yields dh = gelu'(h) g and the bias gradient (column sums of dh) together."""

    @staticmethod
    def forward(ctx, x, w, b, cache):
        ops._require_cuda(x)
        dtype = torch.bfloat16
        xb = _c(x if x.dtype == dtype else x.to(dtype))
        wb, bb = cache.get(w, b, dtype)
        with torch.autocast("cuda", enabled=False):
            h = torch.nn.functional.linear(xb, wb, bb)
            y = torch.nn.functional.gelu(h)
        ctx.save_for_backward(xb, wb, h)
        ctx.has_b, ctx.x_dtype = b is not None, x.dtype
        return y

    @staticmethod
    def backward(ctx, gy):
        xb, wb, h = ctx.saved_tensors
        dh, db = ops.dense_gelu_bwd(h, _c(gy.to(h.dtype)), want_colsum=ctx.has_b)
        g2, x2 = dh.reshape(-1, wb.shape[0]), xb.reshape(-1, wb.shape[1])
        with torch.autocast("cuda", enabled=False):
            gx = (g2 @ wb).view(xb.shape).to(ctx.x_dtype) if ctx.needs_input_grad[0] else None
            gw = (g2.t() @ x2).float()
        return gx, gw, db, None


class LinearScaleResidualFn(torch.autograd.Function):
    """Tail of a standard block branch:  out = x + rs * gamma * (a W^T + b)  with x the f32 residual stream, a the
    compute-dtype branch activations, gamma the layer scale [d] and rs the per-sample stochastic-depth factor
    (both optional).  The GEMMs are the library's; everything around them is one HIP pass each way, which also
    yields d gamma and the bias gradient."""

    @staticmethod
    def forward(ctx, x, a, w, b, gamma, rs, rps, dtype, cache):
        ops._require_cuda(x)
        x = _c(x)
        ab = _c(a if a.dtype == dtype else a.to(dtype))
        wb, bb = cache.get(w, b, dtype)
        y = _linear_lib(ab.reshape(-1, wb.shape[1]), wb, bb).view(*ab.shape[:-1], wb.shape[0])
        g32 = None if gamma is None else _c(gamma.detach().float())
        rs32 = None if rs is None else _c(rs.detach().float())
        out = ops.scale_residual_fwd(x, y, g32, rs32, rps)
        ctx.save_for_backward(ab, wb, y, g32, rs32)
        ctx.meta = (rps, b is not None, gamma is not None, a.dtype)
        ctx.wparam = w
        return out

    @staticmethod
    def backward(ctx, gout):
        ab, wb, y, g32, rs32 = ctx.saved_tensors
        rps, has_b, has_gamma, a_dtype = ctx.meta
        gout = _c(gout.float())
        gy, dgamma, colsum = ops.scale_residual_bwd(gout, y, g32, rs32, rps, want_gamma=has_gamma, want_colsum=has_b)
        g2, a2 = gy.reshape(-1, wb.shape[0]), ab.reshape(-1, wb.shape[1])
        ga = _mm_lib(g2, wb).view(ab.shape).to(a_dtype) if ctx.needs_input_grad[1] else None
        gw = _wgrad_lib(g2, a2, ctx.wparam)
        return gout, ga, gw, colsum, dgamma, None, None, None, None


def invalidate_weight_caches(model):
    """Drop every cached compute-dtype weight copy below ``model`` (LinearD8 ``WeightPrep``s, the standard half's
    ``DenseWeightCache``s): the next forward re-casts from the master parameters.

    The caches key on ``(data_ptr, tensor._version)``.  An optimizer that writes through ``p.data`` or raw pointers
    (apex FusedLAMB — the reference recipe's optimizer — updates ``p.data`` in a multi-tensor kernel) does not bump
    ``_version``, so such optimizers MUST call this after every step; ``track_optimizer`` installs it as a
    step post-hook.  The repo's own ``FusedLamb`` advances the version counters itself."""
    n = 0
    for m in model.modules():
        for attr in ("_prep", "_c1", "_c2"):
            c = getattr(m, attr, None)
            if isinstance(c, (WeightPrep, DenseWeightCache)):
                c.key = None
                if isinstance(c, DenseWeightCache):
                    # the transposed copy (operand of the hand-written input-gradient GEMM) keys on `key`; a re-cast
                    # restores the same key (data_ptr and _version are unchanged by a raw write), so drop it explicitly
                    c.wt, c.wt_key = None, None
                n += 1
    return n


def track_optimizer(model, optimizer):
    """Register ``invalidate_weight_caches(model)`` as a post-step hook of a torch.optim.Optimizer (any optimizer,
    including ones that update ``p.data`` behind autograd's back).  Returns the hook handle."""
    return optimizer.register_step_post_hook(lambda *_a, **_k: invalidate_weight_caches(model))


# --------------------------------------------------------------- standard half on the hand-written MFMA GEMMs
# Which GEMMs of a standard block run on csrc/dense_gemm.hip (the others would go to the BLAS library + the row kernels
# of csrc/dense.hip).  Measured in the train step on MI355X (same device, back to back; DESIGN.md section 3.5;
# tools/ab_dense.sh): since round 3 (tail tiles in front of the grid, rows past M never travel, no split where the slab
# round trip costs more than the K loop it saves, fc1's bias gradient out of the GELU' epilogue) all eight on the
# hand-written kernel is level with or ahead of every mixed routing (72.2 vs 72.3-72.9 ms per step), so the whole
# standard half is hand-written; per shape the library is still ahead on proj (N = K = 1280: 62 vs 67 us, two rounds of
# 256 x 256 tiles against its 256 x 192 ones) and that is paid back by the fused tails (no scale_residual_fwd, no
# dense_gelu_bwd pass).
WGRAD_F32_OUT = False      # True: library weight gradients as torch.mm(..., out_dtype=float32) (no bf16 rounding of
                           # dW; measured equal in step time, but outside the shipped TunableOp table)
DENSE_HIP = {"qkv", "dqkv", "proj", "dproj", "fc1", "dfc1", "fc2", "dfc2"}    # any subset (bench.py --dense-hip)
# proj / fc2 on the hand-written kernel: x + rs*gamma*y inside the GEMM's epilogue (True) or as the separate row kernel
# after a plain epilogue (False).  The fused tail reads and writes the f32 residual stream tile by tile at the end of a
# workgroup with nothing to overlap it (one workgroup per CU): in the step it costs more than the 35 us row pass.
DENSE_RESID_FUSED = False
# The residual add of a branch and the LayerNorm that opens the next one (norm2 of the same block, norm1 of the next
# block) as ONE row pass (csrc/dense.hip dense_resid_ln_fwd_kernel): the f32 stream is written once and not read back.
NEXT_NORM_FUSED = True
# Octic half: the residual-fused proj / fc2 and the LayerNormD8 that follows as one autograd node (LinearD8NormFn): the
# norm's backward also emits the drop-path-scaled bf16 cotangent of the branch (no cast_rowscale pass).
OCTIC_NEXT_NORM = True
# ... and in the backward of that pair: LayerNorm backward + the backward of the residual tail in front of it as one row
# pass (dense_ln_bwd_tail_kernel) instead of dense_ln_bwd + scale_residual_bwd (the stream cotangent is not read back).
LN_TAIL_FUSED = True


def dense_hip_ok(x, w, which=None):
    """bf16 autocast on the GPU, a shape csrc/dense_gemm.hip covers in BOTH directions (forward: K = in_features,
    input gradient: K = out_features; the kernel needs K % 64 == 0, K >= 128, N % 8 == 0) and routed."""
    rows = x.numel() // max(1, x.shape[-1])
    return ((which is None or which in DENSE_HIP) and x.is_cuda and w.shape[1] % 64 == 0 and w.shape[1] >= 128
            and w.shape[0] % 64 == 0 and w.shape[0] >= 128
            and rows * max(w.shape) * 2 < 2 ** 31)          # 32-bit buffer offsets in the kernel (it refuses larger operands)


def _f32(t):
    return None if t is None else _c(t.detach().float())


def _timed_lib(kind, fn, M, N, K):
    """Run a BLAS-library GEMM under the kernel timer (bench.py's `kernels` / `roofline` cover the whole step, not only
    the engine's own launches): name = library_gemm<kind N x K>, 2 M N K flops, operand + result bytes."""
    t = ops.KERNEL_TIMER.start()
    out = fn()
    if t is not None:
        ops.KERNEL_TIMER.stop(t, f"library_gemm<{kind} {N}x{K}>", 2 * (M * K + N * K + M * N), 2.0 * M * N * K)
    return out


# Row slabs of the library weight gradients (the fallback path: shapes csrc/dense_wgrad.hip refuses, or WGRAD_HIP off).
# dW [N,K] = g^T x reduces over the token rows into few output tiles (100 tiles of 256x256 for 5120x1280: the single GEMM
# leaves more than a third of the 256 CUs idle and runs at 0.43-0.79 PFLOP/s).  As a batched GEMM over S row slabs the
# same library kernels fill the chip; the slab partials (bf16, like the single GEMM's result) are summed in f32 by one
# small reduction that replaces the bf16 -> f32 cast.  Measured on MI355X (tools/probe_wgrad_split.py): 311 -> 263,
# 284 -> 247, 217 -> 188, 124 -> 92 us.  WGRAD_SLABS holds the measured choices; other shapes take the rule
# "one round of 256 x 256 tiles over the CUs" (wgrad_slabs), False switches slabs off (bench.py --no-wgrad-slabs).
WGRAD_SLABS = {(5120, 1280): 4, (1280, 5120): 4, (3840, 1280): 2, (1280, 1280): 2}


def wgrad_slabs(M, N, K):
    if WGRAD_SLABS is False:
        return 1
    S = WGRAD_SLABS.get((N, K))
    if S is None:
        tiles = -(-N // 256) * -(-K // 256)
        S = max(1, min(8, 256 // max(1, tiles)))
    while S > 1 and M % S:
        S -= 1
    return S


# Weight gradients of the standard half on csrc/dense_wgrad.hip (dW = dY^T X straight from the row-major operands, f32
# result, fixed-order slab reduction) wherever the kernel takes the shape (ops.dense_wgrad_ok: N, K multiples of 256, at
# most 256 output tiles - ViT-H/14, ViT-L/16, ...), not for a literal list of shapes (round-3 review).  Measured against
# the library's batched row slabs on one MI355X (tools/bench_tn.py): 5120x1280 224-234 vs 296 us, 1280x5120 211 vs 246,
# 3840x1280 156 vs 187, 1280x1280 93 vs 100.  False = every weight gradient on the BLAS library (bench.py --lib-wgrad).
WGRAD_HIP = True


def _wgrad_lib(g2, x2, wparam=None):
    """dW = g^T x (f32 result): the hand-written TN kernel wherever it takes the shape, else the BLAS library.  wparam: the
    parameter this is the gradient of - under DDP the result is written into its bucket view (ops.GRAD_DEST)."""
    M, N, K = g2.shape[0], g2.shape[1], x2.shape[1]
    if (WGRAD_HIP and g2.is_cuda and g2.dtype == torch.bfloat16 and x2.dtype == torch.bfloat16
            and g2.stride(1) == 1 and x2.stride(1) == 1 and ops.dense_wgrad_ok(M, N, K)
            and M * max(g2.stride(0), x2.stride(0)) * 2 < 2 ** 31):
        dest = ops.grad_dest(wparam, (N, K))
        dw = ops.dense_wgrad_tn(g2, x2, out=dest)
        if dest is not None:
            ops.grad_written(wparam)
        return dw
    S = wgrad_slabs(M, N, K)
    if S > 1 and g2.is_contiguous() and x2.is_contiguous():
        with torch.autocast("cuda", enabled=False):
            return _timed_lib("wgrad", lambda: torch.bmm(g2.view(S, M // S, N).transpose(1, 2), x2.view(S, M // S, K))
                              .sum(0, dtype=torch.float32), M, N, K)
    with torch.autocast("cuda", enabled=False):
        if WGRAD_F32_OUT:
            # bf16 operands, f32 result straight from the GEMM: no bf16 rounding of the gradient, no cast launch
            return _timed_lib("wgrad", lambda: torch.mm(g2.t(), x2, out_dtype=torch.float32), g2.shape[0], g2.shape[1],
                              x2.shape[1])
        w = _timed_lib("wgrad", lambda: g2.t() @ x2, g2.shape[0], g2.shape[1], x2.shape[1])
        return w.float()


def _mm_lib(a2, b):
    """a2 [M,N] @ b [N,K]: library input-gradient GEMM of bf16 2-D tensors (no autocast re-casts)."""
    with torch.autocast("cuda", enabled=False):
        return _timed_lib("dgrad", lambda: a2 @ b, a2.shape[0], b.shape[1], b.shape[0])


def _linear_lib(x2, wb, bias):
    with torch.autocast("cuda", enabled=False):
        return _timed_lib("fwd", lambda: torch.nn.functional.linear(x2, wb, bias), x2.shape[0], wb.shape[0], wb.shape[1])


# The qkv and proj weight gradients of a standard block as ONE launch (ops.dense_wgrad_tn_pair): proj's backward runs first and
# parks its operands in the Attention module's WgradPair; qkv's backward, one attention backward later, launches both.  The
# parked gradient tensor is handed to autograd UNWRITTEN in between - only where nothing can read it before the end of the
# backward pass, i.e. under the same condition as ops.DEFERRED_FINISHES (train.Trainer: one micro-batch into .grad = None, no
# DDP hooks); elsewhere both run at once as before.  1280 x 1280 alone is 25 tiles x 8 row slabs at 0.6 PFLOP/s (90 us); beside
# the 75 tiles of 3840 x 1280 the pair is a 100-tile, two-slab launch like the MLP weights.
WGRAD_PAIRED = True
GELU_FACTOR = True     # standard MLP: keep gelu'(h) instead of h between fc1's forward and fc2's input gradient (bench --no-gelu-factor)


class WgradPair:
    __slots__ = ("pending",)

    def __init__(self):
        self.pending = None

    def park(self, g2, x2, wparam=None):
        """Postpone dW = g2^T x2; returns the (not yet written) result tensor - the registered destination of `wparam`'s
        gradient where there is one (ops.GRAD_DEST)."""
        shape = (g2.shape[1], x2.shape[1])
        dw = ops.grad_dest(wparam, shape)
        landed = wparam if dw is not None else None
        if dw is None:
            dw = torch.empty(shape, dtype=torch.float32, device=g2.device)
        # (storage, not tensor: a second tensor reference would make AccumulateGrad clone the unwritten gradient)
        self.pending = (g2, x2, dw.untyped_storage(), dw.data_ptr(), tuple(dw.shape), dw.storage_offset(), landed)
        ops.DEFERRED_FINISHES.add_pair(self)
        return dw

    def take(self):
        """(g2, x2, dw, landed): landed = the parameter whose registered gradient destination dw is, or None."""
        p, self.pending = self.pending, None
        if p is None:
            return None
        g2, x2, storage, ptr, shape, offset, landed = p
        dw = torch.empty(0, dtype=torch.float32, device=g2.device).set_(storage, offset, shape)
        assert dw.data_ptr() == ptr
        return g2, x2, dw, landed

    def flush(self):
        p = self.take()
        if p is not None:
            g2, x2, dw, landed = p
            dw.copy_(_wgrad_lib(g2, x2))
            ops.grad_written(landed)


def _pair_ready(pair, g2, x2):
    return (pair is not None and WGRAD_PAIRED and WGRAD_HIP and ops.DEFERRED_FINISHES.enabled
            and ops.DEFERRED_FINISHES.allow_pairs and ops._in_backward()
            and g2.is_cuda and g2.dtype == torch.bfloat16 and x2.dtype == torch.bfloat16 and g2.stride(1) == 1
            and x2.stride(1) == 1)


def _tok(shape):
    """Tokens per image of a [B, tokens, C] activation (0 = not known to be whole images): lets csrc/dense_gemm.hip take
    per-image row panels for ViT-H/14's 257-token images (octic_dense_gemm_nt_tokens)."""
    return int(shape[-2]) if len(shape) >= 3 else 0


class DenseLinearNTFn(torch.autograd.Function):
    """nn.Linear (bf16 operands, f32 accumulate, f32 bias added before the one rounding to bf16): forward and input
    gradient on csrc/dense_gemm.hip.  deit/vit.py:33 (``self.qkv(x)``)."""

    @staticmethod
    def forward(ctx, x, w, b, cache, tag, pair=None):
        ctx.pair = pair
        ctx.wparam = w
        xb = _c(x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16))
        wb, wt = cache.get_nt(w, b, ("d" + tag) in DENSE_HIP and ctx.needs_input_grad[0])
        x2 = xb.reshape(-1, wb.shape[1])
        if tag in DENSE_HIP:
            y = ops.dense_gemm_nt(x2, wb, 0, bias=_f32(b), name="dense_nt_kernel<plain>", tokens=_tok(x.shape))
        else:
            y = _linear_lib(x2, wb, None if b is None else cache.b)
        ctx.save_for_backward(x2, wb, wt)          # wt is None unless the input gradient is routed to the HIP kernel
        ctx.meta = (b is not None, x.dtype, x.shape, tag)
        return y.view(*x.shape[:-1], wb.shape[0])

    @staticmethod
    def backward(ctx, gy):
        x2, wb, wt = ctx.saved_tensors
        has_b, x_dtype, x_shape, tag = ctx.meta
        g2 = _c(gy).reshape(-1, wb.shape[0])
        gx = None
        if ctx.needs_input_grad[0]:
            gx = (ops.dense_gemm_nt(g2, wt, 0, name="dense_nt_kernel<dgrad>", tokens=_tok(x_shape)) if wt is not None
                  else _mm_lib(g2, wb)).view(x_shape).to(x_dtype)
        gb = None
        if has_b:
            gb = (ops.dense_colsum(g2) if g2.is_cuda and g2.dtype == torch.bfloat16 and g2.shape[1] % 8 == 0
                  else g2.sum(0, dtype=torch.float32))
        pair = ctx.pair
        if pair is not None and pair.pending is not None:
            pg, px = pair.pending[0], pair.pending[1]
            if (_pair_ready(pair, g2, x2) and pg.shape[0] == g2.shape[0] and px.shape[1] == x2.shape[1]
                    and ops.dense_wgrad_pair_ok(g2.shape[0], g2.shape[1], pg.shape[1], x2.shape[1])):
                pg, px, pdw, landed = pair.take()
                dest = ops.grad_dest(ctx.wparam, (g2.shape[1], x2.shape[1]))
                dw, _ = ops.dense_wgrad_tn_pair(g2, x2, pg, px, dw1=pdw, dw0=dest)
                ops.grad_written(landed, ctx.wparam if dest is not None else None)
                return gx, dw, gb, None, None, None
            pair.flush()
        return gx, _wgrad_lib(g2, x2, ctx.wparam), gb, None, None, None


class DenseProjResidFn(torch.autograd.Function):
    """out = x + rs * gamma * (a W^T + b): the tail of a standard-block branch (deit/vit.py:131-134) as ONE GEMM with a
    fused epilogue; backward = one HIP row pass (gy, d gamma, bias gradient) + the input-gradient GEMM."""

    @staticmethod
    def forward(ctx, x, a, w, b, gamma, rs, rps, cache, nw=None, nb=None, neps=None, pair=None, rows_to=None, stream=None):
        """neps is not None: also return LayerNorm(out; nw, nb, neps) in bf16 - the norm that opens the next branch - from
        the same row pass that adds the residual (NEXT_NORM_FUSED).  rows_to (RowsTo; stream = rows_to.stream, passed as a
        tensor so that autograd sees the in-place edit): x are compact rows of the stream, the result goes back into it."""
        ctx.pair = pair
        ctx.rows_to = rows_to
        ctx.wparam = w
        x = _c(x)
        ab = _c(a if a.dtype == torch.bfloat16 else a.to(torch.bfloat16))
        wb, wt = cache.get_nt(w, b, "dproj" in DENSE_HIP and ctx.needs_input_grad[1])
        a2 = ab.reshape(-1, wb.shape[1])
        g32, rs32 = _f32(gamma), _f32(rs)
        ctx.norm = neps is not None
        ctx.x_shape = x.shape
        if ctx.norm:
            y = ops.dense_gemm_nt(a2, wb, 0, bias=_f32(b), name="dense_nt_kernel<proj>", tokens=_tok(a.shape))
            nw32, nb32 = _f32(nw), _f32(nb)
            out, yn, stats = ops.dense_resid_layernorm_fwd(x.view(-1, wb.shape[0]), y, g32, rs32, rps, nw32, nb32, neps,
                                                           torch.bfloat16)
            ctx.save_for_backward(a2, wb, wt, y, g32, rs32, out, stats, nw32)
            ctx.meta = (rps, b is not None, gamma is not None, a.dtype, a.shape, nw is not None, nb is not None)
            ctx.set_materialize_grads(False)
            return out.view(x.shape), yn.view(x.shape)
        if rows_to is not None:
            y = ops.dense_gemm_nt(a2, wb, 0, bias=_f32(b), name="dense_nt_kernel<proj>", tokens=_tok(a.shape))
            ops.scale_residual_fwd_rows_(stream, rows_to.rowmap, x.view(-1, wb.shape[0]), y, g32, rs32, rps)
            ctx.save_for_backward(a2, wb, wt, y, g32, rs32, rows_to.rowmap)
            ctx.meta = (rps, b is not None, gamma is not None, a.dtype, a.shape)
            ctx.mark_dirty(stream)
            return stream
        if DENSE_RESID_FUSED:
            y, out = ops.dense_gemm_nt(a2, wb, 2, bias=_f32(b), gamma=g32, rs=rs32, rps=rps, x=x.view(-1, wb.shape[0]),
                                       name="dense_nt_kernel<resid>")
        else:
            y = ops.dense_gemm_nt(a2, wb, 0, bias=_f32(b), name="dense_nt_kernel<proj>", tokens=_tok(a.shape))
            out = ops.scale_residual_fwd(x.view(-1, wb.shape[0]), y, g32, rs32, rps)
        ctx.save_for_backward(a2, wb, wt, y, g32, rs32)
        ctx.meta = (rps, b is not None, gamma is not None, a.dtype, a.shape)
        return out.view(x.shape)

    @staticmethod
    def backward(ctx, gout, gyn=None):
        dnw = dnb = None
        tail_done = False
        if ctx.norm:
            a2, wb, wt, y, g32, rs32, out, stats, nw32 = ctx.saved_tensors
            rps, has_b, has_gamma, a_dtype, a_shape, has_nw, has_nb = ctx.meta
            if gyn is not None:      # LayerNorm backward with the residual cotangent added in the same pass
                want = has_nw and (ctx.needs_input_grad[8] or ctx.needs_input_grad[9])
                dres = None if gout is None else _c(gout.float()).view(out.shape)
                g2 = _c(gyn).view(out.shape)
                if LN_TAIL_FUSED and ops.dense_ln_bwd_tail_ok(g2, y, out.shape[-1]):
                    gout, dnw, dnb, gy, dgamma, colsum = ops.dense_layernorm_bwd_tail(
                        g2, out, nw32, stats, dres, y, g32, rs32, rps, want_param_grads=want, want_gamma=has_gamma,
                        want_colsum=has_b)
                    tail_done = True
                else:
                    gout, dnw, dnb = ops.dense_layernorm_bwd(g2, out, nw32, stats, dres, want_param_grads=want)
                dnw, dnb = (dnw if has_nw else None), (dnb if has_nb else None)
        elif ctx.rows_to is not None:
            a2, wb, wt, y, g32, rs32, rowmap = ctx.saved_tensors
            rps, has_b, has_gamma, a_dtype, a_shape = ctx.meta
            gout = _c(gout.float())
            ctx.rows_to.link.g = gout             # the LayerNorm of this branch edits it in place and hands it on
            gy, dgamma, colsum = ops.scale_residual_bwd(gout, y, g32, rs32, rps, want_gamma=has_gamma, want_colsum=has_b,
                                                        rowmap=rowmap)
            tail_done = True
        else:
            a2, wb, wt, y, g32, rs32 = ctx.saved_tensors
            rps, has_b, has_gamma, a_dtype, a_shape = ctx.meta
        if not tail_done:
            gout = _c(gout.float())
            gy, dgamma, colsum = ops.scale_residual_bwd(gout.view(y.shape), y, g32, rs32, rps, want_gamma=has_gamma,
                                                        want_colsum=has_b)
        ga = None
        if ctx.needs_input_grad[1]:
            ga = (ops.dense_gemm_nt(gy, wt, 0, name="dense_nt_kernel<dgrad>", tokens=_tok(a_shape)) if wt is not None
                  else _mm_lib(gy, wb)).view(a_shape).to(a_dtype)
        if _pair_ready(ctx.pair, gy, a2) and ops.dense_wgrad_ok(gy.shape[0], gy.shape[1], a2.shape[1]):
            dw = ctx.pair.park(gy, a2, ctx.wparam)   # written by the qkv weight gradient's launch (or at the end of the pass)
        else:
            dw = _wgrad_lib(gy, a2, ctx.wparam)
        gx = None if ctx.rows_to is not None else gout.view(ctx.x_shape)
        return gx, ga, dw, colsum, dgamma, None, None, None, dnw, dnb, None, None, None, None


class DenseMlpFn(torch.autograd.Function):
    """x + rs * gamma * fc2(gelu(fc1(y))) of a standard block (timm Mlp inside deit/vit.py:131-134).  Per GEMM, routed by
    DENSE_HIP: fc1 with bias + exact-erf GELU in the epilogue (pre-activation kept for the backward), fc2 with bias, layer
    scale, stochastic depth and the f32 residual in the epilogue, fc2's input gradient with GELU' in its epilogue, fc1's
    input gradient; otherwise the BLAS library with the row kernels of csrc/dense.hip around it.  Weight gradients:
    csrc/dense_wgrad.hip (WGRAD_HIP)."""

    @staticmethod
    def forward(ctx, y, x, w1, b1, w2, b2, gamma, rs, rps, c1, c2, nw=None, nb=None, neps=None, rows_to=None, stream=None):
        """neps is not None: also return LayerNorm(out; nw, nb, neps) in bf16 (the NEXT block's norm1) from the row pass
        that adds the residual.  rows_to / stream: as in DenseProjResidFn."""
        ctx.rows_to = rows_to
        ctx.wparams = (w1, w2)
        x = _c(x)
        yb = _c(y if y.dtype == torch.bfloat16 else y.to(torch.bfloat16))
        need_t = any(ctx.needs_input_grad[:2])            # (an inference pass - the DINOv2 teacher - never transposes)
        w1b, w1t = c1.get_nt(w1, b1, "dfc1" in DENSE_HIP and need_t)
        w2b, w2t = c2.get_nt(w2, b2, "dfc2" in DENSE_HIP and need_t)
        y2 = yb.reshape(-1, w1b.shape[1])
        g32, rs32 = _f32(gamma), _f32(rs)
        # GELU_FACTOR: fc1's epilogue leaves gelu'(h) (bf16) instead of h; fc2's input gradient multiplies by it (modes 4 / 5 of
        # octic_dense_gemm_nt) - the erf of the backward epilogue is computed once, in the forward, beside gelu's
        factor = GELU_FACTOR and "fc1" in DENSE_HIP and w2t is not None
        ctx.factor = factor
        if "fc1" in DENSE_HIP and not any(ctx.needs_input_grad):
            # no backward will come (inference, the DINOv2 teacher): gelu(h) only, nothing kept (mode 6)
            a = ops.dense_gemm_nt(y2, w1b, 6, bias=_f32(b1), name="dense_nt_kernel<gelu-only>", tokens=_tok(y.shape))
            h = a
        elif "fc1" in DENSE_HIP:
            h, a = ops.dense_gemm_nt(y2, w1b, 4 if factor else 1, bias=_f32(b1), name="dense_nt_kernel<gelu>",
                                     tokens=_tok(y.shape))
        else:
            h = _linear_lib(y2, w1b, None if b1 is None else c1.b)
            a = torch.nn.functional.gelu(h)
        ctx.norm = neps is not None
        ctx.x_shape = x.shape
        if ctx.norm:
            br = (ops.dense_gemm_nt(a, w2b, 0, bias=_f32(b2), name="dense_nt_kernel<fc2>", tokens=_tok(y.shape)) if "fc2" in DENSE_HIP
                  else _linear_lib(a, w2b, None if b2 is None else c2.b))
            nw32, nb32 = _f32(nw), _f32(nb)
            out, yn, stats = ops.dense_resid_layernorm_fwd(x.view(-1, w2b.shape[0]), br, g32, rs32, rps, nw32, nb32, neps,
                                                           torch.bfloat16)
            ctx.save_for_backward(y2, h, a, br, w1b, w1t, w2b, w2t, g32, rs32, out, stats, nw32)
            ctx.meta = (rps, b1 is not None, b2 is not None, gamma is not None, y.dtype, y.shape, nw is not None, nb is not None)
            ctx.set_materialize_grads(False)
            return out.view(x.shape), yn.view(x.shape)
        if rows_to is not None:
            br = (ops.dense_gemm_nt(a, w2b, 0, bias=_f32(b2), name="dense_nt_kernel<fc2>", tokens=_tok(y.shape)) if "fc2" in DENSE_HIP
                  else _linear_lib(a, w2b, None if b2 is None else c2.b))
            ops.scale_residual_fwd_rows_(stream, rows_to.rowmap, x.view(-1, w2b.shape[0]), br, g32, rs32, rps)
            ctx.save_for_backward(y2, h, a, br, w1b, w1t, w2b, w2t, g32, rs32, rows_to.rowmap)
            ctx.meta = (rps, b1 is not None, b2 is not None, gamma is not None, y.dtype, y.shape)
            ctx.mark_dirty(stream)
            return stream
        if "fc2" in DENSE_HIP and DENSE_RESID_FUSED:
            br, out = ops.dense_gemm_nt(a, w2b, 2, bias=_f32(b2), gamma=g32, rs=rs32, rps=rps, x=x.view(-1, w2b.shape[0]),
                                        name="dense_nt_kernel<resid>")
        elif "fc2" in DENSE_HIP:
            br = ops.dense_gemm_nt(a, w2b, 0, bias=_f32(b2), name="dense_nt_kernel<fc2>", tokens=_tok(y.shape))
            out = ops.scale_residual_fwd(x.view(-1, w2b.shape[0]), br, g32, rs32, rps)
        else:
            br = _linear_lib(a, w2b, None if b2 is None else c2.b)
            out = ops.scale_residual_fwd(x.view(-1, w2b.shape[0]), br, g32, rs32, rps)
        ctx.save_for_backward(y2, h, a, br, w1b, w1t, w2b, w2t, g32, rs32)
        ctx.meta = (rps, b1 is not None, b2 is not None, gamma is not None, y.dtype, y.shape)
        return out.view(x.shape)

    @staticmethod
    def backward(ctx, gout, gyn=None):
        dnw = dnb = None
        tail_done = False
        if ctx.norm:
            y2, h, a, br, w1b, w1t, w2b, w2t, g32, rs32, out, stats, nw32 = ctx.saved_tensors
            rps, has_b1, has_b2, has_gamma, y_dtype, y_shape, has_nw, has_nb = ctx.meta
            if gyn is not None:      # LayerNorm backward with the residual cotangent added in the same pass
                want = has_nw and (ctx.needs_input_grad[11] or ctx.needs_input_grad[12])
                dres = None if gout is None else _c(gout.float()).view(out.shape)
                g2 = _c(gyn).view(out.shape)
                if LN_TAIL_FUSED and ops.dense_ln_bwd_tail_ok(g2, br, out.shape[-1]):
                    gout, dnw, dnb, gbr, dgamma, db2 = ops.dense_layernorm_bwd_tail(
                        g2, out, nw32, stats, dres, br, g32, rs32, rps, want_param_grads=want, want_gamma=has_gamma,
                        want_colsum=has_b2)
                    tail_done = True
                else:
                    gout, dnw, dnb = ops.dense_layernorm_bwd(g2, out, nw32, stats, dres, want_param_grads=want)
                dnw, dnb = (dnw if has_nw else None), (dnb if has_nb else None)
        elif ctx.rows_to is not None:
            y2, h, a, br, w1b, w1t, w2b, w2t, g32, rs32, rowmap = ctx.saved_tensors
            rps, has_b1, has_b2, has_gamma, y_dtype, y_shape = ctx.meta
            gout = _c(gout.float())
            ctx.rows_to.link.g = gout
            gbr, dgamma, db2 = ops.scale_residual_bwd(gout, br, g32, rs32, rps, want_gamma=has_gamma, want_colsum=has_b2,
                                                      rowmap=rowmap)
            tail_done = True
        else:
            y2, h, a, br, w1b, w1t, w2b, w2t, g32, rs32 = ctx.saved_tensors
            rps, has_b1, has_b2, has_gamma, y_dtype, y_shape = ctx.meta
        if not tail_done:
            gout = _c(gout.float())
            gbr, dgamma, db2 = ops.scale_residual_bwd(gout.view(br.shape), br, g32, rs32, rps, want_gamma=has_gamma,
                                                      want_colsum=has_b2)
        if w2t is not None:
            md = 5 if ctx.factor else 3
            if has_b1:                                                                  # gelu'(h) * (gbr W2), + db1
                dh, db1 = ops.dense_gemm_nt(gbr, w2t, md, h=h, name="dense_nt_kernel<dgelu>", want_colsum=True,
                                            tokens=_tok(y_shape))
            else:
                dh, db1 = ops.dense_gemm_nt(gbr, w2t, md, h=h, name="dense_nt_kernel<dgelu>", tokens=_tok(y_shape)), None
        else:
            dh, db1 = ops.dense_gelu_bwd(h, _mm_lib(gbr, w2b), want_colsum=has_b1)
        gw2 = _wgrad_lib(gbr, a, ctx.wparams[1])
        gw1 = _wgrad_lib(dh, y2, ctx.wparams[0])
        gy = None
        if ctx.needs_input_grad[0]:
            gy = (ops.dense_gemm_nt(dh, w1t, 0, name="dense_nt_kernel<dgrad>", tokens=_tok(y_shape)) if w1t is not None
                  else _mm_lib(dh, w1b)).view(y_shape).to(y_dtype)
        gx = None if ctx.rows_to is not None else gout.view(ctx.x_shape)
        return gy, gx, gw1, db1, gw2, db2, dgamma, None, None, None, None, dnw, dnb, None, None, None


# -------------------------------------------------------------------------------------- hand-off
class HandoffCatFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, c, out_dtype):
        ctx.c, ctx.in_dtype = c, x.dtype
        return ops.handoff_cat_fwd(_c(x.float()), c, out_dtype)

    @staticmethod
    def backward(ctx, g):
        return ops.handoff_cat_bwd(g, ctx.c).to(ctx.in_dtype), None, None


class PowerSpectrumFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, c, out_dtype):
        x = _c(x.float())
        ctx.save_for_backward(x)
        ctx.c = c
        return ops.power_spectrum_fwd(x, c, out_dtype)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return ops.power_spectrum_bwd(g, x, ctx.c), None, None


# ------------------------------------------------------------------------------------------ lift
class LiftFn(torch.autograd.Function):
    """tokens[b, tok0+n, :] = patches(img)[b,n,:] @ Wfull^T + bias_full + pos[n,:] ; tokens[b, :tok0] = cls."""

    @staticmethod
    def forward(ctx, img, wfull, bias_full, pos, cls_row, p, dtype):
        B, Cin, Hh, Ww = img.shape
        D, K = wfull.shape
        Kpad = (K + 7) // 8 * 8
        n_p = (Hh // p) * (Ww // p)
        tok0 = 0 if cls_row is None else 1
        patches = ops.im2col(img, p, Kpad, dtype)
        wpad = torch.zeros((D, Kpad), dtype=dtype, device=img.device)
        wpad[:, :K] = wfull.detach()
        out = torch.empty((B, tok0 + n_p, D), dtype=torch.float32, device=img.device)
        ops.lift_gemm(patches, wpad, None if bias_full is None else _c(bias_full.detach().float()),
                      None if pos is None else _c(pos.detach().float()), out, B, n_p, tok0, Kpad, D)
        if tok0:
            out[:, 0] = cls_row.detach().float()
        ctx.save_for_backward(patches)
        ctx.meta = (B, n_p, tok0, K, Kpad, D, dtype, bias_full is not None, pos is not None)
        return out

    @staticmethod
    def backward(ctx, dout):
        (patches,) = ctx.saved_tensors
        B, n_p, tok0, K, Kpad, D, dtype, has_bias, has_pos = ctx.meta
        dtok = dout[:, tok0:]
        dcompact = _c(dtok.to(dtype)).reshape(B * n_p, D)
        dw = ops.lift_wgrad(patches, dcompact, Kpad, D)[:, :K]
        dpos = dtok.sum(0) if (has_pos or has_bias) else None
        dbias = dpos.sum(0) if has_bias else None
        dcls = dout[:, 0].sum(0) if tok0 else None
        return None, dw, dbias, dpos if has_pos else None, dcls, None, None
