"""HIP-backed octic layers with the reference's names, constructor signatures and state_dict keys
(reference: octic_vits/d8_layers.py; per-class citations below).  Drop-in at module level: every
module takes and returns the reference's 5-tuple.  Outputs are ``Octic`` tuples — the same five
tensors, as views of one packed token-row buffer, which is what lets the next module skip repacking.

What runs where: all octic arithmetic (LayerNormD8, the five-irrep linears incl. their gradients,
D8-GELU, head packing, layer-scale + drop-path + residual) is HIP (``functional.py``).  torch supplies
memory, RNG for masks, and the attention core (``F.scaled_dot_product_attention``, as the reference).
"""
import collections.abc
import math
from itertools import repeat

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as OF
from .d8_utils import SQRT2, SQRT2_OVER_4, convert_5tuple_to_8tuple, convert_8tuple_to_5tuple, expand_lift_kernel, packed_lift_weight
from .functional import Octic, as_packed, compute_dtype


def _ntuple(n):
    def parse(x):
        if isinstance(x, collections.abc.Iterable) and not isinstance(x, str):
            return tuple(x)
        return tuple(repeat(x, n))
    return parse


to_2tuple = _ntuple(2)

# test hook: a callable (B, keep_prob, device) -> mask[B]; default draws with torch's RNG on `device`
drop_path_mask_source = None


# Inside a model forward (OcticVisionTransformer.forward_features arms the pool) masks are drawn DROP_PATH_POOL at a time:
# one bernoulli_ + one scale launch per pool instead of two launches per mask (a ViT-H step uses 64 masks = 128 launches
# of ~4.5 us).  Same distribution as one draw per call; the order in which torch's generator is consumed differs from
# the reference's per-call draws (the parity tests inject masks through drop_path_mask_source).  The pool is re-drawn at
# the start of every forward, so a seeded forward is reproducible and a captured step contains its own draw.  Blocks
# called on their own (pool not armed) draw per call as before.
# Activation checkpointing: a recompute runs outside the armed window and must see the SAME masks as the forward.  The
# pooled draw is therefore used only when nothing can replay the block later - gradients enabled (a re-entrant
# checkpoint runs its forward under no_grad) and no saved-tensor hooks installed (the non-re-entrant checkpoint and the
# offloading contexts install them).  Otherwise masks are drawn per call from torch's generator, whose state
# torch.utils.checkpoint saves and restores, so forward and recompute agree exactly as they do in the reference.
DROP_PATH_POOL = 64
_mask_pool = {}
_pool_armed = False


def arm_drop_path_pool(on=True):
    """on: start of a model forward (the next mask triggers a fresh pooled draw); off: end of it."""
    global _pool_armed
    _pool_armed = on
    if on:
        for pool in _mask_pool.values():
            pool[1] = DROP_PATH_POOL
        for pool in _compact_pool.values():
            pool[2] = DROP_PATH_POOL


def _pool_usable():
    if not _pool_armed or not torch.is_grad_enabled():
        return False
    try:
        return torch._C._autograd._top_saved_tensors_default_hooks(False) is None
    except AttributeError:      # private API moved: be safe, draw per call
        return False


def _drop_path_mask(B, drop_prob, device, scale_by_keep=True):
    keep = 1.0 - drop_prob
    if torch.compiler.is_compiling():                # traced (torch.compile): an in-graph draw, no Python-side pool
        m = torch.empty(B, device=device, dtype=torch.float32).bernoulli_(keep)
        return m / keep if (keep > 0.0 and scale_by_keep) else m
    if drop_path_mask_source is not None or not _pool_usable():
        if drop_path_mask_source is not None:
            m = drop_path_mask_source(B, keep, device)
        else:
            m = torch.empty(B, device=device, dtype=torch.float32).bernoulli_(keep)
        if keep > 0.0 and scale_by_keep:
            m = m / keep
        return m
    key = (str(device), int(B), float(keep), bool(scale_by_keep))
    pool = _mask_pool.get(key)
    if pool is None or pool[1] >= DROP_PATH_POOL:
        t = torch.empty(DROP_PATH_POOL, B, device=device, dtype=torch.float32).bernoulli_(keep)
        if keep > 0.0 and scale_by_keep:
            t = t / keep
        pool = _mask_pool[key] = [t, 0]
    m = pool[0][pool[1]]
    pool[1] += 1
    return m


# ---- stochastic depth as batch compaction (opt-in) ---------------------------------------------------------------------------
# The reference draws one Bernoulli(keep) per sample and branch, computes the branch for EVERY sample and multiplies the
# dropped ones by zero (drop_path_d8, d8_layers.py:249-270; timm DropPath in deit/vit.py) - with the DeiT-III recipe's constant
# drop_path = 0.5 (model.py:114: `dpr = [drop_path_rate for i in range(depth)]`) half of every branch's GEMMs, attention
# and row passes, forward and backward, is work whose result is discarded and whose gradients are exactly zero.
# COMPACT_DROP_PATH = True computes a branch on the kept samples only:  x[idx] += scale * cs * f(norm(x[idx])),  idx = the samples
# the SAME per-sample Bernoulli mask keeps (drawn on the host's generator, because the launch shapes depend on the count).
# Same loss, same gradients (a dropped sample's contribution to every parameter gradient is an exact zero in the reference
# too); what changes is which rows the kernels are launched on.  Shapes vary from step to step, so this mode runs eagerly
# (no hipGraph of the whole step).  DINOv2's own block does the same with a fixed-size random subset
# (dinov2/layers/block.py:113-140, `drop_add_residual_stochastic_depth`).
COMPACT_DROP_PATH = False
compact_mask_source = None          # test hook: (B, keep_prob) -> CPU float mask [B] of zeros and ones
_compact_pool = {}
_scale_cache = {}


def _compact_indices(B, drop_prob, device):
    """Next branch's kept sample indices (device int64 [n]) and n, from a per-forward pool of host-drawn masks."""
    keep = 1.0 - drop_prob
    if compact_mask_source is not None:
        idx = compact_mask_source(B, keep).flatten().nonzero().flatten()
        return idx.to(device), int(idx.numel())
    key = (str(device), int(B), float(keep))
    pool = _compact_pool.get(key)
    if pool is None or pool[2] >= DROP_PATH_POOL or not _pool_armed:
        m = torch.empty(DROP_PATH_POOL, B).bernoulli_(keep)
        counts = m.sum(1).to(torch.int64).tolist()
        rows = m.nonzero()[:, 1].contiguous()                   # kept sample ids, mask after mask
        host = rows.pin_memory() if device.type == "cuda" else rows
        dev = host.to(device, non_blocking=True)
        offs = [0]
        for cnt in counts:
            offs.append(offs[-1] + cnt)
        pool = _compact_pool[key] = [dev, offs, 0, host]
    i = pool[2]
    pool[2] += 1
    return pool[0][pool[1][i]:pool[1][i + 1]], pool[1][i + 1] - pool[1][i]


def _compact_plan(B, dp, device):
    """(kept sample indices, their count, row scale) of the next compacted branch.  A mask that keeps nobody (probability
    2^-B) still runs the branch - on sample 0 with scale 0 - so that its parameters receive their exact-zero gradients."""
    idx, n = _compact_indices(B, dp.drop_prob, device)
    scale = 1.0 / (1.0 - dp.drop_prob) if (dp.scale_by_keep and dp.drop_prob < 1.0) else 1.0
    if n == 0:
        return torch.zeros(1, dtype=torch.int64, device=device), 1, 0.0
    return idx, n, scale


def _const_scale(n, value, device):
    key = (int(n), float(value), str(device))
    t = _scale_cache.get(key)
    if t is None:
        t = _scale_cache[key] = torch.full((n,), value, dtype=torch.float32, device=device)
    return t


USE_SAMPLE_BLOCKS = True    # developer A/B: csrc sample_blocks_kernel (True) or ATen index_select / index_copy_ (False)


def _gather(full, idx):
    if full.is_cuda and USE_SAMPLE_BLOCKS:
        from . import ops
        if full.is_contiguous() and (full[0].numel() * full.element_size()) % 16 == 0 and 0 < idx.numel() <= 65535:
            return ops.gather_samples(full, idx)
    return full.index_select(0, idx)


def _scatter_(full, idx, compact):
    if full.is_cuda and USE_SAMPLE_BLOCKS:
        from . import ops
        compact = compact if compact.is_contiguous() else compact.contiguous()
        if ops.sample_blocks_ok(full, compact, idx):
            return ops.scatter_samples_(full, idx, compact)
    return full.index_copy_(0, idx, compact)


class _RowLink:
    """Shared by the gather / scatter pair around one compacted branch (see _GatherRowsFn)."""
    __slots__ = ("g",)

    def __init__(self):
        self.g = None


class _GatherRowsFn(torch.autograd.Function):
    """xa = x[idx] (rows = samples) for a compacted branch, paired with _ScatterRowsFn through `link`.
    The pair keeps the stream's cotangent ONE tensor that flows backward through the blocks and is edited in place: the
    scatter's backward hands over the cotangent g of the stream after the branch (and returns no gradient for the stream
    itself); this backward writes the branch's input cotangent over rows idx of g - those rows of the old stream reach the
    new one only through the branch, whose own residual connection is inside g_xa - and returns g as the gradient of the old
    stream.  Composed from index_select / index_copy autograd would clone the stream forward, and clone, zero-fill,
    index_add and add full-size tensors backward: 14 ms per ViT-H step."""

    @staticmethod
    def forward(ctx, x, idx, link):
        ctx.idx, ctx.link = idx, link
        return _gather(x, idx)

    @staticmethod
    def backward(ctx, g_xa):
        g, ctx.link.g = ctx.link.g, None
        if g is None:
            raise RuntimeError("_GatherRowsFn: the stream cotangent of the paired scatter is missing")
        _scatter_(g, ctx.idx, g_xa.to(g.dtype))
        return g, None, None


class _ScatterRowsFn(torch.autograd.Function):
    """x[idx] = out (in place; out = the compacted branch's result INCLUDING its residual connection)."""

    @staticmethod
    def forward(ctx, x, idx, out, link):
        ctx.idx, ctx.link = idx, link
        _scatter_(x, idx, out)
        ctx.mark_dirty(x)
        return x

    @staticmethod
    def backward(ctx, g):
        if not g.is_contiguous():
            g = g.contiguous()
        ctx.link.g = g
        return None, None, _gather(g, ctx.idx), None


def compact_active(dp):
    """Is the compacted stochastic depth in force for this drop-path module right now?  Not where something could replay the
    block later (activation checkpointing: a recompute must see the masks of the forward - the rule of the pooled draw above):
    there the block falls back to masks on the full batch, drawn per call from torch's generator."""
    if not (COMPACT_DROP_PATH and getattr(dp, "training", False) and float(getattr(dp, "drop_prob", 0.)) > 0.
            and torch.is_grad_enabled()):
        return False
    if compact_mask_source is not None:
        return True
    try:
        return torch._C._autograd._top_saved_tensors_default_hooks(False) is None
    except AttributeError:
        return False


class DropoutD8(nn.Module):
    """d8_layers.py:84-96 — independent element masks on each of the five tensors."""

    def __init__(self, p=0.5, inplace=False):
        super().__init__()
        self.dropout = nn.Dropout(p=p, inplace=inplace)

    @property
    def active(self):
        return self.training and self.dropout.p > 0.0

    def forward(self, xs):
        if not self.active:
            return xs
        return tuple(self.dropout(x) for x in xs[:5])


class TritonGeluD8(nn.Module):
    """d8_gelu.py:480-482 (name kept for drop-in; the kernel is HIP, not Triton)."""

    def forward(self, xs):
        # through the dispatcher (custom_ops.py: torch.ops.octic.gelu_d8*, schema + fake kernel + autograd formula), so
        # torch.compile traces through this op instead of breaking the graph at a ctypes call
        from . import custom_ops as _co          # noqa: F401  (registers the ops)
        if isinstance(xs, Octic):
            return Octic(torch.ops.octic.gelu_d8(xs.packed, xs.c), xs.c)
        return tuple(torch.ops.octic.gelu_d8_tuple(xs[0], xs[1], xs[2], xs[3], xs[4]))


class GeluD8(nn.Module):
    """d8_layers.py:98-102 — 8-tuple in / 8-tuple out."""

    def forward(self, xs):
        return convert_5tuple_to_8tuple(TritonGeluD8()(convert_8tuple_to_5tuple(xs)))


_IRREPS = ("A1", "A2", "B1", "B2", "E")


class LinearD8(nn.Module):
    """d8_layers.py:104-130.  One irrep-blocked MFMA launch instead of five nn.Linear calls; the
    nn.Linear submodules only hold the parameters (state_dict keys lin_{A1,A2,B1,B2,E}.weight, lin_A1.bias).
    Optional ``resid/rs/cs`` fuse ``resid + drop_path(gamma * y)`` into the GEMM epilogue."""

    def __init__(self, input_channels, output_channels, bias=True):
        super().__init__()
        if input_channels % 8 != 0 or output_channels % 8 != 0:
            raise ValueError()
        self.bias = bias
        self.input_channels, self.output_channels = input_channels, output_channels
        ci, co = input_channels // 8, output_channels // 8
        self.lin_A1 = nn.Linear(ci, co, bias=bias)
        self.lin_A2 = nn.Linear(ci, co, bias=False)
        self.lin_B1 = nn.Linear(ci, co, bias=False)
        self.lin_B2 = nn.Linear(ci, co, bias=False)
        self.lin_E = nn.Linear(2 * ci, 2 * co, bias=False)
        self._prep = OF.WeightPrep()

    def weights(self):
        return tuple(getattr(self, "lin_" + n).weight for n in _IRREPS)

    def forward(self, x_batched, resid=None, rs=None, cs=None, next_norm=None):
        """next_norm (a LayerNormD8, only with resid): also return next_norm(result) -> (Octic, Octic or None); on the
        bf16 GPU path the two are ONE autograd node whose backward skips the cast pass (OF.LinearD8NormFn)."""
        assert len(x_batched) == 5, "Input should be a 5-tuple"
        xp, cin = as_packed(x_batched)
        if 8 * cin != self.input_channels:
            raise ValueError(f"LinearD8: expected {self.input_channels} channels, got {8 * cin}")
        cout = self.output_channels // 8
        dtype = compute_dtype(xp)
        rps = xp.shape[-2] if xp.dim() >= 2 else 1
        if rs is not None and rs.numel() == xp.numel() // xp.shape[-1] and rs.numel() != xp.shape[0]:
            rps = 1                                   # one factor per ROW (several crop sets in one row tensor: ragged.py)
        cs5 = (None,) * 5 if cs is None else tuple(cs)
        if torch.compiler.is_compiling():
            # being traced: the same kernels through the dispatcher (dispatch.py), no Python-side weight cache
            from . import dispatch as _D   # noqa: F401
            xq = xp if xp.dtype == dtype else xp.to(dtype)
            # (the flat prepared copies functional.PrepBatch keeps current in place, where the fused optimizer owns them)
            fl = self._prep.flat if dtype == torch.bfloat16 else None
            pwb, pwt = fl if fl is not None else (None, None)
            y = torch.ops.octic.linear_d8(xq, *self.weights(), self.lin_A1.bias, resid, rs, *cs5, cin, cout, rps, pwb, pwt)
            return Octic(y, cout) if next_norm is None else (Octic(y, cout), None)
        if (next_norm is not None and resid is not None and OF.OCTIC_NEXT_NORM and xp.is_cuda and dtype == torch.bfloat16
                and resid.dtype == torch.float32 and type(next_norm) is LayerNormD8):
            a, beta = next_norm.affine()
            y, yn = OF.LinearD8NormFn.apply(xp, *self.weights(), self.lin_A1.bias, resid, rs, *cs5, cin, cout, rps, dtype,
                                            self._prep, *a, beta, next_norm.eps)
            return Octic(y, cout), Octic(yn, cout)
        y = OF.LinearD8Fn.apply(xp, *self.weights(), self.lin_A1.bias, resid, rs, *cs5, cin, cout, rps, dtype, self._prep)
        return Octic(y, cout) if next_norm is None else (Octic(y, cout), None)

    def extra_repr(self) -> str:
        return f"in_features={self.input_channels}, out_features={self.output_channels}, bias={self.bias is not None}"


class AffineD8(nn.Module):
    """d8_layers.py:132-158."""

    def __init__(self, dim, bias=True):
        super().__init__()
        if dim % 8 != 0:
            raise ValueError()
        self.alpha_A1 = nn.Parameter(torch.ones(dim // 8))
        self.alpha_A2 = nn.Parameter(torch.ones(dim // 8))
        self.alpha_B1 = nn.Parameter(torch.ones(dim // 8))
        self.alpha_B2 = nn.Parameter(torch.ones(dim // 8))
        self.alpha_E = nn.Parameter(torch.ones(dim // 4))
        self.beta = nn.Parameter(torch.zeros(dim // 8)) if bias else None

    def alphas(self):
        return (self.alpha_A1, self.alpha_A2, self.alpha_B1, self.alpha_B2, self.alpha_E)

    def packed_scale(self):
        return torch.cat(self.alphas() + (self.alpha_E,))

    def forward(self, xs):
        # stand-alone use (inside blocks the scale is fused into the GEMM epilogue)
        xp, c = as_packed(xs)
        y = xp * self.packed_scale().to(xp.dtype)
        if self.beta is not None:
            y = torch.cat((y[..., :c] + self.beta.to(y.dtype), y[..., c:]), dim=-1)
        return Octic(y.contiguous(), c)


class LayerScaleD8(nn.Module):
    """d8_layers.py:189-212."""

    def __init__(self, dim, init_values=1e-5):
        super().__init__()
        if dim % 8 != 0:
            raise ValueError()
        self.alpha_A1 = nn.Parameter(init_values * torch.ones(dim // 8))
        self.alpha_A2 = nn.Parameter(init_values * torch.ones(dim // 8))
        self.alpha_B1 = nn.Parameter(init_values * torch.ones(dim // 8))
        self.alpha_B2 = nn.Parameter(init_values * torch.ones(dim // 8))
        self.alpha_E = nn.Parameter(init_values * torch.ones(dim // 4))

    def alphas(self):
        return (self.alpha_A1, self.alpha_A2, self.alpha_B1, self.alpha_B2, self.alpha_E)

    def forward(self, xs):
        xp, c = as_packed(xs)
        return Octic((xp * torch.cat(self.alphas() + (self.alpha_E,)).to(xp.dtype)).contiguous(), c)


class LayerNormD8(nn.Module):
    """d8_layers.py:161-186 (eps inside the root, per-segment means, shared std, AffineD8 on top)."""

    def __init__(self, channels, eps=1e-05, elementwise_affine=True, bias=True):
        super().__init__()
        self.scaling = AffineD8(channels, bias=bias) if elementwise_affine else nn.Identity()
        self.eps = eps

    def affine(self):
        if isinstance(self.scaling, AffineD8):
            return self.scaling.alphas(), self.scaling.beta
        return (None,) * 5, None

    def forward(self, xs, _out_dtype=None, _with_resid=False):
        xp, c = as_packed(xs)
        out_dtype = _out_dtype or xp.dtype
        a, beta = self.affine()
        if torch.compiler.is_compiling():
            from . import dispatch as _D   # noqa: F401
            y = torch.ops.octic.layernorm_d8(xp, *a, beta, self.eps, c, out_dtype == torch.bfloat16)[0]
            return (Octic(y, c), xp) if _with_resid else Octic(y, c)
        y, xres = OF.LayerNormD8Fn.apply(xp, *a, beta, self.eps, c, out_dtype)
        # _with_resid: also hand back the stream to take the residual from (its cotangent is then folded into this
        # node's backward kernel)
        return (Octic(y, c), xres) if _with_resid else Octic(y, c)


class MlpD8(nn.Module):
    """d8_layers.py:215-247."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=TritonGeluD8, norm_layer=None,
                 bias=True, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        bias = to_2tuple(bias)
        drop_probs = to_2tuple(drop)
        self.fc1 = LinearD8(in_features, hidden_features, bias=bias[0])
        self.act = act_layer()
        self.drop1 = DropoutD8(drop_probs[0])
        self.norm = norm_layer(hidden_features) if norm_layer is not None else nn.Identity()
        self.fc2 = LinearD8(hidden_features, out_features, bias=bias[1])
        self.drop2 = DropoutD8(drop_probs[1])

    def forward(self, xs, resid=None, rs=None, cs=None, next_norm=None):
        xs = self.norm(self.drop1(self.act(self.fc1(xs))))
        if self.drop2.active or resid is None:
            out = _tail(self.drop2(self.fc2(xs)), resid, rs, cs)
            return out if next_norm is None else (out, None)
        return self.fc2(xs, resid=resid, rs=rs, cs=cs, next_norm=next_norm)


def _tail(ys, resid, rs, cs):
    """Unfused ``resid + rs*cs*y`` (only reached with dropout p>0 or foreign sub-modules)."""
    if resid is None:
        return ys
    yp, c = as_packed(ys)
    yp = yp.to(resid.dtype)
    if cs is not None:
        yp = yp * torch.cat(tuple(cs) + (cs[4],)).to(yp.dtype)
    if rs is not None:
        yp = yp * rs.view(-1, *([1] * (yp.dim() - 1))).to(yp.dtype)
    return Octic((resid + yp).contiguous(), c)


def drop_path_d8(xs, drop_prob: float = 0., training: bool = False, scale_by_keep: bool = True):
    """d8_layers.py:249-271 — one Bernoulli mask per sample, shared by the five tensors."""
    if drop_prob == 0. or not training:
        return xs
    xp, c = as_packed(xs)
    m = _drop_path_mask(xp.shape[0], drop_prob, xp.device, scale_by_keep)
    return Octic((xp * m.view(-1, *([1] * (xp.dim() - 1))).to(xp.dtype)).contiguous(), c)


class DropPathD8(nn.Module):
    def __init__(self, drop_prob: float = 0., scale_by_keep: bool = True):
        super().__init__()
        self.drop_prob, self.scale_by_keep = drop_prob, scale_by_keep

    def mask(self, B, device):
        """Per-sample scale for the fused epilogue, or None when inactive."""
        if self.drop_prob == 0. or not self.training:
            return None
        return _drop_path_mask(B, self.drop_prob, device, self.scale_by_keep)

    def forward(self, xs):
        return drop_path_d8(xs, self.drop_prob, self.training, self.scale_by_keep)


# ------------------------------------------------------------------------------------------ lift
class LiftIrrepD8Conv2d(nn.Module):
    """d8_layers.py:284-382.  Holds the learned (p/2)x(p/2) quarter kernel; ``expand_weight`` builds the
    p x p kernel by symmetry (index gather, see d8_utils.expand_lift_kernel).  The convolution itself is
    executed by LiftD8 as one GEMM over all irreps."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, bias, irrep="A1"):
        super().__init__()
        if irrep not in ["A1", "A2", "B1", "B2", "E"]:
            raise ValueError("Invalid irrep.")
        if bias and not (irrep == "A1"):
            raise ValueError("Bias only ok for A1-irrep.")
        kernel_size = to_2tuple(kernel_size)
        if kernel_size[0] != kernel_size[1]:
            raise NotImplementedError("Non-square kernels not implemented")
        if kernel_size[0] % 2 != 0 or kernel_size[1] % 2 != 0:
            raise NotImplementedError("Odd kernel sizes not yet implemented")
        if (kernel_size[0] == 2 or kernel_size[1] == 2) and irrep in ["A2", "B1"]:
            raise ValueError(f"No {irrep} irrep in filter kernels of size 2.")
        self.kernel_size, self.stride, self.irrep = kernel_size, stride, irrep
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, kernel_size[0] // 2, kernel_size[1] // 2))
        self.reset_parameters()

    def reset_parameters(self) -> None:
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in = self.weight.shape[1] * self.weight.shape[2] * self.weight.shape[3]
            if fan_in != 0:
                bound = 1 / math.sqrt(fan_in)
                nn.init.uniform_(self.bias, -bound, bound)

    def expand_weight(self):
        return expand_lift_kernel(self.weight, self.irrep)

    def kernels(self):
        """Full kernel(s) of this conv: one for the 1-D irreps, (K, rot90 K) for E (d8_layers.py:377-381)."""
        k = self.expand_weight()
        return (k, k.rot90(k=1, dims=(-2, -1))) if self.irrep == "E" else (k,)

    def forward(self, x):
        # stand-alone use of a single irrep conv: same GEMM engine with this irrep's rows only
        ks = self.kernels()
        w = torch.cat([k.flatten(1) for k in ks], 0)
        D = w.shape[0]
        bias = None
        if self.bias is not None:
            bias = self.bias
        p = self.kernel_size[0]
        if self.stride != p and tuple(to_2tuple(self.stride)) != tuple(self.kernel_size):
            raise NotImplementedError("lift engine implements stride == kernel_size (patch embedding)")
        if D % 8:
            raise ValueError("out_channels must be a multiple of 8 for the HIP lift")
        out = OF.LiftFn.apply(x, w, bias if bias is None else torch.cat([bias] * len(ks)), None, None, p,
                              compute_dtype(x))
        B, _, Hh, Ww = x.shape
        maps = out.reshape(B, Hh // p, Ww // p, D).permute(0, 3, 1, 2)
        o = self.weight.shape[0]
        return (maps[:, :o], maps[:, o:]) if self.irrep == "E" else maps


class LiftD8(nn.Module):
    """d8_layers.py:384-411."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, bias):
        super().__init__()
        if out_channels % 8 != 0:
            raise ValueError()
        outs = out_channels // 8
        self.conv_A1 = LiftIrrepD8Conv2d(in_channels, outs, kernel_size, stride, bias=bias, irrep="A1")
        self.conv_A2 = LiftIrrepD8Conv2d(in_channels, outs, kernel_size, stride, bias=False, irrep="A2")
        self.conv_B1 = LiftIrrepD8Conv2d(in_channels, outs, kernel_size, stride, bias=False, irrep="B1")
        self.conv_B2 = LiftIrrepD8Conv2d(in_channels, outs, kernel_size, stride, bias=False, irrep="B2")
        self.conv_E_left = LiftIrrepD8Conv2d(in_channels, outs, kernel_size, stride, bias=False, irrep="E")
        self.conv_E_right = LiftIrrepD8Conv2d(in_channels, outs, kernel_size, stride, bias=False, irrep="E")

    def packed_weight(self):
        """[8c, Cin*p*p] kernel matrix with rows in packed channel order A1|A2|B1|B2|E_row0|E_row1:
        E_row0 = (E_left K, E_right K), E_row1 = (E_left rot K, E_right rot K)."""
        ws = [c.weight for c in (self.conv_A1, self.conv_A2, self.conv_B1, self.conv_B2, self.conv_E_left, self.conv_E_right)]
        if all(w.dtype == ws[0].dtype and w.device == ws[0].device and w.shape == ws[0].shape for w in ws):
            return packed_lift_weight(ws)            # one signed gather (d8_utils._LiftWeightFn) instead of ~55 small launches
        return self.packed_weight_composed()

    def packed_weight_composed(self):
        """The same matrix built the way the reference composes it (expand, rot90, flatten, cat)."""
        el, er = self.conv_E_left.kernels(), self.conv_E_right.kernels()
        ks = [self.conv_A1.expand_weight(), self.conv_A2.expand_weight(), self.conv_B1.expand_weight(),
              self.conv_B2.expand_weight(), el[0], er[0], el[1], er[1]]
        return torch.cat([k.flatten(1) for k in ks], dim=0)

    def packed_bias(self):
        if self.conv_A1.bias is None:
            return None
        c = self.conv_A1.bias.shape[0]
        return torch.cat((self.conv_A1.bias, self.conv_A1.bias.new_zeros(7 * c)))

    def tokens(self, img, pos=None, cls_row=None):
        """Packed tokens [B, (1+)G*G, 8c] f32 (+ unfolded positional embedding and cls row, fused)."""
        p = self.conv_A1.kernel_size[0]
        if torch.compiler.is_compiling():
            from . import dispatch as _D   # noqa: F401
            return torch.ops.octic.lift(img, self.packed_weight(), self.packed_bias(), pos, cls_row, p,
                                        compute_dtype(img) == torch.bfloat16)[0]
        return OF.LiftFn.apply(img, self.packed_weight(), self.packed_bias(), pos, cls_row, p, compute_dtype(img))

    def forward(self, img):
        B, _, Hh, Ww = img.shape
        p = self.conv_A1.kernel_size[0]
        out = self.tokens(img)
        c = out.shape[-1] // 8
        maps = out.reshape(B, Hh // p, Ww // p, 8 * c).permute(0, 3, 1, 2)
        a = [maps[:, i * c:(i + 1) * c] for i in range(8)]
        # packed order is (.., x4|x6, x5|x7): back to the reference's (x4, x5, x6, x7)
        return (a[0], a[1], a[2], a[3], a[4], a[6], a[5], a[7])


class PatchEmbedD8(nn.Module):
    """d8_layers.py:413-497."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, norm_layer=None, flatten=True,
                 bias=True, strict_img_size=True):
        super().__init__()
        self.patch_size = to_2tuple(patch_size)
        self.img_size, self.grid_size, self.num_patches = self._init_img_size(img_size)
        if embed_dim % 8 != 0:
            raise ValueError()
        self.flatten = flatten
        self.strict_img_size = strict_img_size
        self.lift8 = LiftD8(in_chans, embed_dim, kernel_size=self.patch_size, stride=self.patch_size, bias=bias)
        self.norm = norm_layer(embed_dim) if norm_layer else nn.Identity()

    def _init_img_size(self, img_size):
        assert self.patch_size
        if img_size is None:
            return None, None, None
        img_size = to_2tuple(img_size)
        grid_size = tuple([s // p for s, p in zip(img_size, self.patch_size)])
        return img_size, grid_size, grid_size[0] * grid_size[1]

    def _check(self, x):
        _, _, H, W = x.shape
        if self.img_size is not None:
            if self.strict_img_size:
                assert H == self.img_size[0], f"Input height ({H}) doesn't match model ({self.img_size[0]})."
                assert W == self.img_size[1], f"Input width ({W}) doesn't match model ({self.img_size[1]})."
            else:
                patch_W, patch_H = self.patch_size
                assert H % (patch_H * 2) == 0, f"Input image height {H} is not an even multiple of patch height {patch_H}"
                assert W % (patch_W * 2) == 0, f"Input image width {W} is not an even multiple of patch width: {patch_W}"

    def tokens(self, x, pos=None, cls_row=None):
        self._check(x)
        return self.lift8.tokens(x, pos, cls_row)

    def forward(self, x):
        self._check(x)
        if not self.flatten:
            xs = self.lift8(x)
            return self.norm(convert_8tuple_to_5tuple(xs))
        t = self.lift8.tokens(x)
        return self.norm(Octic(t, t.shape[-1] // 8))

    def _init_weights(self):
        for conv in (self.lift8.conv_A1, self.lift8.conv_A2, self.lift8.conv_B1, self.lift8.conv_B2,
                     self.lift8.conv_E_left, self.lift8.conv_E_right):
            w = conv.weight.data
            torch.nn.init.xavier_uniform_(w.view([w.shape[0], -1]))


class IsotypicToPatchD8(nn.Module):
    """d8_layers.py:499-588 — irrep features back to pixel patches (not on the training path; the
    linear is the HIP LinearD8, the symmetric unfolding is index glue)."""

    def __init__(self, dim, patch_side, out_channels=3, bias=True, reshape_to_image=False):
        super().__init__()
        if patch_side % 2 != 0:
            raise NotImplementedError("Odd patch side not implemented.")
        self.dim, self.patch_side, self.out_channels = dim, patch_side, out_channels
        self.reshape_to_image = reshape_to_image
        self.lin8 = LinearD8(dim, 2 * (patch_side ** 2 * out_channels), bias=bias)

    def forward(self, xs):
        from .d8_utils import unfold_quarter, unfold_quarter_e_img
        B, L, _ = xs[0].shape
        h = self.patch_side // 2
        q = [0.25 * t.reshape(B, L, h, h, self.out_channels) for t in convert_5tuple_to_8tuple(self.lin8(xs))]
        out = sum(unfold_quarter(q[i], n, 2, 3) for i, n in enumerate(("A1", "A2", "B1", "B2")))
        out = out + unfold_quarter_e_img(SQRT2 * q[4]) + unfold_quarter_e_img(SQRT2 * q[5]).rot90(k=1, dims=(2, 3))
        if self.reshape_to_image:
            H = W = int(math.sqrt(L))
            p = self.patch_side
            out = out.reshape(B, H, W, p, p, self.out_channels).permute(0, 5, 1, 3, 2, 4)
            return out.reshape(B, self.out_channels, H * p, W * p)
        return out.reshape(B, L, self.patch_side ** 2 * self.out_channels)


# ------------------------------------------------------------------------------------- attention
class AttentionD8(nn.Module):
    """d8_layers.py:590-660.  qkv/proj are irrep-blocked GEMMs, head pack/unpack are HIP permutation
    kernels, the softmax core is the HIP attention kernel (csrc/attention.hip) in bf16 and torch SDPA in f32
    (scale = SDPA default 1/sqrt(head_dim); ``self.scale`` is stored but unused, as in the reference)."""

    def __init__(self, dim: int, num_heads: int = 8, qkv_bias: bool = True, proj_bias: bool = True,
                 attn_drop: float = 0.0, proj_drop: float = 0.0, rope=None, qk_scale=None):
        super().__init__()
        assert dim % num_heads == 0, "dim should be divisible by num_heads"
        assert (dim // num_heads) % 8 == 0, "dim should be divisible by 8"
        if rope is not None:
            raise NotImplementedError("RoPE not implemented")
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = qk_scale or head_dim ** -0.5
        self.qkv = LinearD8(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = LinearD8(dim, dim, bias=proj_bias)
        self.proj_drop = DropoutD8(proj_drop)
        self.rope = rope
        self.att = F.scaled_dot_product_attention

    def forward(self, xs, resid=None, rs=None, cs=None, next_norm=None):
        xp, c = as_packed(xs)
        if xp.dim() != 3:
            raise ValueError("AttentionD8 expects [B, N, C] irreps")
        qkv = self.qkv(xs if isinstance(xs, Octic) else Octic(xp, c))
        drop = self.attn_drop.p if self.training else 0.
        rag = OF.RAGGED
        if rag is not None and rag.matches(xp):
            # several crop sets in one row tensor: the softmax core walks the sets, everything around it ran once on all rows
            from . import ragged as _R
            if not (OF.ATTN_PACKED and drop == 0. and qkv.packed.is_cuda
                    and all(OF.ops.attn_packed_ok(T, c, self.num_heads, qkv.packed.dtype) for _, T, _ in rag.sets)):
                raise RuntimeError("AttentionD8: a ragged row tensor needs the packed attention kernels (bf16, head_dim 80 / 64)")
            on = Octic(_R.AttnPackedRaggedFn.apply(qkv.packed, rag, self.num_heads, c, (8 * c // self.num_heads) ** -0.5), c)
        elif (OF.ATTN_PACKED and drop == 0. and qkv.packed.is_cuda
                and OF.ops.attn_packed_ok(qkv.packed.shape[1], c, self.num_heads, qkv.packed.dtype)):
            # head split, softmax core and irrep re-assembly in the attention kernels themselves (head_dim 80)
            if torch.compiler.is_compiling():
                from . import dispatch as _D   # noqa: F401
                on = Octic(torch.ops.octic.attn_packed(qkv.packed, self.num_heads, c, (8 * c // self.num_heads) ** -0.5)[0], c)
            else:
                on = Octic(OF.AttnPackedFn.apply(qkv.packed, self.num_heads, c, (8 * c // self.num_heads) ** -0.5), c)
        else:
            q, k, v = OF.PackHeadsFn.apply(qkv.packed, self.num_heads, c)
            # HIP attention core for the shapes it covers (bf16, T <= 320, no dropout); torch SDPA (== self.att) otherwise
            o = OF.attention_core(q, k, v, dropout_p=drop)
            on = Octic(OF.UnpackHeadsFn.apply(o, c), c)
        if self.proj_drop.active or resid is None:
            out = _tail(self.proj_drop(self.proj(on)), resid, rs, cs)
            return out if next_norm is None else (out, None)
        return self.proj(on, resid=resid, rs=rs, cs=cs, next_norm=next_norm)


def _branch(norm, fn, xs_packed, c, rs, cs, out_dtype, pre=None, next_norm=None):
    """x + drop_path(cs * fn(norm(x))) with the tail fused into fn's last GEMM when fn supports it.
    pre: norm(x) already computed by the previous branch's last layer (then x itself is the residual source).
    next_norm: ask fn's last layer to also return next_norm(result) -> (stream, normalised or None)."""
    x = Octic(xs_packed, c)
    if pre is not None:
        xn = pre
    elif type(norm) is LayerNormD8:
        xn, xs_packed = norm(x, _out_dtype=out_dtype, _with_resid=True)
    else:
        try:
            xn = norm(x, _out_dtype=out_dtype)
        except TypeError:  # foreign norm layer
            xn = norm(x)
    if next_norm is not None:
        try:
            return fn(xn, resid=xs_packed, rs=rs, cs=cs, next_norm=next_norm)
        except TypeError:  # foreign attention / mlp class
            pass
    try:
        out = fn(xn, resid=xs_packed, rs=rs, cs=cs)
    except TypeError:  # foreign attention / mlp class: compose
        out = _tail(fn(xn), xs_packed, rs, cs)
    return out if next_norm is None else (out, None)


def _branch_compact(norm, fn, xs_packed, c, dp, cs, out_dtype):
    """_branch on the samples this branch's stochastic-depth mask keeps (COMPACT_DROP_PATH): the kept rows are gathered,
    run through the same fused branch with the constant 1 / keep as their row scale, and written back."""
    idx, n, scale = _compact_plan(xs_packed.shape[0], dp, xs_packed.device)
    link = _RowLink()
    xa = _GatherRowsFn.apply(xs_packed, idx, link)
    out = _branch(norm, fn, xa, c, _const_scale(n, scale, xa.device), cs, out_dtype)
    return Octic(_ScatterRowsFn.apply(xs_packed, idx, out.packed, link), c)


class Layer_scale_init_BlockD8(nn.Module):
    """d8_layers.py:665-707 (DeiT-III block; gamma_{1,2} = AffineD8(bias=False), one drop_path module
    used twice = two independent per-sample masks)."""

    def __init__(self, dim, num_heads, mlp_ratio=4, qkv_bias=False, qk_scale=None, attn_drop=0., drop=0.,
                 drop_path=0., act_layer=TritonGeluD8, norm_layer=LayerNormD8, Attention_block=AttentionD8,
                 Mlp_block=MlpD8, init_values=1e-4):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention_block(dim=dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale,
                                    attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = DropPathD8(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp_block(in_features=dim, hidden_features=int(mlp_ratio * dim), act_layer=act_layer, drop=drop)
        self.gamma_1 = AffineD8(dim, bias=False)
        self.gamma_2 = AffineD8(dim, bias=False)
        with torch.no_grad():
            for p in list(self.gamma_1.parameters()) + list(self.gamma_2.parameters()):
                p.fill_(init_values)

    def _mask(self, B, device):
        return self.drop_path.mask(B, device) if isinstance(self.drop_path, DropPathD8) else None

    def forward(self, xs):
        xp, c = as_packed(xs)
        dt = compute_dtype(xp)
        if compact_active(self.drop_path) and xp.dtype == torch.float32:
            x1 = _branch_compact(self.norm1, self.attn, xp, c, self.drop_path, self.gamma_1.alphas(), dt)
            return _branch_compact(self.norm2, self.mlp, x1.packed, c, self.drop_path, self.gamma_2.alphas(), dt)
        if torch.compiler.is_compiling() or not (OF.OCTIC_NEXT_NORM and xp.is_cuda and dt == torch.bfloat16
                                                 and type(self.norm2) is LayerNormD8):
            x1 = _branch(self.norm1, self.attn, xp, c, self._mask(xp.shape[0], xp.device), self.gamma_1.alphas(), dt)
            return _branch(self.norm2, self.mlp, x1.packed, c, self._mask(xp.shape[0], xp.device), self.gamma_2.alphas(), dt)
        # norm2 comes out of the attention branch's last layer, norm1 of the NEXT block (link_octic_blocks) out of the
        # MLP's: each pair is one autograd node whose backward needs no cast pass (OF.LinearD8NormFn)
        pre = getattr(xs, "_prenorm", None)
        # valid only for the packed stream exactly as the previous block returned it (version counter: in-place edits)
        pre1 = pre[1] if (pre is not None and pre[0] is self.norm1 and pre[1] is not None
                          and pre[2] == xp._version and pre[3] == xp.data_ptr()) else None
        x1, y2 = _branch(self.norm1, self.attn, xp, c, self._mask(xp.shape[0], xp.device), self.gamma_1.alphas(), dt,
                         pre=pre1, next_norm=self.norm2)
        nn_ = getattr(self, "_next_norm", None)
        nxt = nn_[0] if nn_ else None
        if nxt is None:
            return _branch(self.norm2, self.mlp, x1.packed, c, self._mask(xp.shape[0], xp.device), self.gamma_2.alphas(),
                           dt, pre=y2)
        x2, yn = _branch(self.norm2, self.mlp, x1.packed, c, self._mask(xp.shape[0], xp.device), self.gamma_2.alphas(), dt,
                         pre=y2, next_norm=nxt)
        if yn is not None:
            x2._prenorm = (nxt, yn, x2.packed._version, x2.packed.data_ptr())
        return x2


def link_octic_blocks(blocks):
    """Tell every Layer_scale_init_BlockD8 / BlockD8 which LayerNormD8 follows it (norm1 of the next one in `blocks`); held
    in a tuple: not a sub-module, state_dict keys do not change."""
    seq = [b for b in blocks if isinstance(b, (Layer_scale_init_BlockD8, BlockD8))]
    for cur, nxt in zip(seq[:-1], seq[1:]):
        if type(nxt.norm1) is LayerNormD8:
            cur._next_norm = (nxt.norm1,)


class BlockD8(nn.Module):
    """d8_layers.py:713-776 (DINOv2 / model-default block; ls{1,2} = LayerScaleD8, drop_path{1,2})."""

    def __init__(self, dim: int, num_heads: int, mlp_ratio: float = 4.0, qkv_bias: bool = False,
                 proj_bias: bool = True, ffn_bias: bool = True, drop: float = 0.0, attn_drop: float = 0.0,
                 init_values=None, drop_path: float = 0.0, act_layer=TritonGeluD8, norm_layer=LayerNormD8,
                 attn_class=AttentionD8, ffn_layer=MlpD8) -> None:
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = attn_class(dim, num_heads=num_heads, qkv_bias=qkv_bias, proj_bias=proj_bias, attn_drop=attn_drop,
                               proj_drop=drop)
        self.ls1 = LayerScaleD8(dim, init_values=init_values) if init_values else nn.Identity()
        self.drop_path1 = DropPathD8(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = ffn_layer(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop,
                             bias=ffn_bias)
        self.ls2 = LayerScaleD8(dim, init_values=init_values) if init_values else nn.Identity()
        self.drop_path2 = DropPathD8(drop_path) if drop_path > 0.0 else nn.Identity()
        self.sample_drop_ratio = drop_path

    def forward(self, xs):
        xp, c = as_packed(xs)
        dt = compute_dtype(xp)
        B, dev = xp.shape[0], xp.device
        rag = OF.RAGGED if (OF.RAGGED is not None and OF.RAGGED.matches(xp)) else None
        if rag is not None:                          # one mask per SAMPLE of every crop set, handed on as per-row factors
            B = rag.samples
        pool = rag.masks if rag is not None else None
        if pool is not None:                         # drawn for the whole pass at once (ragged.Ragged.draw_masks)
            m1 = pool.get(id(self.drop_path1))
        else:
            m1 = self.drop_path1.mask(B, dev) if isinstance(self.drop_path1, DropPathD8) else None
            if rag is not None and m1 is not None:
                m1 = rag.row_scale(m1)
        cs1 = self.ls1.alphas() if isinstance(self.ls1, LayerScaleD8) else None
        # norm2 out of the attention branch's last layer, norm1 of the NEXT block out of the MLP's (link_octic_blocks), as in
        # Layer_scale_init_BlockD8: each pair is one autograd node whose backward needs no cast pass (OF.LinearD8NormFn)
        chain = (not torch.compiler.is_compiling() and OF.OCTIC_NEXT_NORM and xp.is_cuda and dt == torch.bfloat16
                 and type(self.norm2) is LayerNormD8)
        y2 = None
        if chain:
            pre = getattr(xs, "_prenorm", None)
            pre1 = pre[1] if (pre is not None and pre[0] is self.norm1 and pre[1] is not None
                              and pre[2] == xp._version and pre[3] == xp.data_ptr()) else None
            x1, y2 = _branch(self.norm1, self.attn, xp, c, m1, cs1, dt, pre=pre1, next_norm=self.norm2)
        else:
            x1 = _branch(self.norm1, self.attn, xp, c, m1, cs1, dt)
        if pool is not None:
            m2 = pool.get(id(self.drop_path2))
        else:
            m2 = self.drop_path2.mask(B, dev) if isinstance(self.drop_path2, DropPathD8) else None
            if rag is not None and m2 is not None:
                m2 = rag.row_scale(m2)
        cs2 = self.ls2.alphas() if isinstance(self.ls2, LayerScaleD8) else None
        nn_ = getattr(self, "_next_norm", None) if chain else None
        nxt = nn_[0] if nn_ else None
        if nxt is None:
            return _branch(self.norm2, self.mlp, x1.packed, c, m2, cs2, dt, pre=y2)
        x2, yn = _branch(self.norm2, self.mlp, x1.packed, c, m2, cs2, dt, pre=y2, next_norm=nxt)
        if yn is not None:
            x2._prenorm = (nxt, yn, x2.packed._version, x2.packed.data_ptr())
        return x2


class NestedTensorBlockD8(BlockD8):
    """d8_layers.py:780-794 — a list of crops is looped over."""

    def forward_nested(self, x_list):
        return [super(NestedTensorBlockD8, self).forward(x) for x in x_list]

    def forward(self, x_or_x_list):
        if isinstance(x_or_x_list, tuple):
            return super().forward(x_or_x_list)
        elif isinstance(x_or_x_list, list):
            return self.forward_nested(x_or_x_list)
        raise AssertionError
