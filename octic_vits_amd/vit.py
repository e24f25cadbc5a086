"""Standard (non-octic) half of the hybrid models: vanilla pre-norm ViT blocks with the state_dict keys of
the reference (deit/vit.py:14-134 ``Attention``/``Layer_scale_init_Block``; timm 1.0.12 ``Block`` for the
bare default, model.py:21,63).  In f32 / on CPU these are stock PyTorch ops like the reference.  Under bf16 autocast on the
GPU a block runs on the engine instead (§8f-1, §8f-3): HIP LayerNorm / attention / layer-scale+drop-path+residual
kernels around hand-written GEMMs (csrc/dense_gemm.hip, csrc/dense_wgrad.hip; the BLAS library only for shapes they
refuse), with cached bf16 weights — same math, same parameters.  Drop-path masks have the reference's distribution;
inside a model forward they are drawn 64 at a time (d8_layers.DROP_PATH_POOL), so torch's generator is consumed in a
different order than by the reference's per-call draws (parity tests inject the reference's masks)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as _OF
from . import ops as _ops


class DropPath(nn.Module):
    def __init__(self, drop_prob: float = 0., scale_by_keep: bool = True):
        super().__init__()
        self.drop_prob, self.scale_by_keep = drop_prob, scale_by_keep

    def forward(self, x):
        if self.drop_prob == 0. or not self.training:
            return x
        keep = 1 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        if keep > 0.0 and self.scale_by_keep:
            mask.div_(keep)
        return x * mask


def _rows_per_scale(y, rs):
    """Rows that share one entry of the per-sample factor rs: the sequence length - or 1 when rs has one entry per ROW (several
    crop sets of different lengths in one [1, R, .] row tensor: ragged.py)."""
    if rs is not None and y.shape[0] == 1 and y.shape[1] > 1 and rs.numel() == y.shape[1]:
        return 1
    return y.shape[1]


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, norm_layer=None,
                 bias=True, drop=0.):
        super().__init__()
        out_features, hidden_features = out_features or in_features, hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features, bias=bias)
        self.act = act_layer()
        self.drop1 = nn.Dropout(drop)
        self.norm = norm_layer(hidden_features) if norm_layer is not None else nn.Identity()
        self.fc2 = nn.Linear(hidden_features, out_features, bias=bias)
        self.drop2 = nn.Dropout(drop)
        self._c1, self._c2 = _OF.DenseWeightCache(), _OF.DenseWeightCache()

    def forward(self, x):
        return self.drop2(self.fc2(self.norm(self.drop1(self.act(self.fc1(x))))))

    def fusable(self):
        drop = self.training and (self.drop1.p > 0. or self.drop2.p > 0.)
        return (not drop and isinstance(self.norm, nn.Identity) and type(self.act) is nn.GELU
                and self.act.approximate == "none" and type(self.fc1) is nn.Linear and type(self.fc2) is nn.Linear)

    def rows_ok(self, y):
        """forward_fused takes `rows_to` for this input (the hand-written path runs)."""
        return (_OF.dense_hip_ok(y, self.fc1.weight) and _OF.dense_hip_ok(y, self.fc2.weight)
                and bool({"fc1", "fc2", "dfc1", "dfc2"} & _OF.DENSE_HIP))

    def forward_fused(self, y, xres, gamma, rs, dtype, next_norm=None, rows_to=None):
        """xres + rs*gamma*fc2(gelu(fc1(y))) with y already normalised and in the compute dtype.  next_norm (an
        nn.LayerNorm): the hand-written path also returns next_norm(result) from the residual row pass -> (x, y_next).
        rows_to (functional.RowsTo): y / xres are compact rows of a stream; the result is written back into it."""
        rps = _rows_per_scale(y, rs)
        if rows_to is not None:
            if not (dtype == torch.bfloat16 and self.rows_ok(y)) or next_norm is not None:
                raise RuntimeError("Mlp.forward_fused: rows_to needs the hand-written bf16 path (check rows_ok first)")
            return _OF.DenseMlpFn.apply(y, xres, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, gamma,
                                        rs, rps, self._c1, self._c2, None, None, None, rows_to, rows_to.stream)
        if (dtype == torch.bfloat16 and _OF.dense_hip_ok(y, self.fc1.weight) and _OF.dense_hip_ok(y, self.fc2.weight)
                and ({"fc1", "fc2", "dfc1", "dfc2"} & _OF.DENSE_HIP)):
            if next_norm is not None:
                return _OF.DenseMlpFn.apply(y, xres, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, gamma,
                                            rs, rps, self._c1, self._c2, next_norm.weight, next_norm.bias,
                                            next_norm.eps)
            return _OF.DenseMlpFn.apply(y, xres, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, gamma,
                                        rs, rps, self._c1, self._c2)
        if next_norm is not None:
            return self.forward_fused(y, xres, gamma, rs, dtype), None
        if dtype == torch.bfloat16 and self.fc1.out_features % 8 == 0:
            h = _OF.DenseLinearGeluFn.apply(y, self.fc1.weight, self.fc1.bias, self._c1)
        else:
            h = F.gelu(_OF.DenseLinearFn.apply(y, self.fc1.weight, self.fc1.bias, dtype, self._c1))
        return _OF.LinearScaleResidualFn.apply(xres, h, self.fc2.weight, self.fc2.bias, gamma, rs, rps, dtype,
                                               self._c2)


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0., fused_attn=True,
                 proj_bias=True):
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = qk_scale or head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim, bias=proj_bias)
        self.proj_drop = nn.Dropout(proj_drop)
        self.fused_attn = fused_attn
        self._c1, self._c2 = _OF.DenseWeightCache(), _OF.DenseWeightCache()
        self._wgpair = _OF.WgradPair()            # qkv + proj weight gradients as one launch (functional.WGRAD_PAIRED)

    def fusable(self, N, dtype):
        drop = self.training and (self.attn_drop.p > 0. or self.proj_drop.p > 0.)
        return (not drop and self.fused_attn and type(self.qkv) is nn.Linear and type(self.proj) is nn.Linear
                and _ops.attn_supported(N, self.qkv.in_features // self.num_heads, dtype))

    def rows_ok(self, y):
        return _OF.dense_hip_ok(y, self.proj.weight, "proj")

    def forward_fused(self, y, xres, gamma, rs, dtype, next_norm=None, rows_to=None):
        """xres + rs*gamma*proj(attention(qkv(y))) with y already normalised and in the compute dtype.  next_norm (an
        nn.LayerNorm): the hand-written path also returns next_norm(result) from the residual row pass -> (x, y_next).
        rows_to (functional.RowsTo): y / xres are compact rows of a stream; the result is written back into it."""
        B, N, C = y.shape
        hd = C // self.num_heads
        bf = dtype == torch.bfloat16
        if bf and _OF.dense_hip_ok(y, self.qkv.weight) and ({"qkv", "dqkv"} & _OF.DENSE_HIP):
            qkv = _OF.DenseLinearNTFn.apply(y, self.qkv.weight, self.qkv.bias, self._c1, "qkv", self._wgpair)
        else:
            qkv = _OF.DenseLinearFn.apply(y, self.qkv.weight, self.qkv.bias, dtype, self._c1)
        rag = _OF.RAGGED
        if rag is not None and rag.matches(y):       # several crop sets in one row tensor (ragged.py): attention per set
            from . import ragged as _R
            a = _R.AttnQKVRaggedFn.apply(qkv, rag, self.num_heads, hd ** -0.5)
        else:
            a = _OF.AttnFusedQKVFn.apply(qkv.view(B, N, 3, self.num_heads, hd), hd ** -0.5)
        rps = _rows_per_scale(y, rs)
        if rows_to is not None:
            if not (bf and self.rows_ok(y)) or next_norm is not None:
                raise RuntimeError("Attention.forward_fused: rows_to needs the hand-written bf16 path (check rows_ok first)")
            return _OF.DenseProjResidFn.apply(xres, a, self.proj.weight, self.proj.bias, gamma, rs, rps, self._c2,
                                              None, None, None, self._wgpair, rows_to, rows_to.stream)
        if bf and _OF.dense_hip_ok(y, self.proj.weight, "proj"):
            if next_norm is not None:
                return _OF.DenseProjResidFn.apply(xres, a, self.proj.weight, self.proj.bias, gamma, rs, rps, self._c2,
                                                  next_norm.weight, next_norm.bias, next_norm.eps, self._wgpair)
            return _OF.DenseProjResidFn.apply(xres, a, self.proj.weight, self.proj.bias, gamma, rs, rps, self._c2,
                                              None, None, None, self._wgpair)
        out = _OF.LinearScaleResidualFn.apply(xres, a, self.proj.weight, self.proj.bias, gamma, rs, rps, dtype, self._c2)
        return out if next_norm is None else (out, None)

    def forward(self, x):
        B, N, C = x.shape
        hd = C // self.num_heads
        qkv = self.qkv(x)
        drop = self.attn_drop.p if self.training else 0.
        if self.fused_attn and drop == 0. and qkv.is_cuda and _ops.attn_supported(N, hd, qkv.dtype):
            # HIP attention core reading q/k/v through strides of the fused projection output and writing [B,N,C]
            x = _OF.AttnFusedQKVFn.apply(qkv.view(B, N, 3, self.num_heads, hd), hd ** -0.5)
            return self.proj_drop(self.proj(x))
        qkv = qkv.reshape(B, N, 3, self.num_heads, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        if self.fused_attn:
            x = F.scaled_dot_product_attention(q, k, v, dropout_p=self.attn_drop.p if self.training else 0.)
        else:
            attn = self.attn_drop(((q * self.scale) @ k.transpose(-2, -1)).softmax(dim=-1))
            x = attn @ v
        return self.proj_drop(self.proj(x.transpose(1, 2).reshape(B, N, C)))


def _drop_path_scale(dp, x):
    """Per-sample stochastic-depth factor [B] drawn exactly like DropPath.forward draws its mask, or None."""
    if not isinstance(dp, DropPath) or dp.drop_prob == 0. or not dp.training:
        return None
    from . import d8_layers as _L
    if x.dtype == torch.float32 and _L._pool_armed and not torch.compiler.is_compiling():   # the pooled draw (one launch pair per 64 masks)
        return _L._drop_path_mask(x.shape[0], dp.drop_prob, x.device, dp.scale_by_keep)
    keep = 1 - dp.drop_prob
    mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
    if keep > 0.0 and dp.scale_by_keep:
        mask.div_(keep)
    return mask.view(-1)


def _fused_block(x, norm1, attn, gamma1, dp1, norm2, mlp, gamma2, dp2, next_norm=None):
    """bf16-autocast forward of one standard block on the engine's row kernels + GEMMs, or None when the block is not
    in that regime (f32 run, CPU, dropout active, exotic sub-modules): the caller then runs eager.
    With functional.NEXT_NORM_FUSED the residual add of a branch and the LayerNorm that opens the next branch are one row
    pass: norm2 comes out of the attention branch's tail; norm1 of the NEXT block (next_norm, set by link_blocks) out of
    the MLP's tail and travels as an attribute of the returned stream tensor (`_octic_prenorm`), which that block picks
    up instead of running its own norm1."""
    if not (x.is_cuda and x.ndim == 3 and x.dtype == torch.float32 and torch.is_autocast_enabled("cuda")
            and torch.get_autocast_dtype("cuda") == torch.bfloat16):
        return None
    d = x.shape[-1]
    for n in (norm1, norm2):
        if type(n) is not nn.LayerNorm or tuple(n.normalized_shape) != (d,) or d % 4 or d > 2048:
            return None
    if not (isinstance(attn, Attention) and isinstance(mlp, Mlp) and attn.fusable(x.shape[1], torch.bfloat16)
            and mlp.fusable()):
        return None
    for dp in (dp1, dp2):
        if not isinstance(dp, (DropPath, nn.Identity)):
            return None
    dt = torch.bfloat16
    from . import d8_layers as _L
    if torch.compiler.is_compiling():
        return _traced_block(x, norm1, attn, gamma1, dp1, norm2, mlp, gamma2, dp2)
    if _L.compact_active(dp1) and _L.compact_active(dp2):
        # stochastic depth as batch compaction (d8_layers.COMPACT_DROP_PATH): each branch on the samples its mask keeps
        for norm, branch, gamma, dp in ((norm1, attn, gamma1, dp1), (norm2, mlp, gamma2, dp2)):
            idx, n, scale = _L._compact_plan(x.shape[0], dp, x.device)
            link = _L._RowLink()
            xa = _L._GatherRowsFn.apply(x, idx, link)
            y, xres = _OF.DenseLayerNormFn.apply(xa, norm.weight, norm.bias, norm.eps, dt)
            out = branch.forward_fused(y, xres, gamma, _L._const_scale(n, scale, x.device), dt)
            x = _L._ScatterRowsFn.apply(x, idx, out, link)
        return x
    fuse = _OF.NEXT_NORM_FUSED
    pre = getattr(x, "_octic_prenorm", None)
    # the carried norm is only valid for the stream exactly as the previous block returned it: an in-place edit in between
    # (a forward hook doing x.mul_(), a token edit) bumps the version counter and the block normalises again
    if pre is not None and pre[0] is norm1 and pre[1] is not None and pre[2] == x._version:
        y, xres = pre[1], x                   # normalised by the previous block's residual pass
    else:
        y, xres = _OF.DenseLayerNormFn.apply(x, norm1.weight, norm1.bias, norm1.eps, dt)
    y2 = None
    if fuse:
        x, y2 = attn.forward_fused(y, xres, gamma1, _drop_path_scale(dp1, x), dt, next_norm=norm2)
    else:
        x = attn.forward_fused(y, xres, gamma1, _drop_path_scale(dp1, x), dt)
    if y2 is not None:
        y, xres = y2, x
    else:
        y, xres = _OF.DenseLayerNormFn.apply(x, norm2.weight, norm2.bias, norm2.eps, dt)
    if fuse and type(next_norm) is nn.LayerNorm and tuple(next_norm.normalized_shape) == (d,):
        out, yn = mlp.forward_fused(y, xres, gamma2, _drop_path_scale(dp2, x), dt, next_norm=next_norm)
        if yn is not None:
            out._octic_prenorm = (next_norm, yn, out._version)
        return out
    return mlp.forward_fused(y, xres, gamma2, _drop_path_scale(dp2, x), dt)


def _traced_block(x, norm1, attn, gamma1, dp1, norm2, mlp, gamma2, dp2):
    """The same block through the dispatcher ops of dispatch.py (torch.compile is tracing): eight custom ops + two in-graph
    mask draws, no Python-side state (weight caches, carried next-norm) - one graph per block."""
    from . import dispatch as _D   # noqa: F401
    o = torch.ops.octic
    B, N, C = x.shape
    H = attn.num_heads
    hd = C // H
    # the bf16 / transposed weight copies the fused optimizer keeps current in place (None, None: the ops cast per call)
    y = o.dense_layernorm(x, norm1.weight, norm1.bias, norm1.eps, True)[0]
    qkv = o.dense_linear(y, attn.qkv.weight, attn.qkv.bias, False, *attn._c1.static_nt())[0]
    a = o.attn_qkv(qkv.view(B, N, 3, H, hd), hd ** -0.5)[0]
    p = o.dense_linear(a, attn.proj.weight, attn.proj.bias, False, *attn._c2.static_nt())[0]
    x = o.scale_residual(x, p, gamma1, _drop_path_scale(dp1, x), N)
    y = o.dense_layernorm(x, norm2.weight, norm2.bias, norm2.eps, True)[0]
    Hd = mlp.fc1.weight.shape[0]
    if _D._mlp_hip_ok(B * N, C, Hd):
        # fc1 + GELU + fc2 as one op: gelu'(h) is kept as a bf16 factor and applied inside fc2's input-gradient epilogue
        f = o.dense_mlp(y, mlp.fc1.weight, mlp.fc1.bias, mlp.fc2.weight, mlp.fc2.bias, *mlp._c1.static_nt(),
                        *mlp._c2.static_nt())[0]
    else:
        h = o.dense_linear(y, mlp.fc1.weight, mlp.fc1.bias, True, *mlp._c1.static_nt())[0]
        f = o.dense_linear(h, mlp.fc2.weight, mlp.fc2.bias, False, *mlp._c2.static_nt())[0]
    return o.scale_residual(x, f, gamma2, _drop_path_scale(dp2, x), N)


def link_blocks(blocks):
    """Tell every standard block which LayerNorm follows it (norm1 of the next standard block in `blocks`), so that its
    MLP tail can emit that norm's output (functional.NEXT_NORM_FUSED).  The reference is held in a tuple: it is not a
    sub-module, state_dict keys do not change."""
    seq = [b for b in blocks if isinstance(b, (Layer_scale_init_Block, Block))]
    for cur, nxt in zip(seq[:-1], seq[1:]):
        cur._next_norm = (nxt.norm1,)


class Layer_scale_init_Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0.,
                 drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm, Attention_block=Attention, Mlp_block=Mlp,
                 init_values=1e-4, use_fused_attn=True):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention_block(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop,
                                    proj_drop=drop, fused_attn=use_fused_attn)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp_block(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        self.gamma_1 = nn.Parameter(init_values * torch.ones((dim)), requires_grad=True)
        self.gamma_2 = nn.Parameter(init_values * torch.ones((dim)), requires_grad=True)

    def forward(self, x):
        nn_ = getattr(self, "_next_norm", None)
        out = _fused_block(x, self.norm1, self.attn, self.gamma_1, self.drop_path, self.norm2, self.mlp, self.gamma_2,
                           self.drop_path, next_norm=nn_[0] if nn_ else None)
        if out is not None:
            return out
        x = x + self.drop_path(self.gamma_1 * self.attn(self.norm1(x)))
        x = x + self.drop_path(self.gamma_2 * self.mlp(self.norm2(x)))
        return x


class LayerScale(nn.Module):
    def __init__(self, dim, init_values=1e-5, inplace=False):
        super().__init__()
        self.inplace = inplace
        self.gamma = nn.Parameter(init_values * torch.ones(dim))

    def forward(self, x):
        return x.mul_(self.gamma) if self.inplace else x * self.gamma


class Block(nn.Module):
    """timm-1.0.12-compatible default block (keys norm1, attn.qkv, attn.proj, ls1.gamma, norm2, mlp.fc1/fc2, ls2.gamma)."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_norm=False, proj_drop=0., attn_drop=0.,
                 init_values=None, drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm, mlp_layer=Mlp, **kwargs):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, attn_drop=attn_drop, proj_drop=proj_drop)
        self.ls1 = LayerScale(dim, init_values=init_values) if init_values else nn.Identity()
        self.drop_path1 = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = mlp_layer(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=proj_drop)
        self.ls2 = LayerScale(dim, init_values=init_values) if init_values else nn.Identity()
        self.drop_path2 = DropPath(drop_path) if drop_path > 0. else nn.Identity()

    def forward(self, x):
        ls_ok = all(isinstance(ls, nn.Identity) or (isinstance(ls, LayerScale) and not ls.inplace)
                    for ls in (self.ls1, self.ls2))
        if ls_ok:
            nn_ = getattr(self, "_next_norm", None)
            out = _fused_block(x, self.norm1, self.attn, getattr(self.ls1, "gamma", None), self.drop_path1, self.norm2,
                               self.mlp, getattr(self.ls2, "gamma", None), self.drop_path2,
                               next_norm=nn_[0] if nn_ else None)
            if out is not None:
                return out
        x = x + self.drop_path1(self.ls1(self.attn(self.norm1(x))))
        x = x + self.drop_path2(self.ls2(self.mlp(self.norm2(x))))
        return x


SUBSET_FUSED = True             # developer A/B: False = the eager composition of drop_add_residual_stochastic_depth
ROW_MAPS = True                 # ragged pass: the subset's rows through row maps in the LayerNorm / tail kernels (False: gather + scatter)
STREAM_OWNED = [False]          # set by a model's block loop: the tensors handed from block to block belong to the loop
MemEffAttention = Attention     # dinov2.layers.MemEffAttention: same parameters; the HIP core replaces xformers


def drop_add_residual_stochastic_depth(x, residual_func, sample_drop_ratio: float = 0.0):
    """dinov2/layers/block.py:113-140: residual branch on a random batch subset, added back scaled by b / subset."""
    b = x.shape[0]
    keep = max(int(b * (1 - sample_drop_ratio)), 1)
    brange = torch.randperm(b, device=x.device)[:keep]
    residual = residual_func(x[brange]).flatten(1)
    return torch.index_add(x.flatten(1), 0, brange, residual.to(x.dtype), alpha=b / keep).view_as(x)


class NestedTensorBlock(Block):
    """dinov2.layers ``Block`` / ``NestedTensorBlock`` (dinov2/layers/block.py:43-111, 234-260), the standard block of
    the DINOv2 octic models (octic_vits/dinov2_models.py:12,55).  Same parameter names as the timm block.  Eval and
    drop_path <= 0.1 run through the fused engine path of ``Block``; drop_path > 0.1 in training uses the
    reference's batch-subset stochastic depth.  A list of crop batches is processed crop by crop (attention never
    mixes samples, so this equals the reference's packed xformers path)."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, proj_bias=True, ffn_bias=True, drop=0.0,
                 attn_drop=0.0, init_values=None, drop_path=0.0, act_layer=nn.GELU, norm_layer=nn.LayerNorm,
                 attn_class=None, ffn_layer=None, **kwargs):
        super().__init__(dim, num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, proj_drop=drop, attn_drop=attn_drop,
                         init_values=init_values, drop_path=drop_path, act_layer=act_layer, norm_layer=norm_layer)
        if not proj_bias:
            self.attn.proj.bias = None
        if not ffn_bias:
            self.mlp.fc1.bias = None
            self.mlp.fc2.bias = None
        self.sample_drop_ratio = drop_path

    def _subset_fused(self, x):
        """Both branches of ``drop_add_residual_stochastic_depth`` on the engine (bf16 autocast, GPU): the residual branch runs
        on the random batch subset through the same fused kernels as a full-batch block - rows gathered, LayerNorm -> qkv ->
        attention -> proj (or fc1 + GELU -> fc2) with ``x_subset + (b / keep) * gamma * f`` in the branch's tail, rows written
        back - instead of the eager composition (ATen LayerNorm, library GEMMs, layer-scale multiply, index_add).  The subsets
        are drawn exactly like the reference draws them (one randperm per branch, dinov2/layers/block.py:121-123), so a
        seeded run picks the same samples as the eager path.  None when the block is not in that regime."""
        if not (x.is_cuda and x.ndim == 3 and x.dtype == torch.float32 and torch.is_autocast_enabled("cuda")
                and torch.get_autocast_dtype("cuda") == torch.bfloat16 and not torch.compiler.is_compiling()):
            return None
        d = x.shape[-1]
        for n in (self.norm1, self.norm2):
            if type(n) is not nn.LayerNorm or tuple(n.normalized_shape) != (d,) or d % 4 or d > 2048:
                return None
        if not (isinstance(self.attn, Attention) and isinstance(self.mlp, Mlp)
                and self.attn.fusable(x.shape[1], torch.bfloat16) and self.mlp.fusable()):
            return None
        gammas = []
        for ls in (self.ls1, self.ls2):
            if isinstance(ls, nn.Identity):
                gammas.append(None)
            elif isinstance(ls, LayerScale) and not ls.inplace:
                gammas.append(ls.gamma)
            else:
                return None
        from . import d8_layers as _L
        b = x.shape[0]
        keep = max(int(b * (1 - self.sample_drop_ratio)), 1)
        if not STREAM_OWNED[0]:
            x = x.clone()           # the reference's index_add is out of place: the caller's tensor stays what it was.  A model
                                    # loop that owns the stream between its blocks (dinov2_models.forward_features*) says so.
        for norm, branch, gamma in ((self.norm1, self.attn, gammas[0]), (self.norm2, self.mlp, gammas[1])):
            idx = torch.randperm(b, device=x.device)[:keep]
            link = _L._RowLink()
            xa = _L._GatherRowsFn.apply(x, idx, link)
            y, xres = _OF.DenseLayerNormFn.apply(xa, norm.weight, norm.bias, norm.eps, torch.bfloat16)
            out = branch.forward_fused(y, xres, gamma, _L._const_scale(keep, b / keep, x.device), torch.bfloat16)
            x = _L._ScatterRowsFn.apply(x, idx, out, link)
        return x

    def _ragged(self, x, rag):
        """One block on the rows of several crop sets at once (ragged.py).  Training with drop_path > 0.1: the batch-subset
        stochastic depth per set (one randperm per set and branch), the kept samples of all sets gathered into one compact row
        tensor, the branch once on all of them with the set's b / keep as per-row factor; otherwise the fused block with
        per-row stochastic-depth masks.  None when the block cannot take the engine path (the caller splits the sets)."""
        from . import d8_layers as _L
        from . import ragged as _R
        if not (x.is_cuda and x.dtype == torch.float32 and torch.is_autocast_enabled("cuda")
                and torch.get_autocast_dtype("cuda") == torch.bfloat16 and not torch.compiler.is_compiling()):
            return None
        d = x.shape[-1]
        for n in (self.norm1, self.norm2):
            if type(n) is not nn.LayerNorm or tuple(n.normalized_shape) != (d,) or d % 4 or d > 2048:
                return None
        hd = d // self.attn.num_heads if isinstance(self.attn, Attention) else 0
        if not (isinstance(self.attn, Attention) and isinstance(self.mlp, Mlp) and self.mlp.fusable()
                and self.attn.fusable(rag.sets[0][1], torch.bfloat16) and _R.attn_sets_supported(rag, hd, torch.bfloat16)):
            return None
        gammas = []
        for ls in (self.ls1, self.ls2):
            if isinstance(ls, nn.Identity):
                gammas.append(None)
            elif isinstance(ls, LayerScale) and not ls.inplace:
                gammas.append(ls.gamma)
            else:
                return None
        bf = torch.bfloat16
        if self.training and self.sample_drop_ratio > 0.1:
            if not STREAM_OWNED[0]:
                x = x.clone()
            keeps = [max(int(B * (1 - self.sample_drop_ratio)), 1) for B, _, _ in rag.sets]
            sub = _R.Ragged([(k, T) for k, (_, T, _) in zip(keeps, rag.sets)])
            scale = sub.const_row_scale([B / k for k, (B, _, _) in zip(keeps, rag.sets)], x.device)
            prev = _OF.RAGGED
            try:
                _OF.RAGGED = sub
                # (rows_ok looks at the weights and an upper bound of the row count: decided once, before anything runs)
                maps_ok = ROW_MAPS and d % 4 == 0 and x.is_contiguous() and self.attn.rows_ok(x) and self.mlp.rows_ok(x)
                for norm, branch, gamma in ((self.norm1, self.attn, gammas[0]), (self.norm2, self.mlp, gammas[1])):
                    idxs, rowmap = rag.take_subset(keeps, x.device)
                    if rowmap is not None and maps_ok:
                        # the kept rows read / written THROUGH a row map by the LayerNorm and the residual tail themselves
                        link = _OF._Link()
                        y, xa = _OF.DenseLayerNormRowsFn.apply(x, rowmap, norm.weight, norm.bias, norm.eps, bf, link)
                        x = branch.forward_fused(y, xa, gamma, scale, bf, rows_to=_OF.RowsTo(x, rowmap, link))
                        continue
                    link = _L._RowLink()
                    xa = _R.GatherSetsFn.apply(x, idxs, rag, sub, link)
                    y, xres = _OF.DenseLayerNormFn.apply(xa, norm.weight, norm.bias, norm.eps, bf)
                    out = branch.forward_fused(y, xres, gamma, scale, bf)
                    x = _R.ScatterSetsFn.apply(x, idxs, rag, sub, out, link)
            finally:
                _OF.RAGGED = prev
            return x
        scales = []
        for dp in (self.drop_path1, self.drop_path1 if self.training and self.sample_drop_ratio > 0.0 else self.drop_path2):
            if isinstance(dp, DropPath) and dp.drop_prob > 0. and dp.training:      # (block.py:104 reuses drop_path1 in training)
                keep = 1 - dp.drop_prob
                m = torch.empty(rag.samples, device=x.device, dtype=torch.float32).bernoulli_(keep)
                scales.append(rag.row_scale(m / keep if (keep > 0.0 and dp.scale_by_keep) else m))
            else:
                scales.append(None)
        y, xres = _OF.DenseLayerNormFn.apply(x, self.norm1.weight, self.norm1.bias, self.norm1.eps, bf)
        x = self.attn.forward_fused(y, xres, gammas[0], scales[0], bf)
        y, xres = _OF.DenseLayerNormFn.apply(x, self.norm2.weight, self.norm2.bias, self.norm2.eps, bf)
        return self.mlp.forward_fused(y, xres, gammas[1], scales[1], bf)

    def _one(self, x):
        rag = _OF.RAGGED
        if rag is not None and rag.matches(x):
            out = self._ragged(x, rag)
            if out is not None:
                return out
            prev, _OF.RAGGED = _OF.RAGGED, None      # not in the engine's regime: set by set, as the reference loops
            try:
                parts = [self._one(v) for v in rag.views(x)]
            finally:
                _OF.RAGGED = prev
            from . import ragged as _R
            return _R.concat(parts)[0]
        if self.training and self.sample_drop_ratio > 0.1:
            out = self._subset_fused(x) if SUBSET_FUSED else None
            if out is not None:
                return out
            x = drop_add_residual_stochastic_depth(x, lambda t: self.ls1(self.attn(self.norm1(t))), self.sample_drop_ratio)
            return drop_add_residual_stochastic_depth(x, lambda t: self.ls2(self.mlp(self.norm2(t))),
                                                      self.sample_drop_ratio)
        if self.training and self.sample_drop_ratio > 0.0:
            x = x + self.drop_path1(self.ls1(self.attn(self.norm1(x))))
            return x + self.drop_path1(self.ls2(self.mlp(self.norm2(x))))     # block.py:104 reuses drop_path1
        return super().forward(x)

    def forward(self, x_or_x_list):
        if isinstance(x_or_x_list, torch.Tensor):
            return self._one(x_or_x_list)
        if isinstance(x_or_x_list, list):
            return [self._one(x) for x in x_or_x_list]
        raise AssertionError
