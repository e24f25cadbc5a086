"""Import the real reference (``/root/reference``) on CPU, in THIS container only.

Used solely by ``tests/golden/make_golden.py`` to generate the committed golden
vectors and by the optional ``-m "not gpu"`` cross-check test that is skipped
when ``/root/reference`` is absent (e.g. on the GPU box).  Nothing here is
product code and nothing here travels as reference source: the reference is
imported from where it lies.

The reference depends on ``timm`` (not installed here, pinned to 1.0.12 by
``deit/environment.yml:85-89``).  A minimal in-memory stand-in is injected for
the handful of names the hot path touches (call sites: ``octic_vits/d8_layers.py:12,456-463``,
``octic_vits/model.py:17,21``, ``octic_vits/deit_models.py:7``, ``deit/vit.py:9-12``).
``timm``'s published arithmetic for those names is restated here (``Mlp`` =
fc1→act→drop→fc2→drop, ``Block`` = pre-norm block with optional LayerScale,
``DropPath`` = per-sample Bernoulli mask scaled by keep-prob).

The reference's Triton GELU cannot launch without a GPU; it is replaced by the
reference's own pure-torch twin ``GeluD8`` exactly as ``octic_vits/d8_gelu.py:517-541``
does for its self-test.
"""
import math
import os
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

REFERENCE_ROOT = os.environ.get("OCTIC_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "octic_vits"))


def _install_timm_stub():
    if "timm" in sys.modules and not getattr(sys.modules["timm"], "_octic_stub", False):
        return  # a real timm is present: use it
    timm = types.ModuleType("timm")
    timm._octic_stub = True
    layers = types.ModuleType("timm.layers")
    trace_utils = types.ModuleType("timm.layers.trace_utils")
    models = types.ModuleType("timm.models")
    vit = types.ModuleType("timm.models.vision_transformer")

    def _assert(cond, msg):
        assert cond, msg

    trace_utils._assert = _assert

    def trunc_normal_(tensor, mean=0.0, std=1.0, a=-2.0, b=2.0):
        return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)

    def to_2tuple(x):
        return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

    class DropPath(nn.Module):
        def __init__(self, drop_prob=0.0, scale_by_keep=True):
            super().__init__()
            self.drop_prob, self.scale_by_keep = drop_prob, scale_by_keep

        def forward(self, x):
            if self.drop_prob == 0.0 or not self.training:
                return x
            keep = 1 - self.drop_prob
            mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
            if keep > 0.0 and self.scale_by_keep:
                mask.div_(keep)
            return x * mask

    class Mlp(nn.Module):
        def __init__(self, in_features, hidden_features=None, out_features=None,
                     act_layer=nn.GELU, norm_layer=None, bias=True, drop=0.0, **_):
            super().__init__()
            out_features = out_features or in_features
            hidden_features = hidden_features or in_features
            self.fc1 = nn.Linear(in_features, hidden_features, bias=bias)
            self.act = act_layer()
            self.drop1 = nn.Dropout(drop)
            self.norm = nn.Identity()
            self.fc2 = nn.Linear(hidden_features, out_features, bias=bias)
            self.drop2 = nn.Dropout(drop)

        def forward(self, x):
            return self.drop2(self.fc2(self.norm(self.drop1(self.act(self.fc1(x))))))

    class _Attention(nn.Module):
        def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0.0, proj_drop=0.0):
            super().__init__()
            self.num_heads = num_heads
            self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
            self.attn_drop = nn.Dropout(attn_drop)
            self.proj = nn.Linear(dim, dim)
            self.proj_drop = nn.Dropout(proj_drop)

        def forward(self, x):
            B, N, C = x.shape
            qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
            x = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2],
                                               dropout_p=self.attn_drop.p if self.training else 0.0)
            return self.proj_drop(self.proj(x.transpose(1, 2).reshape(B, N, C)))

    class _LayerScale(nn.Module):
        def __init__(self, dim, init_values=1e-5):
            super().__init__()
            self.gamma = nn.Parameter(init_values * torch.ones(dim))

        def forward(self, x):
            return x * self.gamma

    class Block(nn.Module):
        def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, qk_norm=False,
                     proj_drop=0.0, attn_drop=0.0, init_values=None, drop_path=0.0,
                     act_layer=nn.GELU, norm_layer=nn.LayerNorm, mlp_layer=Mlp, **_):
            super().__init__()
            self.norm1 = norm_layer(dim)
            self.attn = _Attention(dim, num_heads, qkv_bias, attn_drop, proj_drop)
            self.ls1 = _LayerScale(dim, init_values) if init_values else nn.Identity()
            self.drop_path1 = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
            self.norm2 = norm_layer(dim)
            self.mlp = mlp_layer(in_features=dim, hidden_features=int(dim * mlp_ratio),
                                 act_layer=act_layer, drop=proj_drop)
            self.ls2 = _LayerScale(dim, init_values) if init_values else nn.Identity()
            self.drop_path2 = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()

        def forward(self, x):
            x = x + self.drop_path1(self.ls1(self.attn(self.norm1(x))))
            return x + self.drop_path2(self.ls2(self.mlp(self.norm2(x))))

    class PatchEmbed(nn.Module):  # only imported by deit/vit.py, never built on our path
        def __init__(self, *a, **k):
            super().__init__()
            raise NotImplementedError

    _registry = {}

    def register_model(fn):
        _registry[fn.__name__] = fn
        return fn

    def create_model(name, **kwargs):
        kwargs.pop("pretrained", None)
        return _registry[name](**kwargs)

    def _cfg(**kwargs):
        return dict(kwargs)

    layers.trunc_normal_, layers.DropPath, layers.to_2tuple = trunc_normal_, DropPath, to_2tuple
    layers.trace_utils = trace_utils
    models.register_model, models.create_model, models._registry = register_model, create_model, _registry
    vit.Block, vit.Mlp, vit.PatchEmbed, vit._cfg = Block, Mlp, PatchEmbed, _cfg
    models.vision_transformer = vit
    timm.layers, timm.models = layers, models
    for name, mod in [("timm", timm), ("timm.layers", layers), ("timm.layers.trace_utils", trace_utils),
                      ("timm.models", models), ("timm.models.vision_transformer", vit)]:
        sys.modules[name] = mod


_REF = None


def load_reference():
    """Returns a namespace with the reference modules (CPU-runnable)."""
    global _REF
    if _REF is not None:
        return _REF
    if not reference_available():
        raise RuntimeError(f"reference not found at {REFERENCE_ROOT}")
    _install_timm_stub()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import octic_vits.d8_gelu as d8_gelu
    import octic_vits.d8_layers as d8_layers
    import octic_vits.d8_utils as d8_utils
    import octic_vits.d8_invariantization as d8_inv
    import octic_vits.model as model
    import octic_vits.deit_models as deit_models
    import deit.vit as deit_vit

    def _cpu_gelu_forward(self, xs):  # the reference's own torch twin of its Triton kernel
        return d8_utils.convert_8tuple_to_5tuple(
            d8_layers.GeluD8()(d8_utils.convert_5tuple_to_8tuple(xs)))

    d8_gelu.TritonGeluD8.forward = _cpu_gelu_forward
    _REF = types.SimpleNamespace(d8_gelu=d8_gelu, d8_layers=d8_layers, d8_utils=d8_utils,
                                 d8_inv=d8_inv, model=model, deit_models=deit_models,
                                 deit_vit=deit_vit,
                                 create_model=sys.modules["timm.models"].create_model)
    return _REF
