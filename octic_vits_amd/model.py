"""OcticVisionTransformer on the HIP engine — same constructor, attributes, methods and state_dict keys as
the reference (octic_vits/model.py:25-234), so checkpoints and ``timm.create_model``-style factories carry over.

Forward = one lift GEMM (patch embedding + positional embedding + cls row fused), k octic blocks on packed
token rows, the hand-off kernel (8-tuple channel order, or PowerSpectrum + projection), then the standard half.
"""
from functools import partial
from typing import Callable

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as OF
from .d8_invariantization import PowerSpectrumInvariant
from .d8_layers import BlockD8, LayerNormD8, PatchEmbedD8, TritonGeluD8
from .d8_utils import SQRT2_OVER_2, packed_pos_embed
from .functional import Octic, compute_dtype
from .vit import Block


def trunc_normal_(tensor, mean=0., std=1., a=-2., b=2.):
    return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)


class OcticVisionTransformer(nn.Module):
    def __init__(self, img_size: int = 224, patch_size: int = 16, in_chans: int = 3, num_classes: int = 1000,
                 embed_dim: int = 768, depth: int = 12, num_heads: int = 12, mlp_ratio: float = 4.,
                 qkv_bias: bool = False, drop_rate: float = 0., attn_drop_rate: float = 0., drop_path_rate: float = 0.,
                 octic_block_layers: Callable = BlockD8, standard_block_layers: Callable = Block,
                 Patch_layer: Callable = PatchEmbedD8, init_scale: float = 1e-4, num_register_tokens: int = 0,
                 global_pool: bool = False, invariant: bool = False, octic_equi_break_layer=None, **kwargs):
        super().__init__()
        assert embed_dim % 8 == 0, "embed_dim must be divisible by 8"
        self.dropout_rate = drop_rate
        self.global_pool = global_pool
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        if octic_equi_break_layer is None:
            assert depth % 2 == 0, "depth must be even"
            octic_equi_break_layer = depth // 2
        else:
            assert octic_equi_break_layer >= 0, "octic_equi_break_layer must be non-negative"
            assert octic_equi_break_layer < depth, "octic_equi_break_layer must be less than depth"
        self.octic_equi_break_layer = octic_equi_break_layer
        self.invariant = invariant
        self.num_register_tokens = num_register_tokens
        if self.invariant:
            self.invariantization = PowerSpectrumInvariant(embed_dim)
            self.invariant_proj = torch.nn.Linear(self.invariantization.output_dim, embed_dim)
        self.patch_embed = Patch_layer(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim)
        norm_layer = partial(nn.LayerNorm, eps=1e-6)
        c = embed_dim // 8
        if not global_pool:
            self.cls_token = nn.ParameterList(
                [nn.Parameter(torch.zeros(1, 1, c), requires_grad=(i == 0)) for i in range(4)]
                + [nn.Parameter(torch.zeros(1, 1, 2, 2 * c), requires_grad=False)])
        assert num_register_tokens >= 0
        if num_register_tokens > 0:
            self.register_tokens = nn.ParameterList(
                [nn.Parameter(torch.zeros(1, num_register_tokens, c), requires_grad=(i == 0)) for i in range(8)])
        g2 = img_size // patch_size // 2
        self.pos_embed = nn.ParameterList([nn.Parameter(torch.empty(g2, g2, c)) for _ in range(6)])
        dpr = [drop_path_rate for _ in range(depth)]
        self.blocks = nn.ModuleList([
            octic_block_layers(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                               attn_drop=attn_drop_rate, drop_path=dpr[i], norm_layer=LayerNormD8,
                               act_layer=TritonGeluD8, init_values=init_scale)
            if i < self.octic_equi_break_layer else
            standard_block_layers(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                                  attn_drop=attn_drop_rate, drop_path=dpr[i], norm_layer=norm_layer,
                                  act_layer=nn.GELU, init_values=init_scale)
            for i in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        from .vit import link_blocks
        link_blocks(self.blocks[self.octic_equi_break_layer:])      # residual add + next norm1 as one row pass
        from .d8_layers import link_octic_blocks
        link_octic_blocks(self.blocks[:self.octic_equi_break_layer])

        std = 8 * .02  # model.py:147
        if self.num_register_tokens > 0:
            nn.init.normal_(self.register_tokens[0], std=1e-6)
        for p in self.pos_embed:
            trunc_normal_(p, std=SQRT2_OVER_2 * std)
        if not global_pool:
            for p in self.cls_token:
                if p.requires_grad:
                    trunc_normal_(p, std=std)
        self.apply(self._init_weights)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.weight, 1.0)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)

    def _cls_row(self):
        """Packed cls token row [8c] (only cls_token.0 is trainable; the rest are frozen zeros, model.py:99-105)."""
        t = self.cls_token
        return torch.cat([t[0].flatten(), t[1].flatten(), t[2].flatten(), t[3].flatten(), t[4].flatten()])

    def forward_features(self, x):
        from .d8_layers import arm_drop_path_pool
        arm_drop_path_pool(True)
        try:
            return self._forward_features(x)
        finally:
            arm_drop_path_pool(False)

    def _forward_features(self, x):
        B, C_in, H, W = x.shape
        if self.num_register_tokens > 0:
            raise RuntimeError("num_register_tokens > 0 is broken in the reference (model.py:183-191) and unsupported")
        gh, gw = self.pos_embed[0].shape[0] * 2, self.pos_embed[0].shape[1] * 2
        ps = self.patch_embed.patch_size
        if (H // ps[0], W // ps[1]) != (gh, gw):
            raise NotImplementedError("non-native resolutions (the reference raises TypeError there, d8_utils.py:475)")
        pos = packed_pos_embed(self.pos_embed)                      # [G*G, D], f32, rebuilt per call (tiny)
        cls_row = None if self.global_pool else self._cls_row()
        tokens = self.patch_embed.tokens(x, pos, cls_row)           # [B, T, D] f32 packed
        c = self.embed_dim // 8
        xs = Octic(tokens, c)
        for blk in self.blocks[:self.octic_equi_break_layer]:
            xs = blk(xs)
        dt = compute_dtype(xs.packed)
        if self.invariant:
            x = self.invariant_proj(self.invariantization(xs, _out_dtype=dt))
        else:
            if torch.compiler.is_compiling():
                from . import dispatch as _D   # noqa: F401
                x = torch.ops.octic.handoff_cat(xs.packed, c, xs.packed.dtype == torch.bfloat16)
            else:
                x = OF.HandoffCatFn.apply(xs.packed, c, xs.packed.dtype)
        for blk in self.blocks[self.octic_equi_break_layer:]:
            x = blk(x)
        if self.global_pool:
            return self.norm(x).mean(dim=1)
        from . import d8_layers as _L
        if _L.COMPACT_DROP_PATH:
            # (with the other opt-in elimination of discarded work, d8_layers.COMPACT_DROP_PATH: LayerNorm is row-wise, so
            # normalising the cls rows alone gives the same numbers and the same parameter gradients as the reference's
            # norm over all B T rows followed by x[:, 0], octic_vits/model.py:204-211)
            return self.norm(x[:, 0])
        return self.norm(x)[:, 0]

    def forward(self, x):
        x = self.forward_features(x)
        if self.dropout_rate:
            x = F.dropout(x, p=float(self.dropout_rate), training=self.training)
        return self.head(x)

    def invalidate_weight_caches(self):
        """See functional.invalidate_weight_caches: required after an optimizer that updates ``p.data`` / raw
        pointers (apex FusedLAMB) — such writes do not bump the version counters the bf16 weight caches key on."""
        return OF.invalidate_weight_caches(self)

    @torch.jit.ignore
    def no_weight_decay(self):
        base_names = [f'pos_embed.{i}' for i in range(6)] + ['cls_token.0']
        return set(base_names + [f'_orig_mod.{name}' for name in base_names])
