"""DINOv2 entry points on the HIP engine: ``OcticDinoVisionTransformer`` and its four factories, with the
constructor, methods, output dictionaries and ``state_dict`` keys of the reference
(octic_vits/dinov2_models.py:41-329).  First slice of SURVEY.md section 8f row 4: the model side of the DINOv2
student/teacher step (mask tokens, register tokens, list-of-crops forward, intermediate layers); the losses and the
teacher update stay with the reference's recipe.

Token preparation works on packed rows: the lift GEMM produces the patch tokens, mask tokens / positional
embedding / cls and register rows are packed-row glue (once per forward, not on the block path)."""
from functools import partial
from typing import Callable, Sequence, Union

import torch
import torch.nn as nn

from . import functional as OF
from .d8_layers import NestedTensorBlockD8 as BlockD8
from .d8_utils import SQRT2_OVER_2, packed_pos_embed
from .deit_models import register_model
from .functional import Octic, compute_dtype
from .model import OcticVisionTransformer, trunc_normal_
from .vit import MemEffAttention, NestedTensorBlock as Block


def _pack8(ts):
    """8 tensors [..., c] in the reference's 8-tuple order -> packed channels [A1|A2|B1|B2|x4|x6|x5|x7]."""
    return torch.cat((ts[0], ts[1], ts[2], ts[3], ts[4], ts[6], ts[5], ts[7]), dim=-1)


RAGGED_LISTS = True     # a list of crop sets runs as one token-row tensor where the engine's regime allows it (ragged.py)
_RESIZE = {}


def _resize_matrix(gh, gw, oh, ow, device):
    key = (gh, gw, oh, ow, str(device))
    m = _RESIZE.get(key)
    if m is None:
        with torch.no_grad():
            eye = torch.eye(gh * gw, dtype=torch.float32).view(gh * gw, 1, gh, gw)
            m = torch.nn.functional.interpolate(eye, size=(oh, ow), mode="bicubic", antialias=False)
            m = _RESIZE[key] = m.reshape(gh * gw, oh * ow).t().contiguous().to(device)
    return m


class OcticDinoVisionTransformer(OcticVisionTransformer):
    def __init__(self, img_size: int = 224, patch_size: int = 16, embed_dim: int = 768, depth: int = 12,
                 num_heads: int = 12, mlp_ratio: float = 4.0, num_register_tokens: int = 0,
                 drop_path_rate: float = 0.0, octic_block_layers: Callable = BlockD8,
                 standard_block_layers: Callable = partial(Block, attn_class=MemEffAttention),
                 invariant: bool = False, **kwargs):
        super().__init__(img_size=img_size, patch_size=patch_size, embed_dim=embed_dim, depth=depth,
                         num_heads=num_heads, mlp_ratio=mlp_ratio, num_register_tokens=0,
                         octic_block_layers=octic_block_layers, standard_block_layers=standard_block_layers,
                         drop_path_rate=drop_path_rate, invariant=invariant, qkv_bias=True, ffn_bias=True,
                         proj_bias=True)
        self.depth = depth
        self.num_register_tokens = num_register_tokens
        c = embed_dim // 8
        g2 = img_size // patch_size // 2
        self.cls_token = nn.ParameterList([nn.Parameter(torch.zeros(1, 1, c), requires_grad=(i == 0)) for i in range(8)])
        self.pos_embed = nn.ParameterList([nn.Parameter(torch.empty(g2, g2, c)) for _ in range(6)])
        assert num_register_tokens >= 0
        self.register_tokens = (nn.ParameterList(
            [nn.Parameter(torch.zeros(1, num_register_tokens, c), requires_grad=(i == 0)) for i in range(8)])
            if num_register_tokens else None)
        assert depth % 2 == 0, "depth should be even!"
        self.chunked_blocks = False
        self.mask_token = nn.ParameterList([nn.Parameter(torch.zeros(1, c), requires_grad=(i == 0)) for i in range(8)])
        self.head = nn.Identity()
        self.init_weights()

    def init_weights(self):
        std = 8 * 0.02
        for p in list(self.pos_embed):
            trunc_normal_(p, std=std * SQRT2_OVER_2)
        nn.init.normal_(self.cls_token[0], std=1e-6)
        if self.register_tokens is not None:
            nn.init.normal_(self.register_tokens[0], std=1e-6)
        for m in self.modules():                      # named_apply(init_weights_vit_timm): Linear layers only
            if isinstance(m, nn.Linear):
                trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    # ------------------------------------------------------------------------------------------ tokens
    def prepare_tokens_with_masks(self, x, masks=None):
        """Packed token rows [B, 1 + registers + G*G, 8c] as an ``Octic`` (the reference returns the 5-tuple,
        dinov2_models.py:113-136; ``Octic`` is that tuple as views of one tensor)."""
        B, _, h, w = x.shape
        gh, gw = self.pos_embed[0].shape[0] * 2, self.pos_embed[0].shape[1] * 2
        ps = self.patch_embed.patch_size
        pos = packed_pos_embed(self.pos_embed)                       # [G*G, 8c]
        if (h // ps[0], w // ps[1]) != (gh, gw) or h != w:
            # d8_utils.interpolate_spatial_tuple (:453-499): bicubic resize of the unfolded grids to the crop's token grid
            # (channel-wise, so the packed tensor is resized in one call).  The reference divides by the TUPLE patch size
            # there and raises TypeError as shipped; the integer patch side is the evident intent (the
            # DINOv2 multi-crop recipe needs 96 x 96 local crops on a 224 model).  Parity: oracle only.
            # Bicubic resampling is linear in its input and its weights depend on the two grid sizes only: the resize is
            # one [oh ow, gh gw] matrix (built once per size pair by pushing the identity through F.interpolate) times the
            # position table - the same sums as F.interpolate(pos) in another order, and two small GEMMs per step instead of
            # ATen's per-output-element kernels (0.5 ms forward + 0.75 ms atomics backward for a 14 x 14 -> 6 x 6 table).
            pos = _resize_matrix(gh, gw, h // ps[0], w // ps[1], pos.device) @ pos.float()
        cls_row = _pack8([t.flatten() for t in self.cls_token])
        c = self.embed_dim // 8
        if masks is None and self.register_tokens is None:
            return Octic(self.patch_embed.tokens(x, pos, cls_row), c)     # everything fused in the lift epilogue
        tok = self.patch_embed.tokens(x)                                  # [B, G*G, 8c] f32, no pos / cls
        if masks is not None:
            mrow = _pack8([t.flatten() for t in self.mask_token]).to(tok.dtype)
            tok = torch.where(masks.unsqueeze(-1), mrow, tok)
        tok = tok + pos.to(tok.dtype)
        rows = [cls_row.to(tok.dtype).expand(B, 1, -1)]
        if self.register_tokens is not None:
            rows.append(_pack8([t[0] for t in self.register_tokens]).to(tok.dtype).expand(B, -1, -1))
        return Octic(torch.cat(rows + [tok], dim=1).contiguous(), c)

    def _hand_off(self, xs):
        c = self.embed_dim // 8
        if self.invariant:
            return self.invariant_proj(self.invariantization(xs, _out_dtype=compute_dtype(xs.packed)))
        return OF.HandoffCatFn.apply(xs.packed, c, xs.packed.dtype)

    def _out(self, x, masks):
        x_norm = self.norm(x)
        r = self.num_register_tokens
        return {"x_norm_clstoken": x_norm[:, 0], "x_norm_regtokens": x_norm[:, 1:r + 1],
                "x_norm_patchtokens": x_norm[:, r + 1:], "x_prenorm": x, "masks": masks}

    # ------------------------------------------------------------------------------------------ forward
    def forward_features_list(self, x_list, masks_list):
        xs = [self.prepare_tokens_with_masks(x, m) for x, m in zip(x_list, masks_list)]
        # (ssl.SSLTrainer: gradients of block parameters may be written at the END of backward only if every block module
        # contributed one gradient - true for the one-row-tensor pass, false for the set-by-set loop)
        self._single_use_pass = False
        if RAGGED_LISTS and self._ragged_ok(xs):
            self._single_use_pass = True
            return self._forward_features_ragged(xs, masks_list)
        for blk in self.blocks[:self.depth // 2]:
            xs = blk(xs)
        x = [self._hand_off(t) for t in xs]
        x = self._standard_half(x)
        return [self._out(t, m) for t, m in zip(x, masks_list)]

    def _ragged_ok(self, xs):
        """Can the crop sets run as ONE row tensor (ragged.py)?  The engine's bf16 training / inference regime on the GPU with
        attention shapes the packed kernels take, the hybrid (non-invariant) hand-off, the stock block classes."""
        from . import ops, vit
        from .d8_layers import BlockD8
        t = xs[0].packed
        if not (len(xs) > 1 and t.is_cuda and t.dtype == torch.float32 and torch.is_autocast_enabled("cuda")
                and torch.get_autocast_dtype("cuda") == torch.bfloat16 and not torch.compiler.is_compiling()
                and not self.invariant):
            return False
        c, brk = self.embed_dim // 8, self.depth // 2
        H = getattr(getattr(self.blocks[0], "attn", None), "num_heads", 0)
        if not (H and all(ops.attn_packed_ok(x.packed.shape[1], c, H, torch.bfloat16) for x in xs)):
            return False
        return (all(isinstance(b, BlockD8) and b.attn.attn_drop.p == 0. for b in self.blocks[:brk])
                and all(isinstance(b, vit.NestedTensorBlock) for b in self.blocks[brk:]))

    def _forward_features_ragged(self, xs, masks_list):
        """Both halves on the token rows of all crop sets at once: every row-wise kernel of a block runs once, every parameter
        enters the graph once, attention walks the sets (ragged.py)."""
        from . import ragged as R
        from . import vit
        rows, rag = R.concat([x.packed for x in xs])
        c = self.embed_dim // 8
        prev_r, prev_o = OF.RAGGED, vit.STREAM_OWNED[0]
        OF.RAGGED, vit.STREAM_OWNED[0] = rag, True
        try:
            if self.training:                        # every random draw of the pass up front: a handful of launches
                brk, dev = self.depth // 2, rows.device
                rag.draw_masks([d for b in self.blocks[:brk] for d in (b.drop_path1, b.drop_path2)], dev)
                sub = [b for b in self.blocks[brk:] if b.sample_drop_ratio > 0.1]
                if sub:
                    ratios = {b.sample_drop_ratio for b in sub}
                    keeps = ([max(int(B * (1 - sub[0].sample_drop_ratio)), 1) for B, _, _ in rag.sets]
                             if len(ratios) == 1 else None)        # one ratio for all blocks (the reference's uniform dpr)
                    rag.draw_perms(2 * len(sub), dev, keeps)
            t = Octic(rows, c)
            for blk in self.blocks[:self.depth // 2]:
                t = blk(t)
            x = self._hand_off(t)
            for blk in self.blocks[self.depth // 2:]:
                x = blk(x)
            xn = self.norm(x)
        finally:
            OF.RAGGED, vit.STREAM_OWNED[0] = prev_r, prev_o
        r = self.num_register_tokens
        return [{"x_norm_clstoken": n[:, 0], "x_norm_regtokens": n[:, 1:r + 1], "x_norm_patchtokens": n[:, r + 1:],
                 "x_prenorm": p, "masks": m} for n, p, m in zip(rag.views(xn), rag.views(x), masks_list)]

    def forward_features(self, x, masks=None):
        if isinstance(x, list):
            return self.forward_features_list(x, masks)
        xs = self.prepare_tokens_with_masks(x, masks)
        for blk in self.blocks[:self.depth // 2]:
            xs = blk(xs)
        x = self._standard_half(self._hand_off(xs))
        return self._out(x, masks)

    def _standard_half(self, x):
        """blocks[depth // 2:] on a tensor or a list of crop batches.  The stream handed from block to block is this loop's own
        (the hand-off made it, nobody else holds it), so the blocks may edit it in place (vit.STREAM_OWNED: no defensive copy
        in front of the batch-subset stochastic depth)."""
        from . import vit
        prev, vit.STREAM_OWNED[0] = vit.STREAM_OWNED[0], True
        try:
            for blk in self.blocks[self.depth // 2:]:
                x = blk(x)
        finally:
            vit.STREAM_OWNED[0] = prev
        return x

    def _get_intermediate_layers_not_chunked(self, x, n=1):
        xs = self.prepare_tokens_with_masks(x)
        total = len(self.blocks)
        take = range(total - n, total) if isinstance(n, int) else n
        assert all(i > self.depth // 2 for i in take), f"All block indices must be > half depth, got {take}"
        for blk in self.blocks[:self.depth // 2]:
            xs = blk(xs)
        t = self._hand_off(xs)
        output = []
        for i, blk in enumerate(self.blocks[self.depth // 2:], start=self.depth // 2):
            t = blk(t)
            if i in take:
                output.append(t)
        assert len(output) == len(take), f"only {len(output)} / {len(take)} blocks found"
        return output

    def get_intermediate_layers(self, x: torch.Tensor, n: Union[int, Sequence] = 1, reshape: bool = False,
                                return_class_token: bool = False, norm=True):
        if self.chunked_blocks:
            raise NotImplementedError("Chunked blocks not supported yet")
        outputs = self._get_intermediate_layers_not_chunked(x, n)
        if norm:
            outputs = [self.norm(out) for out in outputs]
        class_tokens = [out[:, 0] for out in outputs]
        outputs = [out[:, 1 + self.num_register_tokens:] for out in outputs]
        if reshape:
            B, _, w, h = x.shape
            p = self.patch_embed.patch_size[0]       # the reference divides by the tuple here (TypeError as shipped)
            outputs = [out.reshape(B, w // p, h // p, -1).permute(0, 3, 1, 2).contiguous() for out in outputs]
        if return_class_token:
            return tuple(zip(outputs, class_tokens))
        return tuple(outputs)

    def forward(self, *args, is_training=False, **kwargs):
        ret = self.forward_features(*args, **kwargs)
        return ret if is_training else self.head(ret["x_norm_clstoken"])

    @torch.jit.ignore
    def no_weight_decay(self):
        base_names = [f'pos_embed.{i}' for i in range(6)] + ['cls_token.0']
        return set(base_names + [f'_orig_mod.{name}' for name in base_names])


def _dinov2(patch_size, dim, depth, heads, invariant, num_register_tokens, kwargs):
    return OcticDinoVisionTransformer(
        patch_size=patch_size, embed_dim=dim, depth=depth, num_heads=heads, mlp_ratio=4, invariant=invariant,
        standard_block_layers=partial(Block, attn_class=MemEffAttention, init_values=1.0e-05),
        octic_block_layers=partial(BlockD8, init_values=1.0e-05), num_register_tokens=num_register_tokens, **kwargs)


@register_model
def hybrid_dinov2_vit_large_patch16(patch_size=16, num_register_tokens=0, **kwargs):
    return _dinov2(patch_size, 1024, 24, 16, False, num_register_tokens, kwargs)


@register_model
def hybrid_dinov2_vit_huge_patch16(patch_size=16, num_register_tokens=0, **kwargs):
    return _dinov2(patch_size, 1280, 32, 16, False, num_register_tokens, kwargs)


@register_model
def d8_inv_early_dinov2_vit_large_patch16(patch_size=16, num_register_tokens=0, **kwargs):
    return _dinov2(patch_size, 1024, 24, 16, True, num_register_tokens, kwargs)


@register_model
def d8_inv_early_dinov2_vit_huge_patch16(patch_size=16, num_register_tokens=0, **kwargs):
    return _dinov2(patch_size, 1280, 32, 16, True, num_register_tokens, kwargs)
