"""ORACLE — CPU restatement of the reference's octic hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package, and only as the checker / reported baseline — never as a product path
(the product, ``octic_vits_amd``, fails loudly when its HIP library is missing).

Parity status: **pinned**.  Every function/class here is checked bit-for-tolerance against
golden vectors produced by running the real reference on CPU in the build container
(``tests/golden/make_golden.py`` → ``tests/golden/*.npz``; checked by
``tests/test_oracle_golden.py``).  One exception, stated where it occurs: the *bare default*
standard block (``Block``) restates timm 1.0.12's published ``vision_transformer.Block``
(``deit/environment.yml:85-89`` pins the version; timm is a dependency absent from
``/root/reference``); it is pinned only through the reference's own call site
(``octic_vits/model.py:127-137``) run with the same restatement.

The code is written from the math, not transliterated: transforms are sign
matrices, group actions are generated from the two generators, spatial unfolds are built
from one quadrant-assembly helper.  Each public name cites the reference lines it restates.
All names, constructor signatures and ``state_dict`` keys follow the reference so that the
shared case runner (``tests/golden/cases.py``) can drive reference, oracle and product alike.
"""
import math
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F

SQRT2 = math.sqrt(2.0)
SQRT2_OVER_2 = SQRT2 / 2
SQRT2_OVER_4 = SQRT2 / 4

# ----------------------------------------------------------------------------------------------
# D8 Fourier transform (octic_vits/d8_utils.py:276-356).  Rows = output index.
# ----------------------------------------------------------------------------------------------
_S_ISO2REG = torch.tensor([
    [1, 1, 1, 1, 1, 1, 1, -1],
    [1, 1, -1, -1, 1, -1, -1, -1],
    [1, 1, 1, 1, -1, -1, -1, 1],
    [1, 1, -1, -1, -1, 1, 1, 1],
    [1, -1, 1, -1, -1, 1, -1, -1],
    [1, -1, -1, 1, -1, -1, 1, -1],
    [1, -1, 1, -1, 1, -1, 1, 1],
    [1, -1, -1, 1, 1, 1, -1, 1],
], dtype=torch.float32)
_S_REG2ISO = _S_ISO2REG.t().contiguous()  # the transform is orthogonal: inverse = transpose


def _apply8(mat, xs):
    """ys[i] = (√2/4) Σ_j mat[i,j] xs[j]   (exact ±1 matrix, one scale at the end)."""
    x = torch.stack(tuple(xs), dim=0)
    m = mat.to(device=x.device, dtype=x.dtype)
    y = torch.einsum("ij,j...->i...", m, x) * SQRT2_OVER_4
    return tuple(y.unbind(0))


def isotypic_to_regular_D8(xs):
    """d8_utils.py:276-303 (explicit form :305-315)."""
    return _apply8(_S_ISO2REG, xs)


def regular_to_isotypic_D8(xs):
    """d8_utils.py:317-344 (explicit form :346-356)."""
    return _apply8(_S_REG2ISO, xs)


# ----------------------------------------------------------------------------------------------
# 5-tuple <-> 8-tuple (d8_utils.py:358-385):  E[..., r, 0:c] = x(4+r),  E[..., r, c:2c] = x(6+r)
# ----------------------------------------------------------------------------------------------
def convert_8tuple_to_5tuple(xs):
    e = torch.stack((torch.cat((xs[4], xs[6]), -1), torch.cat((xs[5], xs[7]), -1)), dim=-2)
    return (xs[0], xs[1], xs[2], xs[3], e)


def convert_5tuple_to_8tuple(xs):
    e = xs[4]
    c = e.shape[-1] // 2
    return (xs[0], xs[1], xs[2], xs[3], e[..., 0, :c], e[..., 1, :c], e[..., 0, c:], e[..., 1, c:])


# ----------------------------------------------------------------------------------------------
# Group D8 = <r, m>.  Element "m r^k" acts as: r applied k times, then m  (d8_utils.py:16,76-260).
# ----------------------------------------------------------------------------------------------
group_elements = ("e", "r", "rr", "rrr", "m", "mr", "mrr", "mrrr")
irreps = ("A1", "A2", "B1", "B2", "E11", "E21", "E21", "E22")


def _word(g):
    if g not in group_elements:
        raise ValueError("Invalid group element")
    return g.count("r"), g.startswith("m")


def _iso_r(x):  # d8_utils.py:182-192
    return (x[0], x[1], -x[2], -x[3], -x[5], x[4], -x[7], x[6])


def _iso_m(x):  # d8_utils.py:215-225
    return (x[0], -x[1], x[2], -x[3], -x[4], x[5], -x[6], x[7])


def _reg_r(x):  # d8_utils.py:99-109
    return (x[1], x[2], x[3], x[0], x[7], x[4], x[5], x[6])


def _reg_m(x):  # d8_utils.py:132-142
    return (x[4], x[5], x[6], x[7], x[0], x[1], x[2], x[3])


def _act(g, xs, rot, mir):
    k, m = _word(g)
    xs = tuple(xs)
    for _ in range(k):
        xs = rot(xs)
    return mir(xs) if m else xs


def isotypic_group_action(g, xs):
    return _act(g, xs, _iso_r, _iso_m)


def regular_group_action(g, xs):
    return _act(g, xs, _reg_r, _reg_m)


def image_space_group_action(g, img):
    """d8_utils.py:76-94: r = rot90 CCW on the last two dims, m = flip of the last dim."""
    k, m = _word(g)
    if k:
        img = img.rot90(k=k, dims=(-2, -1))
    return img.flip(-1) if m else img


def spatial_and_isotypic_group_action(g, xs):
    """d8_utils.py:262-274: tokens as a square grid, image action then isotypic action."""
    B, L, C = xs[0].shape
    side = int(math.sqrt(L))
    moved = tuple(
        image_space_group_action(g, x.transpose(1, 2).reshape(B, C, side, side)).flatten(2).transpose(1, 2)
        for x in xs)
    return isotypic_group_action(g, moved)


def _compose_table():
    """Multiplication triples (g1, g2, g1g2) with a(g1, a(g2, x)) = a(g1g2, x) — d8_utils.py:18-74,
    derived from the permutation representation instead of listed."""
    idx = tuple(range(8))
    perm = {g: regular_group_action(g, idx) for g in group_elements}
    inv = {v: k for k, v in perm.items()}
    table = []
    for g1 in group_elements[1:]:
        for g2 in group_elements[1:]:
            table.append([g1, g2, inv[regular_group_action(g1, regular_group_action(g2, idx))]])
    return table


mult_table = _compose_table()

# ----------------------------------------------------------------------------------------------
# Spatial quadrant assembly shared by the lift kernels and the positional-embedding unfold.
# ----------------------------------------------------------------------------------------------


def _quad(tl, bl, tr, br, d0, d1):
    """2h×2h block matrix over dims (d0, d1): tl top-left, bl bottom-left, tr top-right, br bottom-right."""
    return torch.cat((torch.cat((tl, bl), d0), torch.cat((tr, br), d0)), d1)


def _unfold_1d(q, kind, d0, d1):
    """A1/A2/B1/B2 symmetrisation of a quarter ``q``."""
    s = -1.0 if kind in ("B1", "B2") else 1.0
    rot = lambda k: q.rot90(k=k, dims=(d0, d1))
    full = _quad(q, s * rot(1), s * rot(3), rot(2), d0, d1)
    return full + full.flip(d1) if kind in ("A1", "B1") else full - full.flip(d1)


def _unfold_e(q, d0, d1):
    half = torch.cat((q, q.flip(d0)), d0)
    return torch.cat((half, -half.flip(d1)), d1)


def isotypic_dim_interpolation(xs, dim: int = 0):
    """d8_utils.py:388-451: six learned quarter grids [.., G/2, G/2, c] → eight full grids [.., G, G, c]."""
    d0, d1 = dim, dim + 1
    el, er = _unfold_e(xs[4], d0, d1), _unfold_e(xs[5], d0, d1)
    return (
        _unfold_1d(xs[0], "A1", d0, d1), _unfold_1d(xs[1], "A2", d0, d1),
        _unfold_1d(xs[2], "B1", d0, d1), _unfold_1d(xs[3], "B2", d0, d1),
        el, el.rot90(dims=(d0, d1)), er, er.rot90(dims=(d0, d1)),
    )


def interpolate_spatial_tuple(xs, interpolant, h, w, patch_size):
    """d8_utils.py:453-499.  Native resolution returns the interpolant untouched (:467-471).  The
    reference's resize branch divides by a *tuple* patch size and raises TypeError as shipped
    (SURVEY §5); the evident intent (integer patch side) is implemented and is **parity unpinned**."""
    n_native = interpolant[0].shape[0] ** 2
    if xs[0].shape[1] == n_native and w == h:
        return interpolant
    p = patch_size[0] if isinstance(patch_size, (tuple, list)) else patch_size
    stacked = torch.stack([t.float().reshape(t.shape[0], t.shape[1], -1) for t in interpolant], 0)
    out = F.interpolate(stacked.permute(0, 3, 1, 2), size=(h // p, w // p), mode="bicubic", antialias=False)
    out = out.permute(0, 2, 3, 1)
    return [out[i].reshape(out.shape[1], out.shape[2], *interpolant[i].shape[2:]).to(xs[0].dtype)
            for i in range(len(interpolant))]


# ----------------------------------------------------------------------------------------------
# Layers (octic_vits/d8_layers.py)
# ----------------------------------------------------------------------------------------------
def trunc_normal_(t, std=1.0):
    return nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2.0, b=2.0)


class DropoutD8(nn.Module):
    """d8_layers.py:84-96: one nn.Dropout applied to each of the 5 tensors (independent masks)."""

    def __init__(self, p=0.5, inplace=False):
        super().__init__()
        self.dropout = nn.Dropout(p=p, inplace=inplace)

    def forward(self, xs):
        return tuple(self.dropout(x) for x in xs[:5])


class GeluD8(nn.Module):
    """d8_layers.py:98-102 (8-tuple in/out): iso→regular, exact-erf GELU, regular→iso."""

    def forward(self, xs):
        return regular_to_isotypic_D8([F.gelu(r) for r in isotypic_to_regular_D8(xs)])


class TritonGeluD8(nn.Module):
    """d8_gelu.py:480-482 (5-tuple in/out); arithmetic = GeluD8 (pinned by d8_gelu.py:484-714)."""

    def forward(self, xs):
        return convert_8tuple_to_5tuple(GeluD8()(convert_5tuple_to_8tuple(xs)))


class LinearD8(nn.Module):
    """d8_layers.py:104-130: y_k = x_k W_kᵀ (+b on A1 only); both E rows share W_E."""

    def __init__(self, input_channels, output_channels, bias=True):
        super().__init__()
        if input_channels % 8 or output_channels % 8:
            raise ValueError()
        self.bias, self.input_channels, self.output_channels = bias, input_channels, output_channels
        ci, co = input_channels // 8, output_channels // 8
        self.lin_A1 = nn.Linear(ci, co, bias=bias)
        self.lin_A2 = nn.Linear(ci, co, bias=False)
        self.lin_B1 = nn.Linear(ci, co, bias=False)
        self.lin_B2 = nn.Linear(ci, co, bias=False)
        self.lin_E = nn.Linear(2 * ci, 2 * co, bias=False)

    def forward(self, xs):
        assert len(xs) == 5, "Input should be a 5-tuple"
        lins = (self.lin_A1, self.lin_A2, self.lin_B1, self.lin_B2, self.lin_E)
        return tuple(l(x) for l, x in zip(lins, xs))


class AffineD8(nn.Module):
    """d8_layers.py:132-158."""

    def __init__(self, dim, bias=True):
        super().__init__()
        if dim % 8:
            raise ValueError()
        for n in ("A1", "A2", "B1", "B2"):
            setattr(self, "alpha_" + n, nn.Parameter(torch.ones(dim // 8)))
        self.alpha_E = nn.Parameter(torch.ones(dim // 4))
        self.beta = nn.Parameter(torch.zeros(dim // 8)) if bias else None

    def forward(self, xs):
        a = (self.alpha_A1, self.alpha_A2, self.alpha_B1, self.alpha_B2, self.alpha_E)
        ys = [ai * x for ai, x in zip(a, xs)]
        if self.beta is not None:
            ys[0] = ys[0] + self.beta
        return tuple(ys)


class LayerNormD8(nn.Module):
    """d8_layers.py:161-186.  S = Σ_{1-D irreps} var + mean_rows(var(E rows)) + eps;
    std = (√2/4)·√S (eps inside the root); every segment has its own mean removed."""

    def __init__(self, channels, eps=1e-05, elementwise_affine=True, bias=True):
        super().__init__()
        self.scaling = AffineD8(channels, bias=bias) if elementwise_affine else nn.Identity()
        self.eps = eps

    def forward(self, xs):
        var = lambda t: t.var(dim=-1, unbiased=False, keepdim=True)
        S = var(xs[0]) + var(xs[1]) + var(xs[2]) + var(xs[3]) + var(xs[4]).mean(dim=-2) + self.eps
        std = SQRT2_OVER_4 * torch.sqrt(S)
        cen = [x - x.mean(dim=-1, keepdim=True) for x in xs]
        out = tuple(c / std for c in cen[:4]) + (cen[4] / std.unsqueeze(-1),)
        return self.scaling(out)


class LayerScaleD8(nn.Module):
    """d8_layers.py:189-212."""

    def __init__(self, dim, init_values=1e-5):
        super().__init__()
        if dim % 8:
            raise ValueError()
        for n in ("A1", "A2", "B1", "B2"):
            setattr(self, "alpha_" + n, nn.Parameter(init_values * torch.ones(dim // 8)))
        self.alpha_E = nn.Parameter(init_values * torch.ones(dim // 4))

    def forward(self, xs):
        a = (self.alpha_A1, self.alpha_A2, self.alpha_B1, self.alpha_B2, self.alpha_E)
        return tuple(ai * x for ai, x in zip(a, xs))


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


class MlpD8(nn.Module):
    """d8_layers.py:215-247."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=TritonGeluD8,
                 norm_layer=None, bias=True, drop=0.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        b, d = _pair(bias), _pair(drop)
        self.fc1 = LinearD8(in_features, hidden_features, bias=b[0])
        self.act = act_layer()
        self.drop1 = DropoutD8(d[0])
        self.norm = norm_layer(hidden_features) if norm_layer is not None else nn.Identity()
        self.fc2 = LinearD8(hidden_features, out_features, bias=b[1])
        self.drop2 = DropoutD8(d[1])

    def forward(self, xs):
        return self.drop2(self.fc2(self.norm(self.drop1(self.act(self.fc1(xs))))))


# test hook: callable (B, keep_prob) -> 0/1 mask [B] (float).  None = the reference's own draw.  Lets a test run the oracle on
# CHUNKS of a batch with the rows of the masks the whole batch would have drawn (tests/test_vith_gpu.py, batch 64).
drop_path_mask_source = None


def _draw_mask(x, keep):
    shape = (x.shape[0],) + (1,) * (x.ndim - 1)
    if drop_path_mask_source is not None:
        return drop_path_mask_source(x.shape[0], keep).to(x.dtype).reshape(shape).clone()
    return x.new_empty(shape).bernoulli_(keep)


def drop_path_d8(xs, drop_prob: float = 0.0, training: bool = False, scale_by_keep: bool = True):
    """d8_layers.py:249-271: ONE Bernoulli mask per sample shared by all five tensors.  The mask is
    drawn with the same call (``new_empty(B,1,1).bernoulli_``) so a seeded run reproduces the reference."""
    if drop_prob == 0.0 or not training:
        return xs
    keep = 1 - drop_prob
    mask = _draw_mask(xs[0], keep)
    if keep > 0.0 and scale_by_keep:
        mask.div_(keep)
    return tuple(x * mask for x in xs[:4]) + (xs[4] * mask.unsqueeze(-1),)


class DropPathD8(nn.Module):
    def __init__(self, drop_prob=0.0, scale_by_keep=True):
        super().__init__()
        self.drop_prob, self.scale_by_keep = drop_prob, scale_by_keep

    def forward(self, xs):
        return drop_path_d8(xs, self.drop_prob, self.training, self.scale_by_keep)


class LiftIrrepD8Conv2d(nn.Module):
    """d8_layers.py:284-382: learn a (p/2)² quarter kernel, expand to p×p by symmetry, strided conv."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, bias, irrep="A1"):
        super().__init__()
        if irrep not in ("A1", "A2", "B1", "B2", "E"):
            raise ValueError("Invalid irrep.")
        if bias and irrep != "A1":
            raise ValueError("Bias only ok for A1-irrep.")
        ks = _pair(kernel_size)
        if ks[0] != ks[1]:
            raise NotImplementedError("Non-square kernels not implemented")
        if ks[0] % 2:
            raise NotImplementedError("Odd kernel sizes not yet implemented")
        if ks[0] == 2 and irrep in ("A2", "B1"):
            raise ValueError(f"No {irrep} irrep in filter kernels of size 2.")
        self.kernel_size, self.stride, self.irrep = ks, stride, irrep
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, ks[0] // 2, ks[1] // 2))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            bound = 1 / math.sqrt(in_channels * (ks[0] // 2) * (ks[1] // 2))
            nn.init.uniform_(self.bias, -bound, bound)

    def expand_weight(self):
        if self.irrep == "E":
            return _unfold_e(0.5 * self.weight, -2, -1)
        return _unfold_1d(SQRT2_OVER_4 * self.weight, self.irrep, -2, -1)

    def forward(self, x):
        k = self.expand_weight()
        y = F.conv2d(x, k, self.bias, stride=self.stride)
        if self.irrep == "E":
            return y, F.conv2d(x, k.rot90(k=1, dims=(-2, -1)), self.bias, stride=self.stride)
        return y


class LiftD8(nn.Module):
    """d8_layers.py:384-411: (A1, A2, B1, B2, E_left→(x4,x5), E_right→(x6,x7))."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, bias):
        super().__init__()
        if out_channels % 8:
            raise ValueError()
        o = out_channels // 8
        mk = lambda irrep, b=False: LiftIrrepD8Conv2d(in_channels, o, kernel_size, stride, bias=b, irrep=irrep)
        self.conv_A1, self.conv_A2, self.conv_B1, self.conv_B2 = mk("A1", bias), mk("A2"), mk("B1"), mk("B2")
        self.conv_E_left, self.conv_E_right = mk("E"), mk("E")

    def forward(self, img):
        return (self.conv_A1(img), self.conv_A2(img), self.conv_B1(img), self.conv_B2(img),
                *self.conv_E_left(img), *self.conv_E_right(img))


class PatchEmbedD8(nn.Module):
    """d8_layers.py:413-497."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, norm_layer=None,
                 flatten=True, bias=True, strict_img_size=True):
        super().__init__()
        self.patch_size = _pair(patch_size)
        if img_size is None:
            self.img_size = self.grid_size = self.num_patches = None
        else:
            self.img_size = _pair(img_size)
            self.grid_size = tuple(s // p for s, p in zip(self.img_size, self.patch_size))
            self.num_patches = self.grid_size[0] * self.grid_size[1]
        if embed_dim % 8:
            raise ValueError()
        self.flatten, self.strict_img_size = flatten, strict_img_size
        self.lift8 = LiftD8(in_chans, embed_dim, kernel_size=self.patch_size, stride=self.patch_size, bias=bias)
        self.norm = norm_layer(embed_dim) if norm_layer else nn.Identity()

    def forward(self, x):
        _, _, H, W = x.shape
        if self.img_size is not None:
            if self.strict_img_size:
                assert H == self.img_size[0], f"Input height ({H}) doesn't match model ({self.img_size[0]})."
                assert W == self.img_size[1], f"Input width ({W}) doesn't match model ({self.img_size[1]})."
            else:
                assert H % (self.patch_size[1] * 2) == 0 and W % (self.patch_size[0] * 2) == 0
        xs = self.lift8(x)
        if self.flatten:
            xs = tuple(t.flatten(2).transpose(1, 2) for t in xs)
        return self.norm(convert_8tuple_to_5tuple(xs))


class IsotypicToPatchD8(nn.Module):
    """d8_layers.py:499-588: inverse of the lift — irrep features back to p×p pixel patches."""

    def __init__(self, dim, patch_side, out_channels=3, bias=True, reshape_to_image=False):
        super().__init__()
        if patch_side % 2:
            raise NotImplementedError("Odd patch side not implemented.")
        self.dim, self.patch_side, self.out_channels = dim, patch_side, out_channels
        self.reshape_to_image = reshape_to_image
        self.lin8 = LinearD8(dim, 2 * (patch_side ** 2 * out_channels), bias=bias)

    def forward(self, xs):
        B, L, _ = xs[0].shape
        h = self.patch_side // 2
        q = [0.25 * t.reshape(B, L, h, h, self.out_channels) for t in convert_5tuple_to_8tuple(self.lin8(xs))]
        out = sum(_unfold_1d(q[i], n, 2, 3) for i, n in enumerate(("A1", "A2", "B1", "B2")))

        def e_img(t):  # :561-567 — note: differs from the lift's E unfold (flip of dim 3 / 180° turn)
            t = SQRT2 * t
            return _quad(t, t.flip(2), -t.flip(3), -t.rot90(k=2, dims=(2, 3)), 2, 3)

        out = out + e_img(q[4]) + e_img(q[5]).rot90(k=1, dims=(2, 3))
        if self.reshape_to_image:
            s = int(math.sqrt(L))
            p = self.patch_side
            out = out.reshape(B, s, s, p, p, self.out_channels).permute(0, 5, 1, 3, 2, 4)
            return out.reshape(B, self.out_channels, s * p, s * p)
        return out.reshape(B, L, self.patch_side ** 2 * self.out_channels)


def pack_heads(qkvs, num_heads):
    """d8_layers.py:631-643: 5-tuple with 3·D channels → q,k,v [B,H,T,D/H]; per-head vector =
    [A1 w | A2 w | B1 w | B2 w | E_row0 2w | E_row1 2w]."""
    B, T, c3 = qkvs[0].shape
    c, H = c3 // 3, num_heads
    w = c // H
    parts = [t.reshape(B, T, 3, H, w) for t in qkvs[:4]]
    e = qkvs[4].reshape(B, T, 2, 3, H, 2 * w)
    parts += [e[:, :, 0], e[:, :, 1]]
    qkv = torch.cat(parts, dim=-1).permute(2, 0, 3, 1, 4)  # [3,B,H,T,8w]
    return qkv[0], qkv[1], qkv[2]


def unpack_heads(o):
    """d8_layers.py:650-656: [B,H,T,8w] → 5-tuple."""
    B, H, T, hd = o.shape
    w = hd // 8
    o = o.transpose(1, 2)  # [B,T,H,8w]
    ones = tuple(o[..., i * w:(i + 1) * w].reshape(B, T, H * w) for i in range(4))
    e = torch.stack((o[..., 4 * w:6 * w].reshape(B, T, 2 * H * w), o[..., 6 * w:].reshape(B, T, 2 * H * w)), dim=2)
    return ones + (e,)


class AttentionD8(nn.Module):
    """d8_layers.py:590-660.  ``scale``/``qk_scale`` are stored but unused; SDPA's default 1/√(D/H) applies."""

    def __init__(self, dim, num_heads=8, qkv_bias=True, proj_bias=True, attn_drop=0.0, proj_drop=0.0,
                 rope=None, qk_scale=None):
        super().__init__()
        assert dim % num_heads == 0, "dim should be divisible by num_heads"
        assert (dim // num_heads) % 8 == 0, "dim should be divisible by 8"
        if rope is not None:
            raise NotImplementedError("RoPE not implemented")
        self.num_heads = num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.qkv = LinearD8(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = LinearD8(dim, dim, bias=proj_bias)
        self.proj_drop = DropoutD8(proj_drop)
        self.rope = rope

    def forward(self, xs):
        q, k, v = pack_heads(self.qkv(xs), self.num_heads)
        o = F.scaled_dot_product_attention(q, k, v, dropout_p=self.attn_drop.p if self.training else 0.0)
        return self.proj_drop(self.proj(unpack_heads(o)))


def _add(xs, ys):
    return tuple(x + y for x, y in zip(xs, ys))


class Layer_scale_init_BlockD8(nn.Module):
    """d8_layers.py:665-707 (DeiT-III block): gamma_{1,2} = AffineD8(bias=False), one shared drop_path."""

    def __init__(self, dim, num_heads, mlp_ratio=4, qkv_bias=False, qk_scale=None, attn_drop=0.0, drop=0.0,
                 drop_path=0.0, act_layer=TritonGeluD8, norm_layer=LayerNormD8, Attention_block=AttentionD8,
                 Mlp_block=MlpD8, init_values=1e-4):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention_block(dim=dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale,
                                    attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = DropPathD8(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp_block(in_features=dim, hidden_features=int(mlp_ratio * dim), act_layer=act_layer, drop=drop)
        self.gamma_1 = AffineD8(dim, bias=False)
        self.gamma_2 = AffineD8(dim, bias=False)
        with torch.no_grad():
            for p in list(self.gamma_1.parameters()) + list(self.gamma_2.parameters()):
                p.fill_(init_values)

    def forward(self, xs):
        xs = _add(xs, self.drop_path(self.gamma_1(self.attn(self.norm1(xs)))))
        return _add(xs, self.drop_path(self.gamma_2(self.mlp(self.norm2(xs)))))


class BlockD8(nn.Module):
    """d8_layers.py:713-776 (DINOv2 / model-default block): ls{1,2} = LayerScaleD8, drop_path{1,2}."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, proj_bias=True, ffn_bias=True, drop=0.0,
                 attn_drop=0.0, init_values=None, drop_path=0.0, act_layer=TritonGeluD8, norm_layer=LayerNormD8,
                 attn_class=AttentionD8, ffn_layer=MlpD8):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = attn_class(dim, num_heads=num_heads, qkv_bias=qkv_bias, proj_bias=proj_bias,
                               attn_drop=attn_drop, proj_drop=drop)
        self.ls1 = LayerScaleD8(dim, init_values=init_values) if init_values else nn.Identity()
        self.drop_path1 = DropPathD8(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = ffn_layer(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer,
                             drop=drop, bias=ffn_bias)
        self.ls2 = LayerScaleD8(dim, init_values=init_values) if init_values else nn.Identity()
        self.drop_path2 = DropPathD8(drop_path) if drop_path > 0.0 else nn.Identity()
        self.sample_drop_ratio = drop_path

    def forward(self, xs):
        xs = _add(xs, self.drop_path1(self.ls1(self.attn(self.norm1(xs)))))
        return _add(xs, self.drop_path2(self.ls2(self.mlp(self.norm2(xs)))))


class NestedTensorBlockD8(BlockD8):
    """d8_layers.py:780-794: a list of crops is just looped over."""

    def forward(self, x_or_x_list):
        if isinstance(x_or_x_list, tuple):
            return super().forward(x_or_x_list)
        if isinstance(x_or_x_list, list):
            return [super(NestedTensorBlockD8, self).forward(x) for x in x_or_x_list]
        raise AssertionError


# ----------------------------------------------------------------------------------------------
# Invariants (octic_vits/d8_invariantization.py)
# ----------------------------------------------------------------------------------------------
class Invariant(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.output_dim = dim


class NonInvariant(Invariant):  # :29-42
    def __init__(self, C):
        super().__init__(C)

    def forward(self, xt):
        return torch.cat([t.abs() for t in convert_5tuple_to_8tuple(xt)], dim=-1)


class LinearInvariant(Invariant):  # :43-48
    def __init__(self, C):
        super().__init__(C // 8)

    def forward(self, xt):
        return xt[0].abs()


class PowerSpectrumInvariant(Invariant):  # :49-64
    def __init__(self, C):
        super().__init__(6 * C // 8)

    def forward(self, xt):
        return torch.cat((xt[0], xt[1].abs(), xt[2].abs(), xt[3].abs(), xt[4].norm(dim=-2)), dim=-1)


def _monomials(xt, third_order):
    """Generators of the polynomial invariant ring used by :66-141 (same ordering)."""
    x0, x1, x2, x3, x4, x5, x6, x7 = convert_5tuple_to_8tuple(xt)
    deg2 = [x6 ** 2 + x7 ** 2, x4 * x6 + x5 * x7, x4 ** 2 + x5 ** 2, x3 ** 2, x2 ** 2, x1 ** 2]
    deg3 = [x3 * x6 * x7, x3 * x5 * x6 + x3 * x4 * x7, x3 * x4 * x5, x2 * x6 ** 2 - x2 * x7 ** 2,
            x2 * x4 * x6 - x2 * x5 * x7, x2 * x4 ** 2 - x2 * x5 ** 2, x1 * x5 * x6 - x1 * x4 * x7, x1 * x2 * x3]
    if third_order:
        return [x0 ** 3] + [x0 * t for t in deg2] + deg3
    deg4 = [x6 ** 4 + x7 ** 4, x4 * x6 ** 3 + x5 * x7 ** 3, x4 ** 2 * x6 ** 2 + x5 ** 2 * x7 ** 2,
            x4 ** 3 * x6 + x5 ** 3 * x7, x4 ** 4 + x5 ** 4,
            x2 * x3 * x5 * x6 - x2 * x3 * x4 * x7, x1 * x3 * x6 ** 2 - x1 * x3 * x7 ** 2,
            x1 * x3 * x4 * x6 - x1 * x3 * x5 * x7, x1 * x3 * x4 ** 2 - x1 * x3 * x5 ** 2,
            x1 * x2 * x6 * x7, x1 * x2 * x5 * x6 + x1 * x2 * x4 * x7, x1 * x2 * x4 * x5,
            x1 * x6 ** 3 * x7 - x1 * x6 * x7 ** 3, x1 * x5 * x6 ** 3 - x1 * x4 * x7 ** 3,
            x1 * x4 * x5 * x6 ** 2 - x1 * x4 * x5 * x7 ** 2, x1 * x4 ** 2 * x5 * x6 - x1 * x4 * x5 ** 2 * x7,
            x1 * x4 ** 3 * x5 - x1 * x4 * x5 ** 3]
    return [x0] + deg2 + deg3 + deg4


class PolynomialInvariant(Invariant):  # :66-112
    def __init__(self, C):
        super().__init__(32 * C // 8)

    def forward(self, xt):
        return torch.cat(_monomials(xt, False), dim=-1)


class ThirdOrderInvariant(Invariant):  # :114-141
    def __init__(self, C):
        super().__init__(15 * C // 8)

    def forward(self, xt):
        return torch.cat(_monomials(xt, True), dim=-1)


def _action_matrices():
    """8×8 matrices of the isotypic action of the 8 group elements in the order
    (e, r, r², r³, m, m r, m r², m r³) used by :160-210 (matrix products ``m @ r^k``)."""
    eye = torch.eye(8)
    r = torch.stack(_iso_r(tuple(eye)), 0)   # rows: images of the coordinate functionals
    m = torch.stack(_iso_m(tuple(eye)), 0)
    mats, p = [], eye
    for _ in range(4):
        mats.append(p)
        p = r @ p
    return mats + [m @ q for q in mats], r, m


class MaxFilteringInvariant(Invariant):  # :142-210
    def __init__(self, input_channels, num_references=None, learnable_references=True, global_avg=False):
        num_references = input_channels * 2 if num_references is None else num_references
        super().__init__(num_references)
        self.references = nn.Parameter(
            F.normalize(torch.randn(num_references, input_channels // 8, 8), dim=(1, 2)),
            requires_grad=learnable_references)
        _, r, m = _action_matrices()
        self.rotation_action = nn.Parameter(r.clone(), requires_grad=False)
        self.reflection_action = nn.Parameter(m.clone(), requires_grad=False)
        self.global_avg = global_avg

    def forward(self, xt):
        x = torch.cat(convert_5tuple_to_8tuple(xt), dim=-1)  # [..., 8c] irrep-major
        mats, _, _ = _action_matrices()
        refs = torch.stack([torch.einsum("ij,dcj->dic", g.to(x), self.references) for g in mats], 0).flatten(-2)
        eq = "kdc,bc->bkd" if self.global_avg else "kdc,bnc->bnkd"
        return torch.einsum(eq, refs, x).max(dim=-2).values


class CanonizationInvariant(Invariant):  # :212-280
    def __init__(self, dim, learnable_reference=True, global_avg=False):
        super().__init__(dim)
        self.reference = nn.Parameter(F.normalize(torch.randn(dim), dim=0), requires_grad=learnable_reference)
        _, r, m = _action_matrices()
        self.rotation_action = nn.Parameter(r.clone(), requires_grad=False)
        self.reflection_action = nn.Parameter(m.clone(), requires_grad=False)
        self.global_avg = global_avg

    def forward(self, xt):
        x = torch.stack(convert_5tuple_to_8tuple(xt), dim=-1)  # [B,N,c,8]
        if self.global_avg:
            x = x.unsqueeze(1)
        mats, _, _ = _action_matrices()
        orbit = torch.stack([torch.einsum("ij,...cj->...ic", g.to(x), x) for g in mats], dim=-3).flatten(-2)
        best = torch.einsum("c,bnkc->bnk", self.reference, orbit).argmax(dim=-1, keepdim=True)
        out = torch.gather(orbit, 2, best.unsqueeze(-1).expand(-1, -1, -1, orbit.shape[-1])).squeeze(-2)
        return out.squeeze(1) if self.global_avg else out


# ----------------------------------------------------------------------------------------------
# Standard half (deit/vit.py:14-134) and timm's default Block (restated, see module docstring)
# ----------------------------------------------------------------------------------------------
class DropPath(nn.Module):
    def __init__(self, drop_prob=0.0, scale_by_keep=True):
        super().__init__()
        self.drop_prob, self.scale_by_keep = drop_prob, scale_by_keep

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1 - self.drop_prob
        mask = _draw_mask(x, keep)
        if keep > 0.0 and self.scale_by_keep:
            mask.div_(keep)
        return x * mask


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0, bias=True):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features, bias=bias)
        self.act = act_layer()
        self.drop1 = nn.Dropout(drop)
        self.norm = nn.Identity()
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features, bias=bias)
        self.drop2 = nn.Dropout(drop)

    def forward(self, x):
        return self.drop2(self.fc2(self.norm(self.drop1(self.act(self.fc1(x))))))


class Attention(nn.Module):
    """deit/vit.py:14-56 (fused SDPA branch)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0.0, proj_drop=0.0, fused_attn=True):
        super().__init__()
        self.num_heads = num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.fused_attn = fused_attn

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        o = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2], dropout_p=self.attn_drop.p if self.training else 0.0)
        return self.proj_drop(self.proj(o.transpose(1, 2).reshape(B, N, C)))


class Layer_scale_init_Block(nn.Module):
    """deit/vit.py:90-134."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, qk_scale=None, drop=0.0, attn_drop=0.0,
                 drop_path=0.0, act_layer=nn.GELU, norm_layer=nn.LayerNorm, Attention_block=Attention, Mlp_block=Mlp,
                 init_values=1e-4, use_fused_attn=True):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention_block(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale,
                                    attn_drop=attn_drop, proj_drop=drop, fused_attn=use_fused_attn)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp_block(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        self.gamma_1 = nn.Parameter(init_values * torch.ones(dim))
        self.gamma_2 = nn.Parameter(init_values * torch.ones(dim))

    def forward(self, x):
        x = x + self.drop_path(self.gamma_1 * self.attn(self.norm1(x)))
        return x + self.drop_path(self.gamma_2 * self.mlp(self.norm2(x)))


class _LayerScale(nn.Module):
    def __init__(self, dim, init_values=1e-5):
        super().__init__()
        self.gamma = nn.Parameter(init_values * torch.ones(dim))

    def forward(self, x):
        return x * self.gamma


class Block(nn.Module):
    """timm 1.0.12 ``vision_transformer.Block`` (dependency absent from /root/reference; restated)."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, qk_norm=False, proj_drop=0.0, attn_drop=0.0,
                 init_values=None, drop_path=0.0, act_layer=nn.GELU, norm_layer=nn.LayerNorm, mlp_layer=Mlp, **_):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, attn_drop=attn_drop, proj_drop=proj_drop)
        self.ls1 = _LayerScale(dim, init_values) if init_values else nn.Identity()
        self.drop_path1 = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = mlp_layer(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=proj_drop)
        self.ls2 = _LayerScale(dim, init_values) if init_values else nn.Identity()
        self.drop_path2 = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()

    def forward(self, x):
        x = x + self.drop_path1(self.ls1(self.attn(self.norm1(x))))
        return x + self.drop_path2(self.ls2(self.mlp(self.norm2(x))))


# ----------------------------------------------------------------------------------------------
# Model (octic_vits/model.py) and DeiT factories (octic_vits/deit_models.py)
# ----------------------------------------------------------------------------------------------
class OcticVisionTransformer(nn.Module):
    """model.py:25-234."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4.0, qkv_bias=False, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.0,
                 octic_block_layers=BlockD8, standard_block_layers=Block, Patch_layer=PatchEmbedD8, init_scale=1e-4,
                 num_register_tokens=0, global_pool=False, invariant=False, octic_equi_break_layer=None, **kwargs):
        super().__init__()
        assert embed_dim % 8 == 0, "embed_dim must be divisible by 8"
        self.dropout_rate, self.global_pool, self.num_classes = drop_rate, global_pool, num_classes
        self.num_features = self.embed_dim = embed_dim
        if octic_equi_break_layer is None:
            assert depth % 2 == 0, "depth must be even"
            octic_equi_break_layer = depth // 2
        else:
            assert octic_equi_break_layer >= 0, "octic_equi_break_layer must be non-negative"
            assert octic_equi_break_layer < depth, "octic_equi_break_layer must be less than depth"
        self.octic_equi_break_layer, self.invariant = octic_equi_break_layer, invariant
        self.num_register_tokens = num_register_tokens
        if invariant:
            self.invariantization = PowerSpectrumInvariant(embed_dim)
            self.invariant_proj = nn.Linear(self.invariantization.output_dim, embed_dim)
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
        self.blocks = nn.ModuleList([
            octic_block_layers(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                               attn_drop=attn_drop_rate, drop_path=drop_path_rate, norm_layer=LayerNormD8,
                               act_layer=TritonGeluD8, init_values=init_scale)
            if i < octic_equi_break_layer else
            standard_block_layers(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                                  attn_drop=attn_drop_rate, drop_path=drop_path_rate, norm_layer=norm_layer,
                                  act_layer=nn.GELU, init_values=init_scale)
            for i in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        std = 8 * 0.02
        if num_register_tokens > 0:
            nn.init.normal_(self.register_tokens[0], std=1e-6)
        for p in self.pos_embed:
            trunc_normal_(p, std=SQRT2_OVER_2 * std)
        if not global_pool:
            trunc_normal_(self.cls_token[0], std=std)
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.weight, 1.0)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)

    def forward_features(self, x):
        B, _, H, W = x.shape
        xs = self.patch_embed(x)
        pos = convert_8tuple_to_5tuple(isotypic_dim_interpolation(tuple(self.pos_embed), dim=0))
        pos = interpolate_spatial_tuple(xs, pos, H, W, self.patch_embed.patch_size)
        xs = tuple(t + p.flatten(0, 1) for t, p in zip(xs, pos))
        if not self.global_pool:
            xs = tuple(torch.cat((self.cls_token[i].expand(B, *self.cls_token[i].shape[1:]), xs[i]), dim=1)
                       for i in range(5))
        if self.num_register_tokens > 0:
            raise RuntimeError("register tokens are broken in the reference (model.py:183-191 loops 8 over a 5-tuple)")
        k = self.octic_equi_break_layer
        for blk in self.blocks[:k]:
            xs = blk(xs)
        if self.invariant:
            x = self.invariant_proj(self.invariantization(xs))
        else:
            x = torch.cat(convert_5tuple_to_8tuple(xs), dim=-1)
        for blk in self.blocks[k:]:
            x = blk(x)
        x = self.norm(x)
        return x.mean(dim=1) if self.global_pool else x[:, 0]

    def forward(self, x):
        x = self.forward_features(x)
        if self.dropout_rate:
            x = F.dropout(x, p=float(self.dropout_rate), training=self.training)
        return self.head(x)

    @torch.jit.ignore
    def no_weight_decay(self):
        base = [f"pos_embed.{i}" for i in range(6)] + ["cls_token.0"]
        return set(base + ["_orig_mod." + n for n in base])


# ----------------------------------------------------------------------------------------------
# DINOv2 entry points (octic_vits/dinov2_models.py) with the standard blocks of dinov2/layers/block.py
# ----------------------------------------------------------------------------------------------
subset_observer = None      # test hook: callable (batch, kept indices) called for every batch-subset draw, in draw order


def drop_add_residual_stochastic_depth(x, residual_func, sample_drop_ratio=0.0):
    """dinov2/layers/block.py:113-140: the residual branch runs on a random subset of the batch and is added back
    scaled by batch / subset (training with drop_path > 0.1)."""
    b = x.shape[0]
    keep = max(int(b * (1 - sample_drop_ratio)), 1)
    brange = torch.randperm(b, device=x.device)[:keep]
    if subset_observer is not None:                  # test hook: sees every subset in the order the reference draws them
        subset_observer(b, brange)
    residual = residual_func(x[brange]).flatten(1)
    out = torch.index_add(x.flatten(1), 0, brange, residual.to(x.dtype), alpha=b / keep)
    return out.view_as(x)


class NestedTensorBlock(nn.Module):
    """dinov2/layers/block.py:43-111, 234-260 (``Block`` / ``NestedTensorBlock``); same parameter names as the timm
    block.  A list input (crops of different resolutions) is processed crop by crop: attention never mixes samples,
    so this equals the reference's packed xformers path, which cannot run without xformers - **parity unpinned**."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, proj_bias=True, ffn_bias=True, drop=0.0,
                 attn_drop=0.0, init_values=None, drop_path=0.0, act_layer=nn.GELU, norm_layer=nn.LayerNorm,
                 attn_class=None, ffn_layer=None, **_):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, attn_drop=attn_drop, proj_drop=drop)
        if not proj_bias:
            self.attn.proj.bias = None
        self.ls1 = _LayerScale(dim, init_values) if init_values else nn.Identity()
        self.drop_path1 = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop,
                       bias=ffn_bias)
        self.ls2 = _LayerScale(dim, init_values) if init_values else nn.Identity()
        self.drop_path2 = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.sample_drop_ratio = drop_path

    def _one(self, x):
        attn = lambda t: self.ls1(self.attn(self.norm1(t)))
        ffn = lambda t: self.ls2(self.mlp(self.norm2(t)))
        if self.training and self.sample_drop_ratio > 0.1:
            x = drop_add_residual_stochastic_depth(x, attn, self.sample_drop_ratio)
            return drop_add_residual_stochastic_depth(x, ffn, self.sample_drop_ratio)
        if self.training and self.sample_drop_ratio > 0.0:
            x = x + self.drop_path1(attn(x))
            return x + self.drop_path1(ffn(x))          # the reference reuses drop_path1 (block.py:104, "FIXME")
        x = x + attn(x)
        return x + ffn(x)

    def forward(self, x_or_x_list):
        if isinstance(x_or_x_list, torch.Tensor):
            return self._one(x_or_x_list)
        if isinstance(x_or_x_list, list):
            return [self._one(x) for x in x_or_x_list]
        raise AssertionError


class OcticDinoVisionTransformer(OcticVisionTransformer):
    """dinov2_models.py:41-267."""

    def __init__(self, img_size=224, patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0,
                 num_register_tokens=0, drop_path_rate=0.0, octic_block_layers=NestedTensorBlockD8,
                 standard_block_layers=NestedTensorBlock, invariant=False, **kwargs):
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
        for p in self.pos_embed:
            trunc_normal_(p, std=std * SQRT2_OVER_2)
        nn.init.normal_(self.cls_token[0], std=1e-6)
        if self.register_tokens is not None:
            nn.init.normal_(self.register_tokens[0], std=1e-6)
        for m in self.modules():                      # named_apply(init_weights_vit_timm): Linear layers only
            if isinstance(m, nn.Linear):
                trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def prepare_tokens_with_masks(self, x, masks=None):
        B, _, h, w = x.shape
        xs = convert_5tuple_to_8tuple(self.patch_embed(x))
        if masks is not None:
            xs = tuple(torch.where(masks.unsqueeze(-1), self.mask_token[i].to(xs[i].dtype).unsqueeze(0), xs[i])
                       for i in range(8))
        pos = isotypic_dim_interpolation(tuple(self.pos_embed), dim=0)
        pos = interpolate_spatial_tuple(xs, pos, h, w, self.patch_embed.patch_size)
        xs = tuple(t + v.flatten(0, 1) for t, v in zip(xs, pos))
        xs = tuple(torch.cat((self.cls_token[i].expand(B, -1, -1), xs[i]), dim=1) for i in range(8))
        if self.register_tokens is not None:
            xs = tuple(torch.cat((xs[i][:, :1], self.register_tokens[i].expand(B, -1, -1), xs[i][:, 1:]), dim=1)
                       for i in range(8))
        return convert_8tuple_to_5tuple(xs)

    def _hand_off(self, xs):
        if self.invariant:
            return self.invariant_proj(self.invariantization(xs))
        return torch.cat(convert_5tuple_to_8tuple(xs), dim=-1)

    def _out(self, x, masks):
        x_norm = self.norm(x)
        r = self.num_register_tokens
        return {"x_norm_clstoken": x_norm[:, 0], "x_norm_regtokens": x_norm[:, 1:r + 1],
                "x_norm_patchtokens": x_norm[:, r + 1:], "x_prenorm": x, "masks": masks}

    def forward_features_list(self, x_list, masks_list):
        xs = [self.prepare_tokens_with_masks(x, m) for x, m in zip(x_list, masks_list)]
        for blk in self.blocks[:self.depth // 2]:
            xs = blk(xs)
        x = [self._hand_off(t) for t in xs]
        for blk in self.blocks[self.depth // 2:]:
            x = blk(x)
        return [self._out(t, m) for t, m in zip(x, masks_list)]

    def forward_features(self, x, masks=None):
        if isinstance(x, list):
            return self.forward_features_list(x, masks)
        xs = self.prepare_tokens_with_masks(x, masks)
        for blk in self.blocks[:self.depth // 2]:
            xs = blk(xs)
        x = self._hand_off(xs)
        for blk in self.blocks[self.depth // 2:]:
            x = blk(x)
        return self._out(x, masks)

    def get_intermediate_layers(self, x, n=1, reshape=False, return_class_token=False, norm=True):
        xs = self.prepare_tokens_with_masks(x)
        total = len(self.blocks)
        take = range(total - n, total) if isinstance(n, int) else n
        assert all(i > self.depth // 2 for i in take), f"All block indices must be > half depth, got {take}"
        for blk in self.blocks[:self.depth // 2]:
            xs = blk(xs)
        t = self._hand_off(xs)
        outputs = []
        for i, blk in enumerate(self.blocks[self.depth // 2:], start=self.depth // 2):
            t = blk(t)
            if i in take:
                outputs.append(t)
        assert len(outputs) == len(take), f"only {len(outputs)} / {len(take)} blocks found"
        if norm:
            outputs = [self.norm(o) for o in outputs]
        cls = [o[:, 0] for o in outputs]
        outputs = [o[:, 1 + self.num_register_tokens:] for o in outputs]
        if reshape:
            B, _, w, h = x.shape
            p = self.patch_embed.patch_size
            p = p[0] if isinstance(p, (tuple, list)) else p      # the reference divides by the tuple here (TypeError)
            outputs = [o.reshape(B, w // p, h // p, -1).permute(0, 3, 1, 2).contiguous() for o in outputs]
        return tuple(zip(outputs, cls)) if return_class_token else tuple(outputs)

    def forward(self, *args, is_training=False, **kwargs):
        ret = self.forward_features(*args, **kwargs)
        return ret if is_training else self.head(ret["x_norm_clstoken"])


_REGISTRY = {}


def register_model(fn):
    _REGISTRY[fn.__name__] = fn
    return fn


def create_model(name, pretrained=False, **kwargs):
    kwargs.pop("drop_block_rate", None)
    return _REGISTRY[name](**kwargs)


def _deit(img_size, patch, dim, depth, heads, invariant, **kw):
    return OcticVisionTransformer(img_size=img_size, patch_size=patch, embed_dim=dim, depth=depth, num_heads=heads,
                                  mlp_ratio=4, qkv_bias=True, standard_block_layers=Layer_scale_init_Block,
                                  octic_block_layers=Layer_scale_init_BlockD8, invariant=invariant, **kw)


@register_model
def hybrid_deit_large_patch16(img_size=224, **kw):  # deit_models.py:11-24
    return _deit(img_size, 16, 1024, 24, 16, False, **kw)


@register_model
def hybrid_deit_huge_patch14(img_size=224, **kw):  # deit_models.py:27-40
    return _deit(img_size, 14, 1280, 32, 16, False, **kw)


@register_model
def d8_inv_early_deit_huge_patch14(img_size=224, **kw):  # deit_models.py:42-56
    return _deit(img_size, 14, 1280, 32, 16, True, **kw)


@register_model
def d8_inv_early_deit_large_patch16(img_size=224, **kw):  # deit_models.py:58-72
    return _deit(img_size, 16, 1024, 24, 16, True, **kw)


def _dinov2(patch_size, dim, depth, heads, invariant, num_register_tokens, kw):
    from functools import partial
    return OcticDinoVisionTransformer(
        patch_size=patch_size, embed_dim=dim, depth=depth, num_heads=heads, mlp_ratio=4, invariant=invariant,
        standard_block_layers=partial(NestedTensorBlock, init_values=1.0e-05),
        octic_block_layers=partial(NestedTensorBlockD8, init_values=1.0e-05),
        num_register_tokens=num_register_tokens, **kw)


@register_model
def hybrid_dinov2_vit_large_patch16(patch_size=16, num_register_tokens=0, **kw):  # dinov2_models.py:269-282
    return _dinov2(patch_size, 1024, 24, 16, False, num_register_tokens, kw)


@register_model
def hybrid_dinov2_vit_huge_patch16(patch_size=16, num_register_tokens=0, **kw):  # dinov2_models.py:284-297
    return _dinov2(patch_size, 1280, 32, 16, False, num_register_tokens, kw)


@register_model
def d8_inv_early_dinov2_vit_large_patch16(patch_size=16, num_register_tokens=0, **kw):  # dinov2_models.py:299-313
    return _dinov2(patch_size, 1024, 24, 16, True, num_register_tokens, kw)


@register_model
def d8_inv_early_dinov2_vit_huge_patch16(patch_size=16, num_register_tokens=0, **kw):  # dinov2_models.py:315-329
    return _dinov2(patch_size, 1280, 32, 16, True, num_register_tokens, kw)
