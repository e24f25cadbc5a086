"""D8 helpers of the product package (reference: octic_vits/d8_utils.py).

Host-side glue only (tiny tensors: group tables, weight/pos-embed symmetric unfolding).  Unlike the
reference's chains of rot90/flip/cat, unfolding here is ONE gather with precomputed closed-form
index/sign tables per (quarter size, irrep) — cached, autograd-friendly, and cheap enough to run every
step for the 6 quarter kernels and 6 positional grids.
"""
import math
from functools import lru_cache

import torch

SQRT2 = math.sqrt(2)
SQRT2_OVER_2 = 0.5 * SQRT2
SQRT2_OVER_4 = 0.5 * SQRT2_OVER_2

group_elements = ("e", "r", "rr", "rrr", "m", "mr", "mrr", "mrrr")
irreps = ("A1", "A2", "B1", "B2", "E11", "E21", "E21", "E22")

# --------------------------------------------------------------------------------------------
# Group actions as (source index, sign) tables, generated from the two generators
#   r: iso (x0,x1,-x2,-x3,-x5,x4,-x7,x6)   reg (x1,x2,x3,x0,x7,x4,x5,x6)      (d8_utils.py:99-109,182-192)
#   m: iso (x0,-x1,x2,-x3,-x4,x5,-x6,x7)   reg (x4,x5,x6,x7,x0,x1,x2,x3)      (d8_utils.py:132-142,215-225)
# "m r^k" = r applied k times, then m.
# --------------------------------------------------------------------------------------------
_GEN = {
    "iso": {"r": ((0, 1, 2, 3, 5, 4, 7, 6), (1, 1, -1, -1, -1, 1, -1, 1)),
            "m": ((0, 1, 2, 3, 4, 5, 6, 7), (1, -1, 1, -1, -1, 1, -1, 1))},
    "reg": {"r": ((1, 2, 3, 0, 7, 4, 5, 6), (1,) * 8),
            "m": ((4, 5, 6, 7, 0, 1, 2, 3), (1,) * 8)},
}


def _compose(first, then):
    """table of: apply `first`, then `then`."""
    (p1, s1), (p2, s2) = first, then
    return tuple(p1[p2[i]] for i in range(8)), tuple(s2[i] * s1[p2[i]] for i in range(8))


def _build_tables(kind):
    ident = (tuple(range(8)), (1,) * 8)
    tabs = {}
    for g in group_elements:
        t = ident
        for _ in range(g.count("r")):
            t = _compose(t, _GEN[kind]["r"])
        if g.startswith("m"):
            t = _compose(t, _GEN[kind]["m"])
        tabs[g] = t
    return tabs


_ISO, _REG = _build_tables("iso"), _build_tables("reg")


def _apply_table(tab, g, xs):
    if g not in tab:
        raise ValueError("Invalid group element")
    perm, sign = tab[g]
    return tuple(xs[p] if s > 0 else -xs[p] for p, s in zip(perm, sign))


def isotypic_group_action(group_element, xs):
    return _apply_table(_ISO, group_element, xs)


def regular_group_action(group_element, xs):
    return _apply_table(_REG, group_element, xs)


def image_space_group_action(group_element, img):
    if group_element not in group_elements:
        raise ValueError("Invalid group element")
    k = group_element.count("r")
    if k:
        img = img.rot90(k=k, dims=(-2, -1))
    return img.flip(-1) if group_element.startswith("m") else img


def spatial_and_isotypic_group_action(group_element, xs):
    B, L, C = xs[0].shape
    H = W = int(math.sqrt(L))
    return isotypic_group_action(group_element, tuple(
        image_space_group_action(group_element, x.transpose(1, 2).reshape(B, C, H, W)).flatten(2).transpose(1, 2)
        for x in xs))


def _mult_table():
    inv = {v[0]: k for k, v in _REG.items()}
    return [[g1, g2, inv[_compose(_REG[g2], _REG[g1])[0]]] for g1 in group_elements[1:] for g2 in group_elements[1:]]


mult_table = _mult_table()

# --------------------------------------------------------------------------------------------
# Fourier transform between isotypic and regular coordinates (d8_utils.py:276-356)
# --------------------------------------------------------------------------------------------
_S_ISO2REG = (
    (1, 1, 1, 1, 1, 1, 1, -1), (1, 1, -1, -1, 1, -1, -1, -1), (1, 1, 1, 1, -1, -1, -1, 1), (1, 1, -1, -1, -1, 1, 1, 1),
    (1, -1, 1, -1, -1, 1, -1, -1), (1, -1, -1, 1, -1, -1, 1, -1), (1, -1, 1, -1, 1, -1, 1, 1), (1, -1, -1, 1, 1, 1, -1, 1))


def _signed_sum(rows, xs):
    out = []
    for row in rows:
        acc = None
        for s, x in zip(row, xs):
            acc = (x if s > 0 else -x) if acc is None else (acc + x if s > 0 else acc - x)
        out.append(SQRT2_OVER_4 * acc)
    return tuple(out)


def isotypic_to_regular_D8(xs):
    return _signed_sum(_S_ISO2REG, xs)


def regular_to_isotypic_D8(xs):
    return _signed_sum(tuple(zip(*_S_ISO2REG)), xs)


def convert_8tuple_to_5tuple(xs):
    return (xs[0], xs[1], xs[2], xs[3],
            torch.stack((torch.cat((xs[4], xs[6]), dim=-1), torch.cat((xs[5], xs[7]), dim=-1)), dim=-2))


def convert_5tuple_to_8tuple(xs):
    e = xs[4]
    c = e.shape[-1] // 2
    return (xs[0], xs[1], xs[2], xs[3], e[..., 0, :c], e[..., 1, :c], e[..., 0, c:], e[..., 1, c:])


# --------------------------------------------------------------------------------------------
# Symmetric unfolding of a quarter [h,h] to the full [2h,2h] grid as gather tables.
# With P = 2h and W the quadrant assembly (top-left q, bottom-left s*rot90(q), top-right s*rot270(q),
# bottom-right rot180(q); s = -1 for B1/B2):
#     W[i,j] = q[i,j] | s q[j,P-1-i] | s q[P-1-j,i] | q[P-1-i,P-1-j]       (by quadrant)
#     A1,B1: full = W + W[:, ::-1]      A2,B2: full = W - W[:, ::-1]          (d8_layers.py:338-373)
#     E:     full[i,j] = t q[i',j'],  i' = i or P-1-i,  j' = j or P-1-j,  t = -1 on the right half
# --------------------------------------------------------------------------------------------
@lru_cache(maxsize=None)
def _unfold_tables(h: int, kind: str):
    P = 2 * h
    i = torch.arange(P).view(P, 1).expand(P, P)
    j = torch.arange(P).view(1, P).expand(P, P)

    def quadrant(i, j, s):
        top, left = i < h, j < h
        si = torch.where(top & left, i, torch.where(~top & left, j, torch.where(top & ~left, P - 1 - j, P - 1 - i)))
        sj = torch.where(top & left, j, torch.where(~top & left, P - 1 - i, torch.where(top & ~left, i, P - 1 - j)))
        sg = torch.where(top ^ left, torch.full_like(i, s), torch.ones_like(i))
        return si * h + sj, sg

    if kind == "E":
        si = torch.where(i < h, i, P - 1 - i)
        sj = torch.where(j < h, j, P - 1 - j)
        sg = torch.where(j < h, torch.ones_like(i), -torch.ones_like(i))
        return (si * h + sj).flatten(), sg.flatten().float(), None, None
    s = -1 if kind in ("B1", "B2") else 1
    ia, sa = quadrant(i, j, s)
    ib, sb = quadrant(i, P - 1 - j, s)
    if kind in ("A2", "B2"):
        sb = -sb
    return ia.flatten(), sa.flatten().float(), ib.flatten(), sb.flatten().float()


@lru_cache(maxsize=None)
def _unfold_tables_on(h: int, kind: str, device: str, dtype):
    """The tables resident on `device` (uploaded once: a training step does no host-to-device copies, which also
    keeps it capturable in a hipGraph)."""
    with torch.inference_mode(False):      # cached: must be ordinary tensors even if the first caller is in inference mode
        ia, sa, ib, sb = _unfold_tables(h, kind)
        dev = torch.device(device)
        t = lambda x, *a: None if x is None else x.to(dev, *a).clone()
        return t(ia), t(sa, dtype), t(ib), t(sb, dtype)


@lru_cache(maxsize=None)
def _fold_tables_on(h: int, kind: str, device: str, dtype):
    """Inverse of the unfold tables: for every quarter entry the (2h)^2-grid positions that read it, as a dense
    [h*h, n] index table per gather (every quarter entry is read the same number of times: 4 quadrants), with the
    signs.  The backward of the unfold is then a GATHER over these tables - a fixed-order sum, bitwise reproducible -
    instead of ATen's index_select backward (index_add_ with float atomics, whose summation order changes from run
    to run and leaks into every tensor of a LAMB step through the global gradient norm)."""
    with torch.inference_mode(False):
        ia, sa, ib, sb = _unfold_tables(h, kind)
        dev = torch.device(device)
        out = []
        for idx, sg in ((ia, sa), (ib, sb)):
            if idx is None:
                out += [None, None]
                continue
            order = torch.argsort(idx, stable=True)
            counts = torch.bincount(idx, minlength=h * h)
            n = int(counts[0])
            assert bool((counts == n).all()), "unfold table is not uniform"
            out += [order.view(h * h, n).to(dev).clone(), sg[order].view(h * h, n).to(dev, dtype).clone()]
        return tuple(out)


class _GatherUnfoldFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flat, dim, h, kind):
        ia, sa, ib, sb = _unfold_tables_on(h, kind, str(flat.device), flat.dtype)
        shape = [1] * flat.dim()
        shape[dim] = -1
        out = flat.index_select(dim, ia) * sa.view(shape)
        if ib is not None:
            out = out + flat.index_select(dim, ib) * sb.view(shape)
        ctx.meta = (dim, h, kind)
        return out

    @staticmethod
    def backward(ctx, g):
        dim, h, kind = ctx.meta
        pa, ga, pb, gb = _fold_tables_on(h, kind, str(g.device), g.dtype)
        n = pa.shape[1]
        shape = [1] * (g.dim() + 1)
        shape[dim], shape[dim + 1] = h * h, n
        gf = (g.index_select(dim, pa.flatten()).unflatten(dim, (h * h, n)) * ga.view(shape)).sum(dim + 1)
        if pb is not None:
            gf = gf + (g.index_select(dim, pb.flatten()).unflatten(dim, (h * h, n)) * gb.view(shape)).sum(dim + 1)
        return gf, None, None, None


def _gather_unfold(flat, dim, h, kind):
    """flat: tensor whose dimension `dim` enumerates the h*h quarter; returns it with (2h)^2 entries."""
    return _GatherUnfoldFn.apply(flat, dim, h, kind)


def unfold_quarter(q, kind, d0, d1):
    """Unfold dims (d0, d1=d0+1) of q from [h,h] to [2h,2h] (no scaling)."""
    d0 = d0 % q.dim()
    assert d1 % q.dim() == d0 + 1 and q.shape[d0] == q.shape[d0 + 1]
    h = q.shape[d0]
    full = _gather_unfold(q.flatten(d0, d0 + 1), d0, h, kind)
    return full.unflatten(d0, (2 * h, 2 * h))


def unfold_quarter_e_img(t):
    """IsotypicToPatchD8's E assembly over dims (2,3) (d8_layers.py:561-567) — the same table as the lift's E."""
    return unfold_quarter(t, "E", 2, 3)


def expand_lift_kernel(weight, irrep):
    """LiftIrrepD8Conv2d.expand_weight (d8_layers.py:329-373): [o,i,h,h] -> [o,i,2h,2h]."""
    scale = 0.5 if irrep == "E" else SQRT2_OVER_4
    return unfold_quarter(scale * weight, irrep, -2, -1)


# --------------------------------------------------------------------------------------------
# The lift's whole kernel matrix in one gather.  LiftD8.packed_weight (d8_layers.py:384-411 + 329-381 of the
# reference) = six expand_lift_kernel calls, two rot90s, eight flattens and a cat: ~55 launches of a few
# microseconds forward and as many backward, every step.  All of it is one signed gather from the concatenated
# quarter kernels: W[row, col] = P[idx1] * s1 + P[idx2] * s2 (second term: the one-dimensional irreps), and its
# backward a gather over the inverse table (every quarter entry is read 8 times: fixed-order sum, reproducible).
# Same products and the same order of additions as the composed version: bitwise equal forward.
# --------------------------------------------------------------------------------------------
_LIFT_BLOCKS = ((0, "A1", 0), (1, "A2", 0), (2, "B1", 0), (3, "B2", 0), (4, "E", 0), (5, "E", 0), (4, "E", 1), (5, "E", 1))


@torch.compiler.disable          # index tables built once per shape with nonzero / bincount: not for Dynamo to trace
@lru_cache(maxsize=None)
def _lift_tables_on(O: int, Cin: int, h: int, device: str, dtype):
    with torch.inference_mode(False):
        P2, Q = (2 * h) * (2 * h), h * h
        rot = torch.arange(P2).view(2 * h, 2 * h).rot90(1, (0, 1)).flatten()     # source position of every rotated position
        base = (torch.arange(O * Cin) * Q).view(O, Cin, 1)
        i1, s1, i2, s2 = [], [], [], []
        for conv, kind, r in _LIFT_BLOCKS:
            ia, sa, ib, sb = _unfold_tables(h, kind)
            scale = 0.5 if kind == "E" else SQRT2_OVER_4      # factors are built in float64 and cast to `dtype` at the end:
            if r:                                                # an f32 table would cost a float64 caller 1e-9 (advisor, r3)
                ia, sa = ia[rot], sa[rot]
            off = conv * O * Cin * Q
            i1.append((off + base + ia.view(1, 1, P2)).reshape(-1))
            s1.append((scale * sa.double()).view(1, 1, P2).expand(O, Cin, P2).reshape(-1))
            if ib is None:
                i2.append(torch.zeros(O * Cin * P2, dtype=torch.long))
                s2.append(torch.zeros(O * Cin * P2, dtype=torch.float64))
            else:
                i2.append((off + base + ib.view(1, 1, P2)).reshape(-1))
                s2.append((scale * sb.double()).view(1, 1, P2).expand(O, Cin, P2).reshape(-1))
        i1, s1, i2, s2 = torch.cat(i1), torch.cat(s1), torch.cat(i2), torch.cat(s2)
        # inverse: for every parameter entry the output positions that read it (either term), with their factors
        used2 = s2 != 0
        src = torch.cat([i1, i2[used2]])
        pos = torch.cat([torch.arange(i1.numel()), torch.arange(i2.numel())[used2]])
        fac = torch.cat([s1, s2[used2]])
        order = torch.argsort(src, stable=True)
        counts = torch.bincount(src, minlength=6 * O * Cin * Q)
        n = int(counts[0])
        assert bool((counts == n).all()), "lift table is not uniform"
        dev = torch.device(device)
        return (i1.to(dev), s1.to(dev, dtype), i2.to(dev), s2.to(dev, dtype),
                pos[order].view(-1, n).to(dev).clone(), fac[order].view(-1, n).to(dev, dtype).clone())


class _LiftWeightFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *ws):
        O, Cin, h, _ = ws[0].shape
        i1, s1, i2, s2, _, _ = _lift_tables_on(O, Cin, h, str(ws[0].device), ws[0].dtype)
        flat = torch.cat([w.reshape(-1) for w in ws])
        out = flat.index_select(0, i1) * s1 + flat.index_select(0, i2) * s2
        ctx.meta = (O, Cin, h)
        return out.view(8 * O, Cin * 4 * h * h)

    @staticmethod
    def backward(ctx, g):
        O, Cin, h = ctx.meta
        _, _, _, _, inv, fac = _lift_tables_on(O, Cin, h, str(g.device), g.dtype)
        gp = (g.reshape(-1).index_select(0, inv.reshape(-1)).view(inv.shape) * fac).sum(1)
        return tuple(gp.view(6, O, Cin, h, h).unbind(0))


def packed_lift_weight(ws):
    """[8 O, Cin (2h)^2] kernel matrix of LiftD8 (rows A1|A2|B1|B2|E_left K|E_right K|E_left rot K|E_right rot K) from the
    six quarter kernels [O, Cin, h, h] (order A1, A2, B1, B2, E_left, E_right)."""
    return _LiftWeightFn.apply(*ws)


def isotypic_dim_interpolation(xs, dim: int = 0):
    """d8_utils.py:388-451: 6 quarter grids [.., G/2, G/2, c] -> 8 full grids [.., G, G, c]."""
    d0, d1 = dim, dim + 1
    el, er = unfold_quarter(xs[4], "E", d0, d1), unfold_quarter(xs[5], "E", d0, d1)
    return (unfold_quarter(xs[0], "A1", d0, d1), unfold_quarter(xs[1], "A2", d0, d1),
            unfold_quarter(xs[2], "B1", d0, d1), unfold_quarter(xs[3], "B2", d0, d1),
            el, el.rot90(dims=(d0, d1)), er, er.rot90(dims=(d0, d1)))


@torch.compiler.disable          # index tables built once per shape with nonzero / bincount: not for Dynamo to trace
@lru_cache(maxsize=None)
def _pos_tables_on(h: int, device: str, dtype):
    """Row tables of packed_pos_embed: output row block b (A1|A2|B1|B2|E_left|E_right|rot E_left|rot E_right) x grid
    position -> row of the concatenated quarter grids [6 h^2, c], two terms, no scaling; and the inverse table."""
    with torch.inference_mode(False):
        P2, Q = 4 * h * h, h * h
        rot = torch.arange(P2).view(2 * h, 2 * h).rot90(1, (0, 1)).flatten()
        i1, s1, i2, s2 = [], [], [], []
        for conv, kind, r in _LIFT_BLOCKS:
            ia, sa, ib, sb = _unfold_tables(h, kind)
            if r:
                ia, sa = ia[rot], sa[rot]
            i1.append(conv * Q + ia)
            s1.append(sa)
            i2.append(torch.zeros(P2, dtype=torch.long) if ib is None else conv * Q + ib)
            s2.append(torch.zeros(P2) if ib is None else sb)
        i1, s1, i2, s2 = torch.cat(i1), torch.cat(s1), torch.cat(i2), torch.cat(s2)
        used2 = s2 != 0
        src = torch.cat([i1, i2[used2]])
        pos = torch.cat([torch.arange(i1.numel()), torch.arange(i2.numel())[used2]])
        fac = torch.cat([s1, s2[used2]])
        order = torch.argsort(src, stable=True)
        counts = torch.bincount(src, minlength=6 * Q)
        n = int(counts[0])
        assert bool((counts == n).all()), "pos-embed table is not uniform"
        dev = torch.device(device)
        return (i1.to(dev), s1.to(dev, dtype).view(-1, 1), i2.to(dev), s2.to(dev, dtype).view(-1, 1),
                pos[order].view(-1, n).to(dev).clone(), fac[order].view(-1, n, 1).to(dev, dtype).clone())


class _PosEmbedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *ps):
        h, _, c = ps[0].shape
        i1, s1, i2, s2, _, _ = _pos_tables_on(h, str(ps[0].device), ps[0].dtype)
        flat = torch.cat([p.reshape(h * h, c) for p in ps])                       # [6 h^2, c]
        out = flat.index_select(0, i1) * s1 + flat.index_select(0, i2) * s2         # [8 G^2, c], block-major
        ctx.meta = (h, c)
        return out.view(8, 4 * h * h, c).permute(1, 0, 2).reshape(4 * h * h, 8 * c)

    @staticmethod
    def backward(ctx, g):
        h, c = ctx.meta
        _, _, _, _, inv, fac = _pos_tables_on(h, str(g.device), g.dtype)
        gb = g.view(4 * h * h, 8, c).permute(1, 0, 2).reshape(8 * 4 * h * h, c)
        gp = (gb.index_select(0, inv.reshape(-1)).view(inv.shape[0], inv.shape[1], c) * fac).sum(1)
        return tuple(gp.view(6, h, h, c).unbind(0))


def packed_pos_embed_composed(pos_params):
    """The reference's composition (d8_utils.py:388-451): eight unfolded grids, packed order A1|A2|B1|B2|x4|x6|x5|x7."""
    a = isotypic_dim_interpolation(tuple(pos_params), dim=0)
    return torch.cat([t.flatten(0, 1) for t in (a[0], a[1], a[2], a[3], a[4], a[6], a[5], a[7])], dim=-1)


def packed_pos_embed(pos_params):
    """Unfolded positional embedding as packed rows [G*G, 8c] (order A1|A2|B1|B2|x4|x6|x5|x7): one signed row gather from
    the six quarter grids (same products and order of additions as the composition; ~25 launches fewer each way)."""
    ps = tuple(pos_params)
    if len(ps) == 6 and all(p.dim() == 3 and p.shape == ps[0].shape and p.dtype == ps[0].dtype and p.device == ps[0].device
                            and p.shape[0] == p.shape[1] for p in ps):
        return _PosEmbedFn.apply(*ps)
    return packed_pos_embed_composed(ps)


def interpolate_spatial_tuple(xs, interpolant, h: int, w: int, patch_size):
    """d8_utils.py:453-499.  Native resolution: returned untouched.  (The reference's resize branch raises
    TypeError as shipped because patch_size is a tuple; the integer-side intent is implemented.)"""
    n_native = interpolant[0].shape[0] ** 2
    if xs[0].shape[1] == n_native and w == h:
        return interpolant
    p = patch_size[0] if isinstance(patch_size, (tuple, list)) else patch_size
    stacked = torch.stack([t.float().reshape(t.shape[0], t.shape[1], -1) for t in interpolant], 0)
    out = torch.nn.functional.interpolate(stacked.permute(0, 3, 1, 2), size=(h // p, w // p), mode="bicubic",
                                          antialias=False).permute(0, 2, 3, 1)
    return [out[i].reshape(out.shape[1], out.shape[2], *interpolant[i].shape[2:]).to(xs[0].dtype)
            for i in range(len(interpolant))]
