"""Invariant maps (reference: octic_vits/d8_invariantization.py).

PowerSpectrumInvariant — the one the models use (model.py:89-91) — is a HIP kernel (fwd + bwd).
The other maps are only exercised by the equivariance tests; they are table-driven composites:
polynomial invariants are evaluated from a compact monomial table ("356+347" = x3*x5*x6 + x3*x4*x7),
orbit-based ones from the group-action tables of d8_utils.
"""
import re

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as OF
from .d8_utils import _ISO, convert_5tuple_to_8tuple, group_elements
from .functional import as_packed


class Invariant(nn.Module):
    """Base class for invariants (d8_invariantization.py:12-18)."""

    def __init__(self, dim: int):
        super().__init__()
        self.output_dim = dim


def invariant_head_factory(invariant: Invariant, C: int, num_classes: int, norm=False):
    """d8_invariantization.py:20-28."""
    return nn.Sequential(
        nn.LayerNorm(invariant.output_dim, eps=1e-6) if norm else nn.Identity(),
        nn.Linear(invariant.output_dim, C), nn.GELU(),
        nn.Linear(C, num_classes) if num_classes > 0 else nn.Identity())


class NonInvariant(Invariant):  # :29-42
    def __init__(self, C: int):
        super().__init__(C)

    def forward(self, xtuple):
        return torch.cat([x.abs() for x in convert_5tuple_to_8tuple(xtuple)], dim=-1)


class LinearInvariant(Invariant):  # :43-48
    def __init__(self, C: int):
        super().__init__(C // 8)

    def forward(self, xtuple):
        return torch.abs(xtuple[0])


class PowerSpectrumInvariant(Invariant):  # :49-64 -> octic_power_spectrum_fwd/bwd
    def __init__(self, C: int):
        super().__init__(6 * C // 8)

    def forward(self, xtuple, _out_dtype=None):
        xp, c = as_packed(xtuple)
        return OF.PowerSpectrumFn.apply(xp, c, _out_dtype or xp.dtype)


# monomial tables: digits are 8-tuple indices, repeated digits are powers
_DEG2 = ["66+77", "46+57", "44+55", "33", "22", "11"]
_DEG3 = ["367", "356+347", "345", "266-277", "246-257", "244-255", "156-147", "123"]
_DEG4 = ["6666+7777", "4666+5777", "4466+5577", "4446+5557", "4444+5555", "2356-2347", "1366-1377", "1346-1357",
         "1344-1355", "1267", "1256+1247", "1245", "16667-16777", "15666-14777", "14566-14577", "14456-14557",
         "14445-14555"]


def _poly(spec, x, prefix=""):
    out = None
    for sign, digits in re.findall(r"([+-]?)(\d+)", spec):
        term = None
        for d in prefix + digits:
            term = x[int(d)] if term is None else term * x[int(d)]
        out = (term if sign != "-" else -term) if out is None else (out + term if sign != "-" else out - term)
    return out


class PolynomialInvariant(Invariant):  # :66-112
    def __init__(self, C: int):
        super().__init__(32 * C // 8)

    def forward(self, xtuple):
        x = convert_5tuple_to_8tuple(xtuple)
        return torch.cat([x[0]] + [_poly(s, x) for s in _DEG2 + _DEG3 + _DEG4], dim=-1)


class ThirdOrderInvariant(Invariant):  # :114-141
    def __init__(self, C: int):
        super().__init__(15 * C // 8)

    def forward(self, xtuple):
        x = convert_5tuple_to_8tuple(xtuple)
        return torch.cat([x[0] ** 3] + [_poly(s, x, prefix="0") for s in _DEG2] + [_poly(s, x) for s in _DEG3], dim=-1)


def _action_matrix(g):
    perm, sign = _ISO[g]
    m = torch.zeros(8, 8)
    for i in range(8):
        m[i, perm[i]] = sign[i]
    return m


def _orbit_matrices():
    return [_action_matrix(g) for g in group_elements]  # e, r, rr, rrr, m, mr, mrr, mrrr  (:186-199)


class MaxFilteringInvariant(Invariant):  # :142-210
    def __init__(self, input_channels: int, num_references=None, learnable_references: bool = True,
                 global_avg: bool = False):
        if num_references is None:
            num_references = input_channels * 2
        super().__init__(num_references)
        self.references = nn.Parameter(F.normalize(torch.randn(num_references, input_channels // 8, 8), dim=(1, 2)),
                                       requires_grad=learnable_references)
        self.rotation_action = nn.Parameter(_action_matrix("r"), requires_grad=False)
        self.reflection_action = nn.Parameter(_action_matrix("m"), requires_grad=False)
        self.global_avg = global_avg

    def expand_references_d8(self):
        mats = torch.stack(_orbit_matrices()).to(self.references)          # [8 g, 8, 8]
        return torch.einsum("gij,dcj->gdic", mats, self.references).flatten(start_dim=-2)

    def forward(self, xtuple):
        x = torch.cat(convert_5tuple_to_8tuple(xtuple), dim=-1)
        refs = self.expand_references_d8()
        prod = torch.einsum("kdc,bc->bkd", refs, x) if self.global_avg else torch.einsum("kdc,bnc->bnkd", refs, x)
        return prod.max(dim=-2).values


class CanonizationInvariant(Invariant):  # :212-280
    def __init__(self, dim, learnable_reference=True, global_avg=False):
        super().__init__(dim)
        self.reference = nn.Parameter(F.normalize(torch.randn(dim), dim=0), requires_grad=learnable_reference)
        self.rotation_action = nn.Parameter(_action_matrix("r"), requires_grad=False)
        self.reflection_action = nn.Parameter(_action_matrix("m"), requires_grad=False)
        self.global_avg = global_avg

    def expand_x_d8(self, x):
        mats = torch.stack(_orbit_matrices()).to(x)
        return torch.einsum("gij,...cj->...gic", mats, x).flatten(start_dim=-2)

    def forward(self, xtuple):
        x = torch.stack(convert_5tuple_to_8tuple(xtuple), dim=-1)
        if self.global_avg:
            x = x.unsqueeze(1)
        orbit = self.expand_x_d8(x)
        best = torch.einsum("c,bnkc->bnk", self.reference, orbit).argmax(dim=-1, keepdim=True)
        out = torch.gather(orbit, 2, best.unsqueeze(-1).expand(-1, -1, -1, orbit.shape[-1])).squeeze(-2)
        return out.squeeze(1) if self.global_avg else out
