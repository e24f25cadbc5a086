"""Parity cases for the DINOv2 entry points (octic_vits/dinov2_models.py: OcticDinoVisionTransformer).

``run_dino_case(ns, name)`` drives the real reference (golden generation), the oracle and the HIP-backed product
through the same calls.  ``ns`` exposes ``OcticDinoVisionTransformer``, ``NestedTensorBlockD8`` and ``DinoBlock``
(the standard block class the reference takes from dinov2.layers).  All cases run in eval mode at the native
resolution (the only one the reference can run)."""
from functools import partial

import numpy as np
import torch

import cases

SPEC = dict(img_size=32, patch_size=4, embed_dim=64, depth=4, num_heads=2)
DINO_CASES = {
    "dino_hybrid": dict(kw=dict(num_register_tokens=0, invariant=False)),
    "dino_hybrid_reg": dict(kw=dict(num_register_tokens=2, invariant=False)),
    "dino_invariant": dict(kw=dict(num_register_tokens=0, invariant=True)),
}


def build(ns, kw):
    m = ns.OcticDinoVisionTransformer(
        **SPEC, octic_block_layers=partial(ns.NestedTensorBlockD8, init_values=1e-5),
        standard_block_layers=partial(ns.DinoBlock, init_values=1e-5), **kw)
    cases.fill_parameters(m)
    return m.eval()


def run_dino_case(ns, name, device="cpu", to_module=None):
    case = DINO_CASES[name]
    m = build(ns, case["kw"]).to(device)
    if to_module is not None:
        m = to_module(m)
    B, G = 2, SPEC["img_size"] // SPEC["patch_size"]
    x = cases.randn(name + ".img", B, 3, SPEC["img_size"], SPEC["img_size"]).to(device)
    masks = (cases.randn(name + ".mask", B, G * G) > 0.6).to(device)
    res = {}
    with torch.no_grad():
        res["forward"] = m(x).float().cpu().numpy()
        inter = m.get_intermediate_layers(x, n=[3], return_class_token=True)
        res["inter.patch"] = inter[0][0].float().cpu().numpy()
        res["inter.cls"] = inter[0][1].float().cpu().numpy()
    out = m.forward_features(x, masks)
    for k in ("x_norm_clstoken", "x_norm_regtokens", "x_norm_patchtokens", "x_prenorm"):
        res["ff." + k] = out[k].detach().float().cpu().numpy()
    loss = (out["x_norm_patchtokens"].float() * cases.randn(name + ".cot.p", *out["x_norm_patchtokens"].shape).to(device)).sum() \
        + (out["x_norm_clstoken"].float() * cases.randn(name + ".cot.c", *out["x_norm_clstoken"].shape).to(device)).sum()
    loss.backward()
    for pname, p in m.named_parameters():
        if p.grad is None:
            continue
        g = p.grad.detach().float().cpu().flatten()
        res["gnorm." + pname] = np.array([float(g.norm())], dtype=np.float32)
        res["gsample." + pname] = g[:: max(1, g.numel() // 64)][:64].numpy()
    return res
