"""Parity cases shared by the golden-vector generator and the tests.

A *case* names a reference class (or function), its constructor kwargs, and a
seeded input recipe.  ``run_case(ns, case)`` executes it against any namespace
``ns`` exposing the reference's public names (the real reference imported on
CPU, the oracle, or the HIP-backed product package) and returns a flat
``{name: ndarray}`` dict: outputs, input gradients and parameter gradients.
Because all three implementations keep the reference's class names, ctor
signatures and ``state_dict`` keys, the same runner drives all
of them and the golden files hold only data.

Parameters are filled deterministically from the parameter *name* (not from
the RNG stream of the constructor) so that implementations whose constructors
draw random numbers in a different order still get identical weights.
"""
import zlib

import numpy as np
import torch

# --------------------------------------------------------------------------- helpers


def _gen(tag: str) -> torch.Generator:
    return torch.Generator().manual_seed(zlib.crc32(tag.encode()) & 0x7FFFFFFF)


def randn(tag, *shape):
    return torch.randn(*shape, generator=_gen(tag), dtype=torch.float32)


def fill_parameters(module: torch.nn.Module, salt: str = ""):
    """Deterministic, name-keyed parameter values (trainable params only)."""
    with torch.no_grad():
        for name, p in module.named_parameters():
            if not p.requires_grad:
                continue  # e.g. cls_token.1-4 stay frozen zeros (reference model.py:99-105)
            r = randn(salt + name, *p.shape)
            leaf = name.split(".")[-1]
            if name.startswith("pos_embed") or name.startswith("cls_token"):
                v = 0.5 * r
            elif "gamma" in name or name.split(".")[-2:-1] in (["ls1"], ["ls2"], ["gamma_1"], ["gamma_2"]):
                v = 0.5 + 0.2 * r  # layer-scale: O(1) so the branch matters in parity checks
            elif leaf.startswith("alpha"):
                v = 1.0 + 0.2 * r
            elif leaf in ("beta", "bias"):
                v = 0.1 * r
            elif leaf == "weight" and p.ndim == 1:  # nn.LayerNorm weight
                v = 1.0 + 0.2 * r
            elif p.ndim >= 2:
                fan_in = max(1, int(np.prod(p.shape[1:])))
                v = r / np.sqrt(fan_in)
            else:
                v = 0.5 * r
            p.copy_(v.to(p.dtype))
    return module


def tuple5(tag, B, T, c, offset=True):
    """5-tuple (A1,A2,B1,B2:[B,T,c]; E:[B,T,2,2c]) with a per-row offset like
    test_equivariance.py:124-127 so LayerNorm means are exercised."""
    xs = []
    for i in range(4):
        x = randn(f"{tag}.{i}", B, T, c)
        if offset:
            x = x + randn(f"{tag}.{i}.off", B, T, 1)
        xs.append(x)
    e = randn(f"{tag}.4", B, T, 2, 2 * c)
    if offset:
        e = e + randn(f"{tag}.4.off", B, T, 2, 1)
    xs.append(e)
    return tuple(xs)


def tuple8(tag, B, T, c):
    return tuple(randn(f"{tag}.{i}", B, T, c) for i in range(8))


def _flatten_out(out):
    if isinstance(out, torch.Tensor):
        return [out]
    return list(out)


# --------------------------------------------------------------------------- case table

B, T, D = 2, 5, 64
C = D // 8

CASES = {
    # name: (kind, target, ctor kwargs, input recipe)
    "linear_bias": dict(cls="LinearD8", kw=dict(input_channels=64, output_channels=128, bias=True), inp=("tuple5", 2, 5, 8)),
    "linear_nobias": dict(cls="LinearD8", kw=dict(input_channels=128, output_channels=64, bias=False), inp=("tuple5", 2, 5, 16)),
    "affine": dict(cls="AffineD8", kw=dict(dim=64, bias=True), inp=("tuple5", 2, 5, 8)),
    "layerscale": dict(cls="LayerScaleD8", kw=dict(dim=64, init_values=0.1), inp=("tuple5", 2, 5, 8)),
    "layernorm": dict(cls="LayerNormD8", kw=dict(channels=64), inp=("tuple5", 2, 5, 8)),
    "layernorm_128": dict(cls="LayerNormD8", kw=dict(channels=128, eps=1e-6), inp=("tuple5", 3, 7, 16)),
    "gelu": dict(cls="TritonGeluD8", kw=dict(), inp=("tuple5", 2, 5, 8)),
    "mlp": dict(cls="MlpD8", kw=dict(in_features=64, hidden_features=256), inp=("tuple5", 2, 5, 8)),
    "attention": dict(cls="AttentionD8", kw=dict(dim=64, num_heads=2, qkv_bias=True), inp=("tuple5", 2, 17, 8)),
    "attention_h4": dict(cls="AttentionD8", kw=dict(dim=128, num_heads=4, qkv_bias=False), inp=("tuple5", 2, 9, 16)),
    "block_deit": dict(cls="Layer_scale_init_BlockD8", kw=dict(dim=64, num_heads=2, qkv_bias=True, init_values=0.1), inp=("tuple5", 2, 17, 8)),
    "block_deit_droppath": dict(cls="Layer_scale_init_BlockD8", kw=dict(dim=64, num_heads=2, qkv_bias=True, drop_path=0.5, init_values=0.1),
                                inp=("tuple5", 4, 17, 8), train=True, rng_seed=1234),
    "block_dino": dict(cls="BlockD8", kw=dict(dim=64, num_heads=2, init_values=0.1), inp=("tuple5", 2, 17, 8)),
    "block_dino_droppath": dict(cls="BlockD8", kw=dict(dim=64, num_heads=2, init_values=0.1, drop_path=0.25),
                                inp=("tuple5", 4, 17, 8), train=True, rng_seed=4321),
    "lift": dict(cls="LiftD8", kw=dict(in_channels=3, out_channels=64, kernel_size=4, stride=4, bias=True), inp=("image", 2, 3, 16, 16)),
    "patch_embed": dict(cls="PatchEmbedD8", kw=dict(img_size=16, patch_size=4, in_chans=3, embed_dim=64), inp=("image", 2, 3, 16, 16)),
    "patch_embed_p14": dict(cls="PatchEmbedD8", kw=dict(img_size=28, patch_size=14, in_chans=3, embed_dim=64), inp=("image", 2, 3, 28, 28)),
    "iso_to_patch": dict(cls="IsotypicToPatchD8", kw=dict(dim=64, patch_side=4, reshape_to_image=True), inp=("tuple5", 2, 16, 8, False)),
    "inv_linear": dict(cls="LinearInvariant", kw=dict(C=64), inp=("tuple5", 2, 5, 8)),
    "inv_power_spectrum": dict(cls="PowerSpectrumInvariant", kw=dict(C=64), inp=("tuple5", 2, 5, 8)),
    "inv_polynomial": dict(cls="PolynomialInvariant", kw=dict(C=64), inp=("tuple5", 2, 5, 8)),
    "inv_thirdorder": dict(cls="ThirdOrderInvariant", kw=dict(C=64), inp=("tuple5", 2, 5, 8)),
    "inv_maxfilter": dict(cls="MaxFilteringInvariant", kw=dict(input_channels=64, num_references=24), inp=("tuple5", 2, 5, 8)),
    "inv_canonization": dict(cls="CanonizationInvariant", kw=dict(dim=64), inp=("tuple5", 2, 5, 8), no_grad=True),
    "inv_noninvariant": dict(cls="NonInvariant", kw=dict(C=64), inp=("tuple5", 2, 5, 8)),
    # whole models (img 32, patch 4 -> 8x8 grid, T=65)
    "model_hybrid": dict(model=dict(img_size=32, patch_size=4, num_classes=10, embed_dim=128, depth=4, num_heads=4,
                                    qkv_bias=True, blocks="deit", init_scale=0.1), inp=("image", 2, 3, 32, 32)),
    "model_invariant": dict(model=dict(img_size=32, patch_size=4, num_classes=10, embed_dim=128, depth=4, num_heads=4,
                                       qkv_bias=True, blocks="deit", invariant=True, init_scale=0.1), inp=("image", 2, 3, 32, 32)),
    "model_default_blocks": dict(model=dict(img_size=32, patch_size=4, num_classes=10, embed_dim=64, depth=2, num_heads=2,
                                            blocks="default", octic_equi_break_layer=1, init_scale=0.1), inp=("image", 2, 3, 32, 32)),
    "model_global_pool": dict(model=dict(img_size=32, patch_size=4, num_classes=10, embed_dim=64, depth=2, num_heads=2,
                                         qkv_bias=True, blocks="deit", global_pool=True, init_scale=0.1), inp=("image", 2, 3, 32, 32)),
    # BASELINE.json configs[0]: ViT-S/16, one 1x3x224x224 forward (logits only; 12.44 M params are
    # filled by name so no weights need to be stored)
    "vit_s16_forward": dict(model=dict(img_size=224, patch_size=16, num_classes=1000, embed_dim=384, depth=12, num_heads=6,
                                       blocks="default", init_scale=0.1), inp=("image", 1, 3, 224, 224), no_grad=True),
}

# function-style cases (no module): handled explicitly by run_case
FUNC_CASES = ["transforms", "pos_unfold", "expand_weight", "converters", "group_actions"]


def make_input(case_name, recipe):
    kind = recipe[0]
    if kind == "tuple5":
        return tuple5(case_name + ".in", *recipe[1:])
    if kind == "tuple8":
        return tuple8(case_name + ".in", *recipe[1:])
    if kind == "image":
        return randn(case_name + ".in", *recipe[1:])
    raise ValueError(kind)


def build_model(ns, spec):
    spec = dict(spec)
    blocks = spec.pop("blocks")
    if blocks == "deit":
        spec["octic_block_layers"] = ns.Layer_scale_init_BlockD8
        spec["standard_block_layers"] = ns.Layer_scale_init_Block
    return ns.OcticVisionTransformer(**spec)


def run_module_case(ns, name, device="cpu", dtype=torch.float32, to_module=None):
    """Build, fill, run fwd(+bwd) of one module case.  Returns {key: np.ndarray}."""
    case = CASES[name]
    if "model" in case:
        mod = build_model(ns, case["model"])
    else:
        mod = getattr(ns, case["cls"])(**case["kw"])
    fill_parameters(mod)
    mod = mod.to(device)
    if to_module is not None:
        mod = to_module(mod)
    mod.train(bool(case.get("train", False)))
    inp = make_input(name, case["inp"])
    leaves = [t.clone().to(device).requires_grad_(not case.get("no_grad", False))
              for t in (_flatten_out(inp))]
    arg = leaves[0] if isinstance(inp, torch.Tensor) else tuple(leaves)
    if "rng_seed" in case:
        torch.manual_seed(case["rng_seed"])
    res = {}
    if case.get("no_grad", False):
        with torch.no_grad():
            outs = _flatten_out(mod(arg))
        for i, o in enumerate(outs):
            res[f"out.{i}"] = o.detach().float().cpu().numpy()
        return res
    outs = _flatten_out(mod(arg))
    loss = 0.0
    for i, o in enumerate(outs):
        res[f"out.{i}"] = o.detach().float().cpu().numpy()
        cot = randn(f"{name}.cot.{i}", *o.shape).to(o.device)
        loss = loss + (o.float() * cot).sum()
    loss.backward()
    is_image = case["inp"][0] == "image"
    if not is_image:
        for i, l in enumerate(leaves):
            g = l.grad if l.grad is not None else torch.zeros_like(l)  # unused inputs (e.g. LinearInvariant)
            res[f"gin.{i}"] = g.detach().float().cpu().numpy()
    big = "model" in case
    for pname, p in mod.named_parameters():
        if p.grad is None:
            continue
        g = p.grad.detach().float().cpu()
        if big and g.numel() > 4096:
            # large models: keep a strided sample + norm so the fixture stays small
            flat = g.flatten()
            res[f"gpar_sample.{pname}"] = flat[:: max(1, flat.numel() // 256)][:256].numpy()
            res[f"gpar_norm.{pname}"] = np.array([float(flat.norm())], dtype=np.float32)
        else:
            res[f"gpar.{pname}"] = g.numpy()
    return res


def run_func_case(ns, name):
    """Function-level cases.  ``ns`` must expose the reference's d8_utils names."""
    res = {}
    if name == "transforms":
        xs = tuple8("transforms.in", 2, 5, 8)
        for i, y in enumerate(ns.isotypic_to_regular_D8(xs)):
            res[f"iso2reg.{i}"] = y.numpy()
        for i, y in enumerate(ns.regular_to_isotypic_D8(xs)):
            res[f"reg2iso.{i}"] = y.numpy()
    elif name == "converters":
        xs = tuple8("converters.in", 2, 5, 8)
        five = ns.convert_8tuple_to_5tuple(xs)
        for i, y in enumerate(five):
            res[f"to5.{i}"] = y.numpy()
        for i, y in enumerate(ns.convert_5tuple_to_8tuple(five)):
            res[f"to8.{i}"] = y.numpy()
    elif name == "pos_unfold":
        for g in (2, 7):
            xs = tuple(randn(f"pos_unfold.{g}.{i}", g, g, 8) for i in range(6))
            for i, y in enumerate(ns.isotypic_dim_interpolation(xs, dim=0)):
                res[f"g{g}.{i}"] = y.contiguous().numpy()
    elif name == "expand_weight":
        for irrep in ("A1", "A2", "B1", "B2", "E"):
            for p in (4, 14, 16):
                conv = ns.LiftIrrepD8Conv2d(3, 4, p, p, bias=False, irrep=irrep)
                with torch.no_grad():
                    conv.weight.copy_(randn(f"expand_weight.{irrep}.{p}", *conv.weight.shape))
                    res[f"{irrep}.{p}"] = conv.expand_weight().contiguous().numpy()
    elif name == "group_actions":
        xs = tuple8("group_actions.in", 2, 16, 4)
        img = randn("group_actions.img", 2, 3, 6, 6)
        for g in ns.group_elements:
            for i, y in enumerate(ns.isotypic_group_action(g, xs)):
                res[f"iso.{g}.{i}"] = y.numpy()
            for i, y in enumerate(ns.regular_group_action(g, xs)):
                res[f"reg.{g}.{i}"] = y.numpy()
            for i, y in enumerate(ns.spatial_and_isotypic_group_action(g, xs)):
                res[f"spat.{g}.{i}"] = y.contiguous().numpy()
            res[f"img.{g}"] = ns.image_space_group_action(g, img).contiguous().numpy()
    else:
        raise ValueError(name)
    return res
