"""GPU parity of the HIP-backed modules against the golden vectors generated from the REAL reference
(tests/golden/*.npz) — same class names, same ctor kwargs, same name-keyed parameters, same seeded inputs.

fp32 path: exact-f32 MFMA + f32 everywhere -> rtol/atol 1e-4 of the output scale (models: 1e-3; the
BASELINE north star asks forward <= 1e-3 rel).  bf16 autocast path: documented looser tolerance (5e-2 of
scale on block outputs with O(1) layer-scale; bf16 has 8 mantissa bits)."""
import os
import types

import numpy as np
import pytest
import torch

import cases

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def product_ns():
    import octic_vits_amd.d8_invariantization as I
    import octic_vits_amd.d8_layers as L
    import octic_vits_amd.d8_utils as U
    import octic_vits_amd.model as M
    import octic_vits_amd.vit as V
    ns = types.SimpleNamespace()
    for mod in (U, I, L, M):
        for k, v in vars(mod).items():
            if not k.startswith("_"):
                setattr(ns, k, v)
    ns.Layer_scale_init_Block = V.Layer_scale_init_Block
    return ns


@pytest.fixture(autouse=True)
def _reference_drop_path_stream():
    """Draw drop-path masks exactly like the reference does on CPU (new_empty(B,1,1).bernoulli_) so the seeded
    golden cases are reproducible on the GPU."""
    import octic_vits_amd.d8_layers as L
    L.drop_path_mask_source = lambda B, keep, device: torch.empty((B, 1, 1)).bernoulli_(keep).flatten().to(device)
    yield
    L.drop_path_mask_source = None


def _check(name, got, want, rtol, atol, grad_tol=None):
    assert set(got) == set(want.files), sorted(set(got) ^ set(want.files))
    worst = 0.0
    for k in want.files:
        if grad_tol is not None and not k.startswith("out."):
            rtol = atol = grad_tol
        w = want[k].astype(np.float64)
        g = got[k].astype(np.float64)
        scale = max(1.0, float(np.abs(w).max()))
        err = float(np.abs(g - w).max())
        worst = max(worst, err / scale)
        assert np.allclose(g, w, rtol=rtol, atol=atol * scale), f"{name}:{k} max err {err:.3e} scale {scale:.3g}"
    return worst


SKIP_HIP = {"inv_linear", "inv_polynomial", "inv_thirdorder", "inv_maxfilter", "inv_canonization", "inv_noninvariant"}


@pytest.mark.parametrize("name", [n for n in cases.CASES if n not in SKIP_HIP])
def test_fp32_matches_reference_golden(name):
    got = cases.run_module_case(product_ns(), name, device="cuda")
    want = np.load(os.path.join(GOLD, name + ".npz"))
    tol = 1e-3 if (name.startswith("model") or name.startswith("vit_")) else 1e-4
    _check(name, got, want, tol, tol)


class _Autocast(torch.nn.Module):
    def __init__(self, mod):
        super().__init__()
        self.mod = mod

    def named_parameters(self, *a, **k):  # keep the reference's parameter names in the result dict
        return self.mod.named_parameters(*a, **k)

    def forward(self, x):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            return self.mod(x)


class _AutocastFp16(_Autocast):
    def forward(self, x):
        with torch.autocast("cuda", dtype=torch.float16):
            return self.mod(x)


@pytest.mark.parametrize("name", ["mlp", "attention", "block_deit", "block_dino", "model_hybrid"])
def test_fp16_autocast_runs_the_octic_half_in_f32(name):
    """fp16 autocast is the reference's DeiT default (deit/engine.py:56).  The engine has no fp16 kernels: the octic
    blocks then compute in float32 (functional.compute_dtype), the standard blocks run the reference's eager fp16 ops.
    Outputs and gradients must stay within fp16-autocast distance of the f32 goldens: 1e-4 where everything is octic,
    2e-2 relative L2 for the hybrid model (its standard half really is fp16)."""
    got = cases.run_module_case(product_ns(), name, device="cuda", to_module=_AutocastFp16)
    want = np.load(os.path.join(GOLD, name + ".npz"))
    lim = 2e-2 if name.startswith("model") else 2e-3
    for k in want.files:
        if k not in got:
            continue
        a, b = got[k].astype(np.float64).ravel(), want[k].astype(np.float64).ravel()
        rel = np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-12)
        assert rel <= lim, f"{name}:{k} rel L2 err {rel:.5f} > {lim}"


BF16_CASES = ["linear_bias", "layernorm", "gelu", "mlp", "attention", "block_deit", "block_deit_droppath", "block_dino",
              "patch_embed", "model_hybrid", "model_invariant"]


@pytest.mark.parametrize("name", BF16_CASES)
def test_bf16_autocast_close_to_reference_golden(name):
    got = cases.run_module_case(product_ns(), name, device="cuda", to_module=_Autocast)
    want = np.load(os.path.join(GOLD, name + ".npz"))
    if not name.startswith("model"):
        _check(name, got, want, 5e-2, 5e-2)
        return
    # Whole models.  The yardstick for "what bf16 costs" is the reference itself under bf16 autocast: the oracle (pinned to the
    # reference by the f32 goldens) is run on the CPU under torch.autocast(bfloat16) - the same operands rounded at the
    # same places - and the product's distance to the f32 golden must stay within max(3e-2, 2 x the oracle's own distance),
    # per tensor, in relative L2 (measured on MI355X, round 2: product median 1.6e-2 / max 3.9e-2, oracle median 1.8e-2 /
    # max 4.1e-2 on model_hybrid).
    # One documented exception: in the invariant model the A2 / B1 / B2 streams reach the head through |x| (PowerSpectrum,
    # d8_invariantization.py:49-64), whose cotangent is g * sign(x).  The head reads the cls token only, so g is concentrated
    # on the cls row - whose A2/B1/B2 components start as frozen zeros (model.py:99-105) and are of the size of bf16 rounding
    # noise at the hand-off: sign(x) of those few dozen elements is decided by rounding, differently in any two bf16
    # implementations (tools/inv_diag.py: cotangent at the invariant output within 0.8 %, kernel dx equal to the formula on
    # its own (g, x) to 1e-8, yet 9-21 % on the A2/B1 stream cotangent).  Those tensors are held to 0.35.
    from oracle import octic_ref as R

    class _CpuAutocast(torch.nn.Module):
        def __init__(self, mod):
            super().__init__()
            self.mod = mod

        def named_parameters(self, *a, **k):
            return self.mod.named_parameters(*a, **k)

        def forward(self, x):
            with torch.autocast("cpu", dtype=torch.bfloat16):
                return self.mod(x)

    yard = cases.run_module_case(R, name, device="cpu", to_module=_CpuAutocast)
    assert set(got) == set(want.files)
    for k in want.files:
        g, w, y = got[k].astype(np.float64), want[k].astype(np.float64), yard[k].astype(np.float64)
        if k.startswith("out."):
            scale = max(1.0, float(np.abs(w).max()))
            assert np.allclose(g, w, rtol=5e-2, atol=5e-2 * scale), f"{name}:{k}"
            continue
        den = max(np.linalg.norm(w), 1e-3) if not k.startswith("gpar_norm.") else max(abs(w[0]), 1e-3)
        rel = (np.linalg.norm(g - w) if not k.startswith("gpar_norm.") else abs(g[0] - w[0])) / den
        rel_oracle = (np.linalg.norm(y - w) if not k.startswith("gpar_norm.") else abs(y[0] - w[0])) / den
        lim = max(3e-2, 2.0 * rel_oracle)
        # everything upstream of the |.| is touched by it (the A2/B1/B2 tensors directly: up to 0.32 measured; the A1 / E
        # tensors through the attention of the octic blocks, whose q.k products mix all irreps: up to 0.08 measured);
        # tensors downstream of the hand-off (standard blocks, final norm, head, invariant_proj) keep the tight bound
        downstream = any(t in k for t in (".blocks.2.", ".blocks.3.", ".norm.", ".head.", ".invariant_proj."))
        fragile = name == "model_invariant" and not downstream
        if fragile:
            lim = max(lim, 0.35)
        assert rel <= lim, f"{name}:{k} rel L2 err {rel:.4f} > {lim:.4f} (reference under bf16 autocast: {rel_oracle:.4f})"


def test_gelu_function_is_a_drop_in_for_the_reference_custom_op():
    """TritonGeluD8Function.apply(x_A1, x_A2, x_B1, x_B2, x_2d) -> 5 tensors (d8_gelu.py:456-478)."""
    from octic_vits_amd.functional import GeluD8Function
    from oracle import octic_ref as R
    xs = tuple(t.cuda().requires_grad_(True) for t in cases.tuple5("dropin", 3, 11, 24))
    ys = GeluD8Function.apply(*xs)
    ref_in = tuple(t.detach().cpu().requires_grad_(True) for t in xs)
    yr = R.TritonGeluD8()(ref_in)
    for a, b in zip(ys, yr):
        assert a.shape == b.shape and torch.allclose(a.cpu(), b, atol=2e-6, rtol=1e-5)
    cot = [cases.randn(f"dropin.cot{i}", *y.shape) for i, y in enumerate(yr)]
    torch.autograd.backward(ys, [c.cuda() for c in cot])
    torch.autograd.backward(yr, cot)
    for a, b in zip(xs, ref_in):
        assert torch.allclose(a.grad.cpu(), b.grad, atol=2e-6, rtol=1e-5)


def test_reference_state_dict_loads_into_product_model():
    """Drop-in checkpoint surface: an oracle (== reference-keyed) state_dict loads strictly and gives the same logits."""
    from octic_vits_amd.deit_models import create_model
    from oracle import octic_ref as R
    kw = dict(img_size=32, num_classes=7)
    ref = cases.fill_parameters(R.OcticVisionTransformer(patch_size=4, embed_dim=128, depth=4, num_heads=4, qkv_bias=True,
                                                         octic_block_layers=R.Layer_scale_init_BlockD8,
                                                         standard_block_layers=R.Layer_scale_init_Block,
                                                         init_scale=0.1, **kw)).eval()
    from octic_vits_amd.d8_layers import Layer_scale_init_BlockD8
    from octic_vits_amd.model import OcticVisionTransformer
    from octic_vits_amd.vit import Layer_scale_init_Block
    mine = OcticVisionTransformer(patch_size=4, embed_dim=128, depth=4, num_heads=4, qkv_bias=True,
                                  octic_block_layers=Layer_scale_init_BlockD8,
                                  standard_block_layers=Layer_scale_init_Block, **kw)
    sd = {("_orig_mod." + k if False else k): v for k, v in ref.state_dict().items()}
    mine.load_state_dict(sd, strict=True)
    mine = mine.cuda().eval()
    img = cases.randn("sd.img", 2, 3, 32, 32)
    with torch.no_grad():
        a, b = mine(img.cuda()).cpu(), ref(img)
    assert torch.allclose(a, b, atol=1e-3 * float(b.abs().max()), rtol=1e-3)


# ---- DINOv2 entry points (first slice): product vs goldens from the real reference
import dino_cases  # noqa: E402


def _product_dino_ns():
    import types
    from functools import partial

    from octic_vits_amd import d8_layers, dinov2_models, vit
    ns = types.SimpleNamespace()
    ns.OcticDinoVisionTransformer = dinov2_models.OcticDinoVisionTransformer
    ns.NestedTensorBlockD8 = d8_layers.NestedTensorBlockD8
    ns.DinoBlock = partial(vit.NestedTensorBlock, attn_class=vit.MemEffAttention)
    return ns


@pytest.mark.parametrize("name", list(dino_cases.DINO_CASES))
def test_dino_fp32_matches_reference_golden(name):
    """Forward (cls feature, feature dict with mask / register tokens, intermediate layers) within the north-star 1e-3;
    parameter gradients by norm and strided sample."""
    got = dino_cases.run_dino_case(_product_dino_ns(), name, device="cuda")
    want = np.load(os.path.join(GOLD, name + ".npz"))
    assert set(got) == set(want.files), sorted(set(got) ^ set(want.files))
    for k in want.files:
        w = want[k].astype(np.float64)
        assert got[k].shape == w.shape, f"{name}:{k}"
        if w.size == 0:
            continue
        scale = max(1.0, float(np.abs(w).max()))
        assert np.allclose(got[k].astype(np.float64), w, rtol=1e-3, atol=1e-3 * scale), \
            f"{name}:{k} max err {float(np.abs(got[k] - w).max()):.3e} (scale {scale:.3g})"


def test_dino_list_forward_and_bf16():
    """List-of-crops forward equals the per-crop forward; bf16 autocast stays within 5e-2 of the f32 features."""
    m = dino_cases.build(_product_dino_ns(), dict(num_register_tokens=1, invariant=False)).cuda()
    a, b = cases.randn("dino.list.a", 2, 3, 32, 32).cuda(), cases.randn("dino.list.b", 3, 3, 32, 32).cuda()
    with torch.no_grad():
        outs = m.forward_features([a, b], [None, None])
        for o, x in zip(outs, (a, b)):
            ref = m.forward_features(x)
            for k in ("x_norm_clstoken", "x_norm_patchtokens", "x_prenorm"):
                torch.testing.assert_close(o[k], ref[k], rtol=1e-5, atol=1e-5)
        f32 = m.forward_features(a)["x_norm_patchtokens"]
        with torch.autocast("cuda", dtype=torch.bfloat16):
            b16 = m.forward_features(a)["x_norm_patchtokens"].float()
    assert float((b16 - f32).abs().max()) < 5e-2 * max(1.0, float(f32.abs().max()))


def test_octic_next_norm_node_is_bitwise_the_separate_nodes():
    """Three chained Layer_scale_init_BlockD8 (d8_layers.link_octic_blocks) under bf16 autocast: with the residual-fused
    proj / fc2 and the following LayerNormD8 as one autograd node (its backward emits the scaled bf16 cotangent, no
    cast_rowscale pass) the output and every gradient equal the separate nodes bit for bit."""
    import octic_vits_amd.d8_layers as L
    import octic_vits_amd.functional as OF
    from octic_vits_amd import ops
    torch.manual_seed(5)
    dim, heads, B, T = 8 * 32, 4, 4, 33
    blocks = torch.nn.ModuleList([L.Layer_scale_init_BlockD8(dim, heads, drop_path=0.3, init_values=0.5) for _ in range(3)]).cuda()
    for p in blocks.parameters():
        p.data.add_(torch.randn_like(p) * 0.05)
    L.link_octic_blocks(blocks)
    blocks.train()
    x = torch.randn(B, T, dim, device="cuda")
    gout = torch.randn(B, T, dim, device="cuda")
    res = {}
    saved = OF.OCTIC_NEXT_NORM
    try:
        for mode in (True, False):
            OF.OCTIC_NEXT_NORM = mode
            for p in blocks.parameters():
                p.grad = None
            xg = x.clone().requires_grad_(True)
            torch.manual_seed(11)
            ops.KERNEL_TIMER.enable()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                h = OF.Octic(xg, dim // 8)
                for blk in blocks:
                    h = blk(h)
            h.packed.backward(gout)
            summ = ops.KERNEL_TIMER.summary()
            ops.KERNEL_TIMER.disable()
            res[mode] = (h.packed.detach().clone(), xg.grad.clone(), [p.grad.clone() for p in blocks.parameters()],
                         sum(v["launches"] for n, v in summ.items() if n.startswith("cast_rowscale")))
    finally:
        OF.OCTIC_NEXT_NORM = saved
    assert res[False][3] - res[True][3] == 5          # every residual-fused layer but the last one lost its cast pass
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    for (n, _), a, b in zip(blocks.named_parameters(), res[True][2], res[False][2]):
        assert torch.equal(a, b), n
