"""DINOv2 SSL step without a GPU: the oracle's restatement of the losses / head / masking / collate (oracle/ssl_ref.py)
and the product's torch classes (octic_vits_amd/ssl.py, device-agnostic) against tests/golden/ssl_pieces.npz, which
tests/golden/make_ssl_golden.py produced by running the REAL reference classes on CPU."""
import os
import types

import numpy as np
import pytest
import torch

import ssl_case

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _check(res, tol=1e-5):
    want = np.load(os.path.join(GOLD, "ssl_pieces.npz"))
    assert set(res) == set(want.files), sorted(set(res) ^ set(want.files))
    for k in want.files:
        w = want[k]
        if w.dtype.kind in "iu":
            assert np.array_equal(res[k], w), k
            continue
        scale = max(1.0, float(np.abs(w).max()))
        assert np.allclose(res[k], w, rtol=tol, atol=tol * scale), f"{k}: max err {np.abs(res[k] - w).max():.3e}"


def oracle_ns():
    from oracle import ssl_ref as S
    return types.SimpleNamespace(DINOLoss=S.DINOLoss, iBOTPatchLoss=S.iBOTPatchLoss, KoLeoLoss=S.KoLeoLoss,
                                 DINOHead=S.DINOHead, MaskingGenerator=S.MaskingGenerator, collate=S.collate)


def product_ns():
    from octic_vits_amd import ssl as S
    return types.SimpleNamespace(DINOLoss=S.DINOLoss, iBOTPatchLoss=S.iBOTPatchLoss, KoLeoLoss=S.KoLeoLoss,
                                 DINOHead=S.DINOHead, MaskingGenerator=S.MaskingGenerator, collate=S.collate)


def test_oracle_ssl_pieces_match_reference_golden():
    _check(ssl_case.run_pieces(oracle_ns()))


def test_product_ssl_pieces_match_reference_golden_on_cpu():
    """The SSL losses / head / data helpers of the product are plain torch (no HIP kernels): checked here on the host."""
    _check(ssl_case.run_pieces(product_ns()))


def test_oracle_ssl_step_runs_and_teacher_follows():
    """Composition check of the restated forward_backward on the oracle backbones (CPU, tiny): every loss term is finite,
    gradients reach backbone and head, the teacher is an EMA of the student and receives no gradient."""
    from oracle import octic_ref as R
    from oracle import ssl_ref as S
    import random
    torch.manual_seed(0)
    def make():
        m = R.OcticDinoVisionTransformer(img_size=32, patch_size=4, embed_dim=64, depth=2, num_heads=2,
                                         octic_block_layers=R.NestedTensorBlockD8,
                                         standard_block_layers=lambda **kw: R.NestedTensorBlock(**kw))
        m.patch_embed.strict_img_size = False      # local crops are smaller than the model's native size (see ssl.py)
        return m
    try:
        arch = S.SSLMetaArch(make, 64, head_n_prototypes=32, head_hidden_dim=48, head_bottleneck_dim=16, local_crops_number=2)
    except TypeError:
        pytest.skip("oracle block factory signature differs")
    arch.train()
    random.seed(3)
    g = torch.Generator().manual_seed(1)
    gc, lc = torch.randn(4, 3, 32, 32, generator=g), torch.randn(4, 3, 16, 16, generator=g)
    images = S.collate(gc, lc, (0.1, 0.5), 0.5, 64, S.MaskingGenerator((8, 8), max_num_patches=32))
    before = [p.detach().clone() for p in arch.teacher.parameters()]
    out = arch.forward_backward(images, teacher_temp=0.05)
    assert all(torch.isfinite(v).all() for v in out.values()) and set(out) >= {"dino_local_crops_loss", "dino_global_crops_loss", "koleo_loss", "ibot_loss"}
    assert all(p.grad is None for p in arch.teacher.parameters())
    assert sum(float(p.grad.abs().sum()) for p in arch.student.backbone.parameters() if p.grad is not None) > 0
    assert float(arch.student.dino_head.last_layer.weight_v.grad.abs().sum()) > 0
    with torch.no_grad():
        for p in arch.student.parameters():
            p.add_(0.01)
    arch.update_teacher(0.9)
    for b, t, s in zip(before, arch.teacher.parameters(), arch.student.parameters()):
        if s.requires_grad:
            assert torch.allclose(t, 0.9 * b + 0.1 * s.detach(), atol=1e-6)
