"""DINOv2 SSL student/teacher step on the HIP engine (octic_vits_amd/ssl.py) against the CPU oracle's restatement
(oracle/ssl_ref.py, whose pieces are pinned to the real reference by tests/golden/ssl_pieces.npz): same parameters, same
multi-crop batch (2 global + 2 local crops at a non-native resolution, iBOT masks), every loss term and a sample of the
student gradients.  f32 path 1e-3 (north-star forward tolerance), bf16 autocast 5e-2 on the losses."""
import random

import pytest
import torch

import cases
from oracle import octic_ref as R
from oracle import ssl_ref as SR

pytestmark = pytest.mark.gpu

KW = dict(head_n_prototypes=64, head_hidden_dim=48, head_bottleneck_dim=16, local_crops_number=2)
DIM = 128


def _oracle_backbone():
    m = R.OcticDinoVisionTransformer(img_size=32, patch_size=4, embed_dim=DIM, depth=4, num_heads=4,
                                     octic_block_layers=lambda **kw: R.NestedTensorBlockD8(init_values=0.1, **{k: v for k, v in kw.items() if k != "init_values"}),
                                     standard_block_layers=lambda **kw: R.NestedTensorBlock(init_values=0.1, **{k: v for k, v in kw.items() if k != "init_values"}))
    m.patch_embed.strict_img_size = False
    return m


def _product_backbone():
    from functools import partial
    from octic_vits_amd import d8_layers, dinov2_models, vit
    return dinov2_models.OcticDinoVisionTransformer(
        img_size=32, patch_size=4, embed_dim=DIM, depth=4, num_heads=4,
        octic_block_layers=lambda **kw: d8_layers.NestedTensorBlockD8(init_values=0.1, **{k: v for k, v in kw.items() if k != "init_values"}),
        standard_block_layers=lambda **kw: vit.NestedTensorBlock(attn_class=vit.MemEffAttention, init_values=0.1,
                                                                 **{k: v for k, v in kw.items() if k != "init_values"}))


def _batch():
    random.seed(11)
    gc, lc = cases.randn("ssl.step.g", 8, 3, 32, 32), cases.randn("ssl.step.l", 8, 3, 16, 16)
    return SR.collate(gc, lc, (0.1, 0.5), 0.5, 64, SR.MaskingGenerator((8, 8), max_num_patches=32))


def _to(images, dev):
    return {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in images.items()}


def _pair():
    from octic_vits_amd import ssl as S
    torch.manual_seed(0)
    ref = SR.SSLMetaArch(_oracle_backbone, DIM, **KW)
    cases.fill_parameters(ref.student, salt="ssl.")
    for k in ref.student:
        ref.teacher[k].load_state_dict(ref.student[k].state_dict())
    mine = S.SSLMetaArch(_product_backbone, DIM, **KW)
    mine.student.load_state_dict(ref.student.state_dict(), strict=True)
    mine.teacher.load_state_dict(ref.teacher.state_dict(), strict=True)
    return ref.train(), mine.cuda().train()


def test_non_native_resolution_tokens_match_oracle():
    """Local crops: pos-embed resize branch (raises TypeError in the reference as shipped; oracle = evident intent)."""
    ref, mine = _pair()
    x = cases.randn("ssl.res.x", 2, 3, 16, 16)
    with torch.no_grad():
        a = mine.student.backbone.forward_features(x.cuda())
        b = ref.student.backbone.forward_features(x)
    for k in ("x_norm_clstoken", "x_norm_patchtokens"):
        scale = max(1.0, float(b[k].abs().max()))
        assert float((a[k].cpu() - b[k]).abs().max()) <= 1e-3 * scale, k


@pytest.mark.parametrize("centering", ["centering", "sinkhorn_knopp"])
def test_ssl_forward_backward_matches_oracle_f32(centering):
    ref, mine = _pair()
    ref.centering = mine.centering = centering
    images = _batch()
    want = ref.forward_backward(images, teacher_temp=0.05)
    got = mine.forward_backward(_to(images, "cuda"), teacher_temp=0.05)
    assert set(got) == set(want)
    for k in want:
        assert float(got[k]) == pytest.approx(float(want[k]), rel=1e-3, abs=1e-4), k
    # centres queued for the next step agree, too
    ref.dino_loss.apply_center_update(); mine.dino_loss.apply_center_update()
    assert torch.allclose(mine.dino_loss.center.cpu(), ref.dino_loss.center, atol=1e-5)
    n = 0
    mine_by_name = dict(mine.student.named_parameters())
    for name, p in ref.student.named_parameters():
        q = mine_by_name[name]
        if p.grad is None:
            assert q.grad is None or float(q.grad.abs().max()) == 0.0, name
            continue
        scale = max(1e-6, float(p.grad.abs().max()))
        err = float((q.grad.cpu() - p.grad).abs().max())
        if ".qkv." in name and name.endswith("bias"):
            continue        # K third: exactly zero gradient, rounding noise only
        assert err <= 2e-3 * scale + 1e-7, f"{name}: {err:.3e} vs scale {scale:.3e}"
        n += 1
    assert n > 60


def test_ssl_trainer_steps_bf16():
    """Two full iterations (bf16 autocast student/teacher forward, clip, AdamW, teacher EMA): losses finite and within
    5e-2 of the f32 oracle on the first step, the teacher moves towards the student, its bf16 weight caches follow."""
    from octic_vits_amd import ssl as S
    ref, mine = _pair()
    images = _batch()
    want = ref.forward_backward(images, teacher_temp=0.05, backward=False)
    tr = S.SSLTrainer(mine, lr=1e-3)
    t0 = [p.detach().clone() for p in mine.teacher.parameters()]
    out1 = tr.step(_to(images, "cuda"), teacher_temp=0.05, momentum=0.9)
    for k in want:
        assert float(out1[k]) == pytest.approx(float(want[k]), rel=5e-2, abs=5e-3), k
    out2 = tr.step(_to(images, "cuda"), teacher_temp=0.05, momentum=0.9)
    assert all(torch.isfinite(v).all() for v in out2.values())
    trainable = [p.requires_grad for p in mine.student.parameters()]     # frozen cls/mask token copies stay zeros
    moved = [not torch.equal(a, b) for a, b, tr_ in zip(t0, mine.teacher.parameters(), trainable) if tr_ and a.ndim >= 2]
    assert all(moved) and len(moved) > 40 and all(p.grad is None for p in mine.teacher.parameters())
    for (nt, t), (ns_, s_) in zip(mine.teacher.named_parameters(), mine.student.named_parameters()):
        assert nt == ns_ and t.shape == s_.shape


@pytest.mark.timeout(900)
def test_ssl_step_hybrid_vit_huge_multicrop():
    """BASELINE configs[4] geometry at the reference's largest octic DINOv2 model (hybrid ViT-H/16; the reference has no
    ViT-g): 2 x 224^2 + 8 x 96^2 crops, 4 images, 65536 prototypes - one full bf16 iteration, finite."""
    from octic_vits_amd import ssl as S
    from octic_vits_amd.dinov2_models import hybrid_dinov2_vit_huge_patch16
    torch.manual_seed(0)
    arch = S.SSLMetaArch(lambda: hybrid_dinov2_vit_huge_patch16(img_size=224, drop_path_rate=0.0), 1280).cuda()
    tr = S.SSLTrainer(arch, lr=1e-4)
    images = S.synthetic_multicrop_batch(4, "cuda", seed=5)
    out = tr.step(images, teacher_temp=0.04, momentum=0.992)
    assert all(torch.isfinite(v).all() for v in out.values()), out
    assert float(out["dino_local_crops_loss"]) > 0 and float(out["ibot_loss"]) > 0
