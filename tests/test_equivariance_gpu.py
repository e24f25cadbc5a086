"""The reference's acceptance script (experiments/test_equivariance.py, 21 checks) against the HIP-backed
modules: same layers, same shapes, same tolerances, fp32, on the GPU.  layer(g.x) must equal g.layer(x) for
all 8 elements of D8."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def U():
    import octic_vits_amd.d8_utils as u
    return u


def _tuple8(shape, offset=True):
    return tuple(torch.randn(shape, device=DEV) + (torch.randn(*shape[:-1], 1, device=DEV) if offset else 0.0)
                 for _ in range(8))


def test_group_action():
    u = U()
    xs = tuple(torch.randn(2, 64, 256) for _ in range(8))
    for a in (u.isotypic_group_action, u.regular_group_action):
        for g1, g2, g12 in u.mult_table:
            for x, y in zip(a(g1, a(g2, xs)), a(g12, xs)):
                assert torch.allclose(x, y)


def test_image_space_and_spatial_group_action():
    u = U()
    img = torch.randn(8, 3, 224, 224, device=DEV)
    xs = tuple(torch.randn(8, 196, 48, device=DEV) for _ in range(8))
    for g1, g2, g12 in u.mult_table:
        assert torch.allclose(u.image_space_group_action(g1, u.image_space_group_action(g2, img)),
                              u.image_space_group_action(g12, img))
        for x, y in zip(u.spatial_and_isotypic_group_action(g1, u.spatial_and_isotypic_group_action(g2, xs)),
                        u.spatial_and_isotypic_group_action(g12, xs)):
            assert torch.allclose(x, y)


def test_fourier_transforms():
    u = U()
    xs = tuple(torch.randn(2, 64, 256) for _ in range(8))
    for x, y in zip(u.regular_to_isotypic_D8(u.isotypic_to_regular_D8(xs)), xs):
        assert torch.allclose(x, y, atol=1e-6)
    for x, y in zip(u.isotypic_to_regular_D8(u.regular_to_isotypic_D8(xs)), xs):
        assert torch.allclose(x, y, atol=1e-6)
    for g in u.group_elements:
        for x, y in zip(u.isotypic_group_action(g, u.regular_to_isotypic_D8(xs)),
                        u.regular_to_isotypic_D8(u.regular_group_action(g, xs))):
            assert torch.allclose(x, y, atol=1e-6)
        for x, y in zip(u.isotypic_to_regular_D8(u.isotypic_group_action(g, xs)),
                        u.regular_group_action(g, u.isotypic_to_regular_D8(xs))):
            assert torch.allclose(x, y, atol=1e-6)


def _equi_iso_to_iso(layer, name, irrep_size=(128, 64, 256), atol=1e-6):
    """test_equivariance.py:145-161"""
    u = U()
    with torch.inference_mode():
        xs = _tuple8(list(irrep_size))
        base = u.convert_5tuple_to_8tuple(layer(u.convert_8tuple_to_5tuple(xs)))
        for g in u.group_elements:
            lhs = u.isotypic_group_action(g, base)
            rhs = u.convert_5tuple_to_8tuple(layer(u.convert_8tuple_to_5tuple(u.isotypic_group_action(g, xs))))
            for x, y, irrep in zip(lhs, rhs, u.irreps):
                assert not torch.allclose(x, torch.zeros_like(x), atol=atol), f"Bad test: {name} outputs 0 ({irrep}, {g})"
                assert torch.allclose(x, y, atol=atol), \
                    f"{name} doesn't commute with group action, irrep {irrep}, g={g}: {(x - y).abs().max().item():.2e}"


def test_equi_gelu_d8():
    from octic_vits_amd.d8_layers import TritonGeluD8
    _equi_iso_to_iso(TritonGeluD8(), "GeluD8")


def test_equi_linear_d8():
    from octic_vits_amd.d8_layers import LinearD8
    _equi_iso_to_iso(LinearD8(8 * 256, 384).to(DEV), "LinearD8")


def test_equi_layernorm_d8():
    from octic_vits_amd.d8_layers import LayerNormD8
    _equi_iso_to_iso(LayerNormD8(8 * 256).to(DEV), "LayerNormD8")


def test_equi_mlp_d8():
    from octic_vits_amd.d8_layers import MlpD8
    _equi_iso_to_iso(MlpD8(8 * 256).to(DEV), "MlpD8")


def test_equi_attention_d8():
    from octic_vits_amd.d8_layers import AttentionD8
    _equi_iso_to_iso(AttentionD8(dim=512).to(DEV), "AttentionD8", irrep_size=(32, 196, 64))


def test_equi_d8_block():
    from octic_vits_amd.d8_layers import BlockD8
    _equi_iso_to_iso(BlockD8(dim=768, num_heads=12).to(DEV), "BlockD8", irrep_size=(32, 196, 96))


def test_equi_lift_d8():
    """test_equivariance.py:182-195, 241-244"""
    from octic_vits_amd.d8_layers import LiftD8
    u = U()
    layer = LiftD8(3, 768, bias=True, kernel_size=16, stride=16).to(DEV)
    with torch.inference_mode():
        img = torch.randn(32, 3, 224, 224, device=DEV)
        base = layer(img)
        for g in u.group_elements:
            lhs = tuple(u.image_space_group_action(g, f) for f in u.isotypic_group_action(g, base))
            rhs = layer(u.image_space_group_action(g, img))
            for x, y, irrep in zip(lhs, rhs, u.irreps):
                assert not torch.allclose(x, torch.zeros_like(x), atol=1e-5)
                assert torch.allclose(x, y, atol=1e-5), f"LiftD8 {irrep} {g}: {(x - y).abs().max().item():.2e}"


def test_equi_patch_embed_d8():
    from octic_vits_amd.d8_layers import PatchEmbedD8
    u = U()
    layer = PatchEmbedD8(flatten=True).to(DEV)
    with torch.inference_mode():
        img = torch.randn(32, 3, 224, 224, device=DEV)
        base = u.convert_5tuple_to_8tuple(layer(img))
        for g in u.group_elements:
            lhs = u.spatial_and_isotypic_group_action(g, base)
            rhs = u.convert_5tuple_to_8tuple(layer(u.image_space_group_action(g, img)))
            for x, y, irrep in zip(lhs, rhs, u.irreps):
                assert not torch.allclose(x, torch.zeros_like(x), atol=1e-5)
                assert torch.allclose(x, y, atol=1e-5), f"PatchEmbedD8 {irrep} {g}: {(x - y).abs().max().item():.2e}"


def _equi_flat_iso_to_img(f, name, dim, atol=1e-5):
    """test_equivariance.py:257-268 / 324-335"""
    u = U()
    with torch.inference_mode():
        xs = tuple(torch.randn(32, 196, dim // 8, device=DEV) for _ in range(8))
        base = f(u.convert_8tuple_to_5tuple(xs))
        for g in u.group_elements:
            x = u.image_space_group_action(g, base)
            y = f(u.convert_8tuple_to_5tuple(u.spatial_and_isotypic_group_action(g, xs)))
            assert not torch.allclose(x, torch.zeros_like(x), atol=atol)
            assert torch.allclose(x, y, atol=atol), f"{name} g={g}: {(x - y).abs().max().item():.2e}"


def test_equi_iso_to_patch_d8():
    from octic_vits_amd.d8_layers import IsotypicToPatchD8
    _equi_flat_iso_to_img(IsotypicToPatchD8(dim=768, patch_side=16, reshape_to_image=True).to(DEV), "IsotypicToPatchD8", 768)


def test_invariance_deit_inv_early():
    """test_equivariance.py:302-322: logits invariant under D8, not invariant under a colour-channel flip."""
    from octic_vits_amd.model import OcticVisionTransformer
    u = U()
    net = OcticVisionTransformer(depth=4, embed_dim=768, invariant=True).to(DEV)
    with torch.inference_mode():
        imgs = torch.randn(33, 3, 224, 224, device=DEV)
        out1 = net(imgs)
        out3 = net(imgs.flip(-3))
        assert not torch.allclose(out1, torch.zeros_like(out1), atol=1e-4)
        assert not torch.allclose(out1, out3, atol=1e-4)
        for g in u.group_elements:
            out2 = net(u.image_space_group_action(g, imgs))
            assert torch.allclose(out1, out2, atol=1e-4), f"not invariant under {g}: {(out1 - out2).abs().max().item():.2e}"


@pytest.mark.parametrize("cls,mult,atol,kw", [
    ("LinearInvariant", 1, 1e-5, {}), ("PowerSpectrumInvariant", 6, 1e-5, {}), ("PolynomialInvariant", 32, 1e-4, {}),
    ("ThirdOrderInvariant", 15, 1e-5, {}), ("MaxFilteringInvariant", None, 1e-5, {"num_references": 1024}),
    ("CanonizationInvariant", 8, 1e-5, {})])
def test_invariants(cls, mult, atol, kw):
    """test_equivariance.py:338-391"""
    import octic_vits_amd.d8_invariantization as I
    net = getattr(I, cls)(8 * 256, **kw).to(DEV)
    f = lambda x: net(x).reshape(x[0].shape[0], 14, 14, -1).permute(0, 3, 1, 2)
    _equi_flat_iso_to_img(f, cls, 8 * 256, atol=atol)
