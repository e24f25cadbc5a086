"""DeiT-III octic entry points with the reference's registry names and hyper-parameters
(octic_vits/deit_models.py:11-72).  Registered with timm when timm is importable, otherwise in a local
registry reachable through ``create_model`` (same call shape as ``timm.create_model``, deit/main.py:272-279)."""
from .d8_layers import Layer_scale_init_BlockD8
from .model import OcticVisionTransformer
from .vit import Layer_scale_init_Block

_LOCAL_REGISTRY = {}

try:  # pragma: no cover - timm is optional
    from timm.models import register_model as _timm_register
except Exception:  # noqa: BLE001
    _timm_register = None


def register_model(fn):
    _LOCAL_REGISTRY[fn.__name__] = fn
    return _timm_register(fn) if _timm_register is not None else fn


def create_model(model_name, pretrained=False, **kwargs):
    if pretrained:
        raise RuntimeError("no pretrained weights are bundled (load a reference checkpoint with load_state_dict)")
    kwargs.pop("drop_block_rate", None)
    if model_name not in _LOCAL_REGISTRY:
        raise RuntimeError(f"Unknown model ({model_name}); known: {sorted(_LOCAL_REGISTRY)}")
    return _LOCAL_REGISTRY[model_name](**kwargs)


def _octic_deit(img_size, patch_size, embed_dim, depth, num_heads, invariant, kwargs):
    return OcticVisionTransformer(
        img_size=img_size, patch_size=patch_size, embed_dim=embed_dim, depth=depth, num_heads=num_heads, mlp_ratio=4,
        qkv_bias=True, invariant=invariant, standard_block_layers=Layer_scale_init_Block,
        octic_block_layers=Layer_scale_init_BlockD8, **kwargs)


@register_model
def hybrid_deit_large_patch16(img_size=224, **kwargs):
    return _octic_deit(img_size, 16, 1024, 24, 16, False, kwargs)


@register_model
def hybrid_deit_huge_patch14(img_size=224, **kwargs):
    return _octic_deit(img_size, 14, 1280, 32, 16, False, kwargs)


@register_model
def d8_inv_early_deit_huge_patch14(img_size=224, **kwargs):
    return _octic_deit(img_size, 14, 1280, 32, 16, True, kwargs)


@register_model
def d8_inv_early_deit_large_patch16(img_size=224, **kwargs):
    return _octic_deit(img_size, 16, 1024, 24, 16, True, kwargs)
