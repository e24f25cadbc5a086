"""Generate the committed golden vectors by RUNNING THE REAL REFERENCE on CPU.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

Runs only where ``/root/reference`` exists (the build container).  The outputs
are data (inputs are re-derived from seeds; files hold the reference's outputs
and gradients) — no reference source is copied.  See ``_ref_import.py`` for how
the reference is made importable without ``timm``/GPU.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import cases  # noqa: E402
from _ref_import import load_reference  # noqa: E402


def reference_namespace():
    ref = load_reference()
    ns = types.SimpleNamespace()
    for mod in (ref.d8_utils, ref.d8_inv, ref.d8_layers, ref.model):
        for k, v in vars(mod).items():
            if not k.startswith("_"):
                setattr(ns, k, v)
    ns.TritonGeluD8 = ref.d8_gelu.TritonGeluD8
    ns.Layer_scale_init_Block = ref.deit_vit.Layer_scale_init_Block
    ns.create_model = ref.create_model
    return ns


def dino_namespace():
    """The reference's DINOv2 entry points (octic_vits/dinov2_models.py) and the dinov2.layers block they use."""
    import importlib
    from functools import partial
    load_reference()
    dm = importlib.import_module("octic_vits.dinov2_models")
    ns = types.SimpleNamespace()
    ns.OcticDinoVisionTransformer = dm.OcticDinoVisionTransformer
    ns.NestedTensorBlockD8 = dm.BlockD8
    ns.DinoBlock = partial(dm.Block, attn_class=dm.MemEffAttention)
    return ns


def main():
    torch.set_num_threads(8)
    ns = reference_namespace()
    total = 0
    for name in cases.CASES:
        res = cases.run_module_case(ns, name)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **res)
        total += os.path.getsize(path)
        print(f"{name:28s} {len(res):4d} arrays  {os.path.getsize(path) / 1024:8.1f} KiB")
    for name in cases.FUNC_CASES:
        res = cases.run_func_case(ns, name)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **res)
        total += os.path.getsize(path)
        print(f"{name:28s} {len(res):4d} arrays  {os.path.getsize(path) / 1024:8.1f} KiB")
    import dino_cases
    dns = dino_namespace()
    for name in dino_cases.DINO_CASES:
        res = dino_cases.run_dino_case(dns, name)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **res)
        total += os.path.getsize(path)
        print(f"{name:28s} {len(res):4d} arrays  {os.path.getsize(path) / 1024:8.1f} KiB")
    # known-answer facts about the BASELINE models: parameter counts
    facts = {}
    for mname in ("hybrid_deit_huge_patch14", "d8_inv_early_deit_huge_patch14",
                  "hybrid_deit_large_patch16", "d8_inv_early_deit_large_patch16"):
        with torch.device("meta"):
            m = ns.create_model(mname, num_classes=1000)
        facts[mname + ".params"] = np.array([sum(p.numel() for p in m.parameters())], dtype=np.int64)
        facts[mname + ".tensors"] = np.array([len(list(m.parameters()))], dtype=np.int64)
        keys = sorted(m.state_dict().keys())
        facts[mname + ".keys_crc"] = np.array([__import__("zlib").crc32("\n".join(keys).encode())], dtype=np.int64)
    path = os.path.join(HERE, "model_facts.npz")
    np.savez_compressed(path, **facts)
    total += os.path.getsize(path)
    for k, v in facts.items():
        print(k, int(v[0]))
    print(f"total {total / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
