"""Generate tests/golden/ssl_pieces.npz by RUNNING THE REAL REFERENCE classes of the DINOv2 SSL step on CPU:
dinov2/loss/{dino_clstoken_loss,ibot_patch_loss,koleo_loss}.py, dinov2/layers/dino_head.py, dinov2/data/masking.py,
dinov2/data/collate.py (build container only; nothing of the reference is copied, the file holds numbers).

    python tests/golden/make_ssl_golden.py
"""
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import ssl_case  # noqa: E402
from _ref_import import load_reference  # noqa: E402


def reference_ssl_namespace():
    import importlib
    import types
    load_reference()
    ns = types.SimpleNamespace()
    ns.DINOLoss = importlib.import_module("dinov2.loss.dino_clstoken_loss").DINOLoss
    ns.iBOTPatchLoss = importlib.import_module("dinov2.loss.ibot_patch_loss").iBOTPatchLoss
    ns.KoLeoLoss = importlib.import_module("dinov2.loss.koleo_loss").KoLeoLoss
    ns.DINOHead = importlib.import_module("dinov2.layers.dino_head").DINOHead
    # dinov2/data/__init__.py pulls in torchvision (not installed): the two data modules needed here are self-contained,
    # so they are executed from their files where they lie, without the package initialiser
    import importlib.util
    from _ref_import import REFERENCE_ROOT

    def by_path(name):
        spec = importlib.util.spec_from_file_location("_ref_dinov2_data_" + name,
                                                      os.path.join(REFERENCE_ROOT, "dinov2", "data", name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    ns.MaskingGenerator = by_path("masking").MaskingGenerator
    coll = by_path("collate").collate_data_and_cast

    def collate(global_crops, local_crops, mask_ratio_tuple, mask_probability, n_tokens, mask_generator):
        # the reference collates per-sample dicts: rebuild them from the stacked crops (crop-major order)
        B2 = len(global_crops)
        B = B2 // 2
        nl = len(local_crops) // B
        samples = [({"global_crops": [global_crops[i * B + b] for i in range(2)],
                     "local_crops": [local_crops[i * B + b] for i in range(nl)]},) for b in range(B)]
        return coll(samples, mask_ratio_tuple, mask_probability, torch.float32, n_tokens=n_tokens, mask_generator=mask_generator)
    ns.collate = collate
    return ns


if __name__ == "__main__":
    torch.set_num_threads(8)
    res = ssl_case.run_pieces(reference_ssl_namespace())
    path = os.path.join(HERE, "ssl_pieces.npz")
    np.savez_compressed(path, **res)
    print(f"{len(res)} arrays, {os.path.getsize(path) / 1024:.1f} KiB")
    for k in sorted(res):
        if res[k].size <= 4:
            print("  ", k, res[k])
