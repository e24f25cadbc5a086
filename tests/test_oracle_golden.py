"""The oracle must reproduce every golden vector generated from the real reference
(tests/golden/make_golden.py).  CPU only; fp32; tolerances stated per family."""
import os

import numpy as np
import pytest
import torch

import cases
from oracle import octic_ref

GOLD = os.path.join(os.path.dirname(__file__), "golden")

# fp32 CPU vs fp32 CPU: differences come only from summation order (matrix-form butterfly,
# einsum vs add chains).  Models accumulate over depth; polynomial invariants reach |x|^4.
TOL = {"default": (2e-5, 2e-5), "model": (2e-4, 2e-4), "inv_polynomial": (1e-4, 1e-3), "inv_thirdorder": (1e-4, 1e-4)}


def _tol(name):
    if name.startswith("model") or name.startswith("vit_"):
        return TOL["model"]
    return TOL.get(name, TOL["default"])


def _compare(name, got, want):
    rtol, atol = _tol(name)
    assert set(got) == set(want.files), (sorted(set(got) ^ set(want.files)))
    for k in want.files:
        np.testing.assert_allclose(got[k], want[k], rtol=rtol, atol=atol, err_msg=f"{name}:{k}")


@pytest.mark.parametrize("name", list(cases.CASES))
def test_module_case(name):
    torch.manual_seed(0)
    got = cases.run_module_case(octic_ref, name)
    _compare(name, got, np.load(os.path.join(GOLD, name + ".npz")))


@pytest.mark.parametrize("name", cases.FUNC_CASES)
def test_func_case(name):
    got = cases.run_func_case(octic_ref, name)
    want = np.load(os.path.join(GOLD, name + ".npz"))
    assert set(got) == set(want.files)
    for k in want.files:
        np.testing.assert_allclose(got[k], want[k], rtol=1e-6, atol=1e-6, err_msg=f"{name}:{k}")


def test_mult_table_matches_reference_known_answers():
    # spot values of the reference's hand-written table (d8_utils.py:18-74)
    t = {(a, b): c for a, b, c in octic_ref.mult_table}
    assert len(octic_ref.mult_table) == 49
    assert t[("r", "m")] == "mrrr" and t[("m", "r")] == "mr" and t[("mr", "mr")] == "e"
    assert t[("rrr", "mrrr")] == "m" and t[("mrr", "rr")] == "m" and t[("mrrr", "mrr")] == "rrr"


def test_model_facts():
    facts = np.load(os.path.join(GOLD, "model_facts.npz"))
    import zlib
    for mname in ("hybrid_deit_huge_patch14", "d8_inv_early_deit_huge_patch14",
                  "hybrid_deit_large_patch16", "d8_inv_early_deit_large_patch16"):
        with torch.device("meta"):
            m = octic_ref.create_model(mname, num_classes=1000)
        assert sum(p.numel() for p in m.parameters()) == int(facts[mname + ".params"][0])
        assert len(list(m.parameters())) == int(facts[mname + ".tensors"][0])
        keys = sorted(m.state_dict().keys())
        assert zlib.crc32("\n".join(keys).encode()) == int(facts[mname + ".keys_crc"][0])


# ---- DINOv2 entry points (first slice): goldens from the real OcticDinoVisionTransformer
import types as _types

import dino_cases


def _oracle_dino_ns():
    ns = _types.SimpleNamespace()
    ns.OcticDinoVisionTransformer = octic_ref.OcticDinoVisionTransformer
    ns.NestedTensorBlockD8 = octic_ref.NestedTensorBlockD8
    ns.DinoBlock = octic_ref.NestedTensorBlock
    return ns


@pytest.mark.parametrize("name", list(dino_cases.DINO_CASES))
def test_dino_case(name):
    got = dino_cases.run_dino_case(_oracle_dino_ns(), name)
    want = np.load(os.path.join(GOLD, name + ".npz"))
    assert set(got) == set(want.files), sorted(set(got) ^ set(want.files))
    for k in want.files:
        np.testing.assert_allclose(got[k], want[k], rtol=2e-4, atol=2e-4, err_msg=f"{name}:{k}")


def test_dino_list_forward_equals_per_crop_forward():
    """The reference's list path needs xformers (parity unpinned); what it computes is the per-crop forward."""
    m = dino_cases.build(_oracle_dino_ns(), dict(num_register_tokens=1, invariant=False))
    a, b = cases.randn("dino.list.a", 2, 3, 32, 32), cases.randn("dino.list.b", 3, 3, 32, 32)
    with torch.no_grad():
        outs = m.forward_features([a, b], [None, None])
        for o, x in zip(outs, (a, b)):
            ref = m.forward_features(x)
            for k in ("x_norm_clstoken", "x_norm_patchtokens", "x_prenorm"):
                torch.testing.assert_close(o[k], ref[k])
