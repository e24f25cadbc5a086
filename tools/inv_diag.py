"""Developer diagnostic: why are B1-irrep parameter gradients of the small invariant model 16-30 % off under bf16 autocast?"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402
from oracle import octic_ref as R  # noqa: E402


def run(ns, dev, autocast):
    case = cases.CASES["model_invariant"]
    mod = cases.fill_parameters(cases.build_model(ns, case["model"])).to(dev).eval()
    x = cases.make_input("model_invariant", case["inp"]).to(dev)
    grabbed = {}

    def hook(m, inp, out):
        grabbed["inv_in"] = inp[0]
        grabbed["inv_out"] = out
        for t in (inp[0] if isinstance(inp[0], (tuple, list)) else [inp[0]]):
            if t.requires_grad:
                t.retain_grad()
    mod.invariantization.register_forward_hook(hook)
    ctx = torch.autocast(dev, dtype=torch.bfloat16) if autocast else torch.autocast(dev, enabled=False)
    with ctx:
        out = mod(x)
    cot = cases.randn("model_invariant.cot.0", *out.shape).to(dev)
    (out.float() * cot).sum().backward()
    xs = grabbed["inv_in"]
    res = {"out": out.detach().float().cpu()}
    for i, t in enumerate(xs):
        res[f"x{i}"] = t.detach().float().cpu()
        res[f"g{i}"] = None if t.grad is None else t.grad.detach().float().cpu()
    return res


def main():
    from test_modules_gpu import product_ns
    ref = run(R, "cpu", False)
    oac = run(R, "cpu", True)
    got = run(product_ns(), "cuda", True)
    for i in range(5):
        a, b, c = ref[f"x{i}"], oac[f"x{i}"], got[f"x{i}"]
        print(f"irrep {i}: |x| mean {a.abs().mean():.4f}  oracle-ac err {float((b - a).abs().mean()):.5f}  product err {float((c - a).abs().mean()):.5f}"
              f"  sign flips oracle-ac {float((torch.sign(b) != torch.sign(a)).float().mean()):.4f} product {float((torch.sign(c) != torch.sign(a)).float().mean()):.4f}")
        if ref[f"g{i}"] is not None and got[f"g{i}"] is not None:
            ga, gb, gc = ref[f"g{i}"], oac[f"g{i}"], got[f"g{i}"]
            rel = lambda u, v: float((u - v).norm() / v.norm())
            print(f"          grad rel err oracle-ac {rel(gb, ga) if gb is not None else -1:.4f}  product {rel(gc, ga):.4f}")
        else:
            print("          (no grad captured)", ref[f"g{i}"] is None, got[f"g{i}"] is None)


if __name__ == "__main__":
    main()
