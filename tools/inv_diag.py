"""Developer diagnostic: where do the B1-irrep gradients of the small invariant model pick up their bf16 error?
Captures the cotangents on both sides of the PowerSpectrum hand-off for: oracle f32, oracle under CPU bf16 autocast, product
under CUDA bf16 autocast."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402
from oracle import octic_ref as R  # noqa: E402


def run(ns, dev, autocast, product):
    case = cases.CASES["model_invariant"]
    mod = cases.fill_parameters(cases.build_model(ns, case["model"])).to(dev).eval()
    x = cases.make_input("model_invariant", case["inp"]).to(dev)
    got = {}
    inv = mod.invariantization
    orig = inv.forward

    def fwd(xs, *a, **k):
        if product:
            from octic_vits_amd.functional import as_packed, Octic
            xp, c = as_packed(xs)
            xp = xp.clone() if not xp.requires_grad else xp * 1.0
            xp.register_hook(lambda g: got.__setitem__("gx", g.detach().float().cpu()))
            got["x"] = xp.detach().float().cpu()
            out = orig(Octic(xp, c), *a, **k)
        else:
            xs = tuple(t * 1.0 for t in xs)
            packed = [None] * 5
            for i, t in enumerate(xs):
                t.register_hook(lambda g, i=i: packed.__setitem__(i, g.detach().float().cpu()))
            got["_parts"] = packed
            got["x"] = torch.cat([t.detach().float().flatten(2) for t in xs], -1).cpu()
            out = orig(xs, *a, **k)
        out.register_hook(lambda g: got.__setitem__("gout", g.detach().float().cpu()))
        got["out"] = out.detach().float().cpu()
        return out
    inv.forward = fwd
    ctx = torch.autocast(dev, dtype=torch.bfloat16) if autocast else torch.autocast(dev, enabled=False)
    with ctx:
        out = mod(x)
    cot = cases.randn("model_invariant.cot.0", *out.shape).to(dev)
    (out.float() * cot).sum().backward()
    if "_parts" in got:
        got["gx"] = torch.cat([t.flatten(2) for t in got.pop("_parts")], -1)
    got["logits"] = out.detach().float().cpu()
    return got


def main():
    from test_modules_gpu import product_ns
    ref = run(R, "cpu", False, False)
    oac = run(R, "cpu", True, False)
    got = run(product_ns(), "cuda", True, True)
    c = ref["x"].shape[-1] // 8
    rel = lambda u, v: float((u - v).norm() / v.norm().clamp_min(1e-20))
    print("cotangent at the invariant OUTPUT (6c): rel err vs f32: oracle-ac %.4f product %.4f" % (rel(oac["gout"], ref["gout"]), rel(got["gout"], ref["gout"])))
    segs = [("A1", 0, c), ("|A2|", c, 2 * c), ("|B1|", 2 * c, 3 * c), ("|B2|", 3 * c, 4 * c), ("|E|", 4 * c, 6 * c)]
    for n, a, b in segs:
        print(f"   out seg {n:5s}: |g| {float(ref['gout'][..., a:b].norm()):.4e}  oracle-ac {rel(oac['gout'][..., a:b], ref['gout'][..., a:b]):.4f} product {rel(got['gout'][..., a:b], ref['gout'][..., a:b]):.4f}")
    isegs = [("A1", 0, c), ("A2", c, 2 * c), ("B1", 2 * c, 3 * c), ("B2", 3 * c, 4 * c), ("E", 4 * c, 8 * c)]
    for n, a, b in isegs:
        print(f"   stream {n:3s}: |x| {float(ref['x'][..., a:b].norm()):.4e} x err oracle-ac {rel(oac['x'][..., a:b], ref['x'][..., a:b]):.4f} product {rel(got['x'][..., a:b], ref['x'][..., a:b]):.4f}"
              f" | |gx| {float(ref['gx'][..., a:b].norm()):.4e} gx err oracle-ac {rel(oac['gx'][..., a:b], ref['gx'][..., a:b]):.4f} product {rel(got['gx'][..., a:b], ref['gx'][..., a:b]):.4f}")
    # is the product's dx consistent with ITS OWN g and x (the kernel's math)?
    g, x = got["gout"], got["x"]
    want = torch.cat([g[..., :c], g[..., c:4 * c] * torch.sign(x[..., c:4 * c]),
                      (x[..., 4 * c:].unflatten(-1, (2, 2 * c)) * (g[..., 4 * c:] / x[..., 4 * c:].unflatten(-1, (2, 2 * c)).norm(dim=-2)).unsqueeze(-2)).flatten(-2)], -1)
    print("product: kernel dx vs formula on its own (g, x): rel %.3e" % rel(got["gx"], want))


if __name__ == "__main__":
    main()
