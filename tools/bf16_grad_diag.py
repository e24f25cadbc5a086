"""Developer diagnostic: per-tensor relative L2 error of the product's bf16-autocast model gradients against
(a) the f32 golden of the real reference, (b) the oracle run under torch.autocast('cpu', bfloat16) (= what autocast does to
the reference, same bf16 rounding points), (c) the oracle in f32 with bf16-rounded parameters."""
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


class AC(torch.nn.Module):
    def __init__(self, mod, dev):
        super().__init__()
        self.mod, self.dev = mod, dev

    def named_parameters(self, *a, **k):
        return self.mod.named_parameters(*a, **k)

    def forward(self, x):
        with torch.autocast(self.dev, dtype=torch.bfloat16):
            return self.mod(x)


def main():
    from test_modules_gpu import product_ns
    for name in ("model_hybrid", "model_invariant"):
        got = cases.run_module_case(product_ns(), name, device="cuda", to_module=lambda m: AC(m, "cuda"))
        gold = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
        oac = cases.run_module_case(R, name, device="cpu", to_module=lambda m: AC(m, "cpu"))
        rows = []
        for k in gold.files:
            if not (k.startswith("gpar_sample") or k.startswith("gpar.")):
                continue
            g, w, a = got[k].astype(np.float64), gold[k].astype(np.float64), oac[k].astype(np.float64)
            rel = lambda x, y: np.linalg.norm(x - y) / max(np.linalg.norm(y), 1e-12)
            rows.append((rel(g, w), rel(g, a), rel(a, w), k))
        rows.sort(reverse=True)
        print(f"== {name}: product-bf16 vs f32 golden | product vs oracle-cpu-autocast | oracle-autocast vs golden")
        for r in rows[:15]:
            print(f"  {r[0]:.4f}  {r[1]:.4f}  {r[2]:.4f}  {r[3]}")
        print("  median", np.median([r[0] for r in rows]), np.median([r[1] for r in rows]), np.median([r[2] for r in rows]))
        for k in gold.files:
            if k.startswith("out."):
                print("  out err vs golden", np.abs(got[k] - gold[k]).max(), "scale", np.abs(gold[k]).max(),
                      "| oracle-autocast", np.abs(oac[k] - gold[k]).max())


if __name__ == "__main__":
    main()
