"""Generate tests/golden/train_hybrid.npz by RUNNING THE REAL REFERENCE MODEL on CPU (build container only) through
three optimizer steps of the recipe.  The optimizer is the numpy restatement of apex FusedLAMB (oracle/lamb_ref.py):
apex itself is not part of /root/reference, so the optimizer half of this fixture is "parity unpinned" (see the
header of oracle/lamb_ref.py); the model half (losses / gradients entering the optimizer) is the reference's own.

    python tests/golden/make_train_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import train_case  # noqa: E402
from make_golden import reference_namespace  # noqa: E402

if __name__ == "__main__":
    torch.set_num_threads(8)
    res = train_case.run_train_case(reference_namespace())
    path = os.path.join(HERE, "train_hybrid.npz")
    np.savez_compressed(path, **res)
    print("losses", res["losses"], "grad norms", res["grad_norms"], f"{os.path.getsize(path) / 1024:.1f} KiB")
