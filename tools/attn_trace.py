"""Developer tool: phase timeline of the attention kernels (library built with -DOCTIC_ATTN_TRACE)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octic_vits_amd import _lib, ops  # noqa: E402
from octic_vits_amd import functional as OF  # noqa: E402

_lib.LIB_PATH = os.environ.get("OCTIC_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "micro", "liboctic_trace.so"))
L = _lib.lib()
L.octic_dbg_attn_trace.restype = ctypes.c_void_p
B, H, T, hd = int(os.environ.get("B", "64")), 16, 257, 80
PACKED = os.environ.get("PACKED", "0") == "1"       # AttentionD8 core on packed rows (octic_attn_*_packed)
if PACKED:
    c = 10 * H
    qkv = (torch.randn(B, T, 3 * 8 * c, device="cuda") * 0.7).bfloat16().requires_grad_(True)
    dop = torch.randn(B, T, 8 * c, device="cuda").bfloat16()

    def run():
        o = OF.AttnPackedFn.apply(qkv, H, c, hd ** -0.5)
        torch.autograd.grad(o, qkv, dop)
else:
    q, k, v = (torch.randn(B, H, T, hd, device="cuda").bfloat16().requires_grad_(True) for _ in range(3))
    do = torch.randn(B, H, T, hd, device="cuda").bfloat16()

    def run():
        o = OF.AttnFn.apply(q, k, v, hd ** -0.5)
        torch.autograd.grad(o, (q, k, v), do)
for _ in range(2):
    run()
torch.cuda.synchronize()
n = 3 * 256 * 10 * 16
hip = ctypes.CDLL("libamdhip64.so")
src = L.octic_dbg_attn_trace()
hip.hipMemset(ctypes.c_void_p(src), 0, n * 8)
run()
torch.cuda.synchronize()
buf = torch.zeros(n, dtype=torch.int64, device="cuda")
hip.hipMemcpy(ctypes.c_void_p(buf.data_ptr()), ctypes.c_void_p(src), n * 8, 3)
tr = buf.cpu().numpy().reshape(3, 256, 10, 16)
names = {0: ["wait q + issue", "main pass", "O store + q load", "shared pass", "barrier A", "write V", "vmcnt(0)", "barrier B", "merge"], 1: ["stage issue", "barrier", "main pass", "store", "shared pass", "barrier", "combine"],
         2: ["stage issue", "barrier", "main pass", "store", "shared pass", "barrier", "combine"]}
for kern, label in ((0, "fwd"), (1, "dq"), (2, "dkv")):
    rows = []
    for wg in range(256):
        for w in range(10):
            t = tr[kern, wg, w]
            if t[0] == 0:
                continue
            nz = np.nonzero(t)[0].max()
            rows.append([t[i + 1] - t[i] if i + 1 <= nz else 0 for i in range(len(names[kern]))] + [t[nz] - t[0]])
    rows = np.array(rows)
    if rows.ndim != 2:
        print(f"{label}: no waves traced")
        continue
    print(f"{label}: {len(rows)} waves; mean cycles per phase:")
    for i, nm in enumerate(names[kern]):
        print(f"   {nm:14s} {rows[:, i].mean():9.0f}")
    print(f"   {'wave total':14s} {rows[:, -1].mean():9.0f}")
