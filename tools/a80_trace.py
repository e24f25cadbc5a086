"""Developer tool: phase timeline of the persistent head_dim-80 attention kernels (library built with -DA80_TRACE)."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octic_vits_amd import _lib, ops  # noqa: E402
from octic_vits_amd import functional as OF  # noqa: E402
_lib.LIB_PATH = os.environ.get("OCTIC_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "micro", "liboctic_a80trace.so"))
L = _lib.lib()
raw = ctypes.CDLL(_lib.LIB_PATH)
raw.octic_dbg_a80_trace.restype = ctypes.c_void_p
raw.octic_dbg_a80_variant(int(os.environ.get("A80_VARIANT", "0")))
B, H, T, hd = 64, 16, 257, 80
PACKED = os.environ.get("PACKED", "0") == "1"
if PACKED:
    c = 10 * H
    qkv = (torch.randn(B, T, 3 * 8 * c, device="cuda") * 0.7).bfloat16().requires_grad_(True)
    dop = torch.randn(B, T, 8 * c, device="cuda").bfloat16()
    def run():
        o = OF.AttnPackedFn.apply(qkv, H, c, hd ** -0.5)
        torch.autograd.grad(o, qkv, dop)
else:
    qkv = torch.randn(B, T, 3, H, hd, device="cuda").bfloat16().requires_grad_(True)
    do = torch.randn(B, T, H * hd, device="cuda").bfloat16()
    def run():
        o = OF.AttnFusedQKVFn.apply(qkv, hd ** -0.5)
        torch.autograd.grad(o, qkv, do)
for _ in range(2):
    run()
torch.cuda.synchronize()
n = 3 * 256 * 8 * 16
hip = ctypes.CDLL("libamdhip64.so")
src = raw.octic_dbg_a80_trace()
hip.hipMemset(ctypes.c_void_p(src), 0, n * 8)
run()
torch.cuda.synchronize()
buf = torch.zeros(n, dtype=torch.int64, device="cuda")
hip.hipMemcpy(ctypes.c_void_p(buf.data_ptr()), ctypes.c_void_p(src), n * 8, 3)
tr = buf.cpu().numpy().reshape(3, 256, 8, 16)
for kern, label in ((0, "fwd"), (1, "dq"), (2, "dkv")):
    t = tr[kern].reshape(-1, 16)
    t = t[t[:, 0] != 0]
    if len(t) == 0:
        print(label, ": not traced"); continue
    d = np.diff(t.astype(np.int64), axis=1)
    last = (t != 0).sum(1).min() - 1
    print(f"{label}: {len(t)} waves; mean cycles between stamps 0..{last}: " + " ".join(f"{d[:, i].mean():.0f}" for i in range(last)) + f" | total {np.mean(t[:, last] - t[:, 0]):.0f}")
