"""Developer tool: where a workgroup of the single-pass attention backward (csrc/attn80_bwd.hip) spends its cycles.
Builds a side library with -DA80_TRACE; prints per-phase cycle sums (mean over waves) per head."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from octic_vits_amd import build as Bd
out = os.path.join(ROOT, "gpurun_out", "liboctic_a80bwd.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call([Bd.HIPCC, *Bd.FLAGS, "-shared", "-DA80_TRACE", "-o", out] + [os.path.join(Bd.CSRC, s) for s in Bd.SOURCES])
import numpy as np
import torch
from octic_vits_amd import _lib
_lib.LIB_PATH = out
from octic_vits_amd import ops, functional as OF
L = _lib.lib()
raw = ctypes.CDLL(out)
raw.octic_dbg_a80_bwd_trace.restype = ctypes.c_void_p
hip = ctypes.CDLL("libamdhip64.so")
B, H, T, hd = 64, 16, 257, 80
names = ["prologue", "-", "S' dP", "softmax+dS", "dV dK", "dQ blocks", "delta+k256", "wait+barrier", "stores+DMA", "epilogue", "total"]
for packed in (False, True):
    if packed:
        c = 10 * H
        qkv = (torch.randn(B, T, 3 * 8 * c, device="cuda") * 0.7).bfloat16().requires_grad_(True)
        dop = torch.randn(B, T, 8 * c, device="cuda").bfloat16()
        o = OF.AttnPackedFn.apply(qkv, H, c, hd ** -0.5)
        run = lambda: torch.autograd.grad(o, qkv, dop, retain_graph=True)
    else:
        qkv = torch.randn(B, T, 3, H, hd, device="cuda").bfloat16().requires_grad_(True)
        do = torch.randn(B, T, H * hd, device="cuda").bfloat16()
        o = OF.AttnFusedQKVFn.apply(qkv, hd ** -0.5)
        run = lambda: torch.autograd.grad(o, qkv, do, retain_graph=True)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    n = 1024 * 8 * 16
    src = raw.octic_dbg_a80_bwd_trace()
    hip.hipMemset(ctypes.c_void_p(src), 0, n * 8)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record()
    torch.cuda.synchronize()
    buf = torch.zeros(n, dtype=torch.int64, device="cuda")
    hip.hipMemcpy(ctypes.c_void_p(buf.data_ptr()), ctypes.c_void_p(src), n * 8, 3)
    tr = buf.cpu().numpy().reshape(1024, 8, 16)[:, :, :11].astype(np.float64)
    print(("packed rows" if packed else "strided rows") + f": event {e0.elapsed_time(e1) * 1e3:.0f} us (traced build)")
    m = tr.reshape(-1, 11).mean(0)
    print("  " + "  ".join(f"{nm} {v:.0f}" for nm, v in zip(names, m)))
    w = tr.mean(0)
    for wv in range(8):
        print(f"   wave {wv}: " + " ".join(f"{v:7.0f}" for v in w[wv]))
