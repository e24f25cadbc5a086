"""Developer tool: s_memtime timeline of workgroup 0 of the dense GEMM (build with -DDG_TRACE into a side library)."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CS = os.path.join(ROOT, "octic_vits_amd", "csrc")
out = os.path.join(ROOT, "gpurun_out", "liboctic_trace.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
srcs = ["elementwise.hip", "layernorm.hip", "gemm.hip", "wgrad.hip", "lamb.hip", "attention.hip", "dense.hip", "dense_gemm.hip"]
variant = sys.argv[1] if len(sys.argv) > 1 else "10"
out = out.replace(".so", f"_{variant}.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DDG_TRACE",
                       f"-DDG_NSLOT={variant}", "-Wno-unused-value", "-o", out] + [os.path.join(CS, s) for s in srcs])
os.environ["OCTIC_LIB"] = out
import torch
from octic_vits_amd import ops, _lib
L = _lib.lib()
L.octic_dbg_dense_trace.restype = ctypes.c_void_p
M, N, K = 16448, 5120, 1280
a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
b = (torch.randn(N, K, device="cuda") / 36).to(torch.bfloat16)
for _ in range(3):
    ops.dense_gemm_nt(a, b, 0)
torch.cuda.synchronize()
ptr = L.octic_dbg_dense_trace()
buf = (ctypes.c_ulonglong * (8 * 256))()
import ctypes as C
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpy(buf, C.c_void_p(ptr), 8 * 256 * 8, 2)
import numpy as np
t = np.array(buf[:], dtype=np.int64).reshape(8, 256)
print("variant NSLOT =", variant)
for w in range(8):
    d = np.diff(t[w][:200])
    x = d[1:1 + 196].reshape(-1, 4)
    real = (t[w][255] - t[w][254]) / 100e6      # seconds (100 MHz counter)
    clk = (t[w][199] - t[w][0]) / real / 1e9 if real > 0 else 0
    print(f"wave {w}: prologue {d[0]:5d}  mean per phase R {x[:,0].mean():6.0f} sync {x[:,1].mean():6.0f} M {x[:,2].mean():6.0f} post {x[:,3].mean():6.0f}"
          f"  total/phase {x.sum(1).mean():6.0f}  s_memtime clock {clk:.2f} GHz")
