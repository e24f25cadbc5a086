"""Developer micro-benchmark: standard-half row kernels called straight through the C ABI with preallocated buffers
(no host allocation between launches), ViT-H shapes.  OCTIC_LIB selects the library build."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octic_vits_amd import _lib
if os.environ.get("OCTIC_LIB"):
    _lib.LIB_PATH = os.environ["OCTIC_LIB"]
from octic_vits_amd import ops
from octic_vits_amd.ops import _p, _stream, check, lib, dt_code

def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

M, d = 16448, 1280
bf = torch.bfloat16
gout = torch.randn(M, d, device="cuda")
y = torch.randn(M, d, device="cuda").to(bf)
gamma = torch.rand(d, device="cuda")
rs = torch.ones(64, device="cuda")
gy = torch.empty_like(y)
nblk = lib().octic_dense_blocks(M)
partials = torch.empty(nblk, 2, d, device="cuda")
t = timeit(lambda: check(lib().octic_scale_residual_bwd(_p(gout), _p(y), dt_code(bf), _p(gamma), _p(rs), 257, _p(gy), _p(partials), M, d, _stream(gout))))
print(f"scale_residual_bwd  {t:6.1f} us  {M * d * 8 / t / 1e3:7.1f} GB/s")
x = torch.randn(M, d, device="cuda")
out = torch.empty_like(x)
t = timeit(lambda: check(lib().octic_scale_residual_fwd(_p(x), _p(y), dt_code(bf), _p(gamma), _p(rs), 257, _p(out), M, d, _stream(x))))
print(f"scale_residual_fwd  {t:6.1f} us  {M * d * 10 / t / 1e3:7.1f} GB/s")
w = torch.rand(d, device="cuda"); b = torch.rand(d, device="cuda")
yb = torch.empty(M, d, device="cuda", dtype=bf); stats = torch.empty(M, 2, device="cuda")
t = timeit(lambda: check(lib().octic_dense_layernorm_fwd(_p(x), _p(yb), dt_code(bf), _p(w), _p(b), _p(stats), M, d, 1e-6, _stream(x))))
print(f"dense_ln_fwd        {t:6.1f} us  {M * d * 6 / t / 1e3:7.1f} GB/s")
dx = torch.empty_like(x)
t = timeit(lambda: check(lib().octic_dense_layernorm_bwd(_p(y), dt_code(bf), _p(x), _p(w), _p(stats), _p(gout), _p(dx), _p(partials), M, d, _stream(x))))
print(f"dense_ln_bwd +dres  {t:6.1f} us  {M * d * 14 / t / 1e3:7.1f} GB/s")
h = torch.randn(M, 4 * d, device="cuda").to(bf); g = torch.randn(M, 4 * d, device="cuda").to(bf); dh = torch.empty_like(h)
pg = torch.empty(lib().octic_dense_gelu_blocks(), 4 * d, device="cuda")
t = timeit(lambda: check(lib().octic_dense_gelu_bwd(_p(h), _p(g), _p(dh), _p(pg), M, 4 * d, _stream(h))))
print(f"dense_gelu_bwd      {t:6.1f} us  {M * 4 * d * 6 / t / 1e3:7.1f} GB/s")
