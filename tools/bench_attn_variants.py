"""Developer A/B: attention forward/backward at the ViT-H shape for several builds of the library (OCTIC_LIBS=path,path)
and a80 kernel variants (octic_route_override, OCTIC_ROUTE_ATTN_ONLINE), each in its own process section."""
import ctypes, os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from octic_vits_amd import _lib
    _lib.LIB_PATH = sys.argv[2]
    from octic_vits_amd import ops
    raw = ctypes.CDLL(_lib.LIB_PATH)
    def timeit(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    B, H, T, hd = 64, 16, 257, 80
    qkv = torch.randn(B, T, 3, H, hd, device="cuda").bfloat16()
    q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    c = 160
    pk = (torch.randn(B, T, 3 * 8 * c, device="cuda") * 0.7).bfloat16()
    out = []
    for var in [int(x) for x in sys.argv[3].split(",")]:
        if hasattr(raw, "octic_route_override"):
            raw.octic_route_override(7, var)
        tf = min(timeit(lambda: ops.attn_fwd(q, k, v, hd ** -0.5)) for _ in range(3))
        tp = min(timeit(lambda: ops.attn_fwd_packed(pk, H, c, hd ** -0.5)) for _ in range(3))
        out.append(f"var{var}: fused-qkv {tf:6.1f} us  packed {tp:6.1f} us")
    print(os.path.basename(sys.argv[2]), " | ".join(out), flush=True)
else:
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libs = os.environ.get("OCTIC_LIBS", os.path.join(root, "octic_vits_amd", "liboctic_hip.so")).split(",")
    for lib in libs:
        subprocess.run([sys.executable, os.path.abspath(__file__), "child", lib, os.environ.get("A80_VARIANTS", "0,1")])
