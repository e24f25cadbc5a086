"""Developer A/B: variants of csrc/dense_wgrad.hip (tools/variants/tn_*.so, built with -D flags) on the ViT-H weight-gradient
shapes, interleaved rounds in one process, plus a correctness check of each variant against torch."""
import ctypes, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

def timeit(fn, n=8):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "variants")
names = sys.argv[1].split(",") if len(sys.argv) > 1 else sorted(os.path.basename(f)[3:-3] for f in glob.glob(here + "/tn_*.so"))
Ss = [int(s) for s in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["0"])]
shapes = [(5120, 1280), (1280, 5120), (3840, 1280), (1280, 1280)]
libs = {}
for n in names:
    L = ctypes.CDLL(os.path.join(here, f"tn_{n}.so"))
    L.octic_dense_wgrad_workspace_bytes.restype = ctypes.c_int64
    libs[n] = L
M = 16448
vp = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for (N, K) in shapes:
    dy = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    xx = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    want = (dy.t().float() @ xx.float())
    fl = 2.0 * M * N * K
    for S in Ss:
        res = {n: [] for n in names}
        ws, dw = {}, {}
        for n, L in libs.items():
            L.octic_route_override(2, S)
            ws[n] = torch.zeros(int(L.octic_dense_wgrad_workspace_bytes(M, N, K)), dtype=torch.uint8, device="cuda")
            dw[n] = torch.empty(N, K, device="cuda")
        def run(n):
            rc = libs[n].octic_dense_wgrad_tn(vp(dy), vp(xx), M, N, K, ctypes.c_int64(N), ctypes.c_int64(K), vp(dw[n]), vp(ws[n]), st)
            assert rc == 0, rc
        for n in names:
            run(n); run(n)
            err = float((dw[n] - want).abs().max() / want.abs().max())
            if err >= 2e-3: print(f'   WRONG: {n} err {err:.3g}', flush=True)
        for rnd in range(3):
            for n in names:
                res[n].append(timeit(lambda: run(n)))
        print(f"dW {N:5d}x{K:5d} S={S or 'auto'}: " + "  ".join(f"{n} {min(v):6.1f} ({fl / min(v) / 1e6:4.0f})" for n, v in res.items()), flush=True)
