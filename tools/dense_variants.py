"""Developer tool: A/B builds of csrc/dense_gemm.hip with different -D switches, benchmarked back to back on one device.
    python tools/dense_variants.py "name1:-DDG_X=1 -DDG_Y=2" "name2:..."   """
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "octic_vits_amd", "csrc")
outdir = os.path.join(ROOT, "gpurun_out", "variants")
os.makedirs(outdir, exist_ok=True)
objs = [os.path.join(CS, "build", f) for f in os.listdir(os.path.join(CS, "build")) if f.endswith(".o") and f != "dense_gemm.o"]
rounds = int(os.environ.get("ROUNDS", "2"))
specs = [a.split(":", 1) for a in sys.argv[1:]]
libs = []
for name, flags in specs:
    o = os.path.join(outdir, f"dg_{name}.o")
    so = os.path.join(outdir, f"lib_{name}.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-c",
                           os.path.join(CS, "dense_gemm.hip"), "-o", o] + flags.split())
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, o] + objs)
    libs.append((name, so))
code = r'''
import sys, os, torch
sys.path.insert(0, %r)
from octic_vits_amd import ops
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M = 16448
out = []
for (N, K) in [(3840, 1280), (1280, 1280), (5120, 1280), (1280, 5120), (1280, 3840)]:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    b = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    t = timeit(lambda: ops.dense_gemm_nt(a, b, 0))
    out.append("%%6.1f" %% (2.0 * M * N * K / t / 1e6))
if os.environ.get("TAILS"):            # fused tails at the MLP shape: us of fc1 + GELU and of the fc2 input gradient x GELU'
    a = torch.randn(M, 1280, device="cuda").to(torch.bfloat16)
    b = (torch.randn(5120, 1280, device="cuda") * 1280 ** -0.5).to(torch.bfloat16)
    h = torch.randn(M, 5120, device="cuda").to(torch.bfloat16)
    bias = torch.randn(5120, device="cuda")
    out.append("| gelu %%6.1f us" %% timeit(lambda: ops.dense_gemm_nt(a, b, 1, bias=bias)))
    out.append("dgelu %%6.1f us" %% timeit(lambda: ops.dense_gemm_nt(a, b, 3, h=h, want_colsum=True)))
    out.append("plain %%6.1f us" %% timeit(lambda: ops.dense_gemm_nt(a, b, 0, bias=bias)))
print(" ".join(out))
''' % ROOT
for r in range(rounds):
    for name, so in libs:
        env = dict(os.environ, OCTIC_LIB=so)
        res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print(f"round {r} {name:24s} TF [qkv proj fc1 fc2 dqkv]: {res.stdout.strip()} {res.stderr.strip()[-200:] if res.returncode else ''}", flush=True)
