"""Developer tool: A/B builds of csrc/gemm.hip with different -D switches (ablations), benchmarked back to back on one device.
    python tools/dense_variants.py "name1:-DOCTIC_XREG_ABL=1" "name2:..."   """
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "octic_vits_amd", "csrc")
outdir = os.path.join(ROOT, "tools", "micro", "variants")
os.makedirs(outdir, exist_ok=True)
objs = [os.path.join(CS, "build", f) for f in os.listdir(os.path.join(CS, "build")) if f.endswith(".o") and f != "gemm.o"]
rounds = int(os.environ.get("ROUNDS", "2"))
specs = [a.split(":", 1) for a in sys.argv[1:]]
libs = []
for name, flags in specs:
    o = os.path.join(outdir, f"xr_{name}.o")
    so = os.path.join(outdir, f"lib_{name}.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-c",
                           os.path.join(CS, "gemm.hip"), "-o", o] + flags.split())
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, o] + objs)
    libs.append((name, so))
for r in range(rounds):
    for name, so in libs:
        env = dict(os.environ, OCTIC_LIB=so)
        res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_xreg.py")], env=env, capture_output=True, text=True)
        res2 = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_ring.py")], env=env, capture_output=True, text=True)
        res.stdout = res.stdout.strip() + "   " + res2.stdout.strip()
        print(f"round {r} {name}:\n{res.stdout.strip()} {res.stderr.strip()[-200:] if res.returncode else ''}", flush=True)
