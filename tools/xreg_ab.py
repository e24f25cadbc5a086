"""Developer A/B: prebuilt variant libraries (tools/micro/variants/lib_*.so) of csrc/gemm.hip on the X-stationary shapes."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sorted(glob.glob(os.path.join(ROOT, "tools", "micro", "variants", "lib_*.so")))
for r in range(int(os.environ.get("ROUNDS", "2"))):
    for so in libs:
        env = dict(os.environ, OCTIC_LIB=so)
        res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_xreg.py")], env=env, capture_output=True, text=True)
        print(f"round {r} {os.path.basename(so)[4:-3]:18s} {res.stdout.strip()} {res.stderr.strip()[-200:] if res.returncode else ''}", flush=True)
