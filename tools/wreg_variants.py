"""Developer tool: A/B builds of csrc/gemm_wreg.hip with different -D switches (ablations, ring depth), benchmarked back to back on
one device.    python tools/wreg_variants.py "base:" "nostore:-DOCTIC_WREG_ABL=2" ...   """
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "octic_vits_amd", "csrc")
SRC = os.environ.get("SRC", "gemm_wreg")    # SRC=gemm: variants of csrc/gemm.hip (ring kernels; CHECK_ARGS=--long)
outdir = os.path.join(ROOT, "tools", "micro", "variants")
os.makedirs(outdir, exist_ok=True)
objs = [os.path.join(CS, "build", f) for f in os.listdir(os.path.join(CS, "build")) if f.endswith(".o") and f != SRC + ".o"]
rounds = int(os.environ.get("ROUNDS", "2"))
specs = [a.split(":", 1) for a in sys.argv[1:]]
libs = []
for name, flags in specs:
    o = os.path.join(outdir, f"wr_{name}.o")
    so = os.path.join(outdir, f"lib_{name}.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-c",
                           os.path.join(CS, SRC + ".hip"), "-o", o] + flags.split())
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, o] + objs)
    libs.append((name, so))
for r in range(rounds):
    for name, so in libs:
        env = dict(os.environ, OCTIC_LIB=so)
        if os.environ.get("RUN"):
            res = subprocess.run([sys.executable] + os.environ["RUN"].split(), env=env, capture_output=True, text=True)
            flt = os.environ.get("GREP", "")
            out = [ln for ln in res.stdout.splitlines() if flt in ln]
            print(f"round {r} {name:12s} " + " | ".join(" ".join(ln.split()) for ln in out) + (res.stderr[-300:] if res.returncode else ""), flush=True)
            continue
        if name.startswith("trace"):
            if r == 0:
                res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ring_trace.py" if SRC == "gemm" else "wreg_trace.py")] + os.environ.get("TRACE_ARGS", "").split(), env=env, capture_output=True, text=True)
                print(f"---- {name}\n{res.stdout}{res.stderr[-400:] if res.returncode else ''}", flush=True)
            continue
        res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "wreg_check.py")] + os.environ.get("CHECK_ARGS", "--bench").split(), env=env, capture_output=True, text=True)
        us = [ln.split("wreg")[-1].split("us")[0].strip() for ln in res.stdout.splitlines() if " us " in ln]
        print(f"round {r} {name:14s} qkv/fc1/proj+res/dgrad-fc2/proj-bf16: {' '.join(us)} {res.stderr.strip()[-200:] if res.returncode not in (0, 1) else ''}", flush=True)
