"""Build liboctic_hip.so (gfx950) in-tree with hipcc.  No torch headers: the library is a plain C ABI.

    python -m octic_vits_amd.build        # or: from octic_vits_amd.build import build; build()
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "liboctic_hip.so")
SOURCES = ["elementwise.hip", "layernorm.hip", "gemm.hip", "gemm_wreg.hip", "wgrad.hip", "lamb.hip", "attention.hip", "attn80.hip", "attn80_bwd.hip", "dense.hip", "dense_gemm.hip", "dense_wgrad.hip", "ssl_loss.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(os.path.join(CSRC, "build"), exist_ok=True)
    headers = [os.path.join(CSRC, "octic_common.hpp"), os.path.join(CSRC, "attn_common.hpp"), os.path.join(CSRC, "attn80_common.hpp"), os.path.join(CSRC, "gemm_args.hpp"),
               os.path.join(HERE, "..", "include", "octic_hip.h")]
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, "build", src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([HIPCC, *FLAGS, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=4) as ex:
            list(ex.map(run, jobs))
    if force or jobs or _stale(OUT, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs])
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
