# developer A/B (GPU box): waves per workgroup of dense_ln_bwd_wide_kernel
cd /root/repo
python - <<'PY'
import subprocess, os
from octic_vits_amd import build as B
for w in (8, 12, 16):
    out = f"/root/repo/gpurun_out/liboctic_dw{w}.so"
    subprocess.check_call([B.HIPCC, *B.FLAGS, "-shared", f"-DOCTIC_DLNBWD_WAVES={w}", "-o", out] + [os.path.join(B.CSRC, s) for s in B.SOURCES])
PY
for w in 8 12 16 8 12; do echo "waves=$w"; OCTIC_LIB=/root/repo/gpurun_out/liboctic_dw$w.so python tools/bench_kernels.py 2>&1 | grep "dense_ln_bwd"; done
