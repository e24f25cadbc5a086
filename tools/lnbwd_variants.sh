cd /root/repo
python - <<'PY'
import subprocess, os
from octic_vits_amd import build as B
for tag, flags in (("w0", ["-DOCTIC_LNBWD_WIDE=0"]), ("w1cap256", ["-DOCTIC_LNBWD_CAP=256"])):
    out = f"/root/repo/gpurun_out/liboctic_{tag}.so"
    subprocess.check_call([B.HIPCC, *B.FLAGS, "-shared", *flags, "-o", out] + [os.path.join(B.CSRC, s) for s in B.SOURCES])
PY
for lib in "" /root/repo/gpurun_out/liboctic_w0.so /root/repo/gpurun_out/liboctic_w1cap256.so; do echo "lib=$lib"; OCTIC_LIB=$lib python tools/bench_kernels.py 2>&1 | grep "ln_bwd"; done
