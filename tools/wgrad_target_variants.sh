# developer A/B (GPU box): work-group target of the octic weight-gradient split (slab traffic vs parallelism)
cd /root/repo
python - <<'PY'
import subprocess, os
from octic_vits_amd import build as B
for t in (256, 384, 512, 768):
    out = f"/root/repo/gpurun_out/liboctic_wg{t}.so"
    subprocess.check_call([B.HIPCC, *B.FLAGS, "-shared", f"-DOCTIC_WG_TARGET={t}.0", "-o", out] + [os.path.join(B.CSRC, s) for s in B.SOURCES])
PY
for t in 512 256 384 768 512 256; do echo "target=$t"; OCTIC_LIB=/root/repo/gpurun_out/liboctic_wg$t.so python tools/bench_wgrad.py 2>&1 | tail -1; done
