set -e
cd /root/repo
python - <<'PY'
import subprocess, os
from octic_vits_amd import build as B
out = "/root/repo/gpurun_out/liboctic_attn_trace.so"
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call([B.HIPCC, *B.FLAGS, "-shared", "-DOCTIC_ATTN_TRACE", "-o", out] + [os.path.join(B.CSRC, s) for s in B.SOURCES])
PY
OCTIC_LIB=/root/repo/gpurun_out/liboctic_attn_trace.so timeout 300 python tools/attn_trace.py 2>&1 | grep -v "^/opt" | tail -32
