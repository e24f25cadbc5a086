"""Host profile of the eager train step with stochastic depth as batch compaction (d8_layers.COMPACT_DROP_PATH)."""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from octic_vits_amd import d8_layers as L
from octic_vits_amd.deit_models import create_model
from octic_vits_amd.train import Trainer, synthetic_batch
L.COMPACT_DROP_PATH = "--full" not in sys.argv
L.USE_SAMPLE_BLOCKS = "--aten" not in sys.argv
m = create_model("hybrid_deit_huge_patch14", num_classes=1000, drop_path_rate=0.5, img_size=224).cuda()
tr = Trainer(m, check_every=1000)
x, y = synthetic_batch(64, 1000, "cuda", 1)
W = int(os.environ.get("WARM", "50"))
for _ in range(W):
    tr.step(x, y)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    tr.step(x, y)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"10 steps: host issue {(t1 - t0) * 100:.1f} ms/step, wall {(t2 - t0) * 100:.1f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    tr.step(x, y)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
