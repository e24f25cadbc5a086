"""Host-side cost of one train step: run the flagship model at a tiny batch (GPU time negligible) under cProfile."""
import cProfile
import pstats
import sys
import time

import torch

sys.path.insert(0, ".")
from octic_vits_amd.deit_models import create_model
from octic_vits_amd.train import Trainer, synthetic_batch

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 2
model = create_model("hybrid_deit_huge_patch14", num_classes=1000, drop_path_rate=0.5, img_size=224).cuda()
tr = Trainer(model)
x, y = synthetic_batch(batch, 1000, "cuda", 1)
for _ in range(3):
    tr.step(x, y)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    tr.step(x, y)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"batch {batch}: eager issue {1e3 * (t1 - t0) / 5:.1f} ms/step, wall {1e3 * (t2 - t0) / 5:.1f} ms/step")
# the same step captured once and replayed as one hipGraph (Trainer.capture)
tr.check_every = 1 << 30            # no host read of the loss: pure launch cost
gs = tr.capture(x, y, warmup=1)
for _ in range(3):
    gs.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    gs.replay()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"batch {batch}: graph replay issue {1e3 * (t1 - t0) / 10:.2f} ms/step, wall {1e3 * (t2 - t0) / 10:.1f} ms/step")
if "--profile" not in sys.argv:
    sys.exit(0)
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    tr.step(x, y)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
