"""Developer probe: eager DDP step on a one-rank RCCL group - host issue time against wall time, with the small gradients
inside DDP (train.DDP_FLAT_SMALL_NUMEL = 0) and as one flat all-reduce (default).  usage: ddp_host_probe.py [numel]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29544")
import torch
from octic_vits_amd import train as TR
from octic_vits_amd.deit_models import create_model
from octic_vits_amd.train import Trainer, init_distributed, synthetic_batch

if len(sys.argv) > 1 and sys.argv[1].isdigit():
    TR.DDP_FLAT_SMALL_NUMEL = int(sys.argv[1])
init_distributed(force=True)
model = create_model("hybrid_deit_huge_patch14", num_classes=1000, drop_path_rate=0.5, img_size=224).cuda()
tr = Trainer(model, distributed=os.environ.get("PROBE_NO_DDP") is None, local_rank=0)
BATCH = int(os.environ.get("PROBE_BATCH", "64"))
x, y = synthetic_batch(BATCH, 1000, "cuda", 1)
for _ in range(4):
    tr.step(x, y)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(8):
    tr.step(x, y)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"DDP_FLAT_SMALL_NUMEL {TR.DDP_FLAT_SMALL_NUMEL}: batch {BATCH}, small tensors {len(getattr(tr, '_small', []))}; host issue {1e3 * (t1 - t0) / 8:.1f} ms/step, "
      f"wall {1e3 * (t2 - t0) / 8:.1f} ms/step")
if "--profile" in sys.argv:
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        tr.step(x, y)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
