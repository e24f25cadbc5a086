import sys, torch
sys.path.insert(0, "/root/repo")
from octic_vits_amd.deit_models import create_model
from octic_vits_amd.train import Trainer, synthetic_batch
from torch.profiler import profile, ProfilerActivity
model = create_model("hybrid_deit_huge_patch14", num_classes=1000, drop_path_rate=0.5, img_size=224).cuda()
tr = Trainer(model)
x, y = synthetic_batch(8, 1000, "cuda", 1)
for _ in range(3):
    tr.step(x, y)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    tr.step(x, y)
    torch.cuda.synchronize()
ev = prof.key_averages()
rows = sorted(ev, key=lambda e: -e.count)
for e in rows[:45]:
    print(f"{e.key[:70]:70s} n={e.count:5d} cpu={e.cpu_time_total/1e3:8.2f}ms dev={e.device_time_total/1e3:8.2f}ms")
