"""Developer check: per-step wall times of the bench loop (graph replay with eager kernel-timing steps in between)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octic_vits_amd import ops
from octic_vits_amd.deit_models import create_model
from octic_vits_amd.train import Trainer, synthetic_batch

torch.manual_seed(1337)
model = create_model("hybrid_deit_huge_patch14", num_classes=1000, drop_path_rate=0.5, img_size=224).cuda()
tr = Trainer(model)
x, y = synthetic_batch(64, 1000, "cuda", 4242)
for _ in range(5):
    tr.step(x, y)
gs = tr.capture(x, y, warmup=1)
for _ in range(2):
    gs.replay()
torch.cuda.synchronize()
print("reserved GB", torch.cuda.memory_reserved() / 2**30, "allocated GB", torch.cuda.memory_allocated() / 2**30)
ts = []
for i in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eager = i % 10 == 5
    if eager:
        ops.KERNEL_TIMER.enable() if "--timer" in sys.argv else None
        tr.step(x, y)
        ops.KERNEL_TIMER.disable()
    else:
        gs.replay()
    torch.cuda.synchronize()
    ts.append(((time.perf_counter() - t0) * 1e3, "E" if eager else "g"))
print(" ".join(f"{t:.0f}{k}" for t, k in ts))
print("reserved GB", torch.cuda.memory_reserved() / 2**30)
