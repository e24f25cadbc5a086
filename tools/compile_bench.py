"""Developer measurement (round-4 review item 4): the ViT-H bf16 train step issued (a) eagerly through the autograd.Function
path, (b) by torch.compile(model) through the dispatcher ops of dispatch.py (backend aot_eager: Dynamo + AOTAutograd graphs
calling the HIP ops; no code generation), at batch 64 (GPU-bound) and batch 2 (host-bound: the step time is the host's cost
of issuing it).  usage: compile_bench.py [backend=aot_eager]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octic_vits_amd.deit_models import create_model
from octic_vits_amd.train import Trainer, synthetic_batch

backend = sys.argv[1] if len(sys.argv) > 1 else "aot_eager"
torch.manual_seed(0)
model = create_model("hybrid_deit_huge_patch14", num_classes=1000, drop_path_rate=0.5, img_size=224).cuda()
tr = Trainer(model)
cm = torch.compile(model, backend=backend)


def timed(step, n):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, host / n * 1e3


for B in (64, 2):
    x, y = synthetic_batch(B, 1000, "cuda", 1)
    e_ms, e_host = timed(lambda: tr.step(x, y), 8)
    tr.model = cm
    t0 = time.time()
    tr.step(x, y)
    torch.cuda.synchronize()
    first = time.time() - t0
    c_ms, c_host = timed(lambda: tr.step(x, y), 8)
    tr.model = model
    print(f"batch {B}: eager {e_ms:.1f} ms/step (host issue {e_host:.1f}) | compiled[{backend}] {c_ms:.1f} ms/step (host issue {c_host:.1f}; "
          f"first call {first:.1f} s)")
