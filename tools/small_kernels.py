"""Developer tool: which host-side ops launch the tiny (< 10 us) kernels of a train step?  One eager step under
torch.profiler; prints (aten op / autograd node, input shapes) -> kernel count and device time."""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from octic_vits_amd.deit_models import create_model
from octic_vits_amd.train import Trainer, synthetic_batch

model = create_model("hybrid_deit_huge_patch14", num_classes=1000, drop_path_rate=0.5).cuda()
tr = Trainer(model)
x, y = synthetic_batch(64, 1000, "cuda", seed=0)
for _ in range(3): tr.step(x, y)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.step(x, y)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    ks = getattr(ev, "kernels", None)
    if not ks: continue
    for k in ks:
        dur = k.duration if hasattr(k, "duration") else 0.0
        if dur < 10.0:
            key = (ev.name, str(getattr(ev, "input_shapes", ""))[:70], k.name[:50])
            agg[key][0] += 1; agg[key][1] += dur
tot = sum(v[0] for v in agg.values())
print(f"{tot} kernels under 10 us in one step")
for key, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:45]:
    print(f"{n:5d} {us:8.1f} us  {key[0][:40]:40s} {key[1]:70s} {key[2]}")
