"""Developer probe: which ops of a train step issue device-to-device memcpys (hipMemcpyAsync -> __amd_rocclr_copyBuffer)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from octic_vits_amd.deit_models import create_model
from octic_vits_amd.train import Trainer, synthetic_batch
model = create_model("hybrid_deit_huge_patch14", num_classes=1000, drop_path_rate=0.5, img_size=224).cuda()
tr = Trainer(model)
x, y = synthetic_batch(8, 1000, "cuda", 1)
for _ in range(2):
    tr.step(x, y)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False, record_shapes=True) as prof:
    tr.step(x, y)
    torch.cuda.synchronize()
evs = prof.events()
cnt = collections.Counter()
for e in evs:
    n = e.name
    if "Memcpy" in n or "memcpy" in n or "copyBuffer" in n:
        p = e.cpu_parent
        chain = []
        while p is not None and len(chain) < 4:
            chain.append(p.name)
            p = p.cpu_parent
        cnt[(n, tuple(chain), str(getattr(e, "input_shapes", "")))] += 1
for k, v in cnt.most_common(25):
    print(v, k)
cc = collections.Counter(e.name for e in evs if e.device_type == torch.autograd.DeviceType.CUDA)
print([ (k, v) for k, v in cc.most_common(60) if "copy" in k.lower() or "Memcpy" in k])
# CPU-side ops that call copy_
c2 = collections.Counter()
for e in evs:
    if e.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::_to_copy"):
        p = e.cpu_parent
        c2[(e.name, p.name if p is not None else None, str(e.input_shapes)[:80])] += 1
for k, v in c2.most_common(30):
    print(v, k)
