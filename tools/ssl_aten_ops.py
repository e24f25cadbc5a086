"""Developer probe: which ATen operators of one DINOv2 student/teacher step (tools/bench_ssl.py's workload) still cost GPU time,
by operator + input shapes + the innermost octic_vits_amd frame that issued them (torch.profiler, one step).
usage: ssl_aten_ops.py [images_per_gpu=32] [rows=40] [--by-count]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from octic_vits_amd import ssl as S
from octic_vits_amd.dinov2_models import hybrid_dinov2_vit_huge_patch16

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 40
torch.manual_seed(0)
arch = S.SSLMetaArch(lambda: hybrid_dinov2_vit_huge_patch16(img_size=224, drop_path_rate=0.4), 1280).cuda()
tr = S.SSLTrainer(arch, lr=1e-4)
images = S.synthetic_multicrop_batch(batch, "cuda", seed=5)
for _ in range(2):
    tr.step(images, teacher_temp=0.04, momentum=0.992)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    tr.step(images, teacher_temp=0.04, momentum=0.992)
    torch.cuda.synchronize()
agg = {}
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.self_device_time_total <= 0:
        continue
    frame = next((f for f in ev.stack if "octic_vits_amd" in f), ev.stack[0] if ev.stack else "?")
    frame = frame.split("octic_vits_amd/")[-1]
    key = (ev.name, str(ev.input_shapes)[:70], frame[:60])
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1
    a[1] += ev.self_device_time_total
tot = sum(a[1] for a in agg.values())
print(f"ATen self device time of one step: {tot / 1e3:.2f} ms")
by_count = "--by-count" in sys.argv
for (name, shapes, frame), (n, us) in sorted(agg.items(), key=lambda kv: -(kv[1][0] if by_count else kv[1][1]))[:rows]:
    print(f"{us / 1e3:7.3f} ms {n:5d}  {name:28s} {shapes:70s} {frame}")
