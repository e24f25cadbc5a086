"""Developer benchmark of BASELINE configs[4]: one DINOv2 student/teacher iteration (ssl.SSLTrainer) of the hybrid octic
ViT-H/16 on 2 x 224^2 + 8 x 96^2 crops per image, bf16 autocast, synthetic crops resident in HBM.
usage: bench_ssl.py [images_per_gpu=32] [steps=5]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octic_vits_amd import ssl as S
from octic_vits_amd.dinov2_models import hybrid_dinov2_vit_huge_patch16

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
torch.manual_seed(0)
arch = S.SSLMetaArch(lambda: hybrid_dinov2_vit_huge_patch16(img_size=224, drop_path_rate=0.4), 1280).cuda()
tr = S.SSLTrainer(arch, lr=1e-4)
images = S.synthetic_multicrop_batch(batch, "cuda", seed=5)
for _ in range(2):
    out = tr.step(images, teacher_temp=0.04, momentum=0.992)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    out = tr.step(images, teacher_temp=0.04, momentum=0.992)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"ssl step: {batch} images/GPU ({2 * batch} global + {8 * batch} local crops): {dt * 1e3:.1f} ms/step, {batch / dt:.1f} images/s, "
      f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB, losses " + ", ".join(f"{k}={float(v):.3f}" for k, v in out.items()))
