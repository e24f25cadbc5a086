"""Developer probe: which Python call sites of one DINOv2 step issue tensor copies (copy_ / clone / contiguous / to / float on
CUDA tensors), by caller line inside octic_vits_amd - the host-side view of the `__amd_rocclr_copyBuffer` entries of a profile.
usage: ssl_copy_sites.py [images_per_gpu=8]"""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octic_vits_amd import ssl as S
from octic_vits_amd.dinov2_models import hybrid_dinov2_vit_huge_patch16

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.manual_seed(0)
arch = S.SSLMetaArch(lambda: hybrid_dinov2_vit_huge_patch16(img_size=224, drop_path_rate=0.4), 1280).cuda()
tr = S.SSLTrainer(arch, lr=1e-4)
images = S.synthetic_multicrop_batch(batch, "cuda", seed=5)
for _ in range(2):
    tr.step(images, teacher_temp=0.04, momentum=0.992)
torch.cuda.synchronize()
sites = collections.Counter()


def wrap(name):
    orig = getattr(torch.Tensor, name)

    def f(self, *a, **k):
        out = orig(self, *a, **k)
        if self.is_cuda and (name in ("copy_", "clone") or (torch.is_tensor(out) and out.data_ptr() != self.data_ptr())):
            fr = [x for x in traceback.extract_stack(limit=12) if "octic_vits_amd" in x.filename]
            where = f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno}" if fr else "?"
            sites[(name, tuple(self.shape), str(self.dtype).replace("torch.", ""), where)] += 1
        return out
    setattr(torch.Tensor, name, f)


for n in ("copy_", "clone", "contiguous", "to", "float", "bfloat16"):
    wrap(n)


def where_():
    fr = [x for x in traceback.extract_stack(limit=14) if "octic_vits_amd" in x.filename]
    return f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno}" if fr else "?"


# host -> device uploads: Tensor.to / .cuda on CPU tensors, torch.tensor / as_tensor / scalar_tensor with a CUDA device
for n in ("to", "cuda"):
    orig = getattr(torch.Tensor, n)

    def f(self, *a, _o=orig, _n=n, **k):
        out = _o(self, *a, **k)
        if not self.is_cuda and torch.is_tensor(out) and out.is_cuda:
            sites[("H2D " + _n, tuple(self.shape), str(self.dtype).replace("torch.", ""), where_())] += 1
        return out
    setattr(torch.Tensor, n, f)
for n in ("tensor", "as_tensor", "scalar_tensor", "full", "from_numpy"):
    orig = getattr(torch, n)

    def f(*a, _o=orig, _n=n, **k):
        out = _o(*a, **k)
        if torch.is_tensor(out) and out.is_cuda and _n != "full":
            sites[("H2D torch." + _n, tuple(out.shape), str(out.dtype).replace("torch.", ""), where_())] += 1
        return out
    setattr(torch, n, f)
tr.step(images, teacher_temp=0.04, momentum=0.992)
torch.cuda.synchronize()
for (name, shape, dt, where), n in sorted(sites.items(), key=lambda kv: -kv[1])[:45]:
    print(f"{n:5d}  {name:11s} {str(shape):28s} {dt:9s} {where}")
