import torch
a = torch.randn(1000, 256, device="cuda").bfloat16(); b = torch.randn(1000, 512, device="cuda").bfloat16()
try:
    c = torch.mm(a.t(), b, out_dtype=torch.float32)
    print("mm out_dtype ok", c.dtype, float((c - a.float().t() @ b.float()).abs().max()))
except Exception as e:
    print("mm out_dtype failed:", type(e).__name__, str(e)[:200])
try:
    out = torch.empty(256, 512, device="cuda")
    torch.mm(a.t(), b, out=out)
    print("mm out= f32 ok")
except Exception as e:
    print("mm out=f32 failed:", type(e).__name__, str(e)[:200])
