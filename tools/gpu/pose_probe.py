"""GPU-box probe: in which order does torch add the three products of `(d[..., None, :] * R).sum(-1)`?  (csrc/losses.hip pose_apply_kernel)"""
import torch
DEV = torch.device("cuda", 0)
torch.manual_seed(0)
d = torch.nn.functional.normalize(torch.randn(4096, 3, device=DEV), dim=-1)
R = torch.randn(1, 3, 3, device=DEV)
ref = (d.view(1, 4096, 1, 3) * R[:, None]).sum(-1).view(-1, 3)
p = d[:, None, :] * R[0][None]            # [N,3,3] products
a = (p[..., 0] + p[..., 1]) + p[..., 2]
b = p[..., 0] + (p[..., 1] + p[..., 2])
c = (p[..., 0] + p[..., 2]) + p[..., 1]
f = torch.addcmul(torch.addcmul(p[..., 0], d[:, None, 1].expand(-1, 3), R[0][None, :, 1].expand(4096, -1)), d[:, None, 2].expand(-1, 3), R[0][None, :, 2].expand(4096, -1))
for name, v in (("(0+1)+2", a), ("0+(1+2)", b), ("(0+2)+1", c), ("fma chain", f)):
    print(name, int((v != ref).sum()), "of", ref.numel())
