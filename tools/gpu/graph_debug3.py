"""GPU-box debug: which C-ABI op is not idempotent under HIP-graph replay?  Each op is captured alone and replayed 3 times."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from morpheus_amd import harness, ops, synth
DEV = torch.device("cuda", 0)
torch.manual_seed(0)
model = harness.build_model("b", DEV).train()
M, N = 40000, 500
x = (torch.rand(M, 3, device=DEV) * 1.6 - 0.8)
ri = torch.arange(M, device=DEV, dtype=torch.int32) // (M // N)
ts = torch.rand(M, device=DEV) + 0.5
te = ts + 0.01
dep = torch.rand(N, device=DEV) + 0.6
msk = (torch.rand(N, device=DEV) > 0.3).float()


def case_sdf_losses():
    p = x[:, 0].clone().requires_grad_(True)
    fs, sl = ops.sdf_losses(p, ts, te, ri, dep, msk, 0.1)
    (fs + sl).backward()
    return [fs.detach(), sl.detach(), p.grad]


def case_grid():
    xx = x.clone().requires_grad_(True)
    e = model.encoder.embeddings.detach().clone().requires_grad_(True)
    out = ops.grid_encode(xx, e, model.encoder._offsets_np, model.encoder._res_np, 1.01)
    (out ** 2).sum().backward()
    return [out.detach(), xx.grad, e.grad]


def case_field():
    xx = x.clone().requires_grad_(True)
    with model.operand_scope():
        sdf, sig, alb = model.get_sigma_albedo(xx, None)
        ((sdf ** 2).sum() + (alb ** 2).sum() + 1e-3 * (sig ** 2).mean()).backward()
    return [sdf.detach(), alb.detach(), xx.grad, model.encoder.embeddings.grad.clone(), model.sdf_net.net[0].weight.grad.clone()]


def case_warp():
    xx = x.clone().requires_grad_(True)
    t = torch.full((1, 1), 0.3, device=DEV).expand(M, 1)
    with model.operand_scope():
        d, tp, _ = model.warp(xx, t)
        ((d ** 2).sum() + (tp ** 2).sum()).backward()
    return [d.detach(), tp.detach(), xx.grad, model.deform_net.net[1].weight_v.grad.clone(), model.deform_code.volumes[2].grad.clone()]


def case_normal():
    xx = x.clone().requires_grad_(True)
    with model.operand_scope():
        n, raw = model.normal(xx, topo=None)
        (n ** 2 + raw).sum().backward()
    return [n.detach(), raw.detach(), xx.grad]


def case_composite():
    sig = (torch.rand(M, device=DEV) * 20).requires_grad_(True)
    rgb = torch.rand(M, 3, device=DEV).requires_grad_(True)
    rs = (torch.arange(N, device=DEV, dtype=torch.int32) * (M // N))
    rc = torch.full((N,), M // N, device=DEV, dtype=torch.int32)
    w, o, dd, c = ops.composite(sig, ts, te, rgb, rs, rc)
    ((c ** 2).sum() + (dd ** 2).sum() + o.sum()).backward()
    return [w.detach(), c.detach(), sig.grad, rgb.grad]


for name, fn in list(globals().items()):
    if not name.startswith("case_"):
        continue
    model.zero_grad(set_to_none=True)
    ref = [t.clone() for t in fn()]
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            model.zero_grad(set_to_none=True)
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    model.zero_grad(set_to_none=True)
    with torch.cuda.graph(g):
        outs = fn()
    res = []
    for rep in range(3):
        g.replay()
        torch.cuda.synchronize()
        res.append([float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)) for a, b in zip(outs, ref)])
    print(name, " | ".join(" ".join("%.1e" % v for v in r) for r in res))
