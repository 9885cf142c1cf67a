"""Is the b3 mode's error on d(loss)/d(encoder.embeddings) of the 72 x 72 virtual-view step (14 x the reference's own fp32 error on the
fixture's 64 strided samples, tools/gpu/virt_double_diag.py) ARITHMETIC (broad, every row a little off) or a DISCRETE event (a tap point
whose canonical position moved by an ulp sits in another grid cell at some level: a few rows far off, the rest at round-off)?
The same step in the b3 and in the f32 mode, one process; the full table gradients compared row by row."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from morpheus_amd import harness, synth
from bench_support import trainstep
from oracle import field as of
from tests.util import DrawInjector
DEV = torch.device("cuda", 0)
hw, S, frame = 72, 32, 140
o, d = synth.camera_rays(hw, hw, synth.look_at_pose(70.0, 35.0, 1.5))
N = o.shape[0]
smp = of.uniform_samples(o, d, synth.ray_jitter(N), S, 1.01)
light = of.safe_normalize(o + torch.tensor([0.3, -0.2, 0.5])).to(DEV)


def run(mode):
    model = harness.build_model("b", DEV, 0.75).train()
    model.mlp_mode = mode
    model.config["train"]["normal_smoothness"] = 0.0
    rend = harness.make_renderer(model, S, samples=tuple(v.to(DEV) for v in smp))
    ts = trainstep.VirtualViewTrainStep(rend, res=hw, guidance=trainstep.InjectedGuidance(hw, hw, DEV, scale=5e-3))
    ts.epoch, ts.global_step = 1000, 999
    data = dict(H=hw, W=hw, rays_o=o[None].to(DEV), rays_d=d[None].to(DEV), rays_t=torch.full((1, N, 1), frame / 200, device=DEV),
                rays_id=torch.full((1, N, 1), frame, device=DEV, dtype=torch.int64))
    model.zero_grad()
    with DrawInjector():
        loss = ts(data=data, shading="lambertian", ambient_ratio=0.55, bg_color=torch.tensor([0.2, 0.5, 0.7], device=DEV), light_d=light)
    loss.backward()
    return {k: p.grad.detach().double().cpu() for k, p in model.named_parameters() if p.grad is not None}, float(loss)


ga, la = run("b3")
gb, lb = run("f32")
gc, lc = run("b3")
print("loss b3 %.9f  f32 %.9f  b3 again %.9f" % (la, lb, lc))
for k in ("encoder.embeddings", "encoder_c.embeddings"):
    a, b, c = ga[k], gb[k], gc[k]
    scale = float(b.abs().max())
    dab = (a - b).abs().max(dim=1).values / scale           # per table row
    dac = (a - c).abs().max(dim=1).values / scale
    touched = (b.abs().max(dim=1).values > 0)
    srt = torch.sort(dab[touched], descending=True).values
    print("%s: rows touched %d, max|grad| %.3e" % (k, int(touched.sum()), scale))
    print("   b3 vs f32, per-row max |diff| / max|grad|: max %.2e, 10th largest %.2e, 100th %.2e, 1000th %.2e, median %.2e" %
          (float(srt[0]), float(srt[9]), float(srt[99]), float(srt[999]), float(srt[len(srt) // 2])))
    for th in (1e-4, 3e-5, 1e-5, 3e-6):
        print("   rows off by more than %.0e of max|grad|: %d" % (th, int((dab > th).sum())))
    print("   b3 vs b3 again (atomics order only): max %.2e" % float(dac.max()))
    idx = torch.linspace(0, a.numel() - 1, 64).long()
    fa, fb = a.reshape(-1)[idx], b.reshape(-1)[idx]
    j = int((fa - fb).abs().argmax())
    print("   the fixture's 64 strided samples: largest b3 - f32 difference %.2e of max sample, at flat index %d (row %d)" %
          (float((fa - fb).abs().max() / fb.abs().max()), int(idx[j]), int(idx[j]) // 2))
