"""How sparse are the parked tiles of the benchmark step?  Fraction of exact zeros in the warp nets' parked activations
(post-ReLU) and in the dPre tiles, per layer, for the bench model (closed-form state b, frame 0, 16384 x 128)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from morpheus_amd import harness, synth, ops, _lib
dev = torch.device("cuda", 0)
model = harness.build_model("b", dev).train()
for k in ("normal_smoothness", "normal_smooth_3d", "code_reg", "ori_weight"):
    model.config["train"][k] = 0.0
HW, S = 128, 128
o, d, t, rid = [v.to(dev) for v in synth.frame_rays(0, HW, HW)]
N = o.shape[1]
rend = harness.make_renderer(model, S, jitter=synth.ray_jitter(N).to(dev))
timg, tdep = [v.to(dev) for v in synth.targets(N)]
cap = {}
orig = ops._wgrad
def spy(lib, acts, dpre, *a, **kw):
    if kw.get("b3") is not None and a[-1] == "warp" or (len(a) >= 10 and a[9] == "warp"):
        cap["acts"], cap["dpre"] = acts, dpre
    return orig(lib, acts, dpre, *a, **kw)
ops._wgrad = spy
res = rend.render_rays(o, d, t, rid, HW, HW, ambient_ratio=1.0, shading="albedo", light_d=torch.nn.functional.normalize(o[0] + 0.3, dim=-1))
harness.bench_loss(res, timg, tdep).backward()
lib = _lib.load()
M = N * S
nt = lib.mh_mlp_tiles(M)
a = cap["acts"].view(nt, -1, 32); dp = cap["dpre"].view(nt, -1, 32)
print("tiles", nt)
for net in range(2):
    for l in range(5):
        r0 = 64 + net * 640 + l * 128
        z = float((a[:, r0:r0 + 128] == 0).float().mean())
        zd = float((dp[:, net * 672 + l * 128: net * 672 + (l + 1) * 128] == 0).float().mean())
        print(f"net {net} layer {l + 1}: zeros in H {z:.3f}   zeros in dPre_{l} {zd:.3f}")
