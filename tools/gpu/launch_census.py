"""GPU-box tool: every aten op of one eager real-view training step that reaches the device, attributed to the project line that
issued it -- forward ops by their Python stack, backward ops by the autograd node that runs them and (anomaly mode) the project
line that created that node in forward.  torch.profiler's with_stack leaves the stacks empty on this build; a TorchDispatchMode
sees every op on the calling thread and on the autograd thread alike.

    python tools/gpu/launch_census.py [--glue reference] [--virtual RES]
"""
import argparse, collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from morpheus_amd import harness
from bench_support import trainstep
from morpheus_amd.occgrid import OccupancyGrid
from morpheus_amd.optim import FlatAdam
from morpheus_amd.render import HotPathRenderer

ap = argparse.ArgumentParser()
ap.add_argument("--glue", default="fused")
ap.add_argument("--top", type=int, default=120)
ap.add_argument("--cfg3", action="store_true", help="the bench's cfg3 step (16 384 rays x 128 samples, fwd + bwd + Adam) instead of the real-view step")
args = ap.parse_args()
DEV = torch.device("cuda", 0)
model = harness.build_model("b", DEV).train()
if args.cfg3:
    from morpheus_amd import synth
    for k in ("normal_smoothness", "normal_smooth_3d", "code_reg", "ori_weight"):
        model.config["train"][k] = 0.0
    o, d, t, rid = [v.to(DEV) for v in synth.frame_rays(0, 128, 128)]
    rend3 = harness.make_renderer(model, 128, jitter=synth.ray_jitter(o.shape[1]).to(DEV))
    light = torch.nn.functional.normalize(o[0] + torch.tensor([0.3, -0.2, 0.5], device=DEV), dim=-1)
    timg, tdep = [v.to(DEV) for v in synth.targets(o.shape[1])]
grid = OccupancyGrid([-model.bound] * 3 + [model.bound] * 3, 128).to(DEV)
rend = HotPathRenderer(model, model.config, grid, 200)
frames = trainstep.make_frames([8 * k for k in range(8)], 256, 256, DEV)
ts = trainstep.RealViewTrainStep(rend, frames, ray_num=2048, glue=args.glue)
ts.epoch = 1000
opt = FlatAdam(model.get_params_all(model.config["train"]["lr"]), betas=(0.9, 0.99), eps=1e-15)
with torch.no_grad():
    trainstep.warm_up_occupancy(ts)
ts.global_step = 4096 + 3


def step():
    opt.bucket.zero()
    if args.cfg3:
        res = rend3.render_rays(o, d, t, rid, 128, 128, ambient_ratio=1.0, light_d=light, shading="albedo", cano=False)
        loss = harness.bench_loss(res, timg, tdep)
    else:
        loss = ts()
    loss.backward()
    opt.bucket.collect()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()

# views, metadata and allocation without a fill never launch
NO_LAUNCH = ("view", "reshape", "expand", "slice", "select", "unsqueeze", "squeeze", "permute", "transpose", "t.", "detach", "alias",
             "as_strided", "empty", "size", "stride", "is_", "_unsafe_view", "unbind", "split", "chunk", "narrow", "lift_fresh",
             "_local_scalar_dense", "sym_", "numel", "dim", "storage_offset", "unfold", "_reshape_alias", "set_", "resize_",
             "result_type", "can_cast", "_has_compatible", "new_empty", "contiguous")


def launches(func, args_):
    name = func.__name__ if hasattr(func, "__name__") else str(func)
    if any(name.startswith(p) for p in NO_LAUNCH):
        return False
    for a in args_:
        if isinstance(a, torch.Tensor):
            return a.is_cuda and (a.numel() > 0 or name.startswith(("zeros", "ones", "full")))
        if isinstance(a, (list, tuple)) and a and isinstance(a[0], torch.Tensor):
            return a[0].is_cuda
    return name.startswith(("zeros", "ones", "full", "arange", "rand", "tensor", "scalar_tensor"))


def project_frame(stack):
    for fr in reversed(stack):
        if "/morpheus_amd/" in fr.filename and "launch_census" not in fr.filename:
            return f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
    return None


class Census(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.rows = collections.Counter()

    def __torch_dispatch__(self, func, types, args_=(), kwargs=None):
        out = func(*args_, **(kwargs or {}))
        if launches(func, args_):
            site = project_frame(traceback.extract_stack())
            node = torch._C._current_autograd_node()
            if node is not None:
                made = None
                tb = node.metadata.get("traceback_") if hasattr(node, "metadata") else None
                if tb:
                    for line in reversed(tb):
                        if "/morpheus_amd/" in line:
                            made = line.strip().split("\n")[0].replace('File "', "").split("/morpheus_amd/")[-1]
                            break
                where = f"bwd {node.name()} <- {site or made or '?'}"
            else:
                where = f"fwd {site or '?'}"
            shape = next((tuple(a.shape) for a in args_ if isinstance(a, torch.Tensor)), ())
            self.rows[(func.__name__ if hasattr(func, "__name__") else str(func), where, str(shape))] += 1
        return out


with torch.autograd.detect_anomaly(check_nan=False):
    c = Census()
    with c:
        step()
torch.cuda.synchronize()
total = sum(c.rows.values())
print(f"aten ops that launch in one step (C-ABI kernels not counted): {total}")
by_site = collections.Counter()
for (name, where, shape), n in c.rows.items():
    by_site[where] += n
print("--- by site ---")
for where, n in by_site.most_common(args.top):
    print(f"{n:4d}  {where}")
print("--- by (op, site, first tensor shape) ---")
for (name, where, shape), n in c.rows.most_common(args.top):
    print(f"{n:4d}  {name:28s} {shape:22s} {where}")
