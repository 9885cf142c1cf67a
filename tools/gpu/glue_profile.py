"""GPU-box tool: which Python lines issue the small torch launches of one eager real-view training step?  (torch.profiler, CPU-side
op records with stacks; the eager step and the replayed graph issue the same kernels)"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from morpheus_amd import harness, trainstep
from morpheus_amd.occgrid import OccupancyGrid
from morpheus_amd.optim import FlatAdam
from morpheus_amd.render import HotPathRenderer
DEV = torch.device("cuda", 0)
model = harness.build_model("b", DEV).train()
grid = OccupancyGrid([-model.bound] * 3 + [model.bound] * 3, 128).to(DEV)
rend = HotPathRenderer(model, model.config, grid, 200)
frames = trainstep.make_frames([8 * k for k in range(8)], 256, 256, DEV)
ts = trainstep.RealViewTrainStep(rend, frames, ray_num=2048)
ts.epoch = 1000
opt = FlatAdam(model.get_params_all(model.config["train"]["lr"]), betas=(0.9, 0.99), eps=1e-15)
with torch.no_grad():
    trainstep.warm_up_occupancy(ts)
ts.global_step = 4096 + 3


def step():
    opt.bucket.zero()
    loss = ts()
    loss.backward()
    opt.bucket.collect()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
LAUNCHERS = ("aten::add", "aten::add_", "aten::mul", "aten::mul_", "aten::fill_", "aten::zero_", "aten::copy_", "aten::sub", "aten::div", "aten::neg",
             "aten::sum", "aten::mean", "aten::index", "aten::cat", "aten::stack", "aten::mm", "aten::addmm", "aten::clamp", "aten::abs", "aten::sign",
             "aten::pow", "aten::sqrt", "aten::linalg_vector_norm", "aten::where", "aten::lt", "aten::gt", "aten::le", "aten::bitwise_and", "aten::cos", "aten::sin",
             "aten::rand", "aten::uniform_", "aten::index_select", "aten::_foreach_copy_", "aten::square", "aten::mse_loss", "aten::mse_loss_backward",
             "aten::index_put_", "aten::_index_put_impl_", "aten::sigmoid", "aten::expand", "aten::unbind")
by_site = collections.Counter()
by_op = collections.Counter()
for e in prof.events():
    if e.name not in LAUNCHERS or e.name == "aten::expand":
        continue
    stack = [s for s in (e.stack or []) if "morpheus_amd" in s or "autograd" in s.lower()]
    site = stack[0] if stack else ("<backward/autograd engine>" if not e.stack else e.stack[0])
    by_site[(e.name, site[-110:], str(e.input_shapes)[:60])] += 1
    by_op[e.name] += 1
print("ops that launch, by name:", by_op.most_common(30))
print("--- by (op, innermost project frame, shapes) ---")
for (name, site, shp), n in by_site.most_common(70):
    print(f"{n:4d}  {name:28s} {shp:60s} {site}")
