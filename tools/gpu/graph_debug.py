"""GPU-box debug: mean loss of the real-view step over 40 steps -- eager vs graphed (with / without the one-step look-ahead)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from morpheus_amd import harness, trainstep
from morpheus_amd.occgrid import OccupancyGrid
from morpheus_amd.optim import FlatAdam
from morpheus_amd.render import HotPathRenderer
DEV = torch.device("cuda", 0)


def build():
    torch.manual_seed(0)
    model = harness.build_model("b", DEV).train()
    grid = OccupancyGrid([-model.bound] * 3 + [model.bound] * 3, 128).to(DEV)
    rend = HotPathRenderer(model, model.config, grid, 200)
    frames = trainstep.make_frames([8 * k for k in range(8)], 256, 256, DEV)
    ts = trainstep.RealViewTrainStep(rend, frames, ray_num=2048)
    ts.epoch = 1000
    opt = FlatAdam(model.get_params_all(model.config["train"]["lr"]), betas=(0.9, 0.99), eps=1e-15)
    with torch.no_grad():
        trainstep.warm_up_occupancy(ts)
    ts.global_step = 4096
    return model, grid, ts, opt


def run(mode, n=40):
    model, grid, ts, opt = build()
    losses = []
    if mode == "eager":
        for _ in range(n):
            opt.bucket.zero()
            loss = ts()
            loss.backward()
            opt.bucket.allreduce_mean()
            opt.step()
            losses.append(float(loss))
    else:
        gs = trainstep.GraphedRealViewStep(ts, opt.bucket, lookahead=(mode == "graph"))
        gs.prepare()
        for _ in range(n):
            loss = gs()
            opt.step()
            losses.append(float(loss))
        print("   captures", gs.n_captures, "overflow", gs.check_overflow())
    print(mode, "mean(last 32) %.4f" % (sum(losses[-32:]) / 32), " ".join("%.3f" % l for l in losses[:6]), "...", " ".join("%.3f" % l for l in losses[-6:]))


for m in sys.argv[1:] or ["eager", "graph_sync", "graph"]:
    run(m)
