"""GPU, world_size 2 on ONE device (gloo for the exchange): the real render step under the data-parallel path of
bench.py -- each rank renders its own frame (cfg5's partitioning), the hash-table gradients leave early on the side
stream, the remainder follows, every rank ends with the MEAN gradient -- against the same two frames rendered one after
the other in a single process."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
HW, S = 16, 32


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _frame_grads(frame, bucket_factory=None, optimize_pose=False):
    from morpheus_amd import harness, synth
    dev = torch.device("cuda", 0)
    model = harness.build_model("b", dev).train()
    for k in ("normal_smoothness", "normal_smooth_3d", "code_reg", "ori_weight"):
        model.config["train"][k] = 0.0
    o, d, t, rid = [v.to(dev) for v in synth.frame_rays(frame, HW, HW)]
    N = o.shape[1]
    rend = harness.make_renderer(model, S, jitter=synth.ray_jitter(N).to(dev))
    timg, tdep = [v.to(dev) for v in synth.targets(N)]
    bucket = bucket_factory(model) if bucket_factory else None
    if bucket is not None:
        bucket.zero()
    res = rend.render_rays(o, d, t, rid, HW, HW, ambient_ratio=1.0, shading="albedo",
                           light_d=torch.nn.functional.normalize(o[0] + 0.3, dim=-1), optimize_pose=optimize_pose)
    harness.bench_loss(res, timg, tdep).backward()
    return model, bucket


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    from morpheus_amd import dist as mdist
    from morpheus_amd.optim import FlatAdam
    mdist.init_from_env(backend="gloo")

    opts = []

    def factory(model):
        opt = FlatAdam(model.get_params_all(5e-4), betas=(0.9, 0.99), eps=1e-15)
        opt.bucket.overlap_early([model.encoder.embeddings, model.encoder_c.embeddings])
        opts.append(opt)
        return opt.bucket

    # rank 0 optimises the pose (its pose_array gets a gradient), rank 1 does not: "has a gradient" must be decided across ranks
    model, bucket = _frame_grads(25 * rank, factory, optimize_pose=(rank == 0))
    assert len(bucket._early_work) == 1, "the hash-table range should have left from the autograd hook"
    bucket.allreduce_mean()
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters() if p.grad is not None}
    before = {k: p.detach().cpu().clone() for k, p in model.named_parameters()}
    opts[0].step()                                           # the device-side path: no host read of the flags
    torch.cuda.synchronize()
    sd = opts[0].state_dict()
    names = [k for k, _ in model.named_parameters()]
    order = {id(p): k for k, p in model.named_parameters()}
    steps = {}
    for gi, g in enumerate(opts[0].param_groups):
        for p in g["params"]:
            steps[order[id(p)]] = int(float(opts[0].state[p]["step"]))
    out[rank] = dict(grads=grads, before=before, after={k: p.detach().cpu().clone() for k, p in model.named_parameters()},
                     steps=steps, names=names)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_render_and_average_gradients():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    ref = []
    for frame, pose in ((0, True), (25, False)):
        model, _ = _frame_grads(frame, optimize_pose=pose)
        ref.append({k: (torch.zeros_like(p) if p.grad is None else p.grad.detach()).cpu() for k, p in model.named_parameters()})
    checked = 0
    for k in ref[0]:
        want = 0.5 * (ref[0][k].double() + ref[1][k].double())
        if float(want.norm()) == 0:
            continue
        for r in (0, 1):
            err = float((out[r]["grads"][k].double() - want).norm() / want.norm())
            assert err < 1e-5, (k, r, err)
        assert torch.equal(out[0]["grads"][k], out[1]["grads"][k]), k       # both ranks hold the same bucket after the exchange
        checked += 1
    assert checked >= 40, checked
    # the optimiser step: identical replicas; a parameter with a gradient on ONE rank (the pose, rank 0 only) is stepped on BOTH
    # with the mean gradient and counts one step; parameters without gradient anywhere (the background net) are untouched, count 0
    for k in out[0]["names"]:
        assert torch.equal(out[0]["after"][k], out[1]["after"][k]), k
        assert out[0]["steps"][k] == out[1]["steps"][k], k
    assert out[0]["steps"]["pose_array.data"] == 1 and not torch.equal(out[0]["after"]["pose_array.data"], out[0]["before"]["pose_array.data"])
    for k in out[0]["names"]:
        if k.startswith("bg_net"):
            assert out[0]["steps"][k] == 0 and torch.equal(out[0]["after"][k], out[0]["before"][k]), k
        elif float(ref[0][k].norm()) + float(ref[1][k].norm()) > 0:
            assert out[0]["steps"][k] == 1, k


def _two_backward_grads(frames, bucket_factory=None):
    """cfg4's accumulation (morpheus.py:1396-1424 with freeze_lr off): TWO backward passes -- here two frames rendered one after
    the other, standing for the virtual-view and the real-view pass -- before one exchange / optimiser step."""
    from morpheus_amd import harness, synth
    dev = torch.device("cuda", 0)
    model = harness.build_model("b", dev).train()
    for k in ("normal_smoothness", "normal_smooth_3d", "code_reg", "ori_weight"):
        model.config["train"][k] = 0.0
    bucket = bucket_factory(model) if bucket_factory else None
    if bucket is not None:
        bucket.zero()
    for frame in frames:
        o, d, t, rid = [v.to(dev) for v in synth.frame_rays(frame, HW, HW)]
        N = o.shape[1]
        rend = harness.make_renderer(model, S, jitter=synth.ray_jitter(N).to(dev))
        timg, tdep = [v.to(dev) for v in synth.targets(N)]
        res = rend.render_rays(o, d, t, rid, HW, HW, ambient_ratio=1.0, shading="albedo",
                               light_d=torch.nn.functional.normalize(o[0] + 0.3, dim=-1))
        harness.bench_loss(res, timg, tdep).backward()
    return model, bucket


def _worker2(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import warnings
    from morpheus_amd import dist as mdist
    from morpheus_amd.optim import FlatAdam
    mdist.init_from_env(backend="gloo")
    opts = []

    def factory(model):
        opt = FlatAdam(model.get_params_all(5e-4), betas=(0.9, 0.99), eps=1e-15)
        opt.bucket.overlap_early([model.encoder.embeddings, model.encoder_c.embeddings], backwards_per_step=2)
        opts.append(opt)
        return opt.bucket

    with warnings.catch_warnings():
        warnings.simplefilter("error")                       # a declared two-backward step must not fall back to the late path
        model, bucket = _two_backward_grads([25 * rank, 25 * rank + 8], factory)
        fired_after = bucket._early_hits
        assert len(bucket._early_work) == 1 and fired_after == 2 * len(bucket._early), "one early exchange, after the SECOND backward"
        bucket.allreduce_mean()
    torch.cuda.synchronize()
    out[rank] = {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters() if p.grad is not None}
    opts[0].step()
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_two_backwards_per_step():
    """overlap_early(backwards_per_step=2): the hash-table range leaves once, after the second backward pass, and every rank ends
    with the mean over ranks of its accumulated (two-pass) gradient."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker2, args=(2, _free_port(), out), nprocs=2, join=True)
    ref = []
    for rank in (0, 1):
        model, _ = _two_backward_grads([25 * rank, 25 * rank + 8])
        ref.append({k: (torch.zeros_like(p) if p.grad is None else p.grad.detach()).cpu() for k, p in model.named_parameters()})
    checked = 0
    for k in ref[0]:
        want = 0.5 * (ref[0][k].double() + ref[1][k].double())
        if float(want.norm()) == 0:
            continue
        for r in (0, 1):
            err = float((out[r][k].double() - want).norm() / want.norm())
            assert err < 1e-5, (k, r, err)
        assert torch.equal(out[0][k], out[1][k]), k
        checked += 1
    assert checked >= 40, checked
