"""CPU, world_size 2, gloo: the N>1 path of bench.py -- rays sharded per rank, replicated parameters,
ONE all-reduce of the flat gradient bucket -- gives every rank the mean gradient and identical
parameters after the optimiser step."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out, overlap=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from morpheus_amd import dist as mdist
    r, l, w = mdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(0)                                   # replicated parameters
    net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 3))
    table = torch.nn.Parameter(torch.randn(64, 2))         # stands in for a hash table (sparse-ish gradient)
    params = list(net.parameters()) + [table]
    bucket = mdist.GradBucket(params)
    if overlap:
        bucket.overlap_early([table])                      # its gradient is exchanged from the autograd hook
    opt = torch.optim.Adam(params, lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    g = torch.Generator().manual_seed(123)
    rays = torch.randn(64, 6, generator=g)                 # the full batch; each rank renders its shard
    idx = torch.randint(0, 64, (64,), generator=g)
    lo, hi = mdist.shard_rays(64, rank, world)
    for _ in range(3):
        bucket.zero()
        y = net(rays[lo:hi]) + table[idx[lo:hi]].sum(-1, keepdim=True)
        (y ** 2).mean().backward()
        bucket.allreduce_mean()
        opt.step()
    out[rank] = torch.cat([p.detach().reshape(-1) for p in params]).clone()
    if rank == 0:
        out["nbytes"] = bucket.nbytes
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_bucket_allreduce_two_ranks():
    mgr = mp.Manager()
    out = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert torch.allclose(out[0], out[1], atol=0, rtol=0), "ranks diverged after all-reduce + Adam"
    assert out["nbytes"] == (6 * 16 + 16 + 16 * 3 + 3 + 128) * 4


def test_early_overlapped_exchange_matches_the_single_allreduce():
    """overlap_early(): the early range leaves from the post-accumulate hook while backward is still running and the
    remainder follows in allreduce_mean(); parameters after 3 Adam steps are bit-identical to the one-call exchange."""
    mgr = mp.Manager()
    ref, ovl = mgr.dict(), mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), ref, False), nprocs=2, join=True)
    mp.spawn(_worker, args=(2, _free_port(), ovl, True), nprocs=2, join=True)
    assert torch.equal(ovl[0], ovl[1]) and torch.equal(ovl[0], ref[0])


def _real_layout_worker(rank, world, port, out):
    """The REAL model's parameter set in optim.FlatAdam's layout (groups of get_params_all, hash tables first) with
    rank-dependent closed-form gradients: the flat bucket after the (early + remainder) exchange must be the mean of the
    two ranks' gradients, element for element, and every p.grad must be a view of the bucket."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from morpheus_amd import dist as mdist
    from morpheus_amd import harness, synth
    from morpheus_amd.optim import FlatAdam
    mdist.init_from_env(backend="gloo")
    model = harness.build_model("b", "cpu")
    opt = FlatAdam(model.get_params_all(5e-4), betas=(0.9, 0.99), eps=1e-15)      # construction works on the CPU; step() does not
    bucket = opt.bucket
    bucket.overlap_early([model.encoder.embeddings, model.encoder_c.embeddings])
    named = dict(model.named_parameters())
    grads = lambda r: {k: synth.hash_tensor(tuple(p.shape), 9000 + 17 * i + 1000 * r, 1.0) for i, (k, p) in enumerate(named.items())}
    mine = grads(rank)
    mean = {k: sum(grads(r)[k] for r in range(world)) / world for k in named}
    bucket.zero()
    # hand the gradients over the way autograd does: through backward, so that the post-accumulate hooks fire
    loss = sum((p * mine[k]).sum() for k, p in named.items())
    loss.backward()
    bucket.allreduce_mean()
    worst = 0.0
    for k, p in named.items():
        worst = max(worst, float((p.grad - mean[k]).abs().max()))
        assert p.grad.data_ptr() >= bucket.flat.data_ptr() and p.grad.data_ptr() < bucket.flat.data_ptr() + bucket.nbytes, k
    out[rank] = (worst, bucket.nbytes, bucket._early_span)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_real_model_bucket_all_ranks_get_the_mean(world):
    """world 8 = the node the path is meant for (cfg5: one frame per GPU); gloo on the CPU exercises the same bucket,
    early range and remainder exchange."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_real_layout_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    for r in range(world):
        worst, nbytes, span = out[r]
        assert worst <= 5e-7, worst
        assert 7.3e6 < nbytes < 7.6e6                      # the 7.45 MB bucket of SURVEY 8e
        assert span == (0, 2 * 839280)                     # both hash tables: the first 6.4 MB, exchanged early


def test_single_process_equivalence():
    """bucket: bound views accumulate in place; after zero() backward hands over fresh tensors that
    allreduce_mean() gathers; the all-reduce itself is a no-op at world size 1."""
    from morpheus_amd import dist as mdist
    torch.manual_seed(0)
    lin = torch.nn.Linear(4, 2)
    b = mdist.GradBucket(lin.parameters())
    x = torch.randn(5, 4)
    lin(x).sum().backward()
    g1 = b.flat.clone()
    b.allreduce_mean()
    assert torch.equal(b.flat, g1)
    assert lin.weight.grad.data_ptr() == b.flat.data_ptr()
    b.zero()
    assert lin.weight.grad is None and float(b.flat.abs().sum()) == 0.0
    lin(x).sum().backward()
    b.allreduce_mean()
    assert torch.equal(b.flat, g1) and lin.weight.grad.data_ptr() == b.flat.data_ptr()
    lo, hi = mdist.shard_rays(10, 3, 4)
    assert (lo, hi) == (9, 10) and mdist.shard_rays(10, 0, 4) == (0, 3)


def _pattern_worker(rank, world, port, out, case):
    """Data- and call-pattern cases the exchange must survive with identical replicas."""
    import warnings
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from morpheus_amd import dist as mdist
    mdist.init_from_env(backend="gloo")
    torch.manual_seed(0)
    table_a = torch.nn.Parameter(torch.randn(32, 2))       # two "hash tables" (the early range) ...
    table_b = torch.nn.Parameter(torch.randn(32, 2))
    w = torch.nn.Parameter(torch.randn(6, 3))              # ... an MLP weight and a pose-like parameter behind them
    pose = torch.nn.Parameter(torch.randn(4))
    params = [table_a, table_b, w, pose]
    bucket = mdist.GradBucket(params)
    g = torch.Generator().manual_seed(7 + rank)
    x = torch.randn(8, 6, generator=g)

    def loss_fn(use_a=True, use_b=True, use_pose=True, scale=1.0):
        y = (x @ w).sum()
        if use_a:
            y = y + (table_a ** 2).sum() * (rank + 1)
        if use_b:
            y = y + (table_b * 3.0).sum() * (rank + 2)
        if use_pose:
            y = y + (pose * (rank + 1.0)).sum()
        return y * scale

    def local_grads(**kw):
        """What this rank's backward produces, computed on clones (no hooks, no bucket)."""
        ps = [p.detach().clone().requires_grad_(True) for p in params]
        saved = [p.data for p in params]
        y = (x @ ps[2]).sum()
        if kw.get("use_a", True):
            y = y + (ps[0] ** 2).sum() * (rank + 1)
        if kw.get("use_b", True):
            y = y + (ps[1] * 3.0).sum() * (rank + 2)
        if kw.get("use_pose", True):
            y = y + (ps[3] * (rank + 1.0)).sum()
        (y * kw.get("scale", 1.0)).backward()
        del saved
        return [torch.zeros_like(p) if p.grad is None else p.grad for p in ps]

    def mean_over_ranks(mine):
        """Reference exchange: a plain all-reduce per tensor."""
        outg = []
        for t in mine:
            t = t.clone()
            dist.all_reduce(t)
            outg.append(t / world)
        return outg

    res = {}
    if case == "asymmetric_none":
        # rank 1's batch gives the pose parameter no gradient at all; rank 0's does -> both ranks must step it
        bucket.zero()
        loss_fn(use_pose=(rank == 0)).backward()
        bucket.allreduce_mean()
        want = mean_over_ranks(local_grads(use_pose=(rank == 0)))
        res["counts"] = bucket.grad_counts.tolist()              # what mh_adam_step_dev reads: ranks with a gradient, per parameter
        res["missing"] = sorted(bucket.resolve_missing())
        res["err"] = max(float((p.grad - t).abs().max()) for p, t in zip(params, want))
        # no rank has a gradient for it -> missing everywhere (torch.optim.Adam's skip)
        bucket.zero()
        loss_fn(use_pose=False).backward()
        bucket.allreduce_mean()
        res["missing_all"] = sorted(bucket.resolve_missing())
        res["counts_all"] = bucket.grad_counts.tolist()
    elif case == "early_param_missing_on_one_rank":
        bucket.overlap_early([table_a, table_b])
        for it in range(2):
            bucket.zero()
            loss_fn(use_b=(rank == 0)).backward()             # rank 1 never touches table_b: its hook does not fire there
            bucket.allreduce_mean()
            want = mean_over_ranks(local_grads(use_b=(rank == 0)))
            res["err%d" % it] = max(float((p.grad - t).abs().max()) for p, t in zip(params, want))
        res["missing"] = sorted(bucket.resolve_missing())
    elif case in ("two_backwards_undeclared", "two_backwards_declared"):
        declared = case.endswith("_declared")
        bucket.overlap_early([table_a, table_b], backwards_per_step=2 if declared else 1)
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            for it in range(3):
                bucket.zero()
                loss_fn(scale=1.0).backward()                 # virtual-view backward ...
                loss_fn(scale=0.5, use_pose=False).backward()  # ... and real-view backward before ONE optimiser step
                bucket.allreduce_mean()
                a, b = local_grads(scale=1.0), local_grads(scale=0.5, use_pose=False)
                want = mean_over_ranks([u + v for u, v in zip(a, b)])
                res["err%d" % it] = max(float((p.grad - t).abs().max()) for p, t in zip(params, want))
        res["warned"] = any("backwards_per_step" in str(c.message) for c in caught)
        res["disabled"] = bucket._early_disabled
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", ["asymmetric_none", "early_param_missing_on_one_rank", "two_backwards_undeclared",
                                  "two_backwards_declared"])
def test_exchange_patterns_keep_replicas_identical(case):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_pattern_worker, args=(2, _free_port(), out, case), nprocs=2, join=True)
    r0, r1 = out[0], out[1]
    errs = [v for r in (r0, r1) for k, v in r.items() if k.startswith("err")]
    assert errs and max(errs) <= 1e-6, (r0, r1)
    if case == "asymmetric_none":
        assert r0["missing"] == [] and r1["missing"] == []            # a gradient on ANY rank steps the parameter everywhere
        assert r0["missing_all"] == [3] and r1["missing_all"] == [3]   # none anywhere: skipped everywhere
        assert r0["counts"] == r1["counts"] == [2.0, 2.0, 2.0, 1.0, 0.0]          # [per parameter | the spare zero slot]
        assert r0["counts_all"] == r1["counts_all"] == [2.0, 2.0, 2.0, 0.0, 0.0]
    if case == "early_param_missing_on_one_rank":
        assert r0["missing"] == [] and r1["missing"] == []
    if case == "two_backwards_undeclared":
        assert r0["warned"] and r1["warned"] and r0["disabled"] and r1["disabled"]
    if case == "two_backwards_declared":
        assert not r0["warned"] and not r0["disabled"] and not r1["disabled"]


# ---------------------------------------------------------------------------------------------------------------------------
# NCCL's async semantics without NCCL: `dist.all_reduce(..., async_op=True)` on the nccl backend returns at once, the reduction
# is ordered on a stream, and `work.wait()` does NOT block the host -- it only makes the current stream wait.  gloo (every other
# test of this file) completes the data before `wait()` returns on the host, so code that reads or rescales the early range
# between the async call and `wait()`, or that relies on `wait()` as a host barrier, passes under gloo and breaks under RCCL.
# Here all_reduce is replaced by a stand-in whose asynchronous form applies the reduction ONLY when `wait()` is called (and never
# blocks), and which fails if the tensor was touched in between.
class _StreamOrderedWork:
    def __init__(self, tensor, log):
        self.tensor, self.log, self.snapshot, self.done = tensor, log, tensor.clone(), False

    def wait(self):
        assert not self.done
        assert torch.equal(self.tensor, self.snapshot), "the early range was modified while its collective was in flight"
        self.tensor.mul_(2.0)                 # "sum over two ranks holding the same gradient"
        self.done = True
        self.log.append(("wait", self.tensor.numel()))
        return True

    def is_completed(self):
        raise AssertionError("GradBucket must not poll: nccl work completion is stream-ordered, not a host event")


@pytest.mark.parametrize("backwards", [1, 2])
def test_early_exchange_assumes_no_host_wait(monkeypatch, backwards):
    from morpheus_amd import dist as mdist
    log, works = [], []

    def fake_all_reduce(t, op=None, async_op=False):
        if async_op:
            w = _StreamOrderedWork(t, log)
            works.append(w)
            log.append(("async", t.numel()))
            return w
        t.mul_(2.0)
        log.append(("sync", t.numel()))
        return None

    monkeypatch.setattr(mdist.dist, "all_reduce", fake_all_reduce)
    monkeypatch.setattr(mdist.dist, "get_world_size", lambda *a, **k: 2)
    monkeypatch.setattr(mdist.GradBucket, "_multi_rank", staticmethod(lambda: True))
    torch.manual_seed(0)
    table = torch.nn.Parameter(torch.randn(32, 2))
    net = torch.nn.Linear(4, 3)
    unused = torch.nn.Parameter(torch.zeros(5))           # no gradient on this rank: its flag must travel as 0
    params = [table] + list(net.parameters()) + [unused]
    bucket = mdist.GradBucket(params)
    bucket.overlap_early([table], backwards_per_step=backwards)
    x = torch.randn(16, 4)
    idx = torch.arange(16) % 32
    for step in range(2):
        bucket.zero()
        del log[:]
        for _ in range(backwards):
            ((net(x) + table[idx].sum(-1, keepdim=True)) ** 2).mean().backward()
        assert log == [("async", 64)], log              # fired from the hook of the LAST declared backward, nothing waited for yet
        assert not works[-1].done
        ref = [p.grad.detach().clone() if p.grad is not None else None for p in params]
        ref_table = bucket.flat[:64].clone()             # the early range as the hooks left it (p.grad was taken over)
        bucket.allreduce_mean()
        # [early async] ... [remainder + flags, sync] [wait] and only then the division by the world size
        assert [e[0] for e in log] == ["async", "sync", "wait"], log
        assert works[-1].done
        assert torch.allclose(table.grad.reshape(-1), ref_table)                    # (g + g) / 2
        assert torch.allclose(net.weight.grad, ref[1]) and torch.allclose(net.bias.grad, ref[2])
        counts = bucket.grad_counts
        assert counts[:3].tolist() == [2.0, 2.0, 2.0] and counts[3].item() == 0.0 and counts[4].item() == 0.0
        assert bucket.resolve_missing() == {3}
    # one flag write per step, whatever the number of gradient-less parameters
    assert len(bucket.__dict__["_missing_idx_cache"]) == 1
