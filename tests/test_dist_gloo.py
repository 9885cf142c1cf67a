"""CPU, world_size 2, gloo: the N>1 path of bench.py -- rays sharded per rank, replicated parameters,
ONE all-reduce of the flat gradient bucket -- gives every rank the mean gradient and identical
parameters after the optimiser step."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
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
