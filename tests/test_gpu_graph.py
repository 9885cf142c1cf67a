"""GPU: HIP-graph hygiene (csrc/graph.hip).  A captured graph whose small memset nodes were replaced by fill kernels replays
the same values every time -- the raw graph does not on ROCm 7.2, which is what broke the captured training step (its loss,
summed over ~10^5 sample points by a multi-block torch reduction whose semaphores are zeroed by a memset node, came back
partial from the second replay on)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _capture(body, sanitize):
    from morpheus_amd import ops
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(g):
        out = body()
    nodes, memsets, smallest = ops.graph_memset_nodes(g)
    replaced = ops.graph_replace_memset_nodes(g) if sanitize else 0
    after = ops.graph_memset_nodes(g)
    g.instantiate()
    return g, out, (nodes, memsets, smallest), replaced, after


@pytest.mark.parametrize("nbytes", [12, 4096, 1 << 22])
def test_memset_nodes_become_fill_kernels_and_replay_correctly(nbytes):
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    n = nbytes // 4
    buf = torch.zeros(n, device=DEV)
    out = torch.zeros(n, device=DEV)
    bytes_ = torch.zeros(nbytes, dtype=torch.uint8, device=DEV)

    def body():
        buf.fill_(5.0)                                   # what a previous owner of the memory left behind
        assert hip.hipMemsetAsync(buf.data_ptr(), 0, nbytes, torch.cuda.current_stream().cuda_stream) == 0
        buf.add_(1.0)                                    # the accumulation that relies on the zeroing
        out.copy_(buf)
        assert hip.hipMemsetAsync(bytes_.data_ptr(), 0xA7, nbytes, torch.cuda.current_stream().cuda_stream) == 0   # a byte pattern
        return out

    g, _, (nodes, memsets, smallest), replaced, after = _capture(body, sanitize=True)
    assert memsets == 2 and smallest == nbytes and replaced == 2 and after[1] == 0 and after[0] == nodes
    for _ in range(4):
        bytes_.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert float(out.min()) == 1.0 and float(out.max()) == 1.0
        assert int(bytes_.min()) == 0xA7 and int(bytes_.max()) == 0xA7
    del g


def test_multi_block_torch_reductions_survive_replay():
    """the pattern of the training step: a freed small tensor's block is handed to the reduction's semaphores"""
    x = torch.rand(1 << 22, device=DEV)
    w = torch.rand(1 << 22, device=DEV, requires_grad=True)
    want = float((x.double() * w.detach().double()).sum())

    def body():
        outs = []
        for _ in range(8):
            t = torch.full((128,), 7, dtype=torch.int32, device=DEV)
            del t
            outs.append((x * w).sum())
        return torch.stack(outs).detach()

    g, r, (nodes, memsets, smallest), replaced, after = _capture(body, sanitize=True)
    assert memsets >= 8 and replaced == memsets and after[1] == 0, (memsets, replaced, after)
    for _ in range(4):
        g.replay()
        torch.cuda.synchronize()
        assert torch.allclose(r.double().cpu(), torch.full((8,), want, dtype=torch.float64), rtol=1e-5), (r.tolist(), want)
    del g
