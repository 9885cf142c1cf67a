"""GPU-box probe: are memset nodes ordered with their neighbours when a captured HIP graph is replayed?"""
import ctypes, sys, torch
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
DEV = torch.device("cuda", 0)


def probe(nbytes, n_before, n_after, use_memset=True):
    n = nbytes // 4
    buf = torch.zeros(n, dtype=torch.float32, device=DEV)
    pad = torch.zeros(64, device=DEV)
    out = torch.zeros(n, device=DEV)
    def body():
        for _ in range(n_before):
            pad.add_(1.0)
        buf.fill_(5.0)                                   # what a previous owner of the memory left behind
        if use_memset:
            hip.hipMemsetAsync(buf.data_ptr(), 0, nbytes, torch.cuda.current_stream().cuda_stream)
        else:
            buf.zero_()
        buf.add_(1.0)                                    # the accumulate that relies on the zeroing
        out.copy_(buf)
        for _ in range(n_after):
            pad.add_(1.0)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    res = []
    for _ in range(4):
        g.replay()
        torch.cuda.synchronize()
        res.append((float(out.min()), float(out.max())))
    return res


for nbytes in (12, 4096, 1 << 22):
    for nb, na in ((0, 0), (50, 50), (3000, 3000)):
        for um in (True, False):
            print("bytes", nbytes, "kernels before/after", nb, na, "hipMemsetAsync" if um else "zero_()", probe(nbytes, nb, na, um), flush=True)

# torch's own memset: the semaphores of a global (multi-block, one output) reduction, in memory that a freed tensor just left
x = torch.rand(1 << 22, device=DEV)
want = float(x.double().sum())
def body2():
    outs = []
    for k in range(8):
        t = torch.full((128,), 7, dtype=torch.int32, device=DEV)
        del t
        outs.append(x.sum())
    return torch.stack(outs)
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    body2()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    r = body2()
for _ in range(4):
    g.replay(); torch.cuda.synchronize()
    print("global-reduce sums", [round(float(v) / want, 6) for v in r])
