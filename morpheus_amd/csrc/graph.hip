// HIP-graph hygiene for the captured training step (trainstep.GraphedRealViewStep).
//
// Measured on ROCm 7.2 / gfx950 (tools/gpu/graph_memset_probe.py, profiles/r03_graph_memset_probe.txt): a SMALL memset node of a
// captured graph (hipMemsetAsync of 12 B or 4 KB under stream capture; 4 MB is fine) does its work on the first launch of the
// instantiated graph and writes garbage -- or nothing -- on every later one.  This library zeroes with a kernel (mh_zero_async),
// but the step also runs PyTorch operators that memset under the hood: every multi-block single-output reduction
// (`x.sum()` over ~10^5 sample points, forward and in autograd's broadcast gradients) zeroes its inter-block semaphores that
// way, and from the second replay on such a sum returns a partial value.  mh_graph_replace_memset_nodes rewrites the captured
// graph before it is instantiated: each memset node becomes a kernel node with the same destination, pattern, predecessors
// and successors.
#include <vector>

#include "common.h"

__global__ __launch_bounds__(256) void graph_fill_kernel(unsigned char *dst, uint32_t value, uint32_t elem, uint64_t width,
                                                         uint64_t height, uint64_t pitch) {
    const uint64_t total = width * height;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (uint64_t)gridDim.x * 256) {
        const uint64_t row = i / width, col = i - row * width;
        unsigned char *p = dst + row * pitch + col * elem;
        if (elem == 4) *reinterpret_cast<uint32_t *>(p) = value;
        else if (elem == 2) *reinterpret_cast<uint16_t *>(p) = (uint16_t)value;
        else *p = (unsigned char)value;
    }
}

static bool graph_nodes(hipGraph_t g, std::vector<hipGraphNode_t> &nodes) {
    size_t n = 0;
    if (hipGraphGetNodes(g, nullptr, &n) != hipSuccess) return false;
    nodes.resize(n);
    if (n && hipGraphGetNodes(g, nodes.data(), &n) != hipSuccess) return false;
    nodes.resize(n);
    return true;
}

extern "C" int mh_graph_count_memset_nodes(void *graph, int64_t *n_nodes, int64_t *n_memset, int64_t *smallest_bytes) {
    if (!graph || !n_nodes || !n_memset || !smallest_bytes) return MH_ERR_ARG;
    std::vector<hipGraphNode_t> nodes;
    if (!graph_nodes(reinterpret_cast<hipGraph_t>(graph), nodes)) return MH_ERR_LAUNCH;
    *n_nodes = (int64_t)nodes.size();
    *n_memset = 0;
    *smallest_bytes = 0;
    for (hipGraphNode_t nd : nodes) {
        hipGraphNodeType ty;
        if (hipGraphNodeGetType(nd, &ty) != hipSuccess) return MH_ERR_LAUNCH;
        if (ty != hipGraphNodeTypeMemset) continue;
        hipMemsetParams mp;
        if (hipGraphMemsetNodeGetParams(nd, &mp) != hipSuccess) return MH_ERR_LAUNCH;
        const int64_t bytes = (int64_t)(mp.width * mp.elementSize * (mp.height ? mp.height : 1));
        if (*n_memset == 0 || bytes < *smallest_bytes) *smallest_bytes = bytes;
        ++*n_memset;
    }
    return MH_OK;
}

extern "C" int mh_graph_replace_memset_nodes(void *graph, int64_t *n_replaced) {
    if (!graph || !n_replaced) return MH_ERR_ARG;
    hipGraph_t g = reinterpret_cast<hipGraph_t>(graph);
    std::vector<hipGraphNode_t> nodes;
    if (!graph_nodes(g, nodes)) return MH_ERR_LAUNCH;
    *n_replaced = 0;
    for (hipGraphNode_t nd : nodes) {
        hipGraphNodeType ty;
        if (hipGraphNodeGetType(nd, &ty) != hipSuccess) return MH_ERR_LAUNCH;
        if (ty != hipGraphNodeTypeMemset) continue;
        hipMemsetParams mp;
        if (hipGraphMemsetNodeGetParams(nd, &mp) != hipSuccess) return MH_ERR_LAUNCH;
        if (mp.elementSize != 1 && mp.elementSize != 2 && mp.elementSize != 4) return MH_ERR_ARG;
        size_t n_in = 0, n_out = 0;
        if (hipGraphNodeGetDependencies(nd, nullptr, &n_in) != hipSuccess) return MH_ERR_LAUNCH;
        if (hipGraphNodeGetDependentNodes(nd, nullptr, &n_out) != hipSuccess) return MH_ERR_LAUNCH;
        std::vector<hipGraphNode_t> in(n_in), out(n_out);
        if (n_in && hipGraphNodeGetDependencies(nd, in.data(), &n_in) != hipSuccess) return MH_ERR_LAUNCH;
        if (n_out && hipGraphNodeGetDependentNodes(nd, out.data(), &n_out) != hipSuccess) return MH_ERR_LAUNCH;

        unsigned char *dst = static_cast<unsigned char *>(mp.dst);
        uint32_t value = mp.value, elem = mp.elementSize;
        uint64_t width = mp.width, height = mp.height ? mp.height : 1, pitch = mp.pitch;
        if (height == 1) pitch = width * elem;
        void *args[] = {&dst, &value, &elem, &width, &height, &pitch};
        const uint64_t total = width * height;
        uint64_t blocks = (total + 255) / 256;
        if (blocks < 1) blocks = 1;
        if (blocks > 4096) blocks = 4096;
        hipKernelNodeParams kp;
        kp.func = reinterpret_cast<void *>(graph_fill_kernel);
        kp.gridDim = dim3((unsigned)blocks);
        kp.blockDim = dim3(256);
        kp.sharedMemBytes = 0;
        kp.kernelParams = args;
        kp.extra = nullptr;
        hipGraphNode_t fresh;
        if (hipGraphAddKernelNode(&fresh, g, n_in ? in.data() : nullptr, n_in, &kp) != hipSuccess) return MH_ERR_LAUNCH;
        for (hipGraphNode_t succ : out)
            if (hipGraphAddDependencies(g, &fresh, &succ, 1) != hipSuccess) return MH_ERR_LAUNCH;
        if (hipGraphDestroyNode(nd) != hipSuccess) return MH_ERR_LAUNCH;   // takes the node's own edges with it
        ++*n_replaced;
    }
    return MH_OK;
}
