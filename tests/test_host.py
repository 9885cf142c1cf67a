"""CPU (no GPU): host-side logic of the product path -- the C-ABI library loads and exports every
symbol the header declares, the product path refuses CPU tensors (no fallback), state_dict /
parameter-group compatibility with the reference, the MFMA packing maps, level tables."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from morpheus_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from morpheus_amd import _lib, build
    build.build()
    return _lib.load()


def test_library_exports_every_declared_symbol(lib):
    from morpheus_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "morpheus_hip.h")).read()
    declared = set(re.findall(r"\b(mh_[a-zA-Z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    raw = ctypes.CDLL(os.path.join(ROOT, "morpheus_amd", "_build", "libmorpheus_hip.so"))
    for name in declared:
        assert hasattr(raw, name), f"{name} declared in include/morpheus_hip.h but not exported"
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert lib.mh_abi_version() == 8
    assert lib.mh_status_string(1).decode().startswith("invalid argument")
    # size queries are pure host functions
    assert lib.mh_mlp_tiles(1) == 4 and lib.mh_mlp_tiles(129) == 8
    assert lib.mh_warp_acts_floats(128) == 4 * (64 + 2 * 640 + 40) * 32      # activations + 40 rows of ReLU masks
    assert lib.mh_field_acts_floats(128) == 4 * (96 + 64 * 5 + 8) * 32
    assert lib.mh_grid_bin_bricks() == 4096 and lib.mh_grid_bin_index_ints() == 2 * 4096 + 8


def test_argument_validation_without_gpu(lib):
    """status codes, never exceptions or launches, for bad arguments"""
    assert lib.mh_grid_encode_fwd(None, None, None, None, None, 10, 16, 16, 1.01, 1, None) == 1
    assert lib.mh_grid_encode_fwd(None, None, None, None, None, 0, 16, 16, 1.01, 1, None) == 0      # empty input is fine
    assert lib.mh_grid_encode_fwd_binned(None, None, None, None, None, None, None, 10, 16, 16, 1.01, None) == 1
    assert lib.mh_grid_encode_fwd_binned(None, None, None, None, None, None, None, 0, 16, 16, 1.01, None) == 0
    # the call-size knob of the brick-staged hash-grid forms: a process-wide host value, set / queried without a device
    before = lib.mh_grid_stage_min_points(-1)
    assert before == 1 << 20 or "MORPHEUS_GRID_STAGE_MIN_POINTS" in os.environ
    assert lib.mh_grid_stage_min_points(12345) == 12345 and lib.mh_grid_stage_min_points(-1) == 12345
    assert lib.mh_grid_stage_min_points(before) == before
    assert lib.mh_composite_fwd(*([None] * 10), 0, None) == 0
    assert lib.mh_composite_fwd(*([None] * 10), 5, None) == 1
    assert lib.mh_warp_fwd(*([None] * 8), 6, None, None, None, 128, None) == 1
    assert lib.mh_field_fwd(*([None] * 7), 7, 1, None, None, None, None, 128, None) == 1        # n_bands > 6
    # round-3 entry points: the caller-side glue kernels and the graph pass
    assert lib.mh_masked_mean_fwd(9, None, None, None, 0, 1, None, None, None, None) == 1                # unknown kind
    assert lib.mh_masked_mean_fwd(4, None, None, None, 0, 2, None, None, None, None) == 1                # eikonal rows are [M,3]
    assert lib.mh_masked_mean_fwd(5, None, None, None, 8, 3, None, None, None, None) == 1                # |a - b| needs b
    assert lib.mh_masked_mean_bwd(2, None, None, None, 0, 1, None, None, None, None, None, None) == 1    # no output asked for
    assert lib.mh_masked_mean_workspace_floats() == 1024
    assert lib.mh_ortho_perturb_fwd(None, None, None, 0.1, 0, None, None) == 0 and lib.mh_ortho_perturb_fwd(None, None, None, 0.1, 4, None, None) == 1
    assert lib.mh_pose_apply_fwd(None, None, None, None, 0, 128, None, None, None) == 0
    assert lib.mh_pose_apply_fwd(None, None, None, None, 2, 128, None, None, None) == 1
    assert lib.mh_pose_apply_bwd(None, None, None, 70000, 1, 10, None, None, None, None, None) == 1      # more rows than a grid dimension
    assert lib.mh_pose_bwd_workspace_floats(3, 2500) == 3 * 3 * 12
    assert lib.mh_render_loss_fwd(*([None] * 9), 5, 1.0, 1.0, 1.0, None, None, None, None) == 1
    assert lib.mh_render_loss_bwd(*([None] * 7), 0, 1.0, 1.0, 1.0, None, None, None, None, None) == 0
    assert lib.mh_graph_count_memset_nodes(None, None, None, None) == 1 and lib.mh_graph_replace_memset_nodes(None, None) == 1


def test_no_cpu_fallback():
    from morpheus_amd import harness, ops
    from morpheus_amd._lib import MorpheusHipError
    model = harness.build_model("a", "cpu")
    x = torch.zeros(8, 3)
    with pytest.raises(MorpheusHipError):
        model(x, torch.zeros(8, 1))
    with pytest.raises(MorpheusHipError):
        model.encoder(x, bound=1.01)
    with pytest.raises(MorpheusHipError):
        ops.composite(torch.zeros(4), torch.zeros(4), torch.ones(4), torch.zeros(4, 3), torch.zeros(1, dtype=torch.int32),
                      torch.full((1,), 4, dtype=torch.int32))
    # nothing under morpheus_amd/ imports the oracle
    for fn in os.listdir(os.path.join(ROOT, "morpheus_amd")):
        if fn.endswith(".py"):
            src = open(os.path.join(ROOT, "morpheus_amd", fn)).read()
            assert "import oracle" not in src and "from oracle" not in src, fn


def test_state_dict_and_param_groups_match_reference_layout():
    from morpheus_amd import harness
    model = harness.build_model("b", "cpu")
    want = synth.make_state("b")
    sd = model.state_dict()
    assert set(sd.keys()) == set(want.keys()), set(sd.keys()) ^ set(want.keys())
    for k, v in want.items():
        assert sd[k].shape == v.shape and sd[k].dtype == v.dtype, k
        assert torch.equal(sd[k], v), k
    groups = model.get_params_all(5e-4)
    assert [g["name"] for g in groups] == ["encoder_sdf", "encoder_color", "decoder_sdf", "decoder_topo", "decoder_color",
                                           "density", "decoder_deform", "code_deform", "pose", "decoder_bg"]
    lrs = {g["name"]: g["lr"] for g in groups}
    assert lrs["density"] == 2.5e-4 and lrs["pose"] == 5e-5
    n_in_groups = sum(p.numel() for g in model.get_params_all(1.0) for p in g["params"])
    assert n_in_groups == sum(p.numel() for p in model.parameters())
    assert float(model.sdf2density.get_beta()) == pytest.approx(0.1001)
    assert model.encoder._res_np.tolist() == [16, 19, 22, 25, 28, 32, 37, 43, 49, 56, 64, 74, 85, 98, 112, 128]
    # a switch the reference itself cannot run fails loudly instead of silently taking another path
    from morpheus_amd.model import scene_representation
    with pytest.raises(NotImplementedError):
        scene_representation(model.config, 1.01, num_frames=200, encode_deform=False, use_joint=True)
    assert not model.composed_field


def _mfma_emulate(wpack, KS, MT, bin_lanes):
    """numpy model of csrc/mlp.hip:mfma_layer -- v_mfma_f32_32x32x2_f32 semantics:
    A[i][k] from lane i+32k, B[k][j] from lane j+32k, D[row][col] in lane col+32*((row>>2)&1), reg (row&3)+4*(row>>3)."""
    wp = wpack.reshape(MT, KS // 4, 64, 4)
    acc = np.zeros((64, MT, 16), dtype=np.float64)
    for mt in range(MT):
        D = np.zeros((32, 32))
        for kk in range(KS):
            a = wp[mt, kk // 4, :, kk % 4]
            b = bin_lanes[:, kk]
            A = np.stack([a[:32], a[32:]], 1)           # [32 rows, k=2]
            B = np.stack([b[:32], b[32:]], 0)           # [k=2, 32 cols]
            D += A @ B
        for lane in range(64):
            for r in range(16):
                acc[lane, mt, r] = D[(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), lane & 31]
    return acc


def test_packing_maps_reproduce_a_linear_layer():
    """Forward pack of a 128->128 hidden layer + the accumulator-as-B-operand convention == W @ h."""
    from morpheus_amd import packing
    pk = packing.warp_packer(3)
    rng = np.random.RandomState(0)
    Ws = [rng.randn(s.out_dim, s.in_dim).astype(np.float32) for s in pk.specs]
    wp, wpT = pk.pack([torch.from_numpy(w) for w in Ws])
    wp, wpT = wp.numpy(), wpT.numpy()
    assert wp.shape[0] == 74752 and wpT.shape[0] == 77824
    # layer 1 (offset 5120): input h in accumulator layout
    h = rng.randn(128, 32)                                  # [feature, point]
    km = packing.kmap_acc(4)
    bin_lanes = np.zeros((64, 64))
    for lane in range(64):
        for kk in range(64):
            bin_lanes[lane, kk] = h[km[kk, lane >> 5], lane & 31]
    acc = _mfma_emulate(wp[5120:5120 + 16384], 64, 4, bin_lanes)
    want = Ws[1].astype(np.float64) @ h
    for lane in (0, 5, 31, 32, 63):
        for t in range(4):
            for r in range(16):
                row = 32 * t + packing.acc_row(r, lane >> 5)
                assert abs(acc[lane, t, r] - want[row, lane & 31]) < 1e-4
    # transposed pack of the same layer (T1 is the 5th block of the backward stream): W^T @ g
    g = rng.randn(128, 32)
    for lane in range(64):
        for kk in range(64):
            bin_lanes[lane, kk] = g[km[kk, lane >> 5], lane & 31]
    off_T1 = 4096 + 3 * 16384
    accT = _mfma_emulate(wpT[off_T1:off_T1 + 16384], 64, 4, bin_lanes)
    wantT = Ws[1].astype(np.float64).T @ g
    for lane in (0, 17, 40, 63):
        for t in range(4):
            for r in range(16):
                row = 32 * t + packing.acc_row(r, lane >> 5)
                assert abs(accT[lane, t, r] - wantT[row, lane & 31]) < 1e-4


def test_packing_first_layers_and_gradient_unpack():
    from morpheus_amd import packing
    # every natural weight appears exactly once in the forward pack and once in the transposed pack
    for pk in (packing.warp_packer(3), packing.warp_packer(2), packing.field_packer()):
        for idx in (pk.fwd_index, pk.bwd_index):
            cnt = np.bincount(idx[idx < pk.n_weights], minlength=pk.n_weights)
            assert (cnt == 1).all(), (cnt != 1).sum()
        # dW / db gather indices are a bijection onto the natural layout
        assert pk.dw_index.shape[0] == pk.n_weights and len(np.unique(pk.dw_index)) == pk.n_weights
        assert pk.db_index.shape[0] == pk.n_biases and len(np.unique(pk.db_index)) == pk.n_biases
        assert pk.dw_index.max() < pk.raw_dw and pk.db_index.max() < pk.raw_db
    # frequency-encoding k-steps cover the reference's 39-vector exactly once
    enc = packing.kmap_enc20()
    assert sorted(enc[enc >= 0].tolist()) == list(range(39))
    # field: sdf layer-0 k-steps cover the 73 inputs, colour layer-0 the 64 inputs
    s0, c0 = packing.field_specs()[0], packing.field_specs()[3]
    assert sorted(s0.kmap[s0.kmap >= 0].tolist()) == list(range(73))
    assert sorted(c0.kmap[c0.kmap >= 0].tolist()) == list(range(64))


def test_level_tables_agree_with_oracle():
    from morpheus_amd import ops
    from oracle.hashgrid import effective_levels, level_resolutions
    _, s = synth.grid_offsets()
    assert ops.level_resolutions(16, s, 16).tolist() == level_resolutions(16, s, 16).tolist()
    for ml in (None, 0.01, 0.5, 0.75, 1.0):
        assert ops.effective_levels(ml, 16) == effective_levels(ml, 16)


def test_config_schema_matches_reference_keys():
    from morpheus_amd import harness
    cfg = harness.load_config("snoopy")
    assert set(cfg) == {"data", "exp", "render", "train", "model", "guidance"}
    assert cfg["render"]["step_size"] == 0.01 and cfg["model"]["use_joint"] is True and cfg["exp"]["fp16"] is False
    for k in ("trunc", "normal_smoothness", "normal_smooth_3d", "code_reg", "ori_weight", "topo_none", "smoothness_std"):
        assert k in cfg["train"]


def test_joint_packer_equals_per_net_packers():
    """One gather each way (JointPacker) produces exactly the per-net packs and gradient un-packs."""
    from morpheus_amd.packing import field_joint_packer, field_packer, warp_joint_packer, warp_packer
    g = torch.Generator().manual_seed(5)
    rn = lambda *s: torch.randn(*s, generator=g)

    def net(n_out):
        return ([rn(128, 39)] + [rn(128, 128) for _ in range(4)] + [rn(n_out, 128)],
                [rn(128) for _ in range(5)] + [rn(n_out)])
    (Wd, Bd), (Wt, Bt) = net(3), net(2)
    jp, pd, pt = warp_joint_packer(), warp_packer(3), warp_packer(2)
    f, b = jp.pack([Wd, Wt], [Bd, Bt])
    for k, (pk, W, B) in enumerate(((pd, Wd, Bd), (pt, Wt, Bt))):
        w, wT = pk.pack(W)
        assert torch.equal(jp.take(f, jp.w[k]), w) and torch.equal(jp.take(b, jp.wT[k]), wT)
        assert torch.equal(jp.take(f, jp.b[k]), pk.pack_biases(B, skip_first=True))
    raw = rn(jp.raw_len)
    (gwd, gwt), (gbd, gbt) = jp.unpack_grads(raw)
    n_dw = pd.raw_dw + pt.raw_dw
    a, ab = pd.unpack_grads(raw[:pd.raw_dw], raw[n_dw:n_dw + pd.raw_db])
    c, cb = pt.unpack_grads(raw[pd.raw_dw:n_dw], raw[n_dw + pd.raw_db:])
    assert all(torch.equal(x, y) for x, y in zip(gwd + gwt + gbd + gbt, a + c + ab + cb))
    Ws = [rn(64, 73), rn(64, 64), rn(33, 64), rn(64, 64), rn(64, 64), rn(3, 64)]
    Bs = [rn(64), rn(64), rn(33), rn(64), rn(64), rn(3)]
    fj, fp = field_joint_packer(), field_packer()
    f, b = fj.pack([Ws], [Bs])
    w, wT = fp.pack(Ws)
    assert torch.equal(fj.take(f, fj.w[0]), w) and torch.equal(fj.take(b, fj.wT[0]), wT)
    assert torch.equal(fj.take(f, fj.b[0]), fp.pack_biases(Bs, skip_first=False))


def test_raw_gradient_map_with_bias0_dropped():
    """JointPacker.unpack_grads(zero_bias0=True): identical to the plain map except that every net's first-layer bias
    gradient is zero (it reaches its parameter through the per-frame bias0), and bias0_raw points at those entries."""
    import torch
    from morpheus_amd.packing import warp_joint_packer
    jp = warp_joint_packer()
    raw = torch.randn(jp.raw_len, generator=torch.Generator().manual_seed(3))
    (wa, ba), (wb, bb) = jp.unpack_grads(raw), jp.unpack_grads(raw, zero_bias0=True)
    for net in range(2):
        for l in range(6):
            assert torch.equal(wa[net][l], wb[net][l])
            if l == 0:
                assert float(bb[net][0].abs().sum()) == 0.0 and float(ba[net][0].abs().sum()) > 0
                assert torch.equal(ba[net][0], raw[jp.bias0_raw[net]:jp.bias0_raw[net] + 128])
            else:
                assert torch.equal(ba[net][l], bb[net][l])


def test_b3_fragment_index_matches_the_mfma_layout():
    """packing._frag_index_b3 against a scalar emulation of v_mfma_f32_32x32x16_bf16's operand layout (lane (i, g) supplies
    A[m = i][k = 8g + e]; the B operand of (s, g, e) is what the chain carries in k-step 8s + e of lane half g): layer 1
    (accumulator inputs) and layer 0 (frequency-encoding inputs) of the deform net reproduce W . x."""
    import numpy as np
    import torch
    from morpheus_amd.packing import acc_row, kmap_enc20, warp_joint_packer
    jp = warp_joint_packer()
    g = torch.Generator().manual_seed(0)
    nets = []
    for nout in (3, 2):
        W = [torch.randn(128, 39, generator=g)] + [torch.randn(128, 128, generator=g) for _ in range(4)] + [torch.randn(nout, 128, generator=g)]
        b = [torch.randn(128, generator=g) for _ in range(5)] + [torch.randn(nout, generator=g)]
        nets.append((W, b))
    flat = jp.flat([n[0] for n in nets], [n[1] for n in nets])
    src = flat[torch.from_numpy(jp.fwd3_index)].numpy()
    rng = np.random.default_rng(1)
    # a 128 x 128 layer is staged as two k-half blocks [khalf][out tile][k16 step 0..3][lane][8] (b3_layers[1], [2])
    X = rng.standard_normal(128).astype(np.float32)
    out = np.zeros(128)
    for kh in range(2):
        so, n, _ = jp.b3_layers[1 + kh]
        frag = src[so:so + n].reshape(4, 4, 64, 8)
        for mt in range(4):
            for i in range(32):
                for sp in range(4):
                    s = 4 * kh + sp
                    for gg in range(2):
                        for e in range(8):
                            out[32 * mt + i] += frag[mt, sp, 32 * gg + i, e] * X[32 * (s >> 1) + acc_row(8 * (s & 1) + e, gg)]
    assert np.abs(out - nets[0][0][1].numpy() @ X).max() < 1e-4
    so, n, _ = jp.b3_layers[0]
    frag = src[so:so + n].reshape(4, 3, 64, 8)
    enc = rng.standard_normal(39).astype(np.float32)
    km = kmap_enc20()
    out = np.zeros(128)
    for mt in range(4):
        for i in range(32):
            for s in range(3):
                for gg in range(2):
                    for e in range(8):
                        kk = 8 * s + e
                        if kk < 20 and km[kk, gg] >= 0:
                            out[32 * mt + i] += frag[mt, s, 32 * gg + i, e] * enc[km[kk, gg]]
    assert np.abs(out - nets[0][0][0].numpy() @ enc).max() < 1e-4
    # pack geometry the kernels hard-wire (csrc/mlp_b3.hip: B3_NET_F4, B3_NETT_F4)
    assert jp.w3 == [(0, 28672), (28672, 28672)] and jp.wT3 == [(0, 29184), (29184, 29184)]


def test_model_switches_shapes_and_refusals():
    """scene_representation's constructor switches (models/model.py:36-53) build the reference's state_dict shapes (the fixtures'
    states load strictly) and parameter groups; use_t / use_joint stay on the fused kernels, use_app / encode_topo /
    color_grid=False select the composed field path; the one switch the reference itself cannot run is refused loudly."""
    import pytest
    from morpheus_amd import harness, synth
    from morpheus_amd.model import scene_representation
    cfg = harness.load_config()
    base = dict(num_frames=200, deform_dim=16, amb_dim=2)
    off = dict(use_t=False, use_joint=True, use_app=False, encode_topo=False, color_grid=True)
    #        switches                                   deform in, sdf in, colour in, composed
    cases = ((dict(use_t=True, use_joint=True), 100, 73, 64, False), (dict(use_t=False, use_joint=False), 87, 37, 64, False),
             (dict(use_t=True, use_joint=False), 100, 37, 64, False), (dict(use_app=True), 87, 73, 112, True),
             (dict(encode_topo=True), 87, 89, 64, True), (dict(color_grid=False), 87, 73, 71, True),
             (dict(use_app=True, encode_topo=True, color_grid=False, use_joint=False), 87, 53, 119, True))
    for sw, d0, s0, c0, composed in cases:
        m = scene_representation(cfg, 1.01, **base, **dict(off, **sw))
        m.load_state_dict(synth.variant_state("b", 200, **sw), strict=True)
        assert tuple(m.deform_net.net[0].weight_v.shape) == (128, d0) and tuple(m.topo_net.net[0].weight_v.shape) == (128, d0)
        assert tuple(m.sdf_net.net[0].weight.shape) == (64, s0) and tuple(m.color_net.net[0].weight_v.shape) == (64, c0)
        assert m.composed_field == composed
        names = [g["name"] for g in m.get_params_all(1e-3)]
        assert ("code_app" in names) == bool(sw.get("use_app")) and names[-1] == ("code_app" if sw.get("use_app") else "decoder_bg")
        assert sum(p.numel() for g in m.get_params_all(1.0) for p in g["params"]) == sum(p.numel() for p in m.parameters())
    with pytest.raises(NotImplementedError):
        scene_representation(cfg, 1.01, **base, **dict(off, encode_deform=False))


def test_weighted_sum_is_the_chain_of_adds_on_any_device():
    """ops.weighted_sum is plain torch (stack, multiply, add): the reference's `loss = loss + w * term` chains in three launches.
    Value, gradients, and the weight-vector cache stays bounded when a caller feeds it changing weights."""
    from morpheus_amd import ops
    ts = [torch.tensor(v, requires_grad=True) for v in (0.3, -1.2, 4.0, 0.07)]
    ws = (1.0, 0.5, 0.01, 30.0)
    got = ops.weighted_sum(list(zip(ws, ts)))
    want = sum(w * float(t) for w, t in zip(ws, ts))
    assert abs(float(got) - want) <= 1e-6 * abs(want)
    (got * 2.0).backward()
    for w, t in zip(ws, ts):
        assert abs(float(t.grad) - 2.0 * w) <= 1e-6 * abs(2.0 * w)
    assert ops.weighted_sum([]) == 0
    for k in range(200):
        ops.weighted_sum([(1.0 + k, ts[0].detach())])
    assert len(ops._WEIGHT_CACHE) <= 64


def test_i32arr_cache_keys_do_not_collide_across_source_kinds():
    """a list [3, 1, 2, 3] and the array [1, 2, 3] used to share one cache entry (tuple(a) vs (size,) + values)."""
    import numpy as np
    from morpheus_amd import ops
    a, _ = ops._i32arr([3, 1, 2, 3])
    b, _ = ops._i32arr(np.array([1, 2, 3], dtype=np.int32))
    c, _ = ops._i32arr((1, 2, 3))
    assert a.tolist() == [3, 1, 2, 3] and b.tolist() == [1, 2, 3] and c.tolist() == [1, 2, 3]
    assert ops._i32arr([3, 1, 2, 3])[0] is a                     # cached by value
    d, _ = ops._i32arr([1.5, 2])                                  # non-integers are converted, never used as a key
    assert d.tolist() == [1, 2]


def test_per_ray_jitter_length_is_checked():
    """the marcher indexes the jitter by ray without a length of its own; a buffer sized for another batch must raise."""
    import pytest
    import torch
    from morpheus_amd import ops
    assert ops._ray_jitter(None, 7) is None
    assert ops._ray_jitter(torch.zeros(7, 1), 7).shape == (7,)
    with pytest.raises(ValueError):
        ops._ray_jitter(torch.zeros(2048), 576)
    with pytest.raises(ValueError):
        ops._ray_jitter(torch.zeros(7, dtype=torch.float64), 7)


def test_virtual_view_step_host_logic():
    """trainstep.VirtualViewTrainStep's host side (morpheus.py:864-903): shading schedule, background draw, the guidance stand-in's
    gradient, the arithmetic-mode tag bound to a pack."""
    import types
    import torch
    from morpheus_amd import harness, ops
    from bench_support import trainstep
    cfg = harness.load_config()
    fake_model = types.SimpleNamespace(config=cfg)
    vs = trainstep.VirtualViewTrainStep(types.SimpleNamespace(model=fake_model, config=cfg, occupancy_grid=None), res=72, seed=1)
    vs.epoch = 100                                               # exp_iter_ratio 0.05 <= albedo_iter_ratio 0.1
    assert vs.get_shading() == (1.0, "albedo")
    vs.epoch = 1000
    draws = [vs.get_shading() for _ in range(400)]
    kinds = {s for _, s in draws}
    assert kinds == {"lambertian", "textureless"} and all(0.1 <= a <= 1.0 for a, _ in draws)
    frac = sum(s == "textureless" for _, s in draws) / 400
    assert 0.1 < frac < 0.3                                      # textureless_ratio 0.2
    bgs = [vs.get_bg_color("cpu") for _ in range(40)]
    assert any(b is None for b in bgs) and any(b is not None and b.shape == (3,) for b in bgs)
    g = trainstep.InjectedGuidance(8, 8, "cpu", scale=5e-3)
    img = torch.rand(1, 3, 8, 8, requires_grad=True)
    g(img).backward()
    assert torch.equal(img.grad, g.grad) and float(g.grad.abs().max()) <= 5e-3
    assert ops._warp_mode("f32") == "" and ops._warp_mode("b3") == "b3" and ops._warp_mode(None) == ops._warp_mode(ops.mlp_mode())


def test_query_accumulator_joins_by_identity_and_resets_per_backward_pass():
    """ops._QueryAccumulator (host logic only): a field query joins the in-place sums only when it was handed the very beta tensor
    and table parameters the operands were prepared with (a colourless query: its one table); outside a backward pass nothing
    accumulates; state left behind by one backward pass is dropped when the next one starts; the private autograd call it keys on
    is looked up once and its absence switches the in-place form off."""
    from morpheus_amd import ops
    beta, other_beta = torch.tensor(0.1), torch.tensor(0.1)
    ts, tc, tx = torch.zeros(4, 2), torch.zeros(4, 2), torch.zeros(4, 2)
    acc = ops._QueryAccumulator(beta, (ts, tc))
    assert acc.joins((id(beta), id(ts), id(tc))) and acc.joins((id(beta), id(ts), id(None)))
    assert not acc.joins((id(other_beta), id(ts), id(tc))) and not acc.joins((id(beta), id(tx), id(tc)))
    assert not acc.joins((id(beta), id(ts), id(tx))) and not ops._QueryAccumulator().joins((id(None), id(ts), id(tc)))
    assert ops._GRAPH_TASK_ID is not None and ops.ACCUMULATE_IN_PLACE
    assert acc.enter() is False                      # no backward pass running: nobody would collect the sums
    seen = []

    class Probe(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x * 2

        @staticmethod
        def backward(ctx, g):
            assert acc.enter()
            seen.append((acc.task, acc.raw))
            acc.raw = torch.ones(3)                  # what a query would leave behind
            w = acc.gmax_words("cpu")
            assert w.numel() == 2 and int(w.abs().sum()) == 0
            return g * 2

    for _ in range(2):
        x = torch.ones(2, requires_grad=True)
        (Probe.apply(x).sum() + Probe.apply(x).sum()).backward()
    (t0, r0), (t1, r1), (t2, r2), (t3, r3) = seen
    assert t0 == t1 and t2 == t3 and t0 != t2        # two nodes of one pass share the state, the next pass starts empty
    assert r0 is None and r1 is not None and r2 is None and r3 is not None
    raw, tabs = acc.collect()                        # a different (here: no) pass: nothing stale is handed over
    assert raw is None and tabs == [None, None]


def test_sliced_pack_sizes_match_the_kernels_staging_sizes():
    """The host side lays the bf16 x 3 operand packs out (packing.py), the kernels stage them into LDS by fixed offsets: the two must
    agree on every total, or a kernel would read past / short of a pack.  Host functions of the library, no GPU."""
    from morpheus_amd import _lib, packing
    lib = _lib.load()
    fj, wj = packing.field_joint_packer(), packing.warp_joint_packer()
    assert fj.fwd3_total_f4 * 16 == lib.mh_field_w3_bytes() and fj.bwd3_total_f4 * 16 == lib.mh_field_w3T_bytes()
    assert wj.fwd3_total_f4 * 16 == 2 * lib.mh_warp_w3_bytes() and wj.bwd3_total_f4 * 16 == 2 * lib.mh_warp_w3T_bytes()
    # the transposed layers are sliced for the bf16 x 3 kernels only (f32 reads the fp32 transposed pack)
    assert fj.sliced_bwd_for("b3") and fj.sliced_bwd_for(True) and not fj.sliced_bwd_for("f32")
    assert wj.sliced_bwd_for("b3") and not wj.sliced_bwd_for("")


def test_step_cache_host_logic():
    """model._step_cache (round 6), the parts that need no GPU: nothing is kept outside a training forward (eval mode, no_grad, the
    switch off), a parameter's version counter or a train() / eval() call ends the step, an explicit operand_scope of a training
    forward stands on the step cache and leaves it in the model for the next call, a fresh scope never does, and the merged
    operand gather of a pack addresses the four section layouts the separate gathers had."""
    import torch
    from morpheus_amd import harness, model as mm, packing
    m = harness.build_model("b", "cpu")
    m.eval()
    assert m._step_cache() is None
    m.train()
    with torch.no_grad():
        assert m._step_cache() is None
    sc = m._step_cache()
    assert sc is not None and m._step_cache() is sc and not sc.stale
    with m.operand_scope():
        assert m._opcache is sc.entries and m._scope_step is sc
        with m.fresh_operand_scope():
            assert m._opcache is not sc.entries and m._scope_step is None
        assert m._opcache is sc.entries
    assert m._opcache is None and m._stepcache is sc
    with torch.no_grad():
        next(iter(m.sdf_net.parameters())).add_(0.0)           # an in-place update, as an optimiser step makes it
    sc2 = m._step_cache()
    assert sc2 is not sc
    sc2._spent()
    assert sc2.stale and m._step_cache() is not sc2             # the backward pass reached a pack: the next forward re-prepares
    m.eval()
    assert m._stepcache is None
    m.train()
    saved = mm.IMPLICIT_OPERANDS
    try:
        mm.IMPLICIT_OPERANDS = False
        assert m._step_cache() is None
        with m.operand_scope():
            assert m._opcache == {} and m._scope_step is None
    finally:
        mm.IMPLICIT_OPERANDS = saved
    for jp in (packing.warp_joint_packer(), packing.field_joint_packer()):
        idx, sect = jp.all_index()
        for name, part in (("fwd", jp.fwd_index), ("bwd", jp.bwd_index), ("fwd3", jp.fwd3_index), ("bwd3", jp.bwd3_index)):
            o, n = sect[name]
            assert o % 4 == 0 and n == len(part) and (idx[o:o + n] == part).all()
        assert (idx < jp.n_flat).all()


def test_nerfacc_check_tool_says_what_it_needs():
    """tools/check_against_nerfacc.py cannot run here -- nerfacc is not in the image -- and says so: exit code 2 and
    its own message, not a traceback."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_against_nerfacc.py")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "nerfacc is not installed" in r.stderr and "Traceback" not in r.stderr


def test_graph_task_id_semantics_the_in_place_sums_rely_on():
    """ops._QueryAccumulator keys its running gradient sums on the PRIVATE torch._C._current_graph_task_id ("a torch
    upgrade can change its semantics silently; the fallback triggers only when the symbol is absent").  This pins what the code
    relies on, on the CPU, so that a changed torch fails HERE and not as wrong gradients: -1 outside a backward pass; one id for every
    node of one pass; a different id for the next pass; another one again for a reentrant (checkpoint-style) pass nested in it; the
    outer id back afterwards; and the same under torch.autograd.grad."""
    import torch
    from morpheus_amd import ops
    tid = getattr(torch._C, "_current_graph_task_id", None)
    if tid is None:                                  # then the per-query form must be the one in use
        assert ops.ACCUMULATE_IN_PLACE is False
        return
    assert tid() == -1
    seen = []

    class Probe(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, tag):
            ctx.tag = tag
            return x * 1.0

        @staticmethod
        def backward(ctx, g):
            seen.append((ctx.tag, tid()))
            return g, None

    class Nested(torch.autograd.Function):           # a backward that runs ANOTHER backward inside itself (reentrant checkpointing)
        @staticmethod
        def forward(ctx, x):
            ctx.x = x.detach()
            return x * 2.0

        @staticmethod
        def backward(ctx, g):
            seen.append(("outer-before", tid()))
            with torch.enable_grad():
                xi = ctx.x.clone().requires_grad_(True)
                (Probe.apply(xi, "inner") * 2.0).sum().backward()
            seen.append(("outer-after", tid()))
            return g * 2.0

    x = torch.ones(3, requires_grad=True)
    (Probe.apply(x, "a") + Probe.apply(x, "b")).sum().backward()
    (Probe.apply(x, "c")).sum().backward()
    ids = dict(seen)
    assert ids["a"] == ids["b"] >= 0 and ids["c"] >= 0 and ids["c"] != ids["a"], seen
    seen.clear()
    (Nested.apply(Probe.apply(x, "tail")) + Probe.apply(x, "side")).sum().backward()
    ids = {}
    for k, v in seen:
        ids.setdefault(k, v)
    assert ids["outer-before"] == ids["outer-after"] == ids["tail"] == ids["side"] >= 0, seen
    assert ids["inner"] >= 0 and ids["inner"] != ids["tail"], seen
    seen.clear()
    torch.autograd.grad((Probe.apply(x, "g1") + Probe.apply(x, "g2")).sum(), x)
    ids = dict(seen)
    assert ids["g1"] == ids["g2"] >= 0, seen
    assert tid() == -1
