"""`scene_representation` -- drop-in for the reference's models.model.scene_representation.

Same constructor keywords, method names, parameter-group names and state_dict keys as
/root/reference/models/model.py:31-533 (keys listed in SURVEY.md C.10), so morpheus.py's
optimisation loop, EMA, checkpoints and LR schedule address it unchanged.  Underneath, the deform
field, canonical field and hash grids run on the HIP kernels of libmorpheus_hip.so through
morpheus_amd.ops; there is no PyTorch fallback for those (CPU tensors raise).

Every constructor switch of the reference is accepted.  use_t and use_joint (either value) run on the fused kernels (they
change per-frame constants or zero first-layer columns); use_app=True, encode_topo=True and color_grid=False -- set by no
shipped YAML -- change what the field nets read per point and take the composed field path (`composed_field`: hash grids,
warp nets, finite-difference taps and compositor on the HIP kernels, the two 3 x 64 field nets through torch / rocBLAS).
"""
from __future__ import annotations

import contextlib
import math
import os
import weakref
from typing import List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import ops


def safe_normalize(x: torch.Tensor, eps: float = 1e-20) -> torch.Tensor:
    """utils.py:70-71."""
    return x / torch.sqrt(torch.clamp((x * x).sum(-1, keepdim=True), min=eps))


# ------------------------------------------------------------------------------------ small modules
class PoseArray(nn.Module):
    """Per-frame 6-DoF pose correction, Euler angles -> R (models/pose.py:4-64)."""

    def __init__(self, num_frames: int):
        super().__init__()
        self.num_frames = num_frames
        self.data = nn.Parameter(torch.zeros(num_frames, 6))

    def get_translations(self, ids):
        t = self.data[:, 3:6][ids]
        return t[None] if t.dim() == 1 else t

    def get_rotation_matrices(self, ids):
        r = self.data[:, 0:3][ids]
        if r.dim() == 1:
            r = r[None]
        ca, cb, cg = torch.cos(r[:, 0]), torch.cos(r[:, 1]), torch.cos(r[:, 2])
        sa, sb, sg = torch.sin(r[:, 0]), torch.sin(r[:, 1]), torch.sin(r[:, 2])
        cols = [torch.stack([ca * cb, sa * cb, -sb], -1),
                torch.stack([ca * sb * sg - sa * cg, sa * sb * sg + ca * cg, cb * sg], -1),
                torch.stack([ca * sb * cg + sa * sg, sa * sb * cg - ca * sg, cb * cg], -1)]
        return torch.stack(cols, -1)


class MultiCode(nn.Module):
    """Multi-resolution 1-D code grid, linear in time (models/deform_code.py:5-43)."""

    def __init__(self, sizes, c):
        super().__init__()
        self.volumes = nn.ParameterList([nn.Parameter(torch.randn(1, c, s, 1)) for s in sizes])

    def sample(self, t: torch.Tensor) -> torch.Tensor:
        """[F] or [F,1] times -> [F, L*c], level-major as the reference (deform_code.py:20-38): all levels from ONE HIP launch
        each way (ops.multicode_sample; GPU only, like every op on the path)."""
        if len(self.volumes) != 3:
            raise NotImplementedError("the code kernel is specialised to the reference's three levels (model.py:88-92)")
        return ops.multicode_sample(t, list(self.volumes))

    def get_code(self, level=-1):
        return self.volumes[level].squeeze().permute(1, 0)


class _WNLinear(nn.Module):
    """Linear with the legacy weight_norm parametrisation: W = g * v / ||v|| (decoders.py:51-52)."""

    def __init__(self, din, dout):
        super().__init__()
        lin = nn.Linear(din, dout)
        self.bias = nn.Parameter(lin.bias.detach().clone())
        v = lin.weight.detach().clone()
        self.weight_g = nn.Parameter(v.norm(dim=1, keepdim=True))
        self.weight_v = nn.Parameter(v)

    def effective(self):
        return self.weight_v * (self.weight_g / self.weight_v.norm(dim=1, keepdim=True))


class _PlainLinear(nn.Module):
    def __init__(self, din, dout):
        super().__init__()
        lin = nn.Linear(din, dout)
        self.weight = nn.Parameter(lin.weight.detach().clone())
        self.bias = nn.Parameter(lin.bias.detach().clone())

    def effective(self):
        return self.weight


def wn_effective_batched(layers) -> List[torch.Tensor]:
    """Effective weights W = g * v / ||v|| of many weight-normed layers: ONE HIP launch forward and one backward
    (ops.weight_norm_all) instead of ~200 tiny torch launches per training step for the 15 layers' norm / divide /
    multiply and their autograd graph."""
    return ops.weight_norm_all([l.weight_v for l in layers], [l.weight_g for l in layers])


class MLP(nn.Module):
    """Parameter container with the reference's layout `net.{l}.{weight_g,weight_v|weight,bias}`
    (decoders.py:9-64, incl. the geometric initialisation of :25-43)."""

    def __init__(self, dim_in, dim_out, dim_hidden, num_layers, bias=True, geo_init=False, geo_bias=0.5,
                 weight_norm=True):
        super().__init__()
        self.dim_in, self.dim_out, self.dim_hidden, self.num_layers = dim_in, dim_out, dim_hidden, num_layers
        layers = []
        for l in range(num_layers):
            i = dim_in if l == 0 else dim_hidden
            o = dim_out if l == num_layers - 1 else dim_hidden
            lin = _WNLinear(i, o) if weight_norm else _PlainLinear(i, o)
            if geo_init:
                assert not weight_norm
                with torch.no_grad():
                    if l == num_layers - 1:
                        lin.weight.normal_(math.sqrt(math.pi) / math.sqrt(i), 1e-4)
                        lin.bias.fill_(-geo_bias)
                    elif l == 0:
                        lin.bias.zero_()
                        lin.weight[:, 3:].zero_()
                        lin.weight[:, :3].normal_(0.0, math.sqrt(2) / math.sqrt(o))
                    else:
                        lin.bias.zero_()
                        lin.weight.normal_(0.0, math.sqrt(2) / math.sqrt(o))
            layers.append(lin)
        self.net = nn.ModuleList(layers)

    def weights(self) -> List[torch.Tensor]:
        return [l.effective() for l in self.net]

    def biases(self) -> List[torch.Tensor]:
        return [l.bias for l in self.net]

    def forward(self, x):
        """Plain PyTorch evaluation -- only for nets that are NOT on the hot path (bg_net)."""
        for l, lin in enumerate(self.net):
            x = torch.addmm(lin.bias, x, lin.effective().t())
            if l != self.num_layers - 1:
                x = torch.relu(x)
        return x


class GridEncoder(nn.Module):
    """Multires hash grid with the reference's table sizing (grid.py:104-147) on the HIP kernels."""

    def __init__(self, input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=15,
                 desired_resolution=128):
        super().__init__()
        assert input_dim == 3 and level_dim == 2, "HIP encoder is specialised to D=3, C=2"
        self.num_levels, self.level_dim, self.base_resolution = num_levels, level_dim, base_resolution
        self.per_level_scale = float(np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1)))
        self.output_dim = num_levels * level_dim
        offs, total = [], 0
        for i in range(num_levels):
            res = int(np.ceil(base_resolution * self.per_level_scale ** i))      # float64, grid.py:129
            n = int(np.ceil(min(2 ** log2_hashmap_size, res ** input_dim) / 8) * 8)
            offs.append(total)
            total += n
        offs.append(total)
        self.register_buffer("offsets", torch.tensor(offs, dtype=torch.int32))
        self._offsets_np = np.asarray(offs, dtype=np.int32)
        self._res_np = ops.level_resolutions(num_levels, self.per_level_scale, base_resolution)
        self.n_params = total * level_dim
        self.embeddings = nn.Parameter(torch.empty(total, level_dim).uniform_(-1e-4, 1e-4))

    def forward(self, inputs, bound=1, max_level=None, group=1):
        return ops.grid_encode(inputs, self.embeddings, self._offsets_np, self._res_np, float(bound), max_level, group)


# LaplaceDensity object -> the scene_representation whose step cache its get_beta() may use (registered by the model's training
# forward; kept outside the modules so that copies and pickles of a model carry no reference to another one)
_DENSITY_OWNER: "weakref.WeakValueDictionary" = weakref.WeakValueDictionary()


class LaplaceDensity(nn.Module):
    """VolSDF density, learnable beta (models/density.py:17-31)."""

    def __init__(self, beta=0.1, beta_min=1e-4):
        super().__init__()
        self.beta = nn.Parameter(torch.tensor(float(beta)))
        self.beta_min = beta_min

    def get_beta(self):
        # inside a training forward of the owning scene_representation the value is prepared once (its step cache): the field
        # kernels and the caller's beta regulariser (morpheus.py:1121-1122) then share ONE tensor and one abs / add each way
        owner = _DENSITY_OWNER.get(id(self))
        if owner is not None and owner.sdf2density is self:
            return owner._cached("beta", self._beta)
        return self._beta()

    def _beta(self):
        return self.beta.abs() + self.beta_min

    def forward(self, sdf, beta=None):
        beta = self.get_beta() if beta is None else beta
        return (1 / beta) * (0.5 + 0.5 * sdf.sign() * torch.expm1(-sdf.abs() / beta))


def _freq_encode_torch(x, n_freqs=6, max_level=None):
    """encodings.py:35-57, plain torch -- used only off the hot path (background net)."""
    keep = n_freqs if max_level is None else int(max_level * n_freqs)
    parts = [x]
    for i in range(keep):
        parts += [torch.sin(x * float(2 ** i)), torch.cos(x * float(2 ** i))]
    if keep < n_freqs:
        parts.append(x.new_zeros(*x.shape[:-1], (n_freqs - keep) * 2 * x.shape[-1]))
    return torch.cat(parts, -1)


# ------------------------------------------------------------------------------------ the model
IMPLICIT_OPERANDS = os.environ.get("MORPHEUS_IMPLICIT_OPERANDS", "1") != "0"
_RAW_CAPTURING = getattr(torch._C, "_cuda_isCurrentStreamCapturing", None)
_HAS_GPU = None


def _capturing() -> bool:
    """is the current stream being captured into a graph?  (torch.cuda.is_available() re-reads the environment on every call and
    the step cache asks per model call: the answer to "is there a GPU" is taken once)"""
    global _HAS_GPU
    if _HAS_GPU is None:
        _HAS_GPU = bool(torch.cuda.is_available())
    if not _HAS_GPU or not torch.cuda.is_initialized():
        return False
    return bool(_RAW_CAPTURING()) if _RAW_CAPTURING is not None else torch.cuda.is_current_stream_capturing()


class _StepCache:
    """Prepared operands of ONE training forward (scene_representation._step_cache)."""
    __slots__ = ("sig", "entries", "stale")

    def __init__(self, sig):
        self.sig, self.entries, self.stale = sig, {}, False

    def _spent(self, *_):
        # the graph behind the entries is being run: drop them now -- nothing spent lingers in the model (a HIP-graph capture later
        # in the run crashed on autograd nodes an earlier eager step had left alive: profiles/r05_soak.txt)
        self.stale = True
        self.entries.clear()

    def watch(self, value):
        """the backward pass reaching any graph-bearing member of a cached value ends the step"""
        if isinstance(value, (tuple, list)):
            for v in value:
                self.watch(v)
        elif isinstance(value, ops.MLPOperands):
            if value.notify is not None:
                value.notify.append(self._spent)
        elif torch.is_tensor(value) and value.requires_grad and value.grad_fn is not None:
            value.register_hook(self._mark)

    def _mark(self, grad):
        self._spent()
        return None


class scene_representation(nn.Module):
    def __init__(self, config, bound, max_level=None, num_layers=3, num_layers_t=6, hidden_dim=64, hidden_dim_t=128,
                 hidden_dim_tpo=128, num_layers_bg=2, geo_dim=32, deform_dim=16, hidden_dim_bg=32, amb_dim=2,
                 num_frames=None, use_app=False, use_t=False, color_grid=True, use_joint=False, encode_topo=False,
                 encode_deform=True):
        super().__init__()
        if not encode_deform:
            raise NotImplementedError("encode_deform=False: the reference itself cannot run it (models/model.py:253,422 call the "
                                      "encoder that get_encodings returned as None)")
        if (num_layers, num_layers_t, hidden_dim, hidden_dim_t, hidden_dim_tpo, geo_dim, deform_dim, amb_dim) != \
                (3, 6, 64, 128, 128, 32, 16, 2):
            raise NotImplementedError("kernel geometry is fixed to the reference defaults (model.py:36-44)")
        self.config, self.bound, self.max_level = config, float(bound), max_level
        self.num_layers, self.hidden_dim, self.geo_dim, self.num_frames = num_layers, hidden_dim, geo_dim, num_frames
        self.use_t, self.use_app, self.use_joint = use_t, use_app, use_joint
        self.encode_topo, self.encode_deform = encode_topo, encode_deform
        self.color_grid = bool(color_grid)
        # use_app / encode_topo / color_grid=False change what the field nets read PER POINT (an appearance code per sample behind
        # the colour net's input, the 18-column encoding of the topology coordinates, a frequency-encoded colour input): the fused
        # field kernels are cut for the shipped layout [enc(x) 39 | hash 32 | topo 2] -> [hash_c 32 | geo 32], so these switches --
        # set by no shipped YAML -- take the COMPOSED field path (_sigma_albedo_composed): warp nets, hash grids, finite-difference
        # taps and compositor on the HIP kernels, the two 3 x 64 field nets as rocBLAS products through torch
        self.composed_field = bool(use_app or encode_topo or not color_grid)
        self.in_dim_t = 13 if use_t else 0                                                # model.py:98-103
        self.in_dim_amb = (amb_dim + 2 * 4 * amb_dim) if encode_topo else amb_dim        # :111-115 (multires 4)
        self.in_dim_deform, self.in_dim_xyz = 39, (39 if use_joint else 3)               # :114-119, :162-167
        self.deform_dim = 3 * deform_dim
        self.app_dim = 3 * deform_dim if use_app else 0                                   # :132-136
        self.encoder_t = self.encoder_topo = None

        self.pose_array = PoseArray(num_frames)
        self.deform_code = MultiCode([num_frames // 8, num_frames // 4, num_frames], deform_dim)
        self.app_code = MultiCode([num_frames // 8, num_frames // 4, num_frames], deform_dim) if use_app else None
        self.deform_net = MLP(self.in_dim_t + self.in_dim_deform + self.deform_dim, 3, hidden_dim_t, num_layers_t)
        self.topo_net = MLP(self.in_dim_t + self.in_dim_deform + self.deform_dim, amb_dim, hidden_dim_tpo, num_layers_t)
        self.encoder = GridEncoder()
        self.encoder_c = GridEncoder() if color_grid else None                           # else frequency_torch(x), 39 columns (:151-157)
        self.in_dim = self.encoder.output_dim
        self.in_dim_c = self.encoder.output_dim if color_grid else 39
        self.sdf_net = MLP(self.in_dim + self.in_dim_amb + self.in_dim_xyz, 1 + geo_dim, hidden_dim, num_layers,
                           geo_init=True, geo_bias=0.4, weight_norm=False)
        self.color_net = MLP(self.in_dim_c + geo_dim + self.app_dim, 3, hidden_dim, num_layers)
        if self.config["model"]["bg_radius"] > 0:
            self.in_dim_bg, self.in_dim_bg_t = 39, 13
            self.bg_net = MLP(self.in_dim_bg + self.in_dim_bg_t, 3, hidden_dim_bg, num_layers_bg)
        self.sdf2density = LaplaceDensity(0.1)
        self._opcache = None        # operand cache of the current operand_scope() (None outside a scope)
        self._scope_step = None     # the step cache that scope stands on (None: a scope of its own)
        self._stepcache = None      # prepared operands of the running training forward (_step_cache)
        self._step_params = None    # the parameters whose version counters say "same step"
        # arithmetic of this model's MLP kernels: "b3" / "f32", or None = the process default (ops.mlp_mode()); bound to
        # the operand packs when they are prepared, so two models of one process may run different forms
        self.mlp_mode: Optional[str] = None

    # -- helpers ----------------------------------------------------------------------------
    def _n_bands(self) -> int:
        return 6 if self.max_level is None else int(self.max_level * 6)

    PER_SAMPLE_SLOTS = 1 << 17   # up to this many samples a non-constant time tensor gets one code slot per sample

    def _slots(self, t: torch.Tensor, frame_slots=None) -> Tuple[torch.Tensor, Optional[torch.Tensor], bool]:
        """-> (slot times [F], per-sample slot ids [M] int32 or None when F == 1, slot ids are the identity map), without
        a device->host sync on any path the training step takes:
          * `frame_slots = (t_rows [F], slot [M])` from the renderer (a batch row is one frame);
          * a time tensor that is an expanded scalar (stride 0: `rays_t[:1].expand(M, 1)`, how the renderer and
            `density(allow_shape=True)` / `density(t=float)` hand over a single frame) is constant by construction;
          * otherwise every sample gets its own slot (the per-frame code bias becomes a per-sample one; exact, costs
            M x 128 floats per net) up to PER_SAMPLE_SLOTS samples -- `get_real_view_point_loss` sends N = 2048;
          * beyond that, torch.unique (sort + sync) keeps the bias table small."""
        if frame_slots is not None:
            return frame_slots[0], frame_slots[1], False
        tf = t.reshape(-1)
        if tf.numel() == 1 or (t.dim() >= 1 and t.shape[0] > 1 and t.stride(0) == 0):
            return tf[:1], None, False
        if tf.numel() <= self.PER_SAMPLE_SLOTS:
            return tf, torch.arange(tf.numel(), device=t.device, dtype=torch.int32), True
        tu, inv = torch.unique(tf, return_inverse=True)
        return tu, inv.to(torch.int32), False

    def _warp_params(self, net: MLP, w):
        """-> (kernel parameters, per-frame columns of W0, b0).  The kernels' first layer reads the 39-column frequency
        encoding of x; its per-frame inputs -- the deform code and, with use_t, the 13-column encoding of t that the reference
        puts between them (model.py:427-432) -- are folded into a per-frame bias (_warp_bias0)."""
        b = net.biases()
        return [w[0][:, :39]] + w[1:] + b, w[0][:, 39:], b[0]

    # -- operands that depend only on the parameters: prepared once per scope ------------------------------------------
    @contextlib.contextmanager
    def operand_scope(self):
        """Within a scope (render_rays opens one; a training step may open one around render + point loss) everything that
        depends only on the parameters -- effective (weight-normed) weights, the MFMA operand packs, the per-frame code
        bias, beta -- is prepared ONCE and shared by all warp / field calls: a real-view step evaluates the warp nets 4x and
        the field nets 6x, and the weight gradients of all those calls meet in one raw-gradient token per net group
        (ops._PackOperands) instead of one accumulation per parameter per call.  Parameters must not change inside a scope.
        In a training forward the scope IS the model's step cache (below): what a later call of the same step needs is there."""
        outer = self._opcache
        if outer is None:
            sc = self._step_cache()
            self._opcache = {} if sc is None else sc.entries
            self._scope_step = sc
        try:
            yield self
        finally:
            if outer is None:
                self._opcache = None
                self._scope_step = None

    @contextlib.contextmanager
    def fresh_operand_scope(self):
        """A scope of its own, whatever scope is open around it: chunking.chunked_query re-runs a chunk's forward INSIDE backward
        (reentrant checkpointing: a nested autograd pass with its own graph-task id).  Operands cached by the step's scope carry
        the step's raw-gradient token and in-place gradient sums (ops._QueryAccumulator, keyed on the graph task): a re-run that
        picked them up would reset the outer pass's sums and run the step's _PackOperands node early.  With operands of its own the
        nested pass is self-contained: its sums are collected by its own pack node and reach the parameters' .grad directly."""
        outer, outer_step = self._opcache, self._scope_step
        self._opcache, self._scope_step = {}, None
        try:
            yield self
        finally:
            self._opcache, self._scope_step = outer, outer_step

    # The step cache: the scope nobody had to open (round 6).  The reference's train_step (morpheus.py:1147-1236) calls the model
    # from three places -- render_rays, the surface-point query of get_real_view_point_loss, get_regularization_loss -- and knows
    # nothing of operand scopes: with a scope per call the weight norm, the operand packs and their slices were made twice per
    # step and every parameter received two gradients for autograd to add (reference-glue step: 333 torch launches, 90 of them
    # this).  So a TRAINING forward (module in train mode, gradients enabled, not inside a stream capture) keeps its prepared
    # operands in the model, and the next call finds them -- until one of these ends the step:
    #   * the backward pass reached an operand pack (ops._PackOperands.backward -> `stale`): the graph behind the cached tensors
    #     is spent; gradient accumulation over several forward / backward pairs re-prepares per pair, as it must;
    #   * a parameter changed in place (optimizer.step: `_version`; FlatAdam bumps the counters itself, its kernel writes through
    #     raw pointers), the module was switched with train() / eval(), moved or re-typed (_apply), the arithmetic mode or the
    #     progressive level changed.
    # Forwards under no_grad (evaluation, the occupancy refresh) are never kept across calls: code that swaps weights through
    # `param.data` -- torch_ema's copy_to / restore around the reference's evaluation -- moves no version counter.
    # Two forwards whose backward passes are run separately, later and out of order, share the pack's graph: the second
    # backward then fails with torch's "backward through the graph a second time"; MORPHEUS_IMPLICIT_OPERANDS=0 (or an explicit
    # scope per forward) restores one set of operands per call.
    def _step_cache(self):
        """-> the live _StepCache of this training forward (created if the last one has ended), or None when nothing may be kept"""
        if not (IMPLICIT_OPERANDS and self.training and torch.is_grad_enabled()):
            return None
        if _capturing():
            return None           # operands prepared inside a capture belong to the graph; operands from outside would freeze in it
        ps = self._step_params
        if ps is None:
            ps = self._step_params = list(self.parameters())
            _DENSITY_OWNER[id(self.sdf2density)] = self
        sig = (tuple([p._version for p in ps]), self.mlp_mode, ops.mlp_mode() if self.mlp_mode is None else None, self.max_level)
        sc = self._stepcache
        if sc is None or sc.stale or sc.sig != sig or len(sc.entries) > 64:
            sc = self._stepcache = _StepCache(sig)
        return sc

    def end_step(self):
        """Drop what the running training forward has prepared (a forward that will never be run backward; before a capture)."""
        self._stepcache = None

    def train(self, mode: bool = True):
        self._stepcache = self._step_params = None
        return super().train(mode)

    def _apply(self, fn, *a, **kw):
        self._stepcache = self._step_params = None
        return super()._apply(fn, *a, **kw)

    def _cached(self, key, build):
        c, sc = self._opcache, self._scope_step
        if c is None:
            sc = self._step_cache()
            if sc is None:
                return build()
            c = sc.entries
        key = (key, torch.is_grad_enabled())
        if key not in c:
            c[key] = v = build()
            if sc is not None and key[1]:
                sc.watch(v)
        return c[key]

    def _warp_operands(self):
        def build():
            w_all = wn_effective_batched(list(self.deform_net.net) + list(self.topo_net.net))
            pd, wcode_d, b0_d = self._warp_params(self.deform_net, w_all[:6])
            pt, wcode_t, b0_t = self._warp_params(self.topo_net, w_all[6:])
            return ops.prepare_warp_operands(pd, pt, mode=self.mlp_mode), (wcode_d, b0_d, wcode_t, b0_t)
        return self._cached("warp", build)

    def _warp_bias0(self, tu, code_w):
        """per-slot first-layer bias W0[:,39:] code(t) + b0 of both nets; cached per scope for a single-frame time (the
        cache entry keeps `tu` alive, so its address cannot be recycled while the entry exists)."""
        wcode_d, b0_d, wcode_t, b0_t = code_w

        def build():
            code = self.deform_code.sample(tu[:, None])                       # [F,48], F = distinct frames / slots
            if self.use_t:                                                    # [t_enc(13), code(48)], model.py:427-432
                code = torch.cat([_freq_encode_torch(tu[:, None], 6, self.max_level), code], -1)
            return torch.addmm(b0_d, code, wcode_d.t()), torch.addmm(b0_t, code, wcode_t.t()), tu
        if tu.numel() != 1:
            return build()[:2]
        return self._cached(("bias0", tu.data_ptr(), tu._version), build)[:2]

    def _field_operands(self):
        def build():
            ws = self.sdf_net.weights()
            if not self.use_joint:
                # sdf input [x(3), hash(32), topo(2)] (model.py:285-289): the kernel reads [enc(x)(39), hash, topo] and the
                # encoding's first three entries are x -- the 36 sin / cos columns get zero weights
                w0 = ws[0]
                ws = [torch.cat([w0[:, :3], w0.new_zeros(w0.shape[0], 36), w0[:, 3:]], 1)] + ws[1:]
            params = ws + wn_effective_batched(list(self.color_net.net)) + self.sdf_net.biases() + \
                self.color_net.biases()
            beta = self.sdf2density.get_beta()
            tables = (self.encoder.embeddings, self.encoder_c.embeddings)
            return ops.prepare_field_operands(params, mode=self.mlp_mode, beta=beta, tables=tables), beta
        return self._cached("field", build)

    # -- public API (names/signatures of the reference) ----------------------------------------
    def get_deform_code(self, t, app=False):
        if app:
            return self.app_code.sample(t)
        return self.deform_code.sample(t)

    def get_RT(self, frame_ids):
        ids = frame_ids.squeeze()
        return self.pose_array.get_rotation_matrices(ids), self.pose_array.get_translations(ids)

    def pose_optimisation(self, rays_o, rays_d, frame_ids, rows=None):
        """models/model.py:335-346.  rows = (B, n): the N = B*n rays are B batch rows of n rays, ONE frame per row (how the
        reference's dataset builds every batch): R and t are evaluated for the B frames and broadcast over
        their rays -- same arithmetic per ray, but the gradient reaches pose_array through a B-row gather instead of an
        N-row one (torch's index backward serialises on 2048 identical indices: 0.57 ms per gather)."""
        if rows is not None:
            B, n = rows
            ids = frame_ids.reshape(B, n)[:, 0]
            if rays_o.requires_grad or rays_d.requires_grad:       # rays with a gradient of their own: the operator chain
                R, t = self.pose_array.get_rotation_matrices(ids), self.pose_array.get_translations(ids)   # [B,3,3], [B,3]
                o = (rays_o.view(B, n, 3) + t[:, None]).view(-1, 3)
                d = (rays_d.view(B, n, 1, 3) * R[:, None]).sum(-1).view(-1, 3)
                return o, d
            # Euler angles -> R, the translation and both products in one launch (three with the backward) instead of ~160
            return ops.pose_apply(rays_o.reshape(-1, 3), rays_d.reshape(-1, 3), self.pose_array.data, ids, n)
        ids = frame_ids.squeeze()
        R, t = self.pose_array.get_rotation_matrices(ids), self.pose_array.get_translations(ids)
        return rays_o + t, (rays_d[..., None, :] * R).sum(-1)

    def warp(self, x, t, frame_slots=None):
        """-> deform [M,3], topo [M,2] (its 18-column encoding with encode_topo), app_code ([M,48] with use_app, else None)
        (model.py:412-437).  `frame_slots`: see `_slots`."""
        tu, slot, identity = self._slots(t, frame_slots)
        opnd, code_w = self._warp_operands()
        bias0_d, bias0_t = self._warp_bias0(tu, code_w)                   # per-frame first-layer bias
        deform, topo = ops.warp_mlp(x, slot, bias0_d, bias0_t, self._n_bands(), opnd, slots_are_identity=identity)
        app_code = None
        if self.use_app:                                                  # model.py:418-420: one code row per sample
            rows = self.app_code.sample(tu[:, None])                      # [F,48], F = distinct frames / slots
            app_code = rows.expand(x.shape[0], -1) if slot is None else rows[slot.long()]
        if self.encode_topo:                                              # :434-435
            topo = _freq_encode_torch(topo, 4, self.max_level)
        return deform, topo, app_code

    def get_topo(self, x, t, frame_slots=None):
        return self.warp(x, t, frame_slots)[1]

    def get_sigma_albedo(self, x, topo=None, app_code=None, return_color=True, _group=1):
        """hash grid(s) -> sdf_net -> Laplace density (-> color_net)   (model.py:273-307), one autograd node
        (ops._FieldQuery).  _group: every `_group` consecutive points are neighbours (6 = finite-difference taps)."""
        if self.composed_field:
            return self._sigma_albedo_composed(x, topo, app_code, return_color, _group)
        opnd, beta = self._field_operands()
        sdf, sigma, albedo = ops.field_query(x, topo, beta, self.encoder.embeddings,
                                             self.encoder_c.embeddings if return_color else None, self.encoder._offsets_np,
                                             self.encoder._res_np, self.bound, self.max_level, _group, self._n_bands(), opnd)
        return sdf, sigma, (albedo if return_color else None)

    def _sigma_albedo_composed(self, x, topo, app_code, return_color, group):
        """get_sigma_albedo (model.py:273-307) for the switches the fused field kernels are not cut for: the same operator
        chain -- hash grid(s) on the HIP kernels, sdf_net / color_net as torch products, the reference's Laplace density."""
        enc = self.encoder(x, bound=self.bound, max_level=self.max_level, group=group)
        if topo is None:
            topo = x.new_zeros(x.shape[0], self.in_dim_amb)
        xyz = _freq_encode_torch(x, 6, self.max_level) if self.use_joint else x
        h = self.sdf_net(torch.cat([xyz, enc, topo], -1))
        sdf = h[..., 0]
        sigma = self.sdf2density(sdf)
        if not return_color:
            return sdf, sigma, None
        if self.color_grid:
            enc_c = self.encoder_c(x, bound=self.bound, max_level=self.max_level, group=group)
        else:
            enc_c = _freq_encode_torch(x, 6, self.max_level)
        feat = [enc_c, h[..., 1:]]
        if self.use_app:
            feat.append(x.new_zeros(x.shape[0], self.app_dim) if app_code is None else app_code)
        return sdf, sigma, torch.sigmoid(self.color_net(torch.cat(feat, -1)))

    def get_params_all(self, lr):
        groups = [
            {"name": "encoder_sdf", "params": self.encoder.parameters(), "lr": lr},
            {"name": "encoder_color", "params": self.encoder_c.parameters() if self.color_grid else [], "lr": lr},
            {"name": "decoder_sdf", "params": self.sdf_net.parameters(), "lr": lr},
            {"name": "decoder_topo", "params": self.topo_net.parameters(), "lr": lr},
            {"name": "decoder_color", "params": self.color_net.parameters(), "lr": lr},
            {"name": "density", "params": self.sdf2density.parameters(), "lr": lr / 2.0},
            {"name": "decoder_deform", "params": self.deform_net.parameters(), "lr": lr},
            {"name": "code_deform", "params": self.deform_code.parameters(), "lr": lr},
            {"name": "pose", "params": self.pose_array.parameters(), "lr": lr / 10.0},
        ]
        if self.config["model"]["bg_radius"] > 0:
            groups.append({"name": "decoder_bg", "params": self.bg_net.parameters(), "lr": lr})
        if self.use_app:
            groups.append({"name": "code_app", "params": self.app_code.parameters(), "lr": lr})
        return groups

    def _fd_normals(self, x, epsilon=2e-3, topo=None):
        """-> (normal, raw).  6 clamped taps of the SDF (model.py:367-385), evaluated as ONE batch of 6M points, point-major
        (a sample's six taps adjacent: shared hash-corner cache lines, same backward brick): one launch builds the taps
        (and replicates topo), one hash-grid pass, one sdf-net pass, one launch turns the six values into the raw and the
        normalised normal (model.py:387-398); the same four launches run backward."""
        M = x.shape[0]
        taps, topo6 = ops.fd_taps(x, topo, epsilon, self.bound)
        sdf = self.get_sigma_albedo(taps, topo=topo6, return_color=False, _group=6)[0].view(M, 6)
        return ops.fd_normal(sdf, epsilon)

    def finite_difference_normal(self, x, epsilon=2e-3, topo=None):
        return self._fd_normals(x, epsilon, topo)[1]

    def normal(self, x, t=None, cano=False, topo=None, frame_slots=None):
        if t is not None and not cano:
            deform, topo, _ = self.warp(x, t, frame_slots)
            x = x + deform
        return self._fd_normals(x, topo=topo)

    def background(self, d, t):
        h = torch.cat([_freq_encode_torch(d, 6, None), _freq_encode_torch(t, 6, self.max_level)], -1)
        return torch.sigmoid(self.bg_net(h))

    def density(self, x, t=None, cano=False, allow_shape=False, return_color=True):
        topo = app_code = None
        if not (cano or t is None):
            # a single time for all points travels as an expanded scalar: `_slots` sees one frame, no per-sample work
            if isinstance(t, float):
                t = torch.full((1, 1), t, device=x.device).expand(x.shape[0], 1)
            if x.shape[0] != t.shape[0]:
                if not allow_shape:
                    raise Exception("Shape inconsistent!!!")
                t = t.reshape(-1)[:1].view(1, 1).expand(x.shape[0], 1)
            deform, topo, app_code = self.warp(x, t)
            x = x + deform
        sdf, sigma, albedo = self.get_sigma_albedo(x, topo=topo, app_code=app_code, return_color=return_color)
        return {"sdf": sdf, "sigma": sigma, "albedo": albedo}

    def forward(self, x, t, light_dir=None, ratio=1, shading="albedo", cano=False, return_color=True, *,
                frame_slots=None):
        deform = topo = app_code = None
        xc = x
        if not cano:
            deform, topo, app_code = self.warp(x, t, frame_slots)
            xc = x + deform
        sdf, sigma, albedo = self.get_sigma_albedo(xc, topo, app_code, return_color)
        if shading == "albedo":
            return sdf, sigma, albedo, None, deform, None
        normal, raw = self.normal(x, topo=topo)        # un-warped x, warped point's topo (model.py:515-521)
        if ratio == 1 and shading not in ("textureless", "normal"):
            # real-view steps shade with ambient_ratio = 1.0 (morpheus.py:869-871): lam = 1 + 0 * (n.l)+ is exactly 1 and
            # carries an exactly-zero gradient to the normal, so the lambertian factor is dropped
            return sdf, sigma, albedo, normal, deform, raw
        lam = ratio + (1 - ratio) * (normal * light_dir).sum(-1).clamp(min=0)
        if shading == "textureless":
            color = lam.unsqueeze(-1).repeat(1, 3)
        elif shading == "normal":
            color = (normal + 1) / 2
        else:
            color = albedo * lam.unsqueeze(-1)
        return sdf, sigma, color, normal, deform, raw
