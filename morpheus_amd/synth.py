"""Closed-form synthetic inputs and weights for the render_rays hot path.

Everything here is a pure function of integer indices, so the GPU box, the CPU
oracle and the golden-fixture generator (which imports the reference in the
build container) all regenerate bit-identical fp32 tensors without shipping
multi-megabyte fixtures.  Shapes follow SURVEY.md section 8(d): camera on a
sphere looking at the origin, AABB +-1.01, S equal bins per ray with one
stratified jitter per ray, frame time t = id / num_frames.

Generator: u(i, c) = ((i * 2654435761 + c * 40503) mod 2^24) / 2^24 - 0.5,
exactly representable in fp32.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np
import torch

_A = np.uint64(2654435761)
_B = np.uint64(40503)
_MASK = np.uint64((1 << 24) - 1)


def unit_hash(n: int, stream: int) -> np.ndarray:
    """n values in [-0.5, 0.5), fp32-exact, from the integer hash above."""
    i = np.arange(n, dtype=np.uint64)
    k = (i * _A + np.uint64(stream) * _B) & _MASK
    # decorrelate consecutive indices a little more (still integer-only)
    k = (k ^ (k >> np.uint64(11)) * np.uint64(0x9E3779B1)) & _MASK
    return (k.astype(np.float64) / float(1 << 24) - 0.5).astype(np.float32)


def hash_tensor(shape, stream: int, scale: float = 1.0, shift: float = 0.0) -> torch.Tensor:
    n = int(np.prod(shape)) if len(shape) else 1
    v = unit_hash(n, stream).astype(np.float64) * (2.0 * scale) + shift
    return torch.from_numpy(v.astype(np.float32).reshape(shape))


# ----------------------------------------------------------------------------
# model dimensions at configs/snoopy.yaml (SURVEY.md section 8 preamble)
# ----------------------------------------------------------------------------
SNOOPY_DIMS = dict(num_frames=200, deform_dim=16, amb_dim=2, hidden_t=128, layers_t=6,
                   hidden=64, layers=3, geo_dim=32, hidden_bg=32, layers_bg=2,
                   grid_levels=16, grid_C=2, grid_base=16, grid_log2T=15, grid_desired=128)


def grid_offsets(levels=16, base=16, log2_T=15, desired=128, D=3) -> Tuple[np.ndarray, float]:
    """Row offsets of the multires table and per_level_scale.

    Sizing follows the reference's host logic (external/encoders/gridencoder/
    grid.py:104-136): float64 ceil(base * s^l), min(2^log2_T, res^D), rounded
    up to a multiple of 8.
    """
    s = np.exp2(np.log2(desired / base) / (levels - 1))
    offs = [0]
    for l in range(levels):
        res = int(np.ceil(base * s ** l))
        n = min(2 ** log2_T, res ** D)
        n = int(np.ceil(n / 8) * 8)
        offs.append(offs[-1] + n)
    return np.asarray(offs, dtype=np.int32), float(s)


def make_state(kind: str = "b", num_frames: int = 200, bg: bool = True) -> Dict[str, torch.Tensor]:
    """A full scene_representation state_dict (keys of SURVEY.md C.10).

    kind 'a': init-like (hash features +-1e-4, sphere-ish SDF, tiny deform);
    kind 'b': non-degenerate (hash features +-0.1, |deform| ~ 0.05) so that
    hash-grid bugs cannot hide behind a first layer that ignores them.
    """
    assert kind in ("a", "b")
    sd: Dict[str, torch.Tensor] = {}
    st = [1000 if kind == "a" else 5000]

    def nxt():
        st[0] += 1
        return st[0]

    F = num_frames
    sd["pose_array.data"] = hash_tensor((F, 6), nxt(), 0.0 if kind == "a" else 0.02)
    for k, size in enumerate((F // 8, F // 4, F)):
        sd[f"deform_code.volumes.{k}"] = hash_tensor((1, 16, size, 1), nxt(), 1.0)

    def wn_mlp(prefix, din, dout, hid, nl, last_scale=1.0):
        for l in range(nl):
            i = din if l == 0 else hid
            o = dout if l == nl - 1 else hid
            bound = 1.0 / math.sqrt(i)
            v = hash_tensor((o, i), nxt(), bound)
            g = v.norm(dim=1, keepdim=True) * (1.0 + hash_tensor((o, 1), nxt(), 0.2))
            if l == nl - 1:
                g = g * last_scale
            sd[f"{prefix}.net.{l}.bias"] = hash_tensor((o,), nxt(), bound * (last_scale if l == nl - 1 else 1.0))
            sd[f"{prefix}.net.{l}.weight_g"] = g.contiguous()
            sd[f"{prefix}.net.{l}.weight_v"] = v

    wn_mlp("deform_net", 39 + 48, 3, 128, 6, last_scale=0.02 if kind == "a" else 0.6)
    wn_mlp("topo_net", 39 + 48, 2, 128, 6, last_scale=0.05 if kind == "a" else 0.6)

    offs, _ = grid_offsets()
    rows = int(offs[-1])
    emb_scale = 1e-4 if kind == "a" else 0.1
    sd["encoder.embeddings"] = hash_tensor((rows, 2), nxt(), emb_scale)
    sd["encoder.offsets"] = torch.from_numpy(offs.copy())
    sd["encoder_c.embeddings"] = hash_tensor((rows, 2), nxt(), emb_scale)
    sd["encoder_c.offsets"] = torch.from_numpy(offs.copy())

    # sdf_net 73 -> 64 -> 64 -> 33, geometric-init-like structure (decoders.py:25-43):
    # layer 0 sees xyz strongly, last layer has positive mean so sdf ~ |x| - 0.4
    w0 = hash_tensor((64, 73), nxt(), math.sqrt(2) / math.sqrt(64) * 1.7)
    if kind == "a":
        w0[:, 3:] = 0.0
    else:
        w0[:, 3:39] *= 0.15
        w0[:, 39:] *= 0.6
    sd["sdf_net.net.0.weight"] = w0
    sd["sdf_net.net.0.bias"] = hash_tensor((64,), nxt(), 0.0 if kind == "a" else 0.05)
    sd["sdf_net.net.1.weight"] = hash_tensor((64, 64), nxt(), math.sqrt(2) / math.sqrt(64) * 1.7)
    sd["sdf_net.net.1.bias"] = hash_tensor((64,), nxt(), 0.0 if kind == "a" else 0.05)
    w2 = hash_tensor((33, 64), nxt(), 0.05 if kind == "b" else 1e-4 * 1.7)
    w2[0] = w2[0] * (0.2 if kind == "b" else 1.0) + math.sqrt(math.pi) / math.sqrt(64)
    sd["sdf_net.net.2.weight"] = w2
    b2 = hash_tensor((33,), nxt(), 0.0 if kind == "a" else 0.05)
    b2[0] = -0.4
    sd["sdf_net.net.2.bias"] = b2

    wn_mlp("color_net", 64, 3, 64, 3, last_scale=1.0 if kind == "a" else 2.0)
    if bg:
        wn_mlp("bg_net", 39 + 13, 3, 32, 2)
    sd["sdf2density.beta"] = torch.tensor(0.1, dtype=torch.float32)
    return sd


def variant_state(kind: str = "b", num_frames: int = 200, use_t: bool = False, use_joint: bool = True, use_app: bool = False,
                  encode_topo: bool = False, color_grid: bool = True) -> Dict[str, torch.Tensor]:
    """make_state() for the model switches that change first-layer shapes (models/model.py:36-53, :195-227):
    use_t (13 time-encoding columns between the position encoding and the deform code of deform_net / topo_net),
    use_joint=False (raw x instead of its 39-column encoding in front of sdf_net), encode_topo (the 2 topology coordinates enter
    sdf_net as their 18-column frequency encoding), use_app (a second MultiCode whose 48 columns follow the colour net's input) and
    color_grid=False (the colour net reads the 39-column frequency encoding of x instead of a second hash table).  Only the
    affected first layers (and the appearance code) get new closed-form tensors."""
    sd = make_state(kind, num_frames)
    st = [7300 if kind == "a" else 7600]

    def nxt():
        st[0] += 1
        return st[0]

    din = 39 + (13 if use_t else 0) + 48
    if din != 39 + 48:
        for prefix in ("deform_net", "topo_net"):
            bound = 1.0 / math.sqrt(din)
            v = hash_tensor((128, din), nxt(), bound)
            sd[f"{prefix}.net.0.weight_v"] = v
            sd[f"{prefix}.net.0.weight_g"] = (v.norm(dim=1, keepdim=True) * (1.0 + hash_tensor((128, 1), nxt(), 0.2))).contiguous()
    n_xyz, n_amb = (39 if use_joint else 3), (18 if encode_topo else 2)
    if (n_xyz, n_amb) != (39, 2):
        st[0] += 10 * (n_amb == 18)          # the no-joint stream of earlier fixtures stays what it was
        w0 = hash_tensor((64, n_xyz + 32 + n_amb), nxt(), math.sqrt(2) / math.sqrt(64) * 1.7)
        if kind == "a":
            w0[:, 3:] = 0.0
        else:
            if n_xyz == 39:
                w0[:, 3:39] *= 0.15
            w0[:, n_xyz:] *= 0.6
        sd["sdf_net.net.0.weight"] = w0
    n_c = (32 if color_grid else 39) + 32 + (48 if use_app else 0)
    if n_c != 64:
        st[0] = (7400 if kind == "a" else 7700) + n_c
        bound = 1.0 / math.sqrt(n_c)
        v = hash_tensor((64, n_c), nxt(), bound)
        sd["color_net.net.0.weight_v"] = v
        sd["color_net.net.0.weight_g"] = (v.norm(dim=1, keepdim=True) * (1.0 + hash_tensor((64, 1), nxt(), 0.2))).contiguous()
    if use_app:
        F = num_frames
        for k, size in enumerate((F // 8, F // 4, F)):
            sd[f"app_code.volumes.{k}"] = hash_tensor((1, 16, size, 1), nxt(), 1.0)
    if not color_grid:
        del sd["encoder_c.embeddings"], sd["encoder_c.offsets"]
    return sd


# ----------------------------------------------------------------------------
# rays and samples (SURVEY.md section 8(d) "Synthetic inputs")
# ----------------------------------------------------------------------------
def look_at_pose(theta_deg: float, phi_deg: float, radius: float = 1.5) -> np.ndarray:
    """OpenGL camera-to-world looking at the origin from a sphere of `radius`."""
    th, ph = math.radians(theta_deg), math.radians(phi_deg)
    c = np.array([radius * math.sin(th) * math.sin(ph), radius * math.cos(th),
                  radius * math.sin(th) * math.cos(ph)], dtype=np.float64)
    fwd = -c / np.linalg.norm(c)              # viewing direction
    up = np.array([0.0, 1.0, 0.0])
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right)
    true_up = np.cross(right, fwd)
    m = np.eye(4, dtype=np.float64)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = right, true_up, -fwd, c
    return m.astype(np.float32)


def camera_rays(H: int, W: int, c2w: np.ndarray, focal_mult: float = 1.2) -> Tuple[torch.Tensor, torch.Tensor]:
    """Pinhole rays, OpenGL convention, directions NOT normalised.

    dirs = [(i+.5-cx)/fx, -(j+.5-cy)/fy, -1]; rays_d = R . dirs; rays_o = c2w[:3,3]
    (datasets/utils.py:28-65, datasets/dataset.py:363-366).
    """
    fx = fy = np.float32(focal_mult * W)
    cx, cy = np.float32(0.5 * W), np.float32(0.5 * H)
    i, j = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing="xy")
    dirs = np.stack([(i + np.float32(0.5) - cx) / fx, -(j + np.float32(0.5) - cy) / fy,
                     -np.ones_like(i)], -1).astype(np.float32)
    R = c2w[:3, :3].astype(np.float32)
    rays_d = (dirs[..., None, :] * R).sum(-1).astype(np.float32).reshape(-1, 3)
    rays_o = np.broadcast_to(c2w[:3, 3].astype(np.float32), rays_d.shape).copy()
    return torch.from_numpy(rays_o), torch.from_numpy(rays_d)


def frame_rays(frame_id: int, H: int, W: int, num_frames: int = 200):
    """Rays of one synthetic frame: (rays_o, rays_d, rays_t, rays_id) with leading batch dim 1."""
    theta = 60.0 + 30.0 * math.sin(0.37 * frame_id)
    phi = (frame_id * 360.0 / num_frames) % 360.0 - 180.0
    o, d = camera_rays(H, W, look_at_pose(theta, phi))
    n = o.shape[0]
    t = torch.full((n, 1), frame_id / num_frames, dtype=torch.float32)
    rid = torch.full((n, 1), frame_id, dtype=torch.int64)
    return o[None], d[None], t[None], rid[None]


def ray_jitter(n_rays: int, stream: int = 77) -> torch.Tensor:
    """One stratified jitter u in [0,1) per ray."""
    return torch.from_numpy(unit_hash(n_rays, stream) + np.float32(0.5))


def targets(n_rays: int, stream: int = 91) -> Tuple[torch.Tensor, torch.Tensor]:
    """Fixed regression targets for loss = MSE(image) + MSE(depth)."""
    img = hash_tensor((n_rays, 3), stream, 0.5, 0.5)
    dep = hash_tensor((n_rays,), stream + 1, 0.4, 1.4)
    return img, dep
