"""The reference's REAL-VIEW optimisation step around `render_rays`, as a caller of the hot path.

`morpheus.py:train_step` (1147-1236, `real_view=True, cano=False, optimize_pose=True`) is the call the reference makes
220 k times per run (10 of every 11 steps, `train_one_epoch` :1399-1424): 2 048 random pixels of ONE random frame,
occupancy-marched ragged samples, `shading='albedo_normal'`, depth + mask supervision, pose optimisation, the in-render
regularisers of :708-792 and three caller-side loss groups.  It stays the reference's Python in a drop-in deployment
(INTEGRATION.md); this module restates it so that `bench.py --workload train_real` and the GPU tests can drive the hot
path with exactly that call pattern on synthetic frames:

    update_occ_grid            morpheus.py:905-913    (every 16th step: model.density on grid cells)
    render_rays                morpheus.py:558-794
    get_pred_from_outputs      morpheus.py:915-928
    get_gt_from_data           morpheus.py:930-945
    get_real_view_render_loss  morpheus.py:946-983
    get_real_view_point_loss   morpheus.py:985-1029
    get_regularization_loss    morpheus.py:1090-1145

Deliberate differences (values identical): boolean-mask indexing (`sdf[depth_mask.bool()]`, :1018) is written as a
masked mean so that the step has no device->host synchronisation of its own; the occupancy refresh asks the field for
the density only (`return_color=False`; the reference evaluates and discards the colour net).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

from . import synth


# ------------------------------------------------------------------------------------------ synthetic real-view frames
def make_frames(frame_ids, H: int, W: int, device, num_frames: int = 200) -> List[Dict[str, torch.Tensor]]:
    """What `DeformDataset.real_view_data` holds per frame (datasets/dataset.py:336-396) for closed-form frames:
    all H*W rays, an RGB image, a depth map and an object mask.  The 'object' is the sphere |x| = 0.45 (the weight
    state of synth.make_state has sdf ~ |x| - 0.4): depth = first intersection, mask = hit, colour = position hash."""
    frames = []
    for fid in frame_ids:
        o, d, t, rid = synth.frame_rays(fid, H, W, num_frames)
        oo, dd = o[0].double(), d[0].double()
        a = (dd * dd).sum(-1)
        b = 2 * (oo * dd).sum(-1)
        c = (oo * oo).sum(-1) - 0.45 ** 2
        disc = b * b - 4 * a * c
        hit = disc > 0
        tz = torch.where(hit, (-b - torch.sqrt(disc.clamp(min=0))) / (2 * a), torch.zeros_like(a))
        depth = tz.float()                                             # along the UN-normalised direction, as the reference
        mask = hit.float()
        rgb = synth.hash_tensor((H * W, 3), 7000 + fid, 0.5, 0.5)
        frames.append({k: v.to(device) for k, v in dict(rays_o=o[0], rays_d=d[0], rays_t=t[0], rays_id=rid[0], image=rgb,
                                                        depth=depth, mask=mask).items()})
    return frames


def sample_real_view_rays(frame: Dict[str, torch.Tensor], ray_num: int, index: Optional[torch.Tensor] = None):
    """datasets/dataset.py:398-433 with `ray_num`: ray_num random pixels of one frame; H = ray_num, W = 1."""
    n = frame["rays_o"].shape[0]
    if index is None:
        index = torch.randint(0, n, (ray_num,), device=frame["rays_o"].device)
    g = lambda k: frame[k][index]
    return dict(rays_o=g("rays_o")[None], rays_d=g("rays_d")[None], rays_t=g("rays_t")[None], rays_id=g("rays_id")[None],
                image=g("image").t().reshape(1, 3, ray_num, 1), depth=g("depth").reshape(1, ray_num, 1),
                mask=g("mask").reshape(1, ray_num, 1), H=ray_num, W=1)


# ------------------------------------------------------------------------------------------ the losses
def get_gt_from_data(data, bg_color, B, H, W):
    """morpheus.py:930-945."""
    gt_rgb, gt_depth, gt_mask = data["image"], data["depth"], data["mask"]
    gt_mask = (gt_mask > 0.5).float()
    gt_rgb = gt_rgb * gt_mask[:, None] + bg_color.reshape(B, H, W, 3).permute(0, 3, 1, 2) * (1 - gt_mask[:, None])
    return gt_rgb, gt_depth, gt_mask


def _valid_depth_mask(gt_depth, gt_mask, rays_o, rays_d):
    """morpheus.py:966-976 / :1001-1011: depth > 0, inside the 1.1 sphere, inside the object mask."""
    xyzs = rays_o + gt_depth.reshape(1, -1, 1) * rays_d
    inside = torch.linalg.norm(xyzs, ord=2, dim=-1, keepdim=True) <= 1.1
    m = (gt_depth > 0) & inside.view(*gt_depth.shape) & (gt_mask > 0.5)
    return m.float(), xyzs


def get_real_view_render_loss(tr, pred_rgb, pred_depth, pred_mask, gt_rgb, gt_depth, gt_mask, rays_o, rays_d):
    """morpheus.py:946-983."""
    loss = 0
    if tr["rgb_weight"] > 0:
        loss = loss + tr["rgb_weight"] * F.mse_loss(pred_rgb, gt_rgb)
    if tr["mask_weight"] > 0:
        loss = loss + tr["mask_weight"] * F.binary_cross_entropy(pred_mask[:, 0].clip(1e-5, 1.0 - 1e-5), gt_mask.float())
    if tr["depth_weight"] > 0:
        depth_mask, _ = _valid_depth_mask(gt_depth, gt_mask, rays_o, rays_d)
        loss = loss + tr["depth_weight"] * F.mse_loss(pred_depth[:, 0] * depth_mask, gt_depth * depth_mask)
    return loss


def get_real_view_point_loss(tr, model, gt_rgb, gt_depth, gt_mask, rays_o, rays_d, rays_t, outputs):
    """morpheus.py:985-1029: SDF / free-space losses from the renderer plus one `model.density` query at the N
    back-projected surface points (x and t of equal length, gradients into both hash tables, the warp and the codes)."""
    loss = 0
    if tr["sdf_weight"] > 0:
        loss = loss + tr["sdf_weight"] * outputs["sdf_loss"]
    if tr["sdf_reg"] > 0:
        loss = loss + tr["sdf_reg"] * torch.mean(outputs["sdf"] ** 2)
    if tr["fs_weight"] > 0:
        loss = loss + tr["fs_weight"] * outputs["fs_loss"]
    if tr["surf_sdf_weight"] > 0:
        depth_mask, xyzs = _valid_depth_mask(gt_depth, gt_mask, rays_o, rays_d)
        results = model.density(xyzs.reshape(-1, 3), t=rays_t.reshape(-1, 1))
        sdf, albedo = results["sdf"], results["albedo"]
        masked_color = albedo.view(*depth_mask.shape, 3).permute(0, 3, 1, 2).contiguous()
        surf_color_loss = tr["surf_color_weight"] * F.mse_loss(masked_color * depth_mask[None, ...], gt_rgb * depth_mask[None, ...])
        # mean of sdf^2 over the valid points == F.mse_loss(sdf[mask], 0) of :1018-1026, without the boolean index
        sq = (sdf.view(*depth_mask.shape) ** 2 * depth_mask).sum() / depth_mask.sum().clamp(min=1.0)
        loss = loss + tr["surf_sdf_weight"] * sq + surf_color_loss
    return loss


def get_regularization_loss(tr, model, outputs, pred_normal, global_step: int, end_iter: int, cano=False):
    """morpheus.py:1090-1145."""
    loss = 0
    if tr["entropy_weight"] > 0:
        alphas = outputs["weights"].clamp(1e-5, 1 - 1e-5)
        ent = (-alphas * torch.log2(alphas) - (1 - alphas) * torch.log2(1 - alphas)).mean()
        loss = loss + tr["entropy_weight"] * min(1, 2 * global_step / end_iter) * ent
    if tr["normal_smooth_2d"] > 0 and pred_normal is not None:
        sm = (pred_normal[:, 1:, :, :] - pred_normal[:, :-1, :, :]).square().mean() + \
             (pred_normal[:, :, 1:, :] - pred_normal[:, :, :-1, :]).square().mean()
        loss = loss + tr["normal_smooth_2d"] * sm
    if tr["ori_weight"] > 0 and "loss_orient" in outputs:
        loss = loss + tr["ori_weight"] * outputs["loss_orient"]
    if tr["normal_smooth_3d"] > 0 and "loss_normal_perturb" in outputs:
        loss = loss + tr["normal_smooth_3d"] * outputs["loss_normal_perturb"]
    if tr["normal_smooth_3d_t"] > 0 and "loss_normal_perturb_t" in outputs:
        loss = loss + tr["normal_smooth_3d_t"] * outputs["loss_normal_perturb_t"]
    if outputs["normal_raw"] is not None and tr["eik_weight"] > 0:
        ge = (torch.linalg.norm(outputs["normal_raw"], ord=2, dim=-1) - 1.0) ** 2
        loss = loss + tr["eik_weight"] * torch.mean(ge)
    if tr["beta_weight"] > 0:
        loss = loss + tr["beta_weight"] * torch.mean(model.sdf2density.get_beta())
    if tr["normal_smoothness"] > 0:
        loss = loss + tr["normal_smoothness"] * outputs["normal_reg"]
    if tr["deform_weight"] > 0:
        loss = loss + tr["deform_weight"] * outputs["deform"].abs().mean()
    for w, k in (("deform_smooth", "loss_deform_perturb"), ("deform_smooth_t", "loss_deform_perturb_t"),
                 ("topo_smooth_t", "loss_topo_perturb_t")):
        if tr[w] > 0 and k in outputs:
            loss = loss + tr[w] * outputs[k]
    if tr["code_reg"] > 0 and not cano and "loss_code" in outputs:
        loss = loss + tr["code_reg"] * outputs["loss_code"]
    return loss


# ------------------------------------------------------------------------------------------ the step
class RealViewTrainStep:
    """`MorpheuS.train_step(real_view=True, cano=False, optimize_pose=True)` (morpheus.py:1147-1236) on synthetic frames.

    renderer: morpheus_amd.render.HotPathRenderer whose `occupancy_grid` is a morpheus_amd.occgrid.OccupancyGrid."""

    def __init__(self, renderer, frames, ray_num: int = 2048, n_epochs: int = 2000, end_iter: int = 220000):
        self.r, self.model, self.cfg = renderer, renderer.model, renderer.config
        self.frames, self.ray_num = frames, ray_num
        self.global_step, self.epoch, self.n_epochs, self.end_iter = 0, 0, n_epochs, end_iter
        self.last_samples = 0

    def update_occ_grid(self, rays_t, cano=False):
        """morpheus.py:905-913."""
        step_size = self.cfg["render"]["step_size"]

        def occ_eval_fn(x):
            return self.model.density(x, rays_t, allow_shape=True, cano=cano, return_color=False)["sigma"] * step_size

        self.r.occupancy_grid.update_every_n_steps(step=self.global_step - 1, occ_eval_fn=occ_eval_fn)

    def __call__(self, frame_index: Optional[int] = None, pixel_index: Optional[torch.Tensor] = None):
        with self.model.operand_scope():      # render_rays and the point loss share one set of prepared weight operands
            return self._step(frame_index, pixel_index)

    def _step(self, frame_index, pixel_index):
        tr = self.cfg["train"]
        self.global_step += 1
        if tr["progressive_level"]:                                   # morpheus.py:808-813
            self.model.max_level = min(1.0, 0.5 + 0.5 * self.epoch / self.n_epochs)
        fi = (self.global_step * 7) % len(self.frames) if frame_index is None else frame_index
        data = sample_real_view_rays(self.frames[fi], self.ray_num, pixel_index)
        rays_o, rays_d, rays_t, rays_id = data["rays_o"], data["rays_d"], data["rays_t"], data["rays_id"]
        B, N = rays_o.shape[:2]
        H, W = data["H"], data["W"]
        rays_depth, rays_mask = data["depth"].view(B, -1, 1), data["mask"].view(B, -1, 1)
        ambient_ratio, shading = 1.0, "albedo_normal"                  # get_shading, real view (:869-871)
        bg_color = torch.rand((B * N, 3), device=rays_o.device)        # get_bg_color, real view (:893-894)
        self.update_occ_grid(rays_t, cano=False)
        outputs = self.r.render_rays(rays_o, rays_d, rays_t, rays_id, H, W, perturb=True, bg_color=bg_color,
                                     ambient_ratio=ambient_ratio, shading=shading, real_view=True, cano=False,
                                     rays_depth=rays_depth, rays_mask=rays_mask, optimize_pose=True)
        self.last_samples = 0 if outputs["sdf"] is None else outputs["sdf"].shape[0]
        pred_depth = outputs["depth"].reshape(B, 1, H, W)
        pred_mask = outputs["weights_sum"].reshape(B, 1, H, W)
        pred_normal = outputs["normal_image"].reshape(B, H, W, 3) if "normal_image" in outputs else None
        pred_rgb = outputs["image"].reshape(B, H, W, 3).permute(0, 3, 1, 2).contiguous()
        gt_rgb, gt_depth, gt_mask = get_gt_from_data(data, bg_color, B, H, W)
        loss = get_real_view_render_loss(tr, pred_rgb, pred_depth, pred_mask, gt_rgb, gt_depth, gt_mask, rays_o, rays_d)
        loss = loss + get_real_view_point_loss(tr, self.model, gt_rgb, gt_depth, gt_mask, rays_o, rays_d, rays_t, outputs)
        loss = loss + get_regularization_loss(tr, self.model, outputs, pred_normal, self.global_step, self.end_iter)
        return loss


def warm_up_occupancy(step: RealViewTrainStep, frame_index: int = 0, n_updates: int = 4):
    """Bring the occupancy grid to a trained-like state with the build's own update rule (occgrid.update_every_n_steps,
    warm-up branch: every cell evaluated) before a timed run."""
    rays_t = step.frames[frame_index]["rays_t"][None, :1]
    saved = step.global_step
    for k in range(n_updates):
        step.global_step = 16 * k + 1
        step.update_occ_grid(rays_t)
    step.global_step = saved
