/*
 * morpheus_hip.h -- C ABI of libmorpheus_hip.so (gfx950 / MI355X).
 *
 * Drop-in native boundary for the render_rays hot path of HengyiWang/MorpheuS.
 * Every entry point takes raw device pointers + sizes + an explicit hipStream_t
 * (passed as void*), returns an int status (0 = ok), never throws, never
 * allocates or frees, keeps no thread-local state and no global state that a
 * call's result depends on (forward and backward arrive on different host
 * threads; the only host statics are idempotent per-DEVICE caches -- a device's
 * CU count, "dynamic-LDS limit of kernel k raised on device d" -- indexed by
 * the current device, csrc/common.h), and launches only on the
 * stream it is given (the reference launched on the legacy default stream --
 * gridencoder.cu:386 -- a defect this ABI deliberately does not reproduce).
 * All buffers are caller-owned, contiguous, fp32 unless stated.
 *
 * Reference interfaces replaced (file:line under the reference tree):
 *   mh_grid_encode_fwd / _bwd   external/encoders/gridencoder/src/gridencoder.h:12-13,
 *                               bindings.cpp:5-10, python callers grid.py:61,91
 *   mh_composite_fwd / _bwd     nerfacc.render_weight_from_density + accumulate_along_rays,
 *                               call sites morpheus.py:675-685 (third-party, un-vendored)
 *   mh_sample_uniform           nerfacc OccGridEstimator.sampling call site morpheus.py:628-638
 *                               (benchmark sampler of SURVEY 8d; marcher is next-tier)
 *   mh_generate_rays            datasets/utils.py:28-65 + datasets/dataset.py:363-366
 *   mh_rays_sample_uniform      the two above + the pixel draw of datasets/dataset.py:412-423, one launch
 *   mh_warp_* / mh_field_*      models/model.py:412-437 (warp), :273-307 (get_sigma_albedo),
 *                               models/decoders.py:59-64, models/encodings.py:35-57,
 *                               models/density.py:22-31 -- plain PyTorch in the reference
 *   mh_mlp_wgrad                the weight-gradient GEMMs autograd ran for those MLPs
 * The warp / field entries exist in two arithmetic forms of the SAME interface -- fp32 in, fp32 out, same parked tiles:
 * native fp32 MFMA (no suffix) and exact three-way bf16 splits (_b3: fp32-faithful, what the Python side calls by default --
 * morpheus_amd/ops.py, MORPHEUS_MLP); mh_b3_slice cuts the weight operands of the latter.
 */
#ifndef MORPHEUS_HIP_H
#define MORPHEUS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MH_OK 0
#define MH_ERR_ARG 1     /* bad size / null pointer / unsupported configuration */
#define MH_ERR_LAUNCH 2  /* hipGetLastError() != hipSuccess after the launch */

#define MH_ABI_VERSION 8   /* 8: mh_smooth_points_*, mh_bg_blend_* (the last operator chains inside render_rays), live-row counts of mh_mlp_wgrad(_b3), mh_warp_wgrad_b3 / mh_warp_regen_dpre4 and skip_dpre4 of mh_warp_bwd_data_b3; 7: the fp16 x 2 (_h2) entry points removed (not fp32-faithful; round-5 verdict item 8); 6: mh_grid_stage_min_points, mh_grid_encode_fwd_binned; 5: accumulate flags of mh_grid_encode_bwd_binned (d/dx) and mh_field_bwd_fused (raw; d(beta) is raw[24 928]); 4: mh_graph_*, mh_masked_mean_*, mh_ortho_perturb_*, mh_pose_apply_*, mh_render_loss_*; 3: round-3 prune (measured-loser entry points removed), n_valid in mh_sdf_losses_* */
#define MH_MAX_LEVELS 32
#define MH_TILE 32       /* sample points per wavefront tile in the MLP kernels */

int mh_abi_version(void);
const char *mh_status_string(int status);

/* ---- multiresolution hash grid (D=3, C=2, linear, align_corners=false, gridtype=hash) --------
 * x:        [M,3] world coordinates; u = (x + bound) / (2*bound) is formed in-kernel
 * emb:      [rows,2] table;  offsets_host: [L+1] HOST ints;  res_host: [L] HOST ints
 *           (res_l = (uint32)ceil(exp2f(l*S)*H) evaluated in float32 by the caller)
 * out:      [M, L*2] point-major, level-major/channel-minor inside a point (grid.py:64)
 * n_levels: levels >= n_levels are written as zero (grid.py:42,53)
 * Out-of-range points give zeros and zero gradients (gridencoder.cu:105-130, :279-284).
 * group:    performance hint, >= 1: every `group` consecutive points lie close together (6 = the finite-difference
 *           taps of one sample, models/model.py:367-385) and may share gathered corners; results do not depend on it. */
int mh_grid_encode_fwd(const float *x, const float *emb, const int32_t *offsets_host,
                       const int32_t *res_host, float *out, int64_t M, int32_t L, int32_t n_levels,
                       float bound, int32_t group, void *stream);
/* grad: [M, L*2]; grad_emb: [rows,2] ACCUMULATED into (caller zeroes it, as grid.py:84 does);
 * grad_x: NULL or [M,3], receives d/dx (the 1/(2*bound) chain factor included).  The slope uses
 * the kernel's dy_dx definition, which ignores the border clamp (gridencoder.cu:205-247). */
int mh_grid_encode_bwd(const float *grad, const float *x, const float *emb,
                       const int32_t *offsets_host, const int32_t *res_host, float *grad_emb,
                       float *grad_x, int64_t M, int32_t L, int32_t n_levels, float bound, void *stream);

/* Brick-binned backward (the fast path; same results as mh_grid_encode_bwd up to summation order).
 * mh_grid_bin_points counting-sorts the points into 16^3 spatial bricks of the box:
 *   workspace: int32[mh_grid_bin_workspace_ints()] scratch; perm: int32[M] point ids grouped by brick
 *   (out-of-box points last); brick_start: int32[mh_grid_bin_index_ints()] = offsets into perm followed
 *   by the work-item table (bricks holding more than 1024 points -- 2048 in calls of >= mh_grid_stage_min_points() points -- are
 *   split over several workgroups) and a
 *   few scratch words used by the backward (max |grad| for its fixed-point on-chip accumulation).
 * One binning serves every encoder evaluated at the same x (sdf and colour tables).
 * mh_grid_encode_bwd_binned: one workgroup per brick accumulates all levels in LDS, then flushes the
 * touched vertices with one global atomic each (grad_emb is ADDED to: zero it, or hand over the running sum of several
 * queries of one table). L must be 16. grad_x (optional): accumulate_dx == 0 -> fully written; != 0 -> the points' d/dx is
 * added to what grad_x holds (the field nets' d/dx of the same points: saves the caller one pass and one launch).
 * gmax_bits: NULL, or a DEVICE word holding max |grad| as raw float bits (the producer of `grad` may compute it on
 * the fly, see mh_field_bwd_fused); NULL costs one extra pass over `grad`. */
int64_t mh_grid_bin_workspace_ints(void);
int32_t mh_grid_bin_bricks(void);
int32_t mh_grid_bin_index_ints(void);
int mh_grid_bin_points(const float *x, int64_t M, float bound, int32_t *workspace, int32_t *perm,
                       int32_t *brick_start, void *stream);
int mh_grid_encode_bwd_binned(const float *grad, const float *x, const float *emb,
                              const int32_t *offsets_host, const int32_t *res_host, const int32_t *perm,
                              const int32_t *brick_start, float *grad_emb, float *grad_x, int32_t accumulate_dx,
                              int64_t M, int32_t L, int32_t n_levels, float bound, const uint32_t *gmax_bits, void *stream);
/* Brick-binned forward: mh_grid_encode_fwd's features (bit-identical), for points binned by mh_grid_bin_points BEFORE the
 * forward: a workgroup stages its brick's table rows in LDS once and its <= 1024 points read their corners there instead of
 * gathering 8 rows per (point, level) through the texture-address path (0.46 -> 0.2x ms per table at 2.1 M points).  The
 * same perm / brick_start then serve mh_grid_encode_bwd_binned.  L must be 16.  Points outside the box get zero rows.
 * (The host side takes it for every call of >= mh_grid_stage_min_points() points, the grouped finite-difference-tap calls included.) */
int mh_grid_encode_fwd_binned(const float *x, const float *emb, const int32_t *offsets_host, const int32_t *res_host,
                              const int32_t *perm, const int32_t *brick_start, float *out, int64_t M, int32_t L,
                              int32_t n_levels, float bound, void *stream);
/* Tuning knob (process-wide) of the d/dx forms of mh_grid_encode_bwd_binned: a call of at least this many points stages each
 * brick's table rows in LDS once per work item (one workgroup per CU) instead of gathering eight rows per (point, level) from
 * global memory; smaller calls (the ~0.1 ms calls of a training step) are faster gathering.  Results are bit-identical either
 * way.  set < 0: query only.  Returns the value in force (default 2^20). */
int64_t mh_grid_stage_min_points(int64_t set);

/* ---- packed transmittance compositor -------------------------------------------------------
 * Samples of ray r are the contiguous range [ray_start[r], ray_start[r]+ray_cnt[r]) of the packed
 * arrays, ordered by t.  w_i = exp(-sum_{j<i} sigma_j dt_j) * (1 - exp(-sigma_i dt_i)).
 * weights [M]; opacity [N]; depth [N] = sum w * (ts+te)/2; color [N,3] = sum w * rgb. */
int mh_composite_fwd(const float *sigma, const float *t_starts, const float *t_ends, const float *rgb,
                     const int32_t *ray_start, const int32_t *ray_cnt, float *weights, float *opacity,
                     float *depth, float *color, int32_t N, void *stream);
/* g_weights may be NULL.  Outputs d_sigma [M], d_rgb [M,3]. */
int mh_composite_bwd(const float *sigma, const float *t_starts, const float *t_ends, const float *rgb,
                     const int32_t *ray_start, const int32_t *ray_cnt, const float *weights,
                     const float *g_weights, const float *g_opacity, const float *g_depth,
                     const float *g_color, float *d_sigma, float *d_rgb, int32_t N, void *stream);

/* ---- ray generation + uniform stratified sampler ------------------------------------------- */
/* K = fx,fy,cx,cy; c2w: [4,4] row-major HOST floats; rays_o/rays_d: [H*W,3] (OpenGL, un-normalised) */
int mh_generate_rays(float fx, float fy, float cx, float cy, const float *c2w_host, int32_t H, int32_t W,
                     float *rays_o, float *rays_d, void *stream);
/* AABB slab clip to [-bound,bound]^3, S bins of (t_far-t_near)/(S+1), comb shifted by jitter*dt.
 * Outputs (length N*S, ray-major): ray_idx int32, t_starts, t_ends, xyz [N*S,3] = o + d*(ts+te)/2;
 * and ray_start/ray_cnt [N] int32. */
int mh_sample_uniform(const float *rays_o, const float *rays_d, const float *jitter, int32_t N, int32_t S,
                      float bound, int32_t *ray_idx, float *t_starts, float *t_ends, float *xyz,
                      int32_t *ray_start, int32_t *ray_cnt, void *stream);

/* The two above in one launch (BASELINE.json north_star "fused ray-generate + stratified sampler"): ray r looks
 * through pixel pix[r] (int32 [N] on the device; NULL = pixel r, N <= H*W) -- the per-iteration pixel draw of
 * datasets/dataset.py:412-423.  Writes rays_o/rays_d [N,3] and every mh_sample_uniform output, bit-identical to
 * mh_generate_rays + gather + mh_sample_uniform. */
int mh_rays_sample_uniform(float fx, float fy, float cx, float cy, const float *c2w_host, int32_t H, int32_t W,
                           const int32_t *pix, const float *jitter, int32_t N, int32_t S, float bound,
                           float *rays_o, float *rays_d, int32_t *ray_idx, float *t_starts, float *t_ends,
                           float *xyz, int32_t *ray_start, int32_t *ray_cnt, void *stream);

/* Occupancy-grid marcher (nerfacc OccGridEstimator.sampling call shape, morpheus.py:628-638): fixed `step`,
 * one jitter per ray (NULL = none), binary grid [R,R,R] uint8 over the AABB [-bound,bound]^3.  Interval k of a ray:
 * ts = t_near + u*step + k*step, te = min(ts+step, t_far), kept iff the cell of its midpoint is occupied.
 * One wavefront per ray (64 steps per iteration, ballot/popcount compaction), single pass:
 * mh_march_slots writes ray_cnt [N] and the kept intervals of ray r to slot_ts/slot_te [r*cap .. r*cap+cnt), cap >=
 * mh_march_cap(step, bound) (steps on the AABB diagonal for unit-or-longer directions); *overflow (device int, zeroed by
 * the caller) is set if some ray needed more than cap slots -- the caller then marches again with a longer slot row.
 * After the exclusive scan of ray_cnt into ray_start, mh_march_pack copies the slot rows to the packed ray_idx /
 * t_starts / t_ends. */
int32_t mh_march_cap(float step, float bound);
int mh_march_slots(const float *rays_o, const float *rays_d, const float *jitter, int32_t N, float step, float bound,
                   int32_t R, const uint8_t *binary, int32_t cap, int32_t *ray_cnt, float *slot_ts, float *slot_te,
                   int32_t *overflow, void *stream);
int mh_march_pack(const int32_t *ray_start, const int32_t *ray_cnt, const float *slot_ts, const float *slot_te, int32_t N,
                  int32_t cap, int32_t *ray_idx, float *t_starts, float *t_ends, void *stream);

/* ---- glue of the field queries as single launches (csrc/normal.hip) ----------------------------------------------------
 * Finite-difference normals (models/model.py:367-398): mh_fd_taps writes the 6 clamped taps of every sample, point-major
 * (+x,-x,+y,-y,+z,-z; taps [6M,3]) and replicates topo [M,topo_dim] to topo6 [6M,topo_dim] (topo NULL: skipped);
 * mh_fd_taps_bwd sums the taps' gradients back (clamp passes inside [-bound, bound]); g_x / g_topo NULL: not computed.
 * mh_fd_normal_fwd: sdf6 [M,6] -> raw [M,3] = 0.5 (s+ - s-) / eps and normal = nan_to_num(raw / sqrt(max(|raw|^2, 1e-20)));
 * mh_fd_normal_bwd: (g_normal, g_raw; either may be NULL) -> g_sdf6 [M,6].
 * Sample assembly (morpheus.py:644-647): xyz[m] = rays_o[r] + rays_d[r] * (t_starts[m] + t_ends[m]) / 2, r = ray_idx[m];
 * backward = per-ray segment sums over the packed samples (ray_start / ray_cnt), one wavefront per ray. */
int mh_fd_taps(const float *x, const float *topo, int32_t topo_dim, float eps, float bound, int64_t M, float *taps,
               float *topo6, void *stream);
int mh_fd_taps_bwd(const float *x, const float *g_taps, const float *g_topo6, int32_t topo_dim, float eps, float bound,
                   int64_t M, float *g_x, float *g_topo, void *stream);
int mh_fd_normal_fwd(const float *sdf6, float eps, int64_t M, float *normal, float *raw, void *stream);
int mh_fd_normal_bwd(const float *sdf6, const float *g_normal, const float *g_raw, float eps, int64_t M, float *g_sdf6,
                     void *stream);
int mh_sample_positions(const float *rays_o, const float *rays_d, const int32_t *ray_idx, const float *t_starts,
                        const float *t_ends, int64_t M, float *xyz, void *stream);
/* MultiCode.sample (models/deform_code.py:20-38): three [C, size_l] tables (the reference's volumes [1,C,size,1]) sampled
 * linearly in time, grid_sample(align_corners=True) coordinate rule, t clamped to [0,1]; out [F, 3*C] level-major.
 * Backward accumulates into g0/g1/g2 (same shapes as the tables, ZEROED by the caller) with atomics. */
int mh_multicode_fwd(const float *t, const float *v0, const float *v1, const float *v2, int32_t s0, int32_t s1, int32_t s2,
                     int32_t C, int32_t F, float *out, void *stream);
int mh_multicode_bwd(const float *t, const float *g_out, float *g0, float *g1, float *g2, int32_t s0, int32_t s1, int32_t s2,
                     int32_t C, int32_t F, void *stream);
/* get_sdf_loss (utils.py:91-113) on packed samples: per-ray depth [N] / mask [N] (NULL: none) are read through ray_idx, the
 * sample depth is (t_starts + t_ends)/2.  sums [3] (device) = free-space sum, near-surface sum, count of samples whose target
 * depth is non-zero; the caller divides the two sums by the count.  Backward: g_fs / g_sl are device scalars (NULL = 0).
 * n_valid: NULL, or a DEVICE int: only the first *n_valid of the M packed entries are samples (fixed-capacity sampling pads
 * the packed arrays so that a captured HIP graph sees constant shapes); the padding adds nothing and gets zero gradient. */
int mh_sdf_losses_fwd(const float *pred_sdf, const float *t_starts, const float *t_ends, const int32_t *ray_idx,
                      const float *rays_depth, const float *rays_mask, float trunc, int64_t M, const int32_t *n_valid,
                      float *sums, void *stream);
int mh_sdf_losses_bwd(const float *pred_sdf, const float *t_starts, const float *t_ends, const int32_t *ray_idx,
                      const float *rays_depth, const float *rays_mask, float trunc, int64_t M, const int32_t *n_valid,
                      const float *sums, const float *g_fs, const float *g_sl, float *g_pred, void *stream);
int mh_sample_positions_bwd(const float *g_xyz, const float *t_starts, const float *t_ends, const int32_t *ray_start,
                            const int32_t *ray_cnt, int32_t N, float *g_o, float *g_d, void *stream);

/* ---- fused tiny-MLP evaluators on fp32 MFMA (v_mfma_f32_32x32x2_f32) -------------------------
 * Weight operands are PRE-PACKED by the host into the MFMA A-fragment order (see
 * morpheus_amd/packing.py): for layer l, tile mt, k-quad q: float4 per lane.  `wpack` is the
 * concatenation of all layers of the net(s); layer geometry is fixed by the kernel.
 * Activation scratch ("acts") is written by the forward when acts != NULL and consumed by the
 * backward: per 32-point tile, per layer, feature-major [F][32] fp32 (the weight-gradient GEMM's operand),
 * followed by the hidden layers' ReLU sign masks (one bit per lane and output row) that backward-data reads
 * instead of the activations themselves.
 *
 * mh_warp_fwd: deform = deform_net([freq(x), code]), topo = topo_net(same)
 *   x [M,3]; slot [M] int32 -> row of bias0 (per-frame first-layer bias  W0[:,39:87].code + b0,
 *   computed by the caller; one row per distinct frame time); bias0_{d,t}: [n_slots,128];
 *   n_bands: frequency bands kept (progressive max_level), 0..6;
 *   out_deform [M,3]; out_topo [M,2];  acts: NULL or scratch of mh_warp_acts_floats(M) floats. */
int64_t mh_mlp_tiles(int64_t M);        /* 32-point tiles the kernels touch: 4 * ceil(M/128) */
int64_t mh_warp_acts_floats(int64_t M);
int64_t mh_warp_dpre_floats(int64_t M);
int64_t mh_warp_wpack_floats(void);   /* per net, forward pack */
int64_t mh_warp_wpackT_floats(void);  /* per net, transposed pack */
int mh_warp_fwd(const float *x, const int32_t *slot, const float *bias0_d, const float *bias0_t,
                const float *wpack_d, const float *wpack_t, const float *bias_d, const float *bias_t,
                int32_t n_bands, float *out_deform, float *out_topo, float *acts, int64_t M, void *stream);
/* ---- the same networks with exact fp32 products on the bf16 matrix pipe (csrc/mlp_b3.hip) ------
 * Every fp32 operand is cut into three bf16 slices (hi + mid + lo == x exactly) and a product is the six significant
 * cross terms through v_mfma_f32_32x32x16_bf16 with fp32 accumulation: fp32-grade results (what is dropped is <= 3 * 2^-24
 * of a product), 6/16 of the fp32-MFMA cycles.  Values, parked tiles and the C-ABI stay fp32; only the weight operand has
 * its own pack.
 *
 * mh_b3_slice: fp32 fragments in the 32x32x16 order (packing.py: fwd3_index; per layer [out tile][k16 step][lane][8]) ->
 *   per layer three bf16 planes [hi | mid | lo][out tile][k16 step][lane][8 bf16].  src_off / n in floats (n % 8 == 0),
 *   dst_off in 16-byte units; host arrays of n_layers entries (<= 16).
 * mh_warp_fwd_b3: mh_warp_fwd with w3_{d,t} = one net's sliced pack (mh_warp_w3_bytes() bytes: layer 0 padded to whole
 *   512 x 16-byte DMA rounds).  Same outputs, same parked tiles (mh_warp_bwd_data / mh_mlp_wgrad consume them).
 * mh_warp_bwd_data_b3: mh_warp_bwd_data with the TRANSPOSED sliced packs (mh_warp_w3T_bytes() bytes per net: T5, T4..T1,
 *   T0; packing.py: bwd3_index).  Same dPre tiles, same g_x.  skip_dpre4 = 1: the 128 dPre4 rows of both nets (16 KB of a tile's
 *   172) are NOT written -- mh_warp_wgrad_b3(regen_dpre4 = 1) makes them again from the incoming gradient, the ReLU sign words and
 *   the T5 slices, bit for bit (dPre4 = relu'(H5) W5^T dPre5 and dPre5 IS the incoming gradient); allowed when
 *   mh_warp_regen_dpre4(M) says so (the large-batch weight-gradient path).
 * mh_warp_wgrad_b3: mh_mlp_wgrad_b3 for the 12 layers of the two warp nets, their tile geometry on this side of the ABI; g_deform /
 *   g_topo / w3T_*: what mh_warp_bwd_data_b3 was given (read only with regen_dpre4 = 1; the gradients may be NULL = zero);
 *   workspace: mh_warp_wgrad_workspace_floats(M) floats; dw_raw / db_raw as mh_mlp_wgrad. */
/* mh_field_fwd_b3: mh_field_fwd with the sliced pack of the six field layers (mh_field_w3_bytes() bytes, resident in LDS:
 *   packing.py field_joint_packer().b3_layers).  Same outputs, same parked tiles (mh_field_bwd_fused consumes them). */
int64_t mh_field_w3_bytes(void);
int mh_field_fwd_b3(const float *xc, const float *feat_s, const float *feat_c, const float *topo, const void *w3,
                    const float *bias, const float *beta, int32_t n_bands, int32_t with_color, float *sdf, float *sigma,
                    float *albedo, float *acts, int64_t M, void *stream);
int mh_b3_slice(const float *src, void *dst, int32_t n_layers, const int32_t *src_off_host, const int32_t *n_host,
                const int32_t *dst_off_f4_host, void *stream);
int64_t mh_warp_w3_bytes(void);
int64_t mh_warp_w3T_bytes(void);
int mh_warp_bwd_data_b3(const float *x, const float *g_deform, const float *g_topo, const void *w3T_d, const void *w3T_t,
                        int32_t n_bands, const float *acts, float *dpre, float *g_x, int64_t M, int32_t skip_dpre4, void *stream);
int32_t mh_warp_regen_dpre4(int64_t M);
int64_t mh_warp_wgrad_workspace_floats(int64_t M);
int mh_warp_wgrad_b3(const float *acts, const float *dpre, const float *g_deform, const float *g_topo, const void *w3T_d,
                     const void *w3T_t, int32_t regen_dpre4, float *workspace, float *dw_raw, float *db_raw, int64_t M, void *stream);
int mh_warp_fwd_b3(const float *x, const int32_t *slot, const float *bias0_d, const float *bias0_t, const void *w3_d,
                   const void *w3_t, const float *bias_d, const float *bias_t, int32_t n_bands, float *out_deform,
                   float *out_topo, float *acts, int64_t M, void *stream);
/* backward-data: consumes g_deform [M,3], g_topo [M,2] (either may be NULL = zero), acts from the
 * forward and the TRANSPOSED packs; writes g_x [M,3] (d/dx through the frequency encoding; pass NULL when the
 * sample positions carry no gradient and the first-layer transposed GEMM is skipped) and
 * dpre scratch (same geometry as acts) for mh_mlp_wgrad. */
int mh_warp_bwd_data(const float *x, const float *g_deform, const float *g_topo, const float *wpackT_d,
                     const float *wpackT_t, int32_t n_bands, const float *acts, float *dpre, float *g_x,
                     int64_t M, void *stream);

/* mh_field_fwd: canonical field  sdf_net([freq(xc), hash, topo]) -> sdf, sigma (Laplace), geo;
 *               color_net([hash_c, geo]) -> sigmoid -> albedo.
 *   xc [M,3]; feat_s / feat_c [M,32] (hash features); topo NULL or [M,2]; beta: DEVICE scalar = |beta_p|+1e-4
 *   (a pointer, so the host never synchronises to read the learnable beta);
 *   with_color = 0 skips color_net (FD-normal taps, occupancy queries). */
int64_t mh_field_acts_floats(int64_t M);
int64_t mh_field_wpack_floats(void);
int64_t mh_field_wpackT_floats(void);
int mh_field_fwd(const float *xc, const float *feat_s, const float *feat_c, const float *topo,
                 const float *wpack, const float *bias, const float *beta, int32_t n_bands, int32_t with_color,
                 float *sdf, float *sigma, float *albedo, float *acts, int64_t M, void *stream);
/* weight gradients: for `n_layers` (<= 16) layers described by HOST arrays (offsets in floats into the
 * acts / dpre tiles, feature counts padded to multiples of 32):
 *   dW_l[out][in] = sum_pts dpre_l[out][pt] * act_l[in][pt],  db_l[out] = sum_pts dpre_l[out][pt]
 * One MFMA launch per layer writes per-chunk partials into `workspace`
 * (mh_mlp_wgrad_workspace_floats(...) floats), one reduction launch sums them into
 *   dw_raw [sum_l out_l*in_l]  followed contiguously by  db_raw [sum_l out_l]   (db_raw == dw_raw + sum_l out_l*in_l)
 * in tile-row order (the caller maps rows back to the natural layout, morpheus_amd/packing.py).
 * in_live_host / out_live_host (HOST arrays, or NULL = every row): how many rows of layer l's input tile / dPre tile carry
 * values (the warp nets: 40 of the first layer's 64 input rows, 3 | 2 of the last layer's 32 dPre rows).  Rows behind them are
 * never read -- the producing kernels do not write them -- and the dW rows / columns of those pad features are unspecified
 * (the caller's row map does not gather them).  128-row layers take out_live = 128. */
int64_t mh_mlp_wgrad_workspace_floats(int32_t n_layers, const int32_t *in_feats_host,
                                      const int32_t *out_feats_host, int64_t n_tiles);
int mh_mlp_wgrad(const float *acts, const float *dpre, int64_t acts_tile_floats, int64_t dpre_tile_floats,
                 int32_t n_layers, const int32_t *act_off_host, const int32_t *dpre_off_host,
                 const int32_t *in_feats_host, const int32_t *out_feats_host, const int32_t *in_live_host,
                 const int32_t *out_live_host, float *workspace, float *dw_raw, float *db_raw, int64_t n_tiles, void *stream);
/* the same GEMM with exact fp32 products from three bf16 slices on the bf16 matrix pipe (see mh_warp_fwd_b3): both fp32
 * operands are sliced on the fly, same arguments, same outputs. */
int mh_mlp_wgrad_b3(const float *acts, const float *dpre, int64_t acts_tile_floats, int64_t dpre_tile_floats,
                    int32_t n_layers, const int32_t *act_off_host, const int32_t *dpre_off_host,
                    const int32_t *in_feats_host, const int32_t *out_feats_host, const int32_t *in_live_host,
                    const int32_t *out_live_host, float *workspace, float *dw_raw, float *db_raw, int64_t n_tiles, void *stream);

/* ---- weight-norm parametrisation of every weight-normed layer, one launch each way ---------- */
/* Replaces nn.utils.weight_norm on the Linear layers of deform_net / topo_net / color_net (models/decoders.py:51-52),
 * recomputed per forward: W_l[r,:] = v_l[r,:] * g_l[r] / ||v_l[r,:]||, and its backward (dv_l, dg_l from dW_l).
 * The *_host arguments are HOST arrays (n_layers <= 32) of DEVICE pointers / sizes; v_l, W_l, dW_l, dv_l are
 * [rows_l, cols_l] row-major, g_l and dg_l are [rows_l].  dw_host[l] may be NULL (no gradient): dv_l = dg_l = 0. */
int mh_weight_norm_fwd(int32_t n_layers, const float *const *v_host, const float *const *g_host, float *const *w_host,
                       const int32_t *rows_host, const int32_t *cols_host, void *stream);
int mh_weight_norm_bwd(int32_t n_layers, const float *const *v_host, const float *const *g_host,
                       const float *const *dw_host, float *const *dv_host, float *const *dg_host,
                       const int32_t *rows_host, const int32_t *cols_host, void *stream);

/* Backward of the field nets: backward-data AND weight gradients in one pass per net; the weight-gradient accumulators
 * stay in registers, so the pre-activation gradients never reach HBM (no `dpre` buffer).
 *   g_sdf, g_sigma [M], g_albedo [M,3] (any may be NULL) -> g_xc [M,3] or NULL (freq path only; the hash path's d/dx comes
 *   from mh_grid_encode_bwd), g_feat_s, g_feat_c [M,32], g_topo [M,2].  sdf / albedo are the forward's outputs (Laplace and
 *   sigmoid derivatives are formed from them).
 *   gmax_bits: NULL, or 2 DEVICE words zeroed by the caller that receive max |g_feat_s| and max |g_feat_c| as raw float
 *   bits (atomicMax) -- what mh_grid_encode_bwd_binned needs for its fixed-point accumulation.
 *   dgeo_scratch [mh_field_dgeo_floats(M)] (d(geo) handed from the colour launch to the sdf launch; may be NULL when
 *   with_color == 0), workspace [mh_field_bwd_fused_workspace_floats(M)] (per-wave partial sums), and
 *   raw [24 928 + 1] = the weight gradient in mh_mlp_wgrad's tile-row format for the layer list s0, s1, s2, c0, c1, c2
 *   (dW tiles | db tiles; the colour part is zero when with_color == 0) followed by dL/dbeta (one float, summed in a fixed
 *   order).  accumulate == 0: raw is written; != 0: this call's gradients are ADDED to raw -- the field queries of one
 *   training step (render, finite-difference taps, regularisers) sum their weight gradients in place, no caller-side adds. */
int64_t mh_field_bwd_fused_workspace_floats(int64_t M);
int64_t mh_field_dgeo_floats(int64_t M);
int mh_field_bwd_fused(const float *xc, const float *sdf, const float *albedo, const float *g_sdf, const float *g_sigma,
                       const float *g_albedo, const float *wpackT, const float *beta, int32_t n_bands, int32_t with_color,
                       const float *acts, float *dgeo_scratch, float *workspace, float *raw, int32_t accumulate, float *g_xc,
                       float *g_feat_s, float *g_feat_c, float *g_topo, uint32_t *gmax_bits, int64_t M, void *stream);
/* The same pass with exact fp32 products from three bf16 slices on the bf16 matrix pipe (see mh_warp_fwd_b3): w3T = the sliced
 * TRANSPOSED pack of the six field layers (packing.py field_joint_packer().b3T_layers: TC2 | TC1 | TC0 | TS2 | TS1 | TS0,
 * mh_field_w3T_bytes() bytes, cut by mh_b3_slice).  Same arguments otherwise, same outputs up to fp32 summation order. */
int64_t mh_field_w3T_bytes(void);
int mh_field_bwd_fused_b3(const float *xc, const float *sdf, const float *albedo, const float *g_sdf, const float *g_sigma,
                          const float *g_albedo, const void *w3T, const float *beta, int32_t n_bands, int32_t with_color,
                          const float *acts, float *dgeo_scratch, float *workspace, float *raw, int32_t accumulate, float *g_xc,
                          float *g_feat_s, float *g_feat_c, float *g_topo, uint32_t *gmax_bits, int64_t M, void *stream);
/* ---- optimiser step over the flat parameter bucket (the step after the path) ---- */
/* Replaces torch.optim.Adam(model.get_params_all(lr), betas=(0.9,0.99), eps=1e-15).step() of morpheus.py:154-155,
 * :1401-1424 (no weight decay, no amsgrad).  params/grads/exp_avg/exp_avg_sq: [n] fp32 device buffers, 16-byte aligned.
 * The bucket is cut into contiguous segments, normally ONE PER PARAMETER TENSOR (plus alignment pads): seg_end_host[s] =
 * exclusive end offset (last == n), seg_lr_host[s] = its group's learning rate, seg_step_host[s] = the 1-based step count
 * of that parameter for its bias corrections, or 0 = the parameter has no gradient this time and is left untouched
 * (moments, value and count), which is what torch.optim.Adam does for grad None (HOST arrays, n_segs <= 160). */
int mh_adam_step(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t n, int32_t n_segs,
                 const int64_t *seg_end_host, const float *seg_lr_host, const int64_t *seg_step_host, float beta1,
                 float beta2, float eps, void *stream);
/* The same step with the per-segment bookkeeping ON THE DEVICE, for data-parallel runs: seg_flag_dev [n_segs] > 0 = the
 * segment's parameter has a gradient this step ON SOME RANK (the all-reduced has-gradient flags of the gradient bucket, which
 * only the device knows without a synchronisation); seg_step_dev [n_segs] int64 in/out = the per-segment step counts,
 * incremented where the flag is set; seg_scratch_dev [2 * n_segs] floats of workspace.  A segment whose flag is 0 is left
 * untouched (value, moments, count).  Two launches (one thread per segment, then the update). */
int mh_adam_step_dev(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t n, int32_t n_segs,
                     const int64_t *seg_end_host, const float *seg_lr_host, const float *seg_flag_dev, int64_t *seg_step_dev,
                     float *seg_scratch_dev, float beta1, float beta2, float eps, void *stream);

/* ---- caller-side glue as single launches (csrc/losses.hip) -------------------------------------------------------------
 * Per-sample means of the training losses.  kind: 0 a, 1 a^2, 2 |a|, 3 binary entropy in bits of clamp(a, 1e-5, 1 - 1e-5)
 * (morpheus.py:1093-1096), 4 eikonal (|row| - 1)^2 of [M,3] rows (morpheus.py:1117-1119), 5 |a - b|, 6 (a - b)^2
 * (morpheus.py:764-777, :556).  a, b: [M, C] contiguous fp32 (b only for kinds 5, 6).  n_valid: device int32 or NULL -- only
 * the first n_valid rows count (fixed-capacity sampling); w_row [M] or NULL -- per-row weights.  out[0] = sum / den, out[1] =
 * den, with den = per_row * max(rows, 1) without weights (the reference's `.mean()`; per_row = C, 1 for kind 4) and
 * max(per_row * sum(w_row), 1) with them (morpheus.py:556).  ws: mh_masked_mean_workspace_floats() floats.  Two launches
 * forward (block partials, then one block adding them in a fixed order: deterministic, nothing to zero), one backward:
 * g_a = g * f'(a) * w_row / den inside the valid rows and 0 behind them, g_b = -g_a (either may be NULL); g: device scalar. */
int64_t mh_masked_mean_workspace_floats(void);
int mh_masked_mean_fwd(int32_t kind, const float *a, const float *b, const float *w_row, int64_t M, int32_t C,
                       const int32_t *n_valid, float *ws, float *out, void *stream);
int mh_masked_mean_bwd(int32_t kind, const float *a, const float *b, const float *w_row, int64_t M, int32_t C,
                       const int32_t *n_valid, const float *out, const float *g, float *g_a, float *g_b, void *stream);
/* out = x + scale * (cos(phi) u + sin(phi) v), u = normalize((n^_y, -n^_x, 0)), v = n^ x u, n^ = normalize(n): the random
 * direction orthogonal to the normal of morpheus.py:518-528 (get_ortho_normal_dir) applied to the points (:549, :766).
 * x, n, out [M,3], phi [M] (the caller's uniform draw times 2 pi).  Backward: g_x = g_out (no launch), g_n from *_bwd. */
int mh_ortho_perturb_fwd(const float *x, const float *n, const float *phi, float scale, int64_t M, float *out, void *stream);
int mh_ortho_perturb_bwd(const float *n, const float *phi, const float *g_out, float scale, int64_t M, float *g_n, void *stream);
/* The points of get_normal_smoothness_loss (morpheus.py:530-547): pts[k, n, :] = (depth[n] + off[k]) * rays_d[n, :] + rays_o[n, :],
 * keep[k, n] = |pts| < 1.1 as 0 / 1 (the reference drops the others with a boolean index, :546-547; here they leave the mean
 * through this weight).  depth [N], off [K], rays_o / rays_d [N,3], pts [K*N,3], keep [K*N].  Every product and sum rounded on
 * its own, in the operator chain's order: the same bits.  Backward: g_depth [N], g_o, g_d [N,3] (any may be NULL), the K points
 * of a ray added in index order.  One launch each way (12 / 8 as torch operators). */
int mh_smooth_points_fwd(const float *depth, const float *off, const float *rays_o, const float *rays_d, int64_t N, int32_t K,
                         float *pts, float *keep, void *stream);
int mh_smooth_points_bwd(const float *g_pts, const float *depth, const float *off, const float *rays_d, int64_t N, int32_t K,
                         float *g_depth, float *g_o, float *g_d, void *stream);
/* image = color + (1 - opacity) * bg with a per-ray background (morpheus.py:686-694; bg from get_bg_color, :887-903): color, bg,
 * image [N,3], opacity [N].  Backward: g_color = g_image (nothing to launch), g_opacity [N] = -sum_c g_image bg, g_bg [N,3] =
 * (1 - opacity) g_image (NULL: not wanted).  One launch each way (3 / 4 as torch operators), the chain's rounding. */
int mh_bg_blend_fwd(const float *color, const float *opacity, const float *bg, int64_t N, float *image, void *stream);
int mh_bg_blend_bwd(const float *g_image, const float *opacity, const float *bg, int64_t N, float *g_opacity, float *g_bg, void *stream);

/* Per-frame pose correction of a ray batch made of B rows of n_per_row rays, ONE frame per row (models/pose.py:4-64 PoseArray,
 * models/model.py:335-346 pose_optimisation): o' = o + t_f, d' = R(a, b, g)_f d with pose [n_frames, 6] = (a, b, g, t) per frame
 * and frame_of_row [B] (int64).  Backward: g_pose [n_frames, 6] = the full gradient (zeroed inside; rows of one frame add up
 * in row order); g_o / g_d may be NULL; ws: mh_pose_bwd_workspace_floats(B, n_per_row) floats.  One launch forward, two back. */
int64_t mh_pose_bwd_workspace_floats(int64_t B, int64_t n_per_row);
int mh_pose_apply_fwd(const float *rays_o, const float *rays_d, const float *pose, const int64_t *frame_of_row, int64_t B,
                      int64_t n_per_row, float *o_out, float *d_out, void *stream);
int mh_pose_apply_bwd(const float *rays_d, const float *pose, const int64_t *frame_of_row, int64_t B, int64_t n_per_row,
                      int64_t n_frames, const float *g_o, const float *g_d, float *ws, float *g_pose, void *stream);

/* The real-view render loss of a ray batch (morpheus.py:930-945 get_gt_from_data + :946-983 get_real_view_render_loss): per ray
 * m = mask > 0.5, gt_rgb = image m + bg (1 - m), valid = depth > 0 and |o + depth d| <= 1.1 and m;  rgb = mean (pred_rgb -
 * gt_rgb)^2, mask = BCE(clip(opacity, 1e-5, 1 - 1e-5), m), depth = mean (valid (pred_depth - depth))^2.  pred_rgb, bg, rays_o,
 * rays_d [N,3]; image [3,N] (channel-major, the dataset's layout); the others [N].  out[0] = w_rgb rgb + w_mask mask + w_depth
 * depth, out[1..3] = the terms; gt_rgb [3,N] and valid [N] are returned for the surface-point loss (:1001-1026).  One launch
 * (one workgroup walks the rays: deterministic) and one backward (g: device scalar; any of g_rgb / g_depth / g_opacity NULL). */
int mh_render_loss_fwd(const float *pred_rgb, const float *pred_depth, const float *opacity, const float *image,
                       const float *depth, const float *mask, const float *bg, const float *rays_o, const float *rays_d,
                       int64_t N, float w_rgb, float w_mask, float w_depth, float *gt_rgb, float *valid, float *out, void *stream);
int mh_render_loss_bwd(const float *pred_rgb, const float *pred_depth, const float *opacity, const float *gt_rgb,
                       const float *depth, const float *mask, const float *valid, int64_t N, float w_rgb, float w_mask,
                       float w_depth, const float *g, float *g_rgb, float *g_depth, float *g_opacity, void *stream);

/* ---- HIP-graph hygiene (no reference counterpart: the reference runs its step eagerly; morpheus.py:1147-1236 is the step
 * trainstep.GraphedRealViewStep captures).  graph = a hipGraph_t obtained by stream capture, not yet instantiated.
 * On ROCm 7.2 a small memset node replays wrongly from the second launch on (csrc/graph.hip); the library itself never
 * memsets, PyTorch's multi-block reductions do.  mh_graph_count_memset_nodes reports what a captured graph holds;
 * mh_graph_replace_memset_nodes turns every memset node into a fill-kernel node with the same edges (1-, 2- or 4-byte
 * patterns, 1-D or pitched 2-D; anything else -> MH_ERR_ARG and the graph is left as far as it got: discard it). */
int mh_graph_count_memset_nodes(void *graph, int64_t *n_nodes, int64_t *n_memset, int64_t *smallest_bytes);
int mh_graph_replace_memset_nodes(void *graph, int64_t *n_replaced);

#ifdef __cplusplus
}
#endif
#endif /* MORPHEUS_HIP_H */
