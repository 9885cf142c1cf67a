/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported, linked or executed by the
 * product path (morpheus_amd/); only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may use it, and there only as the checker.
 *
 * Scalar CPU restatement of the reference's multiresolution hash-grid encoder
 * (CUDA-only in the reference; it cannot be built in this image -- it needs
 * nvcc + torch's CUDA headers -- so parity for this operator is pinned by
 * construction and by the property tests in tests/test_oracle_hashgrid.py:
 * "parity unpinned" at the operator level).
 *
 * Follows /root/reference/external/encoders/gridencoder/src/gridencoder.cu:
 *   fast_hash            :45-58     -> og_hash3
 *   get_grid_index       :61-79     -> og_index
 *   kernel_grid          :83-249    -> oracle_grid_forward
 *   kernel_grid_backward :253-349   -> oracle_grid_backward (embedding grads)
 *   kernel_input_backward:353-378   -> oracle_grid_backward (input grads)
 * and the host logic of grid.py:28-70 (level-major internal layout permuted to
 * [B, L*C]; levels >= max_level are zero).  D = 3, linear interpolation,
 * align_corners = False, gridtype = hash: the only configuration the model
 * instantiates (models/model.py:144-157).
 *
 * Arithmetic follows what nvcc emits for the reference kernel under its default -fmad=true:
 * `u*res - 0.5` and the `+= w * value` accumulations are single-rounding FMAs (explicit fmaf
 * here and in csrc/hashgrid.hip, with contraction otherwise disabled on both sides).
 *
 * Per-level resolutions are supplied by the caller (host-computed float32
 * ceil(exp2f(l*S)*H), gridencoder.cu:133) so that oracle and HIP kernel share
 * one table instead of two libm's.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

static inline uint32_t og_hash3(const uint32_t p[3]) {
    return (p[0] * 1u) ^ (p[1] * 2654435761u) ^ (p[2] * 805459861u);
}

static inline uint32_t og_index(uint32_t hashmap_size, uint32_t res, const uint32_t p[3]) {
    uint32_t stride = 1, index = 0;
    for (int d = 0; d < 3 && stride <= hashmap_size; d++) {
        index += p[d] * stride;
        stride *= res;
    }
    if (stride > hashmap_size) index = og_hash3(p);
    return index % hashmap_size;
}

/* u: [B,3] already normalised to [0,1]; emb: [rows,C]; offsets: [L+1]; res: [L]
 * out: [B, L*C] (level-major, channel-minor); dydx: NULL or [B, L, 3, C]. */
void oracle_grid_forward(const float *u, const float *emb, const int32_t *offsets,
                         const int32_t *res_tab, float *out, float *dydx,
                         int64_t B, int32_t L, int32_t C, int32_t max_level) {
    memset(out, 0, sizeof(float) * (size_t)B * L * C);
    if (dydx) memset(dydx, 0, sizeof(float) * (size_t)B * L * 3 * C);
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; b++) {
        const float *x = u + b * 3;
        int oob = 0;
        for (int d = 0; d < 3; d++)
            if (x[d] < 0.0f || x[d] > 1.0f) oob = 1;
        if (oob) continue; /* zeros, zero slope (gridencoder.cu:105-130) */
        for (int l = 0; l < max_level; l++) {
            const float *g = emb + (size_t)offsets[l] * C;
            uint32_t T = (uint32_t)(offsets[l + 1] - offsets[l]);
            uint32_t res = (uint32_t)res_tab[l];
            float pos[3];
            uint32_t pg[3];
            for (int d = 0; d < 3; d++) {
                float p = fminf(fmaxf(fmaf(x[d], (float)res, -0.5f), 0.0f), (float)(res - 1));
                pg[d] = (uint32_t)floorf(p);
                pos[d] = p - (float)pg[d];
            }
            float *o = out + (b * L + l) * C;
            for (uint32_t idx = 0; idx < 8; idx++) {
                float w = 1.0f;
                uint32_t pl[3];
                for (int d = 0; d < 3; d++) {
                    if ((idx & (1u << d)) == 0) {
                        w *= 1.0f - pos[d];
                        pl[d] = pg[d];
                    } else {
                        w *= pos[d];
                        pl[d] = (pg[d] + 1 < res - 1) ? pg[d] + 1 : res - 1;
                    }
                }
                uint32_t row = og_index(T, res, pl);
                for (int c = 0; c < C; c++) o[c] = fmaf(w, g[(size_t)row * C + c], o[c]);
            }
            if (dydx) {
                float *dd = dydx + ((b * L + l) * 3) * C;
                for (int gd = 0; gd < 3; gd++) {
                    for (uint32_t idx = 0; idx < 4; idx++) {
                        float w = (float)res;
                        uint32_t pl[3];
                        for (int nd = 0; nd < 2; nd++) {
                            int d = (nd >= gd) ? nd + 1 : nd;
                            if ((idx & (1u << nd)) == 0) {
                                w *= 1.0f - pos[d];
                                pl[d] = pg[d];
                            } else {
                                w *= pos[d];
                                pl[d] = (pg[d] + 1 < res - 1) ? pg[d] + 1 : res - 1;
                            }
                        }
                        pl[gd] = pg[gd];
                        uint32_t left = og_index(T, res, pl);
                        pl[gd] = (pg[gd] + 1 < res - 1) ? pg[gd] + 1 : res - 1;
                        uint32_t right = og_index(T, res, pl);
                        for (int c = 0; c < C; c++)
                            dd[gd * C + c] += w * (g[(size_t)right * C + c] - g[(size_t)left * C + c]);
                    }
                }
            }
        }
    }
}

/* grad: [B, L*C]; grad_emb: [rows,C] pre-zeroed by caller; grad_u: NULL or [B,3]
 * (needs dydx from the forward).  Per table row the scatter order is b ascending,
 * corner ascending: one fixed summation order (the reference's float atomics have none). */
void oracle_grid_backward(const float *grad, const float *u, const int32_t *offsets,
                          const int32_t *res_tab, const float *dydx, float *grad_emb,
                          float *grad_u, int64_t B, int32_t L, int32_t C, int32_t max_level) {
    if (grad_u) {
        /* kernel_input_backward sums over ALL L levels; skipped levels carry zero dydx */
#pragma omp parallel for schedule(static)
        for (int64_t b = 0; b < B; b++) {
            for (int d = 0; d < 3; d++) {
                float r = 0.0f;
                for (int l = 0; l < L; l++)
                    for (int c = 0; c < C; c++)
                        r += grad[(b * L + l) * C + c] * dydx[((b * L + l) * 3 + d) * C + c];
                grad_u[b * 3 + d] = r;
            }
        }
    }
    /* levels own disjoint table regions -> parallel over levels, b ascending inside:
     * a fixed, thread-count-independent summation order per table row */
#pragma omp parallel for schedule(dynamic, 1)
    for (int l = 0; l < max_level; l++) {
        float *gg = grad_emb + (size_t)offsets[l] * C;
        uint32_t T = (uint32_t)(offsets[l + 1] - offsets[l]);
        uint32_t res = (uint32_t)res_tab[l];
        for (int64_t b = 0; b < B; b++) {
            const float *x = u + b * 3;
            int oob = 0;
            for (int d = 0; d < 3; d++)
                if (x[d] < 0.0f || x[d] > 1.0f) oob = 1;
            if (oob) continue;
            float pos[3];
            uint32_t pg[3];
            for (int d = 0; d < 3; d++) {
                float p = fminf(fmaxf(fmaf(x[d], (float)res, -0.5f), 0.0f), (float)(res - 1));
                pg[d] = (uint32_t)floorf(p);
                pos[d] = p - (float)pg[d];
            }
            const float *gr = grad + (b * L + l) * C;
            for (uint32_t idx = 0; idx < 8; idx++) {
                float w = 1.0f;
                uint32_t pl[3];
                for (int d = 0; d < 3; d++) {
                    if ((idx & (1u << d)) == 0) {
                        w *= 1.0f - pos[d];
                        pl[d] = pg[d];
                    } else {
                        w *= pos[d];
                        pl[d] = (pg[d] + 1 < res - 1) ? pg[d] + 1 : res - 1;
                    }
                }
                uint32_t row = og_index(T, res, pl);
                for (int c = 0; c < C; c++) gg[(size_t)row * C + c] += w * gr[c];
            }
        }
    }
}
