/*
 * oracle/raster_oracle.c  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the tile-based differentiable 3D-Gaussian-splat rasterizer
 * that DreamWaltz-G calls through `diff_gaussian_rasterization` (ashawkey fork):
 *   call site  : /root/reference/core/gaussian/gaussian_renderer.py:5,60-70,186-195
 *   contract   : SURVEY.md section 8a rows R3 (forward) / R4 (backward)
 *
 * PARITY UNPINNED: the third-party CUDA package (requirements.txt:3
 * `diff_gaussian_rasterization==0.0.0`, git HEAD of ashawkey/diff-gaussian-rasterization,
 * scripts/install.sh:30) is not vendored under /root/reference and cannot be built here.
 * This file restates the *published* 3DGS algorithm (Kerbl et al. 2023, EWA splatting) with
 * every constant listed in SURVEY.md R3:
 *   cull p_view.z <= 0.2 | w + 1e-7 | tx,ty clamp 1.3*tanfov | +0.3 low-pass | radius ceil(3 sqrt(lmax))
 *   with max(0.1, mid^2-det) | 16x16 tiles | power>0 skip | alpha=min(0.99, o*exp(p)) |
 *   alpha<1/255 skip | stop when T*(1-alpha) < 1e-4 | out = C + T_final*bg ; depth/alpha un-normalised.
 * The analytic backward is validated against central finite differences of the forward
 * in float64 (tests/test_oracle_raster.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this.
 *
 * Build twice (see oracle/Makefile): -DREAL=float -> liboracle_raster_f32.so,
 *                                    -DREAL=double -> liboracle_raster_f64.so.
 * Ordering rule (identical in the HIP path): inside a tile Gaussians are composited in
 * ascending (depth, gaussian index) order -- the radix sort of the reference is stable
 * w.r.t. the Gaussian-ordered key emission, which gives the same tie-break.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef REAL
#define REAL float
#endif

#define TILE 16

typedef struct {
    REAL x, y;        /* pixel-space mean */
    REAL ca, cb, cc;  /* conic (inverse 2-D covariance) */
    REAL opacity;
    REAL rgb[3];
    REAL depth;       /* p_view.z */
    int radius;
    int tx0, ty0, tx1, ty1; /* tile rect [tx0,tx1) x [ty0,ty1) */
    int clamped[3];   /* SH colour clamp flags */
} splat_t;

typedef struct { uint64_t key; } pair_t;

/* SH basis constants (same numbers as core/gaussian/spherical_harmonics.py:8-40 of the reference) */
static const double SH_C0 = 0.28209479177387814;
static const double SH_C1 = 0.4886025119029199;
static const double SH_C2[5] = {1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
                                -1.0925484305920792, 0.5462742152960396};
static const double SH_C3[7] = {-0.5900435899266435, 2.890611442640554, -0.4570457994644658,
                                0.3731763325901154, -0.4570457994644658, 1.445305721320277,
                                -0.5900435899266435};

typedef struct {
    int G, H, W, tiles_x, tiles_y;
    splat_t* s;
    REAL* cov3d; /* G*6 */
    int64_t K;
    uint32_t* tile_start; /* tiles+1 */
    uint32_t* sorted;     /* K gaussian ids */
} state_t;

/* Two builds of this file (oracle/Makefile): the CHECKER (single-threaded, deterministic summation order: what tests/ and smoke() compare
 * against) and the `_omp` build (-fopenmp: tiles / Gaussians across all host cores, gradient sums through atomics) that only
 * bench.py's cpu_baseline leg times -- so that the reported host number is not a one-core number. */
#ifdef _OPENMP
#define DWG_ADD(x, v) _Pragma("omp atomic") (x) += (v)
#else
#define DWG_ADD(x, v) (x) += (v)
#endif

static uint32_t fbits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static void quat_to_R(const REAL* q, REAL R[9]) {
    /* q = (r,x,y,z) used UN-NORMALISED (SURVEY R3 / checklist Q2). Row-major standard rotation. */
    REAL r = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - r * z);     R[2] = 2 * (x * z + r * y);
    R[3] = 2 * (x * y + r * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - r * x);
    R[6] = 2 * (x * z - r * y);     R[7] = 2 * (y * z + r * x);     R[8] = 1 - 2 * (x * x + y * y);
}

static void cov3d_from_scale_rot(const REAL* sc, REAL mod, const REAL* q, REAL c6[6]) {
    REAL R[9]; quat_to_R(q, R);
    REAL s[3] = {mod * sc[0], mod * sc[1], mod * sc[2]};
    /* M = R * diag(s); Sigma = M M^T */
    REAL M[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M[i * 3 + j] = R[i * 3 + j] * s[j];
    REAL S[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        REAL a = 0; for (int k = 0; k < 3; k++) a += M[i * 3 + k] * M[j * 3 + k];
        S[i * 3 + j] = a;
    }
    c6[0] = S[0]; c6[1] = S[1]; c6[2] = S[2]; c6[3] = S[4]; c6[4] = S[5]; c6[5] = S[8];
}

/* viewmatrix / projmatrix follow the reference's row-vector convention
 * (gaussian_renderer.py:38-39: viewmatrix = extrinsic^T, projmatrix = viewmatrix @ projection^T),
 * i.e. flat index [4*c + r] holds element (row r, col c) of the column-vector matrix. */
static void xform43(const REAL* m, const REAL* p, REAL o[3]) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
static void xform44(const REAL* m, const REAL* p, REAL o[4]) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
    o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

static void eval_sh_color(int deg, int ncoef, const REAL* sh /* [ncoef][3] */, const REAL* pos,
                          const REAL* campos, REAL rgb[3], int clamped[3]) {
    REAL d[3] = {pos[0] - campos[0], pos[1] - campos[1], pos[2] - campos[2]};
    REAL n = (REAL)sqrt((double)(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]));
    REAL x = d[0] / n, y = d[1] / n, z = d[2] / n;
    (void)ncoef;
    for (int c = 0; c < 3; c++) {
#define SHC(k) sh[(k) * 3 + c]
        REAL r = (REAL)SH_C0 * SHC(0);
        if (deg > 0) {
            r = r - (REAL)SH_C1 * y * SHC(1) + (REAL)SH_C1 * z * SHC(2) - (REAL)SH_C1 * x * SHC(3);
            if (deg > 1) {
                REAL xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                r = r + (REAL)SH_C2[0] * xy * SHC(4) + (REAL)SH_C2[1] * yz * SHC(5) +
                    (REAL)SH_C2[2] * (2 * zz - xx - yy) * SHC(6) + (REAL)SH_C2[3] * xz * SHC(7) +
                    (REAL)SH_C2[4] * (xx - yy) * SHC(8);
                if (deg > 2) {
                    r = r + (REAL)SH_C3[0] * y * (3 * xx - yy) * SHC(9) + (REAL)SH_C3[1] * xy * z * SHC(10) +
                        (REAL)SH_C3[2] * y * (4 * zz - xx - yy) * SHC(11) +
                        (REAL)SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * SHC(12) +
                        (REAL)SH_C3[4] * x * (4 * zz - xx - yy) * SHC(13) +
                        (REAL)SH_C3[5] * z * (xx - yy) * SHC(14) + (REAL)SH_C3[6] * x * (xx - 3 * yy) * SHC(15);
                }
            }
        }
#undef SHC
        r += (REAL)0.5;
        clamped[c] = r < 0;
        rgb[c] = r < 0 ? 0 : r;
    }
}

static int cmp_u64(const void* a, const void* b) {
    uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

static void free_state(state_t* st) {
    free(st->s); free(st->cov3d); free(st->tile_start); free(st->sorted);
}

/* preprocess + binning + per-tile depth sort */
static int build_state(state_t* st, int G, int H, int W, const REAL* means3D, const REAL* colors,
                       const REAL* shs, int sh_degree, int sh_ncoef, const REAL* campos,
                       const REAL* opac, const REAL* scales, const REAL* quats, const REAL* cov3D_precomp,
                       const REAL* viewm, const REAL* projm, REAL tanfovx, REAL tanfovy, REAL scale_mod,
                       int* radii) {
    memset(st, 0, sizeof(*st));
    st->G = G; st->H = H; st->W = W;
    st->tiles_x = (W + TILE - 1) / TILE; st->tiles_y = (H + TILE - 1) / TILE;
    int ntiles = st->tiles_x * st->tiles_y;
    st->s = (splat_t*)calloc((size_t)(G > 0 ? G : 1), sizeof(splat_t));
    st->cov3d = (REAL*)calloc((size_t)(G > 0 ? G : 1) * 6, sizeof(REAL));
    st->tile_start = (uint32_t*)calloc((size_t)ntiles + 1, sizeof(uint32_t));
    const REAL focal_x = W / (2 * tanfovx), focal_y = H / (2 * tanfovy);
    uint32_t* counts = (uint32_t*)calloc((size_t)ntiles, sizeof(uint32_t));
#pragma omp parallel for schedule(static)
    for (int i = 0; i < G; i++) {
        splat_t* s = &st->s[i];
        s->radius = 0; if (radii) radii[i] = 0;
        const REAL* p = means3D + 3 * i;
        REAL pv[3]; xform43(viewm, p, pv);
        if (pv[2] <= (REAL)0.2) continue;
        REAL ph[4]; xform44(projm, p, ph);
        REAL pw = 1 / (ph[3] + (REAL)1e-7);
        REAL pp[3] = {ph[0] * pw, ph[1] * pw, ph[2] * pw};
        REAL* c6 = st->cov3d + 6 * i;
        if (cov3D_precomp) memcpy(c6, cov3D_precomp + 6 * i, 6 * sizeof(REAL));
        else cov3d_from_scale_rot(scales + 3 * i, scale_mod, quats + 4 * i, c6);
        /* EWA 2-D covariance */
        REAL t[3] = {pv[0], pv[1], pv[2]};
        REAL limx = (REAL)1.3 * tanfovx, limy = (REAL)1.3 * tanfovy;
        REAL txtz = t[0] / t[2], tytz = t[1] / t[2];
        t[0] = (txtz < -limx ? -limx : (txtz > limx ? limx : txtz)) * t[2];
        t[1] = (tytz < -limy ? -limy : (tytz > limy ? limy : tytz)) * t[2];
        REAL J[6] = {focal_x / t[2], 0, -(focal_x * t[0]) / (t[2] * t[2]),
                     0, focal_y / t[2], -(focal_y * t[1]) / (t[2] * t[2])};
        /* Wr = rotation rows of the world->camera matrix: Wr[r][c] = viewm[4*c + r] */
        REAL M[6]; /* M = J (2x3) * Wr (3x3) */
        for (int r = 0; r < 2; r++) for (int c = 0; c < 3; c++) {
            REAL a = 0; for (int k = 0; k < 3; k++) a += J[r * 3 + k] * viewm[4 * c + k];
            M[r * 3 + c] = a;
        }
        REAL S[9] = {c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5]};
        REAL MS[6];
        for (int r = 0; r < 2; r++) for (int c = 0; c < 3; c++) {
            REAL a = 0; for (int k = 0; k < 3; k++) a += M[r * 3 + k] * S[k * 3 + c];
            MS[r * 3 + c] = a;
        }
        REAL a = MS[0] * M[0] + MS[1] * M[1] + MS[2] * M[2] + (REAL)0.3;
        REAL b = MS[0] * M[3] + MS[1] * M[4] + MS[2] * M[5];
        REAL c = MS[3] * M[3] + MS[4] * M[4] + MS[5] * M[5] + (REAL)0.3;
        REAL det = a * c - b * b;
        if (det == 0) continue;
        REAL di = 1 / det;
        s->ca = c * di; s->cb = -b * di; s->cc = a * di;
        REAL mid = (REAL)0.5 * (a + c);
        REAL q = mid * mid - det; if (q < (REAL)0.1) q = (REAL)0.1;
        REAL sq = (REAL)sqrt((double)q);
        REAL l1 = mid + sq, l2 = mid - sq;
        REAL lm = l1 > l2 ? l1 : l2;
        int radius = (int)ceil((double)(3 * (REAL)sqrt((double)lm)));
        s->x = ((pp[0] + 1) * W - 1) * (REAL)0.5;
        s->y = ((pp[1] + 1) * H - 1) * (REAL)0.5;
        int gx = st->tiles_x, gy = st->tiles_y;
        int v;
#define CLAMPI(val, hi) (v = (val), v < 0 ? 0 : (v > (hi) ? (hi) : v))
        s->tx0 = CLAMPI((int)((s->x - radius) / TILE), gx);
        s->ty0 = CLAMPI((int)((s->y - radius) / TILE), gy);
        s->tx1 = CLAMPI((int)((s->x + radius + TILE - 1) / TILE), gx);
        s->ty1 = CLAMPI((int)((s->y + radius + TILE - 1) / TILE), gy);
#undef CLAMPI
        if ((s->tx1 - s->tx0) * (s->ty1 - s->ty0) == 0) continue;
        if (colors) { s->rgb[0] = colors[3 * i]; s->rgb[1] = colors[3 * i + 1]; s->rgb[2] = colors[3 * i + 2]; }
        else eval_sh_color(sh_degree, sh_ncoef, shs + (size_t)i * sh_ncoef * 3, p, campos, s->rgb, s->clamped);
        s->depth = pv[2];
        s->opacity = opac[i];
        s->radius = radius; if (radii) radii[i] = radius;
        for (int ty = s->ty0; ty < s->ty1; ty++) for (int tx = s->tx0; tx < s->tx1; tx++) { DWG_ADD(counts[ty * gx + tx], 1u); }
    }
    int64_t K = 0;
    for (int t = 0; t < ntiles; t++) { st->tile_start[t] = (uint32_t)K; K += counts[t]; }
    st->tile_start[ntiles] = (uint32_t)K;
    st->K = K;
    uint64_t* keys = (uint64_t*)malloc((size_t)(K > 0 ? K : 1) * sizeof(uint64_t));
    memset(counts, 0, (size_t)ntiles * sizeof(uint32_t));
    for (int i = 0; i < G; i++) {
        splat_t* s = &st->s[i];
        if (s->radius <= 0) continue;
        uint64_t key = ((uint64_t)fbits((float)s->depth) << 32) | (uint32_t)i;
        for (int ty = s->ty0; ty < s->ty1; ty++) for (int tx = s->tx0; tx < s->tx1; tx++) {
            int t = ty * st->tiles_x + tx;
            keys[st->tile_start[t] + counts[t]++] = key;
        }
    }
#pragma omp parallel for schedule(dynamic, 4)
    for (int t = 0; t < ntiles; t++) {
        uint32_t a = st->tile_start[t], b = st->tile_start[t + 1];
        if (b - a > 1) qsort(keys + a, b - a, sizeof(uint64_t), cmp_u64);
    }
    st->sorted = (uint32_t*)malloc((size_t)(K > 0 ? K : 1) * sizeof(uint32_t));
    for (int64_t k = 0; k < K; k++) st->sorted[k] = (uint32_t)(keys[k] & 0xffffffffu);
    free(keys); free(counts);
    return 0;
}

/* near (optional, [H*W] bytes): which hard thresholds of the compositing loop this pixel sits ON, within a relative margin -- the
 * pixels where two correct implementations whose exp() / products round differently may legitimately take different branches:
 *   bit 0  a splat's alpha within `rel_a` + |grad power| x `pos_eps` of the 1/255 cut (pos_eps: how far two fp32 projections of the same
 *          centre may lie apart, ~2 ulps of the pixel coordinate)    bit 1  test_T within `rel_t` of the 1e-4 termination threshold
 *   bit 2  a contributing splat and its successor in the tile list have depths equal to within 4 fp32 ulps (sort order)
 *   bit 3  power within 1e-6 of 0 (the power > 0 skip)               bit 4  alpha within `rel_a` of the 0.99 clamp: harmless, recorded only
 * The parity tests demand |err| <= 1e-4 on every pixel whose byte is 0 and count / bound the rest (tests/raster_cases.py). */
static void composite_forward(const state_t* st, const REAL* bg, REAL* out_color, REAL* out_depth,
                              REAL* out_alpha, REAL* final_T, int* n_contrib, unsigned char* near, REAL rel_a, REAL rel_t, REAL pos_eps) {
    int H = st->H, W = st->W;
#pragma omp parallel for collapse(2) schedule(dynamic, 2)
    for (int ty = 0; ty < st->tiles_y; ty++) for (int tx = 0; tx < st->tiles_x; tx++) {
        int t = ty * st->tiles_x + tx;
        uint32_t a = st->tile_start[t], b = st->tile_start[t + 1];
        for (int py = ty * TILE; py < (ty + 1) * TILE && py < H; py++)
        for (int px = tx * TILE; px < (tx + 1) * TILE && px < W; px++) {
            REAL T = 1, C[3] = {0, 0, 0}, D = 0, A = 0;
            int contributor = 0, last = 0;
            unsigned char nr = 0;
            for (uint32_t k = a; k < b; k++) {
                contributor++;
                const splat_t* s = &st->s[st->sorted[k]];
                REAL dx = s->x - (REAL)px, dy = s->y - (REAL)py;
                REAL power = (REAL)-0.5 * (s->ca * dx * dx + s->cc * dy * dy) - s->cb * dx * dy;
                if (near && power > (REAL)-1e-6 && power < (REAL)1e-6) nr |= 8;
                if (power > 0) continue;
                REAL alpha = s->opacity * (REAL)exp((double)power);
                if (near) {
                    const REAL cut = (REAL)(1.0 / 255.0);
                    REAL gx = s->ca * dx + s->cb * dy, gy = s->cb * dx + s->cc * dy;
                    if (gx < 0) gx = -gx;
                    if (gy < 0) gy = -gy;
                    const REAL ra = rel_a + (gx + gy) * pos_eps;
                    if (alpha > cut * (1 - ra) && alpha < cut * (1 + ra)) nr |= 1;
                    if (alpha > (REAL)0.99 * (1 - rel_a) && alpha < (REAL)0.99 * (1 + rel_a)) nr |= 16;
                }
                if (alpha > (REAL)0.99) alpha = (REAL)0.99;
                if (alpha < (REAL)(1.0 / 255.0)) continue;
                REAL test_T = T * (1 - alpha);
                if (near) {
                    if (test_T > (REAL)0.0001 * (1 - rel_t) && test_T < (REAL)0.0001 * (1 + rel_t)) nr |= 2;
                    if (k + 1 < b) {
                        const REAL d0 = s->depth, d1 = st->s[st->sorted[k + 1]].depth;
                        const REAL dd = d1 > d0 ? d1 - d0 : d0 - d1, mag = d0 > 0 ? d0 : -d0;
                        if (dd <= (REAL)4.8e-7 * mag) {          /* ... and the successor is visible at this pixel too */
                            const splat_t* s2 = &st->s[st->sorted[k + 1]];
                            REAL ex = s2->x - (REAL)px, ey = s2->y - (REAL)py;
                            REAL pw2 = (REAL)-0.5 * (s2->ca * ex * ex + s2->cc * ey * ey) - s2->cb * ex * ey;
                            if (pw2 <= 0 && s2->opacity * (REAL)exp((double)pw2) >= (REAL)(0.5 / 255.0)) nr |= 4;
                        }
                    }
                }
                if (test_T < (REAL)0.0001) break;
                REAL w = alpha * T;
                C[0] += s->rgb[0] * w; C[1] += s->rgb[1] * w; C[2] += s->rgb[2] * w;
                D += s->depth * w; A += w;
                T = test_T; last = contributor;
            }
            size_t pix = (size_t)py * W + px, P = (size_t)H * W;
            if (final_T) final_T[pix] = T;
            if (n_contrib) n_contrib[pix] = last;
            if (near) near[pix] = nr;
            out_color[pix] = C[0] + T * bg[0];
            out_color[P + pix] = C[1] + T * bg[1];
            out_color[2 * P + pix] = C[2] + T * bg[2];
            out_depth[pix] = D; out_alpha[pix] = A;
        }
    }
}

int dwg_oracle_raster_forward(int G, int H, int W, const REAL* means3D, const REAL* colors,
                              const REAL* shs, int sh_degree, int sh_ncoef, const REAL* campos,
                              const REAL* opac, const REAL* scales, const REAL* quats,
                              const REAL* cov3D_precomp, const REAL* viewm, const REAL* projm,
                              REAL tanfovx, REAL tanfovy, const REAL* bg, REAL scale_mod,
                              REAL* out_color, REAL* out_depth, REAL* out_alpha, int* radii,
                              REAL* final_T, int* n_contrib, int64_t* num_pairs) {
    state_t st;
    build_state(&st, G, H, W, means3D, colors, shs, sh_degree, sh_ncoef, campos, opac, scales, quats,
                cov3D_precomp, viewm, projm, tanfovx, tanfovy, scale_mod, radii);
    composite_forward(&st, bg, out_color, out_depth, out_alpha, final_T, n_contrib, NULL, 0, 0, 0);
    if (num_pairs) *num_pairs = st.K;
    free_state(&st);
    return 0;
}

/* The same forward, plus the per-pixel threshold-proximity byte described at composite_forward (test bookkeeping, not a product of the
 * algorithm restated here). */
int dwg_oracle_raster_forward_near(int G, int H, int W, const REAL* means3D, const REAL* colors,
                                   const REAL* shs, int sh_degree, int sh_ncoef, const REAL* campos,
                                   const REAL* opac, const REAL* scales, const REAL* quats,
                                   const REAL* cov3D_precomp, const REAL* viewm, const REAL* projm,
                                   REAL tanfovx, REAL tanfovy, const REAL* bg, REAL scale_mod,
                                   REAL* out_color, REAL* out_depth, REAL* out_alpha, int* radii,
                                   REAL* final_T, int* n_contrib, int64_t* num_pairs, unsigned char* near, REAL rel_a, REAL rel_t,
                                   REAL pos_eps) {
    state_t st;
    build_state(&st, G, H, W, means3D, colors, shs, sh_degree, sh_ncoef, campos, opac, scales, quats,
                cov3D_precomp, viewm, projm, tanfovx, tanfovy, scale_mod, radii);
    composite_forward(&st, bg, out_color, out_depth, out_alpha, final_T, n_contrib, near, rel_a, rel_t, pos_eps);
    if (num_pairs) *num_pairs = st.K;
    free_state(&st);
    return 0;
}

/* Backward. Gradient conventions follow 3DGS:
 *  - no gradient gating at the 0.99 alpha clamp (the clamp is transparent to the chain rule);
 *  - tx/ty clamp in the EWA Jacobian zeroes the corresponding mean gradient term;
 *  - dL_dmeans2D is expressed in NDC units (factor 0.5*W / 0.5*H applied), z component 0;
 *  - depth gradient reaches means3D through p_view.z (third row of the view matrix).
 * grad_out_color [3,H,W], grad_out_depth [H,W], grad_out_alpha [H,W] (either may be NULL = zero). */
int dwg_oracle_raster_backward(int G, int H, int W, const REAL* means3D, const REAL* colors,
                               const REAL* shs, int sh_degree, int sh_ncoef, const REAL* campos,
                               const REAL* opac, const REAL* scales, const REAL* quats,
                               const REAL* cov3D_precomp, const REAL* viewm, const REAL* projm,
                               REAL tanfovx, REAL tanfovy, const REAL* bg, REAL scale_mod,
                               const REAL* g_color, const REAL* g_depth, const REAL* g_alpha,
                               REAL* dL_dmeans3D, REAL* dL_dmeans2D, REAL* dL_dcolors, REAL* dL_dshs,
                               REAL* dL_dopac, REAL* dL_dscales, REAL* dL_dquats, REAL* dL_dcov3D) {
    state_t st;
    build_state(&st, G, H, W, means3D, colors, shs, sh_degree, sh_ncoef, campos, opac, scales, quats,
                cov3D_precomp, viewm, projm, tanfovx, tanfovy, scale_mod, NULL);
    size_t P = (size_t)H * W;
    REAL* oc = (REAL*)malloc(3 * P * sizeof(REAL)); REAL* od = (REAL*)malloc(P * sizeof(REAL));
    REAL* oa = (REAL*)malloc(P * sizeof(REAL)); REAL* fT = (REAL*)malloc(P * sizeof(REAL));
    int* nc = (int*)malloc(P * sizeof(int));
    composite_forward(&st, bg, oc, od, oa, fT, nc, NULL, 0, 0, 0);
    REAL* g2d = (REAL*)calloc((size_t)(G > 0 ? G : 1) * 2, sizeof(REAL));   /* d/d(pixel xy) * (0.5W,0.5H) */
    REAL* gcon = (REAL*)calloc((size_t)(G > 0 ? G : 1) * 3, sizeof(REAL));
    REAL* gop = (REAL*)calloc((size_t)(G > 0 ? G : 1), sizeof(REAL));
    REAL* gcol = (REAL*)calloc((size_t)(G > 0 ? G : 1) * 3, sizeof(REAL));
    REAL* gdep = (REAL*)calloc((size_t)(G > 0 ? G : 1), sizeof(REAL));
    const REAL ddelx_dx = (REAL)0.5 * W, ddely_dy = (REAL)0.5 * H;
#pragma omp parallel for collapse(2) schedule(dynamic, 2)
    for (int ty = 0; ty < st.tiles_y; ty++) for (int tx = 0; tx < st.tiles_x; tx++) {
        int t = ty * st.tiles_x + tx;
        uint32_t a = st.tile_start[t];
        for (int py = ty * TILE; py < (ty + 1) * TILE && py < H; py++)
        for (int px = tx * TILE; px < (tx + 1) * TILE && px < W; px++) {
            size_t pix = (size_t)py * W + px;
            REAL T_final = fT[pix], T = T_final;
            int last = nc[pix];
            REAL gp[3] = {g_color ? g_color[pix] : 0, g_color ? g_color[P + pix] : 0, g_color ? g_color[2 * P + pix] : 0};
            REAL gpd = g_depth ? g_depth[pix] : 0, gpa = g_alpha ? g_alpha[pix] : 0;
            REAL acc[3] = {0, 0, 0}, accd = 0, acca = 0; /* sum over later splats of value*w */
            REAL bgdot = bg[0] * gp[0] + bg[1] * gp[1] + bg[2] * gp[2];
            for (int j = last - 1; j >= 0; j--) {
                uint32_t gid = st.sorted[a + j];
                const splat_t* s = &st.s[gid];
                REAL dx = s->x - (REAL)px, dy = s->y - (REAL)py;
                REAL power = (REAL)-0.5 * (s->ca * dx * dx + s->cc * dy * dy) - s->cb * dx * dy;
                if (power > 0) continue;
                REAL Gv = (REAL)exp((double)power);
                REAL alpha = s->opacity * Gv; if (alpha > (REAL)0.99) alpha = (REAL)0.99;
                if (alpha < (REAL)(1.0 / 255.0)) continue;
                T = T / (1 - alpha);          /* transmittance in front of this splat */
                REAL w = alpha * T;
                /* d out / d alpha_j = T_j * value_j - (sum_{k>j} value_k w_k + T_final*bg) / (1-alpha_j) */
                REAL dL_dalpha = 0;
                REAL inv1a = 1 / (1 - alpha);
                for (int ch = 0; ch < 3; ch++) {
                    dL_dalpha += (s->rgb[ch] * T - acc[ch] * inv1a) * gp[ch];
                    DWG_ADD(gcol[3 * gid + ch], w * gp[ch]);
                    acc[ch] += s->rgb[ch] * w;
                }
                dL_dalpha += (s->depth * T - accd * inv1a) * gpd;
                DWG_ADD(gdep[gid], w * gpd); accd += s->depth * w;
                dL_dalpha += (T - acca * inv1a) * gpa; acca += w;
                dL_dalpha += (-T_final * inv1a) * bgdot;
                REAL dL_dG = s->opacity * dL_dalpha;
                REAL gdx = s->ca * dx + s->cb * dy, gdy = s->cc * dy + s->cb * dx;
                REAL dG_ddx = -Gv * gdx, dG_ddy = -Gv * gdy;
                DWG_ADD(g2d[2 * gid], dL_dG * dG_ddx * ddelx_dx);
                DWG_ADD(g2d[2 * gid + 1], dL_dG * dG_ddy * ddely_dy);
                DWG_ADD(gcon[3 * gid], (REAL)-0.5 * Gv * dx * dx * dL_dG);
                DWG_ADD(gcon[3 * gid + 1], (REAL)-0.5 * Gv * dx * dy * dL_dG); /* NB: b appears twice: stored as half, doubled below */
                DWG_ADD(gcon[3 * gid + 2], (REAL)-0.5 * Gv * dy * dy * dL_dG);
                DWG_ADD(gop[gid], Gv * dL_dalpha);
            }
        }
    }
    /* per-Gaussian chain */
    const REAL focal_x = W / (2 * tanfovx), focal_y = H / (2 * tanfovy);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < G; i++) {
        const splat_t* s = &st.s[i];
        REAL gm[3] = {0, 0, 0};
        if (dL_dmeans2D) { dL_dmeans2D[3 * i] = g2d[2 * i]; dL_dmeans2D[3 * i + 1] = g2d[2 * i + 1]; dL_dmeans2D[3 * i + 2] = 0; }
        if (dL_dopac) dL_dopac[i] = gop[i];
        if (dL_dcolors) for (int c = 0; c < 3; c++) dL_dcolors[3 * i + c] = gcol[3 * i + c];
        REAL gc6[6] = {0, 0, 0, 0, 0, 0};
        if (s->radius > 0) {
            const REAL* p = means3D + 3 * i;
            const REAL* c6 = st.cov3d + 6 * i;
            REAL pv[3]; xform43(viewm, p, pv);
            REAL limx = (REAL)1.3 * tanfovx, limy = (REAL)1.3 * tanfovy;
            REAL txtz = pv[0] / pv[2], tytz = pv[1] / pv[2];
            REAL t[3] = {(txtz < -limx ? -limx : (txtz > limx ? limx : txtz)) * pv[2],
                         (tytz < -limy ? -limy : (tytz > limy ? limy : tytz)) * pv[2], pv[2]};
            REAL xmul = (txtz < -limx || txtz > limx) ? 0 : 1, ymul = (tytz < -limy || tytz > limy) ? 0 : 1;
            REAL J[6] = {focal_x / t[2], 0, -(focal_x * t[0]) / (t[2] * t[2]),
                         0, focal_y / t[2], -(focal_y * t[1]) / (t[2] * t[2])};
            REAL Wr[9]; for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Wr[r * 3 + c] = viewm[4 * c + r];
            REAL M[6];
            for (int r = 0; r < 2; r++) for (int c = 0; c < 3; c++) {
                REAL acc2 = 0; for (int k = 0; k < 3; k++) acc2 += J[r * 3 + k] * Wr[k * 3 + c];
                M[r * 3 + c] = acc2;
            }
            REAL S[9] = {c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5]};
            REAL MS[6];
            for (int r = 0; r < 2; r++) for (int c = 0; c < 3; c++) {
                REAL acc2 = 0; for (int k = 0; k < 3; k++) acc2 += M[r * 3 + k] * S[k * 3 + c];
                MS[r * 3 + c] = acc2;
            }
            REAL a = MS[0] * M[0] + MS[1] * M[1] + MS[2] * M[2] + (REAL)0.3;
            REAL b = MS[0] * M[3] + MS[1] * M[4] + MS[2] * M[5];
            REAL c = MS[3] * M[3] + MS[4] * M[4] + MS[5] * M[5] + (REAL)0.3;
            REAL det = a * c - b * b;
            /* conic = (c, -b, a)/det ; gcon holds dL/d(ca), half dL/d(cb) (sym.), dL/d(cc) */
            REAL gA = gcon[3 * i], gB = 2 * gcon[3 * i + 1], gC = gcon[3 * i + 2];
            REAL d2 = 1 / (det * det);
            /* derivative of (c/det, -b/det, a/det) w.r.t. a,b,c */
            REAL dL_da = d2 * (-c * c * gA + b * c * gB + (det - a * c) * gC);
            REAL dL_dc = d2 * (-a * a * gC + a * b * gB + (det - a * c) * gA);
            REAL dL_db = d2 * (2 * b * c * gA - (det + 2 * b * b) * gB + 2 * a * b * gC);
            /* cov2D = M S M^T : dL/dS = M^T [[da, db/2],[db/2, dc]] M  (S symmetric, off-diagonals counted twice) */
            REAL Gm[4] = {dL_da, (REAL)0.5 * dL_db, (REAL)0.5 * dL_db, dL_dc};
            REAL dS[9];
            for (int r = 0; r < 3; r++) for (int cc2 = 0; cc2 < 3; cc2++) {
                REAL acc2 = 0;
                for (int u = 0; u < 2; u++) for (int v = 0; v < 2; v++) acc2 += M[u * 3 + r] * Gm[u * 2 + v] * M[v * 3 + cc2];
                dS[r * 3 + cc2] = acc2;
            }
            gc6[0] = dS[0]; gc6[3] = dS[4]; gc6[5] = dS[8];
            gc6[1] = dS[1] + dS[3]; gc6[2] = dS[2] + dS[6]; gc6[4] = dS[5] + dS[7];
            /* dL/dM = 2 * Gm * M * S  (Gm symmetric) */
            REAL dM[6];
            for (int u = 0; u < 2; u++) for (int cc2 = 0; cc2 < 3; cc2++) {
                REAL acc2 = 0; for (int v = 0; v < 2; v++) acc2 += Gm[u * 2 + v] * MS[v * 3 + cc2];
                dM[u * 3 + cc2] = 2 * acc2;
            }
            /* M = J Wr -> dL/dJ = dM Wr^T */
            REAL dJ[6];
            for (int u = 0; u < 2; u++) for (int k = 0; k < 3; k++) {
                REAL acc2 = 0; for (int cc2 = 0; cc2 < 3; cc2++) acc2 += dM[u * 3 + cc2] * Wr[k * 3 + cc2];
                dJ[u * 3 + k] = acc2;
            }
            REAL tz = 1 / t[2], tz2 = tz * tz, tz3 = tz2 * tz;
            REAL dtx = xmul * (-focal_x * tz2 * dJ[2]);
            REAL dty = ymul * (-focal_y * tz2 * dJ[5]);
            REAL dtz = -focal_x * tz2 * dJ[0] - focal_y * tz2 * dJ[4] + (2 * focal_x * t[0]) * tz3 * dJ[2] +
                       (2 * focal_y * t[1]) * tz3 * dJ[5];
            /* depth gradient: depth = p_view.z */
            dtz += gdep[i];
            /* t = Wr p + trans */
            for (int k = 0; k < 3; k++) gm[k] += Wr[0 * 3 + k] * dtx + Wr[1 * 3 + k] * dty + Wr[2 * 3 + k] * dtz;
            /* projection: pixel mean from projmatrix */
            REAL ph[4]; xform44(projm, p, ph);
            REAL mw = 1 / (ph[3] + (REAL)1e-7);
            REAL mul1 = (projm[0] * p[0] + projm[4] * p[1] + projm[8] * p[2] + projm[12]) * mw * mw;
            REAL mul2 = (projm[1] * p[0] + projm[5] * p[1] + projm[9] * p[2] + projm[13]) * mw * mw;
            REAL gx = g2d[2 * i], gy = g2d[2 * i + 1];
            gm[0] += (projm[0] * mw - projm[3] * mul1) * gx + (projm[1] * mw - projm[3] * mul2) * gy;
            gm[1] += (projm[4] * mw - projm[7] * mul1) * gx + (projm[5] * mw - projm[7] * mul2) * gy;
            gm[2] += (projm[8] * mw - projm[11] * mul1) * gx + (projm[9] * mw - projm[11] * mul2) * gy;
            /* SH colour backward */
            if (!colors && shs) {
                REAL d[3] = {p[0] - campos[0], p[1] - campos[1], p[2] - campos[2]};
                REAL n = (REAL)sqrt((double)(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]));
                REAL x = d[0] / n, y = d[1] / n, z = d[2] / n;
                REAL dRGBdx[3] = {0, 0, 0}, dRGBdy[3] = {0, 0, 0}, dRGBdz[3] = {0, 0, 0};
                REAL gdir[3] = {0, 0, 0};
                for (int ch = 0; ch < 3; ch++) {
                    REAL gr = s->clamped[ch] ? 0 : gcol[3 * i + ch];
                    const REAL* sh = shs + (size_t)i * sh_ncoef * 3;
                    REAL* gsh = dL_dshs ? dL_dshs + (size_t)i * sh_ncoef * 3 : NULL;
#define SHC(k) sh[(k) * 3 + ch]
#define GSH(k, v) do { if (gsh) gsh[(k) * 3 + ch] = (v) * gr; } while (0)
                    GSH(0, (REAL)SH_C0);
                    if (sh_degree > 0) {
                        GSH(1, -(REAL)SH_C1 * y); GSH(2, (REAL)SH_C1 * z); GSH(3, -(REAL)SH_C1 * x);
                        dRGBdx[ch] = -(REAL)SH_C1 * SHC(3); dRGBdy[ch] = -(REAL)SH_C1 * SHC(1); dRGBdz[ch] = (REAL)SH_C1 * SHC(2);
                        if (sh_degree > 1) {
                            REAL xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                            GSH(4, (REAL)SH_C2[0] * xy); GSH(5, (REAL)SH_C2[1] * yz); GSH(6, (REAL)SH_C2[2] * (2 * zz - xx - yy));
                            GSH(7, (REAL)SH_C2[3] * xz); GSH(8, (REAL)SH_C2[4] * (xx - yy));
                            dRGBdx[ch] += (REAL)SH_C2[0] * y * SHC(4) + (REAL)SH_C2[2] * 2 * -x * SHC(6) + (REAL)SH_C2[3] * z * SHC(7) + (REAL)SH_C2[4] * 2 * x * SHC(8);
                            dRGBdy[ch] += (REAL)SH_C2[0] * x * SHC(4) + (REAL)SH_C2[1] * z * SHC(5) + (REAL)SH_C2[2] * 2 * -y * SHC(6) + (REAL)SH_C2[4] * 2 * -y * SHC(8);
                            dRGBdz[ch] += (REAL)SH_C2[1] * y * SHC(5) + (REAL)SH_C2[2] * 2 * 2 * z * SHC(6) + (REAL)SH_C2[3] * x * SHC(7);
                            if (sh_degree > 2) {
                                GSH(9, (REAL)SH_C3[0] * y * (3 * xx - yy)); GSH(10, (REAL)SH_C3[1] * xy * z);
                                GSH(11, (REAL)SH_C3[2] * y * (4 * zz - xx - yy)); GSH(12, (REAL)SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy));
                                GSH(13, (REAL)SH_C3[4] * x * (4 * zz - xx - yy)); GSH(14, (REAL)SH_C3[5] * z * (xx - yy));
                                GSH(15, (REAL)SH_C3[6] * x * (xx - 3 * yy));
                                dRGBdx[ch] += (REAL)SH_C3[0] * SHC(9) * 3 * 2 * xy + (REAL)SH_C3[1] * SHC(10) * yz + (REAL)SH_C3[2] * SHC(11) * -2 * xy +
                                              (REAL)SH_C3[3] * SHC(12) * -3 * 2 * xz + (REAL)SH_C3[4] * SHC(13) * (-3 * xx + 4 * zz - yy) +
                                              (REAL)SH_C3[5] * SHC(14) * 2 * xz + (REAL)SH_C3[6] * SHC(15) * 3 * (xx - yy);
                                dRGBdy[ch] += (REAL)SH_C3[0] * SHC(9) * 3 * (xx - yy) + (REAL)SH_C3[1] * SHC(10) * xz + (REAL)SH_C3[2] * SHC(11) * (-3 * yy + 4 * zz - xx) +
                                              (REAL)SH_C3[3] * SHC(12) * -3 * 2 * yz + (REAL)SH_C3[4] * SHC(13) * -2 * xy +
                                              (REAL)SH_C3[5] * SHC(14) * -2 * yz + (REAL)SH_C3[6] * SHC(15) * -3 * 2 * xy;
                                dRGBdz[ch] += (REAL)SH_C3[1] * SHC(10) * xy + (REAL)SH_C3[2] * SHC(11) * 4 * 2 * yz + (REAL)SH_C3[3] * SHC(12) * 3 * (2 * zz - xx - yy) +
                                              (REAL)SH_C3[4] * SHC(13) * 4 * 2 * xz + (REAL)SH_C3[5] * SHC(14) * (xx - yy);
                            }
                        }
                    }
#undef SHC
#undef GSH
                    gdir[0] += dRGBdx[ch] * gr; gdir[1] += dRGBdy[ch] * gr; gdir[2] += dRGBdz[ch] * gr;
                }
                /* normalisation backward: dir = d/|d| */
                REAL inv = 1 / n, inv3 = inv * inv * inv;
                REAL dot = d[0] * gdir[0] + d[1] * gdir[1] + d[2] * gdir[2];
                for (int k = 0; k < 3; k++) gm[k] += gdir[k] * inv - d[k] * dot * inv3;
            }
        } else if (dL_dshs && shs) {
            memset(dL_dshs + (size_t)i * sh_ncoef * 3, 0, (size_t)sh_ncoef * 3 * sizeof(REAL));
        }
        if (dL_dmeans3D) for (int k = 0; k < 3; k++) dL_dmeans3D[3 * i + k] = gm[k];
        if (dL_dcov3D) for (int k = 0; k < 6; k++) dL_dcov3D[6 * i + k] = gc6[k];
        if (!cov3D_precomp && (dL_dscales || dL_dquats)) {
            /* Sigma = M M^T, M = R diag(s) ; dL/dM = 2 dSigma_sym M  with dSigma_sym built from gc6 */
            const REAL* q = quats + 4 * i; const REAL* sc = scales + 3 * i;
            REAL R[9]; quat_to_R(q, R);
            REAL sv[3] = {scale_mod * sc[0], scale_mod * sc[1], scale_mod * sc[2]};
            REAL dSig[9] = {gc6[0], (REAL)0.5 * gc6[1], (REAL)0.5 * gc6[2], (REAL)0.5 * gc6[1], gc6[3], (REAL)0.5 * gc6[4],
                            (REAL)0.5 * gc6[2], (REAL)0.5 * gc6[4], gc6[5]};
            REAL Mm[9]; for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Mm[r * 3 + c] = R[r * 3 + c] * sv[c];
            REAL dMm[9];
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) {
                REAL acc2 = 0; for (int k = 0; k < 3; k++) acc2 += dSig[r * 3 + k] * Mm[k * 3 + c];
                dMm[r * 3 + c] = 2 * acc2;
            }
            if (dL_dscales) for (int c = 0; c < 3; c++)
                dL_dscales[3 * i + c] = scale_mod * (R[0 * 3 + c] * dMm[0 * 3 + c] + R[1 * 3 + c] * dMm[1 * 3 + c] + R[2 * 3 + c] * dMm[2 * 3 + c]);
            if (dL_dquats) {
                REAL dR[9]; for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) dR[r * 3 + c] = dMm[r * 3 + c] * sv[c];
                REAL r = q[0], x = q[1], y = q[2], z = q[3];
                dL_dquats[4 * i + 0] = 2 * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] - y * dR[6] + x * dR[7]);
                dL_dquats[4 * i + 1] = 2 * (y * dR[1] + z * dR[2] + y * dR[3] - 2 * x * dR[4] - r * dR[5] + z * dR[6] + r * dR[7] - 2 * x * dR[8]);
                dL_dquats[4 * i + 2] = 2 * (-2 * y * dR[0] + x * dR[1] + r * dR[2] + x * dR[3] + z * dR[5] - r * dR[6] + z * dR[7] - 2 * y * dR[8]);
                dL_dquats[4 * i + 3] = 2 * (-2 * z * dR[0] - r * dR[1] + x * dR[2] + r * dR[3] - 2 * z * dR[4] + y * dR[5] + x * dR[6] + y * dR[7]);
            }
        } else {
            if (dL_dscales) for (int c = 0; c < 3; c++) dL_dscales[3 * i + c] = 0;
            if (dL_dquats) for (int c = 0; c < 4; c++) dL_dquats[4 * i + c] = 0;
        }
    }
    free(oc); free(od); free(oa); free(fT); free(nc);
    free(g2d); free(gcon); free(gop); free(gcol); free(gdep);
    free_state(&st);
    return 0;
}

/* SH colour of N points (get_colors: core/gaussian/gaussian_utils.py:12-17); used to pin eval_sh_color against the
 * golden vectors captured from the reference's own eval_sh. */
int dwg_oracle_sh_colors(int N, int deg, int ncoef, const REAL* shs, const REAL* pos, const REAL* campos, REAL* rgb) {
    for (int i = 0; i < N; i++) {
        int cl[3];
        eval_sh_color(deg, ncoef, shs + (size_t)i * ncoef * 3, pos + 3 * i, campos, rgb + 3 * i, cl);
    }
    return 0;
}
