/*
 * oracle/raster_ref.c -- CPU restatement of the Gaussian tile rasterizer forward pass.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under audio-motion-avatar_amd/ may include, link or call this
 * file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * PARITY UNPINNED: the reference (liubingqi7/audio-motion-avatar) calls the un-vendored, un-pinned
 * third-party CUDA package `diff_gaussian_rasterization` (reference README.md:117-121, call site
 * src/models/renderer.py:422,516-566) and ships no tests, fixtures or golden vectors for it.  This file
 * restates the published algorithm of that package (graphdeco-inria/diff-gaussian-rasterization, the
 * revision with `antialiasing` and inverse-depth output that the reference's call signature implies:
 * renderer.py:529,557) as recorded in SURVEY.md Appendix A.1.  It is anchored by the analytic
 * known-answer tests in tests/test_oracle_raster.py, not by vectors of the reference.
 *
 * What is restated (one frame per call, all arithmetic in `real`):
 *   preprocess : frustum cull (view z <= 0.2), projection, quaternion+scale -> Sigma3D, EWA Sigma2D,
 *                +0.3 px^2 low-pass, conic, 3-sigma radius, 16x16-tile rectangle      (A.1 steps 1-8)
 *   binning    : instances emitted y-outer/x-inner per Gaussian in ascending Gaussian index, stably
 *                ordered by (tile, depth)                                             (A.1 "Binning")
 *   blend      : per pixel front-to-back alpha compositing with the 0.99 / 1/255 / 1e-4 thresholds,
 *                colour + T*background, inverse depth, alpha = 1 - T                 (A.1 "Blend")
 *
 * Build: see oracle/Makefile (REAL=float -> liboracle_raster_f32.so, REAL=double -> _f64.so).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef REAL
#define REAL float
#endif
typedef REAL real;

#define TILE 16

#if defined(ORACLE_F64)
#define R_EXP exp
#define R_SQRT sqrt
#define R_CEIL ceil
#define R_FMIN fmin
#define R_FMAX fmax
#else
#define R_EXP expf
#define R_SQRT sqrtf
#define R_CEIL ceilf
#define R_FMIN fminf
#define R_FMAX fmaxf
#endif

typedef struct {
    real depth;
    int32_t id;
} inst_t;

/* column-major 4x4 * point, as the upstream kernel reads the (transposed) torch matrices */
static void xform4x4(const real *m, const real *p, real *o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
    o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

static void xform4x3(const real *m, const real *p, real *o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* stable merge sort of instances by depth (ties keep emit order = ascending Gaussian index) */
static void merge_sort(inst_t *a, inst_t *tmp, int n) {
    if (n < 2) return;
    int h = n / 2;
    merge_sort(a, tmp, h);
    merge_sort(a + h, tmp, n - h);
    int i = 0, j = h, k = 0;
    while (i < h && j < n) tmp[k++] = (a[j].depth < a[i].depth) ? a[j++] : a[i++];
    while (i < h) tmp[k++] = a[i++];
    while (j < n) tmp[k++] = a[j++];
    memcpy(a, tmp, (size_t)n * sizeof(inst_t));
}

/*
 * One frame.  Inputs are the arguments the reference hands to GaussianRasterizer (renderer.py:557-566)
 * AFTER its own activations (renderer.py:532-547): scales are metric, opacities in (0,1), colours in [0,1],
 * rotations unit (w,x,y,z).  view/proj are the 16 floats of the transposed matrices (renderer.py:507-509).
 *
 * Outputs: color [3,H,W] planar, alpha [H,W] (= 1 - final T), inv_depth [H,W], radii [N], and (optional, may be
 * NULL) unstable [H,W]: 1 where some blend decision of the pixel sat within a relative margin of 1e-4 of one of the
 * algorithm's discontinuities (power > 0, alpha < 1/255, T' < 1e-4).  At such a pixel two correct fp32
 * implementations may legitimately differ by one Gaussian's contribution (up to ~1/255); parity tests hold the
 * 1e-3 bound on the other pixels and a one-flip bound on these (tests/test_raster_gpu.py).
 * Returns the number of (tile, Gaussian) instances, or -1 on allocation failure.
 */
long oracle_rasterize(int N, int H, int W, const real *means3d, const real *rotations, const real *scales,
                      const real *opacities, const real *colors, const real *view, const real *proj, real tanfovx,
                      real tanfovy, const real *bg, real scale_modifier, int antialiasing, real *out_color,
                      real *out_alpha, real *out_inv_depth, int32_t *out_radii, uint8_t *out_unstable) {
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    const int ntiles = gx * gy;
    const real focal_x = (real)W / ((real)2.0 * tanfovx);
    const real focal_y = (real)H / ((real)2.0 * tanfovy);

    real *xy = (real *)malloc((size_t)N * 2 * sizeof(real));
    real *conic_o = (real *)malloc((size_t)N * 4 * sizeof(real));
    real *depth = (real *)malloc((size_t)N * sizeof(real));
    int *rect = (int *)malloc((size_t)N * 4 * sizeof(int));
    int *tile_count = (int *)calloc((size_t)ntiles + 1, sizeof(int));
    if (!xy || !conic_o || !depth || !rect || !tile_count) return -1;

    /* ---- preprocess ---- */
    for (int i = 0; i < N; ++i) {
        out_radii[i] = 0;
        rect[4 * i] = rect[4 * i + 1] = rect[4 * i + 2] = rect[4 * i + 3] = 0;
        const real *p = means3d + 3 * i;
        real pv[3], ph[4];
        xform4x3(view, p, pv);
        if (pv[2] <= (real)0.2) continue;
        xform4x4(proj, p, ph);
        real pw = (real)1.0 / (ph[3] + (real)0.0000001);
        real ppx = ph[0] * pw, ppy = ph[1] * pw;

        /* Sigma3D = R S^2 R^T from (w,x,y,z) and scale */
        const real *q = rotations + 4 * i;
        real r = q[0], x = q[1], y = q[2], z = q[3];
        real Rm[3][3] = {{(real)1 - (real)2 * (y * y + z * z), (real)2 * (x * y - r * z), (real)2 * (x * z + r * y)},
                         {(real)2 * (x * y + r * z), (real)1 - (real)2 * (x * x + z * z), (real)2 * (y * z - r * x)},
                         {(real)2 * (x * z - r * y), (real)2 * (y * z + r * x), (real)1 - (real)2 * (x * x + y * y)}};
        real s[3] = {scale_modifier * scales[3 * i], scale_modifier * scales[3 * i + 1],
                     scale_modifier * scales[3 * i + 2]};
        /* M = S * R^T (rows of M = scaled columns of R); Sigma = M^T M */
        real M[3][3];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) M[a][b] = s[a] * Rm[b][a];
        real Sg[3][3];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) Sg[a][b] = M[0][a] * M[0][b] + M[1][a] * M[1][b] + M[2][a] * M[2][b];

        /* EWA: cov2D = (J Wv) Sigma (J Wv)^T */
        real t[3] = {pv[0], pv[1], pv[2]};
        real limx = (real)1.3 * tanfovx, limy = (real)1.3 * tanfovy;
        real txtz = t[0] / t[2], tytz = t[1] / t[2];
        t[0] = R_FMIN(limx, R_FMAX(-limx, txtz)) * t[2];
        t[1] = R_FMIN(limy, R_FMAX(-limy, tytz)) * t[2];
        real J0[3] = {focal_x / t[2], (real)0, -(focal_x * t[0]) / (t[2] * t[2])};
        real J1[3] = {(real)0, focal_y / t[2], -(focal_y * t[1]) / (t[2] * t[2])};
        /* Wv = rotation part of the view matrix (row a, col b) = view[b*4 + a] */
        real T0[3], T1[3]; /* rows of J*Wv */
        for (int b = 0; b < 3; ++b) {
            T0[b] = J0[0] * view[b * 4 + 0] + J0[1] * view[b * 4 + 1] + J0[2] * view[b * 4 + 2];
            T1[b] = J1[0] * view[b * 4 + 0] + J1[1] * view[b * 4 + 1] + J1[2] * view[b * 4 + 2];
        }
        real ST0[3], ST1[3];
        for (int a = 0; a < 3; ++a) {
            ST0[a] = Sg[a][0] * T0[0] + Sg[a][1] * T0[1] + Sg[a][2] * T0[2];
            ST1[a] = Sg[a][0] * T1[0] + Sg[a][1] * T1[1] + Sg[a][2] * T1[2];
        }
        real ca = T0[0] * ST0[0] + T0[1] * ST0[1] + T0[2] * ST0[2];
        real cb = T0[0] * ST1[0] + T0[1] * ST1[1] + T0[2] * ST1[2];
        real cc = T1[0] * ST1[0] + T1[1] * ST1[1] + T1[2] * ST1[2];

        const real h_var = (real)0.3;
        real det_cov = ca * cc - cb * cb;
        ca += h_var;
        cc += h_var;
        real det = ca * cc - cb * cb;
        real h_scale = (real)1.0;
        if (antialiasing) h_scale = R_SQRT(R_FMAX((real)0.000025, det_cov / det));
        if (det == (real)0.0) continue;
        real det_inv = (real)1.0 / det;
        real conA = cc * det_inv, conB = -cb * det_inv, conC = ca * det_inv;

        real mid = (real)0.5 * (ca + cc);
        real lam1 = mid + R_SQRT(R_FMAX((real)0.1, mid * mid - det));
        real lam2 = mid - R_SQRT(R_FMAX((real)0.1, mid * mid - det));
        real my_radius = R_CEIL((real)3.0 * R_SQRT(R_FMAX(lam1, lam2)));
        real pix_x = ((ppx + (real)1.0) * (real)W - (real)1.0) * (real)0.5;
        real pix_y = ((ppy + (real)1.0) * (real)H - (real)1.0) * (real)0.5;

        int rx0 = clampi((int)((pix_x - my_radius) / (real)TILE), 0, gx);
        int ry0 = clampi((int)((pix_y - my_radius) / (real)TILE), 0, gy);
        int rx1 = clampi((int)((pix_x + my_radius + (real)(TILE - 1)) / (real)TILE), 0, gx);
        int ry1 = clampi((int)((pix_y + my_radius + (real)(TILE - 1)) / (real)TILE), 0, gy);
        if ((rx1 - rx0) * (ry1 - ry0) == 0) continue;

        depth[i] = pv[2];
        out_radii[i] = (int32_t)my_radius;
        xy[2 * i] = pix_x;
        xy[2 * i + 1] = pix_y;
        conic_o[4 * i] = conA;
        conic_o[4 * i + 1] = conB;
        conic_o[4 * i + 2] = conC;
        conic_o[4 * i + 3] = opacities[i] * h_scale;
        rect[4 * i] = rx0;
        rect[4 * i + 1] = ry0;
        rect[4 * i + 2] = rx1;
        rect[4 * i + 3] = ry1;
        for (int ty = ry0; ty < ry1; ++ty)
            for (int tx = rx0; tx < rx1; ++tx) tile_count[ty * gx + tx + 1]++;
    }

    /* ---- binning: per-tile lists in emit order, then stable sort by depth ---- */
    for (int t = 0; t < ntiles; ++t) tile_count[t + 1] += tile_count[t];
    long total = tile_count[ntiles];
    inst_t *inst = (inst_t *)malloc(((size_t)total + 1) * sizeof(inst_t));
    inst_t *tmp = (inst_t *)malloc(((size_t)total + 1) * sizeof(inst_t));
    int *cursor = (int *)malloc((size_t)ntiles * sizeof(int));
    if (!inst || !tmp || !cursor) return -1;
    memcpy(cursor, tile_count, (size_t)ntiles * sizeof(int));
    for (int i = 0; i < N; ++i) {
        if (out_radii[i] <= 0) continue;
        for (int ty = rect[4 * i + 1]; ty < rect[4 * i + 3]; ++ty)
            for (int tx = rect[4 * i]; tx < rect[4 * i + 2]; ++tx) {
                int k = cursor[ty * gx + tx]++;
                inst[k].depth = depth[i];
                inst[k].id = i;
            }
    }
    for (int t = 0; t < ntiles; ++t) merge_sort(inst + tile_count[t], tmp, tile_count[t + 1] - tile_count[t]);

    /* ---- blend ---- */
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4)
#endif
    for (int t = 0; t < ntiles; ++t) {
        const int tx = t % gx, ty = t / gx;
        const int beg = tile_count[t], end = tile_count[t + 1];
        for (int ly = 0; ly < TILE; ++ly)
            for (int lx = 0; lx < TILE; ++lx) {
                const int px = tx * TILE + lx, py = ty * TILE + ly;
                if (px >= W || py >= H) continue;
                const real pxf = (real)px, pyf = (real)py;
                real T = (real)1.0, C0 = 0, C1 = 0, C2 = 0, D = 0;
                int unstable = 0;
                const real margin = (real)1e-4;
                for (int k = beg; k < end; ++k) {
                    const int id = inst[k].id;
                    real dx = xy[2 * id] - pxf, dy = xy[2 * id + 1] - pyf;
                    const real *co = conic_o + 4 * id;
                    real power = (real)-0.5 * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    if (power > -margin * margin) unstable = 1;
                    if (power > (real)0.0) continue;
                    real alpha = R_FMIN((real)0.99, co[3] * R_EXP(power));
                    if (alpha * (real)255.0 > (real)1.0 - margin && alpha * (real)255.0 < (real)1.0 + margin) unstable = 1;
                    if (alpha < (real)1.0 / (real)255.0) continue;
                    real test_T = T * ((real)1.0 - alpha);
                    if (test_T > (real)0.0001 * ((real)1.0 - margin) && test_T < (real)0.0001 * ((real)1.0 + margin))
                        unstable = 1;
                    if (test_T < (real)0.0001) break; /* pixel done; this Gaussian is not added */
                    C0 += colors[3 * id] * alpha * T;
                    C1 += colors[3 * id + 1] * alpha * T;
                    C2 += colors[3 * id + 2] * alpha * T;
                    D += ((real)1.0 / depth[id]) * alpha * T;
                    T = test_T;
                }
                const size_t pid = (size_t)py * W + px;
                out_color[pid] = C0 + T * bg[0];
                out_color[(size_t)H * W + pid] = C1 + T * bg[1];
                out_color[(size_t)2 * H * W + pid] = C2 + T * bg[2];
                out_alpha[pid] = (real)1.0 - T;
                out_inv_depth[pid] = D;
                if (out_unstable) out_unstable[pid] = (uint8_t)unstable;
            }
    }

    free(xy);
    free(conic_o);
    free(depth);
    free(rect);
    free(tile_count);
    free(inst);
    free(tmp);
    free(cursor);
    return total;
}

int oracle_real_bytes(void) { return (int)sizeof(real); }
