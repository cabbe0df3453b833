// Stage-1 identity encoder: the point <-> triplane-cell reductions of
// SMPLXTriplaneEncoder (src/models/triplane_net.py:124-207,226-244) and the point -> pixel feature lookup of
// points_projection (src/utils/graphic_utils.py:275-331), deterministic.
//
// The reference scatters with torch_scatter (scatter_max / scatter_mean: atomics, sum order undefined) and with an
// index_put that has duplicate indices (which pixel's feature a point receives is undefined on a GPU).  Here every
// reduction is a SEGMENT reduce over points pre-sorted by cell (stable sort: ascending point id inside a cell), so
// sums are evaluated in a fixed order and the result is reproducible bit for bit.
//
//   amav_cell_max     cellmax[b][plane][cell][c] = max over the cell's points of feat[b][point][c]   (pool_local, 1/2)
//   amav_cell_gather  out[b][n][c] = sum over the three planes of cellmax[b][plane][cell_of(n)][c]   (pool_local, 2/2)
//   amav_cell_mean    plane[b][c][cell] = mean over the cell's points of feat[b][point][c], 0 if empty
//                     (generate_plane_features; [B,C,R*R] is the reference's layout, reshaped to [B,C,R,R])
//   amav_points_project  nearest-point-per-pixel z-buffer over discs of `radius_px`, then every point that is the
//                     nearest one somewhere takes the feature of the LAST such pixel in (y, x) order (what a sequential
//                     index_put does), all other points get zeros.
#include "amav_common.h"

namespace amav {
namespace splat {

// one block per (cell, plane, b); thread = channel (strided).  order: point ids sorted by cell, seg: [cells + 1] offsets
__global__ __launch_bounds__(256) void cell_max_kernel(int N, int C, int cells, const float *__restrict__ feat,
                                                       const int *__restrict__ order, const int *__restrict__ seg,
                                                       float *__restrict__ cellmax) {
    const int cell = blockIdx.x, plane = blockIdx.y, b = blockIdx.z;
    const size_t pb = (size_t)b * 3 + plane;
    const int *sg = seg + pb * (cells + 1);
    const int beg = sg[cell], end = sg[cell + 1];
    const int *ord = order + pb * N;
    const float *fb = feat + (size_t)b * N * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float m = 0.0f;  // torch_scatter: a cell without points reads 0
        if (end > beg) {
            m = fb[(size_t)ord[beg] * C + c];
            for (int k = beg + 1; k < end; ++k) m = fmaxf(m, fb[(size_t)ord[k] * C + c]);
        }
        cellmax[(pb * cells + cell) * C + c] = m;
    }
}

// one block per (point tile, b): out[b][n][c] = sum_plane cellmax[b][plane][cell[b][plane][n]][c] in plane order 0,1,2
__global__ __launch_bounds__(256) void cell_gather_kernel(int N, int C, int cells, const float *__restrict__ cellmax,
                                                          const int *__restrict__ cell_of, float *__restrict__ out) {
    const int b = blockIdx.y;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (n >= N) return;
    const int c0 = cell_of[((size_t)b * 3 + 0) * N + n], c1 = cell_of[((size_t)b * 3 + 1) * N + n],
              c2 = cell_of[((size_t)b * 3 + 2) * N + n];
    const float *m0 = cellmax + (((size_t)b * 3 + 0) * cells + c0) * C;
    const float *m1 = cellmax + (((size_t)b * 3 + 1) * cells + c1) * C;
    const float *m2 = cellmax + (((size_t)b * 3 + 2) * cells + c2) * C;
    float *o = out + ((size_t)b * N + n) * C;
    for (int c = lane; c < C; c += 64) o[c] = (m0[c] + m1[c]) + m2[c];  // c_out = 0 + xy + xz + yz (triplane_net.py:236-243)
}

// one block per (cell, b) of ONE plane's index set; out[b][c][cell] (the reference's [B, C, R*R])
__global__ __launch_bounds__(256) void cell_mean_kernel(int N, int C, int cells, const float *__restrict__ feat,
                                                        const int *__restrict__ order, const int *__restrict__ seg,
                                                        float *__restrict__ out) {
    const int cell = blockIdx.x, b = blockIdx.y;
    const int *sg = seg + (size_t)b * (cells + 1);
    const int beg = sg[cell], end = sg[cell + 1];
    const int *ord = order + (size_t)b * N;
    const float *fb = feat + (size_t)b * N * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s = 0.0f;
        for (int k = beg; k < end; ++k) s += fb[(size_t)ord[k] * C + c];  // ascending point id: fixed order
        out[((size_t)b * C + c) * cells + cell] = end > beg ? s / (float)(end - beg) : 0.0f;
    }
}

// ---- points_projection ------------------------------------------------------------------------------------------
// pass 1: one thread per point: project (OpenCV convention: x_cam = R p + t, u = fx X/Z + cx, v = fy Y/Z + cy) and
// atomicMin (depth bits << 32 | point id) into every pixel whose CENTRE lies within radius_px of (u, v).
__global__ __launch_bounds__(256) void project_zbuffer_kernel(int N, int H, int W, const float *__restrict__ points,
                                                              const float *__restrict__ w2c, const float *__restrict__ K,
                                                              float radius_px, unsigned long long *__restrict__ zbuf) {
#pragma clang fp contract(off)  // plain fp32 operations in source order: the disc membership tests are discontinuous
    const int n = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (n >= N) return;
    const float *p = points + ((size_t)b * N + n) * 3;
    const float *E = w2c + (size_t)b * 16, *k = K + (size_t)b * 9;
    const float X = E[0] * p[0] + E[1] * p[1] + E[2] * p[2] + E[3];
    const float Y = E[4] * p[0] + E[5] * p[1] + E[6] * p[2] + E[7];
    const float Z = E[8] * p[0] + E[9] * p[1] + E[10] * p[2] + E[11];
    if (!(Z > 0.0f)) return;  // behind the camera: never visible (pytorch3d keeps z >= 0)
    const float u = k[0] * X / Z + k[2], v = k[4] * Y / Z + k[5];
    const int x0 = max(0, (int)ceilf(u - radius_px - 0.5f)), x1 = min(W - 1, (int)floorf(u + radius_px - 0.5f));
    const int y0 = max(0, (int)ceilf(v - radius_px - 0.5f)), y1 = min(H - 1, (int)floorf(v + radius_px - 0.5f));
    const unsigned long long key = ((unsigned long long)__float_as_uint(Z) << 32) | (unsigned)n;
    const float r2 = radius_px * radius_px;
    for (int y = y0; y <= y1; ++y)
        for (int x = x0; x <= x1; ++x) {
            const float dx = (float)x + 0.5f - u, dy = (float)y + 0.5f - v;
            if (dx * dx + dy * dy < r2) atomicMin(&zbuf[((size_t)b * H + y) * W + x], key);
        }
}

// pass 2: one thread per pixel: the winner of this pixel remembers the LARGEST pixel index it wins
__global__ __launch_bounds__(256) void project_claim_kernel(int N, int H, int W, const unsigned long long *__restrict__ zbuf,
                                                            int *__restrict__ pixel_of) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (i >= H * W) return;
    const unsigned long long key = zbuf[(size_t)b * H * W + i];
    if (key == ~0ull) return;
    atomicMax(&pixel_of[(size_t)b * N + (unsigned)key], i);
}

// pass 3: one wave per point: copy the claimed pixel's C features (features are [B, C, H, W] as the reference holds them)
__global__ __launch_bounds__(256) void project_fetch_kernel(int N, int C, int H, int W, const float *__restrict__ feat,
                                                            const int *__restrict__ pixel_of, float *__restrict__ out) {
    const int b = blockIdx.y;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    const int px = pixel_of[(size_t)b * N + n];
    float *o = out + ((size_t)b * N + n) * C;
    const float *f = feat + (size_t)b * C * H * W + (px >= 0 ? px : 0);
    for (int c = lane; c < C; c += 64) o[c] = px >= 0 ? f[(size_t)c * H * W] : 0.0f;
}

__global__ __launch_bounds__(256) void fill_u64_kernel(size_t n, unsigned long long v, unsigned long long *p) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ __launch_bounds__(256) void fill_i32_kernel(size_t n, int v, int *p) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

}  // namespace splat
}  // namespace amav

using namespace amav;
using namespace amav::splat;

extern "C" int amav_cell_max(int B, int N, int C, int cells, const float *feat, const int32_t *order,
                             const int32_t *seg, float *cellmax, void *stream) {
    AMAV_REQUIRE(B > 0 && N > 0 && C > 0 && cells > 0 && cells <= 65535 * 16 && B <= 65535, "amav_cell_max: bad sizes");
    AMAV_REQUIRE(feat && order && seg && cellmax, "amav_cell_max: NULL pointer");
    cell_max_kernel<<<dim3((unsigned)cells, 3, (unsigned)B), 256, 0, static_cast<hipStream_t>(stream)>>>(
        N, C, cells, feat, order, seg, cellmax);
    return check_launch("amav_cell_max");
}

extern "C" int amav_cell_gather(int B, int N, int C, int cells, const float *cellmax, const int32_t *cell_of,
                                float *out, void *stream) {
    AMAV_REQUIRE(B > 0 && N > 0 && C > 0 && cells > 0 && B <= 65535, "amav_cell_gather: bad sizes");
    AMAV_REQUIRE(cellmax && cell_of && out, "amav_cell_gather: NULL pointer");
    cell_gather_kernel<<<dim3((unsigned)((N + 3) / 4), (unsigned)B), 256, 0, static_cast<hipStream_t>(stream)>>>(
        N, C, cells, cellmax, cell_of, out);
    return check_launch("amav_cell_gather");
}

extern "C" int amav_cell_mean(int B, int N, int C, int cells, const float *feat, const int32_t *order,
                              const int32_t *seg, float *out_planes, void *stream) {
    AMAV_REQUIRE(B > 0 && N > 0 && C > 0 && cells > 0 && B <= 65535, "amav_cell_mean: bad sizes");
    AMAV_REQUIRE(feat && order && seg && out_planes, "amav_cell_mean: NULL pointer");
    cell_mean_kernel<<<dim3((unsigned)cells, (unsigned)B), 256, 0, static_cast<hipStream_t>(stream)>>>(N, C, cells, feat,
                                                                                                     order, seg, out_planes);
    return check_launch("amav_cell_mean");
}

extern "C" size_t amav_points_project_workspace_bytes(int B, int N, int H, int W) {
    if (B <= 0 || N <= 0 || H <= 0 || W <= 0) return 0;
    return align_up((size_t)B * H * W * sizeof(unsigned long long), 256) + align_up((size_t)B * N * sizeof(int), 256);
}

extern "C" int amav_points_project(int B, int N, int C, int H, int W, const float *points, const float *w2c,
                                   const float *intrinsics, const float *features, float radius_px, float *out,
                                   void *workspace, size_t workspace_bytes, void *stream_) {
    AMAV_REQUIRE(B > 0 && N > 0 && C > 0 && H > 0 && W > 0 && B <= 65535 && radius_px > 0.0f,
                 "amav_points_project: bad sizes");
    AMAV_REQUIRE((long long)H * W < (1ll << 31), "amav_points_project: image too large");
    AMAV_REQUIRE(points && w2c && intrinsics && features && out && workspace, "amav_points_project: NULL pointer");
    const size_t need = amav_points_project_workspace_bytes(B, N, H, W);
    if (workspace_bytes < need)
        return fail(AMAV_ERR_WORKSPACE, "amav_points_project: workspace %zu < required %zu", workspace_bytes, need);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    Carver c(workspace);
    unsigned long long *zbuf = c.take<unsigned long long>((size_t)B * H * W);
    int *pixel_of = c.take<int>((size_t)B * N);
    const size_t npix = (size_t)B * H * W, npts = (size_t)B * N;
    fill_u64_kernel<<<(unsigned)((npix + 255) / 256), 256, 0, stream>>>(npix, ~0ull, zbuf);
    fill_i32_kernel<<<(unsigned)((npts + 255) / 256), 256, 0, stream>>>(npts, -1, pixel_of);
    project_zbuffer_kernel<<<dim3((unsigned)((N + 255) / 256), (unsigned)B), 256, 0, stream>>>(N, H, W, points, w2c,
                                                                                              intrinsics, radius_px, zbuf);
    project_claim_kernel<<<dim3((unsigned)(((size_t)H * W + 255) / 256), (unsigned)B), 256, 0, stream>>>(N, H, W, zbuf,
                                                                                                         pixel_of);
    project_fetch_kernel<<<dim3((unsigned)((N + 3) / 4), (unsigned)B), 256, 0, stream>>>(N, C, H, W, features, pixel_of,
                                                                                        out);
    return check_launch("amav_points_project");
}
