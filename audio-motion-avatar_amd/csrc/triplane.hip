// Triplane decode for gfx950: grid_sample x 3 planes + five linear Gaussian heads + construct_gaussians.
//
// Replaces src/models/renderer.py:136,158 (F.grid_sample, bilinear, align_corners=False, zero padding, via
// sample_from_triplane :292-317), :165-171 (xyz/rotation/scaling/opacity/shs heads on [p; features]) and :319-346
// (construct_gaussians).  The reference materialises [N, 3C] features (30.7 MB per frame at C=256, N=10k) from an
// NCHW gather that touches C strided addresses per tap.  Here the two linear maps are swapped:
//
//   project_kernel        G[plane][texel][16] = W_plane[16 x C] . slab[:, texel]   -- streams the frame's token slab
//                         [C][3 R^2] exactly once, fully coalesced (16 B per lane along the texel axis, two register
//                         buffers so the stream never drains: 5.6 TB/s), with the plane's weights broadcast from LDS.
//                         This is the HBM-bound kernel of the stage.
//   sample_decode_kernel  4 lanes per point, each lane owns 4 of the 16 projected channels: 12 taps x 16 B loads
//                         from the L2-resident projected planes (issued back to back, zero-weight taps for the
//                         padding), + W_xyz p + bias, then the per-head epilogue (normalise rot, sigmoid colour,
//                         xyz + offset + transl) and one packed 64-byte record.  The indexed form gathers the
//                         point from the posed vertices through the baked subdivision table (one base vertex per
//                         lane of the quad, exchanged by DPP).
//
// Same real-number function as the reference; rounding differs only by summation order.
#include <cstdlib>

#include "amav_common.h"

namespace amav {
namespace triplane {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// grid: (ceil(R*R/4 / 256), 3, F); each thread projects 4 consecutive texels of one plane.  The plane's [C][16] weight
// block is staged in LDS once per workgroup and read back as broadcasts (as scalar loads its 16 KB per plane thrash
// the 16 KB scalar cache when workgroups of different planes share a CU, putting an L2 round trip in every iteration).
// The texels of plane `plane` that bilinear taps of points inside the box [lo, hi] (world space) can touch, with
// sample_decode_kernel's own arithmetic (u = clamp(p / radius), pixel = ((u + 1) R - 1) / 2, taps floor and floor + 1,
// out-of-range taps read the clamped address): every step is monotonic in p, so the taps of any point of the box --
// in particular of any point the subdivision table averages from vertices inside it -- lie in the returned rectangle
// [x0, x1] x [y0, y1] (inclusive, already clamped to the plane).
struct TexelRect {
    int x0, x1, y0, y1;
};
__device__ __forceinline__ int tap_floor(float p, float radius, int R) {
    const float u = fminf(fmaxf(p / radius, -1.0f), 1.0f);
    return (int)floorf(((u + 1.0f) * (float)R - 1.0f) * 0.5f);
}
__device__ __forceinline__ TexelRect region_of(const float *__restrict__ box, int plane, float radius, int R) {
    // plane 0 <- (x, y), plane 1 <- (x, z), plane 2 <- (y, z); grid x indexes W, grid y indexes H
    const int ax = plane == 2 ? 1 : 0, ay = plane == 0 ? 1 : 2;
    TexelRect r;
    r.x0 = min(max(tap_floor(box[ax], radius, R), 0), R - 1);
    r.x1 = min(max(tap_floor(box[3 + ax], radius, R) + 1, 0), R - 1);
    r.y0 = min(max(tap_floor(box[ay], radius, R), 0), R - 1);
    r.y1 = min(max(tap_floor(box[3 + ay], radius, R) + 1, 0), R - 1);
    return r;
}

// `boxes` (NULL = all texels): per frame the bounding box {min xyz, max xyz} of the points that will be sampled
// (amav_points_bbox); texel quads outside the frame's region_of() are neither read nor written -- the body covers a
// fifth to a third of each plane, and the slab is the largest stream of the whole path.
template <int kUnroll, bool kNT>
__global__ __launch_bounds__(256) void project_kernel(int C, int RR, const float *__restrict__ tokens,
                                                      long long frame_stride, const float *__restrict__ wplane,
                                                      float *__restrict__ out, const float *__restrict__ boxes,
                                                      float radius, int R) {
    extern __shared__ __align__(16) float w_lds[];  // [C][16]
    // grid = (3 planes, F, blocks per plane): the block index inside the plane is the SLOWEST dimension, because with a
    // region only the first few blocks of every plane have work and workgroups go to the 8 XCDs round-robin by their
    // linear id -- as the fastest dimension the busy blocks of all planes would share a few XCDs
    const int bx = blockIdx.z;
    int q = bx * blockDim.x + threadIdx.x;  // texel quad within the plane
    const int plane = blockIdx.x, f = blockIdx.y;
    if (boxes) {
        // the quads of the frame's rectangle, numbered row by row: thread group i of the plane's grid takes the i-th
        // of them (full waves whatever the rectangle's width; the blocks past its last quad leave before the barrier)
        const TexelRect rc = region_of(boxes + (size_t)f * 6, plane, radius, R);
        const int c0 = rc.x0 >> 2, per_row = (rc.x1 >> 2) - c0 + 1, rows = rc.y1 - rc.y0 + 1;
        if (bx * (int)blockDim.x >= per_row * rows) return;  // block-uniform
        const int row = q / per_row;
        q = row < rows ? (rc.y0 + row) * (R >> 2) + c0 + (q - row * per_row) : RR;  // RR: past the end, leaves below
    }
    {
        const float4 *src4 = reinterpret_cast<const float4 *>(wplane + (size_t)plane * C * 16);
        float4 *dst4 = reinterpret_cast<float4 *>(w_lds);
        for (int i = threadIdx.x; i < C * 4; i += blockDim.x) dst4[i] = src4[i];
    }
    __syncthreads();
    if (q * 4 >= RR) return;
    const int S = 3 * RR;
    const float *src = tokens + (size_t)f * frame_stride + (size_t)plane * RR + (size_t)q * 4;
    float acc[4][16];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int o = 0; o < 16; ++o) acc[t][o] = 0.0f;
    // read-once stream: non-temporal so the slab does not evict the projected planes / weights from L2.  Two register
    // buffers of kUnroll channels: the loads of the next group are issued before the current group's FMAs, so a wave
    // always has kUnroll..2*kUnroll kilobytes in flight (a single unrolled loop drains to zero at every trip).
    auto load = [&](int c) {
        return kNT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(src + (size_t)c * S))
                   : *reinterpret_cast<const f32x4 *>(src + (size_t)c * S);
    };
    auto fma_channel = [&](int c, const f32x4 &x) {
        const float4 *wc = reinterpret_cast<const float4 *>(w_lds + c * 16);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 ww = wc[g];
            const float wv[4] = {ww.x, ww.y, ww.z, ww.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[0][g * 4 + e] += wv[e] * x.x;
                acc[1][g * 4 + e] += wv[e] * x.y;
                acc[2][g * 4 + e] += wv[e] * x.z;
                acc[3][g * 4 + e] += wv[e] * x.w;
            }
        }
    };
    f32x4 xa[kUnroll], xb[kUnroll];
    const int groups = C / kUnroll;  // full groups; the tail is handled below
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) xa[u] = load(min(u, C - 1));
    for (int g0 = 0; g0 < groups; g0 += 2) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) xb[u] = load(min((g0 + 1) * kUnroll + u, C - 1));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) fma_channel(g0 * kUnroll + u, xa[u]);
        __builtin_amdgcn_sched_barrier(0);
        if (g0 + 1 >= groups) break;
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) xa[u] = load(min((g0 + 2) * kUnroll + u, C - 1));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) fma_channel((g0 + 1) * kUnroll + u, xb[u]);
        __builtin_amdgcn_sched_barrier(0);
    }
    for (int c = groups * kUnroll; c < C; ++c) fma_channel(c, load(c));
    float4 *dst = reinterpret_cast<float4 *>(out + (((size_t)f * 3 + plane) * RR + (size_t)q * 4) * 16);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            dst[t * 4 + g] = make_float4(acc[t][g * 4], acc[t][g * 4 + 1], acc[t][g * 4 + 2], acc[t][g * 4 + 3]);
}

// scalar fallback for R*R not a multiple of 4 or unaligned slabs: one texel per thread
__global__ __launch_bounds__(256) void project_kernel_scalar(int C, int RR, const float *__restrict__ tokens,
                                                             long long frame_stride,
                                                             const float *__restrict__ wplane,
                                                             float *__restrict__ out) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    const int plane = blockIdx.y, f = blockIdx.z;
    if (s >= RR) return;
    const int S = 3 * RR;
    const float *src = tokens + (size_t)f * frame_stride + (size_t)plane * RR + s;
    const float *w = wplane + (size_t)plane * C * 16;
    float acc[16];
#pragma unroll
    for (int o = 0; o < 16; ++o) acc[o] = 0.0f;
    for (int c = 0; c < C; ++c) {
        const float x = src[(size_t)c * S];
#pragma unroll
        for (int o = 0; o < 16; ++o) acc[o] += w[c * 16 + o] * x;
    }
    float *dst = out + (((size_t)f * 3 + plane) * RR + s) * 16;
#pragma unroll
    for (int o = 0; o < 16; ++o) dst[o] = acc[o];
}

// torch grid_sampler, bilinear, align_corners=False, padding zeros: pixel = ((g + 1) * size - 1) / 2
struct Taps {
    int ix0, iy0;
    float wx0, wx1, wy0, wy1;
};

__device__ __forceinline__ Taps make_taps(float gx, float gy, int R) {
    const float ix = ((gx + 1.0f) * (float)R - 1.0f) * 0.5f;
    const float iy = ((gy + 1.0f) * (float)R - 1.0f) * 0.5f;
    const float fx = floorf(ix), fy = floorf(iy);
    Taps t;
    t.ix0 = (int)fx, t.iy0 = (int)fy;
    t.wx1 = ix - fx, t.wx0 = (fx + 1.0f) - ix;
    t.wy1 = iy - fy, t.wy0 = (fy + 1.0f) - iy;
    return t;
}

// Lane group of 4 per point; lane q owns projected channels 4q..4q+3:
//   q0 = (xyz_offset, opacity), q1 = rotation, q2 = (scaling, pad), q3 = (shs, pad)
// kIndexed: the point is gathered from the posed vertices through the baked subdivision table (lbs.hip gather_kernel
// fused in: 1/2 (1/2 (v[a0]+v[b0]) + 1/2 (v[a1]+v[b1])), the same operation order, so the same bits).
template <int K>
__device__ __forceinline__ float quad_bcast(float v) {  // value of lane K of this lane's quad
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), K * 0x55, 0xf, 0xf, true));
}

template <int K>
__device__ __forceinline__ int quad_bcast_i(int v) {
    return __builtin_amdgcn_mov_dpp(v, K * 0x55, 0xf, 0xf, true);
}

template <bool kIndexed>
__global__ __launch_bounds__(256) void sample_decode_kernel(int F, int N, int R, int V,
                                                            const float *__restrict__ proj,
                                                            const float *__restrict__ points,
                                                            const int4 *__restrict__ idx4,
                                                            const float *__restrict__ transl, float radius,
                                                            const float *__restrict__ wpoint,
                                                            float *__restrict__ out) {
    // 1-D grid, frames grouped per XCD: blocks are dealt round-robin over the 8 XCDs, so block b serves frame
    // (b/8 / bpf) * 8 + b%8 and one XCD's L2 keeps a frame's 196 KB of projected planes to itself (speed only)
    const int bpf = (4 * N + 255) / 256;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int f = (j / bpf) * 8 + xcd;
    if (f >= F) return;
    const int gid = (j % bpf) * blockDim.x + threadIdx.x;
    const int n = gid >> 2, q = gid & 3;
    if (n >= N) return;
    float p0, p1, p2;
    if (kIndexed) {
        // the four lanes of a point fetch one base vertex each and trade them inside the quad (DPP), instead of
        // every lane gathering all four
        const int4 id = idx4[n];
        const int mine = q == 0 ? id.x : (q == 1 ? id.y : (q == 2 ? id.z : id.w));
        const float *vp = points + ((size_t)f * V + mine) * 3;
        const float v0 = vp[0], v1 = vp[1], v2 = vp[2];
        p0 = ((quad_bcast<0>(v0) + quad_bcast<1>(v0)) * 0.5f + (quad_bcast<2>(v0) + quad_bcast<3>(v0)) * 0.5f) * 0.5f;
        p1 = ((quad_bcast<0>(v1) + quad_bcast<1>(v1)) * 0.5f + (quad_bcast<2>(v1) + quad_bcast<3>(v1)) * 0.5f) * 0.5f;
        p2 = ((quad_bcast<0>(v2) + quad_bcast<1>(v2)) * 0.5f + (quad_bcast<2>(v2) + quad_bcast<3>(v2)) * 0.5f) * 0.5f;
    } else {
        const float *pp = points + ((size_t)f * N + n) * 3;
        p0 = pp[0], p1 = pp[1], p2 = pp[2];
    }
    const float u0 = fminf(fmaxf(p0 / radius, -1.0f), 1.0f);  // IEEE division, as torch: the taps depend on it
    const float u1 = fminf(fmaxf(p1 / radius, -1.0f), 1.0f);
    const float u2 = fminf(fmaxf(p2 / radius, -1.0f), 1.0f);
    const int RR = R * R;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // The four lanes of a point need the same twelve taps.  Lane q < 3 works out plane q's four (texel, weight)
    // pairs -- branch-free: an out-of-range texel (zero padding) is a clamped address with weight 0 -- and the quad
    // trades them by DPP; then every lane issues its twelve 16-byte loads back to back.
    // plane 0 <- (x, y), plane 1 <- (x, z), plane 2 <- (y, z); grid x indexes W, grid y indexes H
    int my_off[4];
    float my_w[4];
    {
        const Taps t = make_taps(q == 2 ? u1 : u0, q == 0 ? u1 : u2, R);
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int ix = t.ix0 + dx, iy = t.iy0 + dy;
                const bool in = ix >= 0 && ix < R && iy >= 0 && iy < R;
                const int cx = min(max(ix, 0), R - 1), cy = min(max(iy, 0), R - 1);
                my_w[dy * 2 + dx] = in ? (dx ? t.wx1 : t.wx0) * (dy ? t.wy1 : t.wy0) : 0.0f;
                my_off[dy * 2 + dx] = (cy * R + cx) * 4;  // in float4 units
            }
    }
    float4 tv[12];
    float tw[12];
    const float4 *pl0 = reinterpret_cast<const float4 *>(proj + ((size_t)f * 3 * RR) * 16) + q;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        tw[k] = quad_bcast<0>(my_w[k]), tw[4 + k] = quad_bcast<1>(my_w[k]), tw[8 + k] = quad_bcast<2>(my_w[k]);
        tv[k] = pl0[quad_bcast_i<0>(my_off[k])];
        tv[4 + k] = pl0[(size_t)RR * 4 + quad_bcast_i<1>(my_off[k])];
        tv[8 + k] = pl0[(size_t)RR * 8 + quad_bcast_i<2>(my_off[k])];
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) {  // plane, then dy, then dx
        const float w = tw[k];
        const float4 v = tv[k];
        acc.x += w * v.x, acc.y += w * v.y, acc.z += w * v.z, acc.w += w * v.w;
    }
    // + W_xyz p + bias  (wpoint [16][4]: 3 xyz weights, bias)
    const float4 *wp = reinterpret_cast<const float4 *>(wpoint) + q * 4;
    const float4 w0 = wp[0], w1 = wp[1], w2 = wp[2], w3 = wp[3];
    acc.x += w0.x * p0 + w0.y * p1 + w0.z * p2 + w0.w;
    acc.y += w1.x * p0 + w1.y * p1 + w1.z * p2 + w1.w;
    acc.z += w2.x * p0 + w2.y * p1 + w2.z * p2 + w2.w;
    acc.w += w3.x * p0 + w3.y * p1 + w3.z * p2 + w3.w;

    float4 rec;
    if (q == 0) {
        float tx = 0.f, ty = 0.f, tz = 0.f;
        if (transl) tx = transl[f * 3], ty = transl[f * 3 + 1], tz = transl[f * 3 + 2];
        rec = make_float4(p0 + acc.x + tx, p1 + acc.y + ty, p2 + acc.z + tz, acc.w);
    } else if (q == 1) {
        // F.normalize(dim=-1): v / max(||v||, 1e-12).  Hardware sqrt / reciprocal (1 ulp): the four lane roles of a
        // quad run one after the other, so the IEEE division and exp sequences were a third of the kernel's issue slots
        const float nrm = fmaxf(__fsqrt_rn(acc.x * acc.x + acc.y * acc.y + acc.z * acc.z + acc.w * acc.w), 1e-12f);
        const float inv = __frcp_rn(nrm);
        rec = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
    } else if (q == 2) {
        rec = make_float4(acc.x, acc.y, acc.z, 0.0f);
    } else {
        rec = make_float4(__frcp_rn(1.0f + __expf(-acc.x)), __frcp_rn(1.0f + __expf(-acc.y)),
                          __frcp_rn(1.0f + __expf(-acc.z)), 0.0f);
    }
    reinterpret_cast<float4 *>(out)[((size_t)f * N + n) * 4 + q] = rec;
}

// Plain sample_from_triplane: grid (N, F), lanes along the 3C output channels (coalesced writes).
__global__ __launch_bounds__(256) void sample_features_kernel(int N, int C, int R, const float *__restrict__ planes,
                                                              long long frame_stride, long long plane_stride,
                                                              long long chan_stride, const float *__restrict__ points,
                                                              float radius,
                                                              float *__restrict__ out) {
    const int n = blockIdx.x, f = blockIdx.y;
    const float *pp = points + ((size_t)f * N + n) * 3;
    const float u0 = fminf(fmaxf(pp[0] / radius, -1.0f), 1.0f);
    const float u1 = fminf(fmaxf(pp[1] / radius, -1.0f), 1.0f);
    const float u2 = fminf(fmaxf(pp[2] / radius, -1.0f), 1.0f);
    for (int oc = threadIdx.x; oc < 3 * C; oc += blockDim.x) {
        const int plane = oc / C, c = oc - plane * C;
        const float gx = plane == 2 ? u1 : u0;
        const float gy = plane == 0 ? u1 : u2;
        const Taps t = make_taps(gx, gy, R);
        const float *pl = planes + (size_t)f * frame_stride + (size_t)c * chan_stride + (size_t)plane * plane_stride;
        float acc = 0.0f;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int ix = t.ix0 + dx, iy = t.iy0 + dy;
                if (ix >= 0 && ix < R && iy >= 0 && iy < R)
                    acc += pl[iy * R + ix] * ((dx ? t.wx1 : t.wx0) * (dy ? t.wy1 : t.wy0));
            }
        out[((size_t)f * N + n) * 3 * C + oc] = acc;
    }
}

// The same for C % 64 == 0 (the point refiner's first sampling, 768 channels per point): a block is 64 points x 64
// channels of one plane.  READ phase: lanes run along the POINTS, one channel image at a time, so a wave's 256 taps
// fall into one 4 KB channel image (R = 32) instead of 64 images 12 KB apart; the 64 x 64 results go through LDS and
// the WRITE phase runs lanes along the channels (256-byte rows of the [F, N, 3C] output).  Same arithmetic per
// (point, channel) as the kernel above: bit-identical results.
__global__ __launch_bounds__(256) void sample_features_tiled_kernel(int N, int C, int R, const float *__restrict__ planes,
                                                                    long long frame_stride, long long plane_stride,
                                                                    long long chan_stride,
                                                                    const float *__restrict__ points, float radius,
                                                                    float *__restrict__ out) {
    __shared__ float tile[64 * 65];  // [channel][point]
    const int f = blockIdx.z, oc0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int plane = oc0 / C, c0 = oc0 - plane * C;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = min(n0 + lane, N - 1);
    const float *pp = points + ((size_t)f * N + n) * 3;
    const float u0 = fminf(fmaxf(pp[0] / radius, -1.0f), 1.0f);
    const float u1 = fminf(fmaxf(pp[1] / radius, -1.0f), 1.0f);
    const float u2 = fminf(fmaxf(pp[2] / radius, -1.0f), 1.0f);
    const Taps t = make_taps(plane == 2 ? u1 : u0, plane == 0 ? u1 : u2, R);
    int off[4];
    float wt[4];
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const int ix = t.ix0 + dx, iy = t.iy0 + dy;
            const bool in = ix >= 0 && ix < R && iy >= 0 && iy < R;
            off[2 * dy + dx] = in ? iy * R + ix : -1;
            wt[2 * dy + dx] = (dx ? t.wx1 : t.wx0) * (dy ? t.wy1 : t.wy0);
        }
    const float *pl = planes + (size_t)f * frame_stride + (size_t)plane * plane_stride + (size_t)c0 * chan_stride;
    for (int k = wave; k < 64; k += 4) {
        const float *img = pl + (size_t)k * chan_stride;
        float acc = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (off[q] >= 0) acc += img[off[q]] * wt[q];
        tile[k * 65 + lane] = acc;
    }
    __syncthreads();
    for (int p = wave; p < 64; p += 4) {
        if (n0 + p < N) out[((size_t)f * N + n0 + p) * 3 * C + oc0 + lane] = tile[lane * 65 + p];
    }
}

// One 1024-thread block per frame: {min x, y, z, max x, y, z} of the frame's points, read as a flat run of 3N floats
// (element e is component e % 3).  A frame that holds a NaN gets the infinite box (its sampling coordinates clamp to a
// plane border the finite points may not reach).
__global__ __launch_bounds__(1024) void bbox_kernel(int N, const float *__restrict__ points, float *__restrict__ boxes) {
    __shared__ float part[16][7];
    const int f = blockIdx.x;
    const float *pp = points + (size_t)f * N * 3;
    const int total = 3 * N;
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    float bad = 0.f;
    // three consecutive elements per thread and trip = one point when the frame starts on a point boundary (it does):
    // element e0 + d is component d
    for (int e0 = threadIdx.x * 3; e0 < total; e0 += blockDim.x * 3) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float v = pp[e0 + d];
            lo[d] = fminf(lo[d], v), hi[d] = fmaxf(hi[d], v);
            if (v != v) bad = 1.f;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            lo[d] = fminf(lo[d], __shfl_xor(lo[d], o, 64));
            hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], o, 64));
        }
        bad = fmaxf(bad, __shfl_xor(bad, o, 64));
    }
    const int wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) part[wave][d] = lo[d], part[wave][3 + d] = hi[d];
        part[wave][6] = bad;
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int d = threadIdx.x;
        float any_bad = 0.f, v = part[0][d];
        for (int w = 0; w < nwaves; ++w) {
            any_bad += part[w][6];
            v = d < 3 ? fminf(v, part[w][d]) : fmaxf(v, part[w][d]);
        }
        if (any_bad > 0.f) v = d < 3 ? -__builtin_inff() : __builtin_inff();
        boxes[(size_t)f * 6 + d] = v;
    }
}

}  // namespace triplane
}  // namespace amav

using namespace amav;
using namespace amav::triplane;

extern "C" int amav_points_bbox(int F, int N, const float *points, float *out_boxes, void *stream) {
    AMAV_REQUIRE(F > 0 && N > 0 && points && out_boxes, "amav_points_bbox: bad sizes F=%d N=%d or NULL pointer", F, N);
    bbox_kernel<<<F, 1024, 0, static_cast<hipStream_t>(stream)>>>(N, points, out_boxes);
    return check_launch("amav_points_bbox");
}

extern "C" int amav_triplane_project(int F, int C, int R, const float *tokens, int64_t frame_stride,
                                     const float *wplane, float *out, void *stream) {
    return amav_triplane_project_region(F, C, R, tokens, frame_stride, wplane, out, nullptr, 1.0f, stream);
}

extern "C" int amav_triplane_project_region(int F, int C, int R, const float *tokens, int64_t frame_stride,
                                            const float *wplane, float *out, const float *boxes, float radius,
                                            void *stream_) {
    AMAV_REQUIRE(boxes == nullptr || radius > 0.0f, "amav_triplane_project: radius must be positive");
    AMAV_REQUIRE(F > 0 && C > 0 && R > 0, "amav_triplane_project: bad sizes F=%d C=%d R=%d", F, C, R);
    AMAV_REQUIRE(tokens && wplane && out, "amav_triplane_project: NULL pointer");
    AMAV_REQUIRE(F <= 65535, "amav_triplane_project: F=%d exceeds grid.z", F);
    AMAV_REQUIRE(((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(wplane)) & 15) == 0,
                 "amav_triplane_project: out / head_w_plane not 16-B aligned");
    AMAV_REQUIRE((size_t)C * 64 <= 64 * 1024, "amav_triplane_project: C=%d needs more than 64 KiB of LDS for the weights", C);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int RR = R * R;
    const bool vec = (RR % 4 == 0) && (frame_stride % 4 == 0) && ((reinterpret_cast<uintptr_t>(tokens) & 15) == 0);
    if (R % 4 != 0) boxes = nullptr;  // the rectangle is walked in quads of one row; other resolutions project everything
    if (vec) {
        // 786 MB slab: 140 us (5.6 TB/s) with the double-buffered loads at 4 channels per buffer (8: 149 us, 2: 145 us); the
        // single unrolled loop it replaces drained its loads at every trip and stopped at 216 us (3.6 TB/s).
        // AMAV_PROJECT_UNROLL = 2 / 4 / 8 is a tuning aid.
        // with a region and many channels the launch is a serial chain of C / unroll load groups per wave on a part of the
        // chip: 8 per group there (32 x 3 x 128^2 x 512 channels: 0.273 -> 0.228 ms; 250 x 3 x 32^2 x 256: 4 is better)
        static const int env_unroll = getenv("AMAV_PROJECT_UNROLL") ? atoi(getenv("AMAV_PROJECT_UNROLL")) : 0;
        const int unroll = env_unroll ? env_unroll : (boxes && C >= 512 ? 8 : 4);
        const size_t lds = (size_t)C * 16 * sizeof(float);
        const dim3 grid(3, F, (RR / 4 + 255) / 256);
        if (unroll == 8)
            project_kernel<8, true><<<grid, 256, lds, stream>>>(C, RR, tokens, frame_stride, wplane, out, boxes, radius, R);
        else if (unroll == 2)
            project_kernel<2, true><<<grid, 256, lds, stream>>>(C, RR, tokens, frame_stride, wplane, out, boxes, radius, R);
        else
            project_kernel<4, true><<<grid, 256, lds, stream>>>(C, RR, tokens, frame_stride, wplane, out, boxes, radius, R);
    } else {
        const dim3 grid((RR + 255) / 256, 3, F);
        project_kernel_scalar<<<grid, 256, 0, stream>>>(C, RR, tokens, frame_stride, wplane, out);
    }
    return check_launch("amav_triplane_project");
}

extern "C" int amav_triplane_sample_decode(int F, int N, int R, const float *proj, const float *points,
                                           const float *transl, float radius, const float *wpoint, float *out,
                                           void *stream_) {
    AMAV_REQUIRE(F > 0 && N > 0 && R > 0, "amav_triplane_sample_decode: bad sizes");
    AMAV_REQUIRE(F <= 65535, "amav_triplane_sample_decode: F=%d exceeds grid.y", F);
    AMAV_REQUIRE(proj && points && wpoint && out, "amav_triplane_sample_decode: NULL pointer");
    AMAV_REQUIRE(radius > 0.0f, "amav_triplane_sample_decode: radius must be positive");
    AMAV_REQUIRE(((reinterpret_cast<uintptr_t>(proj) | reinterpret_cast<uintptr_t>(out) |
                   reinterpret_cast<uintptr_t>(wpoint)) & 15) == 0,
                 "amav_triplane_sample_decode: proj/out/head_w_point not 16-B aligned");
    const unsigned grid = (unsigned)(((size_t)N * 4 + 255) / 256) * 8u * (unsigned)((F + 7) / 8);
    sample_decode_kernel<false><<<grid, 256, 0, static_cast<hipStream_t>(stream_)>>>(F, N, R, 0, proj, points, nullptr,
                                                                                    transl, radius, wpoint, out);
    return check_launch("amav_triplane_sample_decode");
}

extern "C" int amav_triplane_sample_decode_indexed(int F, int N, int R, int V, const float *proj,
                                                   const float *vertices, const int32_t *idx4, const float *transl,
                                                   float radius, const float *wpoint, float *out, void *stream_) {
    AMAV_REQUIRE(F > 0 && N > 0 && R > 0 && V > 0, "amav_triplane_sample_decode_indexed: bad sizes");
    AMAV_REQUIRE(F <= 65535, "amav_triplane_sample_decode_indexed: F=%d exceeds grid.y", F);
    AMAV_REQUIRE(proj && vertices && idx4 && wpoint && out, "amav_triplane_sample_decode_indexed: NULL pointer");
    AMAV_REQUIRE(radius > 0.0f, "amav_triplane_sample_decode_indexed: radius must be positive");
    AMAV_REQUIRE(((reinterpret_cast<uintptr_t>(proj) | reinterpret_cast<uintptr_t>(out) |
                   reinterpret_cast<uintptr_t>(wpoint) | reinterpret_cast<uintptr_t>(idx4)) & 15) == 0,
                 "amav_triplane_sample_decode_indexed: proj/out/head_w_point/idx not 16-B aligned");
    const unsigned grid = (unsigned)(((size_t)N * 4 + 255) / 256) * 8u * (unsigned)((F + 7) / 8);
    sample_decode_kernel<true><<<grid, 256, 0, static_cast<hipStream_t>(stream_)>>>(
        F, N, R, V, proj, vertices, reinterpret_cast<const int4 *>(idx4), transl, radius, wpoint, out);
    return check_launch("amav_triplane_sample_decode_indexed");
}

extern "C" int amav_triplane_sample_features(int F, int N, int C, int R, const float *planes, int64_t frame_stride,
                                             int64_t plane_stride, int64_t chan_stride, const float *points, float radius, float *out,
                                             void *stream_) {
    AMAV_REQUIRE(F > 0 && N > 0 && C > 0 && R > 0, "amav_triplane_sample_features: bad sizes");
    AMAV_REQUIRE(F <= 65535, "amav_triplane_sample_features: F=%d exceeds grid.y", F);
    AMAV_REQUIRE(planes && points && out, "amav_triplane_sample_features: NULL pointer");
    AMAV_REQUIRE(radius > 0.0f, "amav_triplane_sample_features: radius must be positive");
    if (C % 64 == 0 && 3 * C / 64 <= 65535) {
        const dim3 grid((unsigned)((N + 63) / 64), (unsigned)(3 * C / 64), (unsigned)F);
        sample_features_tiled_kernel<<<grid, 256, 0, static_cast<hipStream_t>(stream_)>>>(
            N, C, R, planes, frame_stride, plane_stride, chan_stride, points, radius, out);
    } else {
        const dim3 grid(N, F);
        sample_features_kernel<<<grid, 256, 0, static_cast<hipStream_t>(stream_)>>>(N, C, R, planes, frame_stride,
                                                                                   plane_stride, chan_stride, points,
                                                                                   radius, out);
    }
    return check_launch("amav_triplane_sample_features");
}
