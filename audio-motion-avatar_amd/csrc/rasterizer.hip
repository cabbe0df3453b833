// Gaussian tile rasterizer (forward) for gfx950, all frames of a shard per launch.
//
// Replaces diff_gaussian_rasterization's GaussianRasterizer.forward as the reference calls it
// (src/models/renderer.py:555-566) plus the activations around it (renderer.py:532-547,568).  The algorithm being
// replaced fixes the numbers: 16x16 tiles, (tile, depth, index) order, the 0.99 / 1/255 /
// 1e-4 blend thresholds.  Everything else is laid out for CDNA4 (three launches per shard instead of upstream's
// six launches + two CUB passes + a host sync per frame):
//
//   bin_kernel     ONE 1024-THREAD BLOCK PER FRAME does preprocess, per-tile counting, the scan and the key scatter
//                  for its frame, with the tile counters in LDS.  Device-scope atomics on scattered addresses run at
//                  the memory side on MI355X (the eight XCD L2s are not coherent; measured 0.43 ms per 8.4 M adds);
//                  LDS atomics do not.  Frames own fixed instance regions, so there is no cross-frame scan and no
//                  host round trip (upstream syncs to read the instance count).
//   sort_big       persistent blocks sort the tile lists longer than 512 keys: an exact bucket sort in LDS (<= 2 K
//                  keys), a bitonic network for clustered depths or longer lists.
//   render_kernel  ONE WAVEFRONT PER TILE (and per workgroup): loads its keys and sorts them in its LDS slice -- rank
//                  sort up to 64 keys, an exact depth-bucket sort up to 512 (comparison sorts when the depths are
//                  too clustered); keys are unique, so the order equals upstream's stable radix sort by
//                  (tile, depth) -- then blends 4 pixels per lane, one in each 8x8 quadrant of the tile.  Gaussians
//                  are staged 64 at a time through LDS; the staging lane tests the Gaussian's exact alpha >= 1/255
//                  bounding box against the four quadrants, drops Gaussians that cannot touch the tile (ballot +
//                  mbcnt compaction) and records a 4-bit quadrant mask, so the wave only evaluates quadrants the
//                  Gaussian can reach (wave-uniform branches; skipped evaluations are ones the reference would
//                  reject with alpha < 1/255, so the output is unchanged).  Early-out by __any(); the state of a
//                  finished pixel is the sign of its transmittance.  Output is pixel-interleaved RGBA, so every
//                  store instruction writes eight full 128-byte lines.  Tiles reach the waves through eight work
//                  queues (one per XCD: tile-row bands, rotating with the frame) bucketed by list length and
//                  dispatched longest first.
#include <algorithm>
#include <cstdlib>

#include "amav_common.h"

namespace amav {
namespace raster {

#ifndef AMAV_ABLATE
#define AMAV_ABLATE 0  /* diagnostic builds of the blend kernel only (tools/): never set in the product */
#endif
constexpr int kTile = AMAV_TILE;
constexpr int kRenderWavesPerSimd = 5;  // blend kernel: one-wave workgroups resident per SIMD (register cap)
constexpr int kSortCap = 512;      // keys a wave sorts in its LDS slice (4 KiB); longer lists go to sort_big
constexpr int kBigLdsCap = 2048;   // keys a sort_big block sorts in LDS (16 KiB); longer lists are sorted in place
constexpr int kBigBlocks = 1280;
constexpr float kLog2e = 1.4426950408889634f;

constexpr int kQueues = 8;    // one work queue per XCD (frame f feeds queue f % 8, so an XCD's L2 sees whole frames)
constexpr int kBuckets = 17;  // list-length classes: bucket 0 = longer than 512, then 481..512, ..., 1..32

__host__ __device__ inline int bucket_of(int n) { return n > 512 ? 0 : 16 - (n - 1) / 32; }

struct Status {
    long long total;      // sum over frames of (tile, Gaussian) instances as upstream counts them (3-sigma rectangles)
    long long max_frame;  // largest per-frame count of EMITTED instances (what the instance regions must hold)
    long long emitted;    // instances actually binned: rectangle tiles that the alpha >= 1/255 box can reach
    int overflow;         // some frame exceeded its region
    int big_count;        // tiles queued for sort_big
    // work lists of the blend kernel: per XCD queue, bucketed by list length (bucket 0 = longest)
    int qcount[kQueues][kBuckets];
    int nempty;           // tiles without Gaussians (background fill)
    // blend kernel: next position of each queue that no wave has taken yet (beyond the first round).  One 128-byte line
    // per queue: device-scope atomics are executed at the memory side line by line, so cursors sharing a line would
    // serialise all eight queues (measured: 1.23 ms instead of 0.6 for the kernel)
    int next[kQueues][32];
};

struct Buffers {
    float4 *geom;              // [F*N][3]: {x, y, conA', conB'} {conC', opacity, r, g} {b, 1/depth, hx, hy}
    uint4 *rectd;              // [F*N]: {rx0 | ry0 << 16, rx1 | ry1 << 16, depth bits, radius}
    int *tile_off;             // [F*(T+1)] exclusive scan within the frame
    unsigned long long *keys;  // [F * cap_per_frame]
    unsigned *sorted;          // [F * cap_per_frame] blend order of the big tiles only
    int *big_list;             // [F*T] (frame * T + tile) of lists longer than kSortCap
    int *queue;                // [kQueues][kBuckets][qcap] (frame * T + tile) of non-empty tiles
    int *empty_list;           // [F*T] (frame * T + tile) of empty tiles
    Status *status;
};

// a queue holds, of every frame, one band of tile rows (bin_kernel): at most ceil(gy / kQueues) rows of gx tiles
static inline size_t queue_capacity(int F, int gx, int gy) { return (size_t)F * ((gy + kQueues - 1) / kQueues) * gx; }

static Buffers carve(void *ws, int F, int N, int gx, int gy, long long cap, size_t *bytes) {
    const int T = gx * gy;
    Carver c(ws);
    Buffers b;
    b.status = c.take<Status>(1);
    b.geom = c.take<float4>((size_t)F * N * 3);
    b.rectd = c.take<uint4>((size_t)F * N);
    b.tile_off = c.take<int>((size_t)F * (T + 1));
    b.keys = c.take<unsigned long long>((size_t)cap);
    b.sorted = c.take<unsigned>((size_t)cap);
    b.big_list = c.take<int>((size_t)F * T);
    b.queue = c.take<int>((size_t)kQueues * kBuckets * queue_capacity(F, gx, gy));
    b.empty_list = c.take<int>((size_t)F * T);
    if (bytes) *bytes = c.total();
    return b;
}

struct Params {
    int F, N, H, W, gx, gy, T;
    amav_attr means3d, rotations, scales, opacities, colors;
    const float *view, *proj, *tanfov;
    float bg[3];
    float scale_modifier;
    int apply_activations;
    float scale_bias, scale_max, opacity_bias;
    int antialiasing, clamp_output;
    float *out_rgba;
    float *out_inv_depth;
    int *out_radii;
    long long cap_per_frame;
    int qcap;  // entries per work queue
    unsigned long long *stamps;  // diagnostic: [F*T][6] s_memtime stamps per tile wave, or NULL
    Buffers buf;
};

__device__ __forceinline__ const float *at(const amav_attr &a, int f, int i) {
    return a.ptr + (long long)f * a.frame_stride + (long long)i * a.elem_stride;
}

__device__ __forceinline__ void wave_sync() {
    // LDS operations of one wave execute in order; this only stops the compiler from moving them.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---------------------------------------------------------------------------------------------------- preprocess
// One Gaussian of frame f.  Contraction is off and the operation order is the oracle's (oracle/raster_ref.c): +, *,
// /, sqrt are correctly rounded on both sides, so depth keys, radii and tile rectangles -- the decisions that move
// whole Gaussians between tiles or swap their blend order -- come out bit-identical to the CPU restatement.
struct GaussRec {
    float4 r0, r1, r2, r3;  // xyz,opacity | rotation | scale,- | colour,-
};

// kPacked: the five attributes are views of one packed [.., 16] record (triplane.hip layout): four 16-byte loads
template <bool kPacked>
__device__ __forceinline__ GaussRec load_gaussian(const Params &p, int f, int i) {
    GaussRec g;
    if (kPacked) {
        const float4 *rec = reinterpret_cast<const float4 *>(at(p.means3d, f, i));
        g.r0 = rec[0], g.r1 = rec[1], g.r2 = rec[2], g.r3 = rec[3];
    } else {
        const float *m_ = at(p.means3d, f, i), *q_ = at(p.rotations, f, i), *s_ = at(p.scales, f, i);
        const float *c_ = at(p.colors, f, i);
        g.r0 = make_float4(m_[0], m_[1], m_[2], at(p.opacities, f, i)[0]);
        g.r1 = make_float4(q_[0], q_[1], q_[2], q_[3]);
        g.r2 = make_float4(s_[0], s_[1], s_[2], 0.f);
        g.r3 = make_float4(c_[0], c_[1], c_[2], 0.f);
    }
    return g;
}

__device__ __forceinline__ uint4 preprocess_one(const Params &p, int f, int i, const GaussRec &rec, const float *vm,
                                                const float *pm, float tanx, float tany, int &upstream_tiles) {
#pragma clang fp contract(off)
    const size_t gi = (size_t)f * p.N + i;
    uint4 rd = make_uint4(0u, 0u, 0u, 0u);
    const float4 rec0 = rec.r0, rec1 = rec.r1, rec2 = rec.r2, rec3 = rec.r3;
    const float px3 = rec0.x, py3 = rec0.y, pz3 = rec0.z;
    const float vx = vm[0] * px3 + vm[4] * py3 + vm[8] * pz3 + vm[12];
    const float vy = vm[1] * px3 + vm[5] * py3 + vm[9] * pz3 + vm[13];
    const float vz = vm[2] * px3 + vm[6] * py3 + vm[10] * pz3 + vm[14];
    if (!(vz > 0.2f)) return rd;
    const float hx_ = pm[0] * px3 + pm[4] * py3 + pm[8] * pz3 + pm[12];
    const float hy_ = pm[1] * px3 + pm[5] * py3 + pm[9] * pz3 + pm[13];
    const float hw = pm[3] * px3 + pm[7] * py3 + pm[11] * pz3 + pm[15];
    const float pw = 1.0f / (hw + 0.0000001f);
    const float ppx = hx_ * pw, ppy = hy_ * pw;

    const float r = rec1.x, x = rec1.y, y = rec1.z, z = rec1.w;
    float s0 = rec2.x, s1 = rec2.y, s2 = rec2.z;
    float opacity = rec0.w;
    float c0 = rec3.x, c1 = rec3.y, c2 = rec3.z;
    if (p.apply_activations) {
        s0 = fminf(expf(s0 - p.scale_bias), p.scale_max);
        s1 = fminf(expf(s1 - p.scale_bias), p.scale_max);
        s2 = fminf(expf(s2 - p.scale_bias), p.scale_max);
        opacity = 1.0f / (1.0f + expf(-(opacity - p.opacity_bias)));
        c0 = fminf(fmaxf(c0, 0.0f), 1.0f);
        c1 = fminf(fmaxf(c1, 0.0f), 1.0f);
        c2 = fminf(fmaxf(c2, 0.0f), 1.0f);
    }
    s0 *= p.scale_modifier;
    s1 *= p.scale_modifier;
    s2 *= p.scale_modifier;

    // Sigma3D = R diag(s)^2 R^T
    const float R00 = 1.f - 2.f * (y * y + z * z), R01 = 2.f * (x * y - r * z), R02 = 2.f * (x * z + r * y);
    const float R10 = 2.f * (x * y + r * z), R11 = 1.f - 2.f * (x * x + z * z), R12 = 2.f * (y * z - r * x);
    const float R20 = 2.f * (x * z - r * y), R21 = 2.f * (y * z + r * x), R22 = 1.f - 2.f * (x * x + y * y);
    const float M00 = s0 * R00, M01 = s0 * R10, M02 = s0 * R20;
    const float M10 = s1 * R01, M11 = s1 * R11, M12 = s1 * R21;
    const float M20 = s2 * R02, M21 = s2 * R12, M22 = s2 * R22;
    const float S00 = M00 * M00 + M10 * M10 + M20 * M20;
    const float S01 = M00 * M01 + M10 * M11 + M20 * M21;
    const float S02 = M00 * M02 + M10 * M12 + M20 * M22;
    const float S11 = M01 * M01 + M11 * M11 + M21 * M21;
    const float S12 = M01 * M02 + M11 * M12 + M21 * M22;
    const float S22 = M02 * M02 + M12 * M12 + M22 * M22;

    // EWA splat: cov2D = (J Wv) Sigma (J Wv)^T
    const float focal_x = (float)p.W / (2.0f * tanx), focal_y = (float)p.H / (2.0f * tany);
    const float limx = 1.3f * tanx, limy = 1.3f * tany;
    const float tz = vz;
    const float tx = fminf(limx, fmaxf(-limx, vx / tz)) * tz;
    const float ty = fminf(limy, fmaxf(-limy, vy / tz)) * tz;
    const float J00 = focal_x / tz, J02 = -(focal_x * tx) / (tz * tz);
    const float J11 = focal_y / tz, J12 = -(focal_y * ty) / (tz * tz);
    float T0[3], T1[3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        T0[b] = J00 * vm[b * 4 + 0] + J02 * vm[b * 4 + 2];
        T1[b] = J11 * vm[b * 4 + 1] + J12 * vm[b * 4 + 2];
    }
    const float U00 = S00 * T0[0] + S01 * T0[1] + S02 * T0[2];
    const float U01 = S01 * T0[0] + S11 * T0[1] + S12 * T0[2];
    const float U02 = S02 * T0[0] + S12 * T0[1] + S22 * T0[2];
    const float U10 = S00 * T1[0] + S01 * T1[1] + S02 * T1[2];
    const float U11 = S01 * T1[0] + S11 * T1[1] + S12 * T1[2];
    const float U12 = S02 * T1[0] + S12 * T1[1] + S22 * T1[2];
    float ca = T0[0] * U00 + T0[1] * U01 + T0[2] * U02;
    const float cb = T0[0] * U10 + T0[1] * U11 + T0[2] * U12;
    float cc = T1[0] * U10 + T1[1] * U11 + T1[2] * U12;

    const float det_cov = ca * cc - cb * cb;
    ca += 0.3f;
    cc += 0.3f;
    const float det = ca * cc - cb * cb;
    float h_scale = 1.0f;
    if (p.antialiasing) h_scale = sqrtf(fmaxf(0.000025f, det_cov / det));
    if (det == 0.0f) return rd;
    const float det_inv = 1.0f / det;
    const float mid = 0.5f * (ca + cc);
    const float root = sqrtf(fmaxf(0.1f, mid * mid - det));
    const float my_radius = ceilf(3.0f * sqrtf(fmaxf(mid + root, mid - root)));
    const float pix_x = ((ppx + 1.0f) * (float)p.W - 1.0f) * 0.5f;
    const float pix_y = ((ppy + 1.0f) * (float)p.H - 1.0f) * 0.5f;
    const int rx0 = min(p.gx, max(0, (int)((pix_x - my_radius) / (float)kTile)));
    const int ry0 = min(p.gy, max(0, (int)((pix_y - my_radius) / (float)kTile)));
    const int rx1 = min(p.gx, max(0, (int)((pix_x + my_radius + (float)(kTile - 1)) / (float)kTile)));
    const int ry1 = min(p.gy, max(0, (int)((pix_y + my_radius + (float)(kTile - 1)) / (float)kTile)));
    if ((rx1 - rx0) * (ry1 - ry0) <= 0) return rd;

    const float op = opacity * h_scale;
    // Exact support of the blend: alpha = op * exp(power) >= 1/255  <=>  d^T Q d <= 2 ln(255 op), whose axis-aligned
    // half extents are sqrt(2 ln(255 op) * cov_xx|yy).  Padded (1e-3 relative + 0.01 px), so a pixel outside the box
    // is rejected by the reference with a margin far above rounding; render_kernel uses the box to skip quadrants.
    const float tau = 2.0f * logf(255.0f * op);
    float bx = -1e30f, by = -1e30f;
    if (tau > 0.0f) {
        bx = sqrtf(tau * ca) * 1.001f + 0.01f;
        by = sqrtf(tau * cc) * 1.001f + 0.01f;
    }
    upstream_tiles = (rx1 - rx0) * (ry1 - ry0);
    // bin only the tiles of the 3-sigma rectangle that the alpha box reaches: in the others every pixel is rejected
    // (alpha < 1/255), so dropping them changes nothing but the list lengths
    int cx0 = 1, cx1 = 0, cy0 = 1, cy1 = 0;
    if (tau > 0.0f) {
        cx0 = max(rx0, (int)floorf((pix_x - bx) / (float)kTile));
        cx1 = min(rx1, (int)floorf((pix_x + bx) / (float)kTile) + 1);
        cy0 = max(ry0, (int)floorf((pix_y - by) / (float)kTile));
        cy1 = min(ry1, (int)floorf((pix_y + by) / (float)kTile) + 1);
    }
    if (cx1 <= cx0 || cy1 <= cy0) cx0 = cx1 = cy0 = cy1 = 0;
    rd = make_uint4((unsigned)cx0 | ((unsigned)cy0 << 16), (unsigned)cx1 | ((unsigned)cy1 << 16), __float_as_uint(vz),
                    (unsigned)(int)my_radius);
    float4 *g = p.buf.geom + gi * 3;
    // the conic is stored as the coefficients of log2(alpha / op) = A' dx^2 + B' dx dy + C' dy^2, i.e. pre-multiplied
    // by log2(e) and by the -1/2 and -1 of the exponent (both exact): the blend is two multiplies, an add, an fma, exp2
    g[0] = make_float4(pix_x, pix_y, -0.5f * ((cc * det_inv) * kLog2e), (cb * det_inv) * kLog2e);
    g[1] = make_float4(-0.5f * ((ca * det_inv) * kLog2e), op, c0, c1);
    g[2] = make_float4(c2, 1.0f / vz, bx, by);
    return rd;
}

// ------------------------------------------------------------------------------------------------------------ bin
// Block-wide exclusive scan helper (blockDim.x = 1024 = 16 waves).
__device__ __forceinline__ int block_exclusive_scan(int v, int *lds_wave, int *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    if (lane == 63) lds_wave[wave] = incl;
    __syncthreads();
    int wave_prefix = 0, tot = 0;
    for (int w = 0; w < nw; ++w) {
        const int s = lds_wave[w];
        if (w < wave) wave_prefix += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return wave_prefix + incl - v;
}

// grid = F blocks of 1024 threads; dynamic LDS = (2*T + 16 + 3*8*(kBuckets+1)) ints: counts[T], cursor[T], scratch, classes
template <bool kPacked>
__global__ __launch_bounds__(1024) void bin_kernel(Params p) {
    extern __shared__ int bin_lds[];
    int *counts = bin_lds;
    int *cursor = bin_lds + p.T;
    int *scratch = bin_lds + 2 * p.T + 3 * kQueues * (kBuckets + 1);
    const int f = blockIdx.x;
    for (int t = threadIdx.x; t < p.T; t += blockDim.x) counts[t] = 0;
    __syncthreads();

    const float *vm = p.view + f * 16;
    const float *pm = p.proj + f * 16;
    const float tanx = p.tanfov[2 * f], tany = p.tanfov[2 * f + 1];
    // phase 1: preprocess, count instances per tile (LDS atomics)
    int upstream = 0;
    // the next Gaussian's record is in flight while this one goes through the (long, dependent) projection maths
    GaussRec cur = load_gaussian<kPacked>(p, f, min((int)threadIdx.x, p.N - 1));
    for (int i = threadIdx.x; i < p.N; i += blockDim.x) {
        const GaussRec nxt = load_gaussian<kPacked>(p, f, min(i + (int)blockDim.x, p.N - 1));
        int up = 0;
        const uint4 rd = preprocess_one(p, f, i, cur, vm, pm, tanx, tany, up);
        cur = nxt;
        upstream += up;
        const size_t gi = (size_t)f * p.N + i;
        p.buf.rectd[gi] = rd;
        if (p.out_radii) p.out_radii[gi] = (int)rd.w;
        if (rd.w) {
            const int rx0 = rd.x & 0xffff, ry0 = rd.x >> 16, rx1 = rd.y & 0xffff, ry1 = rd.y >> 16;
            for (int ty = ry0; ty < ry1; ++ty)
                for (int tx = rx0; tx < rx1; ++tx) atomicAdd(&counts[ty * p.gx + tx], 1);
        }
    }
    __syncthreads();

    // phase 2: exclusive scan of the tile counters -> list offsets inside this frame's instance region
    const int per = (p.T + (int)blockDim.x - 1) / (int)blockDim.x;
    const int t0 = threadIdx.x * per;
    int local = 0;
    for (int k = 0; k < per; ++k)
        if (t0 + k < p.T) local += counts[t0 + k];
    int total, upstream_total;
    block_exclusive_scan(upstream, scratch, &upstream_total);
    int run = block_exclusive_scan(local, scratch, &total);
    int *off = p.buf.tile_off + (size_t)f * (p.T + 1);
    for (int k = 0; k < per; ++k) {
        const int t = t0 + k;
        if (t < p.T) {
            const int c = counts[t];
            cursor[t] = run;
            off[t] = run;
            run += c;
            if (c > kSortCap) {
                const int slot = atomicAdd(&p.buf.status->big_count, 1);
                p.buf.big_list[slot] = f * p.T + t;
            }
        }
    }
    const bool fits = (long long)total <= p.cap_per_frame;
    // work items of the blend kernel: non-empty tiles into the work queues, bucketed by list length; empty tiles into
    // the fill list.  Queue of a tile = (band of its tile row + frame) % 8: every queue (= XCD, see render_kernel)
    // gets one eighth of EVERY frame, rotating, so the queues carry equal work whatever the frames look like, and a
    // queue's Gaussian records stay spatially local.
    // Two LDS-counted passes: count per (queue, class), reserve ranges with one global atomic each, emit.
    constexpr int kCls = kBuckets + 1, kQK = kQueues * kCls;  // class kBuckets = empty
    int *cls = cursor + p.T;  // [kQK] counts, then bases, then emit cursors
    if (threadIdx.x < 3 * kQK) cls[threadIdx.x] = 0;
    __syncthreads();
    auto queue_of = [&](int t) { return (min(kQueues - 1, (t / p.gx) * kQueues / p.gy) + f) % kQueues; };
    for (int k = 0; k < per; ++k)
        if (t0 + k < p.T) {
            const int c = fits ? counts[t0 + k] : 0;
            atomicAdd(&cls[queue_of(t0 + k) * kCls + (c == 0 ? kBuckets : bucket_of(c))], 1);
        }
    __syncthreads();
    if (threadIdx.x < kQK && cls[threadIdx.x]) {
        const int q = threadIdx.x / kCls, kind = threadIdx.x - q * kCls;
        cls[kQK + threadIdx.x] = kind == kBuckets ? atomicAdd(&p.buf.status->nempty, cls[threadIdx.x])
                                                  : atomicAdd(&p.buf.status->qcount[q][kind], cls[threadIdx.x]);
    }
    __syncthreads();
    for (int k = 0; k < per; ++k)
        if (t0 + k < p.T) {
            const int c = fits ? counts[t0 + k] : 0;
            const int kind = c == 0 ? kBuckets : bucket_of(c);
            const int qi = queue_of(t0 + k), slot = qi * kCls + kind;
            const int pos = cls[kQK + slot] + atomicAdd(&cls[2 * kQK + slot], 1);
            if (kind == kBuckets)
                p.buf.empty_list[pos] = f * p.T + t0 + k;
            else
                p.buf.queue[((size_t)qi * kBuckets + kind) * p.qcap + pos] = f * p.T + t0 + k;
        }
    if (threadIdx.x == 0) {
        off[p.T] = total;
        atomicAdd(reinterpret_cast<unsigned long long *>(&p.buf.status->total), (unsigned long long)upstream_total);
        atomicAdd(reinterpret_cast<unsigned long long *>(&p.buf.status->emitted), (unsigned long long)total);
        atomicMax(reinterpret_cast<unsigned long long *>(&p.buf.status->max_frame), (unsigned long long)total);
        if (!fits) atomicExch(&p.buf.status->overflow, 1);
    }
    __syncthreads();
    if (!fits) return;

    // phase 3: scatter (depth, index) keys into the tile lists (order inside a list is fixed later by the sort)
    unsigned long long *keys = p.buf.keys + (size_t)f * p.cap_per_frame;
    const uint4 *rect = p.buf.rectd + (size_t)f * p.N;
    uint4 rd_next = rect[min((int)threadIdx.x, p.N - 1)];
    for (int i = threadIdx.x; i < p.N; i += blockDim.x) {
        const uint4 rd = rd_next;
        rd_next = rect[min(i + (int)blockDim.x, p.N - 1)];
        if (rd.w == 0u) continue;
        const int rx0 = rd.x & 0xffff, ry0 = rd.x >> 16, rx1 = rd.y & 0xffff, ry1 = rd.y >> 16;
        const unsigned long long key = ((unsigned long long)rd.z << 32) | (unsigned)i;
        for (int ty = ry0; ty < ry1; ++ty)
            for (int tx = rx0; tx < rx1; ++tx) keys[atomicAdd(&cursor[ty * p.gx + tx], 1)] = key;
    }
}

// ----------------------------------------------------------------------------------------------------------- sort
// Normalised bitonic network: every comparator orders (lo, hi) ascending, so a tail of "+inf" needs no storage:
// comparators whose hi index is past n are skipped.
template <typename Sync>
__device__ __forceinline__ void bitonic_sort(unsigned long long *a, int n, int tid, int nthreads, Sync sync) {
    int P = 1;
    while (P < n) P <<= 1;
    const int half = P >> 1;
    for (int k = 2; k <= P; k <<= 1) {
        const int hk = k >> 1;
        for (int c = tid; c < half; c += nthreads) {
            const int blk = c / hk, o = c - blk * hk;
            const int lo = blk * k + o, hi = blk * k + k - 1 - o;
            if (hi < n) {
                unsigned long long x = a[lo], y = a[hi];
                if (x > y) {
                    a[lo] = y;
                    a[hi] = x;
                }
            }
        }
        sync();
        for (int j = k >> 2; j > 0; j >>= 1) {
            for (int c = tid; c < half; c += nthreads) {
                const int lo = 2 * j * (c / j) + (c % j), hi = lo + j;
                if (hi < n) {
                    unsigned long long x = a[lo], y = a[hi];
                    if (x > y) {
                        a[lo] = y;
                        a[hi] = x;
                    }
                }
            }
            sync();
        }
    }
}

// Lists longer than kSortCap, one 256-thread block per list.  Up to kBigLdsCap keys: the block form of the blend
// kernel's bucket sort (2048 depth buckets, exact: ties inside a bucket are ordered by the full key), about ten
// barriers instead of the bitonic network's sixty-six; clustered depths (a bucket above kBigBucketMax keys) and longer
// lists take the bitonic network.
constexpr int kBigBuckets = 2048;
constexpr int kBigBucketMax = 32;

__global__ __launch_bounds__(256) void sort_big_kernel(Params p) {
    __shared__ unsigned long long big_lds[kBigLdsCap];
    __shared__ __align__(16) unsigned cnt[kBigBuckets / 2 + 4];
    __shared__ unsigned part[12];
    if (p.buf.status->overflow) return;
    const int count = p.buf.status->big_count;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int w = blockIdx.x; w < count; w += gridDim.x) {
        const int gt = p.buf.big_list[w];
        const int f = gt / p.T, t = gt % p.T;
        const int *off = p.buf.tile_off + (size_t)f * (p.T + 1);
        const int beg = off[t], n = off[t + 1] - beg;
        unsigned long long *keys = p.buf.keys + (size_t)f * p.cap_per_frame + beg;
        unsigned *sorted = p.buf.sorted + (size_t)f * p.cap_per_frame + beg;
        bool done = false;
        if (n <= kBigLdsCap) {
            constexpr int KPT = kBigLdsCap / 256;  // 8 keys and 8 buckets per thread
            unsigned long long k[KPT];
            unsigned dmin = 0xffffffffu, dmax = 0u;
#pragma unroll
            for (int m = 0; m < KPT; ++m) {
                const bool in = tid + 256 * m < n;
                k[m] = in ? keys[tid + 256 * m] : ~0ull;
                const unsigned d = (unsigned)(k[m] >> 32);
                if (in) dmin = min(dmin, d), dmax = max(dmax, d);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                dmin = min(dmin, (unsigned)__shfl_xor((int)dmin, o, 64));
                dmax = max(dmax, (unsigned)__shfl_xor((int)dmax, o, 64));
            }
            if (lane == 0) part[wave] = dmin, part[4 + wave] = dmax;
#pragma unroll
            for (int i = 0; i < 4; ++i) cnt[tid + 256 * i] = 0u;
            __syncthreads();
            dmin = min(min(part[0], part[1]), min(part[2], part[3]));
            dmax = max(max(part[4], part[5]), max(part[6], part[7]));
            const float scale = dmax > dmin ? (float)(kBigBuckets - 1) / (float)(dmax - dmin) : 0.0f;
            int b[KPT];
            unsigned pos[KPT];
#pragma unroll
            for (int m = 0; m < KPT; ++m) {
                b[m] = min(kBigBuckets - 1, (int)((float)((unsigned)(k[m] >> 32) - dmin) * scale));
                pos[m] = 0;
                if (tid + 256 * m < n) {
                    const int sh = 16 * (b[m] & 1);
                    pos[m] = (atomicAdd(&cnt[b[m] >> 1], 1u << sh) >> sh) & 0xffffu;
                }
            }
            __syncthreads();
            // exclusive scan of the 2048 counts: thread owns buckets 8*tid .. 8*tid + 7
            const uint4 wv = reinterpret_cast<uint4 *>(cnt)[tid];
            unsigned c[8] = {wv.x & 0xffffu, wv.x >> 16, wv.y & 0xffffu, wv.y >> 16,
                             wv.z & 0xffffu, wv.z >> 16, wv.w & 0xffffu, wv.w >> 16};
            unsigned tot = 0, big = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const unsigned ci = c[i];
                big = max(big, ci);
                c[i] = tot;
                tot += ci;
            }
            unsigned incl = tot;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned up = (unsigned)__shfl_up((int)incl, o, 64);
                if (lane >= o) incl += up;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) big = max(big, (unsigned)__shfl_xor((int)big, o, 64));
            __syncthreads();  // everyone has read part[] and its counters
            if (lane == 63) part[wave] = incl;
            if (lane == 0) part[4 + wave] = big;
            __syncthreads();
            big = max(max(part[4], part[5]), max(part[6], part[7]));
            if (big <= (unsigned)kBigBucketMax) {  // block-uniform
                unsigned base = incl - tot;
                for (int i = 0; i < wave; ++i) base += part[i];
                reinterpret_cast<uint4 *>(cnt)[tid] =
                    make_uint4((base + c[0]) | ((base + c[1]) << 16), (base + c[2]) | ((base + c[3]) << 16),
                               (base + c[4]) | ((base + c[5]) << 16), (base + c[6]) | ((base + c[7]) << 16));
                if (tid == 0) cnt[kBigBuckets / 2] = (unsigned)n;
                __syncthreads();
                unsigned st[KPT], sz[KPT];
#pragma unroll
                for (int m = 0; m < KPT; ++m) {
                    st[m] = (cnt[b[m] >> 1] >> (16 * (b[m] & 1))) & 0xffffu;
                    const int nb = b[m] + 1;
                    sz[m] = ((cnt[nb >> 1] >> (16 * (nb & 1))) & 0xffffu) - st[m];
                    if (tid + 256 * m < n) big_lds[st[m] + pos[m]] = k[m];
                }
                __syncthreads();
#pragma unroll
                for (int m = 0; m < KPT; ++m) {
                    if (tid + 256 * m < n) {
                        unsigned r = pos[m];
                        if (sz[m] > 1u) {
                            r = 0;
                            for (unsigned j = 0; j < sz[m]; ++j) r += (unsigned)(big_lds[st[m] + j] < k[m]);
                        }
                        sorted[st[m] + r] = (unsigned)k[m];
                    }
                }
                done = true;
            }
            __syncthreads();  // LDS is reused by the next list (or by the fallback below)
        }
        if (done) continue;
        if (n <= kBigLdsCap) {
            for (int k = threadIdx.x; k < n; k += blockDim.x) big_lds[k] = keys[k];
            __syncthreads();
            bitonic_sort(big_lds, n, threadIdx.x, blockDim.x, [] { __syncthreads(); });
            for (int k = threadIdx.x; k < n; k += blockDim.x) sorted[k] = (unsigned)(big_lds[k] & 0xffffffffull);
            __syncthreads();
        } else {
            // in place in global memory: one block owns the range, so block-scope visibility is enough
            bitonic_sort(keys, n, threadIdx.x, blockDim.x, [] {
                __threadfence_block();
                __syncthreads();
            });
            for (int k = threadIdx.x; k < n; k += blockDim.x) sorted[k] = (unsigned)(keys[k] & 0xffffffffull);
            __syncthreads();
        }
    }
}

// --------------------------------------------------------------------------------------------------------- render
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct WaveLds {
    unsigned long long keys[kSortCap];  // keys, then (in place) the blend order as 32-bit Gaussian ids
    float4 stage[3][65];                // staged records {x, y, A', B'} {C', opacity, r, g} {b, 1/depth, ..}; [64] = null record
};

// Rank sort of n <= 64 * KPL unique keys held in LDS: rank = number of smaller keys, no cross-lane exchange.  The
// 32-bit ids then overwrite the key slice in blend order (every lane has read all keys by then).
template <int KPL>
__device__ __forceinline__ void rank_sort(unsigned long long *keys, unsigned *order, int n, int lane) {
    unsigned long long my[KPL];
    int rank[KPL];
#pragma unroll
    for (int m = 0; m < KPL; ++m) {
        my[m] = (lane + 64 * m < n) ? keys[lane + 64 * m] : ~0ull;
        rank[m] = 0;
    }
    int j = 0;
    for (; j + 4 <= n; j += 4) {  // four broadcast reads in flight per round
        const unsigned long long k0 = keys[j], k1 = keys[j + 1], k2 = keys[j + 2], k3 = keys[j + 3];
#pragma unroll
        for (int m = 0; m < KPL; ++m)
            rank[m] += (int)(k0 < my[m]) + (int)(k1 < my[m]) + (int)(k2 < my[m]) + (int)(k3 < my[m]);
    }
    for (; j < n; ++j) {
        const unsigned long long kj = keys[j];
#pragma unroll
        for (int m = 0; m < KPL; ++m) rank[m] += (int)(kj < my[m]);
    }
    wave_sync();
#pragma unroll
    for (int m = 0; m < KPL; ++m)
        if (lane + 64 * m < n) order[rank[m]] = (unsigned)my[m];
    wave_sync();
}

// Bucket sort of 64 < n <= 64 * KPL unique keys by one wave, O(n / 64) per lane for well spread depths.  The depth
// word of a key is mapped monotonically onto 512 buckets between the tile's nearest and farthest Gaussian, the keys
// are counted (packed 16-bit LDS counters), the counts are scanned, every key is placed into its bucket's slice, and
// ties inside a bucket are ordered by comparing the full 64-bit keys of the slice, so the result is the exact
// (depth, id) order whatever the depths are.  Returns false (keys stored to LDS, nothing sorted) when a bucket holds
// more than kBucketMax keys: clustered depths are left to the comparison sorts.
constexpr int kBuckets512 = 512;
constexpr int kBucketMax = 16;

template <int KPL>
__device__ __forceinline__ bool bucket_sort(const unsigned long long *__restrict__ gkeys, unsigned long long *slice,
                                            unsigned *cnt, unsigned *order, int n, int lane) {
    unsigned long long k[KPL];
    unsigned dmin = 0xffffffffu, dmax = 0u;
#pragma unroll
    for (int m = 0; m < KPL; ++m) {
        const bool in = lane + 64 * m < n;
        k[m] = in ? gkeys[lane + 64 * m] : ~0ull;
        const unsigned d = (unsigned)(k[m] >> 32);
        if (in) dmin = min(dmin, d), dmax = max(dmax, d);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        dmin = min(dmin, (unsigned)__shfl_xor((int)dmin, o, 64));
        dmax = max(dmax, (unsigned)__shfl_xor((int)dmax, o, 64));
    }
    const float scale = dmax > dmin ? (float)(kBuckets512 - 1) / (float)(dmax - dmin) : 0.0f;
    int b[KPL];
#pragma unroll
    for (int m = 0; m < KPL; ++m)  // monotone in the depth word: conversion, product and truncation all are
        b[m] = min(kBuckets512 - 1, (int)((float)((unsigned)(k[m] >> 32) - dmin) * scale));
    // counters: 256 words of two 16-bit counts (+ one word past the end for the scan's total)
    reinterpret_cast<uint4 *>(cnt)[lane] = make_uint4(0u, 0u, 0u, 0u);
    wave_sync();
    unsigned pos[KPL];
#pragma unroll
    for (int m = 0; m < KPL; ++m) {
        pos[m] = 0;
        if (lane + 64 * m < n) {
            const int sh = 16 * (b[m] & 1);
            pos[m] = (atomicAdd(&cnt[b[m] >> 1], 1u << sh) >> sh) & 0xffffu;  // arrival index inside the bucket
        }
    }
    wave_sync();
    // exclusive scan of the 512 counts: lane owns buckets 8*lane .. 8*lane + 7
    const uint4 w = reinterpret_cast<uint4 *>(cnt)[lane];
    unsigned c[8] = {w.x & 0xffffu, w.x >> 16, w.y & 0xffffu, w.y >> 16, w.z & 0xffffu, w.z >> 16, w.w & 0xffffu, w.w >> 16};
    unsigned tot = 0, big = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const unsigned ci = c[i];
        big = max(big, ci);
        c[i] = tot;
        tot += ci;
    }
    unsigned incl = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned up = (unsigned)__shfl_up((int)incl, o, 64);
        if (lane >= o) incl += up;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) big = max(big, (unsigned)__shfl_xor((int)big, o, 64));
    if (big > (unsigned)kBucketMax) {  // wave-uniform
        wave_sync();
#pragma unroll
        for (int m = 0; m < KPL; ++m)
            if (lane + 64 * m < n) slice[lane + 64 * m] = k[m];
        wave_sync();
        return false;
    }
    const unsigned base = incl - tot;
    reinterpret_cast<uint4 *>(cnt)[lane] =
        make_uint4((base + c[0]) | ((base + c[1]) << 16), (base + c[2]) | ((base + c[3]) << 16),
                   (base + c[4]) | ((base + c[5]) << 16), (base + c[6]) | ((base + c[7]) << 16));
    if (lane == 0) cnt[kBuckets512 / 2] = (unsigned)n;  // start of the bucket past the last one
    wave_sync();
    unsigned st[KPL], sz[KPL];
#pragma unroll
    for (int m = 0; m < KPL; ++m) {
        st[m] = (cnt[b[m] >> 1] >> (16 * (b[m] & 1))) & 0xffffu;
        const int nb = b[m] + 1;
        sz[m] = ((cnt[nb >> 1] >> (16 * (nb & 1))) & 0xffffu) - st[m];
        if (lane + 64 * m < n) slice[st[m] + pos[m]] = k[m];
    }
    wave_sync();
#pragma unroll
    for (int m = 0; m < KPL; ++m) {
        unsigned r = pos[m];
        if (lane + 64 * m < n && sz[m] > 1u) {  // order the bucket by the full key
            r = 0;
            for (unsigned j = 0; j < sz[m]; ++j) r += (unsigned)(slice[st[m] + j] < k[m]);
        }
        pos[m] = st[m] + r;
    }
    wave_sync();  // the ids overwrite the slice: every lane has finished reading it
#pragma unroll
    for (int m = 0; m < KPL; ++m)
        if (lane + 64 * m < n) order[pos[m]] = (unsigned)k[m];
    wave_sync();
    return true;
}

// Bitonic sort of n (wave-uniform, n <= kSortCap) unique keys in LDS by one wave: O(n log^2 n / 64) comparators
// against the rank sort's O(n^2 / 64) compares; it wins above 256 keys (measured: 31 us against 52 us per tile for
// n in [256, 512), 16 against 13 for [128, 256)).  Normalised network (every comparator orders lo < hi ascending), so
// the "+inf" tail up to the next power of two needs no storage: comparators that reach past n are skipped.
__device__ __forceinline__ void wave_bitonic_sort(unsigned long long *a, unsigned *order, int n, int lane) {
    const int lp = 32 - __builtin_clz(n - 1);  // P = 2^lp >= n
    const int half = 1 << (lp - 1);
    auto cmpswap = [&](int lo, int hi) {
        if (hi < n) {
            const unsigned long long x = a[lo], y = a[hi];
            if (x > y) {
                a[lo] = y;
                a[hi] = x;
            }
        }
    };
    for (int lk = 1; lk <= lp; ++lk) {
        const int k = 1 << lk, hk = k >> 1;
        for (int c = lane; c < half; c += 64) {
            const int base = (c >> (lk - 1)) << lk, o = c & (hk - 1);
            cmpswap(base + o, base + k - 1 - o);
        }
        wave_sync();
        for (int lj = lk - 2; lj >= 0; --lj) {
            const int j = 1 << lj;
            for (int c = lane; c < half; c += 64) {
                const int lo = ((c >> lj) << (lj + 1)) + (c & (j - 1));
                cmpswap(lo, lo + j);
            }
            wave_sync();
        }
    }
    // the 32-bit ids overwrite the key slice in blend order (id k lands inside key k/2: read everything first)
    unsigned id[kSortCap / 64];
#pragma unroll
    for (int m = 0; m < kSortCap / 64; ++m) id[m] = (lane + 64 * m < n) ? (unsigned)a[lane + 64 * m] : 0u;
    wave_sync();
#pragma unroll
    for (int m = 0; m < kSortCap / 64; ++m)
        if (lane + 64 * m < n) order[lane + 64 * m] = id[m];
    wave_sync();
}

// One pixel, one Gaussian.  T > 0: live transmittance; T < 0: pixel finished, |T| is its final transmittance.
// A finished pixel needs no test of its own: with T < 0 the weight alpha*T is negative, T - alpha*T < 1e-4 holds, so
// the Gaussian is either invalid (w = 0) or takes the "finished" branch (w = 0, T <- -|T| = T).
// Record words: a = {x, y, A', B'}, b = {C', opacity, r, g}, c = {b, 1/depth}; A' B' C' are the coefficients of
// log2(alpha / opacity) = A' dx^2 + B' dx dy + C' dy^2 (bin_kernel), evaluated as dx (A' dx + B' dy) + (C' dy) dy.
template <bool kInvDepth>
__device__ __forceinline__ void blend_px(const float4 &a, const float4 &b, const float2 &c, float px, float py,
                                         float &T, float &Cr, float &Cg, float &Cb, float &Dp) {
    const float dx = a.x - px, dy = a.y - py;
    const float power2 = fmaf(dx, fmaf(a.w, dy, a.z * dx), (b.x * dy) * dy);
#if AMAV_ABLATE == 2  /* diagnostic build: no transcendental */
    float alpha = fminf(0.99f, b.y * (power2 + 1.0f));
#else
    float alpha = fminf(0.99f, b.y * __builtin_amdgcn_exp2f(power2));
#endif
    // The three decisions of the reference (skip power > 0, skip alpha < 1/255, stop at T' < 1e-4) as compare + select
    // pairs on VCC only: no SGPR-pair logic between vector instructions (each v_cmp -> s_and -> v_cndmask round trip
    // parks the wave until the vector pipe has drained).
    alpha = power2 <= 0.0f ? alpha : 0.0f;
    alpha = alpha >= (1.0f / 255.0f) ? alpha : 0.0f;  // 0: this Gaussian does not touch the pixel
    const float w0 = alpha * T;
    const float test_T = T - w0;       // = T (1 - alpha) up to one rounding; = T when alpha was zeroed
    // A live pixel has T >= 1e-4 (it would have finished otherwise), so with alpha = 0 the test below is false; a
    // finished pixel (T < 0) gives test_T < 0 and re-takes the "finished" branch, which leaves it as it is.
    const bool fin = test_T < 0.0001f;
    const float w = fin ? 0.0f : w0;
    Cr = fmaf(b.z, w, Cr);
    Cg = fmaf(b.w, w, Cg);
    Cb = fmaf(c.x, w, Cb);
    if (kInvDepth) Dp = fmaf(c.y, w, Dp);
    T -= w;                    // unchanged unless this Gaussian was blended
    T = fin ? -fabsf(T) : T;   // finished: the saturating Gaussian is not blended, the sign marks the pixel done
}

struct StageRec {
    float4 a, b;
    float2 c;
};

__device__ __forceinline__ StageRec read_stage(const WaveLds &L, int j) {  // j is wave-uniform: broadcast reads
    StageRec r;
    r.a = L.stage[0][j];
    r.b = L.stage[1][j];
    r.c = *reinterpret_cast<const float2 *>(&L.stage[2][j]);
    return r;
}

constexpr int kNullSlot = 64;  // staging slot of a record with opacity 0: what an exhausted list keeps reading

// Next set bit of a wave-uniform mask, cleared, in three scalar instructions; an exhausted mask yields kNullSlot, so
// the caller's LDS reads stay unconditional and the compiler can count them exactly in its s_waitcnt (a read under a
// branch makes every later wait assume the worst: "everything issued so far").
__device__ __forceinline__ int next_bit(unsigned long long &mask) {
    int j;
    asm("s_ff1_i32_b64 %0, %1\n\ts_bitset0_b64 %1, %0\n\ts_min_u32 %0, %0, %2" : "=&s"(j), "+s"(mask) : "n"(kNullSlot) : "scc");
    return j;  // s_ff1 of 0 is -1: bit 63 of the (empty) mask is "cleared" and the unsigned minimum gives 64
}

// One 8x8 quadrant of the tile (one pixel per lane) against the staged Gaussians whose bit is set in `mask` (a
// wave-uniform 64-bit ballot, i.e. scalar registers: the loop walks its set bits with scalar instructions, lowest =
// nearest first, so the blend order is the staged order).  Records are read from LDS two Gaussians ahead of their use
// (three register sets, loop unrolled by three); the list is processed in threes, the last
// group padded with the null record, so there is one loop branch per three Gaussians.  Returns false when every
// pixel of the quadrant has finished (checked every six Gaussians).
template <bool kInvDepth>
__device__ __forceinline__ bool blend_quadrant(unsigned long long mask, const WaveLds &L, float px, float py, float &T,
                                               float &R, float &G, float &B, float &D) {
    int groups = (__popcll(mask) + 2) / 3;  // >= 1
    StageRec r0 = read_stage(L, next_bit(mask));
    StageRec r1 = read_stage(L, next_bit(mask)), r2;
    bool alive = true;  // wave-uniform: some pixel of the quadrant still takes Gaussians
    do {  // one back edge, one exit (two exits cost five scalar branches per group instead of two; same speed)
#if AMAV_ABLATE == 1  /* diagnostic build: no LDS reads inside the loop */
        (void)next_bit(mask); (void)next_bit(mask); (void)next_bit(mask);
        r2 = r0;
        asm volatile("" : "+v"(r0.a.x), "+v"(r1.a.x), "+v"(r2.a.x));
        blend_px<kInvDepth>(r0.a, r0.b, r0.c, px, py, T, R, G, B, D);
        blend_px<kInvDepth>(r1.a, r1.b, r1.c, px, py, T, R, G, B, D);
        blend_px<kInvDepth>(r2.a, r2.b, r2.c, px, py, T, R, G, B, D);
#else
        r2 = read_stage(L, next_bit(mask));
        blend_px<kInvDepth>(r0.a, r0.b, r0.c, px, py, T, R, G, B, D);
        r0 = read_stage(L, next_bit(mask));
        blend_px<kInvDepth>(r1.a, r1.b, r1.c, px, py, T, R, G, B, D);
        r1 = read_stage(L, next_bit(mask));
        blend_px<kInvDepth>(r2.a, r2.b, r2.c, px, py, T, R, G, B, D);
#endif
        --groups;
        if ((groups & 1) == 0) alive = __any(T > 0.f);  // a finished quadrant takes no further Gaussians (every six)
    } while (groups != 0 && alive);
    return alive && __any(T > 0.f);
}

#define AMAV_STAMP(slot)                                                                              \
    do {                                                                                              \
        if (p.stamps && lane == 0) p.stamps[(size_t)item * 6 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)

// Background for a tile without Gaussians: colour = bg, alpha = 0 (and inverse depth 0).
template <bool kInvDepth>
__device__ __forceinline__ void fill_tile(const Params &p, int item, int lane) {
    const int f = item / p.T, t = item - f * p.T;
    const int X0 = (t % p.gx) * kTile, Y0 = (t / p.gx) * kTile;
    float r = p.bg[0], g = p.bg[1], bl = p.bg[2];
    if (p.clamp_output) {
        r = fminf(fmaxf(r, 0.f), 1.f);
        g = fminf(fmaxf(g, 0.f), 1.f);
        bl = fminf(fmaxf(bl, 0.f), 1.f);
    }
    // lane -> (column lane & 15, rows (lane >> 4) + 4k): every store instruction covers four full 256-byte tile rows
    const int px = X0 + (lane & 15);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int py = Y0 + (lane >> 4) + 4 * k;
        if (px < p.W && py < p.H) {
            const size_t pid = ((size_t)f * p.H + py) * p.W + px;
            reinterpret_cast<float4 *>(p.out_rgba)[pid] = make_float4(r, g, bl, 0.0f);
            if (kInvDepth) p.out_inv_depth[pid] = 0.0f;
        }
    }
}

// One non-empty tile, one wavefront.
template <bool kInvDepth>
__device__ __forceinline__ void render_tile(const Params &p, WaveLds &L, int item, int lane) {
    const int f = item / p.T, t = item - f * p.T;
    AMAV_STAMP(0);
    const int tx = t % p.gx, ty = t / p.gx;
    const int X0 = tx * kTile, Y0 = ty * kTile;
    // this lane's four pixels: (X0 + 8*qx + lx, Y0 + 8*qy + ly), quadrant q = qx + 2*qy
    const int lx = lane & 7, ly = lane >> 3;
    float pxf0 = (float)(X0 + lx), pxf1 = (float)(X0 + 8 + lx);
    float pyf0 = (float)(Y0 + ly), pyf1 = (float)(Y0 + 8 + ly);
    asm volatile("" : "+v"(pxf0), "+v"(pxf1), "+v"(pyf0), "+v"(pyf1));  // keep them in registers (no re-convert)
    const bool in0 = X0 + lx < p.W, in1 = X0 + 8 + lx < p.W, inr0 = Y0 + ly < p.H, inr1 = Y0 + 8 + ly < p.H;

    float T0 = (in0 & inr0) ? 1.f : -1.f, T1 = (in1 & inr0) ? 1.f : -1.f;
    float T2 = (in0 & inr1) ? 1.f : -1.f, T3 = (in1 & inr1) ? 1.f : -1.f;
    float R0 = 0.f, G0 = 0.f, B0 = 0.f, D0 = 0.f, R1 = 0.f, G1 = 0.f, B1 = 0.f, D1 = 0.f;
    float R2 = 0.f, G2 = 0.f, B2 = 0.f, D2 = 0.f, R3 = 0.f, G3 = 0.f, B3 = 0.f, D3 = 0.f;

    const int *off = p.buf.tile_off + (size_t)f * (p.T + 1);
    const int beg = __builtin_amdgcn_readfirstlane(off[t]);
    const int n = __builtin_amdgcn_readfirstlane(off[t + 1]) - beg;
    AMAV_STAMP(1);
    if (p.stamps && lane == 0) p.stamps[(size_t)item * 6 + 5] = (unsigned long long)n;
    {
        const unsigned long long *keys = p.buf.keys + (size_t)f * p.cap_per_frame + beg;
        const unsigned *order_g = p.buf.sorted + (size_t)f * p.cap_per_frame + beg;  // long lists only
        unsigned *order_l = reinterpret_cast<unsigned *>(L.keys);
        const bool local = n <= kSortCap;
#if AMAV_ABLATE == 4 || AMAV_ABLATE == 8
        if (local) {
            for (int k = lane; k < n; k += 64) order_l[k] = (unsigned)keys[k];
            wave_sync();
        } else
#endif
        if (local) {
            unsigned *cnt = reinterpret_cast<unsigned *>(L.stage);  // the staging buffers are idle while sorting
            bool done = false;
            if (n <= 64) {
                for (int k = lane; k < n; k += 64) L.keys[k] = keys[k];
                wave_sync();
            } else if (n <= 128)
                done = bucket_sort<2>(keys, L.keys, cnt, order_l, n, lane);
            else if (n <= 192)
                done = bucket_sort<3>(keys, L.keys, cnt, order_l, n, lane);
            else if (n <= 256)
                done = bucket_sort<4>(keys, L.keys, cnt, order_l, n, lane);
            else if (n <= 384)
                done = bucket_sort<6>(keys, L.keys, cnt, order_l, n, lane);
            else
                done = bucket_sort<8>(keys, L.keys, cnt, order_l, n, lane);
            if (!done) {  // short list, or depths too clustered for buckets: comparison sorts on the keys in LDS
                if (n <= 64)
                    rank_sort<1>(L.keys, order_l, n, lane);
                else if (n <= 128)
                    rank_sort<2>(L.keys, order_l, n, lane);
                else if (n <= 192)
                    rank_sort<3>(L.keys, order_l, n, lane);
                else if (n <= 256)
                    rank_sort<4>(L.keys, order_l, n, lane);
                else
                    wave_bitonic_sort(L.keys, order_l, n, lane);
            }
        }
        AMAV_STAMP(2);
        // the sorts used the staging buffers as scratch: (re)write the null record (opacity 0 blends nothing)
        if (lane < 3) L.stage[lane][kNullSlot] = make_float4(0.f, 0.f, 0.f, 0.f);

        // ---- blend, 64 Gaussians per staging round
        const float4 *geom = p.buf.geom + (size_t)f * p.N * 3;
        float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0, g2 = g0;
        // blend order of position k: from this wave's LDS slice, or (lists longer than kSortCap) from sort_big's output.
        // Two explicit paths: a select between an LDS and a global pointer becomes a FLAT load, whose completion the
        // hardware can only express as "everything done" (vmcnt(0) + lgkmcnt(0)).
        typedef __attribute__((address_space(3))) const unsigned lds_u32;
        lds_u32 *order_lds = (lds_u32 *)order_l;
        auto load_records = [&](int k) {
            unsigned id;
            if (local)
                id = order_lds[k];
            else
                id = order_g[k];
            const float4 *g = geom + (size_t)id * 3;
            g0 = g[0], g1 = g[1], g2 = g[2];
        };
        if (lane < n) load_records(lane);
        const float X0f = (float)X0, Y0f = (float)Y0;
        int qalive = 15;  // quadrants that still have an unfinished pixel (wave-uniform)
        for (int base = 0; qalive && base < n; base += 64) {
            // quadrant mask of this lane's Gaussian: which live 8x8 quadrants its alpha >= 1/255 box can reach
            int qm = 0;
            if (base + lane < n) {
                const bool hx0 = (g0.x + g2.z >= X0f) & (g0.x - g2.z <= X0f + 7.f);
                const bool hx1 = (g0.x + g2.z >= X0f + 8.f) & (g0.x - g2.z <= X0f + 15.f);
                const bool hy0 = (g0.y + g2.w >= Y0f) & (g0.y - g2.w <= Y0f + 7.f);
                const bool hy1 = (g0.y + g2.w >= Y0f + 8.f) & (g0.y - g2.w <= Y0f + 15.f);
                qm = (int)(hx0 & hy0) | ((int)(hx1 & hy0) << 1) | ((int)(hx0 & hy1) << 2) | ((int)(hx1 & hy1) << 3);
                qm &= qalive;
            }
            // every lane stages its record at its own (= sorted) position; the four ballots are the quadrants' lists
            if (qm != 0) {
                L.stage[0][lane] = g0;
                L.stage[1][lane] = g1;
                L.stage[2][lane] = g2;  // {b, 1/depth, box half extents}: stored whole, so no copy of it is made early
            }
            const unsigned long long m0 = __ballot(qm & 1), m1 = __ballot(qm & 2), m2 = __ballot(qm & 4),
                                     m3 = __ballot(qm & 8);
            wave_sync();
            // prefetch the next round's records while this one is blended
            if (base + 64 + lane < n) load_records(base + 64 + lane);
#if AMAV_ABLATE == 3 || AMAV_ABLATE == 7 || AMAV_ABLATE == 8  /* diagnostic build: no blending at all */
            if (false)
#endif
            if (m0 && !blend_quadrant<kInvDepth>(m0, L, pxf0, pyf0, T0, R0, G0, B0, D0)) qalive &= ~1;
#if AMAV_ABLATE != 3 && AMAV_ABLATE != 7 && AMAV_ABLATE != 8
            if (m1 && !blend_quadrant<kInvDepth>(m1, L, pxf1, pyf0, T1, R1, G1, B1, D1)) qalive &= ~2;
            if (m2 && !blend_quadrant<kInvDepth>(m2, L, pxf0, pyf1, T2, R2, G2, B2, D2)) qalive &= ~4;
            if (m3 && !blend_quadrant<kInvDepth>(m3, L, pxf1, pyf1, T3, R3, G3, B3, D3)) qalive &= ~8;
#endif
            wave_sync();
        }
    }

    AMAV_STAMP(3);
    // ---- write back: quadrant q of the wave = 8 rows x 128 B
    const float Tq[4] = {fabsf(T0), fabsf(T1), fabsf(T2), fabsf(T3)};
    const float Rq[4] = {R0, R1, R2, R3}, Gq[4] = {G0, G1, G2, G3}, Bq[4] = {B0, B1, B2, B3}, Dq[4] = {D0, D1, D2, D3};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int px = X0 + 8 * (q & 1) + lx, py = Y0 + 8 * (q >> 1) + ly;
        if (px < p.W && py < p.H) {
            float r = Rq[q] + Tq[q] * p.bg[0], g = Gq[q] + Tq[q] * p.bg[1], bl = Bq[q] + Tq[q] * p.bg[2];
            if (p.clamp_output) {
                r = fminf(fmaxf(r, 0.f), 1.f);
                g = fminf(fmaxf(g, 0.f), 1.f);
                bl = fminf(fmaxf(bl, 0.f), 1.f);
            }
            const size_t pid = ((size_t)f * p.H + py) * p.W + px;
#if AMAV_ABLATE == 6 || AMAV_ABLATE == 8
            if (r == 123.f)  /* diagnostic build: no tile stores */
#endif
            reinterpret_cast<float4 *>(p.out_rgba)[pid] = make_float4(r, g, bl, 1.0f - Tq[q]);
            if (kInvDepth) p.out_inv_depth[pid] = Dq[q];
        }
    }
    AMAV_STAMP(4);
}

// Blend kernel: PERSISTENT waves (one wave per workgroup, the grid is what the chip holds at once).  The waves of
// queue q (= blockIdx % 8: blocks are dealt round-robin over the XCDs, so an XCD keeps seeing the tile-row bands of
// its own queue -- a placement assumption that only affects speed) walk the queue's buckets, which are ordered
// LONGEST LISTS FIRST, so the kernel ends on the shortest tiles.  The first position of a wave is static; every later
// one comes from the queue's cursor (one returning atomic per tile on one of eight words, issued a tile ahead so its
// round trip hides under the blending: ~16 dequeues per microsecond and word, far below the ~88 a word sustains).
// (Round 1 launched one workgroup per possible tile -- 256 000 of them for 51 000 non-empty tiles; the 205 000 waves
// that only wrote background held an eighth of the wave slots.)  After each tile the wave writes a share of its
// background tiles, so those stores stay spread over the whole kernel, under the VALU-bound blending.
template <bool kInvDepth>
__global__ __launch_bounds__(64, kRenderWavesPerSimd) void render_kernel(Params p) {
    __shared__ WaveLds lds;
    const int lane = threadIdx.x;
    const Status *st = p.buf.status;
    const int nempty = st->nempty;
    const int nw = gridDim.x, per = (nempty + nw - 1) / nw;
    int e0 = min(nempty, (int)blockIdx.x * per);
    const int e1 = min(nempty, e0 + per);
    if (!st->overflow) {
        const int q = blockIdx.x % kQueues, stride = gridDim.x / kQueues;
        int total = 0;
        for (int b = 0; b < kBuckets; ++b) total += st->qcount[q][b];
        // expected tiles per wave, to spread this wave's background tiles over its blended ones
        const int expect = max(1, (total + stride - 1) / stride);
        const int fill_chunk = (e1 - e0 + expect - 1) / expect;
        int *next = const_cast<int *>(&st->next[q][0]);
        int i = blockIdx.x / kQueues;  // first round: static; afterwards the queue's shared cursor
        while (i < total) {
            // take the following position now: the atomic's round trip hides under this tile
            int nxt = 0;
            if (lane == 0) nxt = stride + atomicAdd(next, 1);
            int b = 0, acc = 0;
            while (i >= acc + st->qcount[q][b]) acc += st->qcount[q][b++];  // i < total: b stays inside the table
            // wave-uniform: keep the tile id (and everything derived from it) in scalar registers
            const int item = __builtin_amdgcn_readfirstlane(p.buf.queue[((size_t)q * kBuckets + b) * p.qcap + (i - acc)]);
            render_tile<kInvDepth>(p, lds, item, lane);
#if AMAV_ABLATE != 5 && AMAV_ABLATE != 7 && AMAV_ABLATE != 8
            for (const int stop = min(e1, e0 + fill_chunk); e0 < stop; ++e0) fill_tile<kInvDepth>(p, p.buf.empty_list[e0], lane);
#endif
            i = __builtin_amdgcn_readfirstlane(nxt);
        }
    }
    // the rest of this wave's background tiles (all of them when it had no tile to blend; every tile of the launch
    // when the instance regions overflowed: the caller must retry)
#if AMAV_ABLATE != 5 && AMAV_ABLATE != 7 && AMAV_ABLATE != 8
    for (; e0 < e1; ++e0) fill_tile<kInvDepth>(p, p.buf.empty_list[e0], lane);
#endif
}

}  // namespace raster
}  // namespace amav

using namespace amav;
using namespace amav::raster;

// Resident waves of the blend kernel (the persistent grid).  A grid above the true residency only delays the surplus
// waves' first tile; AMAV_RENDER_WAVES (waves per CU) overrides the default for tuning.
static unsigned render_grid(bool /*inv_depth*/) {
    static unsigned g = 0;
    if (g) return g;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        cus = 256;
    (void)hipGetLastError();
    // __launch_bounds__(64, kRenderWavesPerSimd) caps the registers so that many one-wave workgroups fit on a SIMD;
    // their LDS (7 KiB each) fits 22 times into a CU's 160 KiB.  (hipOccupancyMaxActiveBlocksPerMultiprocessor
    // answers 16 for this kernel on ROCm 7.2 although 20 are resident.)
    int per_cu = 4 * kRenderWavesPerSimd;
    if (const char *env = getenv("AMAV_RENDER_WAVES")) {
        const int v = atoi(env);
        if (v >= 1 && v <= 32) per_cu = v;
    }
    g = (unsigned)(cus * per_cu) / kQueues * kQueues;
    if (g < (unsigned)kQueues) g = kQueues;
    return g;
}

extern "C" size_t amav_rasterize_workspace_bytes(int F, int N, int H, int W, int64_t capacity) {
    if (F <= 0 || N <= 0 || H <= 0 || W <= 0 || capacity < 0) return 0;
    const int gx = (W + kTile - 1) / kTile, gy = (H + kTile - 1) / kTile;
    size_t bytes = 0;
    carve(nullptr, F, N, gx, gy, capacity / F * F, &bytes);
    return bytes;
}

extern "C" int amav_rasterize_forward(const amav_raster_args *a, void *stream_) {
    AMAV_REQUIRE(a != nullptr, "amav_rasterize_forward: args is NULL");
    AMAV_REQUIRE(a->num_frames > 0 && a->num_gaussians > 0 && a->height > 0 && a->width > 0,
                 "amav_rasterize_forward: bad sizes F=%d N=%d H=%d W=%d", a->num_frames, a->num_gaussians, a->height,
                 a->width);
    AMAV_REQUIRE(a->means3d.ptr && a->rotations.ptr && a->scales.ptr && a->opacities.ptr && a->colors.ptr,
                 "amav_rasterize_forward: NULL Gaussian attribute");
    AMAV_REQUIRE(a->viewmatrix && a->projmatrix && a->tanfov, "amav_rasterize_forward: NULL camera");
    AMAV_REQUIRE(a->out_rgba != nullptr, "amav_rasterize_forward: out_rgba is NULL");
    AMAV_REQUIRE(a->workspace != nullptr, "amav_rasterize_forward: workspace is NULL");
    AMAV_REQUIRE(a->instance_capacity >= 0, "amav_rasterize_forward: negative instance_capacity");
    AMAV_REQUIRE((reinterpret_cast<uintptr_t>(a->out_rgba) & 15) == 0, "amav_rasterize_forward: out_rgba not 16-B aligned");
    const int F = a->num_frames, N = a->num_gaussians;
    const int gx = (a->width + kTile - 1) / kTile, gy = (a->height + kTile - 1) / kTile;
    AMAV_REQUIRE(gx < 65536 && gy < 65536, "amav_rasterize_forward: image too large");
    const int T = gx * gy;
    AMAV_REQUIRE((long long)F * T < (1ll << 31), "amav_rasterize_forward: F * tiles overflows int32");
    const size_t bin_lds = ((size_t)2 * T + 16 + 3 * kQueues * (kBuckets + 1)) * sizeof(int);
    AMAV_REQUIRE(bin_lds <= 160 * 1024, "amav_rasterize_forward: %d tiles need %zu B of LDS in the binning block (max 160 KiB)",
                 T, bin_lds);
    const long long cap_per_frame = a->instance_capacity / F;
    AMAV_REQUIRE(cap_per_frame < (1ll << 31), "amav_rasterize_forward: per-frame instance capacity overflows int32");
    size_t need = 0;
    Params p;
    p.buf = carve(a->workspace, F, N, gx, gy, cap_per_frame * F, &need);
    if (a->workspace_bytes < need)
        return fail(AMAV_ERR_WORKSPACE, "amav_rasterize_forward: workspace %zu < required %zu", a->workspace_bytes, need);
    p.F = F, p.N = N, p.H = a->height, p.W = a->width, p.gx = gx, p.gy = gy, p.T = T;
    p.means3d = a->means3d, p.rotations = a->rotations, p.scales = a->scales, p.opacities = a->opacities;
    p.colors = a->colors;
    p.view = a->viewmatrix, p.proj = a->projmatrix, p.tanfov = a->tanfov;
    p.bg[0] = a->bg[0], p.bg[1] = a->bg[1], p.bg[2] = a->bg[2];
    p.scale_modifier = a->scale_modifier;
    p.apply_activations = a->apply_activations;
    p.scale_bias = a->scale_bias, p.scale_max = a->scale_max, p.opacity_bias = a->opacity_bias;
    p.antialiasing = a->antialiasing, p.clamp_output = a->clamp_output;
    p.out_rgba = a->out_rgba, p.out_inv_depth = a->out_inv_depth, p.out_radii = a->out_radii;
    p.cap_per_frame = cap_per_frame;
    p.qcap = (int)queue_capacity(F, gx, gy);
    p.stamps = static_cast<unsigned long long *>(a->debug_stamps);

    hipStream_t stream = static_cast<hipStream_t>(stream_);
    static const hipError_t attr[2] = {
        hipFuncSetAttribute(reinterpret_cast<const void *>(&bin_kernel<false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
        hipFuncSetAttribute(reinterpret_cast<const void *>(&bin_kernel<true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)};
    if (attr[0] != hipSuccess || attr[1] != hipSuccess)
        return fail(AMAV_ERR_LAUNCH, "amav_rasterize_forward: cannot raise the dynamic LDS limit");
    // packed-record fast path: the attributes are the xyz|opacity|rot|scale|color views of one [.., 16] buffer
    const float *b0 = a->means3d.ptr;
    const bool packed = (reinterpret_cast<uintptr_t>(b0) & 15) == 0 && a->means3d.elem_stride == 16 &&
                        a->means3d.frame_stride % 4 == 0 && a->opacities.ptr == b0 + 3 && a->rotations.ptr == b0 + 4 &&
                        a->scales.ptr == b0 + 8 && a->colors.ptr == b0 + 12 && a->opacities.elem_stride == 16 &&
                        a->rotations.elem_stride == 16 && a->scales.elem_stride == 16 && a->colors.elem_stride == 16 &&
                        a->opacities.frame_stride == a->means3d.frame_stride &&
                        a->rotations.frame_stride == a->means3d.frame_stride &&
                        a->scales.frame_stride == a->means3d.frame_stride &&
                        a->colors.frame_stride == a->means3d.frame_stride;
    if (zero_async(p.buf.status, sizeof(Status), stream) != hipSuccess)
        return fail(AMAV_ERR_LAUNCH, "amav_rasterize_forward: status clear failed");

    if (packed)
        bin_kernel<true><<<F, 1024, bin_lds, stream>>>(p);
    else
        bin_kernel<false><<<F, 1024, bin_lds, stream>>>(p);
    sort_big_kernel<<<kBigBlocks, 256, 0, stream>>>(p);
    // persistent grid: the waves the chip holds at once (a multiple of kQueues)
    const unsigned blocks = render_grid(a->out_inv_depth != nullptr);
    if (a->profile_start_event) (void)hipEventRecord(static_cast<hipEvent_t>(a->profile_start_event), stream);
    if (a->out_inv_depth)
        render_kernel<true><<<blocks, 64, 0, stream>>>(p);
    else
        render_kernel<false><<<blocks, 64, 0, stream>>>(p);
    if (a->profile_stop_event) (void)hipEventRecord(static_cast<hipEvent_t>(a->profile_stop_event), stream);
    return check_launch("amav_rasterize_forward");
}

namespace amav {
namespace raster {
__global__ __launch_bounds__(256) void tile_counts_kernel(int F, int T, const Status *__restrict__ st,
                                                          const int *__restrict__ tile_off, int *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= F * T) return;
    const int f = i / T, t = i - f * T;
    const int *off = tile_off + (size_t)f * (T + 1);
    out[i] = st->overflow ? 1 : off[t + 1] - off[t];  // after an overflow nothing is known: every tile "may be drawn"
}
}  // namespace raster
}  // namespace amav

extern "C" int amav_rasterize_tile_counts(const void *workspace, int F, int N, int H, int W, int64_t capacity,
                                          int32_t *out_counts, void *stream_) {
    AMAV_REQUIRE(workspace && out_counts, "amav_rasterize_tile_counts: NULL pointer");
    AMAV_REQUIRE(F > 0 && N > 0 && H > 0 && W > 0 && capacity >= 0, "amav_rasterize_tile_counts: bad sizes");
    const int gx = (W + kTile - 1) / kTile, gy = (H + kTile - 1) / kTile, T = gx * gy;
    size_t bytes = 0;
    const Buffers b = carve(const_cast<void *>(workspace), F, N, gx, gy, capacity / F * F, &bytes);
    tile_counts_kernel<<<(unsigned)(((long long)F * T + 255) / 256), 256, 0, static_cast<hipStream_t>(stream_)>>>(
        F, T, b.status, b.tile_off, out_counts);
    return check_launch("amav_rasterize_tile_counts");
}

extern "C" int amav_rasterize_status(const void *workspace, int64_t *total, int64_t *max_frame, int32_t *overflow,
                                     void *stream_) {
    AMAV_REQUIRE(workspace != nullptr, "amav_rasterize_status: workspace is NULL");
    Status s;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    hipError_t e = hipMemcpyAsync(&s, workspace, sizeof(Status), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return fail(AMAV_ERR_LAUNCH, "amav_rasterize_status: %s", hipGetErrorString(e));
    if (total) *total = s.total;
    if (max_frame) *max_frame = s.max_frame;
    if (overflow) *overflow = s.overflow;
    return AMAV_OK;
}
