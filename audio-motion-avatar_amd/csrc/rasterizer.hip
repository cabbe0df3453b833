// Gaussian tile rasterizer (forward) for gfx950, all frames of a shard per launch.
//
// Replaces diff_gaussian_rasterization's GaussianRasterizer.forward as the reference calls it
// (src/models/renderer.py:555-566) plus the activations around it (renderer.py:532-547,568).  The algorithm being
// replaced fixes the numbers: 16x16 tiles, (tile, depth, index) order, the 0.99 / 1/255 /
// 1e-4 blend thresholds.  Everything else is laid out for CDNA4:
//
//   preprocess  1 thread / (frame, Gaussian): cull, project, conic, 3-sigma tile rectangle; counts instances per
//               tile with integer atomics (L2-resident counters, ~3 per Gaussian).
//   scan        per-frame block scan of the tile counters, then one block scans the frame totals: instance
//               ranges for every (frame, tile) without a host round trip (upstream syncs to read the total).
//   scatter     every visible Gaussian drops (depth_bits << 32 | index) keys into its tiles' ranges.
//   sort        ONE WAVEFRONT PER TILE sorts its range in LDS with a normalised bitonic network (all comparators
//               ascending, so ragged lengths need no padding); keys are unique, so the result equals upstream's
//               stable radix sort by (tile, depth).  Oversized ranges go to a persistent big-tile kernel.
//   render      ONE WAVEFRONT PER TILE, 4 pixels per lane (same column, rows 4 apart): the tile's Gaussians are
//               staged 64 at a time in LDS and read back as wave-uniform broadcasts; early-out by __all();
//               output is pixel-interleaved RGBA so each store instruction writes four full 256-byte tile rows.
//               Block ids are remapped so that one XCD's L2 sees whole frames (the per-frame Gaussian records
//               are fetched into one L2, not eight).
#include "amav_common.h"

namespace amav {
namespace raster {

constexpr int kTile = AMAV_TILE;
constexpr int kSmallCap = 1024;   // keys one wave sorts in its LDS slice (8 KiB)
constexpr int kBigLdsCap = 16384; // keys a 1024-thread block sorts in LDS (128 KiB)
constexpr int kBigBlocks = 256;

struct Status {
    long long total;
    int overflow;
    int big_count;
};

struct Buffers {
    float4 *geom;              // [F*N][3]: {x, y, conA, conB} {conC, opacity, r, g} {b, 1/depth, -, -}
    uint4 *rectd;              // [F*N]: {rx0 | ry0 << 16, rx1 | ry1 << 16, depth bits, radius}
    int *tile_count;           // [F*T]
    int *tile_off;             // [F*(T+1)] exclusive scan within the frame
    long long *frame_total;    // [F]
    long long *frame_base;     // [F]
    unsigned long long *keys;  // [capacity]
    unsigned *sorted;          // [capacity] Gaussian indices in blend order
    int *big_list;             // [F*T] (frame * T + tile) of ranges longer than kSmallCap
    Status *status;
};

static Buffers carve(void *ws, int F, int N, int T, long long cap, size_t *bytes) {
    Carver c(ws);
    Buffers b;
    b.status = c.take<Status>(1);
    b.geom = c.take<float4>((size_t)F * N * 3);
    b.rectd = c.take<uint4>((size_t)F * N);
    b.tile_count = c.take<int>((size_t)F * T);
    b.tile_off = c.take<int>((size_t)F * (T + 1));
    b.frame_total = c.take<long long>(F);
    b.frame_base = c.take<long long>(F);
    b.keys = c.take<unsigned long long>((size_t)cap);
    b.sorted = c.take<unsigned>((size_t)cap);
    b.big_list = c.take<int>((size_t)F * T);
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
    long long capacity;
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
__global__ __launch_bounds__(256) void preprocess_kernel(Params p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y;
    if (i >= p.N) return;
    const size_t gi = (size_t)f * p.N + i;
    uint4 rd = make_uint4(0u, 0u, 0u, 0u);
    int radius_out = 0;

    const float *vm = p.view + f * 16;
    const float *pm = p.proj + f * 16;
    const float *m = at(p.means3d, f, i);
    const float px3 = m[0], py3 = m[1], pz3 = m[2];
    const float vx = vm[0] * px3 + vm[4] * py3 + vm[8] * pz3 + vm[12];
    const float vy = vm[1] * px3 + vm[5] * py3 + vm[9] * pz3 + vm[13];
    const float vz = vm[2] * px3 + vm[6] * py3 + vm[10] * pz3 + vm[14];
    if (vz > 0.2f) {
        const float hx = pm[0] * px3 + pm[4] * py3 + pm[8] * pz3 + pm[12];
        const float hy = pm[1] * px3 + pm[5] * py3 + pm[9] * pz3 + pm[13];
        const float hw = pm[3] * px3 + pm[7] * py3 + pm[11] * pz3 + pm[15];
        const float pw = 1.0f / (hw + 0.0000001f);
        const float ppx = hx * pw, ppy = hy * pw;

        const float *q = at(p.rotations, f, i);
        const float r = q[0], x = q[1], y = q[2], z = q[3];
        const float *sc = at(p.scales, f, i);
        float s0 = sc[0], s1 = sc[1], s2 = sc[2];
        float opacity = at(p.opacities, f, i)[0];
        const float *cl = at(p.colors, f, i);
        float c0 = cl[0], c1 = cl[1], c2 = cl[2];
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
        const float tanx = p.tanfov[2 * f], tany = p.tanfov[2 * f + 1];
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
        if (det != 0.0f) {
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
            if ((rx1 - rx0) * (ry1 - ry0) > 0) {
                radius_out = (int)my_radius;
                rd = make_uint4((unsigned)rx0 | ((unsigned)ry0 << 16), (unsigned)rx1 | ((unsigned)ry1 << 16),
                                __float_as_uint(vz), (unsigned)radius_out);
                float4 *g = p.buf.geom + gi * 3;
                g[0] = make_float4(pix_x, pix_y, cc * det_inv, -cb * det_inv);
                g[1] = make_float4(ca * det_inv, opacity * h_scale, c0, c1);
                g[2] = make_float4(c2, 1.0f / vz, 0.f, 0.f);
                int *cnt = p.buf.tile_count + (size_t)f * p.T;
                for (int ty_ = ry0; ty_ < ry1; ++ty_)
                    for (int tx_ = rx0; tx_ < rx1; ++tx_) atomicAdd(cnt + ty_ * p.gx + tx_, 1);
            }
        }
    }
    p.buf.rectd[gi] = rd;
    if (p.out_radii) p.out_radii[gi] = radius_out;
}

// ---------------------------------------------------------------------------------------------------------- scans
// Exclusive scan of v over the block (256 threads); returns the prefix of this thread, *total = block sum.
__device__ __forceinline__ long long block_exclusive_scan(long long v, long long *lds_wave, long long *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    long long incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        long long o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    if (lane == 63) lds_wave[wave] = incl;
    __syncthreads();
    long long wave_prefix = 0, tot = 0;
    for (int w = 0; w < nw; ++w) {
        long long s = lds_wave[w];
        if (w < wave) wave_prefix += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return wave_prefix + incl - v;
}

__global__ __launch_bounds__(256) void scan_tiles_kernel(Params p) {
    __shared__ long long lds_wave[4];
    const int f = blockIdx.x;
    const int per = (p.T + 255) / 256;
    const int t0 = threadIdx.x * per;
    const int *cnt = p.buf.tile_count + (size_t)f * p.T;
    int *off = p.buf.tile_off + (size_t)f * (p.T + 1);
    long long local = 0;
    for (int k = 0; k < per; ++k) {
        int t = t0 + k;
        if (t < p.T) local += cnt[t];
    }
    long long total;
    long long prefix = block_exclusive_scan(local, lds_wave, &total);
    int run = (int)prefix;
    for (int k = 0; k < per; ++k) {
        int t = t0 + k;
        if (t < p.T) {
            int c = cnt[t];
            off[t] = run;
            run += c;
            if (c > kSmallCap) {
                int slot = atomicAdd(&p.buf.status->big_count, 1);
                p.buf.big_list[slot] = f * p.T + t;
            }
        }
    }
    if (threadIdx.x == 0) {
        off[p.T] = (int)total;
        p.buf.frame_total[f] = total;
    }
}

__global__ __launch_bounds__(256) void scan_frames_kernel(Params p) {
    __shared__ long long lds_wave[4];
    long long carry = 0;
    for (int base = 0; base < p.F; base += 256) {
        int f = base + threadIdx.x;
        long long v = f < p.F ? p.buf.frame_total[f] : 0;
        long long total;
        long long prefix = block_exclusive_scan(v, lds_wave, &total);
        if (f < p.F) p.buf.frame_base[f] = carry + prefix;
        carry += total;
    }
    if (threadIdx.x == 0) {
        p.buf.status->total = carry;
        p.buf.status->overflow = carry > p.capacity ? 1 : 0;
    }
}

// -------------------------------------------------------------------------------------------------------- scatter
__global__ __launch_bounds__(256) void scatter_kernel(Params p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y;
    if (i >= p.N || p.buf.status->overflow) return;
    const uint4 rd = p.buf.rectd[(size_t)f * p.N + i];
    if (rd.w == 0u) return;
    const int rx0 = rd.x & 0xffff, ry0 = rd.x >> 16, rx1 = rd.y & 0xffff, ry1 = rd.y >> 16;
    const unsigned long long key = ((unsigned long long)rd.z << 32) | (unsigned)i;
    int *cnt = p.buf.tile_count + (size_t)f * p.T;
    const int *off = p.buf.tile_off + (size_t)f * (p.T + 1);
    unsigned long long *keys = p.buf.keys + p.buf.frame_base[f];
    for (int ty = ry0; ty < ry1; ++ty)
        for (int tx = rx0; tx < rx1; ++tx) {
            const int t = ty * p.gx + tx;
            const int slot = atomicSub(cnt + t, 1) - 1;  // counters drain back to zero
            keys[off[t] + slot] = key;
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

__global__ __launch_bounds__(256) void sort_small_kernel(Params p) {
    __shared__ unsigned long long lds[4][kSmallCap];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long gt = (long long)blockIdx.x * 4 + wave;  // frame * T + tile
    if (gt >= (long long)p.F * p.T || p.buf.status->overflow) return;
    const int f = (int)(gt / p.T), t = (int)(gt % p.T);
    const int *off = p.buf.tile_off + (size_t)f * (p.T + 1);
    const int beg = off[t], n = off[t + 1] - beg;
    if (n == 0 || n > kSmallCap) return;
    const unsigned long long *keys = p.buf.keys + p.buf.frame_base[f] + beg;
    unsigned *sorted = p.buf.sorted + p.buf.frame_base[f] + beg;
    unsigned long long *a = lds[wave];
    for (int k = lane; k < n; k += 64) a[k] = keys[k];
    wave_sync();
    bitonic_sort(a, n, lane, 64, [] { wave_sync(); });
    for (int k = lane; k < n; k += 64) sorted[k] = (unsigned)(a[k] & 0xffffffffull);
}

__global__ __launch_bounds__(1024) void sort_big_kernel(Params p) {
    extern __shared__ unsigned long long big_lds[];
    if (p.buf.status->overflow) return;
    const int count = p.buf.status->big_count;
    for (int w = blockIdx.x; w < count; w += gridDim.x) {
        const int gt = p.buf.big_list[w];
        const int f = gt / p.T, t = gt % p.T;
        const int *off = p.buf.tile_off + (size_t)f * (p.T + 1);
        const int beg = off[t], n = off[t + 1] - beg;
        unsigned long long *keys = p.buf.keys + p.buf.frame_base[f] + beg;
        unsigned *sorted = p.buf.sorted + p.buf.frame_base[f] + beg;
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
// Blocks are dealt round-robin over the 8 XCDs; give each XCD a contiguous range of logical blocks (= frames).
__device__ __forceinline__ unsigned xcd_remap(unsigned b, unsigned nb) {
    const unsigned xcd = b & 7u, q = nb >> 3, r = nb & 7u;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

template <bool kInvDepth>
__global__ __launch_bounds__(256) void render_kernel(Params p, int blocks_per_frame) {
    __shared__ float4 stage[4][3][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned b = xcd_remap(blockIdx.x, gridDim.x);
    const int f = b / blocks_per_frame;
    const int t = (b - f * blocks_per_frame) * 4 + wave;
    if (t >= p.T) return;
    const int tx = t % p.gx, ty = t / p.gx;
    const int px = tx * kTile + (lane & 15);
    const int py0 = ty * kTile + (lane >> 4);  // this lane's pixels: (px, py0 + 4k), k = 0..3
    const float pxf = (float)px;

    float T[4] = {1.f, 1.f, 1.f, 1.f};
    float Cr[4] = {0.f, 0.f, 0.f, 0.f}, Cg[4] = {0.f, 0.f, 0.f, 0.f}, Cb[4] = {0.f, 0.f, 0.f, 0.f};
    float Dp[4] = {0.f, 0.f, 0.f, 0.f};
    bool done[4];
    float pyf[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        pyf[k] = (float)(py0 + 4 * k);
        done[k] = !(px < p.W && py0 + 4 * k < p.H);
    }

    const bool overflow = p.buf.status->overflow != 0;
    const int *off = p.buf.tile_off + (size_t)f * (p.T + 1);
    const int beg = off[t];
    const int n = overflow ? 0 : off[t + 1] - beg;
    if (n > 0) {
        const unsigned *sorted = p.buf.sorted + p.buf.frame_base[f] + beg;
        const float4 *geom = p.buf.geom + (size_t)f * p.N * 3;
        float4(*st)[64] = stage[wave];
        float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0, g2 = g0;
        if (lane < n) {
            const float4 *g = geom + (size_t)sorted[lane] * 3;
            g0 = g[0], g1 = g[1], g2 = g[2];
        }
        for (int base = 0; base < n; base += 64) {
            const int cnt = min(64, n - base);
            st[0][lane] = g0;
            st[1][lane] = g1;
            st[2][lane] = g2;
            wave_sync();
            // prefetch the next chunk's records while this one is blended
            if (base + 64 + lane < n) {
                const float4 *g = geom + (size_t)sorted[base + 64 + lane] * 3;
                g0 = g[0], g1 = g[1], g2 = g[2];
            }
            for (int j = 0; j < cnt; ++j) {
                const float4 a = st[0][j], bq = st[1][j], c = st[2][j];
                const float dx = a.x - pxf;
                const float adx2 = a.z * dx * dx;
                const float bdx = a.w * dx;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float dy = a.y - pyf[k];
                    const float power = -0.5f * (adx2 + bq.x * dy * dy) - bdx * dy;
                    const float alpha = fminf(0.99f, bq.y * __expf(power));
                    bool valid = !done[k] && power <= 0.0f && alpha >= (1.0f / 255.0f);
                    const float test_T = T[k] * (1.0f - alpha);
                    const bool fin = valid && test_T < 0.0001f;
                    done[k] = done[k] || fin;
                    valid = valid && !fin;
                    const float w = valid ? alpha * T[k] : 0.0f;
                    Cr[k] += bq.z * w;
                    Cg[k] += bq.w * w;
                    Cb[k] += c.x * w;
                    if (kInvDepth) Dp[k] += c.y * w;
                    T[k] = valid ? test_T : T[k];
                }
                if ((j & 15) == 15 && __all(done[0] && done[1] && done[2] && done[3])) {
                    base = n;  // every pixel of the tile is saturated
                    break;
                }
            }
            wave_sync();
        }
    }

    if (px < p.W) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int py = py0 + 4 * k;
            if (py < p.H) {
                float r = Cr[k] + T[k] * p.bg[0], g = Cg[k] + T[k] * p.bg[1], bl = Cb[k] + T[k] * p.bg[2];
                if (p.clamp_output) {
                    r = fminf(fmaxf(r, 0.f), 1.f);
                    g = fminf(fmaxf(g, 0.f), 1.f);
                    bl = fminf(fmaxf(bl, 0.f), 1.f);
                }
                const size_t pid = ((size_t)f * p.H + py) * p.W + px;
                reinterpret_cast<float4 *>(p.out_rgba)[pid] = make_float4(r, g, bl, 1.0f - T[k]);
                if (kInvDepth) p.out_inv_depth[pid] = Dp[k];
            }
        }
    }
}

}  // namespace raster
}  // namespace amav

using namespace amav;
using namespace amav::raster;

extern "C" size_t amav_rasterize_workspace_bytes(int F, int N, int H, int W, int64_t capacity) {
    if (F <= 0 || N <= 0 || H <= 0 || W <= 0 || capacity < 0) return 0;
    const int gx = (W + kTile - 1) / kTile, gy = (H + kTile - 1) / kTile;
    size_t bytes = 0;
    carve(nullptr, F, N, gx * gy, capacity, &bytes);
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
    size_t need = 0;
    Params p;
    p.buf = carve(a->workspace, F, N, T, a->instance_capacity, &need);
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
    p.capacity = a->instance_capacity;

    hipStream_t stream = static_cast<hipStream_t>(stream_);
    // tile counters + status start at zero (the counters also drain to zero in scatter; this covers first use
    // and an overflowed previous call)
    if (hipMemsetAsync(p.buf.status, 0, sizeof(Status), stream) != hipSuccess ||
        hipMemsetAsync(p.buf.tile_count, 0, (size_t)F * T * sizeof(int), stream) != hipSuccess)
        return fail(AMAV_ERR_LAUNCH, "amav_rasterize_forward: hipMemsetAsync failed");

    const dim3 ggrid((N + 255) / 256, F);
    preprocess_kernel<<<ggrid, 256, 0, stream>>>(p);
    scan_tiles_kernel<<<F, 256, 0, stream>>>(p);
    scan_frames_kernel<<<1, 256, 0, stream>>>(p);
    scatter_kernel<<<ggrid, 256, 0, stream>>>(p);
    const long long ntile = (long long)F * T;
    sort_small_kernel<<<(unsigned)((ntile + 3) / 4), 256, 0, stream>>>(p);
    static const hipError_t big_attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&sort_big_kernel),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                                           kBigLdsCap * sizeof(unsigned long long));
    if (big_attr != hipSuccess)
        return fail(AMAV_ERR_LAUNCH, "amav_rasterize_forward: cannot reserve %zu B of LDS for the big-tile sort",
                    kBigLdsCap * sizeof(unsigned long long));
    sort_big_kernel<<<kBigBlocks, 1024, kBigLdsCap * sizeof(unsigned long long), stream>>>(p);
    const int bpf = (T + 3) / 4;
    if (a->out_inv_depth)
        render_kernel<true><<<(unsigned)(bpf * F), 256, 0, stream>>>(p, bpf);
    else
        render_kernel<false><<<(unsigned)(bpf * F), 256, 0, stream>>>(p, bpf);
    return check_launch("amav_rasterize_forward");
}

extern "C" int amav_rasterize_status(const void *workspace, int64_t *total, int32_t *overflow, void *stream_) {
    AMAV_REQUIRE(workspace != nullptr, "amav_rasterize_status: workspace is NULL");
    Status s;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    hipError_t e = hipMemcpyAsync(&s, workspace, sizeof(Status), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return fail(AMAV_ERR_LAUNCH, "amav_rasterize_status: %s", hipGetErrorString(e));
    if (total) *total = s.total;
    if (overflow) *overflow = s.overflow;
    return AMAV_OK;
}
