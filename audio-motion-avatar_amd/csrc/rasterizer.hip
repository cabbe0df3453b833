// Gaussian tile rasterizer (forward) for gfx950, all frames of a shard per launch.
//
// Replaces diff_gaussian_rasterization's GaussianRasterizer.forward as the reference calls it
// (src/models/renderer.py:555-566) plus the activations around it (renderer.py:532-547,568).  The algorithm being
// replaced fixes the numbers: 16x16 tiles, (tile, depth, index) order, the 0.99 / 1/255 /
// 1e-4 blend thresholds.  Everything else is laid out for CDNA4 (three launches per shard instead of upstream's
// six launches + two CUB passes + a host sync per frame):
//
//   bin_kernel     ONE 1024-THREAD BLOCK PER FRAME does preprocess (for shards of many frames; a launch of its own,
//                  preprocess_kernel, when there are fewer frames than CUs), per-tile counting, the scan and the key scatter
//                  for its frame, with the tile counters in LDS.  Device-scope atomics on scattered addresses run at
//                  the memory side on MI355X (the eight XCD L2s are not coherent; measured 0.43 ms per 8.4 M adds);
//                  LDS atomics do not.  Frames own fixed instance regions, so there is no cross-frame scan and no
//                  host round trip (upstream syncs to read the instance count).
//   sort_big       persistent blocks sort the tile lists longer than 512 keys: an exact bucket sort in LDS (<= 2 K
//                  keys), a bitonic network for clustered depths or longer lists.
//   render_kernel  PERSISTENT WAVES, one tile at a time per wave (one wave per workgroup): the tile's keys are sorted
//                  in the wave's LDS slice -- rank sort up to 64 keys, an exact depth-bucket sort up to 512
//                  (comparison sorts when the depths are too clustered); keys are unique, so the order equals
//                  upstream's stable radix sort by (tile, depth) -- then blended 4 pixels per lane, one in each 8x8
//                  quadrant of the tile.  Gaussians are staged 64 at a time through LDS, moved into the tile's frame;
//                  the staging lane tests the Gaussian's exact alpha >= 1/255 bounding box against the four
//                  quadrants and enters the record's LDS address into the lists of the quadrants it can reach
//                  (ballot + mbcnt), so a quadrant's blend loop walks a dense address list and skips only
//                  evaluations the reference would reject with alpha < 1/255.  The blend loop itself is hand-scheduled
//                  assembly (17 vector instructions per pixel and Gaussian, LDS reads pipelined with counted waits).
//                  The next tile's start-up chain (queue entry -> keys -> sort -> first records) runs under the
//                  current tile's blending.  Output is pixel-interleaved RGBA, so every store instruction writes
//                  full 128-byte lines.  Tiles reach the waves through eight work queues (one per XCD: tile-row
//                  bands, rotating with the frame) bucketed by list length and dispatched longest first.
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdlib>

#include "amav_common.h"

namespace amav {
namespace raster {

#ifndef AMAV_BIN_ABLATE
#define AMAV_BIN_ABLATE 0  /* diagnostic builds of the binning kernel only (tools/stamp_bin.sh) */
#endif
#ifndef AMAV_ABLATE
#define AMAV_ABLATE 0  /* diagnostic builds of the blend kernel only (tools/): never set in the product */
#endif
constexpr int kTile = AMAV_TILE;
constexpr int kRenderWavesPerSimd = 4;  // blend kernel: one-wave workgroups resident per SIMD (register cap)
constexpr int kFusedMinFrames = 96;  // binning: shards below this project their Gaussians in a launch of their own
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
    float4 *geom;              // [F*N][3]: {x, y, qa, qb} {qc, log2(opacity), r, g} {b, 1/depth, hx, hy} (preprocess_one)
    uint4 *rectd;              // [F*N]: {rx0 | ry0 << 16, rx1 | ry1 << 16, depth bits, radius}
    int *tile_off;             // [F*(T+1)] exclusive scan within the frame
    unsigned long long *keys;  // [F * cap_per_frame]
    unsigned *sorted;          // [F * cap_per_frame] blend order of the big tiles only
    int *big_list;             // [F*T] (frame * T + tile) of lists longer than kSortCap
    int4 *queue;               // [kQueues][kBuckets][qcap] non-empty tiles: {frame * T + tile, list offset in the frame's region, list length, 0}
    int *empty_list;           // [F*T] (frame * T + tile) of empty tiles
    int *slice_counts;         // [F][slices][T] few-frame shards: per-tile counts of every slice of a frame, then their prefix over the slices
    Status *status;
};

// Few frames (F < kFusedMinFrames): a frame's Gaussians are counted and scattered by several blocks ("slices"), so that
// 6 or 32 frames still spread over the chip; ~256 blocks in all, at most 16 per frame.
static inline int bin_slices(int F) { return F >= kFusedMinFrames ? 0 : std::min(16, std::max(1, (256 + F - 1) / F)); }

// a queue holds, of every frame, one band of tile rows (bin_kernel): at most ceil(gy / kQueues) rows of gx tiles
static inline size_t queue_capacity(int F, int gx, int gy) { return (size_t)F * ((gy + kQueues - 1) / kQueues) * gx; }

static Buffers carve(void *ws, int F, int N, int gx, int gy, long long cap, size_t *bytes) {
    const int T = gx * gy;
    Carver c(ws);
    Buffers b;
    b.status = c.take<Status>(1);
    // + 1024 spare records behind the last frame: where the binning block's lanes past the end of a frame store
    b.geom = c.take<float4>(((size_t)F * N + 1024) * 3);
    b.rectd = c.take<uint4>((size_t)F * N + 1024);
    b.tile_off = c.take<int>((size_t)F * (T + 1));
    b.keys = c.take<unsigned long long>((size_t)cap);
    b.sorted = c.take<unsigned>((size_t)cap);
    b.big_list = c.take<int>((size_t)F * T);
    b.queue = c.take<int4>((size_t)kQueues * kBuckets * queue_capacity(F, gx, gy));
    b.empty_list = c.take<int>((size_t)F * T);
    b.slice_counts = c.take<int>((size_t)F * bin_slices(F) * T);
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
    // direct emission of the exchange's wire format (amav_raster_args.wire), or wire_header == NULL
    int *wire_header, *wire_frame_counts, *wire_offsets;
    unsigned char *wire_payload;
    int wire_cap;
    unsigned wire_bg;
    int stash;   // fused binning: the binning records of the frame also go to LDS (8 B each) for the key scatter
    int slices;  // few-frame shards: blocks per frame of slice_count_kernel / slice_scatter_kernel (bin_slices)
    Buffers buf;
};

__device__ __forceinline__ const float *at(const amav_attr &a, int f, int i) {
    return a.ptr + (long long)f * a.frame_stride + (long long)i * a.elem_stride;
}

// The value of `x`, opaque to the optimiser: what is derived from the result cannot be hoisted out of the enclosing loop.
// The persistent kernels below are one long loop around a lot of inlined code; left alone, the compiler hoists every
// lane-derived address and predicate out of it and then spills them (90 registers to scratch, whose reloads are
// vector-memory operations that queue behind the wave's stores): recomputing them costs a few instructions per tile.
__device__ __forceinline__ int opaque(int x) {
    asm volatile("" : "+v"(x));
    return x;
}
// The lane id, recomputed where it is asked for (two instructions): kept in a register across the blend kernel's tile loop
// it was spilled, and its reload -- a scratch load, i.e. a vector-memory operation -- waited for the wave's stores.
__device__ __forceinline__ int lane_now() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
// A zero the optimiser cannot hoist (and then spill): materialised where it is used.
__device__ __forceinline__ unsigned zero_now() {
    unsigned z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    return z;
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

// A frame's camera, read ONCE per block into scalar registers.  Passed as pointers, the binning loop re-read the two
// matrices from memory for every Gaussian (the compiler cannot prove that the loop's stores leave them alone): eight
// vector loads per iteration that queued behind the previous iteration's stores -- a wave's vector-memory operations
// complete in order -- and, waited for with vmcnt(0), drained the prefetch of the next record as well.
struct FrameCamera {
    float vm[16], pm[16];  // column-major view / projection (as the rasterizer's arguments hold them)
    float tanx, tany;
};
__device__ __forceinline__ FrameCamera load_camera(const Params &p, int f) {
    FrameCamera c;
    auto uniform = [](float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); };
#pragma unroll
    for (int k = 0; k < 16; ++k) c.vm[k] = uniform(p.view[f * 16 + k]), c.pm[k] = uniform(p.proj[f * 16 + k]);
    c.tanx = uniform(p.tanfov[2 * f]), c.tany = uniform(p.tanfov[2 * f + 1]);
    return c;
}

// Returns the binning record {tile box, depth bits, radius} (radius 0 = culled) and, for a kept Gaussian, its blend record
// in g[0..2] (the caller stores it to buf.geom).
__device__ __forceinline__ uint4 preprocess_one(const Params &p, const GaussRec &rec, const FrameCamera &cam,
                                                int &upstream_tiles, float4 (&g)[3]) {
#pragma clang fp contract(off)
    const float *vm = cam.vm, *pm = cam.pm;
    const float tanx = cam.tanx, tany = cam.tany;
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
    // upstream skips det == 0.  det < 0 cannot happen for a real covariance (J W Sigma W^T J^T is positive
    // semi-definite and 0.3 is added to its diagonal); a record that gets there by overflow / NaN inputs is dropped too
    // (upstream would blend an indefinite form), which is what lets the blend kernel use the square-root form below.
    if (!(det > 0.0f)) return rd;
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
    // The blend needs log2(alpha) = log2(op) + log2(e) * power, power = -1/2 (A dx^2 + C dy^2) - B dx dy with the conic
    // (A, B, C) = (cc, -cb, ca) / det.  The quadratic form is stored as its Cholesky factor: with k = log2(e) / 2,
    //     -log2(e) * power = (a dx + b dy)^2 + (c dy)^2,   a = sqrt(k A), b = k B / a, c = sqrt(k (C - B^2 / A)) = sqrt(k / cc)
    // (A C - B^2 = 1 / det).  A sum of squares is >= 0 in floating point too, so upstream's "power > 0 -> skip" guard
    // (which only ever fires on rounding noise of ITS three-term form) has nothing left to catch, and the blend kernel
    // evaluates log2(alpha) in five fused multiply-adds (blend_px).
    const float kk = 0.5f * kLog2e;
    const float qa = sqrtf(kk * (cc * det_inv));
    const float qb = -(kk * (cb * det_inv)) / qa;
    const float qc = sqrtf(kk / cc);
    g[0] = make_float4(pix_x, pix_y, qa, qb);
    g[1] = make_float4(qc, log2f(op), c0, c1);
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

// Projection of every Gaussian of every frame (preprocess_one): one thread per (frame, Gaussian), any number of frames
// fills the chip.  Round 2 ran this inside the per-frame binning block: with the reference's own window (6 frames of
// 30 000 Gaussians) six of the 256 CUs did ~1200 instructions per Gaussian while the others idled.
template <bool kPacked>
__global__ __launch_bounds__(256) void preprocess_kernel(Params p) {
    // grid = (blocks per frame, frames): the frame is block-uniform, so its camera is read with scalar loads
    const int f = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    int up = 0;
    if (i < p.N) {
        const GaussRec rec = load_gaussian<kPacked>(p, f, i);
        float4 g[3];
        const uint4 rd = preprocess_one(p, rec, load_camera(p, f), up, g);
        const size_t gi = (size_t)f * p.N + i;
        p.buf.rectd[gi] = rd;
        if (p.out_radii) p.out_radii[gi] = (int)rd.w;
        if (rd.w) {
            float4 *dst = p.buf.geom + gi * 3;
            dst[0] = g[0], dst[1] = g[1], dst[2] = g[2];
        }
    }
    // upstream's instance count (3-sigma rectangles), summed per block
    __shared__ int part[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) up += __shfl_xor(up, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = up;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int sum = part[0] + part[1] + part[2] + part[3];
        if (sum) atomicAdd(reinterpret_cast<unsigned long long *>(&p.buf.status->total), (unsigned long long)sum);
    }
}

// grid = F blocks of 1024 threads; dynamic LDS = (2*T + 48 + 3*8*(kBuckets+1)) ints: counts[T], cursor[T], classes, scratch[48]
// kFused: the block also projects its frame's Gaussians (shards of >= kFusedMinFrames frames: every CU has a frame, and
// the projection's memory traffic hides under the counting); otherwise preprocess_kernel has done that for all frames.
template <bool kFused, bool kPacked>
__global__ __launch_bounds__(1024) void bin_kernel(Params p) {
    extern __shared__ int bin_lds[];
    int *counts = bin_lds;
    int *cursor = bin_lds + p.T;
    int *scratch = bin_lds + 2 * p.T + 3 * kQueues * (kBuckets + 1);
    uint2 *stash = reinterpret_cast<uint2 *>(scratch + 48);  // [N] when p.stash (8-byte aligned: 2 T + 480 ints in front)
    const int f = blockIdx.x;
#ifdef AMAV_BIN_STAMPS  /* diagnostic build (tools/stamp_bin.sh): block-level phase stamps behind the blend kernel's */
#define AMAV_BIN_STAMP(k)                                                                                        \
    do {                                                                                                         \
        __syncthreads();                                                                                         \
        if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)p.F * p.T * 6 + (size_t)f * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define AMAV_BIN_STAMP(k)
#endif
    AMAV_BIN_STAMP(0);
    for (int t = threadIdx.x; t < p.T; t += blockDim.x) counts[t] = 0;
    __syncthreads();

    // phase 1: count instances per tile (LDS atomics)
    if (kFused) {
        const FrameCamera cam = load_camera(p, f);
        int upstream = 0;
        // Two Gaussians per thread and round, the NEXT round's two records requested before this round's results are
        // stored.  A wave's vector-memory operations complete in order, so a wait for a load is also a wait for every
        // store issued before it, and a store takes ~9 us to be acknowledged on a chip that is writing: with the plain
        // order (store the result, fetch the next record, wait for it) every Gaussian paid for its predecessor's
        // stores (tools/stamp_bin.sh at 250 x 10 000: this phase 88 us; 50.6 us without loads and stores, 52.9 with the
        // stores only, 59.0 with the loads only).  Here the wait at the end of a round is for loads that were issued
        // BEFORE the round's stores, and the stores it has to cover are a whole round old: 78 us.  For that to work the
        // count of stores between a load and its wait must be the same on every path, so that `s_waitcnt vmcnt(8)` can
        // step over them: every lane stores every time -- culled Gaussians a zero record (nobody reads it), lanes past
        // the end of the frame into a spare slot behind the last frame.
        auto pin = [](GaussRec &r) {  // the values count as produced here: the compiler's wait for the loads lands here
            asm volatile(""
                         : "+v"(r.r0.x), "+v"(r.r0.y), "+v"(r.r0.z), "+v"(r.r0.w), "+v"(r.r1.x), "+v"(r.r1.y), "+v"(r.r1.z),
                           "+v"(r.r1.w), "+v"(r.r2.x), "+v"(r.r2.y), "+v"(r.r2.z), "+v"(r.r3.x), "+v"(r.r3.y), "+v"(r.r3.z));
        };
        const int B = blockDim.x;
        auto project = [&](const GaussRec &rec, int i) {
            int up = 0;
            float4 g[3] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
            uint4 rd = preprocess_one(p, rec, cam, up, g);
            const bool real = i < p.N;
            if (!real) rd = make_uint4(0u, 0u, 0u, 0u), up = 0;
            upstream += up;
#if AMAV_BIN_ABLATE != 1
            if (rd.w) {
                const int rx0 = rd.x & 0xffff, ry0 = rd.x >> 16, rx1 = rd.y & 0xffff, ry1 = rd.y >> 16;
                for (int ty = ry0; ty < ry1; ++ty)
                    for (int tx = rx0; tx < rx1; ++tx) atomicAdd(&counts[ty * p.gx + tx], 1);
            }
#endif
            if (p.stash && real)  // tile box (8 bits per bound) + depth bits: what the key scatter needs, without a reload
                stash[i] = make_uint2((rd.x & 0xff) | ((rd.x >> 16) << 8) | ((rd.y & 0xff) << 16) | ((rd.y >> 16) << 24), rd.z);
            const size_t gi = real ? (size_t)f * p.N + i : (size_t)p.F * p.N + threadIdx.x;  // spare slots: carve()
            p.buf.rectd[gi] = rd;
            float4 *dst = p.buf.geom + gi * 3;
            dst[0] = g[0], dst[1] = g[1], dst[2] = g[2];
        };
        GaussRec a0 = load_gaussian<kPacked>(p, f, min((int)threadIdx.x, p.N - 1));
        GaussRec a1 = load_gaussian<kPacked>(p, f, min((int)threadIdx.x + B, p.N - 1));
        for (int base = 0; base < p.N; base += 2 * B) {  // block-uniform trip count
            const int i = base + threadIdx.x;
            GaussRec n0 = load_gaussian<kPacked>(p, f, min(i + 2 * B, p.N - 1));
            GaussRec n1 = load_gaussian<kPacked>(p, f, min(i + 3 * B, p.N - 1));
            project(a0, i);
            project(a1, i + B);
            pin(n0);
            pin(n1);
            a0 = n0, a1 = n1;
        }
        int upstream_total;
        block_exclusive_scan(upstream, scratch, &upstream_total);
        if (threadIdx.x == 0)
            atomicAdd(reinterpret_cast<unsigned long long *>(&p.buf.status->total), (unsigned long long)upstream_total);
    } else {
        // the frame was counted in slices (slice_count_kernel): sum them, and leave in every slice's entry the number of
        // instances the slices before it put on the tile (slice_scatter_kernel starts its cursors there)
        int *sc = p.buf.slice_counts + (size_t)f * p.slices * p.T;
        for (int t = threadIdx.x; t < p.T; t += blockDim.x) {
            int run = 0;
            for (int b = 0; b < p.slices; ++b) {
                const int c = sc[(size_t)b * p.T + t];
                sc[(size_t)b * p.T + t] = run;
                run += c;
            }
            counts[t] = run;
        }
    }
    __syncthreads();
    AMAV_BIN_STAMP(1);

    // phase 2: exclusive scan of the tile counters -> list offsets inside this frame's instance region
    const int per = (p.T + (int)blockDim.x - 1) / (int)blockDim.x;
    const int t0 = threadIdx.x * per;
    int local = 0;
    for (int k = 0; k < per; ++k)
        if (t0 + k < p.T) local += counts[t0 + k];
    int total;
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
    AMAV_BIN_STAMP(2);
    // slots of this frame's non-empty tiles in the wire buffer's payload: in tile order inside the frame, the frame's
    // range reserved with one atomic on the wire header's count (frames land in completion order; readers go through
    // the offsets table, so the order is immaterial)
    int wire_slot = 0;
    if (p.wire_header) {
        int mine = 0;
        for (int k = 0; k < per; ++k)
            if (t0 + k < p.T && fits && counts[t0 + k] > 0) ++mine;
        int frame_tiles;
        const int before = block_exclusive_scan(mine, scratch, &frame_tiles);
        if (threadIdx.x == 0) {
            scratch[32] = atomicAdd(&p.wire_header[1], frame_tiles);
            p.wire_frame_counts[f] = frame_tiles;
            if (f == 0) {
                p.wire_header[0] = kWireMagic, p.wire_header[2] = p.wire_cap, p.wire_header[3] = p.F;
                p.wire_header[4] = p.T, p.wire_header[5] = p.H, p.wire_header[6] = p.W, p.wire_header[7] = (int)p.wire_bg;
            }
        }
        __syncthreads();
        wire_slot = scratch[32] + before;
        __syncthreads();
    }
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
            if (p.wire_header) p.wire_offsets[(size_t)f * p.T + t0 + k] = kind == kBuckets ? -1 : wire_slot;
            if (kind == kBuckets)
                p.buf.empty_list[pos] = f * p.T + t0 + k;
            else
                p.buf.queue[((size_t)qi * kBuckets + kind) * p.qcap + pos] =
                    make_int4(f * p.T + t0 + k, cursor[t0 + k], c, p.wire_header ? wire_slot++ : 0);
        }
    if (threadIdx.x == 0) {
        off[p.T] = total;
        atomicAdd(reinterpret_cast<unsigned long long *>(&p.buf.status->emitted), (unsigned long long)total);
        atomicMax(reinterpret_cast<unsigned long long *>(&p.buf.status->max_frame), (unsigned long long)total);
        if (!fits) atomicExch(&p.buf.status->overflow, 1);
    }
    __syncthreads();
    if (!fits) return;
    AMAV_BIN_STAMP(3);
    if (!kFused) return;  // few-frame shards: slice_scatter_kernel writes the keys

    // phase 3: scatter (depth, index) keys into the tile lists (order inside a list is fixed later by the sort)
    unsigned long long *keys = p.buf.keys + (size_t)f * p.cap_per_frame;
    // The binning records of kScatterChunk Gaussians per thread are read up front and waited for ONCE, before the first
    // key store: a load issued after the key stores of the previous Gaussian would wait for their acknowledgement
    // (in-order vector memory, see phase 1) once per Gaussian; this way once per chunk, i.e. not at all up to 12 288
    // Gaussians per frame.
    if (kFused && p.stash) {
        // the frame's binning records are in LDS (written by phase 1): no reload -- a load here would first wait for
        // the acknowledgement of phase 2's stores (queues, offsets: 12 of this pass's 36 us)
        for (int i = threadIdx.x; i < p.N; i += blockDim.x) {
            const uint2 e = stash[i];
            const int rx0 = e.x & 0xff, ry0 = (e.x >> 8) & 0xff, rx1 = (e.x >> 16) & 0xff, ry1 = e.x >> 24;
            const unsigned long long key = ((unsigned long long)e.y << 32) | (unsigned)i;
            for (int ty = ry0; ty < ry1; ++ty)
                for (int tx = rx0; tx < rx1; ++tx) keys[atomicAdd(&cursor[ty * p.gx + tx], 1)] = key;
        }
        AMAV_BIN_STAMP(4);
        return;
    }
    const uint4 *rect = p.buf.rectd + (size_t)f * p.N;
    constexpr int kScatterChunk = 12;
    for (int base = threadIdx.x; base < p.N; base += kScatterChunk * (int)blockDim.x) {
        uint4 rds[kScatterChunk];
#pragma unroll
        for (int k = 0; k < kScatterChunk; ++k) rds[k] = rect[min(base + k * (int)blockDim.x, p.N - 1)];
#pragma unroll
        for (int k = 0; k < kScatterChunk; k += 4)
            asm volatile("s_waitcnt vmcnt(0)"
                         : "+v"(rds[k].x), "+v"(rds[k].y), "+v"(rds[k].z), "+v"(rds[k].w), "+v"(rds[k + 1].x), "+v"(rds[k + 1].y),
                           "+v"(rds[k + 1].z), "+v"(rds[k + 1].w), "+v"(rds[k + 2].x), "+v"(rds[k + 2].y), "+v"(rds[k + 2].z),
                           "+v"(rds[k + 2].w), "+v"(rds[k + 3].x), "+v"(rds[k + 3].y), "+v"(rds[k + 3].z), "+v"(rds[k + 3].w));
#pragma unroll
        for (int k = 0; k < kScatterChunk; ++k) {
            const int i = base + k * (int)blockDim.x;
            if (i >= p.N) break;
            const uint4 rd = rds[k];
            if (kFused && p.out_radii) p.out_radii[(size_t)f * p.N + i] = (int)rd.w;
            if (rd.w == 0u) continue;
            const int rx0 = rd.x & 0xffff, ry0 = rd.x >> 16, rx1 = rd.y & 0xffff, ry1 = rd.y >> 16;
            const unsigned long long key = ((unsigned long long)rd.z << 32) | (unsigned)i;
            for (int ty = ry0; ty < ry1; ++ty)
                for (int tx = rx0; tx < rx1; ++tx) keys[atomicAdd(&cursor[ty * p.gx + tx], 1)] = key;
        }
    }
    AMAV_BIN_STAMP(4);
}

// Few-frame shards, pass 1 of 3: grid (slices, F); the block counts the instances of its slice of the frame's Gaussians
// per tile (LDS atomics over the binning records of preprocess_kernel) and leaves the counts in slice_counts.
__global__ __launch_bounds__(1024) void slice_count_kernel(Params p) {
    extern __shared__ int slice_lds[];
    int *counts = slice_lds;
    const int b = blockIdx.x, f = blockIdx.y;
    for (int t = threadIdx.x; t < p.T; t += blockDim.x) counts[t] = 0;
    __syncthreads();
    const int per = (p.N + p.slices - 1) / p.slices;
    const int i1 = min(p.N, (b + 1) * per);
    const uint4 *rect = p.buf.rectd + (size_t)f * p.N;
    for (int i = b * per + threadIdx.x; i < i1; i += blockDim.x) {
        const uint4 rd = rect[i];
        if (rd.w) {
            const int rx0 = rd.x & 0xffff, ry0 = rd.x >> 16, rx1 = rd.y & 0xffff, ry1 = rd.y >> 16;
            for (int ty = ry0; ty < ry1; ++ty)
                for (int tx = rx0; tx < rx1; ++tx) atomicAdd(&counts[ty * p.gx + tx], 1);
        }
    }
    __syncthreads();
    int *out = p.buf.slice_counts + ((size_t)f * p.slices + b) * p.T;
    for (int t = threadIdx.x; t < p.T; t += blockDim.x) out[t] = counts[t];
}

// Pass 3 of 3 (pass 2 is bin_kernel<false, .>: sums the slices, scans the tiles, builds the queues): the block scatters
// the keys of its slice, its cursors starting behind the slices before it.  A frame that overflowed its region
// (tile_off[T] = its instance count) is skipped, as the one-block form does.
__global__ __launch_bounds__(1024) void slice_scatter_kernel(Params p) {
    extern __shared__ int slice_lds[];
    int *cursor = slice_lds;
    const int b = blockIdx.x, f = blockIdx.y;
    const int *off = p.buf.tile_off + (size_t)f * (p.T + 1);
    if ((long long)off[p.T] > p.cap_per_frame) return;
    const int *before = p.buf.slice_counts + ((size_t)f * p.slices + b) * p.T;
    for (int t = threadIdx.x; t < p.T; t += blockDim.x) cursor[t] = off[t] + before[t];
    __syncthreads();
    const int per = (p.N + p.slices - 1) / p.slices;
    const int i1 = min(p.N, (b + 1) * per);
    unsigned long long *keys = p.buf.keys + (size_t)f * p.cap_per_frame;
    const uint4 *rect = p.buf.rectd + (size_t)f * p.N;
    // binning records first, key stores after (one wait per chunk: see bin_kernel's scatter)
    constexpr int kChunk = 8;
    for (int base = b * per + threadIdx.x; base < i1; base += kChunk * (int)blockDim.x) {
        uint4 rds[kChunk];
#pragma unroll
        for (int k = 0; k < kChunk; ++k) rds[k] = rect[min(base + k * (int)blockDim.x, i1 - 1)];
#pragma unroll
        for (int k = 0; k < kChunk; k += 4)
            asm volatile("s_waitcnt vmcnt(0)"
                         : "+v"(rds[k].x), "+v"(rds[k].y), "+v"(rds[k].z), "+v"(rds[k].w), "+v"(rds[k + 1].x), "+v"(rds[k + 1].y),
                           "+v"(rds[k + 1].z), "+v"(rds[k + 1].w), "+v"(rds[k + 2].x), "+v"(rds[k + 2].y), "+v"(rds[k + 2].z),
                           "+v"(rds[k + 2].w), "+v"(rds[k + 3].x), "+v"(rds[k + 3].y), "+v"(rds[k + 3].z), "+v"(rds[k + 3].w));
#pragma unroll
        for (int k = 0; k < kChunk; ++k) {
            const int i = base + k * (int)blockDim.x;
            if (i >= i1) break;
            const uint4 rd = rds[k];
            if (rd.w == 0u) continue;
            const int rx0 = rd.x & 0xffff, ry0 = rd.x >> 16, rx1 = rd.y & 0xffff, ry1 = rd.y >> 16;
            const unsigned long long key = ((unsigned long long)rd.z << 32) | (unsigned)i;
            for (int ty = ry0; ty < ry1; ++ty)
                for (int tx = rx0; tx < rx1; ++tx) keys[atomicAdd(&cursor[ty * p.gx + tx], 1)] = key;
        }
    }
}

// Wave-wide min / max / inclusive sum WITHOUT ds_bpermute: __shfl_xor / __shfl_up keep one address register per
// distance alive (the compiler hoists (lane ^ o) << 2 out of the tile loop: thirteen registers), and what did not fit
// was spilled -- reloaded per tile as scratch loads, i.e. vector-memory operations whose s_waitcnt vmcnt(0) also drains
// every background store issued before them.  DPP permutes inside rows of 16 lanes need no address; the four rows are
// combined on the scalar unit (the results are wave-uniform anyway).
template <int kCtrl>
__device__ __forceinline__ unsigned dpp_permute(unsigned v) {  // every lane has a source for these controls
    return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, kCtrl, 0xf, 0xf, false);
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {  // all 64 lanes active
    v = min(v, dpp_permute<0xB1>(v));   // quad_perm [1,0,3,2]
    v = min(v, dpp_permute<0x4E>(v));   // quad_perm [2,3,0,1]
    v = min(v, dpp_permute<0x141>(v));  // row_half_mirror
    v = min(v, dpp_permute<0x140>(v));  // row_mirror: every lane of a row holds the row's minimum
    const unsigned a = __builtin_amdgcn_readlane((int)v, 0), b = __builtin_amdgcn_readlane((int)v, 16);
    const unsigned c = __builtin_amdgcn_readlane((int)v, 32), d = __builtin_amdgcn_readlane((int)v, 48);
    return min(min(a, b), min(c, d));
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    v = max(v, dpp_permute<0xB1>(v));
    v = max(v, dpp_permute<0x4E>(v));
    v = max(v, dpp_permute<0x141>(v));
    v = max(v, dpp_permute<0x140>(v));
    const unsigned a = __builtin_amdgcn_readlane((int)v, 0), b = __builtin_amdgcn_readlane((int)v, 16);
    const unsigned c = __builtin_amdgcn_readlane((int)v, 32), d = __builtin_amdgcn_readlane((int)v, 48);
    return max(max(a, b), max(c, d));
}
template <int kShift>
__device__ __forceinline__ unsigned dpp_row_shr(unsigned v) {  // lane i of a row <- lane i - kShift, 0 at the row's start
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 + kShift, 0xf, 0xf, false);
}
__device__ __forceinline__ unsigned wave_inclusive_sum_u32(unsigned v, int lane) {  // all 64 lanes active
    v += dpp_row_shr<1>(v);
    v += dpp_row_shr<2>(v);
    v += dpp_row_shr<4>(v);
    v += dpp_row_shr<8>(v);  // inclusive sums inside every row of 16
    const unsigned r0 = __builtin_amdgcn_readlane((int)v, 15), r1 = __builtin_amdgcn_readlane((int)v, 31);
    const unsigned r2 = __builtin_amdgcn_readlane((int)v, 47);
    return v + (lane >= 16 ? r0 : 0u) + (lane >= 32 ? r1 : 0u) + (lane >= 48 ? r2 : 0u);
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
            dmin = wave_min_u32(dmin), dmax = wave_max_u32(dmax);
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
            const unsigned incl = wave_inclusive_sum_u32(tot, lane);
            big = wave_max_u32(big);
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

constexpr int kNullSlot = 64;  // staging slot of the null record (log2(opacity) = -inf): what list padding points at

struct WaveLds {
    // keys of the tile's list while it is sorted, then (in place) the blend order as 32-bit Gaussian ids in the first
    // 2 KiB and, behind them, the four quadrant lists of the current staging round
    unsigned long long keys[kSortCap];
    // staged records, [64] = null record.  Three planes at fixed distances (immediate offsets of the blend loop's reads):
    float4 geo[65];   // +0     {k0, k1, qa, qb}
    float4 col[65];   // +1040  {r, g, b, 1/depth}
    float4 geo2[65];  // +2080  {qc, log2(opacity), -, -}: 16-byte slots like the other planes (one slot address serves all three)
    int bucket_end[32];  // the queue's bucket table: entry b = queue positions before the end of bucket b (render_kernel)
};
static_assert(offsetof(WaveLds, col) - offsetof(WaveLds, geo) == 1040 && offsetof(WaveLds, geo2) - offsetof(WaveLds, geo) == 2080,
              "blend_quadrant's ds_read offsets");
// quadrant lists: LDS byte addresses of the staged records that reach the quadrant, in blend order; blend_quadrant walks
// them two entries at a time and reads up to two pairs past the end, so the tail is padded with the null record's address
constexpr int kListStride = 72;
constexpr int kListTail = 5;  // pad entries behind the last real one: up to entry 2 * ceil(n / 2) + 3
static_assert(64 + kListTail <= kListStride && kListStride % 2 == 0, "list padding");
static_assert(kSortCap * 4 + 4 * kListStride * 4 <= kSortCap * 8, "quadrant lists must fit behind the id list");

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
__device__ __attribute__((noinline)) bool bucket_sort(const unsigned long long *__restrict__ gkeys, unsigned long long *slice,
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
    dmin = wave_min_u32(dmin), dmax = wave_max_u32(dmax);
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
    const unsigned incl = wave_inclusive_sum_u32(tot, lane);
    big = wave_max_u32(big);
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
__device__ __attribute__((noinline)) void wave_bitonic_sort(unsigned long long *a, unsigned *order, int n, int lane) {
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

// ---- the blend loop
// One pixel (lane), one Gaussian.  T > 0: live transmittance; T < 0: pixel finished, |T| is its final transmittance.
// Staged record: geo = {k0, k1, qa, qb}, geo2 = {qc, L = log2(opacity)}, col = {r, g, b, 1/depth}; (lx, ly) = the pixel
// relative to the tile origin.  log2(alpha) = L - u^2 - v^2 with u = qa dx + qb dy = k0 - qa lx - qb ly and
// v = qc dy = k1 - qc ly (preprocess_one; k0, k1 are formed per tile when the record is staged): five fused
// multiply-adds, and never above L, so upstream's "power > 0" skip has no case left.  Then, as upstream:
//     alpha = min(0.99, 2^..);  alpha < 1/255 -> skip;  T' = T (1 - alpha);  T' < 1e-4 -> pixel finished (not blended)
//     C += colour * alpha * T;  T = T'
// A finished pixel needs no test of its own: with T < 0, T - alpha T < 1e-4 holds, the Gaussian takes the "finished"
// branch (weight 0, T <- -|T| = T).  17 vector instructions per pixel and Gaussian (18 with inverse depth).
//
// The loop is written in assembly (one statement, guide section 5.7 form (i): its LDS reads and their waits are all
// inside).  Compiled from C++ the same loop came out at 27 vector + 15 scalar instructions per Gaussian in round 2 (mask
// walking, register copies at the back edge, lgkmcnt(0) before every group) and, rewritten over address lists, lost its
// software pipeline to the scheduler (every record read hoisted to the loop top and waited for at once).
//
// Registers: v68..v95 are named literally and listed as clobbers (two record sets of ten, two pairs of list entries,
// three temporaries); the state (T, R, G, B, D) and inputs are ordinary operands.
// Pipeline (LDS returns in order, so every wait is a count): iteration k blends pair k = records A, B while it
//   * reads the list entries of pair k+2 (top of the iteration),
//   * re-reads each record's registers for pair k+1 as soon as the last instruction using them has issued: the six
//     geometry words after the fourth multiply-add, the colour words after the colour multiply-adds,
// so a read is issued about one Gaussian (~20 instructions of this wave, times the waves sharing the SIMD) before its
// use.  At the top of an iteration the queue holds [entries(k+1)] A.geo A.geo2 A.col B.geo B.geo2 B.col + the new
// entries(k+2) = 8 reads; the waits below are lgkmcnt(5) before each record's geometry and lgkmcnt(6) before its colours.
// Software-managed hazards of gfx950 respected inside the string: a VALU write of VCC needs two instructions before
// the VALU that reads it as a mask; a transcendental's result one before its VALU consumer.
#define AMAV_BLEND_GEO(k0, k1, qa, qb, qc, L) /* u -> v92 = L - u^2 (v^2 still to subtract), v -> v93 */          \
    "v_fma_f32 v92, -" qb ", %[ly], " k0 "\n\t"                                                                    \
    "v_fma_f32 v93, -" qc ", %[ly], " k1 "\n\t"                                                                    \
    "v_fma_f32 v92, -" qa ", %[lx], v92\n\t"                                                                       \
    "v_fma_f32 v92, -v92, v92, " L "\n\t"
#define AMAV_BLEND_REST(r, g, b, d, col_wait, DEPTH)                                                               \
    "v_fma_f32 v92, -v93, v93, v92\n\t"                                                                            \
    "v_exp_f32_e32 v92, v92\n\t"                                                                                   \
    "s_nop 0\n\t"                                                                                                  \
    "v_min_f32_e32 v92, 0x3f7d70a4, v92\n\t"            /* min(0.99, .) */                                        \
    "v_cmp_le_f32_e32 vcc, 0x3b808081, v92\n\t"         /* 1/255 <= alpha */                                      \
    "s_nop 1\n\t"                                                                                                  \
    "v_cndmask_b32_e32 v92, 0, v92, vcc\n\t"            /* else alpha = 0 (also NaN) */                           \
    "v_fma_f32 v93, -v92, %[T], %[T]\n\t"               /* T' = T - alpha T */                                    \
    "v_cmp_gt_f32_e32 vcc, 0x38d1b717, v93\n\t"         /* T' < 1e-4: finished */                                 \
    "v_mul_f32_e32 v94, v92, %[T]\n\t"                  /* weight alpha T */                                      \
    "s_nop 0\n\t"                                                                                                  \
    "v_cndmask_b32_e64 v94, v94, 0, vcc\n\t"                                                                       \
    "v_cndmask_b32_e64 %[T], v93, -|%[T]|, vcc\n\t"                                                                \
    col_wait                                                                                                       \
    "v_fmac_f32_e32 %[R], " r ", v94\n\t"                                                                          \
    "v_fmac_f32_e32 %[G], " g ", v94\n\t"                                                                          \
    "v_fmac_f32_e32 %[B], " b ", v94\n\t" DEPTH(d)
#define AMAV_DEPTH_ON(d) "v_fmac_f32_e32 %[D], " d ", v94\n\t"
#define AMAV_DEPTH_OFF(d)
// one pair: records A = v[68:77] (geo 68..71, geo2 72..73, col 74..77), B = v[78:87]; `use` = the register pair holding
// the entries of the next pair (addresses of its records), `load` = the pair that receives the entries after that
#define AMAV_BLEND_PAIR(use_x, use_y, load, DEPTH)                                                                 \
    "ds_read_b64 " load ", %[list] offset:16\n\t"                                                                  \
    "v_add_u32_e32 %[list], 8, %[list]\n\t"                                                                        \
    "s_waitcnt lgkmcnt(5)\n\t"                                                                                     \
    AMAV_BLEND_GEO("v68", "v69", "v70", "v71", "v72", "v73")                                                       \
    "ds_read_b128 v[68:71], " use_x "\n\t"                                                                         \
    "ds_read_b64 v[72:73], " use_x " offset:2080\n\t"                                                              \
    AMAV_BLEND_REST("v74", "v75", "v76", "v77", "s_waitcnt lgkmcnt(6)\n\t", DEPTH)                                 \
    "ds_read_b128 v[74:77], " use_x " offset:1040\n\t"                                                             \
    "s_waitcnt lgkmcnt(5)\n\t"                                                                                     \
    AMAV_BLEND_GEO("v78", "v79", "v80", "v81", "v82", "v83")                                                       \
    "ds_read_b128 v[78:81], " use_y "\n\t"                                                                         \
    "ds_read_b64 v[82:83], " use_y " offset:2080\n\t"                                                              \
    AMAV_BLEND_REST("v84", "v85", "v86", "v87", "s_waitcnt lgkmcnt(6)\n\t", DEPTH)                                 \
    "ds_read_b128 v[84:87], " use_y " offset:1040\n\t"
#define AMAV_BLEND_LOOP(DEPTH)                                                                                     \
    "s_waitcnt lgkmcnt(0)\n\t"                          /* nothing of the compiler's may be in flight: counted waits */ \
    "ds_read_b64 v[88:89], %[list]\n\t"                                                                            \
    "ds_read_b64 v[90:91], %[list] offset:8\n\t"                                                                   \
    "s_waitcnt lgkmcnt(1)\n\t"                                                                                     \
    "ds_read_b128 v[68:71], v88\n\t"                                                                               \
    "ds_read_b64 v[72:73], v88 offset:2080\n\t"                                                                    \
    "ds_read_b128 v[74:77], v88 offset:1040\n\t"                                                                   \
    "ds_read_b128 v[78:81], v89\n\t"                                                                               \
    "ds_read_b64 v[82:83], v89 offset:2080\n\t"                                                                    \
    "ds_read_b128 v[84:87], v89 offset:1040\n\t"                                                                   \
    "1:\n\t"                                                                                                       \
    AMAV_BLEND_PAIR("v90", "v91", "v[88:89]", DEPTH)                                                               \
    "s_sub_u32 %[groups], %[groups], 1\n\t"                                                                        \
    "s_cmp_eq_u32 %[groups], 0\n\t"                                                                                \
    "s_cbranch_scc1 2f\n\t"                                                                                        \
    AMAV_BLEND_PAIR("v88", "v89", "v[90:91]", DEPTH)                                                               \
    "v_cmp_lt_f32_e32 vcc, 0, %[T]\n\t"                 /* some pixel of the quadrant still takes Gaussians */    \
    "s_sub_u32 %[groups], %[groups], 1\n\t"                                                                        \
    "s_cmp_eq_u32 %[groups], 0\n\t"                                                                                \
    "s_cbranch_scc1 2f\n\t"                                                                                        \
    "s_cbranch_vccnz 1b\n\t"                                                                                       \
    "2:\n\t"                                                                                                       \
    "s_waitcnt lgkmcnt(0)\n\t"                          /* the reads issued ahead land before the registers are reused */
#define AMAV_BLEND_CLOBBERS                                                                                        \
    "memory", "vcc", "scc", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79",    \
        "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94"

// One 8x8 quadrant of the tile (one pixel per lane) against the staged Gaussians of its list: `list` = LDS byte address
// of the list, `groups` = ceil(entries / 2) >= 1.  Returns false when every pixel of the quadrant has finished.
template <bool kInvDepth>
__device__ __forceinline__ bool blend_quadrant(unsigned list, int groups, float lx, float ly, float &T, float &R,
                                               float &G, float &B, float &D) {
    if (kInvDepth)
        asm volatile(AMAV_BLEND_LOOP(AMAV_DEPTH_ON)
                     : [T] "+v"(T), [R] "+v"(R), [G] "+v"(G), [B] "+v"(B), [D] "+v"(D), [list] "+v"(list), [groups] "+s"(groups)
                     : [lx] "v"(lx), [ly] "v"(ly)
                     : AMAV_BLEND_CLOBBERS);
    else
        asm volatile(AMAV_BLEND_LOOP(AMAV_DEPTH_OFF)
                     : [T] "+v"(T), [R] "+v"(R), [G] "+v"(G), [B] "+v"(B), [list] "+v"(list), [groups] "+s"(groups)
                     : [lx] "v"(lx), [ly] "v"(ly)
                     : AMAV_BLEND_CLOBBERS);
    return __any(T > 0.f);
}

#define AMAV_STAMP(slot)                                                                              \
    do {                                                                                              \
        if (p.stamps && lane == 0) p.stamps[(size_t)item * 6 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)

typedef float nt_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_nt(float *rgba, long long pid, float4 v) {  // write-once output: non-temporal
    const nt_f4 x = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(x, reinterpret_cast<nt_f4 *>(rgba) + pid);
}

// Background for the tiles without Gaussians: colour = bg, alpha = 0 (and inverse depth 0).  A wave owns the slice
// [e0, e1) of the empty list and writes it a few tiles at a time between its blended tiles.  The tile ids are fetched
// 64 at a time (one per lane) and decoded in the lanes, so a tile costs two scalar reads of lane registers and its
// stores -- round 2 loaded every id just before its tile: one dependent global round trip per 4 KiB written, 40 of them
// in a row per wave, which is what kept the stores from hiding under the blending (the kernel ran 0.12 ms longer with
// the background than without: the full fill time of the 0.85 GB at the chip's fill rate).
struct FillCursor {
    int next, end;     // slice of the empty list not yet fetched
    int have, pos;     // lanes of the current batch, next lane to write
    long long origin;  // per lane: pixel index of the tile's first pixel
    int xy;            // per lane: X0 | Y0 << 16
};

__device__ __forceinline__ void fill_fetch(const Params &p, FillCursor &c, int lane) {
    c.have = min(64, c.end - c.next);
    c.pos = 0;
    if (c.have <= 0) return;
    const int item = p.buf.empty_list[c.next + min(lane, c.have - 1)];
    c.next += c.have;
    const int f = item / p.T, t = item - f * p.T;
    const int ty = t / p.gx, tx = t - ty * p.gx;
    c.xy = (tx * kTile) | ((ty * kTile) << 16);
    c.origin = ((long long)f * p.H + ty * kTile) * p.W + tx * kTile;
}

// up to `count` tiles of the wave's slice
template <bool kInvDepth>
__device__ __forceinline__ void fill_some(const Params &p, FillCursor &c, int count, int lane) {
    float r = p.bg[0], g = p.bg[1], bl = p.bg[2];
    if (p.clamp_output) {
        r = fminf(fmaxf(r, 0.f), 1.f);
        g = fminf(fmaxf(g, 0.f), 1.f);
        bl = fminf(fmaxf(bl, 0.f), 1.f);
    }
    const float4 px_bg = make_float4(r, g, bl, 0.0f);
    lane = opaque(lane);
    // lane -> (column lane & 15, rows (lane >> 4) + 4k): every store instruction covers four full 256-byte tile rows
    const int cx = lane & 15, cy = lane >> 4;
    while (count > 0) {
        if (c.pos >= c.have) {
            if (c.next >= c.end) return;
            fill_fetch(p, c, lane);
        }
        const int todo = min(count, c.have - c.pos);
        for (int j = c.pos; j < c.pos + todo; ++j) {
            const int xy = __builtin_amdgcn_readlane(c.xy, j);
            const long long origin = ((long long)__builtin_amdgcn_readlane((int)(c.origin >> 32), j) << 32) |
                                     (unsigned)__builtin_amdgcn_readlane((int)c.origin, j);
            const int X0 = xy & 0xffff, Y0 = xy >> 16;
            const bool in_x = X0 + cx < p.W;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (in_x && Y0 + cy + 4 * k < p.H) {
                    const long long pid = origin + (long long)(cy + 4 * k) * p.W + cx;
                    store_nt(p.out_rgba, pid, px_bg);
                    if (kInvDepth) p.out_inv_depth[pid] = 0.0f;
                }
            }
        }
        c.pos += todo;
        count -= todo;
    }
}

// The next tile of the current batch (the caller checks c.pos < c.have): no load, so it can sit inside the blend rounds.
template <bool kInvDepth>
__device__ __forceinline__ void fill_tile_at(const Params &p, FillCursor &c, int lane) {
    const int j = c.pos++;
    const int xy = __builtin_amdgcn_readlane(c.xy, j);
    const long long origin = ((long long)__builtin_amdgcn_readlane((int)(c.origin >> 32), j) << 32) |
                             (unsigned)__builtin_amdgcn_readlane((int)c.origin, j);
    const int X0 = xy & 0xffff, Y0 = xy >> 16;
    float r = p.bg[0], g = p.bg[1], bl = p.bg[2];
    if (p.clamp_output) {
        r = fminf(fmaxf(r, 0.f), 1.f);
        g = fminf(fmaxf(g, 0.f), 1.f);
        bl = fminf(fmaxf(bl, 0.f), 1.f);
    }
    const int cx = lane & 15, cy = lane >> 4;
    const bool in_x = X0 + cx < p.W;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (in_x && Y0 + cy + 4 * k < p.H) {
            const long long pid = origin + (long long)(cy + 4 * k) * p.W + cx;
            store_nt(p.out_rgba, pid, make_float4(r, g, bl, 0.0f));
            if (kInvDepth) p.out_inv_depth[pid] = 0.0f;
        }
    }
}

// ---- tile preparation under the previous tile
struct TileRef {
    int item, beg, n;  // frame * T + tile, list offset inside the frame's key region, list length (all wave-uniform)
    int slot;          // the tile's slot in the wire buffer's payload (direct emission of the exchange format)
};

constexpr int kPrepCap = 256;   // lists up to this length are sorted while the previous tile is still blending
constexpr int kPrepBuckets = 256;

// Exact (depth, id) order of n <= kPrepCap unique keys held in registers (k[m] = key lane + 64 m, ~0 beyond n): the
// 32-bit ids land in order[0 .. n).  `slice` (n keys) and `order` may alias (ids overwrite the slice once every lane
// has read it); `cnt` = kPrepBuckets / 2 + 1 words of scratch.  Same method as bucket_sort / rank_sort above.
__device__ __forceinline__ void sort_prefetched(const unsigned long long (&k)[4], int n, unsigned long long *slice,
                                                unsigned *cnt, unsigned *order, int lane) {
    if (n <= 64) {
        if (lane < n) slice[lane] = k[0];
        wave_sync();
        rank_sort<1>(slice, order, n, lane);
        return;
    }
    unsigned dmin = 0xffffffffu, dmax = 0u;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const unsigned d = (unsigned)(k[m] >> 32);
        if (lane + 64 * m < n) dmin = min(dmin, d), dmax = max(dmax, d);
    }
    dmin = wave_min_u32(dmin), dmax = wave_max_u32(dmax);
    const float scale = dmax > dmin ? (float)(kPrepBuckets - 1) / (float)(dmax - dmin) : 0.0f;
    int b[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) b[m] = min(kPrepBuckets - 1, (int)((float)((unsigned)(k[m] >> 32) - dmin) * scale));
    {
        const unsigned z = zero_now();
        reinterpret_cast<uint2 *>(cnt)[lane] = make_uint2(z, z);  // 128 words of two 16-bit counts
    }
    wave_sync();
    unsigned pos[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        pos[m] = 0;
        if (lane + 64 * m < n) {
            const int sh = 16 * (b[m] & 1);
            pos[m] = (atomicAdd(&cnt[b[m] >> 1], 1u << sh) >> sh) & 0xffffu;  // arrival index inside the bucket
        }
    }
    wave_sync();
    const uint2 w = reinterpret_cast<uint2 *>(cnt)[lane];  // lane owns buckets 4 lane .. 4 lane + 3
    unsigned c[4] = {w.x & 0xffffu, w.x >> 16, w.y & 0xffffu, w.y >> 16};
    unsigned tot = 0, big = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned ci = c[i];
        big = max(big, ci);
        c[i] = tot;
        tot += ci;
    }
    const unsigned incl = wave_inclusive_sum_u32(tot, lane);
    big = wave_max_u32(big);
    if (big > (unsigned)kBucketMax) {  // wave-uniform: clustered depths, comparison sort
        wave_sync();
#pragma unroll
        for (int m = 0; m < 4; ++m)
            if (lane + 64 * m < n) slice[lane + 64 * m] = k[m];
        wave_sync();
        rank_sort<4>(slice, order, n, lane);
        return;
    }
    const unsigned base = incl - tot;
    reinterpret_cast<uint2 *>(cnt)[lane] = make_uint2((base + c[0]) | ((base + c[1]) << 16), (base + c[2]) | ((base + c[3]) << 16));
    if (lane == 0) cnt[kPrepBuckets / 2] = (unsigned)n;  // start of the bucket past the last one
    wave_sync();
    unsigned st[4], sz[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        st[m] = (cnt[b[m] >> 1] >> (16 * (b[m] & 1))) & 0xffffu;
        const int nb = b[m] + 1;
        sz[m] = ((cnt[nb >> 1] >> (16 * (nb & 1))) & 0xffffu) - st[m];
        if (lane + 64 * m < n) slice[st[m] + pos[m]] = k[m];
    }
    wave_sync();
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        unsigned r = pos[m];
        if (lane + 64 * m < n && sz[m] > 1u) {  // order the bucket by the full key
            r = 0;
            for (unsigned j = 0; j < sz[m]; ++j) r += (unsigned)(slice[st[m] + j] < k[m]);
        }
        pos[m] = st[m] + r;
    }
    wave_sync();  // the ids overwrite the slice: every lane has finished reading it
#pragma unroll
    for (int m = 0; m < 4; ++m)
        if (lane + 64 * m < n) order[pos[m]] = (unsigned)k[m];
    wave_sync();
}

// Blend kernel: PERSISTENT waves (one wave per workgroup, the grid is what the chip holds at once).  The waves of
// queue q (= blockIdx % 8: blocks are dealt round-robin over the XCDs, so an XCD keeps seeing the tile-row bands of
// its own queue -- a placement assumption that only affects speed) walk the queue's buckets, which are ordered
// LONGEST LISTS FIRST, so the kernel ends on the shortest tiles.  The first position of a wave is static; every later
// one comes from the queue's cursor (one returning atomic per tile on one of eight words).
//
// A tile's start-up is a chain of dependent global round trips -- queue position -> queue entry -> keys -> (sort) ->
// records of the first staging round -- and vector-memory operations of a wave complete in order, so a load issued
// behind the wave's stores also waits for those to drain.  Round 2 paid that chain once per tile (18 us against 29 us
// of blending).  Here every link is issued while an EARLIER tile is still blending, and ahead of that tile's stores:
//     tile i starts              atomic for the position of tile i+2
//     first round, after q0      that position has come back: request the queue entry of tile i+2
//     last round, at its start   entry of tile i+1 (requested during tile i-1) is consumed: its keys go to registers
//     last round, after q1       keys sorted in the id area of LDS (tile i has issued its last gather: the area is free),
//                                records of tile i+1's first round requested (the prefetch registers are free as well)
//     after q3                   wait for those records (they had q2 and q3 to arrive), THEN store tile i
// Lists of 257..512 keys need the whole key area and are sorted at the start of their own tile, as is a wave's first tile.
// Measured (250 frames, 512^2, 10 k Gaussians; tools/stamp_render.py): sort + range reads fall from 62 us to 20 us of a
// wave's 500 us.
//
// Registers: the kernel is built for 4 waves per SIMD (128 registers).  At 5 (96 registers, round 2's setting) the
// pipeline state on top of the blend loop's 27 fixed registers spills -- and a spill's reload is a vector-memory load
// behind the stores, i.e. the very stall the pipeline removes (measured: 0.76 ms instead of 0.50).
template <bool kInvDepth>
__global__ __launch_bounds__(64, kRenderWavesPerSimd) void render_kernel(Params p) {
    __shared__ WaveLds lds;
    WaveLds &L = lds;
    const int lane = threadIdx.x;
    const Status *st = p.buf.status;
    const int q = blockIdx.x % kQueues, stride = gridDim.x / kQueues;
    // ---- background: every wave owns a slice of the empty list and writes it a few tiles per staging round, right
    // after the round's records have been consumed and before the next round's are requested.  Vector-memory operations
    // of a wave complete in order, so a wave waits for its stores whenever it next needs a loaded value; placed there,
    // that next wait is a whole round of blending away.  (Measured alternatives: stores at the end of each tile, between
    // quadrants, or by dedicated fill-first waves all cost the full fill time of 0.12 ms -- one wave sustains only
    // ~4 GB/s of stores, 63 operations in flight at the ~15 us write latency of a saturated chip, so the background
    // needs most of the chip's waves, thinly.)
    FillCursor fill;
    {
        const int nempty = st->nempty;
        const int per = (nempty + (int)gridDim.x - 1) / (int)gridDim.x;
        fill.next = min(nempty, (int)blockIdx.x * per);
        fill.end = min(nempty, fill.next + per);
        fill.have = fill.pos = 0;
        fill.origin = 0, fill.xy = 0;
    }
    // The queue's bucket table, read ONCE: lane b holds where bucket b starts and ends among the queue's positions.
    // (Looked up in memory per tile, the walk was a chain of up to 17 dependent loads of the status block -- which the
    // kernel also updates atomically, so nothing could be kept in registers -- each queued behind the wave's stores.)
    int bucket_end = (lane < kBuckets && !st->overflow) ? st->qcount[q][lane] : 0;
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) {  // kBuckets <= 32 lanes carry the table
        const int o = __shfl_up(bucket_end, d, 64);
        if (lane >= d) bucket_end += o;
    }
    const int total = __builtin_amdgcn_readlane(bucket_end, kBuckets - 1);
    // the table lives in LDS from here on (in a register it was spilled: see lane_now())
    if (lane < 32) L.bucket_end[lane] = bucket_end;
    wave_sync();
    int i_cur = blockIdx.x / kQueues;  // first round: static; afterwards the queue's shared cursor
    if (i_cur < total) {
        // expected tiles per wave, to spread this wave's background tiles over its blended ones
        const int expect = max(1, (total + stride - 1) / stride);
        const int fill_chunk = (fill.end - fill.next + expect - 1) / expect;
        int *next = const_cast<int *>(&st->next[q][0]);
        const int4 *queue_q = p.buf.queue + (size_t)q * kBuckets * p.qcap;
        auto entry_of = [&](int i) {  // i < total (wave-uniform): the bucket whose range holds position i
            const int l = lane_now();
            const int b = __popcll(__ballot(l < kBuckets && i >= L.bucket_end[l & 31]));
            const int first = b > 0 ? L.bucket_end[b - 1] : 0;
            return queue_q + (size_t)b * p.qcap + (i - __builtin_amdgcn_readfirstlane(first));
        };
        unsigned *order_l = reinterpret_cast<unsigned *>(L.keys);
        typedef __attribute__((address_space(3))) const unsigned lds_u32;
        lds_u32 *order_lds = (lds_u32 *)order_l;
        unsigned *prep_cnt = reinterpret_cast<unsigned *>(&L.keys[kSortCap / 2]);  // lists 0 and 1: idle after quadrant 1
        // LDS byte addresses (the quadrant lists hold addresses, so the blend loop does no address arithmetic)
        const unsigned stage_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)&L.geo[0];
        const unsigned lists_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)&L.keys[kSortCap / 2];
        const unsigned null_slot = stage_base + (unsigned)kNullSlot * 16u;
        auto write_null_record = [&]() {  // log2(opacity) = -inf blends nothing
            if (lane_now() == 0) {
                const float z = __uint_as_float(zero_now());
                L.geo[kNullSlot] = make_float4(z, z, z, z);
                L.col[kNullSlot] = make_float4(z, z, z, z);
                L.geo2[kNullSlot] = make_float4(z, -__builtin_inff(), z, z);
            }
        };
        write_null_record();
#if AMAV_ABLATE != 5 && AMAV_ABLATE != 7
        fill_fetch(p, fill, lane);
#endif
        int idx_v = 0;
        if (lane == 0) idx_v = stride + atomicAdd(next, 1);
        const int4 e0 = *entry_of(i_cur);
        TileRef cur = {__builtin_amdgcn_readfirstlane(e0.x), __builtin_amdgcn_readfirstlane(e0.y),
                       __builtin_amdgcn_readfirstlane(e0.z), __builtin_amdgcn_readfirstlane(e0.w)};
        bool ready = false;  // cur's blend order is in LDS and its first records are on their way
        float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0, g2 = g0;
        int i_next = __builtin_amdgcn_readfirstlane(idx_v);
        int4 e_next = make_int4(0, 0, 0, 0);
        if (i_next < total) e_next = *entry_of(i_next);
        for (;;) {
            const int ln = lane_now();  // per-tile copy of the lane id (see opaque(), lane_now())
            // the position after next: the atomic's round trip hides under this tile
            int idx2 = 0;
            if (ln == 0) idx2 = stride + atomicAdd(next, 1);
            int4 e_after = make_int4(0, 0, 0, 0);  // its queue entry, requested after the first quadrant below
            const int item = cur.item, n = cur.n;
            const unsigned my_slot = stage_base + (unsigned)ln * 16u;
            const int f = item / p.T, t = item - f * p.T;
            AMAV_STAMP(0);
            const int tx = t % p.gx, ty = t / p.gx;
            const int X0 = tx * kTile, Y0 = ty * kTile;
            const unsigned *order_g = p.buf.sorted + (size_t)f * p.cap_per_frame + cur.beg;  // long lists only
            const float4 *geom = p.buf.geom + (size_t)f * p.N * 3;
            const bool local = n <= kSortCap;
            if (p.stamps && ln == 0) p.stamps[(size_t)item * 6 + 5] = (unsigned long long)n;
            // blend order of position k: from this wave's LDS slice, or (lists longer than kSortCap) from sort_big's output.
            // Two explicit paths: a select between an LDS and a global pointer becomes a FLAT load, whose completion the
            // hardware can only express as "everything done" (vmcnt(0) + lgkmcnt(0)).
            auto load_records = [&](int k) {
                unsigned id;
                if (local)
                    id = order_lds[k];
                else
                    id = order_g[k];
                const float4 *g = geom + (size_t)id * 3;
                g0 = g[0], g1 = g[1], g2 = g[2];
            };
#ifndef AMAV_STAMP_DETAIL
            AMAV_STAMP(1);
#endif
#ifdef AMAV_STAMP_DETAIL
            const unsigned long long own_0 = __builtin_amdgcn_s_memrealtime();
#endif
            if (!ready) {  // not prepared under the previous tile: the wave's first tile, lists of 257 .. 512 keys
                const unsigned long long *keys = p.buf.keys + (size_t)f * p.cap_per_frame + cur.beg;
                if (local) {
                    unsigned *cnt = reinterpret_cast<unsigned *>(L.geo);  // the staging buffers are idle while sorting
                    bool done = false;
                    if (n <= 64) {
                        for (int k = ln; k < n; k += 64) L.keys[k] = keys[k];
                        wave_sync();
                    } else if (n <= 128)
                        done = bucket_sort<2>(keys, L.keys, cnt, order_l, n, ln);
                    else if (n <= 192)
                        done = bucket_sort<3>(keys, L.keys, cnt, order_l, n, ln);
                    else if (n <= 256)
                        done = bucket_sort<4>(keys, L.keys, cnt, order_l, n, ln);
                    else if (n <= 384)
                        done = bucket_sort<6>(keys, L.keys, cnt, order_l, n, ln);
                    else
                        done = bucket_sort<8>(keys, L.keys, cnt, order_l, n, ln);
                    if (!done) {  // short list, or depths too clustered for buckets: comparison sorts on the keys in LDS
                        if (n <= 64)
                            rank_sort<1>(L.keys, order_l, n, ln);
                        else if (n <= 128)
                            rank_sort<2>(L.keys, order_l, n, ln);
                        else if (n <= 192)
                            rank_sort<3>(L.keys, order_l, n, ln);
                        else if (n <= 256)
                            rank_sort<4>(L.keys, order_l, n, ln);
                        else
                            wave_bitonic_sort(L.keys, order_l, n, ln);
                    }
                    write_null_record();  // the sorts used the staging buffers as scratch
                }
                if (ln < n) load_records(ln);
            }
            AMAV_STAMP(2);

            // this ln's four pixels: (X0 + 8*qx + lx, Y0 + 8*qy + ly), quadrant q = qx + 2*qy
            const int lx = ln & 7, ly = ln >> 3;
            // pixel coordinates relative to the tile origin (the blend works in the tile's frame)
            float lxf0 = (float)lx, lxf1 = (float)(8 + lx), lyf0 = (float)ly, lyf1 = (float)(8 + ly);
            asm volatile("" : "+v"(lxf0), "+v"(lxf1), "+v"(lyf0), "+v"(lyf1));  // keep them in registers (no re-convert)
            const bool in0 = X0 + lx < p.W, in1 = X0 + 8 + lx < p.W, inr0 = Y0 + ly < p.H, inr1 = Y0 + 8 + ly < p.H;
            float T0 = (in0 & inr0) ? 1.f : -1.f, T1 = (in1 & inr0) ? 1.f : -1.f;
            float T2 = (in0 & inr1) ? 1.f : -1.f, T3 = (in1 & inr1) ? 1.f : -1.f;
            float R0 = 0.f, G0 = 0.f, B0 = 0.f, D0 = 0.f, R1 = 0.f, G1 = 0.f, B1 = 0.f, D1 = 0.f;
            float R2 = 0.f, G2 = 0.f, B2 = 0.f, D2 = 0.f, R3 = 0.f, G3 = 0.f, B3 = 0.f, D3 = 0.f;
            const float X0f = (float)X0, Y0f = (float)Y0;

            // the next tile (filled in at the start of the last round)
            TileRef nxt = {0, 0, 0, 0};
            bool ready_next = false;
            const float4 *geom_next = geom;

#ifdef AMAV_STAMP_DETAIL  /* diagnostic build (tools/stamp_render.py --detail): wave-time inside the blend loops / fill */
            unsigned long long t_asm = 0, t_fill = 0, t_prep = 0, t_stage = 0, t_own = __builtin_amdgcn_s_memrealtime() - own_0;
#define AMAV_TIC(x) const unsigned long long x##_0 = __builtin_amdgcn_s_memrealtime()
#define AMAV_TOC(x, acc) acc += __builtin_amdgcn_s_memrealtime() - x##_0
#else
#define AMAV_TIC(x)
#define AMAV_TOC(x, acc)
#endif
            int qalive = 15;  // quadrants that still have an unfinished pixel (wave-uniform)
            int fill_left = fill_chunk;                                       // background tiles owed by this tile ...
            const int fill_round = (fill_chunk * 64 + n - 1) / max(n, 64) + 0;  // ... per staging round (ceil(n / 64) rounds)
            // ---- blend, 64 Gaussians per staging round; one more pass ("drain") when the pixels finished early, so that
            // the next tile's preparation has exactly one place in the code
            for (int base = 0;; base += 64) {
                const bool more = qalive != 0 && base < n;
                const bool last = !more || base + 64 >= n;
                int n0 = 0, n1 = 0, n2 = 0, n3 = 0;
                // the next tile's keys / first id: live from the start of the last round to its quadrant 1 only
                unsigned long long pk[4] = {~0ull, ~0ull, ~0ull, ~0ull};
                unsigned idf = 0;
                if (last) {
                    // ---- next tile, step 1: its queue entry (fetched in the middle of the previous tile) -> its keys, into
                    // registers (in the last round the record prefetch registers are idle, so they cost no extra ones)
                    nxt.item = __builtin_amdgcn_readfirstlane(e_next.x);
                    nxt.beg = __builtin_amdgcn_readfirstlane(e_next.y);
                    nxt.n = __builtin_amdgcn_readfirstlane(e_next.z);
                    nxt.slot = __builtin_amdgcn_readfirstlane(e_next.w);
                    if (nxt.n > 0) {
                        const int fn = nxt.item / p.T;
                        geom_next = p.buf.geom + (size_t)fn * p.N * 3;
                        if (nxt.n <= kPrepCap) {
                            const unsigned long long *kn = p.buf.keys + (size_t)fn * p.cap_per_frame + nxt.beg;
#pragma unroll
                            for (int m = 0; m < 4; ++m)
                                if (ln + 64 * m < nxt.n) pk[m] = kn[ln + 64 * m];
                        } else if (nxt.n > kSortCap) {
                            if (ln < nxt.n) idf = p.buf.sorted[(size_t)fn * p.cap_per_frame + nxt.beg + ln];
                        }
                    }
                }
                if (more) {
                    AMAV_TIC(s0);
                    // quadrant mask of this ln's Gaussian: which live 8x8 quadrants its alpha >= 1/255 box can reach
                    int qm = 0;
                    if (base + ln < n) {
                        const bool hx0 = (g0.x + g2.z >= X0f) & (g0.x - g2.z <= X0f + 7.f);
                        const bool hx1 = (g0.x + g2.z >= X0f + 8.f) & (g0.x - g2.z <= X0f + 15.f);
                        const bool hy0 = (g0.y + g2.w >= Y0f) & (g0.y - g2.w <= Y0f + 7.f);
                        const bool hy1 = (g0.y + g2.w >= Y0f + 8.f) & (g0.y - g2.w <= Y0f + 15.f);
                        qm = (int)(hx0 & hy0) | ((int)(hx1 & hy0) << 1) | ((int)(hx0 & hy1) << 2) | ((int)(hx1 & hy1) << 3);
                        qm &= qalive;
                    }
                    // every ln stages its record at its own (= sorted) slot, moved into the tile's frame:
                    // u = qa (x - px) + qb (y - py) = k0 - qa lx - qb ly with k0 = qa (x - X0) + qb (y - Y0); v likewise
                    if (qm != 0) {
                        const float rx = g0.x - X0f, ry = g0.y - Y0f;
                        L.geo[ln] = make_float4(fmaf(g0.z, rx, g0.w * ry), g1.x * ry, g0.z, g0.w);
                        *reinterpret_cast<float2 *>(&L.geo2[ln]) = make_float2(g1.x, g1.y);
                        L.col[ln] = make_float4(g1.z, g1.w, g2.x, g2.y);
                    }
                    // the four quadrant lists: a ln's entry goes to the position its bit has in the quadrant's ballot
                    const unsigned long long m0 = __ballot(qm & 1), m1 = __ballot(qm & 2), m2 = __ballot(qm & 4),
                                             m3 = __ballot(qm & 8);
                    n0 = __popcll(m0), n1 = __popcll(m1), n2 = __popcll(m2), n3 = __popcll(m3);
                    auto put = [&](int qd, unsigned long long m, int cnt, int bit) {
                        typedef __attribute__((address_space(3))) unsigned lds_u32w;
                        lds_u32w *list = (lds_u32w *)(uintptr_t)(lists_base + (unsigned)qd * (kListStride * 4u));
                        if (qm & bit)
                            list[__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = my_slot;
                        if (ln < kListTail) list[cnt + ln] = null_slot;
                    };
                    if (n0) put(0, m0, n0, 1);
                    if (n1) put(1, m1, n1, 2);
                    if (n2) put(2, m2, n2, 4);
                    if (n3) put(3, m3, n3, 8);
                    wave_sync();
                    AMAV_TOC(s0, t_stage);
#if AMAV_ABLATE != 5 && AMAV_ABLATE != 7
                    // this round's share of the background (see the kernel's header), from the ids already in registers
                    { AMAV_TIC(f0);
                    for (int k = 0; k < fill_round && fill_left > 0 && fill.pos < fill.have; ++k, --fill_left)
                        fill_tile_at<kInvDepth>(p, fill, ln);
                    AMAV_TOC(f0, t_fill); }
#endif
                    // prefetch the next round's records while this one is blended
                    if (!last && base + 64 + ln < n) load_records(base + 64 + ln);
#if AMAV_ABLATE != 3 && AMAV_ABLATE != 7  /* diagnostic builds: no blending at all */
                    { AMAV_TIC(a0);
                    if (n0 && !blend_quadrant<kInvDepth>(lists_base, (n0 + 1) >> 1, lxf0, lyf0, T0, R0, G0, B0, D0)) qalive &= ~1;
                    AMAV_TOC(a0, t_asm); }
#endif
                }
                if (base == 0) {  // the tile after next: its position has come back, request its queue entry
                    i_next = __builtin_amdgcn_readfirstlane(idx2);
                    if (i_next < total) e_after = *entry_of(i_next);
                }
#if AMAV_ABLATE != 3 && AMAV_ABLATE != 7
                if (more) {
                    AMAV_TIC(a1);
                    if (n1 && !blend_quadrant<kInvDepth>(lists_base + kListStride * 4u, (n1 + 1) >> 1, lxf1, lyf0, T1, R1, G1, B1, D1)) qalive &= ~2;
                    AMAV_TOC(a1, t_asm);
                }
#endif
                if (last) {
                    // ---- next tile, steps 2 and 3: sort its keys into the id area (this tile has requested its last
                    // records, and lists 0 / 1 are done with), then request the records of its first staging round
                    AMAV_TIC(p0);
                    if (nxt.n > 0 && nxt.n <= kPrepCap) {
                        sort_prefetched(pk, nxt.n, L.keys, prep_cnt, order_l, ln);
                        if (ln < nxt.n) {
                            const float4 *g = geom_next + (size_t)order_lds[ln] * 3;
                            g0 = g[0], g1 = g[1], g2 = g[2];
                        }
                        ready_next = true;
                    } else if (nxt.n > kSortCap) {
                        if (ln < nxt.n) {
                            const float4 *g = geom_next + (size_t)idf * 3;
                            g0 = g[0], g1 = g[1], g2 = g[2];
                        }
                        ready_next = true;
                    }
                    AMAV_TOC(p0, t_prep);
                }
#if AMAV_ABLATE != 3 && AMAV_ABLATE != 7
                if (more) {
                    AMAV_TIC(a2);
                    if (n2 && !blend_quadrant<kInvDepth>(lists_base + kListStride * 8u, (n2 + 1) >> 1, lxf0, lyf1, T2, R2, G2, B2, D2)) qalive &= ~4;
                    if (n3 && !blend_quadrant<kInvDepth>(lists_base + kListStride * 12u, (n3 + 1) >> 1, lxf1, lyf1, T3, R3, G3, B3, D3)) qalive &= ~8;
                    AMAV_TOC(a2, t_asm);
                }
#endif

                wave_sync();
                if (last) break;
            }
            // the next tile's first records have had quadrants 2 and 3 to arrive.  Wait for them HERE: vector-memory operations
            // complete in order, so behind the stores below they would only count as arrived once those have drained.
            // (The statement redefines the registers, so the compiler attaches no wait of its own to the loads.)
            asm volatile("s_waitcnt vmcnt(0)"
                         : "+v"(g0.x), "+v"(g0.y), "+v"(g0.z), "+v"(g0.w), "+v"(g1.x), "+v"(g1.y), "+v"(g1.z), "+v"(g1.w),
                           "+v"(g2.x), "+v"(g2.y), "+v"(g2.z), "+v"(g2.w)
                         :
                         : "memory");

#ifdef AMAV_STAMP_DETAIL
            if (p.stamps && lane == 0) {  // slots 1 and 2 carry durations in this build: blend loops; next tile's sort + fill
                p.stamps[(size_t)item * 6 + 1] = t_asm;
                p.stamps[(size_t)item * 6 + 2] = (t_prep << 48) | ((t_fill & 0xffffull) << 32) | ((t_stage & 0xffffull) << 16) | (t_own & 0xffffull);
            }
#endif
            AMAV_STAMP(3);
            // ---- write back: quadrant q of the wave = 8 rows x 128 B
            {
                const float Tq[4] = {fabsf(T0), fabsf(T1), fabsf(T2), fabsf(T3)};
                const float Rq[4] = {R0, R1, R2, R3}, Gq[4] = {G0, G1, G2, G3}, Bq[4] = {B0, B1, B2, B3}, Dq[4] = {D0, D1, D2, D3};
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const int px = X0 + 8 * (qd & 1) + lx, py = Y0 + 8 * (qd >> 1) + ly;
                    if (px < p.W && py < p.H) {
                        float r = Rq[qd] + Tq[qd] * p.bg[0], g = Gq[qd] + Tq[qd] * p.bg[1], bl = Bq[qd] + Tq[qd] * p.bg[2];
                        if (p.clamp_output) {
                            r = fminf(fmaxf(r, 0.f), 1.f);
                            g = fminf(fmaxf(g, 0.f), 1.f);
                            bl = fminf(fmaxf(bl, 0.f), 1.f);
                        }
                        const size_t pid = ((size_t)f * p.H + py) * p.W + px;
                        store_nt(p.out_rgba, pid, make_float4(r, g, bl, 1.0f - Tq[qd]));
                        if (kInvDepth) p.out_inv_depth[pid] = Dq[qd];
                    }
                }
                // ---- the exchange's wire format, straight from the registers: the tile as 16 x 16 x 3 bytes in row
                // order (what amav_frames_pack_tiles would quantise out of the fp32 frame).  The lanes hold one pixel
                // per quadrant; the staging area (idle now) turns that into 12 contiguous bytes per lane.
                if (p.wire_header && cur.slot < p.wire_cap) {
                    unsigned char *bytes = reinterpret_cast<unsigned char *>(L.geo);
                    const unsigned bgw = p.wire_bg;
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) {
                        const int tx16 = 8 * (qd & 1) + lx, ty16 = 8 * (qd >> 1) + ly;
                        unsigned cr = bgw & 255u, cg = (bgw >> 8) & 255u, cb = (bgw >> 16) & 255u;  // outside the image
                        if (X0 + tx16 < p.W && Y0 + ty16 < p.H) {
                            const float r = Rq[qd] + Tq[qd] * p.bg[0], g = Gq[qd] + Tq[qd] * p.bg[1], bl = Bq[qd] + Tq[qd] * p.bg[2];
                            cr = (unsigned)(unsigned char)(fminf(fmaxf(r, 0.f), 1.f) * 255.0f);
                            cg = (unsigned)(unsigned char)(fminf(fmaxf(g, 0.f), 1.f) * 255.0f);
                            cb = (unsigned)(unsigned char)(fminf(fmaxf(bl, 0.f), 1.f) * 255.0f);
                        }
                        unsigned char *px3 = bytes + (ty16 * 16 + tx16) * 3;
                        px3[0] = (unsigned char)cr, px3[1] = (unsigned char)cg, px3[2] = (unsigned char)cb;
                    }
                    wave_sync();
                    const unsigned *words = reinterpret_cast<const unsigned *>(bytes) + ln * 3;
                    const unsigned w0 = words[0], w1 = words[1], w2 = words[2];
                    unsigned *dst = reinterpret_cast<unsigned *>(p.wire_payload + (size_t)cur.slot * kWireTileBytes) + ln * 3;
                    dst[0] = w0, dst[1] = w1, dst[2] = w2;
                    wave_sync();  // the staging area is the next tile's again
                }
            }
            AMAV_STAMP(4);
#if AMAV_ABLATE != 5 && AMAV_ABLATE != 7
            // what the rounds did not take (pixels finished early), and the next batch of tile ids
            if (fill_left > 0 || fill.pos >= fill.have) fill_some<kInvDepth>(p, fill, fill_left, ln);
            if (fill.pos >= fill.have && fill.next < fill.end) fill_fetch(p, fill, ln);
#endif
            if (nxt.n <= 0) break;
            cur = nxt;
            ready = ready_next;
            e_next = e_after;
        }
    }
    // the rest of this wave's background tiles (all of them when it had no tile to blend; every tile of the launch
    // when the instance regions overflowed: the caller must retry)
#if AMAV_ABLATE != 5 && AMAV_ABLATE != 7
    fill_some<kInvDepth>(p, fill, 0x7fffffff, lane_now());
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
    const size_t bin_lds = ((size_t)2 * T + 48 + 3 * kQueues * (kBuckets + 1)) * sizeof(int);
    AMAV_REQUIRE(bin_lds <= 160 * 1024, "amav_rasterize_forward: %d tiles need %zu B of LDS in the binning block (max 160 KiB)",
                 T, bin_lds);
    // fused binning: room in LDS for the frame's binning records (8 B each), tile bounds in 8 bits, no radii output
    // (the scatter pass writes those from the full records)
    const bool stash = bin_lds + 8 + (size_t)N * 8 <= 160 * 1024 && gx <= 255 && gy <= 255 && a->out_radii == nullptr;
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
    p.wire_header = nullptr, p.wire_frame_counts = nullptr, p.wire_offsets = nullptr, p.wire_payload = nullptr;
    p.wire_cap = 0, p.wire_bg = 0;
    p.slices = bin_slices(F);
    p.stash = stash ? 1 : 0;
    if (a->wire) {
        AMAV_REQUIRE(a->clamp_output, "amav_rasterize_forward: the wire output carries the clamped colours (set clamp_output)");
        AMAV_REQUIRE((reinterpret_cast<uintptr_t>(a->wire) & 15) == 0 && a->wire_capacity_tiles >= 0 &&
                         a->wire_capacity_tiles <= (long long)F * T,
                     "amav_rasterize_forward: bad wire buffer (alignment, or capacity %lld outside [0, %lld])",
                     (long long)a->wire_capacity_tiles, (long long)F * T);
        const size_t wire_need = (wire_payload_at(F, F * T) + (size_t)a->wire_capacity_tiles * kWireTileBytes + 15) / 16 * 16;
        if (a->wire_bytes < wire_need)
            return fail(AMAV_ERR_WORKSPACE, "amav_rasterize_forward: wire buffer %zu < required %zu", a->wire_bytes, wire_need);
        p.wire_header = static_cast<int *>(a->wire);
        p.wire_frame_counts = p.wire_header + kWireHeaderInts;
        p.wire_offsets = p.wire_frame_counts + F;
        p.wire_payload = static_cast<unsigned char *>(a->wire) + wire_payload_at(F, F * T);
        p.wire_cap = (int)a->wire_capacity_tiles;
        auto q8 = [](float v) { return (unsigned)(unsigned char)(std::fmin(std::fmax(v, 0.f), 1.f) * 255.0f); };
        p.wire_bg = q8(a->bg[0]) | (q8(a->bg[1]) << 8) | (q8(a->bg[2]) << 16);
    }

    hipStream_t stream = static_cast<hipStream_t>(stream_);
    static const hipError_t attr[5] = {
        hipFuncSetAttribute(reinterpret_cast<const void *>(&slice_count_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
        hipFuncSetAttribute(reinterpret_cast<const void *>(&slice_scatter_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
        hipFuncSetAttribute(reinterpret_cast<const void *>(&bin_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
        hipFuncSetAttribute(reinterpret_cast<const void *>(&bin_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
        hipFuncSetAttribute(reinterpret_cast<const void *>(&bin_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)};
    if (attr[0] != hipSuccess || attr[1] != hipSuccess || attr[2] != hipSuccess || attr[3] != hipSuccess || attr[4] != hipSuccess)
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
    if (p.wire_header && zero_async(p.wire_header, (size_t)kWireHeaderInts * 4, stream) != hipSuccess)
        return fail(AMAV_ERR_LAUNCH, "amav_rasterize_forward: wire header clear failed");

    // Few frames (the reference's own window is 6): the projection of ALL Gaussians runs as its own launch over the
    // whole chip and the per-frame blocks only count and scatter; a shard of many frames keeps the projection inside
    // the per-frame block (one frame per CU anyway; measured 250 x 10 000: 0.12 ms fused, 0.20 ms split -- 6 x 30 000:
    // 81 us split, of which 15 us projection).
    if (F < kFusedMinFrames && F <= 65535) {
        const dim3 pre_grid((unsigned)((N + 255) / 256), (unsigned)F);
        if (packed)
            preprocess_kernel<true><<<pre_grid, 256, 0, stream>>>(p);
        else
            preprocess_kernel<false><<<pre_grid, 256, 0, stream>>>(p);
        const dim3 slice_grid((unsigned)p.slices, (unsigned)F);
        slice_count_kernel<<<slice_grid, 1024, (size_t)T * sizeof(int), stream>>>(p);
        bin_kernel<false, false><<<F, 1024, bin_lds, stream>>>(p);
        slice_scatter_kernel<<<slice_grid, 1024, (size_t)T * sizeof(int), stream>>>(p);
    } else if (packed)
        bin_kernel<true, true><<<F, 1024, bin_lds + (stash ? 8 + (size_t)N * 8 : 0), stream>>>(p);
    else
        bin_kernel<true, false><<<F, 1024, bin_lds + (stash ? 8 + (size_t)N * 8 : 0), stream>>>(p);
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
