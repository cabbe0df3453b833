// Point refiner (second half): the sparse / serialised pieces of the reference's
// PointTransformerV3 (src/models/point_transformer/pointtransformer_v3.py, point_encoder.py:25-40, called from
// src/models/renderer.py:143-151) for a batch of CLOUDS (one per frame), deterministic.
//
// The reference leans on spconv (hash-table submanifold convolution), torch_scatter (segment_csr) and, optionally,
// flash_attn; none has a ROCm build.  What replaces them here:
//
//   amav_cloud_voxelize    grid = floor(res * p) - per-cloud minimum, serialisation depth per cloud      (point_encoder.py:33,
//                          pointtransformer_v3.py:98-101; origin / depth per cloud: DESIGN.md section 4.5)
//   amav_cloud_codes       the four serialisation keys (z, z-trans, hilbert, hilbert-trans) of every point, as
//                          (cloud << 48 | code) so that ONE stable sort orders all clouds    (serialization/*.py)
//   amav_cloud_neighbors   [n, k^3] table of the rows a submanifold convolution gathers: binary search of the
//                          neighbour voxel's z-order key in the cloud's sorted key run (no hash table: the sorted run
//                          is already there, and "first row of the run of equal keys" is a reproducible choice for
//                          voxels that hold several points)                                     (spconv SubMConv3d)
//   amav_subm_pair_gemm    the convolution as a gather-GEMM over the (input row, output row) pairs that exist, grouped
//                          by tap, fp32 MFMA; amav_subm_pair_sum adds each row's products in tap order (no atomics)
//   amav_patch_attention   softmax(Q K^T / sqrt(d)) V inside patches of <= 512 consecutive points of a serialised
//                          order, rows gathered through the order; fp32 MFMA, online softmax   (SerializedAttention)
//   amav_cluster_max       per-cluster channel maximum + BatchNorm(eval) + GELU                 (SerializedPooling)
//   amav_bn_gelu           BatchNorm(eval) + GELU                                                (Embedding, Unpooling)
//   amav_unpool_merge      skip = GELU(BN(x)); sum = skip + up[cluster]                        (SerializedUnpooling)
//   amav_rows_norm         s = base + LN_a(x) (or base + x); n = LN_b(s): the residual + LayerNorm passes of a Block
#include <algorithm>
#include <climits>
#include <cmath>

#include "amav_common.h"

namespace amav {
namespace cloud {

typedef float f32x16 __attribute__((ext_vector_type(16)));


__device__ __forceinline__ float gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// ---- voxelize -----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bounds_init_kernel(int clouds, int *__restrict__ bounds) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < clouds * 6) bounds[i] = (i % 6) < 3 ? INT_MAX : INT_MIN;
}

__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

__global__ __launch_bounds__(256) void bounds_kernel(long long n, const float *__restrict__ points,
                                                     const int *__restrict__ cloud_of, float res,
                                                     int *__restrict__ bounds) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n;
    const long long j = live ? i : n - 1;
    const int c = cloud_of[j];
    const int gx = (int)floorf(points[3 * j] * res), gy = (int)floorf(points[3 * j + 1] * res),
              gz = (int)floorf(points[3 * j + 2] * res);
    const int c0 = __builtin_amdgcn_readfirstlane(c);
    if (__all(c == c0)) {  // the usual case: a wave lies inside one cloud -> six atomics per wave
        const int a = wave_min(gx), b = wave_min(gy), d = wave_min(gz), e = wave_max(gx), f = wave_max(gy), g = wave_max(gz);
        if ((threadIdx.x & 63) == 0) {
            int *bd = bounds + 6 * c0;
            atomicMin(bd, a), atomicMin(bd + 1, b), atomicMin(bd + 2, d);
            atomicMax(bd + 3, e), atomicMax(bd + 4, f), atomicMax(bd + 5, g);
        }
    } else {
        int *bd = bounds + 6 * c;
        atomicMin(bd, gx), atomicMin(bd + 1, gy), atomicMin(bd + 2, gz);
        atomicMax(bd + 3, gx), atomicMax(bd + 4, gy), atomicMax(bd + 5, gz);
    }
}

__global__ __launch_bounds__(256) void grid_kernel(long long n, int clouds, const float *__restrict__ points,
                                                   const int *__restrict__ cloud_of, float res,
                                                   const int *__restrict__ bounds, int *__restrict__ grid,
                                                   int *__restrict__ depth) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < clouds) {
        const int *bd = bounds + 6 * i;
        const int ext = max(max(bd[3] - bd[0], bd[4] - bd[1]), bd[5] - bd[2]);
        depth[i] = ext > 0 ? 32 - __clz(ext) : 0;  // int.bit_length() of the largest grid coordinate
    }
    if (i >= n) return;
    const int *bd = bounds + 6 * cloud_of[i];
    grid[3 * i] = (int)floorf(points[3 * i] * res) - bd[0];
    grid[3 * i + 1] = (int)floorf(points[3 * i + 1] * res) - bd[1];
    grid[3 * i + 2] = (int)floorf(points[3 * i + 2] * res) - bd[2];
}

// ---- serialisation keys ---------------------------------------------------------------------------------------------
// bit i of v -> bit 3 i (16 bits in, 46 bits out)
__device__ __forceinline__ unsigned long long spread3(unsigned int v) {
    unsigned long long x = v & 0xffffu;
    x = (x | (x << 16)) & 0x0000ff0000ffULL;
    x = (x | (x << 8)) & 0x00f00f00f00fULL;
    x = (x | (x << 4)) & 0x0c30c30c30c3ULL;
    x = (x | (x << 2)) & 0x249249249249ULL;
    return x;
}

// z_order.py:42-52: x -> bit 3i+2, y -> 3i+1, z -> 3i
__device__ __forceinline__ unsigned long long z_code(unsigned int x, unsigned int y, unsigned int z) {
    return (spread3(x) << 2) | (spread3(y) << 1) | spread3(z);
}

// hilbert.py:93-190 = Skilling's AxesToTranspose, most significant bit first, then the Gray decode of the
// interleaved bits (axis 0 most significant inside a triple)
__device__ __forceinline__ unsigned long long hilbert_code(unsigned int x0, unsigned int x1, unsigned int x2, int depth) {
    for (int bit = depth - 1; bit > 0; --bit) {  // Q = 1 << bit; the pass with Q = 1 has no lower bits to touch
        const unsigned int Q = 1u << bit, P = Q - 1;
        unsigned int t;
        if (x0 & Q) x0 ^= P;
        if (x1 & Q) x0 ^= P; else { t = (x0 ^ x1) & P; x0 ^= t; x1 ^= t; }
        if (x2 & Q) x0 ^= P; else { t = (x0 ^ x2) & P; x0 ^= t; x2 ^= t; }
    }
    unsigned long long g = z_code(x0, x1, x2);
    g ^= g >> 1, g ^= g >> 2, g ^= g >> 4, g ^= g >> 8, g ^= g >> 16, g ^= g >> 32;
    return g;
}

__global__ __launch_bounds__(256) void codes_kernel(long long n, const int *__restrict__ grid,
                                                    const int *__restrict__ cloud_of, const int *__restrict__ cloud_depth,
                                                    long long *__restrict__ keys) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = cloud_of[i], depth = cloud_depth[c];
    const unsigned int mask = depth >= 16 ? 0xffffu : ((1u << depth) - 1u);
    const unsigned int x = grid[3 * i] & mask, y = grid[3 * i + 1] & mask, z = grid[3 * i + 2] & mask;
    const long long hi = (long long)c << 48;
    keys[i] = hi | (long long)z_code(x, y, z);
    keys[n + i] = hi | (long long)z_code(y, x, z);
    keys[2 * n + i] = hi | (long long)hilbert_code(x, y, z, depth);
    keys[3 * n + i] = hi | (long long)hilbert_code(y, x, z, depth);
}

// ---- neighbour table --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void neighbors_kernel(long long n, int ksize, const int *__restrict__ grid,
                                                        const int *__restrict__ cloud_of,
                                                        const int *__restrict__ cloud_depth,
                                                        const int *__restrict__ cloud_start,
                                                        const long long *__restrict__ sorted_keys,
                                                        const long long *__restrict__ order, int *__restrict__ nbr) {
    const int taps = ksize * ksize * ksize;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n * taps) return;
    const long long i = gid / taps;
    const int t = (int)(gid - i * taps), r = ksize / 2;
    const int da = t / (ksize * ksize) - r, db = (t / ksize) % ksize - r, dc = t % ksize - r;
    if ((da | db | dc) == 0) {  // the centre tap is the point itself, also where a voxel holds several points
        nbr[gid] = (int)i;
        return;
    }
    const int c = cloud_of[i], depth = cloud_depth[c];
    const int x = grid[3 * i] + da, y = grid[3 * i + 1] + db, z = grid[3 * i + 2] + dc;
    const int lim = depth >= 16 ? 65536 : (1 << depth);
    int found = -1;
    if (x >= 0 && y >= 0 && z >= 0 && x < lim && y < lim && z < lim) {
        const long long key = ((long long)c << 48) | (long long)z_code(x, y, z);
        int lo = cloud_start[c], hi = cloud_start[c + 1];
        const int end = hi;
        while (lo < hi) {  // lower bound: first sorted position with key >= target
            const int mid = (lo + hi) >> 1;
            if (sorted_keys[mid] < key) lo = mid + 1; else hi = mid;
        }
        if (lo < end && sorted_keys[lo] == key) found = (int)order[lo];  // stable sort: the voxel's lowest row
    }
    nbr[gid] = found;
}

// ---- submanifold convolution as a gather-GEMM over (input row -> output row) pairs ------------------------------------------
// Pairs are grouped by tap (tap_start [taps + 1]); pair p of tap t multiplies feat[pair_src[p]] [C_in] by
// Wt[t] [C_in][C_out] into products[p] [C_out].  Only voxels that exist are multiplied (a body surface fills 7 of the
// 27 taps of a 3x3x3 kernel and 22 of the 125 of the stem), and the per-row sum over taps is a second, ordered pass
// (pair_sum_kernel): no atomics, fixed evaluation order.
// One workgroup = 128 pairs of one tap x NT output channels; K swept in chunks of 32 through LDS (A rows gathered,
// next chunk's global loads in flight under the MFMAs); wave w owns pairs 32 w .. 32 w + 31, NT / 32 accumulators of
// v_mfma_f32_32x32x2_f32 (exact fp32 products).
template <int NT>
__global__ __launch_bounds__(256) void pair_gemm_kernel(const float *__restrict__ feat, const int *__restrict__ pair_src,
                                                        const int *__restrict__ tap_start,
                                                        const int *__restrict__ tile_start, int taps,
                                                        const float *__restrict__ Wt, float *__restrict__ products,
                                                        int Cin, int Cout) {
    constexpr int NACC = NT / 32, KC = 32, LDA = KC + 1, LDB = NT + 4;
    constexpr int BQ = KC * NT / 4 / 256;  // float4s of the B chunk per thread (1, 2 or 4)
    __shared__ float As[128 * LDA];
    __shared__ float Bs[KC * LDB];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, hh = lane >> 5;
    // which tap this tile belongs to: last t with tile_start[t] <= blockIdx.x (uniform -> scalar loop)
    int lo = 0, hi = taps;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (tile_start[mid] <= (int)blockIdx.x) lo = mid; else hi = mid;
    }
    const int tap = lo;
    const int p0 = tap_start[tap] + ((int)blockIdx.x - tile_start[tap]) * 128, p_end = tap_start[tap + 1];
    const int n0 = blockIdx.y * NT;
    const float *W = Wt + (size_t)tap * Cin * Cout + n0;

    // A staging: thread -> rows (tid / 8) + 32 i, 4 consecutive k at (tid % 8) * 4
    const int arow = tid >> 3, ak = (tid & 7) * 4;
    const float *ap[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) ap[i] = feat + (size_t)pair_src[min(p0 + arow + 32 * i, p_end - 1)] * Cin + ak;
    // B staging: thread -> float4 number tid + 256 i of the [KC][NT] chunk
    float4 a0, a1, a2, a3, b[BQ];
#define AMAV_PG_LOAD(k0_)                                                                            \
    {                                                                                                \
        a0 = *reinterpret_cast<const float4 *>(ap[0] + (k0_));                                       \
        a1 = *reinterpret_cast<const float4 *>(ap[1] + (k0_));                                       \
        a2 = *reinterpret_cast<const float4 *>(ap[2] + (k0_));                                       \
        a3 = *reinterpret_cast<const float4 *>(ap[3] + (k0_));                                       \
        _Pragma("unroll") for (int i = 0; i < BQ; ++i) {                                             \
            const int t_ = tid + 256 * i, kr_ = t_ / (NT / 4), nc_ = (t_ % (NT / 4)) * 4;            \
            b[i] = *reinterpret_cast<const float4 *>(W + (size_t)((k0_) + kr_) * Cout + nc_);        \
        }                                                                                            \
    }
#define AMAV_PG_STAGE_A(i_, v_)                          \
    {                                                    \
        float *d_ = &As[(arow + 32 * (i_)) * LDA + ak];  \
        d_[0] = v_.x, d_[1] = v_.y, d_[2] = v_.z, d_[3] = v_.w; \
    }
    f32x16 acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[a][t] = 0.f;

    AMAV_PG_LOAD(0)
    for (int k0 = 0; k0 < Cin; k0 += KC) {
        AMAV_PG_STAGE_A(0, a0) AMAV_PG_STAGE_A(1, a1) AMAV_PG_STAGE_A(2, a2) AMAV_PG_STAGE_A(3, a3)
#pragma unroll
        for (int i = 0; i < BQ; ++i) {
            const int t_ = tid + 256 * i, kr_ = t_ / (NT / 4), nc_ = (t_ % (NT / 4)) * 4;
            *reinterpret_cast<float4 *>(&Bs[kr_ * LDB + nc_]) = b[i];
        }
        __syncthreads();
        if (k0 + KC < Cin) AMAV_PG_LOAD(k0 + KC)
        const float *arow_l = &As[(wave * 32 + c) * LDA + hh];
        const float *brow_l = &Bs[hh * LDB + c];
        // operands of k-step s + 1 are read from LDS before the MFMAs of step s issue (read just in time, every MFMA
        // group waited a full LDS round trip: ds_read; s_waitcnt 0; mfma ...); the order is pinned, as in attention.hip
        float av[2], bv[2][NACC];
        av[0] = arow_l[0];
#pragma unroll
        for (int a = 0; a < NACC; ++a) bv[0][a] = brow_l[32 * a];
#pragma unroll
        for (int s = 0; s < KC / 2; ++s) {
            if (s + 1 < KC / 2) {
                av[(s + 1) & 1] = arow_l[2 * (s + 1)];
#pragma unroll
                for (int a = 0; a < NACC; ++a) bv[(s + 1) & 1][a] = brow_l[2 * (s + 1) * LDB + 32 * a];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < NACC; ++a)
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s & 1], bv[s & 1][a], acc[a], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
#undef AMAV_PG_LOAD
#undef AMAV_PG_STAGE_A
    // accumulator register t of lane (c, hh): pair 32 wave + (t & 3) + 8 (t >> 2) + 4 hh, channel n0 + 32 a + c
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int p = p0 + wave * 32 + (t & 3) + 8 * (t >> 2) + 4 * hh;
        if (p < p_end) {
#pragma unroll
            for (int a = 0; a < NACC; ++a) products[(size_t)p * Cout + n0 + 32 * a + c] = acc[a][t];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same gather-GEMM on the 16-bit matrix pipe (split-product arithmetic): features and weights
// as two fp16 parts each, (x 2^e = h1 + h2), three partial products per fp32 product on v_mfma_f32_32x32x16_f16 with fp32
// accumulation -- the fp32 product to 2^-22 at a third of the matrix time.  Scaling (exact powers of two): the weights
// by one exponent from their largest magnitude (static: subm_weights_split_kernel writes it into the prepared buffer's
// header), the features by one exponent from the largest |feature| of the call (feat_absmax_kernel -> `scratch`).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
constexpr size_t kSplitHeader = 256;  // prepared weights: [0] float largest |w|, [1] int scale exponent

__device__ __forceinline__ int f16_scale_exp(float amax) {
    if (!(amax > 0.f)) return 0;
    return max(-100, min(100, 14 - ilogbf(amax)));
}
__device__ __forceinline__ void split2(float x, _Float16 &a, _Float16 &b) {
    a = (_Float16)x;
    b = (_Float16)__builtin_fmaf((float)a, -1.0f, x);
}

__global__ __launch_bounds__(256) void absmax_kernel(const float4 *__restrict__ x4, long long n4, unsigned *__restrict__ out) {
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 v = x4[i];
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(out, __float_as_uint(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]))));
}

// weights [taps][cin][cout] fp32 -> [tap][cout / 32][cin / 16][part][lane half hh][column c][8 k] fp16: the B fragment of
// lane (c, hh) for one (k-step, part) is 16 contiguous bytes, and a 32-row chunk of one 32-column block is 4 KB that
// the kernel copies to LDS as is.  One thread per (tap, column block, k-step, hh, c).
__global__ __launch_bounds__(256) void subm_weights_split_kernel(const float *__restrict__ w, int cin, int cout, long long items,
                                                                 unsigned *__restrict__ hdr, _Float16 *__restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int e = f16_scale_exp(__uint_as_float(hdr[0]));
    if (i == 0) reinterpret_cast<int *>(hdr)[1] = e;
    if (i >= items) return;
    const int c = (int)(i & 31), hh = (int)((i >> 5) & 1);
    const long long rest = i >> 6;  // (tap * nblk + nb) * ksteps + ks
    const int ksteps = cin / 16, nblk = cout / 32;
    const int ks = (int)(rest % ksteps), nb = (int)((rest / ksteps) % nblk);
    const long long tap = rest / ksteps / nblk;
    const float scale = ldexpf(1.0f, e);
    f16x8 p1, p2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 16 * ks + 8 * hh + j;
        _Float16 a, b;
        split2(w[(tap * cin + k) * cout + nb * 32 + c] * scale, a, b);
        p1[j] = a, p2[j] = b;
    }
    _Float16 *dst = out + (rest * 2) * 512 + hh * 256 + c * 8;
    *reinterpret_cast<f16x8 *>(dst) = p1;
    *reinterpret_cast<f16x8 *>(dst + 512) = p2;
}

// grid (tiles of 128 pairs, cout / NT), 256 threads = 4 waves of 32 pairs x NT columns, as pair_gemm_kernel
template <int NT>
__global__ __launch_bounds__(256) void pair_gemm_f16_kernel(const float *__restrict__ feat, const int *__restrict__ pair_src,
                                                            const int *__restrict__ tap_start,
                                                            const int *__restrict__ tile_start, int taps,
                                                            const void *__restrict__ wsplit,
                                                            const unsigned *__restrict__ feat_amax,
                                                            float *__restrict__ products, int Cin, int Cout) {
    constexpr int NACC = NT / 32, KC = 32, LDA = KC + 8;  // A rows of 80 bytes: 16-byte fragment reads without conflicts
    __shared__ _Float16 As[2][128 * LDA];                 // [part][pair][k]
    __shared__ float4 Bs[NACC * 256];                     // [column block][k-step][part][hh][c][8 k], 4 KB per block
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, hh = lane >> 5;
    int lo = 0, hi = taps;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (tile_start[mid] <= (int)blockIdx.x) lo = mid; else hi = mid;
    }
    const int tap = lo;
    const int p0 = tap_start[tap] + ((int)blockIdx.x - tile_start[tap]) * 128, p_end = tap_start[tap + 1];
    const int n0 = blockIdx.y * NT;
    const int ea = f16_scale_exp(__uint_as_float(feat_amax[0]));
    const int ew = reinterpret_cast<const int *>(wsplit)[1];
    const float a_scale = ldexpf(1.0f, ea);
    const int ksteps = Cin / 16, nblk = Cout / 32;
    // this block's weights: column block n0 / 32 + a, all k-steps: 2 KB per k-step
    const float4 *wb = reinterpret_cast<const float4 *>(static_cast<const char *>(wsplit) + kSplitHeader) +
                       ((size_t)tap * nblk + n0 / 32) * ksteps * 128;

    const int arow = tid >> 3, ak = (tid & 7) * 4;
    const float *ap0 = feat + (size_t)pair_src[min(p0 + arow, p_end - 1)] * Cin + ak;
    const float *ap1 = feat + (size_t)pair_src[min(p0 + arow + 32, p_end - 1)] * Cin + ak;
    const float *ap2 = feat + (size_t)pair_src[min(p0 + arow + 64, p_end - 1)] * Cin + ak;
    const float *ap3 = feat + (size_t)pair_src[min(p0 + arow + 96, p_end - 1)] * Cin + ak;
    float4 a0, a1, a2, a3, b0, b1, b2, b3;  // named: no arrays behind the prefetch (see lbs.hip, skin_f16_kernel)
#define AMAV_PGH_LOAD(k0_)                                                                   \
    {                                                                                        \
        a0 = *reinterpret_cast<const float4 *>(ap0 + (k0_));                                 \
        a1 = *reinterpret_cast<const float4 *>(ap1 + (k0_));                                 \
        a2 = *reinterpret_cast<const float4 *>(ap2 + (k0_));                                 \
        a3 = *reinterpret_cast<const float4 *>(ap3 + (k0_));                                 \
        const float4 *src_ = wb + (size_t)((k0_) / 16) * 128 + tid;                          \
        b0 = src_[0];                                                                        \
        if (NACC > 1) b1 = src_[(size_t)ksteps * 128];                                       \
        if (NACC > 2) b2 = src_[(size_t)2 * ksteps * 128], b3 = src_[(size_t)3 * ksteps * 128]; \
    }
#define AMAV_PGH_STAGE_A(i_, v_)                                                     \
    {                                                                                \
        const float x_[4] = {v_.x, v_.y, v_.z, v_.w};                                \
        f16x4 h1_, h2_;                                                              \
        _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) {                           \
            _Float16 u_, l_;                                                         \
            split2(x_[e_] * a_scale, u_, l_);                                        \
            h1_[e_] = u_, h2_[e_] = l_;                                              \
        }                                                                            \
        *reinterpret_cast<f16x4 *>(&As[0][(arow + 32 * (i_)) * LDA + ak]) = h1_;     \
        *reinterpret_cast<f16x4 *>(&As[1][(arow + 32 * (i_)) * LDA + ak]) = h2_;     \
    }
    f32x16 acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[a][t] = 0.f;

    AMAV_PGH_LOAD(0)
    for (int k0 = 0; k0 < Cin; k0 += KC) {
        AMAV_PGH_STAGE_A(0, a0) AMAV_PGH_STAGE_A(1, a1) AMAV_PGH_STAGE_A(2, a2) AMAV_PGH_STAGE_A(3, a3)
        Bs[tid] = b0;
        if (NACC > 1) Bs[256 + tid] = b1;
        if (NACC > 2) Bs[512 + tid] = b2, Bs[768 + tid] = b3;
        __syncthreads();
        if (k0 + KC < Cin) AMAV_PGH_LOAD(k0 + KC)
        const _Float16 *al = &As[0][(wave * 32 + c) * LDA + 8 * hh];
        const _Float16 *bl = reinterpret_cast<const _Float16 *>(Bs) + hh * 256 + c * 8;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const f16x8 h1 = *reinterpret_cast<const f16x8 *>(al + 16 * s);
            const f16x8 h2 = *reinterpret_cast<const f16x8 *>(al + 128 * LDA + 16 * s);
#pragma unroll
            for (int a = 0; a < NACC; ++a) {
                // column block a, k-step s, part p at (a * 4 + s * 2 + p) * 512 halfs
                const f16x8 g1 = *reinterpret_cast<const f16x8 *>(bl + (a * 4 + s * 2 + 0) * 512);
                const f16x8 g2 = *reinterpret_cast<const f16x8 *>(bl + (a * 4 + s * 2 + 1) * 512);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h2, g1, acc[a], 0, 0, 0);  // small terms first
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h1, g2, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h1, g1, acc[a], 0, 0, 0);
            }
        }
        __syncthreads();
    }
#undef AMAV_PGH_LOAD
#undef AMAV_PGH_STAGE_A
    const float unscale = ldexpf(1.0f, -(ea + ew));
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int p = p0 + wave * 32 + (t & 3) + 8 * (t >> 2) + 4 * hh;
        if (p < p_end) {
#pragma unroll
            for (int a = 0; a < NACC; ++a) products[(size_t)p * Cout + n0 + 32 * a + c] = acc[a][t] * unscale;
        }
    }
}

// out[i] = bias + sum over taps (ascending) of products[pair_of[i][tap]]; one thread per (row, 4 channels)
__global__ __launch_bounds__(256) void pair_sum_kernel(long long n, int taps, int cout4, const float4 *__restrict__ products,
                                                       const int *__restrict__ pair_of, const float4 *__restrict__ bias,
                                                       float4 *__restrict__ out) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n * cout4) return;
    const long long i = gid / cout4;
    const int c = (int)(gid - i * cout4);
    float4 acc = bias ? bias[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    const int *po = pair_of + i * taps;
#pragma unroll 4
    for (int t = 0; t < taps; ++t) {
        const int p = po[t];
        if (p >= 0) {
            const float4 y = products[(long long)p * cout4 + c];
            acc.x += y.x, acc.y += y.y, acc.z += y.z, acc.w += y.w;
        }
    }
    out[gid] = acc;
}

// ---- patch attention --------------------------------------------------------------------------------------------------
// desc[patch] = {first sorted position, K (keys = queries of the patch), own (leading positions that are stored), 0}.
// Position j of the patch is sorted position first + j, or first + j - K for j >= own: the tail of a cloud's last,
// incomplete patch is filled with the same slots of the patch before it (pointtransformer_v3.py:419-432); those
// borrowed slots take part as keys and their query results are dropped (the `unpad` gather at :462,493).
// One workgroup = 128 queries of one (patch, head); S^T = K Q^T and O^T = V^T P^T on v_mfma_f32_32x32x2_f32 with the
// probabilities kept in registers (layout notes: attention.hip).
template <int D>
__global__ __launch_bounds__(256, 3) void patch_attention_kernel(const float *__restrict__ qkv,
                                                              const long long *__restrict__ order,
                                                              const int4 *__restrict__ desc, float *__restrict__ out,
                                                              int C, float scale_log2e) {
    constexpr int DV = D < 32 ? 32 : D;  // width of the V tile (zero padded: the MFMA produces 32 rows of O^T)
    constexpr int NO = DV / 32;
    constexpr int kLdk = 65;
    __shared__ float Kt[D * kLdk];  // [d][key]
    __shared__ float Vs[64 * DV];   // [key][d]
    __shared__ long long rows[64];  // qkv row of every key of the tile
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c = lane & 31, hh = lane >> 5;
    const int head = blockIdx.y;
    const int4 pd = desc[blockIdx.z];
    const int first = pd.x, K = pd.y, own = pd.z;
    if ((int)blockIdx.x * 128 >= K) return;  // uniform over the workgroup
    const int q0 = blockIdx.x * 128 + wave * 32;
    auto row_of = [&](int j) -> long long {
        j = min(j, K - 1);
        return order[first + j - (j >= own ? K : 0)];
    };
    const long long stride = 3LL * C;

    float Qr[D / 2];
    const long long qrow = row_of(q0 + c);
    {
        const float *qp = qkv + qrow * stride + head * D;
#pragma unroll
        for (int s = 0; s < D / 2; ++s) {
            const float2 t = *reinterpret_cast<const float2 *>(qp + 2 * s);
            Qr[s] = (hh ? t.y : t.x) * scale_log2e;
        }
    }
    if (D < 32) {  // columns D..31 of V stay zero for the whole sweep
        for (int t = tid; t < 64 * (DV - D); t += 256) Vs[(t / (DV - D)) * DV + D + t % (DV - D)] = 0.f;
    }

    f32x16 O[NO];
#pragma unroll
    for (int a = 0; a < NO; ++a)
#pragma unroll
        for (int t = 0; t < 16; ++t) O[a][t] = 0.f;
    float m_run = -1e30f, l_run = 0.f;

    const int ntiles = (K + 63) / 64;
    for (int kt = 0; kt < ntiles; ++kt) {
        if (tid < 64) rows[tid] = row_of(kt * 64 + tid);
        __syncthreads();
        constexpr int kQuads = 64 * D / 4;  // float4s per matrix tile
#pragma unroll
        for (int i = 0; i < (kQuads + 255) / 256; ++i) {
            const int t = tid + 256 * i;
            if (t < kQuads) {
                const int key = t / (D / 4), d4 = (t % (D / 4)) * 4;
                const float *kp = qkv + rows[key] * stride + C + head * D + d4;
                const float4 kv = *reinterpret_cast<const float4 *>(kp);
                const float4 vv = *reinterpret_cast<const float4 *>(kp + C);
                Kt[(d4 + 0) * kLdk + key] = kv.x;
                Kt[(d4 + 1) * kLdk + key] = kv.y;
                Kt[(d4 + 2) * kLdk + key] = kv.z;
                Kt[(d4 + 3) * kLdk + key] = kv.w;
                *reinterpret_cast<float4 *>(&Vs[key * DV + d4]) = vv;
            }
        }
        __syncthreads();

        f32x16 S0, S1;
#pragma unroll
        for (int t = 0; t < 16; ++t) S0[t] = 0.f, S1[t] = 0.f;
        // K operands of step s + 2 are read before the MFMAs of step s issue; order pinned (see attention.hip)
        float ka[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            ka[u][0] = Kt[(2 * u + hh) * kLdk + c];
            ka[u][1] = Kt[(2 * u + hh) * kLdk + 32 + c];
        }
#pragma unroll
        for (int s = 0; s < D / 2; ++s) {
            const float a0 = ka[s & 1][0], a1 = ka[s & 1][1];
            if (s + 2 < D / 2) {
                ka[s & 1][0] = Kt[(2 * (s + 2) + hh) * kLdk + c];
                ka[s & 1][1] = Kt[(2 * (s + 2) + hh) * kLdk + 32 + c];
            }
            __builtin_amdgcn_sched_barrier(0);
            S0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, Qr[s], S0, 0, 0, 0);
            S1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, Qr[s], S1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if ((kt + 1) * 64 > K) {  // accumulator register t of key half kb: key kt*64 + 32 kb + (t&3) + 8 (t>>2) + 4 hh
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int kk = kt * 64 + (t & 3) + 8 * (t >> 2) + 4 * hh;
                if (kk >= K) S0[t] = -1e30f;
                if (kk + 32 >= K) S1[t] = -1e30f;
            }
        }
        float mx = S0[0];
#pragma unroll
        for (int t = 1; t < 16; ++t) mx = fmaxf(mx, S0[t]);
#pragma unroll
        for (int t = 0; t < 16; ++t) mx = fmaxf(mx, S1[t]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float corr = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
#pragma unroll
        for (int a = 0; a < NO; ++a)
#pragma unroll
            for (int t = 0; t < 16; ++t) O[a][t] *= corr;

        // V operands of step u + 2 are read and the probability of step u + 1 is exponentiated before the MFMAs of
        // step u issue: the LDS round trip and the quarter-rate exp run under the matrix pipe
        auto vrow = [&](int u) { return ((u >> 4) * 32 + (u & 3) + 8 * ((u & 15) >> 2) + 4 * hh) * DV + c; };
        auto score = [&](int u) { return (u < 16 ? S0[u] : S1[u - 16]) - m_new; };
        float va[3][NO];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int a = 0; a < NO; ++a) va[u][a] = Vs[vrow(u) + 32 * a];
        float p_cur = __builtin_amdgcn_exp2f(score(0)), psum = 0.f;
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const float pu = p_cur;
            float vu[NO];
#pragma unroll
            for (int a = 0; a < NO; ++a) vu[a] = va[u % 3][a];
            if (u + 2 < 32) {
#pragma unroll
                for (int a = 0; a < NO; ++a) va[(u + 2) % 3][a] = Vs[vrow(u + 2) + 32 * a];
            }
            if (u + 1 < 32) p_cur = __builtin_amdgcn_exp2f(score(u + 1));
            psum += pu;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < NO; ++a) O[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(vu[a], pu, O[a], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        l_run = l_run * corr + psum;
        __syncthreads();
    }

    const float inv = 1.0f / (l_run + __shfl_xor(l_run, 32, 64));
    if (q0 + c < own) {  // own <= K; the slots behind it are another patch's points
        float *orow = out + qrow * C + head * D;
#pragma unroll
        for (int a = 0; a < NO; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) {  // registers 4g..4g+3 are the consecutive d = 32 a + 8 g + 4 hh + (0..3)
                const int d = 32 * a + 8 * g + 4 * hh;
                if (d < D)
                    *reinterpret_cast<float4 *>(orow + d) = make_float4(O[a][4 * g] * inv, O[a][4 * g + 1] * inv,
                                                                        O[a][4 * g + 2] * inv, O[a][4 * g + 3] * inv);
            }
    }
}

// ---- pooling / normalisation ------------------------------------------------------------------------------------------
// one wave per cluster: max over rows members[seg[j] .. seg[j+1]) of x, then y = gelu(x * scale + shift)
__global__ __launch_bounds__(256) void cluster_max_kernel(long long clusters, int C4, const float4 *__restrict__ x,
                                                          const long long *__restrict__ members,
                                                          const long long *__restrict__ seg,
                                                          const float4 *__restrict__ scale, const float4 *__restrict__ shift,
                                                          float4 *__restrict__ out) {
    const long long j = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= clusters) return;
    const int lane = threadIdx.x & 63;
    const long long beg = seg[j], end = seg[j + 1];
    for (int c = lane; c < C4; c += 64) {
        float4 m = x[members[beg] * C4 + c];
        for (long long r = beg + 1; r < end; ++r) {
            const float4 v = x[members[r] * C4 + c];
            m.x = fmaxf(m.x, v.x), m.y = fmaxf(m.y, v.y), m.z = fmaxf(m.z, v.z), m.w = fmaxf(m.w, v.w);
        }
        const float4 s = scale[c], b = shift[c];
        out[j * C4 + c] = make_float4(gelu(m.x * s.x + b.x), gelu(m.y * s.y + b.y), gelu(m.z * s.z + b.z),
                                      gelu(m.w * s.w + b.w));
    }
}

__global__ __launch_bounds__(256) void bn_gelu_kernel(long long quads, int C4, const float4 *__restrict__ x,
                                                      const float4 *__restrict__ scale, const float4 *__restrict__ shift,
                                                      float4 *__restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= quads) return;
    const int c = (int)(i % C4);
    const float4 v = x[i], s = scale[c], b = shift[c];
    out[i] = make_float4(gelu(v.x * s.x + b.x), gelu(v.y * s.y + b.y), gelu(v.z * s.z + b.z), gelu(v.w * s.w + b.w));
}

__global__ __launch_bounds__(256) void unpool_merge_kernel(long long quads, int C4, const float4 *__restrict__ x,
                                                           const float4 *__restrict__ scale, const float4 *__restrict__ shift,
                                                           const float4 *__restrict__ up, const long long *__restrict__ cluster,
                                                           float4 *__restrict__ skip, float4 *__restrict__ sum) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= quads) return;
    const long long row = i / C4;
    const int c = (int)(i - row * C4);
    const float4 v = x[i], s = scale[c], b = shift[c];
    const float4 k = make_float4(gelu(v.x * s.x + b.x), gelu(v.y * s.y + b.y), gelu(v.z * s.z + b.z), gelu(v.w * s.w + b.w));
    const float4 u = up[cluster[row] * C4 + c];
    skip[i] = k;
    sum[i] = make_float4(k.x + u.x, k.y + u.y, k.z + u.z, k.w + u.w);
}

// ---- residual + LayerNorm passes of a Block (pointtransformer_v3.py:595-615) -------------------------------------------
//   s = base + (norm_a ? LN_a(x) : x);  n = LN_b(s)        rows of C = 32 .. 512 floats
// covers `feat + cpe.2(cpe.1(conv))` followed by norm1, and `feat + attn` followed by norm2: one pass over the rows
// instead of LayerNorm + add + LayerNorm.  G lanes share a row (C = 4 G V floats), 64 / G rows per wave; two-pass
// mean / variance on the registers.
template <int G, int V>
__global__ __launch_bounds__(256) void rows_norm_kernel(long long rows, const float4 *__restrict__ x,
                                                        const float4 *__restrict__ base, const float4 *__restrict__ wa,
                                                        const float4 *__restrict__ ba, const float4 *__restrict__ wb,
                                                        const float4 *__restrict__ bb, float eps,
                                                        float4 *__restrict__ out_sum, float4 *__restrict__ out_norm) {
    constexpr int kRow4 = G * V;  // float4s per row
    constexpr float kInvC = 1.0f / (4.0f * kRow4);
    const int lane = threadIdx.x & 63, sub = lane % G;
    const long long row = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * (64 / G) + lane / G;
    const bool live = row < rows;
    const long long r = live ? row : rows - 1;
    auto group_sum = [](float v) {
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        return v;
    };
    auto normalise = [&](float4 (&t)[V], const float4 *w, const float4 *b) {
        float sum = 0.f;
#pragma unroll
        for (int v = 0; v < V; ++v) sum += (t[v].x + t[v].y) + (t[v].z + t[v].w);
        const float mean = group_sum(sum) * kInvC;
        float var = 0.f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const float dx = t[v].x - mean, dy = t[v].y - mean, dz = t[v].z - mean, dw = t[v].w - mean;
            var += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
        const float rstd = 1.0f / sqrtf(group_sum(var) * kInvC + eps);
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const float4 wv = w[sub + G * v], bv = b[sub + G * v];
            t[v] = make_float4((t[v].x - mean) * rstd * wv.x + bv.x, (t[v].y - mean) * rstd * wv.y + bv.y,
                               (t[v].z - mean) * rstd * wv.z + bv.z, (t[v].w - mean) * rstd * wv.w + bv.w);
        }
    };
    float4 t[V];
#pragma unroll
    for (int v = 0; v < V; ++v) t[v] = x[r * kRow4 + sub + G * v];
    if (wa) normalise(t, wa, ba);
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const float4 h = base[r * kRow4 + sub + G * v];
        t[v] = make_float4(h.x + t[v].x, h.y + t[v].y, h.z + t[v].z, h.w + t[v].w);
        if (live) out_sum[r * kRow4 + sub + G * v] = t[v];
    }
    normalise(t, wb, bb);
#pragma unroll
    for (int v = 0; v < V; ++v)
        if (live) out_norm[r * kRow4 + sub + G * v] = t[v];
}

}  // namespace cloud
}  // namespace amav

using namespace amav;

static inline unsigned blocks_for(long long threads) { return (unsigned)((threads + 255) / 256); }
static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int amav_cloud_voxelize(int64_t n, int clouds, const float *points, const int32_t *cloud_of, float resolution,
                                   int32_t *grid, int32_t *cloud_depth, int32_t *bounds, void *stream_) {
    AMAV_REQUIRE(n > 0 && clouds > 0 && clouds < 32768, "amav_cloud_voxelize: bad sizes n=%lld clouds=%d", (long long)n, clouds);
    AMAV_REQUIRE(points && cloud_of && grid && cloud_depth && bounds, "amav_cloud_voxelize: NULL pointer");
    AMAV_REQUIRE(resolution > 0.f, "amav_cloud_voxelize: resolution must be positive");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    cloud::bounds_init_kernel<<<blocks_for(clouds * 6), 256, 0, stream>>>(clouds, bounds);
    cloud::bounds_kernel<<<blocks_for(n), 256, 0, stream>>>(n, points, cloud_of, resolution, bounds);
    cloud::grid_kernel<<<blocks_for(n > clouds ? n : clouds), 256, 0, stream>>>(n, clouds, points, cloud_of, resolution, bounds,
                                                                             grid, cloud_depth);
    return check_launch("amav_cloud_voxelize");
}

extern "C" int amav_cloud_codes(int64_t n, const int32_t *grid, const int32_t *cloud_of, const int32_t *cloud_depth,
                                int64_t *keys, void *stream) {
    AMAV_REQUIRE(n > 0, "amav_cloud_codes: bad size n=%lld", (long long)n);
    AMAV_REQUIRE(grid && cloud_of && cloud_depth && keys, "amav_cloud_codes: NULL pointer");
    cloud::codes_kernel<<<blocks_for(n), 256, 0, static_cast<hipStream_t>(stream)>>>(
        n, grid, cloud_of, cloud_depth, reinterpret_cast<long long *>(keys));
    return check_launch("amav_cloud_codes");
}

extern "C" int amav_cloud_neighbors(int64_t n, int ksize, const int32_t *grid, const int32_t *cloud_of,
                                    const int32_t *cloud_depth, const int32_t *cloud_start, const int64_t *sorted_keys,
                                    const int64_t *order, int32_t *nbr, void *stream) {
    AMAV_REQUIRE(n > 0 && n < INT_MAX && (ksize == 3 || ksize == 5), "amav_cloud_neighbors: bad sizes n=%lld ksize=%d",
                 (long long)n, ksize);
    AMAV_REQUIRE(grid && cloud_of && cloud_depth && cloud_start && sorted_keys && order && nbr,
                 "amav_cloud_neighbors: NULL pointer");
    const long long threads = (long long)n * ksize * ksize * ksize;
    cloud::neighbors_kernel<<<blocks_for(threads), 256, 0, static_cast<hipStream_t>(stream)>>>(
        n, ksize, grid, cloud_of, cloud_depth, cloud_start, reinterpret_cast<const long long *>(sorted_keys),
        reinterpret_cast<const long long *>(order), nbr);
    return check_launch("amav_cloud_neighbors");
}

extern "C" int amav_subm_pair_gemm(int64_t pairs, int tiles, int taps, int cin, int cout, const float *feat,
                                   const int32_t *pair_src, const int32_t *tap_start, const int32_t *tile_start,
                                   const float *weights, float *products, void *stream_) {
    AMAV_REQUIRE(pairs > 0 && pairs < INT_MAX && tiles > 0 && taps > 0, "amav_subm_pair_gemm: bad sizes pairs=%lld tiles=%d taps=%d",
                 (long long)pairs, tiles, taps);
    AMAV_REQUIRE(cin > 0 && cin % 32 == 0 && cout > 0 && cout % 32 == 0,
                 "amav_subm_pair_gemm: channels must be multiples of 32 (C_in %d, C_out %d)", cin, cout);
    AMAV_REQUIRE(feat && pair_src && tap_start && tile_start && weights && products, "amav_subm_pair_gemm: NULL pointer");
    AMAV_REQUIRE(aligned16(feat) && aligned16(weights) && aligned16(products), "amav_subm_pair_gemm: buffers must be 16-byte aligned");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int nt = cout % 128 == 0 ? 128 : (cout % 64 == 0 ? 64 : 32);
    const dim3 grid((unsigned)tiles, (unsigned)(cout / nt));
    if (nt == 128)
        cloud::pair_gemm_kernel<128><<<grid, 256, 0, stream>>>(feat, pair_src, tap_start, tile_start, taps, weights, products, cin, cout);
    else if (nt == 64)
        cloud::pair_gemm_kernel<64><<<grid, 256, 0, stream>>>(feat, pair_src, tap_start, tile_start, taps, weights, products, cin, cout);
    else
        cloud::pair_gemm_kernel<32><<<grid, 256, 0, stream>>>(feat, pair_src, tap_start, tile_start, taps, weights, products, cin, cout);
    return check_launch("amav_subm_pair_gemm");
}

static size_t subm_split_bytes(int taps, int cin, int cout) {
    return cloud::kSplitHeader + (size_t)taps * cin * cout * 2 * sizeof(_Float16);
}

extern "C" size_t amav_subm_weights_split_bytes(int taps, int cin, int cout) {
    if (taps <= 0 || cin <= 0 || cin % 32 || cout <= 0 || cout % 32) return 0;
    return subm_split_bytes(taps, cin, cout);
}

extern "C" int amav_subm_prepare_weights_split(int taps, int cin, int cout, const float *weights, void *out,
                                               size_t out_bytes, void *stream_) {
    AMAV_REQUIRE(taps > 0 && cin > 0 && cin % 32 == 0 && cout > 0 && cout % 32 == 0,
                 "amav_subm_prepare_weights_split: taps=%d, channels must be multiples of 32 (C_in %d, C_out %d)", taps, cin, cout);
    AMAV_REQUIRE(weights && out && aligned16(weights) && (reinterpret_cast<uintptr_t>(out) & 255) == 0,
                 "amav_subm_prepare_weights_split: NULL / misaligned pointer (out: 256 bytes)");
    if (out_bytes < subm_split_bytes(taps, cin, cout))
        return fail(AMAV_ERR_WORKSPACE, "amav_subm_prepare_weights_split: buffer %zu < required %zu", out_bytes,
                    subm_split_bytes(taps, cin, cout));
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    unsigned *hdr = static_cast<unsigned *>(out);
    AMAV_REQUIRE(zero_async(hdr, cloud::kSplitHeader, stream) == hipSuccess, "amav_subm_prepare_weights_split: header clear failed");
    const long long n4 = (long long)taps * cin * cout / 4;
    cloud::absmax_kernel<<<(unsigned)std::min<long long>((n4 + 255) / 256, 256), 256, 0, stream>>>(
        reinterpret_cast<const float4 *>(weights), n4, hdr);
    const long long items = (long long)taps * (cout / 32) * (cin / 16) * 64;
    cloud::subm_weights_split_kernel<<<blocks_for(items), 256, 0, stream>>>(
        weights, cin, cout, items, hdr, reinterpret_cast<_Float16 *>(static_cast<char *>(out) + cloud::kSplitHeader));
    return check_launch("amav_subm_prepare_weights_split");
}

extern "C" int amav_subm_pair_gemm_split(int64_t pairs, int tiles, int taps, int cin, int cout, int64_t n_rows,
                                         const float *feat, const int32_t *pair_src, const int32_t *tap_start,
                                         const int32_t *tile_start, const void *weights_split, void *scratch16,
                                         float *products, void *stream_) {
    AMAV_REQUIRE(pairs > 0 && pairs < INT_MAX && tiles > 0 && taps > 0 && n_rows > 0,
                 "amav_subm_pair_gemm_split: bad sizes pairs=%lld tiles=%d taps=%d rows=%lld", (long long)pairs, tiles, taps,
                 (long long)n_rows);
    AMAV_REQUIRE(cin > 0 && cin % 32 == 0 && cout > 0 && cout % 32 == 0,
                 "amav_subm_pair_gemm_split: channels must be multiples of 32 (C_in %d, C_out %d)", cin, cout);
    AMAV_REQUIRE(feat && pair_src && tap_start && tile_start && weights_split && scratch16 && products,
                 "amav_subm_pair_gemm_split: NULL pointer");
    AMAV_REQUIRE(aligned16(feat) && aligned16(weights_split) && aligned16(products) && aligned16(scratch16),
                 "amav_subm_pair_gemm_split: buffers must be 16-byte aligned");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    unsigned *amax = static_cast<unsigned *>(scratch16);
    AMAV_REQUIRE(zero_async(amax, 16, stream) == hipSuccess, "amav_subm_pair_gemm_split: scratch clear failed");
    const long long n4 = (long long)n_rows * cin / 4;
    cloud::absmax_kernel<<<(unsigned)std::min<long long>((n4 + 255) / 256, 256), 256, 0, stream>>>(
        reinterpret_cast<const float4 *>(feat), n4, amax);
    const int nt = cout % 128 == 0 ? 128 : (cout % 64 == 0 ? 64 : 32);
    const dim3 grid((unsigned)tiles, (unsigned)(cout / nt));
    if (nt == 128)
        cloud::pair_gemm_f16_kernel<128><<<grid, 256, 0, stream>>>(feat, pair_src, tap_start, tile_start, taps, weights_split, amax, products, cin, cout);
    else if (nt == 64)
        cloud::pair_gemm_f16_kernel<64><<<grid, 256, 0, stream>>>(feat, pair_src, tap_start, tile_start, taps, weights_split, amax, products, cin, cout);
    else
        cloud::pair_gemm_f16_kernel<32><<<grid, 256, 0, stream>>>(feat, pair_src, tap_start, tile_start, taps, weights_split, amax, products, cin, cout);
    return check_launch("amav_subm_pair_gemm_split");
}

extern "C" int amav_subm_pair_sum(int64_t n, int taps, int cout, const float *products, const int32_t *pair_of,
                                  const float *bias, float *out, void *stream) {
    AMAV_REQUIRE(n > 0 && taps > 0 && cout > 0 && cout % 4 == 0, "amav_subm_pair_sum: bad sizes n=%lld taps=%d cout=%d",
                 (long long)n, taps, cout);
    AMAV_REQUIRE(products && pair_of && out, "amav_subm_pair_sum: NULL pointer");
    AMAV_REQUIRE(aligned16(products) && aligned16(out) && (!bias || aligned16(bias)), "amav_subm_pair_sum: buffers must be 16-byte aligned");
    cloud::pair_sum_kernel<<<blocks_for((long long)n * (cout / 4)), 256, 0, static_cast<hipStream_t>(stream)>>>(
        n, taps, cout / 4, reinterpret_cast<const float4 *>(products), pair_of, reinterpret_cast<const float4 *>(bias),
        reinterpret_cast<float4 *>(out));
    return check_launch("amav_subm_pair_sum");
}

extern "C" int amav_patch_attention(int patches, int max_patch, int heads, int head_dim, const float *qkv,
                                    const int64_t *order, const int32_t *patch_desc, float *out, float scale,
                                    void *stream_) {
    AMAV_REQUIRE(patches > 0 && patches <= 65535 && heads > 0 && heads <= 65535 && max_patch > 0,
                 "amav_patch_attention: bad sizes patches=%d heads=%d max_patch=%d", patches, heads, max_patch);
    AMAV_REQUIRE(head_dim == 16 || head_dim == 32 || head_dim == 64, "amav_patch_attention: head_dim %d (16, 32, 64 are built)",
                 head_dim);
    AMAV_REQUIRE(qkv && order && patch_desc && out, "amav_patch_attention: NULL pointer");
    AMAV_REQUIRE(aligned16(qkv) && aligned16(out) && aligned16(patch_desc), "amav_patch_attention: buffers must be 16-byte aligned");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const dim3 grid((unsigned)((max_patch + 127) / 128), (unsigned)heads, (unsigned)patches);
    const int C = heads * head_dim;
    const float sl = scale * 1.4426950408889634f;
    const long long *ord = reinterpret_cast<const long long *>(order);
    const int4 *pd = reinterpret_cast<const int4 *>(patch_desc);
    if (head_dim == 16)
        cloud::patch_attention_kernel<16><<<grid, 256, 0, stream>>>(qkv, ord, pd, out, C, sl);
    else if (head_dim == 32)
        cloud::patch_attention_kernel<32><<<grid, 256, 0, stream>>>(qkv, ord, pd, out, C, sl);
    else
        cloud::patch_attention_kernel<64><<<grid, 256, 0, stream>>>(qkv, ord, pd, out, C, sl);
    return check_launch("amav_patch_attention");
}

extern "C" int amav_cluster_max(int64_t clusters, int channels, const float *x, const int64_t *members, const int64_t *seg,
                                const float *scale, const float *shift, float *out, void *stream) {
    AMAV_REQUIRE(clusters > 0 && channels > 0 && channels % 4 == 0, "amav_cluster_max: bad sizes clusters=%lld channels=%d",
                 (long long)clusters, channels);
    AMAV_REQUIRE(x && members && seg && scale && shift && out, "amav_cluster_max: NULL pointer");
    AMAV_REQUIRE(aligned16(x) && aligned16(scale) && aligned16(shift) && aligned16(out), "amav_cluster_max: buffers must be 16-byte aligned");
    cloud::cluster_max_kernel<<<(unsigned)((clusters + 3) / 4), 256, 0, static_cast<hipStream_t>(stream)>>>(
        clusters, channels / 4, reinterpret_cast<const float4 *>(x), reinterpret_cast<const long long *>(members),
        reinterpret_cast<const long long *>(seg), reinterpret_cast<const float4 *>(scale),
        reinterpret_cast<const float4 *>(shift), reinterpret_cast<float4 *>(out));
    return check_launch("amav_cluster_max");
}

extern "C" int amav_bn_gelu(int64_t rows, int channels, const float *x, const float *scale, const float *shift, float *out,
                            void *stream) {
    AMAV_REQUIRE(rows > 0 && channels > 0 && channels % 4 == 0, "amav_bn_gelu: bad sizes rows=%lld channels=%d",
                 (long long)rows, channels);
    AMAV_REQUIRE(x && scale && shift && out, "amav_bn_gelu: NULL pointer");
    AMAV_REQUIRE(aligned16(x) && aligned16(scale) && aligned16(shift) && aligned16(out), "amav_bn_gelu: buffers must be 16-byte aligned");
    const long long quads = (long long)rows * (channels / 4);
    cloud::bn_gelu_kernel<<<blocks_for(quads), 256, 0, static_cast<hipStream_t>(stream)>>>(
        quads, channels / 4, reinterpret_cast<const float4 *>(x), reinterpret_cast<const float4 *>(scale),
        reinterpret_cast<const float4 *>(shift), reinterpret_cast<float4 *>(out));
    return check_launch("amav_bn_gelu");
}

extern "C" int amav_rows_norm(int64_t rows, int channels, const float *x, const float *base, const float *weight_a,
                              const float *bias_a, const float *weight_b, const float *bias_b, float eps, float *out_sum,
                              float *out_norm, void *stream_) {
    AMAV_REQUIRE(rows > 0 && (channels == 32 || channels == 64 || channels == 128 || channels == 256 || channels == 512),
                 "amav_rows_norm: rows=%lld channels=%d (32, 64, 128, 256, 512 are built)", (long long)rows, channels);
    AMAV_REQUIRE(x && base && weight_b && bias_b && out_sum && out_norm && (!weight_a == !bias_a), "amav_rows_norm: NULL pointer");
    AMAV_REQUIRE(aligned16(x) && aligned16(base) && aligned16(weight_b) && aligned16(bias_b) && aligned16(out_sum) &&
                     aligned16(out_norm) && (!weight_a || (aligned16(weight_a) && aligned16(bias_a))),
                 "amav_rows_norm: buffers must be 16-byte aligned");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    auto f4 = [](const float *p) { return reinterpret_cast<const float4 *>(p); };
#define AMAV_ROWS_NORM(G_, V_)                                                                                       \
    cloud::rows_norm_kernel<G_, V_><<<(unsigned)((rows + 4 * (64 / G_) - 1) / (4 * (64 / G_))), 256, 0, stream>>>(   \
        rows, f4(x), f4(base), f4(weight_a), f4(bias_a), f4(weight_b), f4(bias_b), eps,                              \
        reinterpret_cast<float4 *>(out_sum), reinterpret_cast<float4 *>(out_norm))
    if (channels == 32) AMAV_ROWS_NORM(8, 1);
    else if (channels == 64) AMAV_ROWS_NORM(16, 1);
    else if (channels == 128) AMAV_ROWS_NORM(32, 1);
    else if (channels == 256) AMAV_ROWS_NORM(64, 1);
    else AMAV_ROWS_NORM(64, 2);
#undef AMAV_ROWS_NORM
    return check_launch("amav_rows_norm");
}

extern "C" int amav_unpool_merge(int64_t rows, int channels, const float *x, const float *scale, const float *shift,
                                 const float *up, const int64_t *cluster, float *skip, float *sum, void *stream) {
    AMAV_REQUIRE(rows > 0 && channels > 0 && channels % 4 == 0, "amav_unpool_merge: bad sizes rows=%lld channels=%d",
                 (long long)rows, channels);
    AMAV_REQUIRE(x && scale && shift && up && cluster && skip && sum, "amav_unpool_merge: NULL pointer");
    AMAV_REQUIRE(aligned16(x) && aligned16(scale) && aligned16(shift) && aligned16(up) && aligned16(skip) && aligned16(sum),
                 "amav_unpool_merge: buffers must be 16-byte aligned");
    const long long quads = (long long)rows * (channels / 4);
    cloud::unpool_merge_kernel<<<blocks_for(quads), 256, 0, static_cast<hipStream_t>(stream)>>>(
        quads, channels / 4, reinterpret_cast<const float4 *>(x), reinterpret_cast<const float4 *>(scale),
        reinterpret_cast<const float4 *>(shift), reinterpret_cast<const float4 *>(up),
        reinterpret_cast<const long long *>(cluster), reinterpret_cast<float4 *>(skip), reinterpret_cast<float4 *>(sum));
    return check_launch("amav_unpool_merge");
}
