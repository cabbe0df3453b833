// Self-attention of the audio transformer on the gfx950 matrix cores, exact fp32.
//
// Replaces diffusers' Attention -> F.scaled_dot_product_attention as reached from BasicTransformerBlock.attn1
// (src/models/transformers.py:329-336; 8 heads x 64, S = 2 * (3 * 32^2 + 80) = 6304 tokens, no mask).  This is the
// only dense contraction of the hot path (the audio cross-attention has ONE key, so it is a
// broadcast, not a contraction); it runs on v_mfma_f32_32x32x2_f32, whose products and sums are exact IEEE fp32
// (the reference runs fp32 SDPA; bf16 would threaten the 1e-3 image bound through the autoregressive recurrence).
//
// Flash-style, one pass over the keys with an online softmax, never materialising the S x S scores:
//   * a workgroup = 4 waves = 128 queries of one (batch, head); each wave owns 32 queries for the whole key sweep;
//   * the product is computed TRANSPOSED, S^T = K Q^T (keys on the MFMA rows, queries on the columns): a lane then
//     holds one query's scores in its accumulator registers, so the row max / row sum are register reductions plus
//     one exchange between the two lane halves, and P^T is already in the B-operand layout of the next product
//     O^T = V^T P^T -- the probabilities never leave the registers;
//   * K is staged in LDS transposed ([d][key], padded) and V row-major, which makes every operand fetch a
//     conflict-free 32-lane row read; the next K/V tile is prefetched into registers under the current tile's MFMAs.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "amav_common.h"

namespace amav {
namespace attn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kD = 64;        // head dim
constexpr int kBM = 128;      // queries per workgroup (4 waves x 32)
constexpr int kBN = 64;       // keys per tile
constexpr int kLdk = kBN + 1; // padded row of the transposed K tile

// nsplit > 1: the key range is cut into nsplit slices handled by different workgroups (finer tasks balance the
// 1576 wave-tasks of the reference shape over 1024 SIMDs); each slice writes its un-normalised O, running max and
// sum to `part`, and combine_kernel merges them.
__global__ __launch_bounds__(256, 3) void selfattn_kernel(const float *__restrict__ q, const float *__restrict__ k,
                                                       const float *__restrict__ v, float *__restrict__ out, int S,
                                                       long long row_stride, long long out_row_stride,
                                                       float scale_log2e, int nsplit, float *__restrict__ part) {
    __shared__ float Kt[kD * kLdk];   // [d][key]
    __shared__ float Vs[kBN * kD];    // [key][d]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c = lane & 31, hh = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int split = blockIdx.x % nsplit;
    const int q0 = (blockIdx.x / nsplit) * kBM + wave * 32;
    const size_t base = (size_t)b * S;

    // Q fragment (B operand of S^T = K Q^T): Q[q0 + c][2 s + hh], pre-scaled by softmax_scale * log2(e)
    float Qr[32];
    {
        const float *qrow = q + (base + min(q0 + c, S - 1)) * row_stride + head * kD;
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const float2 t = *reinterpret_cast<const float2 *>(qrow + 2 * s);
            Qr[s] = (hh ? t.y : t.x) * scale_log2e;
        }
    }

    f32x16 O0, O1;  // O^T: rows d (0..31 / 32..63), column = this lane's query
#pragma unroll
    for (int t = 0; t < 16; ++t) O0[t] = 0.f, O1[t] = 0.f;
    float m_run = -1e30f, l_run = 0.f;

    // staging map: thread -> key (tid / 16 + 16 i), 4 consecutive d at (tid % 16) * 4
    const int skey = tid >> 4, sd = (tid & 15) * 4;
    float4 k0, k1, k2, k3, v0, v1, v2, v3;  // named registers: an indexed array here is demoted to scratch
#define AMAV_LOAD_ONE(kt_, i_, kr_, vr_)                                             \
    {                                                                                \
        const int key_ = min((kt_) * kBN + skey + 16 * (i_), S - 1);                 \
        const size_t off_ = (base + key_) * row_stride + head * kD + sd;             \
        kr_ = *reinterpret_cast<const float4 *>(k + off_);                           \
        vr_ = *reinterpret_cast<const float4 *>(v + off_);                           \
    }
#define AMAV_LOAD_TILE(kt_) \
    AMAV_LOAD_ONE(kt_, 0, k0, v0) AMAV_LOAD_ONE(kt_, 1, k1, v1) AMAV_LOAD_ONE(kt_, 2, k2, v2) AMAV_LOAD_ONE(kt_, 3, k3, v3)
#define AMAV_STAGE_ONE(i_, kr_, vr_)                                  \
    {                                                                 \
        const int key_ = skey + 16 * (i_);                            \
        Kt[(sd + 0) * kLdk + key_] = kr_.x;                           \
        Kt[(sd + 1) * kLdk + key_] = kr_.y;                           \
        Kt[(sd + 2) * kLdk + key_] = kr_.z;                           \
        Kt[(sd + 3) * kLdk + key_] = kr_.w;                           \
        *reinterpret_cast<float4 *>(&Vs[key_ * kD + sd]) = vr_;      \
    }
    const int ntiles_all = (S + kBN - 1) / kBN;
    const int kt_begin = (int)((long long)ntiles_all * split / nsplit);
    const int ntiles = (int)((long long)ntiles_all * (split + 1) / nsplit);  // end of this slice
    AMAV_LOAD_TILE(kt_begin)
    for (int kt = kt_begin; kt < ntiles; ++kt) {
        // ---- stage tile kt (K transposed, V as is)
        AMAV_STAGE_ONE(0, k0, v0) AMAV_STAGE_ONE(1, k1, v1) AMAV_STAGE_ONE(2, k2, v2) AMAV_STAGE_ONE(3, k3, v3)
        __syncthreads();
        if (kt + 1 < ntiles) {  // in flight under this tile's MFMAs
            AMAV_LOAD_TILE(kt + 1)
        }

        // ---- S^T = K Q^T for the two 32-key halves of the tile
        f32x16 S0, S1;
#pragma unroll
        for (int t = 0; t < 16; ++t) S0[t] = 0.f, S1[t] = 0.f;
        // the K operands of step s + 2 are read from LDS before the MFMAs of step s issue: read just in time, every
        // MFMA pair waited a full LDS round trip (ds_read; s_waitcnt 0; mfma; mfma), which the other waves of the SIMD
        // only partly covered
        float ka[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            ka[u][0] = Kt[(2 * u + hh) * kLdk + c];
            ka[u][1] = Kt[(2 * u + hh) * kLdk + 32 + c];
        }
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const float a0 = ka[s & 1][0], a1 = ka[s & 1][1];
            if (s + 2 < 32) {
                ka[s & 1][0] = Kt[(2 * (s + 2) + hh) * kLdk + c];
                ka[s & 1][1] = Kt[(2 * (s + 2) + hh) * kLdk + 32 + c];
            }
            __builtin_amdgcn_sched_barrier(0);
            S0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, Qr[s], S0, 0, 0, 0);
            S1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, Qr[s], S1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);  // pin the order: read of step s + 2, then this step's two MFMAs
        }
        // accumulator register t of key half kb holds key  kt*64 + kb*32 + (t&3) + 8*(t>>2) + 4*hh
        if ((kt + 1) * kBN > S) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int kk = kt * kBN + (t & 3) + 8 * (t >> 2) + 4 * hh;
                if (kk >= S) S0[t] = -1e30f;
                if (kk + 32 >= S) S1[t] = -1e30f;
            }
        }
        // ---- online softmax over this lane's query
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
        for (int t = 0; t < 16; ++t) O0[t] *= corr, O1[t] *= corr;

        // ---- O^T += V^T P^T: score register t of key half kb is the B operand of k-step {key r, key r + 4}, with
        // r = 32 kb + (t & 3) + 8 (t >> 2) + 4 hh.  Same pinned pipeline as above: the V operands of step u + 2 are
        // read and the probability of step u + 1 is exponentiated before the two MFMAs of step u issue, so the LDS
        // round trip and the quarter-rate exp run under the matrix pipe.
        auto vrow = [&](int u) { return ((u >> 4) * 32 + (u & 3) + 8 * ((u & 15) >> 2) + 4 * hh) * kD + c; };
        auto score = [&](int u) { return (u < 16 ? S0[u] : S1[u - 16]) - m_new; };
        float va[3][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) va[u][0] = Vs[vrow(u)], va[u][1] = Vs[vrow(u) + 32];
        float p_cur = __builtin_amdgcn_exp2f(score(0)), psum = 0.f;
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const float v0 = va[u % 3][0], v1 = va[u % 3][1], pu = p_cur;
            if (u + 2 < 32) va[(u + 2) % 3][0] = Vs[vrow(u + 2)], va[(u + 2) % 3][1] = Vs[vrow(u + 2) + 32];
            if (u + 1 < 32) p_cur = __builtin_amdgcn_exp2f(score(u + 1));
            psum += pu;
            __builtin_amdgcn_sched_barrier(0);
            O0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, pu, O0, 0, 0, 0);
            O1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, pu, O1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        l_run = l_run * corr + psum;  // per-half partial sum; the halves are added once at the end
        __syncthreads();  // every wave is done with this tile before it is overwritten
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    if (nsplit > 1) {
        // partial record of (split, b, head, query): 64 un-normalised O values, then m and l
        if (q0 + c < S) {
            float *prow = part + ((((size_t)split * gridDim.z + b) * gridDim.y + head) * S + q0 + c) * (kD + 2);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = 8 * g + 4 * hh;
                prow[d] = O0[4 * g], prow[d + 1] = O0[4 * g + 1], prow[d + 2] = O0[4 * g + 2], prow[d + 3] = O0[4 * g + 3];
                prow[32 + d] = O1[4 * g], prow[33 + d] = O1[4 * g + 1], prow[34 + d] = O1[4 * g + 2];
                prow[35 + d] = O1[4 * g + 3];
            }
            if (hh == 0) prow[kD] = m_run, prow[kD + 1] = l_tot;
        }
        return;
    }
    const float inv = 1.0f / l_tot;
    if (q0 + c < S) {
        float *orow = out + (base + q0 + c) * out_row_stride + head * kD;
#pragma unroll
        for (int g = 0; g < 4; ++g) {  // registers 4g..4g+3 are 4 consecutive d: 8g + 4hh + (0..3)
            const int d = 8 * g + 4 * hh;
            *reinterpret_cast<float4 *>(orow + d) =
                make_float4(O0[4 * g] * inv, O0[4 * g + 1] * inv, O0[4 * g + 2] * inv, O0[4 * g + 3] * inv);
            *reinterpret_cast<float4 *>(orow + 32 + d) =
                make_float4(O1[4 * g] * inv, O1[4 * g + 1] * inv, O1[4 * g + 2] * inv, O1[4 * g + 3] * inv);
        }
    }
}

// One thread per (b, head, query, 4 consecutive d): merge the nsplit partial softmax states.  kSplitOut: the result goes
// out as the fp16 x 2 operand of the projection that follows (rows of [h2 | h1 | h1], K = H * 64 each, x 2^scale_exp =
// h1 + h2: exactly what split_operand_f16_kernel makes of the fp32 result, without the pass over it).
typedef _Float16 c_f16x4 __attribute__((ext_vector_type(4)));
template <bool kSplitOut>
__global__ __launch_bounds__(256) void combine_kernel(const float *__restrict__ part, float *__restrict__ out, int B,
                                                      int H, int S, int nsplit, long long out_row_stride,
                                                      _Float16 *__restrict__ out_split, float prescale) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int d4 = (int)(gid & 15);
    const long long row = gid >> 4;  // (b * H + head) * S + query
    if (row >= (long long)B * H * S) return;
    const size_t slice = (size_t)B * H * S * (kD + 2);
    const float *p0 = part + (size_t)row * (kD + 2);
    float m = -1e30f;
    for (int s = 0; s < nsplit; ++s) m = fmaxf(m, p0[s * slice + kD]);
    float l = 0.f, o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
    for (int s = 0; s < nsplit; ++s) {
        const float *p = p0 + s * slice;
        const float w = __builtin_amdgcn_exp2f(p[kD] - m);
        l += p[kD + 1] * w;
        o0 += p[4 * d4] * w, o1 += p[4 * d4 + 1] * w, o2 += p[4 * d4 + 2] * w, o3 += p[4 * d4 + 3] * w;
    }
    const float inv = 1.0f / l;
    const int qi = (int)(row % S), head = (int)((row / S) % H), b = (int)(row / ((long long)S * H));
    if (kSplitOut) {
        const float v[4] = {o0 * inv, o1 * inv, o2 * inv, o3 * inv};
        c_f16x4 p1, p2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float x = v[j] * prescale;
            const _Float16 a = (_Float16)x;
            p1[j] = a, p2[j] = (_Float16)__builtin_fmaf((float)a, -1.0f, x);
        }
        const size_t K = (size_t)H * kD;
        _Float16 *dst = out_split + ((size_t)b * S + qi) * 3 * K + head * kD + 4 * d4;
        *reinterpret_cast<c_f16x4 *>(dst) = p2;
        *reinterpret_cast<c_f16x4 *>(dst + K) = p1;
        *reinterpret_cast<c_f16x4 *>(dst + 2 * K) = p1;
        return;
    }
    float *orow = out + ((size_t)b * S + qi) * out_row_stride + head * kD + 4 * d4;
    *reinterpret_cast<float4 *>(orow) = make_float4(o0 * inv, o1 * inv, o2 * inv, o3 * inv);
}

// ---------------------------------------------------------------------------------------------------------------
// The same attention on the bf16 matrix instructions, fp32-equivalent: every fp32 operand x is split into three bf16
// parts x = x1 + x2 + x3 (x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2): 24 mantissa bits, the subtractions
// are exact), and a product is the six partial products x_i y_j with i + j <= 4, each EXACT in the fp32 accumulator
// of v_mfma_f32_32x32x16_bf16; the dropped terms are below 2^-24 of the product, the size of one fp32 rounding
// (tools/attention_accuracy.py: 1.6e-7 max abs against fp64, the fp32-MFMA kernel 2.3e-7, the library's SDPA 4.8e-7).
// Six bf16 MFMAs cost 6/16 of the fp32 MFMA they replace (the bf16 pipe is 16x the fp32 one on gfx950).
//   split_kv_kernel: K parts [3][B H][S][64] and V^T parts [3][B H][64][S_pad] in bf16, written once per call (every
//     one of the 50 query tiles would otherwise split the same K / V again);
//   selfattn_split_kernel: same decomposition as selfattn_kernel (128 queries per workgroup, transposed scores so the
//     softmax stays in registers, key range split over workgroups); S^T = sum K_i Q_j^T with K rows read straight from
//     a row-major LDS tile (16 B per lane), P^T split in registers and used as the B operand of O^T = sum V_i^T P_j^T
//     (accumulator-as-operand: its k order inside a step is key 16 s + 8 (j >> 2) + 4 h + (j & 3), which is how the V^T
//     fragments are gathered).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int kLdK = kD + 8;   // bf16 per K row in LDS (144 B: 16-byte reads of 8 consecutive lanes cover all banks)
constexpr int kLdV = kBN + 4;  // bf16 per V^T row in LDS (136 B: 8-byte reads of 16 consecutive lanes cover all banks)

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// Two-way fp16 split: x = h1 + h2 to 22 bits when the residual stays out of fp16's subnormals, which the callers arrange
// by pre-scaling x with a power of two (exact) so that the tensor's bound sits just under fp16's maximum.
__device__ __forceinline__ void split2(float x, _Float16 &a, _Float16 &b) {
    a = (_Float16)x;
    b = (_Float16)__builtin_fmaf((float)a, -1.0f, x);  // x - a, exact; written as an fma so it can be one mixed-precision op
}

__device__ __forceinline__ void split3(float x, __bf16 &a, __bf16 &b, __bf16 &c) {
    a = (__bf16)x;
    const float r1 = x - (float)a;
    b = (__bf16)r1;
    c = (__bf16)(r1 - (float)b);
}

// grid (key tiles of 64, H, B), 256 threads: a [64 keys][64 d] tile of K and of V of one head
__global__ __launch_bounds__(256) void split_kv_kernel(const float *__restrict__ k, const float *__restrict__ v, int S,
                                                       int Spad, long long row_stride, __bf16 *__restrict__ Kp,
                                                       __bf16 *__restrict__ Vt) {
    __shared__ float vt[kD][kBN + 1];
    const int tid = threadIdx.x, head = blockIdx.y, b = blockIdx.z, H = gridDim.y, B = gridDim.z;
    const int key0 = blockIdx.x * kBN;
    const size_t bh = (size_t)b * H + head, part_k = (size_t)B * H * S * kD, part_v = (size_t)B * H * kD * Spad;
    // thread -> key tid / 4, 16 consecutive d at (tid % 4) * 16
    const int key = tid >> 2, d0 = (tid & 3) * 16;
    const bool live = key0 + key < S;
    const size_t src = ((size_t)b * S + min(key0 + key, S - 1)) * row_stride + head * kD + d0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 kv = *reinterpret_cast<const float4 *>(k + src + 4 * q);
        const float4 vv = *reinterpret_cast<const float4 *>(v + src + 4 * q);
        const float kk[4] = {kv.x, kv.y, kv.z, kv.w};
        bf16x4 p1, p2, p3;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            __bf16 a, bb, c;
            split3(kk[e], a, bb, c);
            p1[e] = a, p2[e] = bb, p3[e] = c;
        }
        if (live) {
            __bf16 *dst = Kp + (bh * S + key0 + key) * kD + d0 + 4 * q;
            *reinterpret_cast<bf16x4 *>(dst) = p1;
            *reinterpret_cast<bf16x4 *>(dst + part_k) = p2;
            *reinterpret_cast<bf16x4 *>(dst + 2 * part_k) = p3;
        }
        vt[d0 + 4 * q + 0][key] = live ? vv.x : 0.f;
        vt[d0 + 4 * q + 1][key] = live ? vv.y : 0.f;
        vt[d0 + 4 * q + 2][key] = live ? vv.z : 0.f;
        vt[d0 + 4 * q + 3][key] = live ? vv.w : 0.f;
    }
    __syncthreads();
    // thread -> d tid / 4, 16 consecutive keys at (tid % 4) * 16 (keys past S are zeros: the padded tail)
    const int d = tid >> 2, kq = (tid & 3) * 16;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        bf16x4 p1, p2, p3;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            __bf16 a, bb, c;
            split3(vt[d][kq + 4 * q + e], a, bb, c);
            p1[e] = a, p2[e] = bb, p3[e] = c;
        }
        __bf16 *dst = Vt + (bh * kD + d) * Spad + key0 + kq + 4 * q;
        *reinterpret_cast<bf16x4 *>(dst) = p1;
        *reinterpret_cast<bf16x4 *>(dst + part_v) = p2;
        *reinterpret_cast<bf16x4 *>(dst + 2 * part_v) = p3;
    }
}

// Workgroup -> work mapping, XCD-aware: the hardware hands consecutive workgroup ids to the 8 XCDs in turn, and each XCD
// has its own 4 MB L2.  Id i runs on XCD i % 8; the query tiles of ONE (batch, head, key-slice) -- which all stream the
// same ~1 MB of K / V parts -- are given to one XCD back to back, so a slice stays in that XCD's L2 while its tiles run.
// (Measured neutral at the reference shape, 0.55 ms either way: the kernel is not fetch-bound.)
__global__ __launch_bounds__(256, 2) void selfattn_split_kernel(const float *__restrict__ q, const __bf16 *__restrict__ Kp,
                                                             const __bf16 *__restrict__ Vt, float *__restrict__ out,
                                                             int S, int Spad, long long row_stride,
                                                             long long out_row_stride, float scale_log2e, int nsplit,
                                                             float *__restrict__ part, int H, int B, int q_tiles) {
    __shared__ __bf16 Ks[3][kBN * kLdK];  // [part][key][d]
    __shared__ __bf16 Vs[3][kD * kLdV];   // [part][d][key]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, hh = lane >> 5;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int slice = (j / q_tiles) * 8 + xcd;  // (batch, head, key-slice) index
    if (slice >= nsplit * H * B) return;        // whole workgroup: the grid is padded to 8 slices per round
    const int split = slice % nsplit, head = (slice / nsplit) % H, b = slice / (nsplit * H);
    const int q0 = (j % q_tiles) * kBM + wave * 32;
    const size_t bh = (size_t)b * H + head, part_k = (size_t)B * H * S * kD, part_v = (size_t)B * H * kD * Spad;

    // Q fragments (B operand of S^T = K Q^T): lane (r, hh) holds Q[q0 + r][16 s + 8 hh + j], pre-scaled, three parts
    bf16x8 Q1[4], Q2[4], Q3[4];
    {
        const float *qrow = q + ((size_t)b * S + min(q0 + r, S - 1)) * row_stride + head * kD;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float4 lo = *reinterpret_cast<const float4 *>(qrow + 16 * s + 8 * hh);
            const float4 hi = *reinterpret_cast<const float4 *>(qrow + 16 * s + 8 * hh + 4);
            const float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                __bf16 a, bb, c;
                split3(x[j] * scale_log2e, a, bb, c);
                Q1[s][j] = a, Q2[s][j] = bb, Q3[s][j] = c;
            }
        }
    }
    f32x16 O0, O1;
#pragma unroll
    for (int t = 0; t < 16; ++t) O0[t] = 0.f, O1[t] = 0.f;
    float m_run = -1e30f, l_run = 0.f;

    const int ntiles_all = (S + kBN - 1) / kBN;
    const int kt_begin = (int)((long long)ntiles_all * split / nsplit);
    const int ntiles = (int)((long long)ntiles_all * (split + 1) / nsplit);
    // staging map: 16-byte chunk number tid + 256 i of a [64 rows][128 B] tile: row (tid + 256 i) / 8, chunk % 8; the
    // next tile's twelve chunks are in flight under the current tile's MFMAs (named registers: an indexed array captured
    // by a lambda was demoted to scratch memory)
    const int srow0 = tid >> 3, sc8 = (tid & 7) * 8;  // chunk i: row srow0 + 32 i
    uint4 ka0, ka1, kb0, kb1, kc0, kc1, va0, va1, vb0, vb1, vc0, vc1;
#define AMAV_SP_LOAD(kt_)                                                                                          \
    {                                                                                                              \
        const int key0_ = (kt_) * kBN;                                                                             \
        const size_t k0_ = (bh * S + min(key0_ + srow0, S - 1)) * kD + sc8;                                        \
        const size_t k1_ = (bh * S + min(key0_ + srow0 + 32, S - 1)) * kD + sc8;                                   \
        const size_t v0_ = (bh * kD + srow0) * Spad + key0_ + sc8, v1_ = v0_ + (size_t)32 * Spad;                  \
        ka0 = *reinterpret_cast<const uint4 *>(Kp + k0_), ka1 = *reinterpret_cast<const uint4 *>(Kp + k1_);        \
        kb0 = *reinterpret_cast<const uint4 *>(Kp + part_k + k0_), kb1 = *reinterpret_cast<const uint4 *>(Kp + part_k + k1_); \
        kc0 = *reinterpret_cast<const uint4 *>(Kp + 2 * part_k + k0_), kc1 = *reinterpret_cast<const uint4 *>(Kp + 2 * part_k + k1_); \
        va0 = *reinterpret_cast<const uint4 *>(Vt + v0_), va1 = *reinterpret_cast<const uint4 *>(Vt + v1_);        \
        vb0 = *reinterpret_cast<const uint4 *>(Vt + part_v + v0_), vb1 = *reinterpret_cast<const uint4 *>(Vt + part_v + v1_); \
        vc0 = *reinterpret_cast<const uint4 *>(Vt + 2 * part_v + v0_), vc1 = *reinterpret_cast<const uint4 *>(Vt + 2 * part_v + v1_); \
    }
#define AMAV_SP_STAGE(p_, i_, kr_, vr_)                                                              \
    {                                                                                                \
        *reinterpret_cast<uint4 *>(&Ks[p_][(srow0 + 32 * (i_)) * kLdK + sc8]) = kr_;                 \
        uint2 *d_ = reinterpret_cast<uint2 *>(&Vs[p_][(srow0 + 32 * (i_)) * kLdV + sc8]); /* 136-byte rows: 8-byte aligned */ \
        d_[0] = make_uint2(vr_.x, vr_.y), d_[1] = make_uint2(vr_.z, vr_.w);                          \
    }
    AMAV_SP_LOAD(kt_begin)
    for (int kt = kt_begin; kt < ntiles; ++kt) {
        const int key0 = kt * kBN;
        AMAV_SP_STAGE(0, 0, ka0, va0) AMAV_SP_STAGE(0, 1, ka1, va1) AMAV_SP_STAGE(1, 0, kb0, vb0)
        AMAV_SP_STAGE(1, 1, kb1, vb1) AMAV_SP_STAGE(2, 0, kc0, vc0) AMAV_SP_STAGE(2, 1, kc1, vc1)
        __syncthreads();
        if (kt + 1 < ntiles) AMAV_SP_LOAD(kt + 1)

        // ---- S^T = K Q^T, one 32-key half at a time: six partial products per 16-wide k-step
        f32x16 S0, S1;
#pragma unroll
        for (int t = 0; t < 16; ++t) S0[t] = 0.f, S1[t] = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16 &Sx = kb ? S1 : S0;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int off = (r + 32 * kb) * kLdK + 16 * s + 8 * hh;
                const bf16x8 a1 = *reinterpret_cast<const bf16x8 *>(&Ks[0][off]);
                const bf16x8 a2 = *reinterpret_cast<const bf16x8 *>(&Ks[1][off]);
                const bf16x8 a3 = *reinterpret_cast<const bf16x8 *>(&Ks[2][off]);
                Sx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, Q1[s], Sx, 0, 0, 0);  // small terms first
                Sx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, Q2[s], Sx, 0, 0, 0);
                Sx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, Q3[s], Sx, 0, 0, 0);
                Sx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, Q1[s], Sx, 0, 0, 0);
                Sx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, Q2[s], Sx, 0, 0, 0);
                Sx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, Q1[s], Sx, 0, 0, 0);
            }
        }
        if ((kt + 1) * kBN > S) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int kk = key0 + (t & 3) + 8 * (t >> 2) + 4 * hh;
                if (kk >= S) S0[t] = -1e30f;
                if (kk + 32 >= S) S1[t] = -1e30f;
            }
        }
        // ---- online softmax over this lane's query (as in selfattn_kernel)
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
        for (int t = 0; t < 16; ++t) O0[t] *= corr, O1[t] *= corr;
        float psum = 0.f;
        // ---- O^T += V^T P^T: the probabilities of registers 8 s2 .. 8 s2 + 7 are the B operand of k-step s2; element j
        // of lane half hh is key 32 kb + 16 s2 + 8 (j >> 2) + 4 hh + (j & 3).  Software pipeline: the exponentials and
        // the three-way split of step u + 1 (~80 vector instructions) are issued between the twelve MFMAs of step u
        // (an MFMA holds the SIMD's issue port for 8 of its 32 cycles), pinned with sched_group_barrier
        auto parts = [&](int u, bf16x8 &P1, bf16x8 &P2, bf16x8 &P3) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float pv = __builtin_amdgcn_exp2f(((u >> 1) ? S1[8 * (u & 1) + j] : S0[8 * (u & 1) + j]) - m_new);
                psum += pv;
                __bf16 a, bb, c;
                split3(pv, a, bb, c);
                P1[j] = a, P2[j] = bb, P3[j] = c;
            }
        };
        bf16x8 Pc1, Pc2, Pc3, Pn1, Pn2, Pn3;
        parts(0, Pc1, Pc2, Pc3);
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // u = 2 kb + s2
            const int kbase = 32 * (u >> 1) + 16 * (u & 1) + 4 * hh;
            bf16x8 vf[2][3];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    const __bf16 *src = &Vs[p][(r + 32 * a) * kLdV + kbase];
                    const bf16x4 lo = *reinterpret_cast<const bf16x4 *>(src);
                    const bf16x4 hi = *reinterpret_cast<const bf16x4 *>(src + 8);
                    vf[a][p][0] = lo[0], vf[a][p][1] = lo[1], vf[a][p][2] = lo[2], vf[a][p][3] = lo[3];
                    vf[a][p][4] = hi[0], vf[a][p][5] = hi[1], vf[a][p][6] = hi[2], vf[a][p][7] = hi[3];
                }
            if (u + 1 < 4) parts(u + 1, Pn1, Pn2, Pn3);
            O0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[0][2], Pc1, O0, 0, 0, 0);
            O1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[1][2], Pc1, O1, 0, 0, 0);
            O0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[0][1], Pc2, O0, 0, 0, 0);
            O1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[1][1], Pc2, O1, 0, 0, 0);
            O0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[0][0], Pc3, O0, 0, 0, 0);
            O1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[1][0], Pc3, O1, 0, 0, 0);
            O0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[0][1], Pc1, O0, 0, 0, 0);
            O1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[1][1], Pc1, O1, 0, 0, 0);
            O0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[0][0], Pc2, O0, 0, 0, 0);
            O1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[1][0], Pc2, O1, 0, 0, 0);
            O0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[0][0], Pc1, O0, 0, 0, 0);
            O1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[1][0], Pc1, O1, 0, 0, 0);
            if (u + 1 < 4) {
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);  // seven vector instructions of the next step
                }
                Pc1 = Pn1, Pc2 = Pn2, Pc3 = Pn3;
            }
        }
        l_run = l_run * corr + psum;
        __syncthreads();
    }
#undef AMAV_SP_LOAD
#undef AMAV_SP_STAGE

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    if (nsplit > 1) {
        if (q0 + r < S) {
            float *prow = part + ((((size_t)split * B + b) * H + head) * S + q0 + r) * (kD + 2);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = 8 * g + 4 * hh;
                prow[d] = O0[4 * g], prow[d + 1] = O0[4 * g + 1], prow[d + 2] = O0[4 * g + 2], prow[d + 3] = O0[4 * g + 3];
                prow[32 + d] = O1[4 * g], prow[33 + d] = O1[4 * g + 1], prow[34 + d] = O1[4 * g + 2];
                prow[35 + d] = O1[4 * g + 3];
            }
            if (hh == 0) prow[kD] = m_run, prow[kD + 1] = l_tot;
        }
        return;
    }
    const float inv = 1.0f / l_tot;
    if (q0 + r < S) {
        float *orow = out + ((size_t)b * S + q0 + r) * out_row_stride + head * kD;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = 8 * g + 4 * hh;
            *reinterpret_cast<float4 *>(orow + d) =
                make_float4(O0[4 * g] * inv, O0[4 * g + 1] * inv, O0[4 * g + 2] * inv, O0[4 * g + 3] * inv);
            *reinterpret_cast<float4 *>(orow + 32 + d) =
                make_float4(O1[4 * g] * inv, O1[4 * g + 1] * inv, O1[4 * g + 2] * inv, O1[4 * g + 3] * inv);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// fp16 x 2 form of the split kernel: x 2^e = h1 + h2 with fp16 parts carries 22 bits, and a product needs THREE partial
// products (h1 g1, h1 g2, h2 g1) instead of six -- half the MFMA work and two thirds of the LDS traffic.  fp16 has five
// exponent bits, so every operand is pre-scaled by a power of two (exact) taken from its measured magnitude:
//   absmax_kernel      max |q|, |k|, |v| of the call -> three words in the workspace (one 39 MB sweep, ~12 us), unless
//                      the caller hands over bounds it can prove (amav_selfattn_forward_bounded)
//   K 2^ek, V 2^ev     split by split_kv_f16_kernel; Q 2^eq (with the softmax scale folded in) split in registers
//   S = acc 2^-(eq+ek) applied inside the exponential's fused multiply-add, no extra instruction
//   P 2^14             the probabilities (<= 1) scaled to fp16's upper range before their split; the same factor is in
//                      the running row sum, so it cancels in O / l, and 2^-ev is folded into the final normalisation.
// e is chosen so the largest magnitude lands in [2^14, 2^15): no overflow, and elements down to 2^-17 of the maximum keep
// their residual out of fp16's subnormals (below that the absolute error is under 2^-25 of the maximum).
struct Magnitudes {  // upper bounds of |q|, |k|, |v| handed over by the caller (instead of the measured maxima)
    float q, k, v;
};

__device__ __forceinline__ int fp16_scale_exp(float amax) {
    if (!(amax > 0.f)) return 0;
    return max(-100, min(100, 14 - ilogbf(amax)));
}

__global__ __launch_bounds__(256) void absmax_kernel(const float *__restrict__ q, const float *__restrict__ k,
                                                     const float *__restrict__ v, long long rows, int row4,
                                                     long long row_stride, unsigned *__restrict__ hdr) {
    float mq = 0.f, mk = 0.f, mv = 0.f;
    const long long total = rows * row4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / row4;
        const size_t off = row * row_stride + (i - row * row4) * 4;
        const float4 a = *reinterpret_cast<const float4 *>(q + off), b = *reinterpret_cast<const float4 *>(k + off);
        const float4 c = *reinterpret_cast<const float4 *>(v + off);
        mq = fmaxf(mq, fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))));
        mk = fmaxf(mk, fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w))));
        mv = fmaxf(mv, fmaxf(fmaxf(fabsf(c.x), fabsf(c.y)), fmaxf(fabsf(c.z), fabsf(c.w))));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mq = fmaxf(mq, __shfl_xor(mq, o, 64)), mk = fmaxf(mk, __shfl_xor(mk, o, 64)), mv = fmaxf(mv, __shfl_xor(mv, o, 64));
    }
    __shared__ float wave_max[4][3];
    if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6][0] = mq, wave_max[threadIdx.x >> 6][1] = mk, wave_max[threadIdx.x >> 6][2] = mv;
    __syncthreads();
    if (threadIdx.x < 3) {  // one atomic per block and tensor (same-address atomics serialise in L2: ~10 ns each);
        const float m = fmaxf(fmaxf(wave_max[0][threadIdx.x], wave_max[1][threadIdx.x]),
                              fmaxf(wave_max[2][threadIdx.x], wave_max[3][threadIdx.x]));
        atomicMax(hdr + threadIdx.x, __float_as_uint(m));  // non-negative floats order like their bit patterns
    }
}

// grid (key tiles of 64, H, B), 256 threads: as split_kv_kernel, two fp16 parts of K 2^ek and V 2^ev.  V^T rows store
// every group of 16 keys in the order the accumulator-as-operand product reads them: positions 0-7 = keys 0-3, 8-11
// (lane half 0), positions 8-15 = keys 4-7, 12-15 (lane half 1).
__global__ __launch_bounds__(256) void split_kv_f16_kernel(const float *__restrict__ k, const float *__restrict__ v, int S,
                                                           int Spad, long long row_stride,
                                                           const unsigned *__restrict__ hdr, Magnitudes given,
                                                           _Float16 *__restrict__ Kp, _Float16 *__restrict__ Vt) {
    __shared__ float vt[kD][kBN + 1];
    const int tid = threadIdx.x, head = blockIdx.y, b = blockIdx.z, H = gridDim.y, B = gridDim.z;
    const int key0 = blockIdx.x * kBN;
    const size_t bh = (size_t)b * H + head, part_k = (size_t)B * H * S * kD, part_v = (size_t)B * H * kD * Spad;
    const float sk = ldexpf(1.0f, fp16_scale_exp(hdr ? __uint_as_float(hdr[1]) : given.k));
    const float sv = ldexpf(1.0f, fp16_scale_exp(hdr ? __uint_as_float(hdr[2]) : given.v));
    const int key = tid >> 2, d0 = (tid & 3) * 16;
    const bool live = key0 + key < S;
    const size_t src = ((size_t)b * S + min(key0 + key, S - 1)) * row_stride + head * kD + d0;
    float kk[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 kv = *reinterpret_cast<const float4 *>(k + src + 4 * q);
        const float4 vv = *reinterpret_cast<const float4 *>(v + src + 4 * q);
        kk[4 * q] = kv.x, kk[4 * q + 1] = kv.y, kk[4 * q + 2] = kv.z, kk[4 * q + 3] = kv.w;
        vt[d0 + 4 * q + 0][key] = live ? vv.x * sv : 0.f;
        vt[d0 + 4 * q + 1][key] = live ? vv.y * sv : 0.f;
        vt[d0 + 4 * q + 2][key] = live ? vv.z * sv : 0.f;
        vt[d0 + 4 * q + 3][key] = live ? vv.w * sv : 0.f;
    }
    if (live) {  // 16 consecutive d of one key: two 16-byte stores per part
        _Float16 *dst = Kp + (bh * S + key0 + key) * kD + d0;
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            f16x8 p1, p2;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                _Float16 a, bb;
                split2(kk[8 * o + e] * sk, a, bb);
                p1[e] = a, p2[e] = bb;
            }
            *reinterpret_cast<f16x8 *>(dst + 8 * o) = p1;
            *reinterpret_cast<f16x8 *>(dst + part_k + 8 * o) = p2;
        }
    }
    __syncthreads();
    // thread -> d tid / 4, the 16-key group (tid % 4); inside a group the keys are stored in the order the PV product's
    // A operand wants them: lane half hh of selfattn_f16_kernel then reads its 8 keys as one 16-byte piece
    const int d = tid >> 2, kq = (tid & 3) * 16;
    _Float16 *dst = Vt + (bh * kD + d) * Spad + key0 + kq;
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        f16x8 p1, p2;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int pos = 8 * o + e;  // position in the group -> key 8 ((pos & 7) >> 2) + 4 (pos >> 3) + (pos & 3)
            _Float16 a, bb;
            split2(vt[d][kq + 8 * ((pos & 7) >> 2) + 4 * (pos >> 3) + (pos & 3)], a, bb);
            p1[e] = a, p2[e] = bb;
        }
        *reinterpret_cast<f16x8 *>(dst + 8 * o) = p1;
        *reinterpret_cast<f16x8 *>(dst + part_v + 8 * o) = p2;
    }
}

// Diagnostic build (-DAMAV_ATTN_STAMPS, tools/attention_stamps.py): per-phase shader-clock totals of every wave of
// selfattn_f16_kernel -- stage + barrier | QK^T | max + correction + first split | PV with the pipelined splits |
// second barrier -- summed into amav_attn_stamp_totals.  The stamps serialise the phases (the stamped kernel is ~1.6x
// slower): read the split, not the total.
#ifdef AMAV_ATTN_STAMPS
__device__ unsigned long long amav_attn_stamp_totals[8];
#define AMAV_STAMP_DECL long long stamp_prev_ = 0, stamp_acc_[6] = {0, 0, 0, 0, 0, 0};
#define AMAV_STAMP(i_)                                                   \
    {                                                                    \
        const long long now_ = clock64();                                \
        if ((i_) != 0) stamp_acc_[i_] += now_ - stamp_prev_;             \
        stamp_prev_ = now_;                                              \
    }
#define AMAV_STAMP_FLUSH                                                                             \
    if (lane == 0) {                                                                                 \
        for (int i_ = 1; i_ < 6; ++i_) atomicAdd(&amav_attn_stamp_totals[i_], (unsigned long long)stamp_acc_[i_]); \
        atomicAdd(&amav_attn_stamp_totals[0], 1ull);                                                 \
    }
#else
#define AMAV_STAMP_DECL
#define AMAV_STAMP(i_)
#define AMAV_STAMP_FLUSH
#endif

__global__ __launch_bounds__(256, 3) void selfattn_f16_kernel(const float *__restrict__ q, const _Float16 *__restrict__ Kp,
                                                           const _Float16 *__restrict__ Vt, float *__restrict__ out,
                                                           int S, int Spad, long long row_stride,
                                                           long long out_row_stride, float scale_log2e, int nsplit,
                                                           float *__restrict__ part, const unsigned *__restrict__ hdr,
                                                           Magnitudes given, int H, int B, int q_tiles) {
    __shared__ _Float16 Ks[2][kBN * kLdK];  // [part][key][d]
    __shared__ _Float16 Vs[2][kD * kLdK];   // [part][d][key, in the group order of split_kv_f16_kernel]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, hh = lane >> 5;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int slice = (j / q_tiles) * 8 + xcd;  // (batch, head, key-slice) index, XCD-aware as in selfattn_split_kernel
    if (slice >= nsplit * H * B) return;
    const int split = slice % nsplit, head = (slice / nsplit) % H, b = slice / (nsplit * H);
    const int q0 = (j % q_tiles) * kBM + wave * 32;
    const size_t bh = (size_t)b * H + head, part_k = (size_t)B * H * S * kD, part_v = (size_t)B * H * kD * Spad;
    const int eq = fp16_scale_exp((hdr ? __uint_as_float(hdr[0]) : given.q) * scale_log2e);
    const int ek = fp16_scale_exp(hdr ? __uint_as_float(hdr[1]) : given.k);
    const int ev = fp16_scale_exp(hdr ? __uint_as_float(hdr[2]) : given.v);
    const float qs = scale_log2e * ldexpf(1.0f, eq);  // Q pre-scale (softmax scale and log2 e folded in)
    const float cs = ldexpf(1.0f, -(eq + ek));        // raw accumulator -> log2-domain score

    f16x8 Q1[4], Q2[4];
    {
        const float *qrow = q + ((size_t)b * S + min(q0 + r, S - 1)) * row_stride + head * kD;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float4 lo = *reinterpret_cast<const float4 *>(qrow + 16 * s + 8 * hh);
            const float4 hi = *reinterpret_cast<const float4 *>(qrow + 16 * s + 8 * hh + 4);
            const float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                _Float16 a, bb;
                split2(x[jj] * qs, a, bb);
                Q1[s][jj] = a, Q2[s][jj] = bb;
            }
        }
    }
    f32x16 O0, O1;
#pragma unroll
    for (int t = 0; t < 16; ++t) O0[t] = 0.f, O1[t] = 0.f;
    float m_run = -1e30f, l_run = 0.f;

    const int ntiles_all = (S + kBN - 1) / kBN;
    const int kt_begin = (int)((long long)ntiles_all * split / nsplit);
    const int ntiles = (int)((long long)ntiles_all * (split + 1) / nsplit);
    const int srow0 = tid >> 3, sc8 = (tid & 7) * 8;  // 16-byte chunk tid + 256 i of a [64 rows][128 B] tile
    uint4 ka0, ka1, kb0, kb1, va0, va1, vb0, vb1;     // the next tile's eight chunks, in flight under this tile's MFMAs
#define AMAV_F16_LOAD(kt_)                                                                                         \
    {                                                                                                              \
        const int key0_ = (kt_) * kBN;                                                                             \
        const size_t k0_ = (bh * S + min(key0_ + srow0, S - 1)) * kD + sc8;                                        \
        const size_t k1_ = (bh * S + min(key0_ + srow0 + 32, S - 1)) * kD + sc8;                                   \
        const size_t v0_ = (bh * kD + srow0) * Spad + key0_ + sc8, v1_ = v0_ + (size_t)32 * Spad;                  \
        ka0 = *reinterpret_cast<const uint4 *>(Kp + k0_), ka1 = *reinterpret_cast<const uint4 *>(Kp + k1_);        \
        kb0 = *reinterpret_cast<const uint4 *>(Kp + part_k + k0_), kb1 = *reinterpret_cast<const uint4 *>(Kp + part_k + k1_); \
        va0 = *reinterpret_cast<const uint4 *>(Vt + v0_), va1 = *reinterpret_cast<const uint4 *>(Vt + v1_);        \
        vb0 = *reinterpret_cast<const uint4 *>(Vt + part_v + v0_), vb1 = *reinterpret_cast<const uint4 *>(Vt + part_v + v1_); \
    }
#define AMAV_F16_STAGE(p_, i_, kr_, vr_)                                                             \
    {                                                                                                \
        *reinterpret_cast<uint4 *>(&Ks[p_][(srow0 + 32 * (i_)) * kLdK + sc8]) = kr_;                 \
        *reinterpret_cast<uint4 *>(&Vs[p_][(srow0 + 32 * (i_)) * kLdK + sc8]) = vr_;                 \
    }
    AMAV_F16_LOAD(kt_begin)
    AMAV_STAMP_DECL
    for (int kt = kt_begin; kt < ntiles; ++kt) {
        const int key0 = kt * kBN;
        AMAV_STAMP(0)
        AMAV_F16_STAGE(0, 0, ka0, va0) AMAV_F16_STAGE(0, 1, ka1, va1) AMAV_F16_STAGE(1, 0, kb0, vb0)
        AMAV_F16_STAGE(1, 1, kb1, vb1)
        __syncthreads();
        AMAV_STAMP(1)
        if (kt + 1 < ntiles) AMAV_F16_LOAD(kt + 1)

        // ---- raw S^T = K' Q'^T: three partial products per 16-wide k-step, small terms first
        f32x16 S0, S1;
#pragma unroll
        for (int t = 0; t < 16; ++t) S0[t] = 0.f, S1[t] = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16 &Sx = kb ? S1 : S0;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int off = (r + 32 * kb) * kLdK + 16 * s + 8 * hh;
                const f16x8 a1 = *reinterpret_cast<const f16x8 *>(&Ks[0][off]);
                const f16x8 a2 = *reinterpret_cast<const f16x8 *>(&Ks[1][off]);
                Sx = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, Q1[s], Sx, 0, 0, 0);
                Sx = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, Q2[s], Sx, 0, 0, 0);
                Sx = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, Q1[s], Sx, 0, 0, 0);
            }
        }
        if ((kt + 1) * kBN > S) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int kk = key0 + (t & 3) + 8 * (t >> 2) + 4 * hh;
                if (kk >= S) S0[t] = -INFINITY;
                if (kk + 32 >= S) S1[t] = -INFINITY;
            }
        }
        // ---- online softmax over this lane's query: the maximum of the raw scores, moved to the log2 domain once
        AMAV_STAMP(2)
        float mx = S0[0];
#pragma unroll
        for (int t = 1; t < 16; ++t) mx = fmaxf(mx, S0[t]);
#pragma unroll
        for (int t = 0; t < 16; ++t) mx = fmaxf(mx, S1[t]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx * cs);
        const float corr = __builtin_amdgcn_exp2f(m_run - m_new);
        if (__any(m_new != m_run)) {  // wave-uniform: after the first tiles the running maxima rarely move
#pragma unroll
            for (int t = 0; t < 16; ++t) O0[t] *= corr, O1[t] *= corr;
        }
        m_run = m_new;
        const float shift = 14.0f - m_new;  // P' = 2^14 exp2(s - m)
        float psum = 0.f;
        // ---- O'^T += V'^T P'^T, software-pipelined: the exponentials and the split of step u + 1 between the six MFMAs
        // of step u
        auto parts = [&](int u, f16x8 &P1, f16x8 &P2) {
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const float raw = (u >> 1) ? S1[8 * (u & 1) + jj] : S0[8 * (u & 1) + jj];
                const float pv = __builtin_amdgcn_exp2f(fmaf(raw, cs, shift));
                psum += pv;
                _Float16 a, bb;
                split2(pv, a, bb);
                P1[jj] = a, P2[jj] = bb;
            }
        };
        f16x8 Pc1, Pc2, Pn1, Pn2;
        parts(0, Pc1, Pc2);
        AMAV_STAMP(3)
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // u = 2 kb + s2
            f16x8 vf[2][2];  // this lane half's 8 keys of k-step u: one 16-byte piece (group order)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int p = 0; p < 2; ++p)
                    vf[a][p] = *reinterpret_cast<const f16x8 *>(&Vs[p][(r + 32 * a) * kLdK + 16 * u + 8 * hh]);
            if (u + 1 < 4) parts(u + 1, Pn1, Pn2);
            O0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[0][1], Pc1, O0, 0, 0, 0);
            O1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[1][1], Pc1, O1, 0, 0, 0);
            O0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[0][0], Pc2, O0, 0, 0, 0);
            O1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[1][0], Pc2, O1, 0, 0, 0);
            O0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[0][0], Pc1, O0, 0, 0, 0);
            O1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[1][0], Pc1, O1, 0, 0, 0);
            if (u + 1 < 4) {
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);  // ten vector instructions of the next step
                }
                Pc1 = Pn1, Pc2 = Pn2;
            }
        }
        l_run = l_run * corr + psum;
        AMAV_STAMP(4)
        __syncthreads();
        AMAV_STAMP(5)
    }
    AMAV_STAMP_FLUSH
#undef AMAV_F16_LOAD
#undef AMAV_F16_STAGE

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);  // carries the 2^14 of P'
    const float unv = ldexpf(1.0f, -ev);
    if (nsplit > 1) {  // partial (O 2^14, m, l 2^14): the common factor cancels in combine_kernel
        if (q0 + r < S) {
            float *prow = part + ((((size_t)split * B + b) * H + head) * S + q0 + r) * (kD + 2);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = 8 * g + 4 * hh;
#pragma unroll
                for (int e = 0; e < 4; ++e) prow[d + e] = O0[4 * g + e] * unv, prow[32 + d + e] = O1[4 * g + e] * unv;
            }
            if (hh == 0) prow[kD] = m_run, prow[kD + 1] = l_tot;
        }
        return;
    }
    const float inv = unv / l_tot;
    if (q0 + r < S) {
        float *orow = out + ((size_t)b * S + q0 + r) * out_row_stride + head * kD;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = 8 * g + 4 * hh;
            *reinterpret_cast<float4 *>(orow + d) =
                make_float4(O0[4 * g] * inv, O0[4 * g + 1] * inv, O0[4 * g + 2] * inv, O0[4 * g + 3] * inv);
            *reinterpret_cast<float4 *>(orow + 32 + d) =
                make_float4(O1[4 * g] * inv, O1[4 * g + 1] * inv, O1[4 * g + 2] * inv, O1[4 * g + 3] * inv);
        }
    }
}

// Key-range split that best balances the (q-tile, head, batch) workgroups over the chip: a CU runs two workgroups
// at a time (LDS / registers), so the kernel lasts ceil(blocks * s / CUs) slices of 1/s of the key sweep.
static int choose_split(int B, int S, int H, int num_cus) {
    static const int forced = getenv("AMAV_ATTN_SPLIT") ? atoi(getenv("AMAV_ATTN_SPLIT")) : 0;  // tuning aid
    if (forced >= 1 && forced <= 16 && ((S + kBN - 1) / kBN) / forced >= 1) return forced;
    const long long blocks = (long long)((S + kBM - 1) / kBM) * H * B;
    const int ntiles = (S + kBN - 1) / kBN;
    int best = 1;
    double best_cost = (double)((blocks + num_cus - 1) / num_cus);
    for (int s = 2; s <= 8; ++s) {
        if (ntiles / s < 8) break;  // keep slices long enough to amortise the Q load and the combine pass
        const double cost = (double)((blocks * s + num_cus - 1) / num_cus) / s + 0.02 * s;
        if (cost < best_cost - 1e-9) best_cost = cost, best = s;
    }
    return best;
}

}  // namespace attn
}  // namespace amav

using namespace amav;

static int attn_num_cus() {
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
            v = 256;
        return v;
    }();
    return n;
}

// AMAV_ATTN=f32 selects the fp32-MFMA kernel, AMAV_ATTN=bf16 the bf16 x 3 split kernel, anything else (default) the
// fp16 x 2 split kernel
static int attn_variant() {  // 0: fp32 MFMA, 1: bf16 x 3, 2: fp16 x 2
    const int chosen = option_attn();  // amav_set_option("attn", ...)
    if (chosen >= 0) return chosen;
    static const int v = [] {
        const char *e = getenv("AMAV_ATTN");
        if (e && strcmp(e, "f32") == 0) return 0;
        return e && strcmp(e, "bf16") == 0 ? 1 : 2;
    }();
    return v;
}
static bool attn_use_split() { return attn_variant() != 0; }
constexpr size_t kAttnHeaderBytes = 256;  // fp16 variant: max |q|, |k|, |v| of the call

static size_t attn_partial_bytes(int B, int S, int H, int ns) {
    return ns > 1 ? align_up((size_t)ns * B * H * S * (attn::kD + 2) * sizeof(float), 256) : 256;
}
static size_t attn_spad(int S) { return ((size_t)S + attn::kBN - 1) / attn::kBN * attn::kBN; }

extern "C" size_t amav_selfattn_workspace_bytes(int B, int S, int H, int D) {
    if (B <= 0 || S <= 0 || H <= 0 || D != attn::kD) return 0;
    const int ns = attn::choose_split(B, S, H, attn_num_cus());
    size_t need = attn_partial_bytes(B, S, H, ns);
    const size_t parts = attn_variant() == 2 ? 2 : 3;
    if (attn_use_split())  // [header] + K parts [parts][B H][S][64] + V^T parts [parts][B H][64][S_pad], 2-byte elements
        need += kAttnHeaderBytes + align_up(parts * B * H * S * attn::kD * 2, 256) +
                align_up(parts * B * H * attn::kD * attn_spad(S) * 2, 256);
    return need;
}

extern "C" int amav_selfattn_forward(int B, int S, int H, int D, const float *q, const float *k, const float *v,
                                     int64_t row_stride, float *out, int64_t out_row_stride, float scale,
                                     void *workspace, size_t workspace_bytes, void *stream_) {
    return amav_selfattn_forward_bounded(B, S, H, D, q, k, v, row_stride, out, out_row_stride, scale, 0.f, 0.f, 0.f,
                                         workspace, workspace_bytes, stream_);
}

extern "C" int amav_selfattn_forward_bounded(int B, int S, int H, int D, const float *q, const float *k, const float *v,
                                             int64_t row_stride, float *out, int64_t out_row_stride, float scale,
                                             float q_bound, float k_bound, float v_bound, void *workspace,
                                             size_t workspace_bytes, void *stream_) {
    return amav_selfattn_forward_split_out(B, S, H, D, q, k, v, row_stride, out, out_row_stride, scale, q_bound, k_bound, v_bound,
                                           nullptr, 0, workspace, workspace_bytes, stream_);
}

extern "C" int amav_selfattn_forward_split_out(int B, int S, int H, int D, const float *q, const float *k, const float *v,
                                               int64_t row_stride, float *out, int64_t out_row_stride, float scale,
                                               float q_bound, float k_bound, float v_bound, void *out_split,
                                               int split_scale_exp, void *workspace, size_t workspace_bytes, void *stream_) {
    AMAV_REQUIRE(out_split == nullptr || ((reinterpret_cast<uintptr_t>(out_split) & 15) == 0 && split_scale_exp >= -126 &&
                                          split_scale_exp <= 126 && out_row_stride == (int64_t)H * D),
                 "amav_selfattn_forward_split_out: out_split must be 16-byte aligned, |scale_exp| <= 126, out rows dense");
    const bool bounded = q_bound > 0.f && k_bound > 0.f && v_bound > 0.f;
    AMAV_REQUIRE(bounded || (q_bound == 0.f && k_bound == 0.f && v_bound == 0.f),
                 "amav_selfattn_forward_bounded: give all three bounds (> 0, finite) or none (0)");
    AMAV_REQUIRE(std::isfinite(q_bound) && std::isfinite(k_bound) && std::isfinite(v_bound),
                 "amav_selfattn_forward_bounded: bounds must be finite");
    AMAV_REQUIRE(B > 0 && S > 0 && H > 0, "amav_selfattn_forward: bad sizes B=%d S=%d H=%d", B, S, H);
    AMAV_REQUIRE(D == attn::kD, "amav_selfattn_forward: head_dim %d (only %d is built)", D, attn::kD);
    AMAV_REQUIRE(q && k && v && out, "amav_selfattn_forward: NULL pointer");
    AMAV_REQUIRE(row_stride >= (int64_t)H * D && out_row_stride >= (int64_t)H * D && row_stride % 4 == 0 &&
                     out_row_stride % 4 == 0,
                 "amav_selfattn_forward: row strides must be multiples of 4 floats and >= H*D");
    AMAV_REQUIRE(((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v) |
                   reinterpret_cast<uintptr_t>(out)) & 15) == 0,
                 "amav_selfattn_forward: q/k/v/out must be 16-byte aligned");
    AMAV_REQUIRE(H <= 65535 && B <= 65535, "amav_selfattn_forward: grid too large");
    const int ns = attn::choose_split(B, S, H, attn_num_cus());
    const bool split = attn_use_split();
    const size_t need = split || ns > 1 ? amav_selfattn_workspace_bytes(B, S, H, D) : 0;
    if (need && (workspace == nullptr || workspace_bytes < need))
        return fail(AMAV_ERR_WORKSPACE, "amav_selfattn_forward: workspace %zu < required %zu", workspace_bytes, need);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const dim3 grid((unsigned)((S + attn::kBM - 1) / attn::kBM) * ns, H, B);
    if (split) {
        char *ws = static_cast<char *>(workspace);
        const int Spad = (int)attn_spad(S);
        const size_t parts = attn_variant() == 2 ? 2 : 3;
        unsigned *hdr = reinterpret_cast<unsigned *>(ws + attn_partial_bytes(B, S, H, ns));
        char *kp = reinterpret_cast<char *>(hdr) + kAttnHeaderBytes;
        char *vt = kp + align_up(parts * B * H * S * attn::kD * 2, 256);
        const dim3 kv_grid((unsigned)(Spad / attn::kBN), H, B);
        const int q_tiles = (S + attn::kBM - 1) / attn::kBM;
        const long long rounds = ((long long)ns * H * B + 7) / 8;  // 8 slices (one per XCD) per round
        const unsigned main_grid = (unsigned)(rounds * 8 * q_tiles);
        const float sl2 = scale * 1.4426950408889634f;
        if (attn_variant() == 2) {
            const attn::Magnitudes given = {q_bound, k_bound, v_bound};
            if (bounded) {
                hdr = nullptr;  // the kernels scale from `given`
            } else {            // measure max |q|, |k|, |v|
                AMAV_REQUIRE(zero_async(hdr, 16, stream) == hipSuccess, "amav_selfattn_forward: header clear failed");
                const long long rows = (long long)B * S;
                const long long quads = rows * (H * attn::kD / 4);
                attn::absmax_kernel<<<(unsigned)std::min<long long>((quads + 255) / 256, 512), 256, 0, stream>>>(
                    q, k, v, rows, H * attn::kD / 4, row_stride, hdr);
            }
            attn::split_kv_f16_kernel<<<kv_grid, 256, 0, stream>>>(k, v, S, Spad, row_stride, hdr, given,
                                                                  reinterpret_cast<_Float16 *>(kp),
                                                                  reinterpret_cast<_Float16 *>(vt));
            attn::selfattn_f16_kernel<<<main_grid, 256, 0, stream>>>(
                q, reinterpret_cast<const _Float16 *>(kp), reinterpret_cast<const _Float16 *>(vt), out, S, Spad,
                row_stride, out_row_stride, sl2, ns, static_cast<float *>(workspace), hdr, given, H, B, q_tiles);
        } else {
            attn::split_kv_kernel<<<kv_grid, 256, 0, stream>>>(k, v, S, Spad, row_stride, reinterpret_cast<__bf16 *>(kp),
                                                              reinterpret_cast<__bf16 *>(vt));
            attn::selfattn_split_kernel<<<main_grid, 256, 0, stream>>>(
                q, reinterpret_cast<const __bf16 *>(kp), reinterpret_cast<const __bf16 *>(vt), out, S, Spad, row_stride,
                out_row_stride, sl2, ns, static_cast<float *>(workspace), H, B, q_tiles);
        }
    } else
        attn::selfattn_kernel<<<grid, 256, 0, stream>>>(q, k, v, out, S, row_stride, out_row_stride,
                                                        scale * 1.4426950408889634f, ns, static_cast<float *>(workspace));
    if (ns > 1) {
        const long long threads = (long long)B * H * S * 16;
        if (out_split)  // the partial states are merged straight into the next projection's fp16 x 2 operand
            attn::combine_kernel<true><<<(unsigned)((threads + 255) / 256), 256, 0, stream>>>(
                static_cast<const float *>(workspace), out, B, H, S, ns, out_row_stride, static_cast<_Float16 *>(out_split),
                std::ldexp(1.0f, split_scale_exp));
        else
            attn::combine_kernel<false><<<(unsigned)((threads + 255) / 256), 256, 0, stream>>>(
                static_cast<const float *>(workspace), out, B, H, S, ns, out_row_stride, nullptr, 1.0f);
    } else if (out_split) {  // one key slice: the kernel wrote `out`; split it as amav_split_operand would
        if (int rc = check_launch("amav_selfattn_forward")) return rc;
        return amav_split_operand((int64_t)B * S, H * D, out, out_row_stride, 0, AMAV_SPLIT_FP16X2, split_scale_exp, out_split, stream_);
    }
    return check_launch("amav_selfattn_forward");
}


// ---------------------------------------------------------------------------------------------------------------
// Split operands of an fp32-equivalent GEMM on the bf16 matrix pipe (the same three-way split as the attention
// kernel above): x = x1 + x2 + x3 with every part a bf16, and x W^T is the sum of the six partial products x_i W_j^T
// with i + j <= 4 (the dropped ones are below 2^-24 of the result).  The six products are ONE bf16 GEMM with fp32
// accumulation over operands concatenated along K, small terms first:
//     activations  A' = [x3 | x2 | x1 | x2 | x1 | x1]      weights  B' = [w1 | w2 | w3 | w1 | w2 | w1]      (K' = 6 K)
// which the matrix pipe runs at 16x the fp32 MFMA rate, i.e. 2.7x faster than the fp32 GEMM for the same result
// (measured 1.9e-6 max abs against fp64 on the 512 -> 4096 projection, 6.0e-6 for the library's fp32 GEMM).
namespace amav {
namespace attn {
template <bool kWeights>
__global__ __launch_bounds__(256) void split_operand_kernel(long long octs, int k8, const float *__restrict__ x,
                                                            long long row_stride, __bf16 *__restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= octs) return;
    const long long row = i / k8;
    const int c = (int)(i - row * k8) * 8;
    const float *src = x + row * row_stride + c;
    const float4 lo = *reinterpret_cast<const float4 *>(src), hi = *reinterpret_cast<const float4 *>(src + 4);
    const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    bf16x8 p1, p2, p3;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        __bf16 a, b, cc;
        split3(v[j], a, b, cc);
        p1[j] = a, p2[j] = b, p3[j] = cc;
    }
    const long long K = (long long)k8 * 8;
    __bf16 *dst = out + row * 6 * K + c;
    auto put = [&](int block, const bf16x8 &p) { *reinterpret_cast<bf16x8 *>(dst + block * K) = p; };
    if (kWeights) {
        put(0, p1), put(1, p2), put(2, p3), put(3, p1), put(4, p2), put(5, p1);
    } else {
        put(0, p3), put(1, p2), put(2, p1), put(3, p2), put(4, p1), put(5, p1);
    }
}

// fp16 x 2 format: (x * prescale) = h1 + h2; activations [h2 | h1 | h1], weights [g1 | g2 | g1] (K' = 3 K), so that
// A' B'^T = h2 g1 + h1 g2 + h1 g1 -- the product to 2^-22, at half the MFMA work of the bf16 x 3 format.
template <bool kWeights>
__global__ __launch_bounds__(256) void split_operand_f16_kernel(long long octs, int k8, const float *__restrict__ x,
                                                                long long row_stride, float prescale,
                                                                _Float16 *__restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= octs) return;
    const long long row = i / k8;
    const int c = (int)(i - row * k8) * 8;
    const float *src = x + row * row_stride + c;
    const float4 lo = *reinterpret_cast<const float4 *>(src), hi = *reinterpret_cast<const float4 *>(src + 4);
    const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    f16x8 p1, p2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        _Float16 a, b;
        split2(v[j] * prescale, a, b);
        p1[j] = a, p2[j] = b;
    }
    const long long K = (long long)k8 * 8;
    _Float16 *dst = out + row * 3 * K + c;
    auto put = [&](int block, const f16x8 &p) { *reinterpret_cast<f16x8 *>(dst + block * K) = p; };
    if (kWeights) {
        put(0, p1), put(1, p2), put(2, p1);
    } else {
        put(0, p2), put(1, p1), put(2, p1);
    }
}
}  // namespace attn
}  // namespace amav

static bool split_format_ok(int format, int scale_exp) {
    return format == AMAV_SPLIT_BF16X3 || (format == AMAV_SPLIT_FP16X2 && scale_exp >= -126 && scale_exp <= 126);
}

extern "C" int amav_split_operand(int64_t rows, int k, const float *x, int64_t x_row_stride, int weights, int format,
                                  int scale_exp, void *out, void *stream_) {
    AMAV_REQUIRE(rows > 0 && k > 0 && k % 8 == 0, "amav_split_operand: rows=%lld k=%d (k must be a multiple of 8)",
                 (long long)rows, k);
    AMAV_REQUIRE(x && out, "amav_split_operand: NULL pointer");
    AMAV_REQUIRE(x_row_stride >= k && x_row_stride % 4 == 0, "amav_split_operand: bad row stride");
    AMAV_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0,
                 "amav_split_operand: buffers must be 16-byte aligned");
    AMAV_REQUIRE(split_format_ok(format, scale_exp), "amav_split_operand: format %d / scale exponent %d", format, scale_exp);
    const long long octs = rows * (k / 8);
    const unsigned grid = (unsigned)((octs + 255) / 256);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (format == AMAV_SPLIT_FP16X2) {
        const float prescale = ldexpf(1.0f, scale_exp);
        _Float16 *o = static_cast<_Float16 *>(out);
        if (weights)
            amav::attn::split_operand_f16_kernel<true><<<grid, 256, 0, stream>>>(octs, k / 8, x, x_row_stride, prescale, o);
        else
            amav::attn::split_operand_f16_kernel<false><<<grid, 256, 0, stream>>>(octs, k / 8, x, x_row_stride, prescale, o);
    } else if (weights) {
        amav::attn::split_operand_kernel<true><<<grid, 256, 0, stream>>>(octs, k / 8, x, x_row_stride,
                                                                       static_cast<__bf16 *>(out));
    } else {
        amav::attn::split_operand_kernel<false><<<grid, 256, 0, stream>>>(octs, k / 8, x, x_row_stride,
                                                                        static_cast<__bf16 *>(out));
    }
    return check_launch("amav_split_operand");
}


// ---------------------------------------------------------------------------------------------------------------
// GEGLU gate of the feed-forward (src/models/transformers.py:484-508: hidden, gate = proj(x).chunk(2); hidden *
// gelu(gate), exact-erf GELU): one pass over the projection's [rows, 2 * inner] output instead of torch's two
// elementwise kernels (gelu, then mul), i.e. 1.5 instead of 2.5 tensor sweeps.
namespace amav {
namespace attn {
__global__ __launch_bounds__(256) void geglu_kernel(long long quads, int inner4, const float4 *__restrict__ in,
                                                    long long in_row4, const float4 *__restrict__ bias,
                                                    float4 *__restrict__ out, _Float16 *__restrict__ out_split,
                                                    float prescale) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= quads) return;
    const long long row = i / inner4;
    const int col = (int)(i - row * inner4);
    float4 h = in[row * in_row4 + col], g = in[row * in_row4 + inner4 + col];
    if (bias) {  // the projection's bias, when its GEMM ran without one
        const float4 bh = bias[col], bg = bias[inner4 + col];
        h = make_float4(h.x + bh.x, h.y + bh.y, h.z + bh.z, h.w + bh.w);
        g = make_float4(g.x + bg.x, g.y + bg.y, g.z + bg.z, g.w + bg.w);
    }
    auto gelu = [](float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); };
    const float4 y = make_float4(h.x * gelu(g.x), h.y * gelu(g.y), h.z * gelu(g.z), h.w * gelu(g.w));
    if (out_split) {  // the fp16 x 2 activation operand of the output projection (split_operand_f16_kernel's layout)
        const float yv[4] = {y.x, y.y, y.z, y.w};
        f16x4 p1, p2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            _Float16 a, b;
            split2(yv[j] * prescale, a, b);
            p1[j] = a, p2[j] = b;
        }
        const long long K = 4LL * inner4;
        _Float16 *dst = out_split + row * 3 * K + 4 * col;
        *reinterpret_cast<f16x4 *>(dst) = p2;
        *reinterpret_cast<f16x4 *>(dst + K) = p1;
        *reinterpret_cast<f16x4 *>(dst + 2 * K) = p1;
    } else {
        out[i] = y;
    }
}
}  // namespace attn
}  // namespace amav

extern "C" int amav_geglu(int64_t rows, int inner, const float *proj, int64_t proj_row_stride, const float *bias,
                          float *out, void *out_split, int split_scale_exp, void *stream) {
    AMAV_REQUIRE(rows > 0 && inner > 0 && inner % 4 == 0, "amav_geglu: bad sizes rows=%lld inner=%d", (long long)rows, inner);
    AMAV_REQUIRE(proj && ((out != nullptr) != (out_split != nullptr)),
                 "amav_geglu: NULL projection, or not exactly one of out (fp32) and out_split (fp16 x 2 operand)");
    AMAV_REQUIRE(split_scale_exp >= -126 && split_scale_exp <= 126, "amav_geglu: scale exponent %d", split_scale_exp);
    AMAV_REQUIRE(proj_row_stride >= 2LL * inner && proj_row_stride % 4 == 0, "amav_geglu: bad row stride");
    AMAV_REQUIRE(((reinterpret_cast<uintptr_t>(proj) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(bias) |
                   reinterpret_cast<uintptr_t>(out_split)) & 15) == 0,
                 "amav_geglu: buffers must be 16-byte aligned");
    const long long quads = rows * (inner / 4);
    amav::attn::geglu_kernel<<<(unsigned)((quads + 255) / 256), 256, 0, static_cast<hipStream_t>(stream)>>>(
        quads, inner / 4, reinterpret_cast<const float4 *>(proj), proj_row_stride / 4,
        reinterpret_cast<const float4 *>(bias), reinterpret_cast<float4 *>(out), static_cast<_Float16 *>(out_split),
        ldexpf(1.0f, split_scale_exp));
    return check_launch("amav_geglu");
}


// ---------------------------------------------------------------------------------------------------------------
// Residual adds + LayerNorm of BasicTransformerBlock (src/models/transformers.py:292-399) in one pass:
//   h = ((a + a_bias) + h);  h = (row_b + h);  n = LayerNorm(h) * w + b
// i.e. `attn1(...) + h`, then `attn2(...) + h` whose value is ONE row per batch item (single audio key), then norm3.
// torch runs two adds and two LayerNorms for this (norm2's result is never used on this path: the cross-attention
// output does not depend on its queries).  One wave per row, row in registers, two-pass mean / variance.
// a_bias is the bias of the projection that produced `a` when its GEMM ran without one; with kSplit the normalised row
// is written as the activation operand of the next projection's split GEMM (split_operand_kernel's layout) instead of
// fp32 -- the row never makes the round trip through HBM in between.
namespace amav {
namespace attn {
template <int kVec, int kSplit>  // float4 per lane: dim = 256 * kVec; kSplit 0: fp32 rows, 1: bf16 x 3, 2: fp16 x 2
__global__ __launch_bounds__(256) void add_layernorm_kernel(long long rows, long long rows_per_batch,
                                                            const float4 *__restrict__ a, const float4 *__restrict__ a_bias,
                                                            const float4 *__restrict__ brow,
                                                            const float4 *__restrict__ h, float4 *__restrict__ h_out,
                                                            const float4 *__restrict__ w,
                                                            const float4 *__restrict__ b, float eps,
                                                            float4 *__restrict__ out, void *__restrict__ out_split,
                                                            float prescale) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    constexpr int kRow4 = 64 * kVec;
    const float4 *br = brow ? brow + (row / rows_per_batch) * kRow4 : nullptr;
    float4 x[kVec];
    float sum = 0.f;
#pragma unroll
    for (int v = 0; v < kVec; ++v) {
        const int c = lane + 64 * v;
        float4 t = h[row * kRow4 + c];
        if (a) {
            float4 av = a[row * kRow4 + c];
            if (a_bias) {
                const float4 ab = a_bias[c];
                av = make_float4(av.x + ab.x, av.y + ab.y, av.z + ab.z, av.w + ab.w);
            }
            t = make_float4(av.x + t.x, av.y + t.y, av.z + t.z, av.w + t.w);
        }
        if (br) {
            const float4 rv = br[c];
            t = make_float4(rv.x + t.x, rv.y + t.y, rv.z + t.z, rv.w + t.w);
        }
        x[v] = t;
        h_out[row * kRow4 + c] = t;
        sum += (t.x + t.y) + (t.z + t.w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float mean = sum * (1.0f / (256.0f * kVec));
    float var = 0.f;
#pragma unroll
    for (int v = 0; v < kVec; ++v) {
        const float dx = x[v].x - mean, dy = x[v].y - mean, dz = x[v].z - mean, dw = x[v].w - mean;
        var += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) var += __shfl_xor(var, o, 64);
    const float rstd = 1.0f / sqrtf(var * (1.0f / (256.0f * kVec)) + eps);
#pragma unroll
    for (int v = 0; v < kVec; ++v) {
        const int c = lane + 64 * v;
        const float4 wv = w[c], bv = b[c];
        const float4 n = make_float4((x[v].x - mean) * rstd * wv.x + bv.x, (x[v].y - mean) * rstd * wv.y + bv.y,
                                     (x[v].z - mean) * rstd * wv.z + bv.z, (x[v].w - mean) * rstd * wv.w + bv.w);
        if (kSplit == 2) {
            constexpr int K = 256 * kVec;
            const float nv[4] = {n.x, n.y, n.z, n.w};
            f16x4 p1, p2;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                _Float16 s1, s2;
                split2(nv[j] * prescale, s1, s2);
                p1[j] = s1, p2[j] = s2;
            }
            _Float16 *dst = static_cast<_Float16 *>(out_split) + row * 3 * K + 4 * c;
            *reinterpret_cast<f16x4 *>(dst) = p2;
            *reinterpret_cast<f16x4 *>(dst + K) = p1;
            *reinterpret_cast<f16x4 *>(dst + 2 * K) = p1;
        } else if (kSplit == 1) {
            constexpr int K = 256 * kVec;
            const float nv[4] = {n.x, n.y, n.z, n.w};
            bf16x4 p1, p2, p3;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                __bf16 s1, s2, s3;
                split3(nv[j], s1, s2, s3);
                p1[j] = s1, p2[j] = s2, p3[j] = s3;
            }
            __bf16 *dst = static_cast<__bf16 *>(out_split) + row * 6 * K + 4 * c;
            auto put = [&](int block, const bf16x4 &p) { *reinterpret_cast<bf16x4 *>(dst + block * K) = p; };
            put(0, p3), put(1, p2), put(2, p1), put(3, p2), put(4, p1), put(5, p1);
        } else {
            out[row * kRow4 + c] = n;
        }
    }
}
}  // namespace attn
}  // namespace amav

extern "C" int amav_add_layernorm(int64_t rows, int dim, int64_t rows_per_batch, const float *add, const float *add_bias,
                                  const float *batch_row, const float *hidden, float *hidden_out, const float *weight,
                                  const float *bias, float eps, float *out_norm, void *out_norm_split, int split_format,
                                  int split_scale_exp, void *stream_) {
    AMAV_REQUIRE(rows > 0 && rows_per_batch > 0 && (dim == 256 || dim == 512 || dim == 768 || dim == 1024),
                 "amav_add_layernorm: rows=%lld dim=%d (dim must be 256, 512, 768 or 1024)", (long long)rows, dim);
    AMAV_REQUIRE(hidden && hidden_out && weight && bias, "amav_add_layernorm: NULL pointer");
    AMAV_REQUIRE((out_norm != nullptr) != (out_norm_split != nullptr),
                 "amav_add_layernorm: give exactly one of out_norm (fp32) and out_norm_split (bf16 split operand)");
    AMAV_REQUIRE(add || !add_bias, "amav_add_layernorm: add_bias without add");
    AMAV_REQUIRE(!out_norm_split || split_format_ok(split_format, split_scale_exp),
                 "amav_add_layernorm: split format %d / scale exponent %d", split_format, split_scale_exp);
    AMAV_REQUIRE(((reinterpret_cast<uintptr_t>(hidden) | reinterpret_cast<uintptr_t>(hidden_out) |
                   reinterpret_cast<uintptr_t>(out_norm) | reinterpret_cast<uintptr_t>(out_norm_split) |
                   reinterpret_cast<uintptr_t>(add) | reinterpret_cast<uintptr_t>(add_bias) |
                   reinterpret_cast<uintptr_t>(batch_row) |
                   reinterpret_cast<uintptr_t>(weight) | reinterpret_cast<uintptr_t>(bias)) & 15) == 0,
                 "amav_add_layernorm: buffers must be 16-byte aligned");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const unsigned grid = (unsigned)((rows + 3) / 4);
    auto p4 = [](const float *p) { return reinterpret_cast<const float4 *>(p); };
    float4 *h4 = reinterpret_cast<float4 *>(hidden_out), *o4 = reinterpret_cast<float4 *>(out_norm);
    const float prescale = ldexpf(1.0f, out_norm_split && split_format == AMAV_SPLIT_FP16X2 ? split_scale_exp : 0);
#define AMAV_LN_ARGS rows, rows_per_batch, p4(add), p4(add_bias), p4(batch_row), p4(hidden), h4, p4(weight), p4(bias), eps, \
                     o4, out_norm_split, prescale
#define AMAV_LN_LAUNCH(V)                                                                                   \
    do {                                                                                                    \
        if (!out_norm_split)                                                                                \
            amav::attn::add_layernorm_kernel<V, 0><<<grid, 256, 0, stream>>>(AMAV_LN_ARGS);                 \
        else if (split_format == AMAV_SPLIT_FP16X2)                                                         \
            amav::attn::add_layernorm_kernel<V, 2><<<grid, 256, 0, stream>>>(AMAV_LN_ARGS);                 \
        else                                                                                                \
            amav::attn::add_layernorm_kernel<V, 1><<<grid, 256, 0, stream>>>(AMAV_LN_ARGS);                 \
    } while (0)
    if (dim == 256) AMAV_LN_LAUNCH(1);
    else if (dim == 512) AMAV_LN_LAUNCH(2);
    else if (dim == 768) AMAV_LN_LAUNCH(3);
    else AMAV_LN_LAUNCH(4);
#undef AMAV_LN_LAUNCH
#undef AMAV_LN_ARGS
    return check_launch("amav_add_layernorm");
}

#ifdef AMAV_ATTN_STAMPS
// diagnostic builds only (not declared in include/amav.h): reads and clears the phase totals
extern "C" int amav_debug_attn_stamps(unsigned long long out[8]) {
    unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(amav::attn::amav_attn_stamp_totals), sizeof(zero)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(amav::attn::amav_attn_stamp_totals), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
#endif
