// SMPL-X forward + linear blend skinning for gfx950, F frames per launch.
//
// Replaces smplx.SMPLX.forward -> smplx.lbs.lbs as the reference reaches it (src/models/renderer.py:261-274) and
// the posed-mesh subdivision + subset that follows (renderer.py:276-288).  SURVEY.md Appendix A.2 is the spec.
//
//   joint_chain_kernel  one 64-lane block per frame, one lane per joint: Rodrigues (angle = ||r + 1e-8||), joint
//                       locations from the pre-regressed tables, the kinematic chain level by level in LDS, the
//                       rest-pose-removed 3x4 transforms A_j, and the frame's blend feature column
//                       [betas; expression; (R_1..R_54 - I)] written TRANSPOSED ([k][frame]): a feature row is the
//                       A-operand row of the MFMA kernel and one LDS slice of the FMA kernel.
//   skin_mfma_kernel    (F > 16) the [F, 506] x [506, 3V] blend product on v_mfma_f32_32x32x2_f32: a block = 128
//                       frames x 32 vertices, the tile-major blend table streamed once through LDS, then
//                       T_v = sum_j w_vj A_j and the 3x4 transform in the accumulator layout.
//   skin_kernel         (F <= 16) one thread per vertex and FT frames: the same product as register-tiled FMAs with
//                       the group's feature slice in LDS; the frame groups of a vertex chunk share an XCD, so a
//                       chunk of the 63.6 MB table is fetched from HBM once.  Nothing but the vertices is written.
//   skin_f16_kernel     the MFMA kernel on the 16-bit matrix pipe: features and table as two fp16 parts each, three
//                       partial products per fp32 product on v_mfma_f32_32x32x16_f16 (16x the fp32 MFMA rate), fp32
//                       accumulation -- the fp32 product to 2^-22.  Needs the table pre-split
//                       (amav_lbs_prepare_blend_split); every frame's features are pre-scaled by their own power of
//                       two, taken back out in the epilogue.
//   gather_kernel       baked subdivision table -> the N sampled points.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "amav_common.h"

#ifndef AMAV_LBS_ABLATE
#define AMAV_LBS_ABLATE 0  /* diagnostic builds only (tools/ablate_lbs.sh) */
#endif

namespace amav {
namespace lbs {

constexpr int kMaxJoints = 64;

struct Tables {
    int V, J, NC, KW, KB;  // KB = NC + (J-1)*9 blend rows
    const float *v_template, *blend, *j_template, *j_dirs;
    const int *parents, *skin_idx;
    const float *skin_w;
    const void *blend_split;  // fp16 x 2 form of `blend` (amav_lbs_prepare_blend_split) or NULL
};

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr size_t kSplitHeaderBytes = 256;  // [0]: float, largest |table entry|; [1]: int, the table's scale exponent

__device__ __forceinline__ int f16_scale_exp(float amax) {
    if (!(amax > 0.f)) return 0;
    return max(-100, min(100, 14 - ilogbf(amax)));
}
__device__ __forceinline__ void split2(float x, _Float16 &a, _Float16 &b) {
    a = (_Float16)x;
    b = (_Float16)__builtin_fmaf((float)a, -1.0f, x);
}
static inline int k16_of(int KB) { return (KB + 31) / 32 * 32; }  // table rows padded to whole 32-row chunks

constexpr int kMaxFeatures = 64 + (kMaxJoints - 1) * 9;  // KB <= num_coeffs + (J - 1) * 9

// Where the pose and the shape / expression coefficients of a frame come from: the keyword arguments of the SMPL-X
// call as the caller holds them (global_orient, body_pose, jaw_pose, ... / betas, expression: renderer.py:261-272),
// concatenated on load -- smplx's torch.cat + `full_pose += pose_mean` + torch.cat were three launches of ~16 us each
// in front of a 12 us kernel.  One part each = an assembled full_pose / coefficient matrix.
struct PoseSource {
    int nparts, ncparts;
    int first[8], cfirst[4];        // first joint / coefficient of every part
    const float *part[8], *cpart[4];
    long long stride[8], cstride[4];  // floats between frames
    const float *mean;              // [J*3] added to the concatenated pose, or NULL
};

// One 64-lane block per frame (kSplit: per padded frame).  kSplit = the feature row of the frame goes straight into the
// fp16 x 2 operand layout of skin_f16_kernel (featH, fscale) instead of the fp32 matrix featT.
template <bool kSplit>
__global__ __launch_bounds__(64) void joint_chain_kernel(Tables t, int F, int Fpad, PoseSource src, float *__restrict__ featT,
                                                        float *__restrict__ A_out, int K16, const unsigned *__restrict__ hdr,
                                                        _Float16 *__restrict__ featH, float *__restrict__ fscale) {
    __shared__ float G[kMaxJoints][12];
    __shared__ float coef[64];
    __shared__ int depth_s[kMaxJoints];
    __shared__ float feat_s[kSplit ? kMaxFeatures : 1];
    const int f = blockIdx.x, j = threadIdx.x;
    const bool live = f < F;  // kSplit only: padded frames carry zero features and scale 1
    auto put_feature = [&](int k, float v) {
        if (kSplit)
            feat_s[k] = v;
        else
            featT[(size_t)k * Fpad + f] = v;
    };
    if (live) {
    if (j < t.NC) {
        const float *base = src.cpart[0];
        long long st = src.cstride[0];
        int fj = 0;
#pragma unroll
        for (int q = 1; q < 4; ++q) {
            const bool in = q < src.ncparts && j >= src.cfirst[q];
            base = in ? src.cpart[q] : base, st = in ? src.cstride[q] : st, fj = in ? src.cfirst[q] : fj;
        }
        const float c = base[(size_t)f * st + (j - fj)];
        coef[j] = c;
        put_feature(j, c);
    }
    __syncthreads();
    float R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    float Jp[3] = {0, 0, 0}, rel[3] = {0, 0, 0};
    int parent = -1, depth = 0;
    if (j < t.J) {
        // Rodrigues, smplx.lbs.batch_rodrigues: eps is added to every component before the norm
        const float *base = src.part[0];
        long long st = src.stride[0];
        int fj = 0;
#pragma unroll
        for (int q = 1; q < 8; ++q) {
            const bool in = q < src.nparts && j >= src.first[q];
            base = in ? src.part[q] : base, st = in ? src.stride[q] : st, fj = in ? src.first[q] : fj;
        }
        base += (size_t)f * st + (j - fj) * 3;
        float rx = base[0], ry = base[1], rz = base[2];
        if (src.mean) rx += src.mean[j * 3], ry += src.mean[j * 3 + 1], rz += src.mean[j * 3 + 2];
        const float ex = rx + 1e-8f, ey = ry + 1e-8f, ez = rz + 1e-8f;
        const float angle = sqrtf(ex * ex + ey * ey + ez * ez);
        const float kx = rx / angle, ky = ry / angle, kz = rz / angle;
        const float s = sinf(angle), c1 = 1.0f - cosf(angle);
        // K = [[0,-kz,ky],[kz,0,-kx],[-ky,kx,0]];  R = I + s K + (1-c) K K
        R[0] = 1.0f + c1 * (-kz * kz - ky * ky);
        R[1] = -s * kz + c1 * (kx * ky);
        R[2] = s * ky + c1 * (kx * kz);
        R[3] = s * kz + c1 * (kx * ky);
        R[4] = 1.0f + c1 * (-kz * kz - kx * kx);
        R[5] = -s * kx + c1 * (ky * kz);
        R[6] = -s * ky + c1 * (kx * kz);
        R[7] = s * kx + c1 * (ky * kz);
        R[8] = 1.0f + c1 * (-ky * ky - kx * kx);
        if (j >= 1) {
#pragma unroll
            for (int e = 0; e < 9; ++e) put_feature(t.NC + (j - 1) * 9 + e, R[e] - ((e == 0 || e == 4 || e == 8) ? 1.0f : 0.0f));
        }
        // joint location: J = J_regressor (v_template + dirs c) = j_template + j_dirs c
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            float acc = t.j_template[j * 3 + d];
            const float *row = t.j_dirs + (size_t)(j * 3 + d) * t.NC;
            for (int l = 0; l < t.NC; ++l) acc += row[l] * coef[l];
            Jp[d] = acc;
        }
        parent = t.parents[j];
        for (int a = parent; a >= 0; a = t.parents[a]) ++depth;
        depth_s[j] = depth;
        G[j][0] = Jp[0], G[j][1] = Jp[1], G[j][2] = Jp[2];  // park J_j for the children
    }
    __syncthreads();
    int max_depth = 0;
    for (int k = 0; k < t.J; ++k) max_depth = max(max_depth, depth_s[k]);
    if (j < t.J && parent >= 0) {
        rel[0] = Jp[0] - G[parent][0], rel[1] = Jp[1] - G[parent][1], rel[2] = Jp[2] - G[parent][2];
    } else {
        rel[0] = Jp[0], rel[1] = Jp[1], rel[2] = Jp[2];
    }
    __syncthreads();
    // chain: G_j = G_parent [R_j | rel_j], level by level
    float Gj[12];
    if (j < t.J && depth == 0) {
        Gj[0] = R[0], Gj[1] = R[1], Gj[2] = R[2], Gj[3] = rel[0];
        Gj[4] = R[3], Gj[5] = R[4], Gj[6] = R[5], Gj[7] = rel[1];
        Gj[8] = R[6], Gj[9] = R[7], Gj[10] = R[8], Gj[11] = rel[2];
#pragma unroll
        for (int e = 0; e < 12; ++e) G[j][e] = Gj[e];
    }
    __syncthreads();
    for (int d = 1; d <= max_depth; ++d) {
        if (j < t.J && depth == d) {
            const float *P = G[parent];  // depth d-1: final since the previous barrier
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float p0 = P[r * 4], p1 = P[r * 4 + 1], p2 = P[r * 4 + 2], p3 = P[r * 4 + 3];
                Gj[r * 4 + 0] = p0 * R[0] + p1 * R[3] + p2 * R[6];
                Gj[r * 4 + 1] = p0 * R[1] + p1 * R[4] + p2 * R[7];
                Gj[r * 4 + 2] = p0 * R[2] + p1 * R[5] + p2 * R[8];
                Gj[r * 4 + 3] = p0 * rel[0] + p1 * rel[1] + p2 * rel[2] + p3;
            }
#pragma unroll
            for (int e = 0; e < 12; ++e) G[j][e] = Gj[e];
        }
        __syncthreads();
    }
    if (j < t.J) {
        // remove the rest pose: translation -= G[:3,:3] J_j
        float *dst = A_out + ((size_t)f * t.J + j) * 12;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            dst[r * 4 + 0] = Gj[r * 4 + 0];
            dst[r * 4 + 1] = Gj[r * 4 + 1];
            dst[r * 4 + 2] = Gj[r * 4 + 2];
            dst[r * 4 + 3] = Gj[r * 4 + 3] - (Gj[r * 4] * Jp[0] + Gj[r * 4 + 1] * Jp[1] + Gj[r * 4 + 2] * Jp[2]);
        }
    }
    }  // live
    if (kSplit) {
        // the frame's largest |feature| -> its scale exponent; features -> featH [part][k / 8][Fpad][8] fp16 (lane
        // (frame, hh) of the skin kernel reads 16 contiguous bytes per part and k-step), fscale[frame] =
        // 2^-(e_frame + e_table) for the epilogue.  Rows KB..K16-1 are zero.  (feat_s is complete: barriers above.)
        const int KB = t.KB;
        float m = 0.f;
        for (int k = j; k < KB; k += 64) m = fmaxf(m, live ? fabsf(feat_s[k]) : 0.f);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        const int e = f16_scale_exp(m);
        const float scale = ldexpf(1.0f, e);
        if (j == 0) fscale[f] = ldexpf(1.0f, -(e + reinterpret_cast<const int *>(hdr)[1]));
        const size_t part = (size_t)(K16 / 8) * Fpad * 8;
        for (int k8 = j; k8 < K16 / 8; k8 += 64) {
            f16x8 p1, p2;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = 8 * k8 + i;
                const float x = (live && k < KB) ? feat_s[k] : 0.f;
                _Float16 a, b;
                split2(x * scale, a, b);
                p1[i] = a, p2[i] = b;
            }
            _Float16 *dst = featH + ((size_t)k8 * Fpad + f) * 8;
            *reinterpret_cast<f16x8 *>(dst) = p1;
            *reinterpret_cast<f16x8 *>(dst + part) = p2;
        }
    }
}

// grid: ceil(V/256) * (Fpad/FT) blocks.  featT rows are [k][Fpad]; A is [Fpad][J][12]; blend is tile-major
// [ceil(V/32)][KB][3][32] (a wave's loads are two coalesced 128-byte rows per component and k).
// Block order: the frame groups of one vertex chunk are adjacent AND land on one XCD (blocks are dealt round-robin
// over 8 XCDs), so each 1.5 MB chunk of the 64 MB blend table is fetched from HBM once and then re-read from that
// XCD's L2 by the other frame groups (placement only affects speed).
template <int FT>
__global__ __launch_bounds__(256) void skin_kernel(Tables t, int F, int Fpad, int nchunks,
                                                   const float *__restrict__ featT, const float *__restrict__ A,
                                                   float *__restrict__ out) {
    extern __shared__ __align__(16) float feat_lds[];  // [KB][FT]: this frame group's slice of the feature matrix
    const int ngroups = Fpad / FT;
    // block id -> (chunk, frame group): ids of one XCD (id % 8) walk chunk-major over that XCD's share of the chunks
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int chunks_per_xcd = (nchunks + 7) / 8;
    const int chunk = (j / ngroups) * 8 + xcd, fg = j % ngroups;
    if (j / ngroups >= chunks_per_xcd || chunk >= nchunks) return;
    const int v = chunk * blockDim.x + threadIdx.x;
    const int f0 = fg * FT;
    // stage the group's features once: the k-loop then reads them as LDS broadcasts (scalar loads of a 32 KB slice
    // per block thrash the scalar cache and put an L2 round trip into every iteration)
    for (int k = threadIdx.x; k < t.KB * FT; k += blockDim.x)
        feat_lds[k] = featT[(size_t)(k / FT) * Fpad + f0 + (k % FT)];
    __syncthreads();
    if (v >= t.V) return;

    float acc[FT][3];
    {
        const float x = t.v_template[v * 3], y = t.v_template[v * 3 + 1], z = t.v_template[v * 3 + 2];
#pragma unroll
        for (int m = 0; m < FT; ++m) acc[m][0] = x, acc[m][1] = y, acc[m][2] = z;
    }
    const float *bl = t.blend + (size_t)(v >> 5) * t.KB * 96 + (v & 31);  // tile-major table: [V/32][KB][3][32]
#pragma unroll 4
    for (int k = 0; k < t.KB; ++k) {
        const float b0 = bl[k * 96], b1 = bl[k * 96 + 32], b2 = bl[k * 96 + 64];
        const float *fk = feat_lds + k * FT;  // same address in every lane: LDS broadcast
#pragma unroll
        for (int m = 0; m < FT; ++m) {
            const float s = fk[m];
            acc[m][0] += s * b0;
            acc[m][1] += s * b1;
            acc[m][2] += s * b2;
        }
    }

    // the vertex's non-zero skinning weights (ascending joint order); 8 live in registers, any more are re-read
    int jidx[8];
    float jw[8];
    const int kw = t.KW;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        jidx[k] = k < kw ? t.skin_idx[(size_t)v * kw + k] : 0;
        jw[k] = k < kw ? t.skin_w[(size_t)v * kw + k] : 0.0f;
    }
#pragma unroll
    for (int m = 0; m < FT; ++m) {
        if (f0 + m >= F) break;
        float Tm[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) Tm[e] = 0.0f;
        const float *Af = A + (size_t)(f0 + m) * t.J * 12;  // 3 x 16 B per (frame, joint), L1/L2 resident
        auto add = [&](int ji, float w) {
            const float4 *a4 = reinterpret_cast<const float4 *>(Af + ji * 12);
            const float4 r0 = a4[0], r1 = a4[1], r2 = a4[2];
            Tm[0] += w * r0.x, Tm[1] += w * r0.y, Tm[2] += w * r0.z, Tm[3] += w * r0.w;
            Tm[4] += w * r1.x, Tm[5] += w * r1.y, Tm[6] += w * r1.z, Tm[7] += w * r1.w;
            Tm[8] += w * r2.x, Tm[9] += w * r2.y, Tm[10] += w * r2.z, Tm[11] += w * r2.w;
        };
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < kw) add(jidx[k], jw[k]);
        for (int k = 8; k < kw; ++k) add(t.skin_idx[(size_t)v * kw + k], t.skin_w[(size_t)v * kw + k]);
        const float x = acc[m][0], y = acc[m][1], z = acc[m][2];
        float *o = out + ((size_t)(f0 + m) * t.V + v) * 3;
        o[0] = Tm[0] * x + Tm[1] * y + Tm[2] * z + Tm[3];
        o[1] = Tm[4] * x + Tm[5] * y + Tm[6] * z + Tm[7];
        o[2] = Tm[8] * x + Tm[9] * y + Tm[10] * z + Tm[11];
    }
}

// MFMA form of the same stage for F > 16: the blend product [F, KB] x [KB, 3V] is a GEMM, so it runs on
// v_mfma_f32_32x32x2_f32 (exact fp32 products and sums; twice the FMA rate of the vector pipe, and one operand word
// per lane per 4096 flops instead of an LDS broadcast per FMA, which is what bounds skin_kernel).
//   block  = 4 waves = 128 frames x one 32-vertex tile; wave = 32 frames x 32 vertices x 3 components = three 32x32
//            accumulators whose column (lane & 31) is the vertex and whose rows (8g + 4*(lane >> 5) + r in register
//            4g + r) are frames: the skinning epilogue finds x, y, z of a (frame, vertex) pair in one lane and a
//            store instruction writes 384 contiguous bytes per frame.
//   table  : tile-major, so a block streams ONE contiguous 194 KB slab, in 12 KB chunks of 32 rows that the block
//            stages through LDS (double buffered, one barrier per chunk): each table byte leaves HBM once per
//            128 frames instead of once per wave that happens to miss L2.
//   feats  : featT[k][frame] rows straight from L2 (0.5 MB, hot), one chunk ahead in registers; rows past KB are zero
//            (host pads featT), so the last chunk needs no special case.
// The frame groups of a vertex tile are adjacent in block order and share an XCD (as in skin_kernel).
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kMfmaWaves = 4;                    // frame tiles (of 32) per block
constexpr int kMfmaKC = 32;                      // table rows per staged chunk
constexpr int kMfmaChunk4 = kMfmaKC * 96 / 4;    // float4 per chunk (768)
constexpr int kMfmaKPad = 2 * kMfmaKC;           // zero rows appended to featT on this path

__global__ __launch_bounds__(64 * kMfmaWaves, 3) void skin_mfma_kernel(Tables t, int F, int Fpad, int ntiles,
                                                                    const float *__restrict__ featT,
                                                                    const float *__restrict__ A,
                                                                    float *__restrict__ out) {
    __shared__ float4 Bs[2][kMfmaChunk4];
    const int ngroups = (Fpad / 32 + kMfmaWaves - 1) / kMfmaWaves;
    const int xcd = blockIdx.x & 7, bj = blockIdx.x >> 3;
    const int tile = (bj / ngroups) * 8 + xcd, fg = bj % ngroups;
    if (tile >= ntiles) return;  // block-uniform
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ftile = fg * kMfmaWaves + wave;
    const bool active = ftile * 32 < Fpad;  // a wave without frames still helps staging and joins the barriers
    const int f0 = active ? ftile * 32 : 0;
    const int c = lane & 31, hh = lane >> 5;
    const int v = tile * 32 + c;

    f32x16 X, Y, Z;
    {
        const int vl = min(v, t.V - 1);
        const float x = t.v_template[vl * 3], y = t.v_template[vl * 3 + 1], z = t.v_template[vl * 3 + 2];
#pragma unroll
        for (int r = 0; r < 16; ++r) X[r] = x, Y[r] = y, Z[r] = z;
    }
    const float4 *bt4 = reinterpret_cast<const float4 *>(t.blend + (size_t)tile * t.KB * 96);
    const int total4 = t.KB * 24;  // float4 in this tile's slab
    const int nchunks = (t.KB + kMfmaKC - 1) / kMfmaKC;
    const unsigned lane_a = (unsigned)(hh * Fpad + f0 + c);
    float4 breg[3];
    float a_cur[kMfmaKC / 2], a_nxt[kMfmaKC / 2];
    auto gload = [&](int ch) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int idx = ch * kMfmaChunk4 + (int)threadIdx.x + 256 * i;
            breg[i] = idx < total4 ? bt4[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto aload = [&](float (&a)[kMfmaKC / 2], int ch) {
        const float *pa = featT + (size_t)(ch * kMfmaKC) * Fpad;
#pragma unroll
        for (int s = 0; s < kMfmaKC / 2; ++s) a[s] = pa[(size_t)(2 * s) * Fpad + lane_a];
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 3; ++i) Bs[buf][threadIdx.x + 256 * i] = breg[i];
    };
    gload(0);
    aload(a_cur, 0);
    stage(0);
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        // always issued (the last iteration re-reads its own chunk and discards it): a conditional prefetch would
        // make the compiler wait for the loads it has just issued
        const int nxt = min(ch + 1, nchunks - 1);
        gload(nxt);
        aload(a_nxt, nxt);
        __builtin_amdgcn_sched_barrier(0);  // the next chunk's loads stay ahead of this chunk's MFMAs
        if (active) {
            const float *bs = reinterpret_cast<const float *>(Bs[buf]) + hh * 96 + c;
            float q0 = bs[0], q1 = bs[32], q2 = bs[64];
#pragma unroll
            for (int s = 0; s < kMfmaKC / 2; ++s) {
                const float p0 = q0, p1 = q1, p2 = q2;
                if (s + 1 < kMfmaKC / 2) q0 = bs[(s + 1) * 192], q1 = bs[(s + 1) * 192 + 32], q2 = bs[(s + 1) * 192 + 64];
                X = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[s], p0, X, 0, 0, 0);
                Y = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[s], p1, Y, 0, 0, 0);
                Z = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[s], p2, Z, 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        stage(buf ^ 1);  // last read in iteration ch - 1, before that iteration's barrier
#pragma unroll
        for (int s = 0; s < kMfmaKC / 2; ++s) a_cur[s] = a_nxt[s];
        __syncthreads();
    }
    if (!active) return;
    if (v >= t.V) return;

    int jidx[8];
    float jw[8];
    const int kw = t.KW;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        jidx[k] = k < kw ? t.skin_idx[(size_t)v * kw + k] : 0;
        jw[k] = k < kw ? t.skin_w[(size_t)v * kw + k] : 0.0f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int f = f0 + 8 * (r >> 2) + 4 * hh + (r & 3);
        if (f >= F) continue;
        float Tm[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) Tm[e] = 0.0f;
        const float *Af = A + (size_t)f * t.J * 12;
        auto add = [&](int ji, float w) {
            const float4 *a4 = reinterpret_cast<const float4 *>(Af + ji * 12);
            const float4 r0 = a4[0], r1 = a4[1], r2 = a4[2];
            Tm[0] += w * r0.x, Tm[1] += w * r0.y, Tm[2] += w * r0.z, Tm[3] += w * r0.w;
            Tm[4] += w * r1.x, Tm[5] += w * r1.y, Tm[6] += w * r1.z, Tm[7] += w * r1.w;
            Tm[8] += w * r2.x, Tm[9] += w * r2.y, Tm[10] += w * r2.z, Tm[11] += w * r2.w;
        };
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < kw) add(jidx[k], jw[k]);
        for (int k = 8; k < kw; ++k) add(t.skin_idx[(size_t)v * kw + k], t.skin_w[(size_t)v * kw + k]);
        const float x = X[r], y = Y[r], z = Z[r];
        float *o = out + ((size_t)f * t.V + v) * 3;
        o[0] = Tm[0] * x + Tm[1] * y + Tm[2] * z + Tm[3];
        o[1] = Tm[4] * x + Tm[5] * y + Tm[6] * z + Tm[7];
        o[2] = Tm[8] * x + Tm[9] * y + Tm[10] * z + Tm[11];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// fp16 x 2 form of the blend product (the split-product arithmetic of csrc/attention.hip):
//   x 2^e = h1 + h2 (two fp16 parts, 22 bits),   a b = (a1 b1 + a1 b2 + a2 b1) 2^-(ea + eb)   to 2^-22,
// three v_mfma_f32_32x32x16_f16 per fp32 product, each 8 x the k depth of the fp32 MFMA at half its cycles.
// Scaling (exact powers of two): the table by ONE exponent from its largest magnitude (static, prepared once); every
// frame's feature column by its own exponent from that frame's largest feature (joint_chain_kernel<true>), undone per
// accumulator row in the epilogue.  Both put the largest magnitude in [2^14, 2^15): nothing overflows, and entries
// down to 2^-17 of the largest keep all 22 bits (smaller ones an absolute error below 2^-39 of it).

__global__ __launch_bounds__(256) void table_absmax_kernel(const float4 *__restrict__ blend4, long long n4,
                                                           unsigned *__restrict__ hdr) {
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 v = blend4[i];
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(hdr, __float_as_uint(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]))));
}

// blend [tile][KB][3][32] fp32 -> split [tile][k-step of 16][part][comp][lane half hh][vertex c][8 k] fp16: the B operand
// fragment of lane (c, hh) for one (k-step, part, component) is 16 contiguous bytes, and a 32-row chunk is 12 KB that
// the skin kernel copies to LDS as is.  One thread per (tile, k-step, comp, hh, c): 8 strided reads, two 16-byte writes.
__global__ __launch_bounds__(256) void table_split_kernel(const float *__restrict__ blend, int KB, int K16, long long items,
                                                          unsigned *__restrict__ hdr, _Float16 *__restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int e = f16_scale_exp(__uint_as_float(hdr[0]));
    if (i == 0) reinterpret_cast<int *>(hdr)[1] = e;
    if (i >= items) return;
    const int c = (int)(i & 31), hh = (int)((i >> 5) & 1), comp = (int)((i >> 6) % 3);
    const long long rest = (i >> 6) / 3;  // tile * steps + step
    const int steps = K16 / 16, step = (int)(rest % steps);
    const long long tile = rest / steps;
    const float scale = ldexpf(1.0f, e);
    f16x8 p1, p2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 16 * step + 8 * hh + j;
        const float x = k < KB ? blend[((tile * KB + k) * 3 + comp) * 32 + c] : 0.f;
        _Float16 a, b;
        split2(x * scale, a, b);
        p1[j] = a, p2[j] = b;
    }
    _Float16 *dst = out + (((tile * steps + step) * 2) * 3 + comp) * 512 + hh * 256 + c * 8;
    *reinterpret_cast<f16x8 *>(dst) = p1;
    *reinterpret_cast<f16x8 *>(dst + 3 * 512) = p2;
}

// Same decomposition as skin_mfma_kernel (block = 4 waves = 128 frames x one 32-vertex tile, the tile's table slab
// streamed once through double-buffered LDS in 12 KB chunks of 32 rows, one barrier per chunk); per chunk a wave issues
// 2 k-steps x 3 components x 3 partial products = 18 MFMAs of 32 cycles where the fp32 kernel issues 48 of 64.
__global__ __launch_bounds__(64 * kMfmaWaves, 3) void skin_f16_kernel(Tables t, int F, int Fpad, int ntiles, int K16,
                                                                   const _Float16 *__restrict__ featH,
                                                                   const float *__restrict__ fscale,
                                                                   const float *__restrict__ A,
                                                                   float *__restrict__ out) {
    __shared__ float4 Bs[2][kMfmaChunk4];
    const int ngroups = (Fpad / 32 + kMfmaWaves - 1) / kMfmaWaves;
    const int xcd = blockIdx.x & 7, bj = blockIdx.x >> 3;
    const int tile = (bj / ngroups) * 8 + xcd, fg = bj % ngroups;
    if (tile >= ntiles) return;  // block-uniform
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ftile = fg * kMfmaWaves + wave;
    const bool active = ftile * 32 < Fpad;
    const int f0 = active ? ftile * 32 : 0;
    const int c = lane & 31, hh = lane >> 5;
    const int v = tile * 32 + c;

    f32x16 X, Y, Z;
#pragma unroll
    for (int r = 0; r < 16; ++r) X[r] = 0.f, Y[r] = 0.f, Z[r] = 0.f;
    const int nchunks = K16 / kMfmaKC;
    const float4 *bt4 = reinterpret_cast<const float4 *>(static_cast<const char *>(t.blend_split) + kSplitHeaderBytes) +
                        (size_t)tile * nchunks * kMfmaChunk4;
    // A fragments: lane (frame f0 + c, hh), k-step s of chunk ch: featH[part][(ch * 32 + 16 s) / 8 + hh][frame][0..7]
    const size_t part_a = (size_t)(K16 / 8) * Fpad * 8;
    const _Float16 *fa = featH + ((size_t)hh * Fpad + f0 + c) * 8;
    // Named registers and macros: arrays captured by lambdas were demoted to scratch memory here.  The table is
    // prefetched TWO chunks ahead (chunk ch in LDS being read, ch + 1 in registers waiting for its LDS buffer, ch + 2 in
    // flight): with one chunk ahead the loop ran at one HBM round trip per chunk (3.6 us against 0.3 us of MFMA work).
    float4 bn0, bn1, bn2, bf0, bf1, bf2;                   // b{n: next chunk, f: the one after}
    f16x8 ac00, ac01, ac10, ac11, an00, an01, an10, an11;  // a{c: current, n: next}{k-step}{part}
#define AMAV_LBS_GLOAD(r0_, r1_, r2_, ch_)                                    \
    {                                                                         \
        const float4 *src_ = bt4 + (size_t)(ch_) * kMfmaChunk4 + threadIdx.x; \
        r0_ = src_[0], r1_ = src_[256], r2_ = src_[512];                      \
    }
#define AMAV_LBS_ALOAD(a00_, a01_, a10_, a11_, ch_)                                          \
    {                                                                                        \
        const _Float16 *p_ = fa + (size_t)((ch_) * 4) * Fpad * 8;                            \
        a00_ = *reinterpret_cast<const f16x8 *>(p_);                                         \
        a01_ = *reinterpret_cast<const f16x8 *>(p_ + part_a);                                \
        a10_ = *reinterpret_cast<const f16x8 *>(p_ + (size_t)2 * Fpad * 8);                  \
        a11_ = *reinterpret_cast<const f16x8 *>(p_ + (size_t)2 * Fpad * 8 + part_a);         \
    }
#define AMAV_LBS_STAGE(buf_, r0_, r1_, r2_) \
    Bs[buf_][threadIdx.x] = r0_, Bs[buf_][threadIdx.x + 256] = r1_, Bs[buf_][threadIdx.x + 512] = r2_;
    // fragment (k-step s, part p, component q) at ((s * 2 + p) * 3 + q) * 512 halfs; one component at a time keeps
    // two fragments live instead of six
#define AMAV_LBS_COMP(s_, q_, acc_, a1_, a2_)                                                              \
    {                                                                                                      \
        const f16x8 b1 = *reinterpret_cast<const f16x8 *>(bs + (((s_) * 2 + 0) * 3 + (q_)) * 512);         \
        const f16x8 b2 = *reinterpret_cast<const f16x8 *>(bs + (((s_) * 2 + 1) * 3 + (q_)) * 512);         \
        acc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2_, b1, acc_, 0, 0, 0); /* small terms first */     \
        acc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1_, b2, acc_, 0, 0, 0);                             \
        acc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1_, b1, acc_, 0, 0, 0);                             \
    }
#define AMAV_LBS_STEP(s_, a1_, a2_) \
    AMAV_LBS_COMP(s_, 0, X, a1_, a2_) AMAV_LBS_COMP(s_, 1, Y, a1_, a2_) AMAV_LBS_COMP(s_, 2, Z, a1_, a2_)
    AMAV_LBS_GLOAD(bn0, bn1, bn2, 0)
    AMAV_LBS_ALOAD(ac00, ac01, ac10, ac11, 0)
    AMAV_LBS_STAGE(0, bn0, bn1, bn2)
    AMAV_LBS_GLOAD(bn0, bn1, bn2, min(1, nchunks - 1))
    __syncthreads();
#if AMAV_LBS_ABLATE == 2  /* diagnostic: one chunk instead of the whole table (epilogue + start-up only) */
    for (int ch = 0; ch < 1; ++ch) {
#else
    for (int ch = 0; ch < nchunks; ++ch) {
#endif
        const int buf = ch & 1;
        // always issued (the last iterations re-read the last chunk and discard it), as in skin_mfma_kernel
        AMAV_LBS_GLOAD(bf0, bf1, bf2, min(ch + 2, nchunks - 1))
        AMAV_LBS_ALOAD(an00, an01, an10, an11, min(ch + 1, nchunks - 1))
        __builtin_amdgcn_sched_barrier(0);
        if (active) {
            const _Float16 *bs = reinterpret_cast<const _Float16 *>(Bs[buf]) + hh * 256 + c * 8;
            AMAV_LBS_STEP(0, ac00, ac01)
            AMAV_LBS_STEP(1, ac10, ac11)
        }
        __builtin_amdgcn_sched_barrier(0);
        AMAV_LBS_STAGE(buf ^ 1, bn0, bn1, bn2)  // chunk ch + 1; its buffer was last read in iteration ch - 1
        bn0 = bf0, bn1 = bf1, bn2 = bf2;
        ac00 = an00, ac01 = an01, ac10 = an10, ac11 = an11;
        __syncthreads();
    }
#undef AMAV_LBS_GLOAD
#undef AMAV_LBS_ALOAD
#undef AMAV_LBS_STAGE
#undef AMAV_LBS_COMP
#undef AMAV_LBS_STEP
    if (!active) return;
    if (v >= t.V) return;
#if AMAV_LBS_ABLATE == 1  /* diagnostic (tools/ablate_lbs.sh, stand-alone timing only): no skinning epilogue */
    if (X[0] + Y[3] + Z[7] == 12345.678f) out[0] = X[1];
    return;
#endif

    const float tx = t.v_template[v * 3], ty = t.v_template[v * 3 + 1], tz = t.v_template[v * 3 + 2];
    float fs_all[16];  // the scale of every accumulator row's frame
#pragma unroll
    for (int r = 0; r < 16; ++r) fs_all[r] = fscale[min(f0 + 8 * (r >> 2) + 4 * hh + (r & 3), Fpad - 1)];
    int jidx[8];
    float jw[8];
    const int kw = t.KW;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        jidx[k] = k < kw ? t.skin_idx[(size_t)v * kw + k] : 0;
        jw[k] = k < kw ? t.skin_w[(size_t)v * kw + k] : 0.0f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int f = f0 + 8 * (r >> 2) + 4 * hh + (r & 3);
        if (f >= F) continue;
        float Tm[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) Tm[e] = 0.0f;
        const float *Af = A + (size_t)f * t.J * 12;
        auto add = [&](int ji, float w) {
            const float4 *a4 = reinterpret_cast<const float4 *>(Af + ji * 12);
            const float4 r0 = a4[0], r1 = a4[1], r2 = a4[2];
            Tm[0] += w * r0.x, Tm[1] += w * r0.y, Tm[2] += w * r0.z, Tm[3] += w * r0.w;
            Tm[4] += w * r1.x, Tm[5] += w * r1.y, Tm[6] += w * r1.z, Tm[7] += w * r1.w;
            Tm[8] += w * r2.x, Tm[9] += w * r2.y, Tm[10] += w * r2.z, Tm[11] += w * r2.w;
        };
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < kw) add(jidx[k], jw[k]);
        for (int k = 8; k < kw; ++k) add(t.skin_idx[(size_t)v * kw + k], t.skin_w[(size_t)v * kw + k]);
        const float fs = fs_all[r];
        const float x = tx + X[r] * fs, y = ty + Y[r] * fs, z = tz + Z[r] * fs;
        float *o = out + ((size_t)f * t.V + v) * 3;
        o[0] = Tm[0] * x + Tm[1] * y + Tm[2] * z + Tm[3];
        o[1] = Tm[4] * x + Tm[5] * y + Tm[6] * z + Tm[7];
        o[2] = Tm[8] * x + Tm[9] * y + Tm[10] * z + Tm[11];
    }
}

__global__ __launch_bounds__(256) void gather_kernel(int V, int N, const float *__restrict__ verts,
                                                     const int4 *__restrict__ idx, float *__restrict__ out) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y;
    if (n >= N) return;
    const int4 id = idx[n];
    const float *vf = verts + (size_t)f * V * 3;
    float *o = out + ((size_t)f * N + n) * 3;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float p0 = (vf[id.x * 3 + d] + vf[id.y * 3 + d]) * 0.5f;
        const float p1 = (vf[id.z * 3 + d] + vf[id.w * 3 + d]) * 0.5f;
        o[d] = (p0 + p1) * 0.5f;
    }
}

// Frames per thread of skin_kernel, or 0: the MFMA kernel (F > 16; frames padded to its 32-row tiles).
static int frame_tile(int F, int /*KB*/) {
    static const int forced = getenv("AMAV_LBS_FT") ? atoi(getenv("AMAV_LBS_FT")) : -1;  // tuning aid: 0, 4, 8, 16, 32
    if (forced == 4 || forced == 8 || forced == 16 || forced == 32) return forced;
    if (forced == 0) return 0;
    return F <= 8 ? 4 : (F <= 16 ? 16 : 0);
}
static int frame_pad(int F, int FT) { return FT ? (F + FT - 1) / FT * FT : (F + 31) / 32 * 32; }

}  // namespace lbs
}  // namespace amav

using namespace amav;
using namespace amav::lbs;

static int validate_tables(const amav_body_tables *t, const char *who) {
    AMAV_REQUIRE(t != nullptr, "%s: tables is NULL", who);
    AMAV_REQUIRE(t->num_verts > 0 && t->num_joints > 0 && t->num_joints <= kMaxJoints, "%s: bad V=%d J=%d", who,
                 t->num_verts, t->num_joints);
    AMAV_REQUIRE(t->num_coeffs > 0 && t->num_coeffs <= 64, "%s: num_coeffs %d not in 1..64", who, t->num_coeffs);
    AMAV_REQUIRE(t->skin_k > 0 && t->skin_k <= t->num_joints, "%s: bad skin_k %d", who, t->skin_k);
    AMAV_REQUIRE(t->v_template && t->blend && t->j_template && t->j_dirs && t->parents && t->skin_idx && t->skin_w,
                 "%s: NULL table", who);
    AMAV_REQUIRE((reinterpret_cast<uintptr_t>(t->blend) & 15) == 0, "%s: blend table not 16-byte aligned", who);
    return AMAV_OK;
}

// AMAV_LBS=f32 keeps the fp32 MFMA kernel even when the tables carry a split blend table
static bool lbs_use_split(const amav_body_tables *t, int FT) {
    static const bool off = [] {
        const char *e = getenv("AMAV_LBS");
        return e && strcmp(e, "f32") == 0;
    }();
    const int chosen = option_lbs();  // amav_set_option("lbs", ...)
    return FT == 0 && t->blend_split != nullptr && (chosen >= 0 ? chosen == 1 : !off);
}

static size_t lbs_ws(int F, const amav_body_tables *t, float **featT, float **A, void *ws, _Float16 **featH = nullptr,
                     float **fscale = nullptr) {
    const int KB = t->num_coeffs + (t->num_joints - 1) * 9;
    const int FT = frame_tile(F, KB);
    const int Fpad = frame_pad(F, FT);
    Carver c(ws);
    float *ft = c.take<float>((size_t)(KB + (FT ? 0 : kMfmaKPad)) * Fpad);
    float *a = c.take<float>((size_t)Fpad * t->num_joints * 12);
    if (featT) *featT = ft;
    if (A) *A = a;
    if (lbs_use_split(t, FT)) {
        _Float16 *fh = c.take<_Float16>((size_t)2 * k16_of(KB) * Fpad);
        float *fs = c.take<float>((size_t)Fpad);
        if (featH) *featH = fh;
        if (fscale) *fscale = fs;
    }
    return c.total();
}

static size_t blend_split_bytes(const amav_body_tables *t) {
    const int KB = t->num_coeffs + (t->num_joints - 1) * 9;
    const size_t ntiles = ((size_t)t->num_verts + 31) / 32;
    return kSplitHeaderBytes + ntiles * (k16_of(KB) / 16) * 2 * 3 * 512 * sizeof(_Float16);
}

extern "C" size_t amav_lbs_blend_split_bytes(const amav_body_tables *t) {
    if (validate_tables(t, "amav_lbs_blend_split_bytes") != AMAV_OK) return 0;
    return blend_split_bytes(t);
}

extern "C" int amav_lbs_prepare_blend_split(const amav_body_tables *tb, void *out, size_t out_bytes, void *stream_) {
    if (int rc = validate_tables(tb, "amav_lbs_prepare_blend_split")) return rc;
    AMAV_REQUIRE(out && (reinterpret_cast<uintptr_t>(out) & 255) == 0, "amav_lbs_prepare_blend_split: out must be 256-byte aligned");
    const size_t need = blend_split_bytes(tb);
    if (out_bytes < need)
        return fail(AMAV_ERR_WORKSPACE, "amav_lbs_prepare_blend_split: buffer %zu < required %zu", out_bytes, need);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int KB = tb->num_coeffs + (tb->num_joints - 1) * 9, K16 = k16_of(KB);
    const long long ntiles = (tb->num_verts + 31) / 32;
    unsigned *hdr = static_cast<unsigned *>(out);
    AMAV_REQUIRE(zero_async(hdr, kSplitHeaderBytes, stream) == hipSuccess, "amav_lbs_prepare_blend_split: header clear failed");
    const long long n4 = ntiles * KB * 24;
    table_absmax_kernel<<<(unsigned)std::min<long long>((n4 + 255) / 256, 512), 256, 0, stream>>>(
        reinterpret_cast<const float4 *>(tb->blend), n4, hdr);
    const long long items = ntiles * (K16 / 16) * 3 * 64;
    table_split_kernel<<<(unsigned)((items + 255) / 256), 256, 0, stream>>>(
        tb->blend, KB, K16, items, hdr, reinterpret_cast<_Float16 *>(static_cast<char *>(out) + kSplitHeaderBytes));
    return check_launch("amav_lbs_prepare_blend_split");
}

extern "C" size_t amav_lbs_workspace_bytes(int F, const amav_body_tables *t) {
    if (F <= 0 || validate_tables(t, "amav_lbs_workspace_bytes") != AMAV_OK) return 0;
    return lbs_ws(F, t, nullptr, nullptr, nullptr);
}

extern "C" int amav_lbs_forward_parts(int F, const amav_body_tables *tb, const amav_pose_parts *pp, float *out_vertices,
                                      float *out_A, void *workspace, size_t workspace_bytes, void *stream_) {
    AMAV_REQUIRE(F > 0, "amav_lbs_forward: F=%d", F);
    if (int rc = validate_tables(tb, "amav_lbs_forward")) return rc;
    AMAV_REQUIRE(pp && out_vertices && workspace, "amav_lbs_forward: NULL pointer");
    PoseSource src;
    {
        AMAV_REQUIRE(pp->num_pose_parts >= 1 && pp->num_pose_parts <= 8 && pp->num_coeff_parts >= 1 && pp->num_coeff_parts <= 4,
                     "amav_lbs_forward: %d pose parts (1..8), %d coefficient parts (1..4)", pp->num_pose_parts, pp->num_coeff_parts);
        int joints = 0, ncoef = 0;
        for (int q = 0; q < 8; ++q) {
            const bool used = q < pp->num_pose_parts;
            AMAV_REQUIRE(!used || (pp->pose[q] && pp->pose_joints[q] > 0 && pp->pose_stride[q] >= 3ll * pp->pose_joints[q]),
                         "amav_lbs_forward: pose part %d: NULL, no joints, or frame stride %lld < 3 * %d", q,
                         (long long)pp->pose_stride[q], pp->pose_joints[q]);
            src.first[q] = joints, src.part[q] = used ? pp->pose[q] : nullptr, src.stride[q] = used ? pp->pose_stride[q] : 0;
            if (used) joints += pp->pose_joints[q];
        }
        for (int q = 0; q < 4; ++q) {
            const bool used = q < pp->num_coeff_parts;
            AMAV_REQUIRE(!used || (pp->coeff[q] && pp->coeff_count[q] > 0 && pp->coeff_stride[q] >= pp->coeff_count[q]),
                         "amav_lbs_forward: coefficient part %d: NULL, empty, or frame stride %lld < %d", q,
                         (long long)pp->coeff_stride[q], pp->coeff_count[q]);
            src.cfirst[q] = ncoef, src.cpart[q] = used ? pp->coeff[q] : nullptr, src.cstride[q] = used ? pp->coeff_stride[q] : 0;
            if (used) ncoef += pp->coeff_count[q];
        }
        AMAV_REQUIRE(joints == tb->num_joints && ncoef == tb->num_coeffs,
                     "amav_lbs_forward: the parts hold %d joints / %d coefficients, the tables %d / %d", joints, ncoef,
                     tb->num_joints, tb->num_coeffs);
        src.nparts = pp->num_pose_parts, src.ncparts = pp->num_coeff_parts, src.mean = pp->pose_mean;
    }
    float *featT = nullptr, *A = nullptr, *fscale = nullptr;
    _Float16 *featH = nullptr;
    const size_t need = lbs_ws(F, tb, &featT, &A, workspace, &featH, &fscale);
    if (workspace_bytes < need)
        return fail(AMAV_ERR_WORKSPACE, "amav_lbs_forward: workspace %zu < required %zu", workspace_bytes, need);
    Tables t;
    t.V = tb->num_verts, t.J = tb->num_joints, t.NC = tb->num_coeffs, t.KW = tb->skin_k;
    t.KB = t.NC + (t.J - 1) * 9;
    t.v_template = tb->v_template, t.blend = tb->blend, t.j_template = tb->j_template, t.j_dirs = tb->j_dirs;
    t.parents = tb->parents, t.skin_idx = tb->skin_idx, t.skin_w = tb->skin_w, t.blend_split = tb->blend_split;
    const int FT = frame_tile(F, t.KB);
    const int Fpad = frame_pad(F, FT);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    float *A_dst = out_A ? out_A : A;
    if (lbs_use_split(tb, FT)) {
        // product path: the chain kernel writes the fp16 x 2 feature operand itself (padded frames included)
        const int K16 = k16_of(t.KB);
        const int ntiles = (t.V + 31) / 32, ngroups = (Fpad / 32 + kMfmaWaves - 1) / kMfmaWaves;
        const unsigned mgrid = (unsigned)(((ntiles + 7) / 8) * 8 * ngroups);
        joint_chain_kernel<true><<<Fpad, 64, 0, stream>>>(t, F, Fpad, src, nullptr, A_dst, K16,
                                                          static_cast<const unsigned *>(t.blend_split), featH, fscale);
        skin_f16_kernel<<<mgrid, 64 * kMfmaWaves, 0, stream>>>(t, F, Fpad, ntiles, K16, featH, fscale, A_dst, out_vertices);
        return check_launch("amav_lbs_forward");
    }
    // padded frame columns of featT must be finite (they feed FMAs whose results are discarded), and the MFMA path
    // reads kMfmaKPad zero rows past the table
    const size_t feat_rows = (size_t)t.KB + (FT ? 0 : kMfmaKPad);
    if ((Fpad != F || FT == 0) && zero_async(featT, feat_rows * Fpad * sizeof(float), stream) != hipSuccess)
        return fail(AMAV_ERR_LAUNCH, "amav_lbs_forward: padding clear failed");
    joint_chain_kernel<false><<<F, 64, 0, stream>>>(t, F, Fpad, src, featT, A_dst, 0, nullptr, nullptr, nullptr);
    if (FT == 0) {
        const int ntiles = (t.V + 31) / 32, ngroups = (Fpad / 32 + kMfmaWaves - 1) / kMfmaWaves;
        const unsigned mgrid = (unsigned)(((ntiles + 7) / 8) * 8 * ngroups);
        skin_mfma_kernel<<<mgrid, 64 * kMfmaWaves, 0, stream>>>(t, F, Fpad, ntiles, featT, A_dst, out_vertices);
        return check_launch("amav_lbs_forward");
    }
    const int nchunks = (t.V + 255) / 256;
    const unsigned grid = (unsigned)(((nchunks + 7) / 8) * 8 * (Fpad / FT));
    const size_t lds = (size_t)FT * t.KB * sizeof(float);
    if (FT == 4)
        skin_kernel<4><<<grid, 256, lds, stream>>>(t, F, Fpad, nchunks, featT, A_dst, out_vertices);
    else if (FT == 8)
        skin_kernel<8><<<grid, 256, lds, stream>>>(t, F, Fpad, nchunks, featT, A_dst, out_vertices);
    else if (FT == 32)
        skin_kernel<32><<<grid, 256, lds, stream>>>(t, F, Fpad, nchunks, featT, A_dst, out_vertices);
    else
        skin_kernel<16><<<grid, 256, lds, stream>>>(t, F, Fpad, nchunks, featT, A_dst, out_vertices);
    return check_launch("amav_lbs_forward");
}

extern "C" int amav_lbs_forward(int F, const amav_body_tables *tb, const float *full_pose, const float *coeffs,
                                float *out_vertices, float *out_A, void *workspace, size_t workspace_bytes,
                                void *stream) {
    AMAV_REQUIRE(tb != nullptr, "amav_lbs_forward: NULL tables");
    AMAV_REQUIRE(full_pose && coeffs, "amav_lbs_forward: NULL pointer");
    amav_pose_parts pp = {};
    pp.num_pose_parts = 1, pp.pose[0] = full_pose, pp.pose_joints[0] = tb->num_joints, pp.pose_stride[0] = 3ll * tb->num_joints;
    pp.num_coeff_parts = 1, pp.coeff[0] = coeffs, pp.coeff_count[0] = tb->num_coeffs, pp.coeff_stride[0] = tb->num_coeffs;
    return amav_lbs_forward_parts(F, tb, &pp, out_vertices, out_A, workspace, workspace_bytes, stream);
}

extern "C" int amav_points_gather(int F, int V, int N, const float *vertices, const int32_t *idx, float *out,
                                  void *stream) {
    AMAV_REQUIRE(F > 0 && V > 0 && N > 0, "amav_points_gather: bad sizes");
    AMAV_REQUIRE(vertices && idx && out, "amav_points_gather: NULL pointer");
    AMAV_REQUIRE((reinterpret_cast<uintptr_t>(idx) & 15) == 0, "amav_points_gather: idx not 16-B aligned");
    const dim3 grid((N + 255) / 256, F);
    gather_kernel<<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(V, N, vertices, reinterpret_cast<const int4 *>(idx),
                                                                    out);
    return check_launch("amav_points_gather");
}
