// Library GEMM entry for the fp16 x 2 split projections of the transformer step: out[rows, n] (fp32) =
// alpha * a[rows, k3] (fp16) x w[n, k3]^T (fp16), fp32 accumulate -- the three partial products of an fp32-equivalent
// nn.Linear (src/models/transformers.py:70-84, 448, 505) concatenated along K.  The arithmetic
// is hipBLASLt's; what this file adds is the choice of kernel: hipBLASLt's first heuristic answer is 39 % slower than
// its best kernel for the 2048 -> 512 feed-forward output projection (K' = 6144: 0.066 vs 0.040 ms at 6304 rows,
// tools/hipblaslt_probe.cpp), so the caller may name the algorithm by its library index (found once per shape by
// amav_gemm_split_fp16_tune, shipped in gemm_split_tuning_gfx950.csv) and gets the heuristic's choice otherwise.
#include <hipblaslt/hipblaslt-ext.hpp>
#include <hipblaslt/hipblaslt.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "amav_common.h"

namespace amav {
namespace gemm {

struct Plan {
    hipblasLtMatmulDesc_t desc = nullptr;
    hipblasLtMatrixLayout_t la = nullptr, lb = nullptr, ld = nullptr;
    hipblasLtMatmulAlgo_t algo;
    size_t workspace = 0;
    bool ok = false;
};

static hipblasLtHandle_t handle() {
    static hipblasLtHandle_t h = [] {
        hipblasLtHandle_t x = nullptr;
        return hipblasLtCreate(&x) == HIPBLAS_STATUS_SUCCESS ? x : nullptr;
    }();
    return h;
}

static std::mutex g_lock;
static std::map<std::tuple<long long, int, int, int>, Plan> g_plans;  // (rows, n, k3, algo index)

static bool make_layouts(Plan &p, long long rows, int n, int k3) {
    const hipblasOperation_t opT = HIPBLAS_OP_T, opN = HIPBLAS_OP_N;
    return hipblasLtMatmulDescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) == HIPBLAS_STATUS_SUCCESS &&
           hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opT, sizeof(opT)) == HIPBLAS_STATUS_SUCCESS &&
           hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opN, sizeof(opN)) == HIPBLAS_STATUS_SUCCESS &&
           // row-major [n, k3] weights = column-major k3 x n; row-major [rows, k3] activations = k3 x rows; out = n x rows
           hipblasLtMatrixLayoutCreate(&p.la, HIP_R_16F, k3, n, k3) == HIPBLAS_STATUS_SUCCESS &&
           hipblasLtMatrixLayoutCreate(&p.lb, HIP_R_16F, k3, rows, k3) == HIPBLAS_STATUS_SUCCESS &&
           hipblasLtMatrixLayoutCreate(&p.ld, HIP_R_32F, n, rows, n) == HIPBLAS_STATUS_SUCCESS;
}

// the plan for (shape, algorithm index); index < 0 or an index this library build does not know: the heuristic's choice
static const Plan *plan_for(long long rows, int n, int k3, int index, size_t ws_bytes) {
    std::lock_guard<std::mutex> guard(g_lock);
    const auto key = std::make_tuple(rows, n, k3, index);
    auto it = g_plans.find(key);
    if (it != g_plans.end()) return it->second.ok && it->second.workspace <= ws_bytes ? &it->second : nullptr;
    Plan p;
    const float one = 1.f, zero = 0.f;
    if (handle() && make_layouts(p, rows, n, k3)) {
        if (index >= 0) {
            std::vector<int> idx{index};
            std::vector<hipblasLtMatmulHeuristicResult_t> res;
            if (hipblaslt_ext::getAlgosFromIndex(handle(), idx, res) == HIPBLAS_STATUS_SUCCESS && !res.empty()) {
                size_t need = 0;
                if (hipblaslt_ext::matmulIsAlgoSupported(handle(), p.desc, &one, p.la, p.lb, &zero, p.ld, p.ld, res[0].algo, need) ==
                    HIPBLAS_STATUS_SUCCESS) {
                    p.algo = res[0].algo, p.workspace = need, p.ok = true;
                }
            }
        }
        if (!p.ok) {
            hipblasLtMatmulPreference_t pref = nullptr;
            hipblasLtMatmulHeuristicResult_t heur[1];
            int got = 0;
            if (hipblasLtMatmulPreferenceCreate(&pref) == HIPBLAS_STATUS_SUCCESS &&
                hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws_bytes, sizeof(ws_bytes)) ==
                    HIPBLAS_STATUS_SUCCESS &&
                hipblasLtMatmulAlgoGetHeuristic(handle(), p.desc, p.la, p.lb, p.ld, p.ld, pref, 1, heur, &got) == HIPBLAS_STATUS_SUCCESS &&
                got > 0) {
                p.algo = heur[0].algo, p.workspace = heur[0].workspaceSize, p.ok = true;
            }
            if (pref) hipblasLtMatmulPreferenceDestroy(pref);
        }
    }
    auto &slot = g_plans[key] = p;
    return slot.ok && slot.workspace <= ws_bytes ? &slot : nullptr;
}


// ---------------------------------------------------------------------------------------------------------------
// Hand-written form of the same product (algo_index = -2): the K-concatenated operands a = [h2 | h1 | h1], w = [g1 | g2 | g1]
// hold each part once in their first two thirds, so the kernel stages h1, h2, g1, g2 once per K-step and issues the three
// partial products h2 g1 + h1 g2 + h1 g1 per fragment pair -- the library GEMM over K' = 3K reads h1 and g1 twice.
// Block = 256 threads = 2 x 2 waves, tile 128 x 128, K-step 32; a wave owns 64 x 64 = 2 x 2 v_mfma_f32_32x32x16_f16 tiles
// (64 accumulator registers).  LDS: [part][row][32 halfs] with the 16-byte chunk index XOR-ed by (row >> 2) & 3, so the
// sixteen lanes of a ds_read_b128 pass (sixteen consecutive rows, one chunk column) hit sixteen different bank quads;
// two stages of 32 KB; the next stage's 16-byte global loads are in flight under the current stage's 24 MFMAs per wave.
// Status (tools/bench_split_gemm.py, 6304 rows): correct to the library's own distance from fp64 (6e-7 of the largest
// output), 470-715 TFLOP/s issued against the tuned library kernels' 510-930 -- the matrix pipe waits for LDS fragment reads
// between small groups of MFMAs and for a barrier per 32 of K (matrix pipe 38 % busy by SQ_VALU_MFMA_BUSY_CYCLES, the library's
// kernels 48 %; all sixteen fragments of a K-step requested before its first MFMA: slower, 62 vs 54 us; the four
// accumulators in turn: no change).
// Not on the product path: a base for the fused forms (GEGLU gate in the epilogue, next operand split in the epilogue)
// that a library GEMM cannot express.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int kBM = 128, kBN = 128, kBK = 32;
constexpr int kStageBytes = 2 * (kBM + kBN) * kBK * 2;  // both parts of both operands: 32 KB

__device__ __forceinline__ int lds_chunk_offset(int row, int chunk) {  // bytes inside one part's [rows][32 halfs] block
    return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4);
}

__global__ __launch_bounds__(256, 2) void split_gemm_kernel(long long M, int N, int K, const _Float16 *__restrict__ A,
                                                            const _Float16 *__restrict__ W, float alpha, float *__restrict__ out) {
    extern __shared__ __align__(16) unsigned char gemm_lds[];  // [stage][A h1 | A h2 | W g1 | W g2] each kBM (kBN) x 64 B
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // block -> tile: consecutive blocks walk down M inside one column of tiles, so the 128 weight rows of the column
    // stay in L2 while the activations stream
    const int tiles_m = (int)((M + kBM - 1) / kBM);
    const int bm = blockIdx.x % tiles_m, bn = blockIdx.x / tiles_m;
    const long long m0 = (long long)bm * kBM;
    const int n0 = bn * kBN;
    const size_t lda = (size_t)3 * K;  // halfs per row of both operands
    // staging: 512 chunks of 16 B per part and operand = 2 per thread; thread t handles rows (t >> 2) and (t >> 2) + 64,
    // chunk t & 3
    const int srow = tid >> 2, schunk = tid & 3;
    const _Float16 *a_src[2], *w_src[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const long long r = min(m0 + srow + 64 * i, M - 1);  // rows past M: a valid row is read, its results never stored
        a_src[i] = A + (size_t)r * lda + schunk * 8;
        w_src[i] = W + (size_t)(n0 + srow + 64 * i) * lda + schunk * 8;
    }
    // Two register sets of staged chunks: the global loads of K-step s + 2 are issued at the top of step s and stored to
    // LDS at the end of step s + 1 (one step of MFMAs, ~0.6 us, is shorter than a global round trip: with the loads one
    // step ahead the K = 2048 projection ran 74 us, with two 66).
    u32x4 st0[8], st1[8];  // [operand part][i]: A h2 (cols 0..K), A h1 (K..2K), W g1 (0..K), W g2 (K..2K)
    const int nsteps = K / kBK;
    auto gload = [&](u32x4 (&st)[8], int step) {
        const int k0 = min(step, nsteps - 1) * kBK;  // past the end: the last step again (never stored)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            st[0 + i] = *reinterpret_cast<const u32x4 *>(a_src[i] + k0);
            st[2 + i] = *reinterpret_cast<const u32x4 *>(a_src[i] + K + k0);
            st[4 + i] = *reinterpret_cast<const u32x4 *>(w_src[i] + k0);
            st[6 + i] = *reinterpret_cast<const u32x4 *>(w_src[i] + K + k0);
        }
    };
    auto lstore = [&](const u32x4 (&st)[8], int stage) {
        unsigned char *base = gemm_lds + stage * kStageBytes;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int off = lds_chunk_offset(srow + 64 * i, schunk);
            *reinterpret_cast<u32x4 *>(base + 0 * kBM * 64 + off) = st[0 + i];                 // A h2
            *reinterpret_cast<u32x4 *>(base + 1 * kBM * 64 + off) = st[2 + i];                 // A h1
            *reinterpret_cast<u32x4 *>(base + 2 * kBM * 64 + 0 * kBN * 64 + off) = st[4 + i];  // W g1
            *reinterpret_cast<u32x4 *>(base + 2 * kBM * 64 + 1 * kBN * 64 + off) = st[6 + i];  // W g2
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int frow = lane & 31, fh = lane >> 5;  // fragment: row (column) inside the 32-tile, k half
    auto compute = [&](int stage) {
        const unsigned char *base = gemm_lds + stage * kStageBytes;
        const unsigned char *a2p = base, *a1p = base + kBM * 64, *b1p = base + 2 * kBM * 64, *b2p = b1p + kBN * 64;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f16x8 a1[2], a2[2], b1[2], b2[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ra = wm * 64 + i * 32 + frow, rb = wn * 64 + i * 32 + frow;
                const int oa = lds_chunk_offset(ra, ks * 2 + fh), ob = lds_chunk_offset(rb, ks * 2 + fh);
                a2[i] = *reinterpret_cast<const f16x8 *>(a2p + oa);
                a1[i] = *reinterpret_cast<const f16x8 *>(a1p + oa);
                b1[i] = *reinterpret_cast<const f16x8 *>(b1p + ob);
                b2[i] = *reinterpret_cast<const f16x8 *>(b2p + ob);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2[i], b1[j], acc[i][j], 0, 0, 0);  // small terms first
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[i], b2[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[i], b1[j], acc[i][j], 0, 0, 0);
                }
        }
    };
    gload(st0, 0);
    gload(st1, 1);
    lstore(st0, 0);
    __syncthreads();
    for (int s = 0; s < nsteps; s += 2) {
        gload(st0, s + 2);
        compute(0);
        if (s + 1 < nsteps) lstore(st1, 1);  // stage 1 was last read in step s - 1, behind the barrier that ended it
        __syncthreads();
        if (s + 1 >= nsteps) break;
        gload(st1, s + 3);
        compute(1);
        if (s + 2 < nsteps) lstore(st0, 0);
        __syncthreads();
    }
    // accumulator register r of lane (frow, fh) of tile (i, j): row 8 (r >> 2) + 4 fh + (r & 3), column frow
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long m = m0 + wm * 64 + i * 32 + 8 * (r >> 2) + 4 * fh + (r & 3);
                if (m < M) out[(size_t)m * N + n0 + wn * 64 + j * 32 + frow] = alpha * acc[i][j][r];
            }
}

static bool split_gemm_supported(long long rows, int n, int k3) {
    return k3 % 3 == 0 && (k3 / 3) % kBK == 0 && n % kBN == 0 && rows > 0 && (rows + kBM - 1) / kBM * (long long)(n / kBN) < (1ll << 31);
}

}  // namespace gemm
}  // namespace amav

using namespace amav;

static int check_gemm_args(const char *who, int64_t rows, int n, int k3, const void *a, const void *w, const float *out) {
    AMAV_REQUIRE(rows > 0 && rows < (1ll << 31) && n > 0 && k3 > 0 && k3 % 8 == 0, "%s: bad sizes rows=%lld n=%d k3=%d", who,
                 (long long)rows, n, k3);
    AMAV_REQUIRE(a && w && out, "%s: NULL pointer", who);
    AMAV_REQUIRE(((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(out)) & 15) == 0,
                 "%s: operands must be 16-byte aligned", who);
    return AMAV_OK;
}

extern "C" int amav_gemm_split_fp16(int64_t rows, int n, int k3, const void *a, const void *w, float alpha, float *out,
                                    int algo_index, void *workspace, size_t workspace_bytes, void *stream) {
    if (int rc = check_gemm_args("amav_gemm_split_fp16", rows, n, k3, a, w, out)) return rc;
    if (algo_index == -2) {  // the hand-written kernel
        AMAV_REQUIRE(gemm::split_gemm_supported(rows, n, k3), "amav_gemm_split_fp16: the hand-written kernel needs n %% 128 == 0 and "
                     "k3 = 3 K with K %% 32 == 0 (rows=%lld n=%d k3=%d)", (long long)rows, n, k3);
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm::split_gemm_kernel),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 2 * gemm::kStageBytes);
        if (attr != hipSuccess) return fail(AMAV_ERR_LAUNCH, "amav_gemm_split_fp16: cannot raise the dynamic LDS limit");
        const unsigned blocks = (unsigned)((rows + gemm::kBM - 1) / gemm::kBM * (n / gemm::kBN));
        gemm::split_gemm_kernel<<<blocks, 256, 2 * gemm::kStageBytes, static_cast<hipStream_t>(stream)>>>(
            rows, n, k3 / 3, static_cast<const _Float16 *>(a), static_cast<const _Float16 *>(w), alpha, out);
        return check_launch("amav_gemm_split_fp16");
    }
    const gemm::Plan *p = gemm::plan_for(rows, n, k3, algo_index, workspace ? workspace_bytes : 0);
    if (!p) return fail(AMAV_ERR_LAUNCH, "amav_gemm_split_fp16: hipBLASLt has no kernel for rows=%lld n=%d k=%d within %zu B of workspace",
                        (long long)rows, n, k3, workspace_bytes);
    const float beta = 0.f;
    hipblasLtMatmulAlgo_t algo = p->algo;
    const hipblasStatus_t st = hipblasLtMatmul(gemm::handle(), p->desc, &alpha, w, p->la, a, p->lb, &beta, out, p->ld, out, p->ld, &algo,
                                               workspace, workspace_bytes, static_cast<hipStream_t>(stream));
    if (st != HIPBLAS_STATUS_SUCCESS) return fail(AMAV_ERR_LAUNCH, "amav_gemm_split_fp16: hipblasLtMatmul failed (%d)", (int)st);
    return AMAV_OK;
}

extern "C" int amav_gemm_split_fp16_tune(int64_t rows, int n, int k3, const void *a, const void *w, float *out, void *workspace,
                                         size_t workspace_bytes, int repeats, int copies, int32_t *best_index, float *best_ms,
                                         float *heuristic_ms, void *stream_) {
    if (int rc = check_gemm_args("amav_gemm_split_fp16_tune", rows, n, k3, a, w, out)) return rc;
    AMAV_REQUIRE(best_index && best_ms && heuristic_ms && repeats > 0 && copies > 0, "amav_gemm_split_fp16_tune: NULL result pointer");
    // `copies` > 1: a, w and out are `copies` consecutive operand sets and run i uses set i % copies, so that a kernel is
    // timed as it runs inside the transformer step (weights and activations come from MALL / HBM, not from a hot L2)
    const size_t a_step = (size_t)rows * k3 * 2, w_step = (size_t)n * k3 * 2, o_step = (size_t)rows * n * 4;
    auto A = [&](int i) { return static_cast<const char *>(a) + (size_t)(i % copies) * a_step; };
    auto W = [&](int i) { return static_cast<const char *>(w) + (size_t)(i % copies) * w_step; };
    auto O = [&](int i) { return reinterpret_cast<float *>(reinterpret_cast<char *>(out) + (size_t)(i % copies) * o_step); };
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    gemm::Plan p;
    AMAV_REQUIRE(gemm::handle() && gemm::make_layouts(p, rows, n, k3), "amav_gemm_split_fp16_tune: hipBLASLt set-up failed");
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return fail(AMAV_ERR_LAUNCH, "amav_gemm_split_fp16_tune: events");
    const float one = 1.f, zero = 0.f;
    auto time_algo = [&](hipblasLtMatmulAlgo_t algo, float *ms) {
        size_t need = 0;
        if (hipblaslt_ext::matmulIsAlgoSupported(gemm::handle(), p.desc, &one, p.la, p.lb, &zero, p.ld, p.ld, algo, need) != HIPBLAS_STATUS_SUCCESS ||
            need > workspace_bytes)
            return false;
        for (int i = 0; i < 2; ++i)
            if (hipblasLtMatmul(gemm::handle(), p.desc, &one, W(i), p.la, A(i), p.lb, &zero, O(i), p.ld, O(i), p.ld, &algo, workspace,
                                workspace_bytes, stream) != HIPBLAS_STATUS_SUCCESS)
                return false;
        (void)hipEventRecord(e0, stream);
        for (int i = 0; i < repeats; ++i)
            (void)hipblasLtMatmul(gemm::handle(), p.desc, &one, W(i), p.la, A(i), p.lb, &zero, O(i), p.ld, O(i), p.ld, &algo, workspace,
                                  workspace_bytes, stream);
        (void)hipEventRecord(e1, stream);
        if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(ms, e0, e1) != hipSuccess) return false;
        *ms /= (float)repeats;
        return true;
    };
    *best_index = -1, *best_ms = 1e30f, *heuristic_ms = -1.f;
    const gemm::Plan *h = gemm::plan_for(rows, n, k3, -1, workspace_bytes);
    if (h) (void)time_algo(h->algo, heuristic_ms);
    std::vector<hipblasLtMatmulHeuristicResult_t> all;
    if (hipblaslt_ext::getAllAlgos(gemm::handle(), hipblaslt_ext::GemmType::HIPBLASLT_GEMM, HIPBLAS_OP_T, HIPBLAS_OP_N, HIP_R_16F, HIP_R_16F,
                                   HIP_R_32F, HIP_R_32F, HIPBLAS_COMPUTE_32F, all) != HIPBLAS_STATUS_SUCCESS)
        return fail(AMAV_ERR_LAUNCH, "amav_gemm_split_fp16_tune: getAllAlgos failed");
    for (auto &r : all) {
        float ms;
        if (time_algo(r.algo, &ms) && ms < *best_ms) *best_ms = ms, *best_index = hipblaslt_ext::getIndexFromAlgo(r.algo);
    }
    (void)hipEventDestroy(e0), (void)hipEventDestroy(e1);
    return *best_index >= 0 ? AMAV_OK : fail(AMAV_ERR_LAUNCH, "amav_gemm_split_fp16_tune: no algorithm ran");
}

extern "C" const char *amav_gemm_library_version(void) {
    static char buf[64] = {0};
    if (!buf[0]) {
        int v = 0;
        if (gemm::handle() && hipblasLtGetVersion(gemm::handle(), &v) == HIPBLAS_STATUS_SUCCESS) snprintf(buf, sizeof(buf), "hipblaslt-%d", v);
        else snprintf(buf, sizeof(buf), "hipblaslt-unknown");
    }
    return buf;
}
