// Diagnostic: every hipBLASLt algorithm for the four fp16 x 2 split GEMMs of a transformer block (fp16 in, fp32 out,
// fp32 accumulate; M = 6304 rows), timed against the library's own first heuristic choice.
//   hipcc --offload-arch=gfx950 -O2 tools/hipblaslt_probe.cpp -lhipblaslt -o /tmp/hipblaslt_probe && /tmp/hipblaslt_probe
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt-ext.hpp>
#include <hipblaslt/hipblaslt.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x) do { auto e_ = (x); if (e_ != 0) { printf("error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 1; } } while (0)

int main() {
    hipblasLtHandle_t h;
    CK(hipblasLtCreate(&h));
    hipStream_t stream;
    CK(hipStreamCreate(&stream));
    const int M = 6304;
    const int shapes[4][2] = {{1536, 1536}, {1536, 512}, {1536, 4096}, {6144, 512}};  // {K' = 3K, N}
    const char *names[4] = {"qkv 512->1536", "to_out 512->512", "ff_in 512->4096", "ff_out 2048->512"};
    size_t ws_bytes = 64u << 20;
    void *ws, *A, *B, *D;
    CK(hipMalloc(&ws, ws_bytes));
    CK(hipMalloc(&A, (size_t)M * 6144 * 2));
    CK(hipMalloc(&B, (size_t)4096 * 6144 * 2));
    CK(hipMalloc(&D, (size_t)M * 4096 * 4));
    CK(hipMemset(A, 0, (size_t)M * 6144 * 2));
    CK(hipMemset(B, 0, (size_t)4096 * 6144 * 2));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int s = 0; s < 4; ++s) {
        const int K = shapes[s][0], N = shapes[s][1];
        hipblasLtMatmulDesc_t desc;
        CK(hipblasLtMatmulDescCreate(&desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
        hipblasOperation_t opT = HIPBLAS_OP_T, opN = HIPBLAS_OP_N;
        CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opT, sizeof(opT)));
        CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opN, sizeof(opN)));
        hipblasLtMatrixLayout_t la, lb, ld;
        CK(hipblasLtMatrixLayoutCreate(&la, HIP_R_16F, K, N, K));  // weights [N, K] row-major = K x N column-major
        CK(hipblasLtMatrixLayoutCreate(&lb, HIP_R_16F, K, M, K));  // activations [M, K] row-major
        CK(hipblasLtMatrixLayoutCreate(&ld, HIP_R_32F, N, M, N));  // out [M, N] row-major
        float alpha = 1.f, beta = 0.f;
        auto time_algo = [&](hipblasLtMatmulAlgo_t &algo, float *ms) -> int {
            size_t need = 0;
            if (hipblaslt_ext::matmulIsAlgoSupported(h, desc, &alpha, la, lb, &beta, ld, ld, algo, need) != HIPBLAS_STATUS_SUCCESS || need > ws_bytes)
                return 1;
            for (int i = 0; i < 2; ++i)
                if (hipblasLtMatmul(h, desc, &alpha, B, la, A, lb, &beta, D, ld, D, ld, &algo, ws, ws_bytes, stream) != HIPBLAS_STATUS_SUCCESS) return 1;
            hipEventRecord(e0, stream);
            for (int i = 0; i < 10; ++i) hipblasLtMatmul(h, desc, &alpha, B, la, A, lb, &beta, D, ld, D, ld, &algo, ws, ws_bytes, stream);
            hipEventRecord(e1, stream);
            hipEventSynchronize(e1);
            hipEventElapsedTime(ms, e0, e1);
            *ms /= 10;
            return 0;
        };
        // the library's own choice
        hipblasLtMatmulPreference_t pref;
        CK(hipblasLtMatmulPreferenceCreate(&pref));
        CK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws_bytes, sizeof(ws_bytes)));
        hipblasLtMatmulHeuristicResult_t heur[8];
        int got = 0;
        CK(hipblasLtMatmulAlgoGetHeuristic(h, desc, la, lb, ld, ld, pref, 8, heur, &got));
        float t_default = -1.f;
        if (got > 0) time_algo(heur[0].algo, &t_default);
        std::vector<hipblasLtMatmulHeuristicResult_t> all;
        CK(hipblaslt_ext::getAllAlgos(h, hipblaslt_ext::GemmType::HIPBLASLT_GEMM, opT, opN, HIP_R_16F, HIP_R_16F, HIP_R_32F, HIP_R_32F,
                                      HIPBLAS_COMPUTE_32F, all));
        std::vector<std::pair<float, int>> res;
        for (auto &r : all) {
            float ms;
            if (time_algo(r.algo, &ms) == 0) res.push_back({ms, hipblaslt_ext::getIndexFromAlgo(r.algo)});
        }
        std::sort(res.begin(), res.end());
        const double flop = 2.0 * M * (double)K * N;
        printf("%-18s K'=%d N=%d: default (index %d) %.4f ms = %.0f TF/s issued; %zu of %zu algorithms run; best:", names[s], K, N,
               got ? hipblaslt_ext::getIndexFromAlgo(heur[0].algo) : -1, t_default, flop / t_default / 1e9, res.size(), all.size());
        for (size_t i = 0; i < std::min<size_t>(4, res.size()); ++i) printf("  [%d] %.4f ms (%.0f TF/s)", res[i].second, res[i].first, flop / res[i].first / 1e9);
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
