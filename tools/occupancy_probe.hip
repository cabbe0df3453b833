// Diagnostic micro-benchmark: how many 256-thread blocks stay resident per CU as a function of LDS per block, and
// how fast blocks that exit immediately are replaced.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(256) void probe(unsigned long long *stamps, int heavy_every, int spin_us) {
    extern __shared__ int lds[];
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const bool heavy = (blockIdx.x % heavy_every) == 0;
    if (heavy) {
        if (threadIdx.x == 0) lds[0] = 1;
        while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_us * 100) __builtin_amdgcn_s_sleep(8);
    }
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = t0;
        stamps[2 * blockIdx.x + 1] = heavy ? __builtin_amdgcn_s_memrealtime() : 0;
    }
}

int main() {
    const int nblocks = 65536;
    unsigned long long *d;
    hipMalloc(&d, nblocks * 16);
    std::vector<unsigned long long> h(nblocks * 2);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&probe), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int heavy_every : {1, 4}) {
        for (int lds_kb : {0, 8, 16, 28, 32, 48, 64}) {
            hipMemset(d, 0, nblocks * 16);
            hipEvent_t a, b;
            hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a);
            probe<<<nblocks, 256, lds_kb * 1024>>>(d, heavy_every, 40);
            hipEventRecord(b);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, a, b);
            hipMemcpy(h.data(), d, nblocks * 16, hipMemcpyDeviceToHost);
            // peak concurrency of heavy blocks
            std::vector<std::pair<unsigned long long, int>> ev;
            for (int i = 0; i < nblocks; ++i) if (h[2 * i + 1]) { ev.push_back({h[2 * i], 1}); ev.push_back({h[2 * i + 1], -1}); }
            std::sort(ev.begin(), ev.end());
            int cur = 0, peak = 0; for (auto &e : ev) { cur += e.second; peak = std::max(peak, cur); }
            printf("heavy 1/%d  lds %2d KiB: %.3f ms, peak resident heavy blocks %d (%.2f per CU)\n", heavy_every, lds_kb, ms, peak, peak / 256.0);
        }
    }
    return 0;
}
