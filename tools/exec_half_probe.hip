// Diagnostic: does a wave64 VALU instruction cost less when one 32-lane half of EXEC is empty?  (Not product code.)
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(64) void probe(float *out, int iters, int mode) {
    const int lane = threadIdx.x;
    // mode 0: all lanes; 1: lanes 0..31 only; 2: even lanes only (both halves half full); 3: lanes 0..15 only
    if (mode == 1 && lane >= 32) return;
    if (mode == 2 && (lane & 1)) return;
    if (mode == 3 && lane >= 16) return;
    float a0 = lane, a1 = lane + 1, a2 = lane + 2, a3 = lane + 3, a4 = lane * 0.5f, a5 = 1.f, a6 = 2.f, a7 = 3.f;
    const float m = 1.0001f, c = 0.0001f;
    for (int i = 0; i < iters; ++i) {
        a0 = fmaf(a0, m, c); a1 = fmaf(a1, m, c); a2 = fmaf(a2, m, c); a3 = fmaf(a3, m, c);
        a4 = fmaf(a4, m, c); a5 = fmaf(a5, m, c); a6 = fmaf(a6, m, c); a7 = fmaf(a7, m, c);
    }
    out[blockIdx.x * 64 + lane] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

int main() {
    float *d;
    (void)hipMalloc(&d, 1 << 24);
    const int blocks = 256 * 4 * 4, iters = 20000;  // 4 waves per SIMD
    for (int mode = 0; mode < 4; ++mode) {
        hipEvent_t a, b;
        (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        probe<<<blocks, 64>>>(d, 100, mode);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(a);
        probe<<<blocks, 64>>>(d, iters, mode);
        (void)hipEventRecord(b);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        const double instr = (double)blocks * iters * 8;
        printf("mode %d: %.3f ms  -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", mode, ms,
               ms * 1e-3 * 2.4e9 / (instr / 1024));
    }
    return 0;
}
