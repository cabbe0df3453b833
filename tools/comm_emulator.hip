// Diagnostic (not product code): a stand-in for a collective's kernel on one GPU -- `blocks` workgroups of 256 threads
// with a register footprint like RCCL's (~100 VGPRs) that stay resident for `microseconds` doing next to nothing
// (a transfer over xGMI keeps its channels' workgroups resident but issues little).  Launched on a side stream next to
// the render step, it shows whether such a kernel can START while the persistent blend kernel owns the wave slots
// (tools/bench_comm_overlap.py).   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o libcomm_emulator.so comm_emulator.hip
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(256) void resident_kernel(unsigned long long ticks, float *sink) {
    float r[96];
#pragma unroll
    for (int i = 0; i < 96; ++i) r[i] = threadIdx.x + i;
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {
#pragma unroll
        for (int i = 0; i < 96; ++i) r[i] = r[i] * 1.0001f + 0.5f;  // keeps the registers live
        __builtin_amdgcn_s_sleep(64);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 96; ++i) s += r[i];
    if (s == 12345.678f) sink[0] = s;
}

extern "C" int comm_emulate(int blocks, int microseconds, void *sink, void *stream) {
    // wall_clock64 ticks at 100 MHz on gfx9
    resident_kernel<<<blocks, 256, 0, static_cast<hipStream_t>(stream)>>>(100ULL * microseconds, static_cast<float *>(sink));
    return (int)hipGetLastError();
}
