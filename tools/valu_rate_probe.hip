// Diagnostic (not product code): VALU issue rate of one SIMD on gfx950 as a function of resident waves and instruction
// kind, measured in shader cycles with s_memtime.  Answers "how many cycles does a wave64 v_fma_f32 / v_pk_fma_f32 /
// v_exp_f32 / a branchy mix occupy a SIMD for" -- the unit behind the blend kernel's VALU ceiling.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate_probe valu_rate_probe.hip && ./valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(64) void probe(float *out, unsigned long long *cyc, int iters, int mask) {
    const int lane = threadIdx.x;
    float a0 = lane, a1 = lane + 1, a2 = lane + 2, a3 = lane + 3, a4 = lane * 0.5f, a5 = 1.f, a6 = 2.f, a7 = 3.f;
    f32x2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float m = 1.0001f, c = 0.0001f;
    const f32x2 m2 = {m, m}, c2 = {c, c};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#define FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(c))
#define PKFMA(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m2), "v"(c2))
#define EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
        if (MODE == 0) {  // 8 independent v_fma_f32
            FMA(a0); FMA(a1); FMA(a2); FMA(a3); FMA(a4); FMA(a5); FMA(a6); FMA(a7);
        } else if (MODE == 1) {  // 8 independent v_pk_fma_f32
            PKFMA(p0); PKFMA(p1); PKFMA(p2); PKFMA(p3); PKFMA(p4); PKFMA(p5); PKFMA(p6); PKFMA(p7);
        } else if (MODE == 2) {  // 8 independent v_exp_f32
            EXP(a0); EXP(a1); EXP(a2); EXP(a3); EXP(a4); EXP(a5); EXP(a6); EXP(a7);
        } else if (MODE == 3) {  // one DEPENDENT chain of 8 v_fma_f32
            FMA(a0); FMA(a0); FMA(a0); FMA(a0); FMA(a0); FMA(a0); FMA(a0); FMA(a0);
        } else if (MODE == 4) {  // 8 v_fma_f32 interleaved with 8 SALU instructions
            int s = i;
#define SALU(x) asm volatile("s_add_i32 %0, %0, 3" : "+s"(x) : : "scc")
            FMA(a0); SALU(s); FMA(a1); SALU(s); FMA(a2); SALU(s); FMA(a3); SALU(s);
            FMA(a4); SALU(s); FMA(a5); SALU(s); FMA(a6); SALU(s); FMA(a7); SALU(s);
            asm volatile("" :: "s"(s));
        } else if (MODE == 5) {  // 8 v_fma_f32 behind 4 wave-uniform, always-taken-in-body branches (blend loop shape)
            const int s = __builtin_amdgcn_readfirstlane(i) | mask;  // mask = 15 at run time
            if (s & 1) { FMA(a0); FMA(a1); }
            if (s & 2) { FMA(a2); FMA(a3); }
            if (s & 4) { FMA(a4); FMA(a5); }
            if (s & 8) { FMA(a6); FMA(a7); }
        } else if (MODE == 6) {  // 4 v_cmp (to VCC) + 4 v_cndmask
#define CMPSEL(x, y, z) asm volatile("v_cmp_gt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(y), "v"(z) : "vcc")
            CMPSEL(a0, a1, a2); CMPSEL(a3, a4, a5); CMPSEL(a6, a7, a0); CMPSEL(a1, a2, a3);
        } else if (MODE == 7) {  // 6 v_fma_f32 + 2 broadcast ds_read_b128 per iteration (the blend loop's record reads)
            __shared__ float4 lds[64];
            if (i == 0) lds[lane] = make_float4(a0, a1, a2, a3);
            const float4 r0 = lds[i & 63], r1 = lds[(i + 7) & 63];
            FMA(a0); FMA(a1); FMA(a2); FMA(a3); FMA(a4); FMA(a5);
            a6 += (r0.x + r0.y) + (r0.z + r0.w) + (r1.x + r1.y) + (r1.z + r1.w);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 64 + lane] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}

template <int MODE>
static void run(const char *name, int per_iter, float *d, unsigned long long *dc) {
    const int iters = 20000;
    for (int wps : {1, 2, 3, 4, 5, 8}) {
        const int blocks = 256 * 4 * wps;  // one 64-thread block per wave slot: wps waves per SIMD when spread evenly
        probe<MODE><<<blocks, 64>>>(d, dc, 100, 15);
        (void)hipDeviceSynchronize();
        hipEvent_t a, b;
        (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        (void)hipEventRecord(a);
        probe<MODE><<<blocks, 64>>>(d, dc, iters, 15);
        (void)hipEventRecord(b);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        std::vector<unsigned long long> h(blocks);
        (void)hipMemcpy(h.data(), dc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double sum = 0; for (auto v : h) sum += (double)v;
        const double wave_cyc = sum / blocks;  // s_memtime ticks = 100 MHz constant clock on gfx9?  report both
        printf("%-28s waves/SIMD %d: kernel %.3f ms; wave ticks/iter %.2f; SIMD-time per wave-instruction %.3f ns "
               "(= %.2f cycles at 2.4 GHz)\n", name, wps, ms, wave_cyc / iters,
               ms * 1e6 / ((double)iters * per_iter * wps), ms * 1e-3 * 2.4e9 / ((double)iters * per_iter * wps));
        fflush(stdout);
    }
}

int main() {
    float *d; unsigned long long *dc;
    (void)hipMalloc(&d, 1 << 24); (void)hipMalloc(&dc, 1 << 20);
    run<0>("8 x v_fma_f32 (indep)", 8, d, dc);
    run<1>("8 x v_pk_fma_f32 (indep)", 8, d, dc);
    run<2>("8 x v_exp_f32 (indep)", 8, d, dc);
    run<3>("8 x v_fma_f32 (one chain)", 8, d, dc);
    run<4>("8 x v_fma_f32 + 8 SALU", 8, d, dc);
    run<5>("8 v_fma behind 4 branches", 8, d, dc);
    run<6>("4 v_cmp + 4 v_cndmask", 8, d, dc);
    run<7>("6 v_fma + 8 v_add + 2 LDS reads", 14, d, dc);
    return 0;
}
