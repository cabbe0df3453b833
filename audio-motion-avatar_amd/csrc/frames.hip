// Frame post-processing for the exchange step: fp32 RGBA -> uint8 RGB, the on-wire format of the all-gather.
// Quantisation is the reference's own (src/main2.py:351: `(frame * 255).astype(np.uint8)`, i.e. truncation).
#include "amav_common.h"

namespace amav {

// one thread per 4 pixels: reads 4 x 16 B (coalesced), writes 12 B
__global__ __launch_bounds__(256) void rgba_to_rgb8_kernel(size_t quads, const float4 *__restrict__ rgba,
                                                           uint3 *__restrict__ out) {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= quads) return;
    unsigned char b[12];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float4 p = rgba[q * 4 + k];
        b[k * 3 + 0] = (unsigned char)(fminf(fmaxf(p.x, 0.f), 1.f) * 255.0f);
        b[k * 3 + 1] = (unsigned char)(fminf(fmaxf(p.y, 0.f), 1.f) * 255.0f);
        b[k * 3 + 2] = (unsigned char)(fminf(fmaxf(p.z, 0.f), 1.f) * 255.0f);
    }
    uint3 w;
    w.x = b[0] | (b[1] << 8) | (b[2] << 16) | ((unsigned)b[3] << 24);
    w.y = b[4] | (b[5] << 8) | (b[6] << 16) | ((unsigned)b[7] << 24);
    w.z = b[8] | (b[9] << 8) | (b[10] << 16) | ((unsigned)b[11] << 24);
    out[q] = w;
}

}  // namespace amav

using namespace amav;

extern "C" int amav_frames_to_rgb8(int64_t num_pixels, const float *rgba, uint8_t *out_rgb8, void *stream) {
    AMAV_REQUIRE(num_pixels > 0 && num_pixels % 4 == 0, "amav_frames_to_rgb8: pixel count %lld not a multiple of 4",
                 (long long)num_pixels);
    AMAV_REQUIRE(rgba && out_rgb8, "amav_frames_to_rgb8: NULL pointer");
    AMAV_REQUIRE((reinterpret_cast<uintptr_t>(rgba) & 15) == 0 && (reinterpret_cast<uintptr_t>(out_rgb8) & 3) == 0,
                 "amav_frames_to_rgb8: misaligned buffer");
    const size_t quads = (size_t)num_pixels / 4;
    rgba_to_rgb8_kernel<<<(unsigned)((quads + 255) / 256), 256, 0, static_cast<hipStream_t>(stream)>>>(
        quads, reinterpret_cast<const float4 *>(rgba), reinterpret_cast<uint3 *>(out_rgb8));
    return check_launch("amav_frames_to_rgb8");
}
