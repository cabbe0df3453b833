// Library-level entry points of libamav_hip.so (version, errors, device probe) and the camera kernel.
#include <atomic>
#include <cstdlib>
#include <cstring>

#include "amav_common.h"

namespace amav {

char *error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}

static std::atomic<int> g_option_attn{-1}, g_option_lbs{-1};
int option_attn() { return g_option_attn.load(std::memory_order_relaxed); }
int option_lbs() { return g_option_lbs.load(std::memory_order_relaxed); }

// 0 = zero-fill kernel (default), 1 = hipMemsetAsync (AMAV_CLEAR=memset, diagnostic only)
static int clear_mode() {
    static const int mode = [] {
        const char *e = getenv("AMAV_CLEAR");
        return (e && e[0] == 'm') ? 1 : 0;
    }();
    return mode;
}

__global__ __launch_bounds__(256) void zero_words_kernel(unsigned *__restrict__ dst, size_t words) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t quads = words >> 2;
    if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
        if (i < quads) reinterpret_cast<uint4 *>(dst)[i] = make_uint4(0u, 0u, 0u, 0u);
        if (i < (words & 3)) dst[quads * 4 + i] = 0u;
    } else {
        for (size_t k = i * 4; k < min(words, i * 4 + 4); ++k) dst[k] = 0u;
    }
}

hipError_t zero_async(void *ptr, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return hipSuccess;
    if (clear_mode() == 1 || (bytes & 3) || (reinterpret_cast<uintptr_t>(ptr) & 3))
        return hipMemsetAsync(ptr, 0, bytes, stream);
    const size_t words = bytes / 4, threads = (words + 3) / 4;
    zero_words_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, stream>>>(static_cast<unsigned *>(ptr), words);
    return hipGetLastError();
}

// Sixteen threads per frame, one per matrix element (one thread per frame walked ~40 dependent scalar loads: 17 us for
// 250 frames).  Replaces src/models/renderer.py:486-510 + src/utils/graphic_utils.py:67-78,103-145.
// The reference inverts [R^T|t] twice (a numerical identity); here the view matrix is E.
__global__ void camera_kernel(int F, const float *K, const float *E, float h, float w, float znear, float zfar,
                              float *view, float *proj, float *tanfov, float *campos) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int f = tid >> 4, r = (tid >> 2) & 3, c = tid & 3;
    if (f >= F) return;
    const float *k = K + f * 9;
    const float *e = E + f * 16;
    const float fx = k[0], fy = k[4], px = k[2], py = k[5];
    // row r of the NDC matrix {{2fx/w, 0, (2px-w)/w, 0}, {0, 2fy/h, (2py-h)/h, 0}, {0, 0, zf/(zf-zn), -zf zn/(zf-zn)}, {0, 0, 1, 0}}
    float n[4] = {0.f, 0.f, 0.f, 0.f};
    if (r == 0) n[0] = 2.f * fx / w, n[2] = (2.f * px - w) / w;
    if (r == 1) n[1] = 2.f * fy / h, n[2] = (2.f * py - h) / h;
    if (r == 2) n[2] = zfar / (zfar - znear), n[3] = -zfar * znear / (zfar - znear);
    if (r == 3) n[2] = 1.f;
    view[f * 16 + c * 4 + r] = e[r * 4 + c];
    float acc = 0.f;
    for (int m = 0; m < 4; ++m) acc += n[m] * e[m * 4 + c];
    proj[f * 16 + c * 4 + r] = acc;
    if ((tid & 15) != 0) return;
    tanfov[2 * f] = w / (2.f * fx);
    tanfov[2 * f + 1] = h / (2.f * fy);
    if (campos) {
        // camera centre = -A^-1 t for E = [A | t]
        const float a = e[0], b = e[1], c = e[2], d = e[4], g = e[5], hh = e[6], i = e[8], j = e[9], l = e[10];
        const float c00 = g * l - hh * j, c01 = hh * i - d * l, c02 = d * j - g * i;
        const float det = a * c00 + b * c01 + c * c02;
        const float inv = 1.f / det;
        const float t0 = e[3], t1 = e[7], t2 = e[11];
        const float i00 = c00 * inv, i01 = (c * j - b * l) * inv, i02 = (b * hh - c * g) * inv;
        const float i10 = c01 * inv, i11 = (a * l - c * i) * inv, i12 = (c * d - a * hh) * inv;
        const float i20 = c02 * inv, i21 = (b * i - a * j) * inv, i22 = (a * g - b * d) * inv;
        campos[3 * f] = -(i00 * t0 + i01 * t1 + i02 * t2);
        campos[3 * f + 1] = -(i10 * t0 + i11 * t1 + i12 * t2);
        campos[3 * f + 2] = -(i20 * t0 + i21 * t1 + i22 * t2);
    }
}

}  // namespace amav

using namespace amav;

extern "C" const char *amav_version(void) { return "amav-hip 0.1.0 (gfx950)"; }

extern "C" const char *amav_last_error(void) { return error_buffer(); }

extern "C" int amav_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(AMAV_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

extern "C" int amav_camera_from_intrinsics(int F, const float *K, const float *E, int height, int width, float znear,
                                           float zfar, float *view, float *proj, float *tanfov, float *campos,
                                           void *stream) {
    AMAV_REQUIRE(F > 0 && height > 0 && width > 0, "amav_camera_from_intrinsics: bad sizes");
    AMAV_REQUIRE(K && E && view && proj && tanfov, "amav_camera_from_intrinsics: NULL pointer");
    camera_kernel<<<(F * 16 + 255) / 256, 256, 0, static_cast<hipStream_t>(stream)>>>(F, K, E, (float)height, (float)width,
                                                                                     znear, zfar, view, proj, tanfov, campos);
    return check_launch("amav_camera_from_intrinsics");
}

extern "C" int amav_event_create(void **event) {
    AMAV_REQUIRE(event != nullptr, "amav_event_create: NULL");
    hipEvent_t e;
    hipError_t rc = hipEventCreate(&e);
    if (rc != hipSuccess) return fail(AMAV_ERR_LAUNCH, "hipEventCreate: %s", hipGetErrorString(rc));
    *event = e;
    return AMAV_OK;
}

extern "C" int amav_event_destroy(void *event) {
    if (event && hipEventDestroy(static_cast<hipEvent_t>(event)) != hipSuccess)
        return fail(AMAV_ERR_LAUNCH, "hipEventDestroy failed");
    return AMAV_OK;
}

extern "C" int amav_event_record(void *event, void *stream) {
    AMAV_REQUIRE(event != nullptr, "amav_event_record: NULL event");
    hipError_t rc = hipEventRecord(static_cast<hipEvent_t>(event), static_cast<hipStream_t>(stream));
    if (rc != hipSuccess) return fail(AMAV_ERR_LAUNCH, "hipEventRecord: %s", hipGetErrorString(rc));
    return AMAV_OK;
}

extern "C" int amav_event_elapsed_ms(void *start, void *stop, float *ms) {
    AMAV_REQUIRE(start && stop && ms, "amav_event_elapsed_ms: NULL");
    hipError_t rc = hipEventSynchronize(static_cast<hipEvent_t>(stop));
    if (rc == hipSuccess) rc = hipEventElapsedTime(ms, static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop));
    if (rc != hipSuccess) return fail(AMAV_ERR_LAUNCH, "hipEventElapsedTime: %s", hipGetErrorString(rc));
    return AMAV_OK;
}

extern "C" int amav_set_option(const char *name, const char *value) {
    AMAV_REQUIRE(name && value, "amav_set_option: NULL argument");
    const bool dflt = strcmp(value, "default") == 0;
    if (strcmp(name, "attn") == 0) {
        const int v = dflt ? -1 : strcmp(value, "f32") == 0 ? 0 : strcmp(value, "bf16") == 0 ? 1 : strcmp(value, "fp16") == 0 ? 2 : -2;
        AMAV_REQUIRE(v != -2, "amav_set_option: attn = \"%s\" (expected fp16, bf16, f32 or default)", value);
        amav::g_option_attn.store(v);
        return AMAV_OK;
    }
    if (strcmp(name, "lbs") == 0) {
        const int v = dflt ? -1 : strcmp(value, "f32") == 0 ? 0 : strcmp(value, "split") == 0 ? 1 : -2;
        AMAV_REQUIRE(v != -2, "amav_set_option: lbs = \"%s\" (expected split, f32 or default)", value);
        amav::g_option_lbs.store(v);
        return AMAV_OK;
    }
    return amav::fail(AMAV_ERR_INVALID_ARG, "amav_set_option: unknown option \"%s\"", name);
}
