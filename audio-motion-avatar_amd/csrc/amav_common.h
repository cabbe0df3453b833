// Shared host-side helpers of libamav_hip.so (error reporting, launch checks, workspace carving).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/amav.h"

namespace amav {

constexpr int kWave = 64;  // gfx950 wavefront width

char *error_buffer();  // thread-local, 512 bytes (api.hip)

inline int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

#define AMAV_REQUIRE(cond, ...)                                         \
    do {                                                                \
        if (!(cond)) return ::amav::fail(AMAV_ERR_INVALID_ARG, __VA_ARGS__); \
    } while (0)

// Check the launch that was just enqueued (no device sync: only launch-time errors).
inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(AMAV_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return AMAV_OK;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Tile-sparse wire format of the frame exchange (frames.hip has the description): shared with the rasterizer, which can
// emit it directly (amav_raster_args.wire)
constexpr int kWireHeaderInts = 16;
constexpr int kWireMagic = 0x414d4156;  // "AMAV"
constexpr int kWireTileBytes = 16 * 16 * 3;
inline size_t wire_payload_at(int F, int tiles) { return ((size_t)(kWireHeaderInts + F + tiles) * 4 + 15) / 16 * 16; }

// amav_set_option overrides (api.hip): -1 = follow the environment variable
int option_attn();  // 0: fp32 MFMA, 1: bf16 x 3, 2: fp16 x 2
int option_lbs();   // 0: fp32 MFMA, 1: split

// Zero `bytes` on `stream` with an ordinary kernel launch (api.hip).  Used instead of hipMemsetAsync so that a captured
// step holds kernel nodes only (HIP-graph note).
hipError_t zero_async(void *ptr, size_t bytes, hipStream_t stream);

// Bump allocator over the caller's workspace; with base == nullptr it only measures.
struct Carver {
    char *base;
    size_t off = 0;
    explicit Carver(void *b) : base(static_cast<char *>(b)) {}
    template <typename T>
    T *take(size_t count) {
        off = align_up(off, 256);
        T *p = base ? reinterpret_cast<T *>(base + off) : nullptr;
        off += count * sizeof(T);
        return p;
    }
    size_t total() const { return align_up(off, 256); }
};

}  // namespace amav
