// Frame post-processing for the exchange step: fp32 RGBA -> uint8 RGB (dense, or as the tile-sparse wire format of
// the all-gather below) and back.  Quantisation is the reference's own (src/main2.py:351:
// `(frame * 255).astype(np.uint8)`, i.e. truncation).
#include <algorithm>
#include <cmath>

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

// ---------------------------------------------------------------------------------------------------------------
// Tile-sparse wire format of the exchange (lossless).  A rendered avatar frame is mostly background: only the 16x16
// tiles that differ from the background colour travel, so the all-gather moves ~1/5 of the bytes of dense uint8 RGB.
//   wire = header (16 int32: magic, count, cap, F, T, H, W, bg) | stored tiles per frame int32[F] | offsets int32[F*T]
//          (-1 = background tile, else the tile's slot in the payload) | payload [cap][16*16*3] uint8 (pack stores
//          the tiles in (frame, tile) order; readers go through `offsets`, so any assignment of slots is valid --
//          the rasterizer's direct emission hands out slots frame by frame in completion order)
// pack  : flags -> exclusive scan -> compaction (three launches, no host sync); tiles past `cap` are dropped and
//         `count` > `cap` tells every receiver (unpack raises its overflow flag; the caller re-packs with more room).
// unpack: every tile of every gathered buffer is written back into dense [frames, H, W, 3] uint8.
constexpr int kTileBytes = kWireTileBytes;  // header constants: amav_common.h

__device__ __forceinline__ unsigned char quant8(float v) { return (unsigned char)(fminf(fmaxf(v, 0.f), 1.f) * 255.0f); }

struct TileGeom {
    int f, x0, y, cq;  // frame, first pixel column of this lane's quad, pixel row, quad column inside the tile
};
__device__ __forceinline__ TileGeom tile_geom(int tile, int gx, int T, int lane) {
    TileGeom g;
    g.f = tile / T;
    const int t = tile - g.f * T;
    g.cq = lane & 3;
    g.x0 = (t % gx) * 16 + 4 * g.cq;
    g.y = (t / gx) * 16 + (lane >> 2);
    return g;
}

// this lane's 4 pixels as 12 bytes (3 words); pixels outside the image read as background
__device__ __forceinline__ uint3 quantise_quad(const float4 *__restrict__ rgba, const TileGeom &g, int H, int W,
                                               unsigned bgw) {
    unsigned char b[12];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int x = g.x0 + k;
        if (x < W && g.y < H) {
            const float4 p = rgba[((size_t)g.f * H + g.y) * W + x];
            b[k * 3] = quant8(p.x), b[k * 3 + 1] = quant8(p.y), b[k * 3 + 2] = quant8(p.z);
        } else {
            b[k * 3] = bgw & 255u, b[k * 3 + 1] = (bgw >> 8) & 255u, b[k * 3 + 2] = (bgw >> 16) & 255u;
        }
    }
    uint3 w;
    w.x = b[0] | (b[1] << 8) | (b[2] << 16) | ((unsigned)b[3] << 24);
    w.y = b[4] | (b[5] << 8) | (b[6] << 16) | ((unsigned)b[7] << 24);
    w.z = b[8] | (b[9] << 8) | (b[10] << 16) | ((unsigned)b[11] << 24);
    return w;
}

__device__ __forceinline__ uint3 background_quad(unsigned bgw) {
    const unsigned r = bgw & 255u, g = (bgw >> 8) & 255u, b = (bgw >> 16) & 255u;
    return make_uint3(r | (g << 8) | (b << 16) | (r << 24), g | (b << 8) | (r << 16) | (g << 24),
                      b | (r << 8) | (g << 16) | (b << 24));
}

// one wave per tile: flag = the tile differs from the background somewhere
__global__ __launch_bounds__(256) void tile_flags_kernel(int tiles, int gx, int T, int H, int W,
                                                         const float4 *__restrict__ rgba, unsigned bgw,
                                                         int *__restrict__ flags) {
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (tile >= tiles) return;
    const uint3 q = quantise_quad(rgba, tile_geom(tile, gx, T, lane), H, W, bgw);
    const uint3 bq = background_quad(bgw);
    const bool differs = (q.x != bq.x) | (q.y != bq.y) | (q.z != bq.z);
    const unsigned long long any = __ballot(differs);
    if (lane == 0) flags[tile] = any ? 1 : 0;
}

// flags from the caller's hint instead (e.g. the rasterizer's per-tile list lengths: a tile without Gaussians IS
// background), which saves reading the fp32 frames once
__global__ __launch_bounds__(256) void tile_hint_kernel(int tiles, const int *__restrict__ hint, int *__restrict__ flags) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < tiles) flags[i] = hint[i] != 0 ? 1 : 0;
}

// one block per frame: stored tiles of the frame
__global__ __launch_bounds__(256) void frame_count_kernel(int T, const int *__restrict__ flags,
                                                          int *__restrict__ frame_counts) {
    __shared__ int part[4];
    const int f = blockIdx.x;
    int c = 0;
    for (int t = threadIdx.x; t < T; t += 256) c += flags[(size_t)f * T + t];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) frame_counts[f] = part[0] + part[1] + part[2] + part[3];
}

// one block per frame: flags -> slots (exclusive scan over the whole clip; -1 for background tiles).  The block first
// sums the counts of the frames before its own, then scans its own T flags; the last block writes the clip total.
__global__ __launch_bounds__(1024) void tile_scan_kernel(int F, int T, int cap, int *__restrict__ header,
                                                         const int *__restrict__ frame_counts,
                                                         int *__restrict__ offsets) {
    __shared__ int wave_tot[16];
    __shared__ int carry;
    const int f = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int mine = 0;
    for (int g = threadIdx.x; g < f; g += 1024) mine += frame_counts[g];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if (lane == 0) wave_tot[wave] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        int b = 0;
        for (int w = 0; w < 16; ++w) b += wave_tot[w];
        carry = b;
    }
    __syncthreads();
    int *off = offsets + (size_t)f * T;
    for (int base = 0; base < T; base += 1024) {
        const int i = base + threadIdx.x;
        const int fl = i < T ? off[i] : 0;
        int incl = fl;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o, 64);
            if (lane >= o) incl += up;
        }
        __syncthreads();  // wave_tot / carry of the previous round have been read
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int before = carry;
        for (int w = 0; w < wave; ++w) before += wave_tot[w];
        if (i < T) off[i] = fl ? before + incl - fl : -1;
        __syncthreads();
        if (threadIdx.x == 1023) carry = before + incl;
    }
    __syncthreads();
    if (f == F - 1 && threadIdx.x == 0) header[1] = carry, header[2] = cap;
}

__global__ void wire_header_kernel(int *header, int F, int T, int H, int W, int bgw) {
    if (threadIdx.x == 0) header[0] = kWireMagic, header[3] = F, header[4] = T, header[5] = H, header[6] = W, header[7] = bgw;
}

__global__ __launch_bounds__(256) void tile_compact_kernel(int tiles, int gx, int T, int H, int W, int cap,
                                                           const float4 *__restrict__ rgba, unsigned bgw,
                                                           const int *__restrict__ offsets,
                                                           unsigned *__restrict__ payload) {
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (tile >= tiles) return;
    const int off = offsets[tile];
    if (off < 0 || off >= cap) return;
    const uint3 q = quantise_quad(rgba, tile_geom(tile, gx, T, lane), H, W, bgw);
    unsigned *dst = payload + (size_t)off * (kTileBytes / 4) + lane * 3;
    dst[0] = q.x, dst[1] = q.y, dst[2] = q.z;
}

// grid over (buffer, tile): dense frames out.  status[0] |= 1 when a sender dropped tiles (count > cap)
__global__ __launch_bounds__(256) void tile_unpack_kernel(int buffers, int tiles, int gx, int T, int H, int W, int cap,
                                                          const unsigned char *__restrict__ wire, size_t wire_stride,
                                                          unsigned char *__restrict__ out, int *__restrict__ status) {
    const long long gt = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (gt >= (long long)buffers * tiles) return;
    const int b = (int)(gt / tiles), tile = (int)(gt - (long long)b * tiles);
    const unsigned char *buf = wire + (size_t)b * wire_stride;
    const int *header = reinterpret_cast<const int *>(buf);
    const int F = tiles / T;
    const int *offsets = header + kWireHeaderInts + F;
    const size_t pay_at = ((size_t)(kWireHeaderInts + F + tiles) * 4 + 15) / 16 * 16;
    const unsigned *payload = reinterpret_cast<const unsigned *>(buf + pay_at);
    const unsigned bgw = (unsigned)header[7];
    const int off = offsets[tile];
    if (lane == 0 && tile == 0 && (header[0] != kWireMagic || header[1] > cap)) atomicOr(status, 1);
    uint3 q = background_quad(bgw);
    if (off >= 0 && off < cap) {
        const unsigned *src = payload + (size_t)off * (kTileBytes / 4) + lane * 3;
        q = make_uint3(src[0], src[1], src[2]);
    }
    const TileGeom g = tile_geom(tile, gx, T, lane);
    if (g.y >= H || g.x0 >= W) return;
    unsigned char *dst = out + ((((size_t)b * F + g.f) * H + g.y) * W + g.x0) * 3;
    if (g.x0 + 3 < W && (W & 3) == 0) {
        unsigned *d = reinterpret_cast<unsigned *>(dst);
        d[0] = q.x, d[1] = q.y, d[2] = q.z;
    } else {
        const unsigned w3[3] = {q.x, q.y, q.z};
        for (int k = 0; k < 12 && g.x0 + k / 3 < W; ++k) dst[k] = (w3[k >> 2] >> (8 * (k & 3))) & 255u;
    }
}

// Band form of the unpack for widths that are multiples of 16 pixels.  grid = (tile rows, frames, buffers): a block
// owns one band of 16 pixel rows of one frame, which is one contiguous run of the dense output, and walks it linearly
// in 16-byte chunks (256 threads x 16 B per trip), so the stores are as linear as a fill.  A chunk lies inside one
// tile row (48 bytes, 16-byte aligned); row and column of a chunk come from a multiply-shift by the host's reciprocal
// of the chunks per row, so no thread divides by a run-time value.
__global__ __launch_bounds__(256) void tile_unpack_rows_kernel(int F, int gx, int T, int H, int W, int cap,
                                                               unsigned per_row_magic,
                                                               const unsigned char *__restrict__ wire,
                                                               size_t wire_stride, unsigned char *__restrict__ out,
                                                               int *__restrict__ status) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const int f = blockIdx.y, b = blockIdx.z;
    const int per_row = W * 3 / 16;
    const unsigned char *buf = wire + (size_t)b * wire_stride;
    const int *header = reinterpret_cast<const int *>(buf);
    const unsigned char *payload = buf + ((size_t)(kWireHeaderInts + F + F * T) * 4 + 15) / 16 * 16;
    if (blockIdx.x == 0 && f == 0 && threadIdx.x == 0 && (header[0] != kWireMagic || header[1] > cap)) atomicOr(status, 1);
    const unsigned bgw = (unsigned)header[7];
    const unsigned ch[3] = {bgw & 255u, (bgw >> 8) & 255u, (bgw >> 16) & 255u};
    u32x4 bgv[3];  // background chunk by position inside the tile row: byte n of chunk `part` is channel (16 part + n) % 3
#pragma unroll
    for (int part = 0; part < 3; ++part)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int n0 = 16 * part + 4 * k;
            bgv[part][k] = ch[n0 % 3] | (ch[(n0 + 1) % 3] << 8) | (ch[(n0 + 2) % 3] << 16) | (ch[(n0 + 3) % 3] << 24);
        }
    __shared__ int slot[1024];
    const int gy = T / gx;
    for (int ty = blockIdx.x; ty < gy; ty += gridDim.x) {  // a block walks several bands: fewer, longer workgroups
        const int *offsets = header + kWireHeaderInts + F + f * T + ty * gx;
        const int rows = min(16, H - ty * 16);
        u32x4 *band = reinterpret_cast<u32x4 *>(out + (((size_t)b * F + f) * H + (size_t)ty * 16) * W * 3);
        __syncthreads();  // the previous band's slots have been consumed
        // the band's tile slots once, into LDS: a load in front of every store would put a memory round trip in each trip
        for (int tx = threadIdx.x; tx < gx; tx += blockDim.x) slot[tx] = offsets[tx];
        __syncthreads();
        for (int cidx = threadIdx.x; cidx < rows * per_row; cidx += blockDim.x) {
            const int r = (int)__umulhi((unsigned)cidx, per_row_magic), j = cidx - r * per_row;
            const int tx = j / 3, part = j - tx * 3;
            const int off = slot[tx];
            u32x4 v = part == 0 ? bgv[0] : (part == 1 ? bgv[1] : bgv[2]);
            if (off >= 0 && off < cap)
                v = *reinterpret_cast<const u32x4 *>(payload + (size_t)off * kTileBytes + r * 48 + part * 16);
            __builtin_nontemporal_store(v, band + cidx);  // streaming store: the frames are not read again on this GPU
        }
    }
}

// Differential form of the unpack.  The dense output buffer is REUSED from step to step (the exchange double-buffers
// two of them) and ~80 % of an avatar frame is background, so most tiles already hold what this step would write.
// `state` remembers per output tile what the buffer holds: the background word 0x00BBGGRR it was cleared to, or -1
// ("holds rendered pixels" / unknown; a fresh buffer starts as all -1).  A tile is written when it is stored on the
// wire (payload copy) or when it is background now but the buffer does not hold this background yet (clear); every
// other tile is skipped.  Bytes written per step drop from the dense 786 KB per frame to the stored tiles (+ the few
// tiles the body moved out of).
// grid = (frames, buffers); block = kDeltaWaves waves, wave w walks tile rows w, w + kDeltaWaves, ...; a tile row's
// selected tiles are written row by row of 16 pixels (48 B = three 16-byte chunks per tile, neighbouring tiles adjoin).
constexpr int kDeltaWaves = 16;  // waves per frame: the copies are latency-bound, so many short walkers
__global__ __launch_bounds__(kDeltaWaves * 64) void tile_unpack_delta_kernel(int F, int gx, int T, int H, int W, int cap,
                                                                const unsigned char *__restrict__ wire,
                                                                size_t wire_stride, unsigned char *__restrict__ out,
                                                                int *__restrict__ state, int *__restrict__ status) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ int delta_lds[];  // [T] source: slot >= 0, -1 = clear to background, -2 = skip; then kDeltaWaves x 128 ints of per-wave lists
    const int f = blockIdx.x, b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned char *buf = wire + (size_t)b * wire_stride;
    const int *header = reinterpret_cast<const int *>(buf);
    const unsigned char *payload = buf + ((size_t)(kWireHeaderInts + F + F * T) * 4 + 15) / 16 * 16;
    if (f == 0 && threadIdx.x == 0 && (header[0] != kWireMagic || header[1] > cap)) atomicOr(status, 1);
    const int bgw = header[7] & 0x00ffffff;
    const int *offsets = header + kWireHeaderInts + F + f * T;
    int *st = state + ((size_t)b * F + f) * T;
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
        const int off = offsets[t], had = st[t];
        int src = -2;
        if (off >= 0 && off < cap)
            src = off, st[t] = -1;
        else if (had != bgw)
            src = -1, st[t] = bgw;
        delta_lds[t] = src;
    }
    __syncthreads();
    const unsigned ch[3] = {(unsigned)bgw & 255u, ((unsigned)bgw >> 8) & 255u, ((unsigned)bgw >> 16) & 255u};
    u32x4 bgv[3];  // background chunk by position inside a tile row: byte n of chunk `part` is channel (16 part + n) % 3
#pragma unroll
    for (int part = 0; part < 3; ++part)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int n0 = 16 * part + 4 * k;
            bgv[part][k] = ch[n0 % 3] | (ch[(n0 + 1) % 3] << 8) | (ch[(n0 + 2) % 3] << 16) | (ch[(n0 + 3) % 3] << 24);
        }
    const int gy = T / gx;
    unsigned char *frame = out + ((size_t)b * F + f) * H * W * 3;
    for (int ty = wave; ty < gy; ty += kDeltaWaves) {
        const int rows = min(16, H - ty * 16);
        // the row's tiles that need a write, compacted (gx <= 64: one ballot; wider rows in rounds of 64)
        for (int tx0 = 0; tx0 < gx; tx0 += 64) {
            const int tx = tx0 + lane;
            const int src = tx < gx ? delta_lds[ty * gx + tx] : -2;
            const unsigned long long live = __ballot(src != -2);
            const int n = __popcll(live);
            if (n == 0) continue;
            // compact (tile column, source) of the selected tiles into this wave's list
            int *list = delta_lds + T + wave * 128;
            if (src != -2) {
                const int pos = __builtin_amdgcn_mbcnt_hi((unsigned)(live >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)live, 0));
                list[2 * pos] = tx, list[2 * pos + 1] = src;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // item k: pixel row r = k / (3 n), entry j = (k % (3 n)) / 3, 16-byte part k % 3 of the tile's 48-byte row;
            // four items per lane per round, all loads issued before the first store (the copies are independent)
            const int per_row = 3 * n, total = rows * per_row;
            for (int k0 = lane; k0 < total; k0 += 256) {
                u32x4 v[4];
                size_t dst[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = min(k0 + 64 * u, total - 1);
                    const int r = k / per_row, rem = k - r * per_row;
                    const int j = rem / 3, part = rem - 3 * j;
                    const int txs = list[2 * j], s2 = list[2 * j + 1];
                    v[u] = part == 0 ? bgv[0] : (part == 1 ? bgv[1] : bgv[2]);
                    if (s2 >= 0) v[u] = *reinterpret_cast<const u32x4 *>(payload + (size_t)s2 * kTileBytes + r * 48 + part * 16);
                    dst[u] = ((size_t)(ty * 16 + r) * W + txs * 16) * 3 + part * 16;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (k0 + 64 * u < total) *reinterpret_cast<u32x4 *>(frame + dst[u]) = v[u];
            }
            __builtin_amdgcn_wave_barrier();  // the list is rewritten by the next round
        }
    }
}


}  // namespace amav

using namespace amav;

extern "C" size_t amav_frames_wire_bytes(int F, int H, int W, int64_t cap_tiles) {
    if (F <= 0 || H <= 0 || W <= 0 || cap_tiles < 0) return 0;
    const long long tiles = (long long)F * ((H + 15) / 16) * ((W + 15) / 16);
    if (tiles > 0x7fffffffLL - kWireHeaderInts || cap_tiles > tiles) return 0;
    return (wire_payload_at(F, (int)tiles) + (size_t)cap_tiles * kTileBytes + 15) / 16 * 16;
}

static unsigned pack_bg(const float *bg) {
    auto q = [](float v) { return (unsigned)(unsigned char)(std::fmin(std::fmax(v, 0.f), 1.f) * 255.0f); };
    return q(bg[0]) | (q(bg[1]) << 8) | (q(bg[2]) << 16);
}

extern "C" int amav_frames_pack_tiles(int F, int H, int W, const float *rgba, const float *bg_host3,
                                      const int32_t *tile_hint, int64_t cap_tiles, void *wire, size_t wire_bytes,
                                      void *stream_) {
    AMAV_REQUIRE(F > 0 && H > 0 && W > 0 && cap_tiles >= 0, "amav_frames_pack_tiles: bad sizes F=%d H=%d W=%d", F, H, W);
    AMAV_REQUIRE(rgba && bg_host3 && wire, "amav_frames_pack_tiles: NULL pointer");
    AMAV_REQUIRE(((reinterpret_cast<uintptr_t>(rgba) | reinterpret_cast<uintptr_t>(wire)) & 15) == 0,
                 "amav_frames_pack_tiles: misaligned buffer");
    const size_t need = amav_frames_wire_bytes(F, H, W, cap_tiles);
    AMAV_REQUIRE(need != 0, "amav_frames_pack_tiles: capacity %lld exceeds the tile count", (long long)cap_tiles);
    if (wire_bytes < need) return fail(AMAV_ERR_WORKSPACE, "amav_frames_pack_tiles: wire buffer %zu < %zu", wire_bytes, need);
    const int gx = (W + 15) / 16, T = gx * ((H + 15) / 16), tiles = F * T;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const unsigned bgw = pack_bg(bg_host3);
    int *header = static_cast<int *>(wire), *frame_counts = header + kWireHeaderInts, *offsets = frame_counts + F;
    if (zero_async(header, (size_t)kWireHeaderInts * 4, stream) != hipSuccess)
        return fail(AMAV_ERR_LAUNCH, "amav_frames_pack_tiles: header clear failed");
    const float4 *px = reinterpret_cast<const float4 *>(rgba);
    const unsigned grid = (unsigned)((tiles + 3) / 4);
    if (tile_hint)
        tile_hint_kernel<<<(unsigned)((tiles + 255) / 256), 256, 0, stream>>>(tiles, tile_hint, offsets);
    else
        tile_flags_kernel<<<grid, 256, 0, stream>>>(tiles, gx, T, H, W, px, bgw, offsets);
    frame_count_kernel<<<F, 256, 0, stream>>>(T, offsets, frame_counts);
    tile_scan_kernel<<<F, 1024, 0, stream>>>(F, T, (int)cap_tiles, header, frame_counts, offsets);
    wire_header_kernel<<<1, 64, 0, stream>>>(header, F, T, H, W, (int)bgw);
    if (cap_tiles > 0)
        tile_compact_kernel<<<grid, 256, 0, stream>>>(tiles, gx, T, H, W, (int)cap_tiles, px, bgw, offsets,
                                                      reinterpret_cast<unsigned *>(static_cast<unsigned char *>(wire) +
                                                                                   wire_payload_at(F, tiles)));
    return check_launch("amav_frames_pack_tiles");
}

extern "C" int amav_frames_unpack_tiles(int num_buffers, int F, int H, int W, int64_t cap_tiles, const void *wire_all,
                                        size_t wire_stride, uint8_t *out_rgb8, int32_t *status, void *stream_) {
    AMAV_REQUIRE(num_buffers > 0 && F > 0 && H > 0 && W > 0 && cap_tiles >= 0, "amav_frames_unpack_tiles: bad sizes");
    AMAV_REQUIRE(wire_all && out_rgb8 && status, "amav_frames_unpack_tiles: NULL pointer");
    const size_t need = amav_frames_wire_bytes(F, H, W, cap_tiles);
    AMAV_REQUIRE(need != 0 && wire_stride >= need && wire_stride % 16 == 0,
                 "amav_frames_unpack_tiles: wire stride %zu does not hold a %zu-byte buffer", wire_stride, need);
    AMAV_REQUIRE((reinterpret_cast<uintptr_t>(wire_all) & 15) == 0 && (reinterpret_cast<uintptr_t>(out_rgb8) & 3) == 0,
                 "amav_frames_unpack_tiles: misaligned buffer");
    const int gx = (W + 15) / 16, T = gx * ((H + 15) / 16), tiles = F * T;
    if (W % 16 == 0 && (reinterpret_cast<uintptr_t>(out_rgb8) & 15) == 0 && F <= 65535 && num_buffers <= 65535) {
        const dim3 grid((unsigned)std::min(T / gx, 4), (unsigned)F, (unsigned)num_buffers);
        const unsigned per_row = (unsigned)(W * 3 / 16);
        const unsigned magic = (unsigned)((0x100000000ULL + per_row - 1) / per_row);  // exact for dividends < 2^16
        AMAV_REQUIRE(16u * per_row < 65536u && gx <= 1024, "amav_frames_unpack_tiles: width %d too large for the band kernel", W);
        tile_unpack_rows_kernel<<<grid, 256, 0, static_cast<hipStream_t>(stream_)>>>(
            F, gx, T, H, W, (int)cap_tiles, magic, static_cast<const unsigned char *>(wire_all), wire_stride, out_rgb8,
            status);
        return check_launch("amav_frames_unpack_tiles");
    }
    const long long total = (long long)num_buffers * tiles;
    tile_unpack_kernel<<<(unsigned)((total + 3) / 4), 256, 0, static_cast<hipStream_t>(stream_)>>>(
        num_buffers, tiles, gx, T, H, W, (int)cap_tiles, static_cast<const unsigned char *>(wire_all), wire_stride,
        out_rgb8, status);
    return check_launch("amav_frames_unpack_tiles");
}

extern "C" int amav_frames_unpack_tiles_delta(int num_buffers, int F, int H, int W, int64_t cap_tiles,
                                              const void *wire_all, size_t wire_stride, uint8_t *out_rgb8,
                                              int32_t *tile_state, int32_t *status, void *stream_) {
    AMAV_REQUIRE(num_buffers > 0 && F > 0 && H > 0 && W > 0 && cap_tiles >= 0, "amav_frames_unpack_tiles_delta: bad sizes");
    AMAV_REQUIRE(wire_all && out_rgb8 && status && tile_state, "amav_frames_unpack_tiles_delta: NULL pointer");
    const size_t need = amav_frames_wire_bytes(F, H, W, cap_tiles);
    AMAV_REQUIRE(need != 0 && wire_stride >= need && wire_stride % 16 == 0,
                 "amav_frames_unpack_tiles_delta: wire stride %zu does not hold a %zu-byte buffer", wire_stride, need);
    AMAV_REQUIRE(W % 16 == 0, "amav_frames_unpack_tiles_delta: width %d is not a multiple of 16 (use amav_frames_unpack_tiles)", W);
    AMAV_REQUIRE((reinterpret_cast<uintptr_t>(wire_all) & 15) == 0 && (reinterpret_cast<uintptr_t>(out_rgb8) & 15) == 0,
                 "amav_frames_unpack_tiles_delta: misaligned buffer");
    AMAV_REQUIRE(F <= 65535 * 32768 && num_buffers <= 65535, "amav_frames_unpack_tiles_delta: grid too large");
    const int gx = W / 16, T = gx * ((H + 15) / 16);
    AMAV_REQUIRE(((size_t)T + kDeltaWaves * 128) * sizeof(int) <= 64 * 1024, "amav_frames_unpack_tiles_delta: %d tiles per frame exceed the LDS table", T);
    const dim3 grid((unsigned)F, (unsigned)num_buffers);
    tile_unpack_delta_kernel<<<grid, kDeltaWaves * 64, ((size_t)T + kDeltaWaves * 128) * sizeof(int), static_cast<hipStream_t>(stream_)>>>(
        F, gx, T, H, W, (int)cap_tiles, static_cast<const unsigned char *>(wire_all), wire_stride, out_rgb8, tile_state,
        status);
    return check_launch("amav_frames_unpack_tiles_delta");
}

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
