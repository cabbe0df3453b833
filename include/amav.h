/*
 * amav.h -- C ABI of libamav_hip.so: the MI355X (gfx950) kernels of the audio-driven avatar rendering hot path.
 *
 * The reference (liubingqi7/audio-motion-avatar) has no FFI layer: its native work is reached through
 * third-party Python wheels.  Each entry point below replaces one of those call sites (cited per function,
 * paths relative to the reference root) and is what a ctypes binding on the reference side would bind
 * (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - Every pointer named *_dev / documented "device" is a caller-owned HIP device pointer (a torch tensor's
 *     data_ptr()).  The library never allocates, frees or retains device memory: scratch is a caller-provided
 *     workspace sized by the matching *_workspace_bytes() query.
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream).  All work is enqueued
 *     on it; no call synchronises unless documented.  No hipMalloc / hipFree / host sync on the hot path, so every
 *     launch function may be captured into a hipGraph.
 *   - Return value: 0 = AMAV_OK, negative = error; amav_last_error() returns a thread-local message.
 *   - All floating-point data is IEEE fp32; indices are int32 unless stated.
 *   - Thread-safe for distinct streams/workspaces; no mutable global state.
 */
#ifndef AMAV_H
#define AMAV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AMAV_OK 0
#define AMAV_ERR_INVALID_ARG (-1)
#define AMAV_ERR_LAUNCH (-2)
#define AMAV_ERR_WORKSPACE (-3)
#define AMAV_ERR_NO_DEVICE (-4)

#define AMAV_TILE 16          /* rasterizer tile edge, pixels (fixed by the algorithm being replaced) */
#define AMAV_GAUSS_STRIDE 16  /* floats per packed Gaussian record written by amav_triplane_sample_decode */

const char *amav_version(void);
const char *amav_last_error(void);
/* Number of visible HIP devices, or a negative error.  Does not create a context on a device. */
int amav_device_count(void);
/* Process-wide arithmetic selection, effective from the next call on (the environment variables AMAV_ATTN / AMAV_LBS give
 * the defaults; there is no reference counterpart: the reference computes these products in library fp32).
 *   name "attn": "fp16" (fp16 x 2 split products, default) | "bf16" (bf16 x 3) | "f32" (exact products on the fp32 MFMA)
 *   name "lbs":  "split" (fp16 x 2 split blend product, default) | "f32" (fp32 MFMA)
 * value "default" returns to the environment's choice.  Not to be changed while calls of that entry point are in
 * flight on another thread (a call reads it more than once: workspace layout, then launch). */
int amav_set_option(const char *name, const char *value);

/* Timing events for callers without a HIP binding of their own (thin hipEvent wrappers; elapsed synchronises). */
int amav_event_create(void **event);
int amav_event_destroy(void *event);
int amav_event_record(void *event, void *stream);
int amav_event_elapsed_ms(void *start, void *stop, float *ms);

/* One per-Gaussian (or per-point) attribute: element (f, i) lives at ptr[f * frame_stride + i * elem_stride].
 * frame_stride = 0 broadcasts one set of Gaussians to every frame (render_multi_view, renderer.py:431-445). */
typedef struct amav_attr {
    const float *ptr;
    int64_t frame_stride; /* in floats */
    int32_t elem_stride;  /* in floats */
    int32_t _pad;
} amav_attr;

/* ------------------------------------------------------------------------------------------------------------
 * Camera.  Replaces the per-frame host code of render_one (src/models/renderer.py:486-510) and
 * getWorld2View2_torch / getProjectionMatrix_torch / focal2fov_torch (src/utils/graphic_utils.py:67-78,103-145):
 * K [F,3,3], E [F,4,4] (row-major, device) -> viewmatrix [F,16] = E^T, projmatrix [F,16] = (K_ndc E)^T (both as
 * the rasterizer reads them: column-major), tanfov [F,2] = (W/(2fx), H/(2fy)), campos [F,3].  No host sync.
 */
int amav_camera_from_intrinsics(int num_frames, const float *K_dev, const float *E_dev, int height, int width,
                                float znear, float zfar, float *viewmatrix_dev, float *projmatrix_dev,
                                float *tanfov_dev, float *campos_dev, void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * Gaussian tile rasterizer, forward, all frames of a shard in one call.
 * Replaces GaussianRasterizer.forward of diff_gaussian_rasterization as called at src/models/renderer.py:555-566
 * (preprocess, per-tile binning, per-tile (depth, index) sort, 16x16-tile front-to-back alpha compositing) and,
 * with apply_activations = 1, the torch ops of renderer.py:532-547,568 that surround it.
 */
typedef struct amav_raster_args {
    int32_t num_frames, num_gaussians, height, width;
    amav_attr means3d;   /* 3 floats: world position */
    amav_attr rotations; /* 4 floats: unit quaternion (w,x,y,z); NOT normalised here (renderer.py:333 did it) */
    amav_attr scales;    /* 3 floats */
    amav_attr opacities; /* 1 float  */
    amav_attr colors;    /* 3 floats (colors_precomp; the SH branch of the reference is dead code) */
    const float *viewmatrix; /* device [F,16] */
    const float *projmatrix; /* device [F,16] */
    const float *tanfov;     /* device [F,2]  */
    float bg[3];             /* background colour (renderer.py:512-514: white by default) */
    float scale_modifier;    /* renderer.py:522: 1.0 */
    /* 1: scales = min(exp(s - scale_bias), scale_max), opacities = sigmoid(o - opacity_bias),
     *    colors = clamp(c, 0, 1) are applied on load (renderer.py:532-533,547 with SCALE_BIAS 3.9, cap 0.1,
     *    OPACITY_BIAS 0.0).  0: inputs are already activated (the op-level GaussianRasterizer contract). */
    int32_t apply_activations;
    float scale_bias, scale_max, opacity_bias;
    int32_t antialiasing; /* renderer.py:529: False */
    int32_t clamp_output; /* 1: clamp RGB to [0,1] (renderer.py:568) */
    /* outputs (device).  out_rgba is required: [F,H,W,4] = (R,G,B,alpha = 1 - T_final), pixel-interleaved so a tile
     * row is one 256-byte run.  out_inv_depth [F,H,W] and out_radii [F,N] (int32) may be NULL. */
    float *out_rgba;
    float *out_inv_depth;
    int32_t *out_radii;
    /* scratch */
    void *workspace;
    size_t workspace_bytes;
    int64_t instance_capacity; /* instances the workspace was sized for; each frame owns capacity / F of them */
    /* optional hipEvent_t pair (amav_event_create) recorded on `stream` right before / after the blend kernel, so a
     * caller can time the dominant kernel live (bench.py roofline); NULL = off */
    void *profile_start_event;
    void *profile_stop_event;
    /* diagnostic only: device buffer of num_frames * tiles * 6 uint64 that receives per-tile-wave clock stamps
     * (start, ranges read, sorted, blended, stored) and the list length; NULL in production */
    void *debug_stamps;
    /* optional: the tile-sparse wire buffer of the frame exchange (amav_frames_wire_bytes(F, H, W, wire_capacity_tiles)
     * bytes, 16-byte aligned), written by the rasterizer itself: every tile that holds a Gaussian is stored (uint8 RGB,
     * quantised as amav_frames_to_rgb8 does, of the clamped colour), the others are background -- the result of
     * amav_frames_pack_tiles with the rasterizer's tile counts as hint, without the pass over the fp32 frames.  Needs
     * clamp_output = 1.  Tiles beyond the capacity are dropped and header.count > capacity tells the receivers
     * (amav_frames_unpack_tiles raises its status flag).  NULL = off. */
    void *wire;
    size_t wire_bytes;
    int64_t wire_capacity_tiles;
} amav_raster_args;

size_t amav_rasterize_workspace_bytes(int num_frames, int num_gaussians, int height, int width,
                                      int64_t instance_capacity);
int amav_rasterize_forward(const amav_raster_args *args, void *stream);
/* Synchronises `stream`, then reports the last forward on this workspace: the total instance count, the largest
 * per-frame count, and whether some frame exceeded its region (instance_capacity / num_frames instances each), in
 * which case the outputs are invalid and the caller must retry with instance_capacity >= num_frames *
 * *max_frame_instances.  The only rasterizer call that waits on the device. */
int amav_rasterize_status(const void *workspace, int64_t *total_instances, int64_t *max_frame_instances,
                          int32_t *overflow, void *stream);

/* Rendered frames -> on-wire format of the multi-GPU exchange: fp32 RGBA [pixels,4] -> uint8 RGB [pixels,3] with the
 * reference's own quantisation (src/main2.py:351: (frame * 255).astype(uint8), truncating). num_pixels % 4 == 0. */
int amav_frames_to_rgb8(int64_t num_pixels, const float *rgba_dev, uint8_t *out_rgb8_dev, void *stream);

/* Tile-sparse form of the same exchange (lossless): an avatar frame is mostly background, so only the 16x16 tiles
 * that differ from the background colour are put on the wire.  One wire buffer per rank:
 *   int32 header[16] = {magic, stored tiles, capacity, F, tiles per frame, H, W, background as 0x00BBGGRR, 0...}
 *   int32 stored tiles per frame [F]
 *   int32 slot of every tile [F * tiles per frame]   (-1: background; else index into the payload)
 *   uint8 payload [capacity][16*16*3]                (16-byte aligned; tiles in (frame, tile) order)
 * amav_frames_wire_bytes: size of such a buffer (0 on bad arguments).
 * amav_frames_pack_tiles: fp32 RGBA frames [F,H,W,4] -> wire (quantised as amav_frames_to_rgb8 does); tiles beyond
 *   `capacity_tiles` are dropped and header[1] > header[2] tells every receiver.  capacity 0 only counts.
 *   tile_hint (optional, int32 [F * tiles per frame]): zero = the caller knows the tile is pure background, e.g.
 *   amav_rasterize_tile_counts (a tile without Gaussians); with a hint the frames are read for the stored tiles only.
 * amav_frames_unpack_tiles: `num_buffers` gathered wire buffers (rank r at wire_all + r * wire_stride) -> dense uint8
 *   RGB [num_buffers * F, H, W, 3]; status[0] |= 1 if any buffer was truncated or is not a wire buffer (the caller
 *   re-packs with more room, as with the rasterizer's instance capacity).  No host synchronisation in either call.
 * amav_frames_unpack_tiles_delta: the same into an output buffer that is REUSED from step to step (width % 16 == 0).
 *   tile_state int32 [num_buffers * F * tiles per frame] belongs to that output buffer and records what every tile
 *   of it holds: the background word 0x00BBGGRR it was last cleared to, or -1 (rendered pixels / unknown -- a fresh
 *   buffer's state is all -1).  Only the tiles stored on the wire, and the tiles that are background now but do not
 *   hold this background yet, are written; the state is updated in the same launch. */
/* Per-tile Gaussian list lengths of the last amav_rasterize_forward on this workspace, int32 [F * tiles per frame]
 * (frame-major, tiles row-major; all ones after an instance-capacity overflow).  Same sizes as the forward call. */
int amav_rasterize_tile_counts(const void *workspace, int num_frames, int num_gaussians, int height, int width,
                               int64_t instance_capacity, int32_t *out_counts_dev, void *stream);
size_t amav_frames_wire_bytes(int num_frames, int height, int width, int64_t capacity_tiles);
int amav_frames_pack_tiles(int num_frames, int height, int width, const float *rgba_dev, const float *background_host3,
                           const int32_t *tile_hint_dev, int64_t capacity_tiles, void *wire_dev, size_t wire_bytes,
                           void *stream);
int amav_frames_unpack_tiles(int num_buffers, int num_frames, int height, int width, int64_t capacity_tiles,
                             const void *wire_all_dev, size_t wire_stride, uint8_t *out_rgb8_dev, int32_t *status_dev,
                             void *stream);
int amav_frames_unpack_tiles_delta(int num_buffers, int num_frames, int height, int width, int64_t capacity_tiles,
                                   const void *wire_all_dev, size_t wire_stride, uint8_t *out_rgb8_dev,
                                   int32_t *tile_state_dev, int32_t *status_dev, void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * SMPL-X forward + linear blend skinning for F frames.
 * Replaces smplx.SMPLX.forward -> smplx.lbs.lbs as called at src/models/renderer.py:261-274 (no transl;
 * use_pca=False).  Model constants are immutable device tables prepared once by the host mirror
 * (audio-motion-avatar_amd/body_model.py) from the SMPL-X arrays:
 *   v_template [V,3]; blend [ceil(V/32), KB, 3, 32] with KB = n_coeff + (J-1)*9: for every tile of 32 vertices (the
 *   last one zero padded) the KB blend rows as x / y / z planes, rows 0..n_coeff-1 = shape + expression directions,
 *   then posedirs; 16-byte aligned; j_template [J,3] = J_regressor v_template; j_dirs [J*3, n_coeff] = J_regressor applied to
 *   the shape directions; parents [J]; skin_idx / skin_w [V, skin_k]: the non-zero LBS weights of each vertex in
 *   ascending joint order, padded with weight 0.
 */
typedef struct amav_body_tables {
    int32_t num_verts, num_joints, num_coeffs, skin_k;
    const float *v_template;
    const float *blend;
    const float *j_template;
    const float *j_dirs;
    const int32_t *parents;
    const int32_t *skin_idx;
    const float *skin_w;
    const void *blend_split; /* amav_lbs_prepare_blend_split's buffer, or NULL (the blend product then runs on fp32 MFMA) */
} amav_body_tables;

/* The blend table as two fp16 parts (scaled by one power of two) in the fragment order of the 16-bit MFMA: with it
 * amav_lbs_forward computes the [F, KB] x [KB, 3V] blend product as three fp16 partial products per fp32 product with
 * fp32 accumulation (the fp32 result to 2^-22, 3x faster than on fp32 MFMA; every frame's features get their own
 * power-of-two scale).  Prepare once per model into a 256-byte aligned device buffer of amav_lbs_blend_split_bytes and
 * put its address into tables->blend_split (`blend` must stay valid too: short batches use it).  AMAV_LBS=f32 in the
 * environment ignores blend_split. */
size_t amav_lbs_blend_split_bytes(const amav_body_tables *tables);
int amav_lbs_prepare_blend_split(const amav_body_tables *tables, void *out_dev, size_t out_bytes, void *stream);

size_t amav_lbs_workspace_bytes(int num_frames, const amav_body_tables *tables);
/* full_pose [F, J*3] axis-angle (pose_mean already added), coeffs [F, n_coeff] (betas then expression).
 * out_vertices [F,V,3]; out_joint_transforms [F,J,12] (rows of the 3x4 rest-pose-removed transforms) may be NULL. */
int amav_lbs_forward(int num_frames, const amav_body_tables *tables, const float *full_pose_dev,
                     const float *coeffs_dev, float *out_vertices_dev, float *out_joint_transforms_dev,
                     void *workspace, size_t workspace_bytes, void *stream);

/* The same with the pose and the coefficients as the caller of the SMPL-X layer holds them -- the keyword arguments of
 * renderer.py:261-272 (global_orient, body_pose, jaw_pose, leye_pose, reye_pose, left_hand_pose, right_hand_pose / betas,
 * expression) -- concatenated in the given order as the first kernel loads them, plus smplx's `full_pose += pose_mean`:
 * what smplx does with torch.cat / add / torch.cat (three launches) in front of the joint chain.  Part p holds
 * pose_joints[p] axis-angle triples per frame, pose_stride[p] floats apart between frames; the joint counts must add up
 * to tables->num_joints and the coefficient counts to tables->num_coeffs.  pose_mean [J*3] may be NULL. */
typedef struct amav_pose_parts {
    int32_t num_pose_parts;  /* 1..8 */
    int32_t num_coeff_parts; /* 1..4 */
    const float *pose[8];
    int32_t pose_joints[8];
    int64_t pose_stride[8];
    const float *pose_mean;
    const float *coeff[4];
    int32_t coeff_count[4];
    int64_t coeff_stride[4];
} amav_pose_parts;
int amav_lbs_forward_parts(int num_frames, const amav_body_tables *tables, const amav_pose_parts *parts,
                           float *out_vertices_dev, float *out_joint_transforms_dev, void *workspace,
                           size_t workspace_bytes, void *stream);

/* Densify + subset of the posed vertices (src/models/renderer.py:276-288: pytorch3d SubdivideMeshes applied to the
 * posed mesh once or twice, then a vertex subset) as one baked table of 4 base-vertex ids per output point:
 *   point n = 1/2 * ( 1/2 * (v[a0] + v[b0]) + 1/2 * (v[a1] + v[b1]) ),   idx[n] = (a0, b0, a1, b1)
 * which is the midpoint-of-midpoints order subdivision evaluates in; an original vertex repeats its id four times
 * and a level-1 midpoint repeats its pair (halving and adding equal values is exact, so all three cases are
 * bit-identical to the sequential subdivision). */
int amav_points_gather(int num_frames, int num_verts, int num_points, const float *vertices_dev,
                       const int32_t *idx_dev, float *out_points_dev, void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * Triplane decode: F.grid_sample x 3 planes + the five Gaussian heads + construct_gaussians
 * (src/models/renderer.py:136,158,165-181,292-346), restructured for HBM: because the heads are linear in the
 * sampled features and bilinear sampling is linear in the texels, each texel is projected through the head
 * weights ONCE (amav_triplane_project: streams the [C, 3R^2] token slab exactly once, coalesced) and the points
 * then sample the 16-channel projected planes (amav_triplane_sample_decode).
 *
 * head_w_plane: device [3, C, 16]: head_w_plane[p][c][o] = Wcat[o][3 + p*C + c] where Wcat [14, 3C+3] stacks the
 * head weights in the output order below (rows 14,15 zero).  head_w_point: device [16,4]: columns 0..2 =
 * Wcat[o][0..2] (the xyz inputs), column 3 = bias.  Output channel order o (= packed record layout):
 *   0-2 xyz_offset, 3 opacity | 4-7 rotation (w,x,y,z) | 8-10 scaling, 11 pad | 12-14 shs, 15 pad.
 */
int amav_triplane_project(int num_frames, int channels, int resolution, const float *tokens_dev,
                          int64_t tokens_frame_stride, const float *head_w_plane_dev, float *out_proj_dev,
                          void *stream);
/* The same, restricted to the texels the sampling can touch: boxes [F,6] = {min x, y, z, max x, y, z} of the points
 * (amav_points_bbox of the point set handed to amav_triplane_sample_decode, or of the posed vertices the subdivision
 * table of amav_triplane_sample_decode_indexed averages: a midpoint lies inside the box of its ends in floating point
 * too), radius as in the sampling call.  Every tap address the sampling kernels form for a point inside the box --
 * including the clamped addresses of zero-padded taps -- lies inside the projected rectangle (each step of
 * p -> texel is monotonic), so the decoded Gaussians are bit-identical; texels outside are left unwritten.  The avatar
 * covers a fifth to a third of each plane, and the token slab is the largest stream of the path.  boxes = NULL: all. */
int amav_triplane_project_region(int num_frames, int channels, int resolution, const float *tokens_dev,
                                 int64_t tokens_frame_stride, const float *head_w_plane_dev, float *out_proj_dev,
                                 const float *boxes_dev, float radius, void *stream);
/* boxes [F,6] = per frame {min x, y, z, max x, y, z} of points [F,N,3]; a frame holding a NaN gets the infinite box. */
int amav_points_bbox(int num_frames, int num_points, const float *points_dev, float *out_boxes_dev, void *stream);
/* points [F,N,3]; transl [F,3] or NULL; proj [F,3,R,R,16] from amav_triplane_project.
 * out_gaussians [F,N,16] packed records: xyz = p + offset + transl, opacity (raw logit) | rot (normalised) |
 * scale (raw), 0 | color = sigmoid(shs), 0.   (renderer.py:333-344) */
int amav_triplane_sample_decode(int num_frames, int num_points, int resolution, const float *proj_dev,
                                const float *points_dev, const float *transl_dev, float radius,
                                const float *head_w_point_dev, float *out_gaussians_dev, void *stream);
/* The same with amav_points_gather fused in: the N points are gathered from the posed vertices [F,V,3] through the
 * baked subdivision table idx [N,4] (identical operation order, so identical bits), saving the [F,N,3] round trip. */
int amav_triplane_sample_decode_indexed(int num_frames, int num_points, int resolution, int num_verts,
                                        const float *proj_dev, const float *vertices_dev, const int32_t *idx_dev,
                                        const float *transl_dev, float radius, const float *head_w_point_dev,
                                        float *out_gaussians_dev, void *stream);
/* Plain Renderer.sample_from_triplane (renderer.py:292-317): element (f,p,c,h,w) of the planes lives at
 * planes[f*frame_stride + p*plane_stride + c*chan_stride + h*R + w] (so both the [F,3,C,R,R] tensor and the
 * [F,C,(3 R R)] token layout are addressable); points [F,N,3] -> features [F,N,3C] in (plane, channel) order. */
int amav_triplane_sample_features(int num_frames, int num_points, int channels, int resolution,
                                  const float *planes_dev, int64_t frame_stride, int64_t plane_stride,
                                  int64_t chan_stride, const float *points_dev, float radius,
                                  float *out_features_dev, void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * Stage-1 identity encoder: the point <-> triplane-cell reductions of
 * SMPLXTriplaneEncoder and the point -> pixel feature lookup, deterministic segment reductions.
 * Replaces torch_scatter.scatter_max / scatter_mean as used at src/models/triplane_net.py:226-244 and the pytorch3d
 * point rasterizer + index_put of points_projection (src/utils/graphic_utils.py:275-331).
 *   order [B,3,N] (amav_cell_max) / [B,N] (amav_cell_mean): point ids sorted by cell (stable); seg [.., cells + 1]:
 *   offsets of every cell's run in `order`; cell_of [B,3,N]: cell of every point per plane.
 * amav_cell_max:    cellmax [B,3,cells,C] = per-cell channel-wise maximum of feat [B,N,C] (0 for an empty cell)
 * amav_cell_gather: out [B,N,C] = cellmax[plane 0] + cellmax[plane 1] + cellmax[plane 2] at the point's cells
 *                   (pool_local, triplane_net.py:226-238)
 * amav_cell_mean:   out_planes [B,C,cells] = per-cell mean of feat [B,N,C], summed in ascending point id, 0 if empty
 *                   (generate_plane_features, triplane_net.py:240-244)
 * amav_points_project: points [B,N,3] (world), w2c [B,4,4] row-major, intrinsics [B,3,3] (OpenCV pixels), features
 *   [B,C,H,W] -> out [B,N,C]: z-buffer of discs of radius_px around the projected points (pixel centres at +0.5);
 *   a point that is the nearest one at some pixel takes the features of the LAST such pixel in (y, x) order, every
 *   other point zeros. */
int amav_cell_max(int batch, int num_points, int channels, int cells, const float *feat_dev, const int32_t *order_dev,
                  const int32_t *seg_dev, float *cellmax_dev, void *stream);
int amav_cell_gather(int batch, int num_points, int channels, int cells, const float *cellmax_dev,
                     const int32_t *cell_of_dev, float *out_dev, void *stream);
int amav_cell_mean(int batch, int num_points, int channels, int cells, const float *feat_dev, const int32_t *order_dev,
                   const int32_t *seg_dev, float *out_planes_dev, void *stream);
size_t amav_points_project_workspace_bytes(int batch, int num_points, int height, int width);
int amav_points_project(int batch, int num_points, int channels, int height, int width, const float *points_dev,
                        const float *w2c_dev, const float *intrinsics_dev, const float *features_dev, float radius_px,
                        float *out_dev, void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * Self-attention of the audio transformer (diffusers Attention -> F.scaled_dot_product_attention as reached from
 * src/models/transformers.py:329-336): softmax(Q K^T * scale) V, fp32 in/out, no mask.  q,k,v,out: [B, S, H*D] with
 * row stride `row_stride` floats (so a fused QKV projection output can be passed without a copy); D must be 64.  The key
 * sweep may be split over several workgroups for load balance; their partial softmax states and the split K / V
 * operands live in the caller's workspace.
 *
 * The products are fp32-equivalent sums of low-precision partial products on the 16-bit MFMA pipe (1.1e-7 max abs
 * against fp64 at the reference shape; the library's fp32 SDPA: 4.8e-7): by default two fp16 parts per operand, three
 * partial products, every operand pre-scaled by a power of two taken from its magnitude -- measured by a pre-pass over
 * q, k, v (amav_selfattn_forward), or handed over by a caller that can PROVE upper bounds of |q|, |k|, |v|
 * (amav_selfattn_forward_bounded with all three > 0; a bound the data exceeds by more than 2x overflows fp16).  The
 * process-wide switch AMAV_ATTN=bf16 selects three bf16 parts / six partial products (no scaling involved),
 * AMAV_ATTN=f32 the exact-product fp32 MFMA kernel (v_mfma_f32_32x32x2_f32).
 */
size_t amav_selfattn_workspace_bytes(int batch, int seq_len, int heads, int head_dim);
int amav_selfattn_forward(int batch, int seq_len, int heads, int head_dim, const float *q_dev, const float *k_dev,
                          const float *v_dev, int64_t row_stride, float *out_dev, int64_t out_row_stride,
                          float scale, void *workspace, size_t workspace_bytes, void *stream);
int amav_selfattn_forward_bounded(int batch, int seq_len, int heads, int head_dim, const float *q_dev,
                                  const float *k_dev, const float *v_dev, int64_t row_stride, float *out_dev,
                                  int64_t out_row_stride, float scale, float q_bound, float k_bound, float v_bound,
                                  void *workspace, size_t workspace_bytes, void *stream);
/* The same, with the result ALSO (or, when the key range is split over workgroups, ONLY) written as the fp16 x 2 activation
 * operand of the projection that follows (amav_split_operand's AMAV_SPLIT_FP16X2 layout, rows of [h2 | h1 | h1] with K =
 * heads * head_dim, x 2^split_scale_exp = h1 + h2): diffusers' to_out after the attention (transformers.py:329-336, 448).
 * out_split_dev [batch * seq_len, 3 K] fp16, 16-byte aligned; out_dev rows dense (out_row_stride = K) -- whether out_dev is
 * written depends on the key split the kernel chooses, so a caller that passes out_split must not read it.  NULL: as above. */
int amav_selfattn_forward_split_out(int batch, int seq_len, int heads, int head_dim, const float *q_dev, const float *k_dev,
                                    const float *v_dev, int64_t row_stride, float *out_dev, int64_t out_row_stride,
                                    float scale, float q_bound, float k_bound, float v_bound, void *out_split_dev,
                                    int split_scale_exp, void *workspace, size_t workspace_bytes, void *stream);

/* Operand of an fp32-equivalent nn.Linear (src/models/transformers.py:70-84, 448, 505: the to_q/k/v, to_out and
 * feed-forward projections) computed as ONE low-precision GEMM with fp32 accumulation over operands split into parts
 * and concatenated along K.  x [rows, k] fp32 (row stride in floats, k a multiple of 8) -> out [rows, parts * k]:
 *   AMAV_SPLIT_BF16X3  x = x1 + x2 + x3 (bf16 each); activations (weights = 0) [x3 x2 x1 x2 x1 x1], weights (weights = 1)
 *                      [w1 w2 w3 w1 w2 w1]: A' B'^T = the six partial products x_i w_j^T with i + j <= 4 (K' = 6 k).
 *                      Any finite input; scale_exp is ignored.
 *   AMAV_SPLIT_FP16X2  x 2^scale_exp = h1 + h2 (fp16 each); activations [h2 h1 h1], weights [g1 g2 g1]: A' B'^T =
 *                      h2 g1 + h1 g2 + h1 g1 (K' = 3 k, half the matrix work).  The caller picks scale_exp from a bound
 *                      on |x| so that |x| 2^scale_exp <= 32768 (no overflow) and typical values keep their residual out
 *                      of fp16's subnormals, and multiplies the product by 2^-(scale_exp_a + scale_exp_w). */
#define AMAV_SPLIT_BF16X3 0
#define AMAV_SPLIT_FP16X2 1
int amav_split_operand(int64_t rows, int k, const float *x_dev, int64_t x_row_stride, int weights, int format,
                       int scale_exp, void *out_dev, void *stream);

/* GEGLU gate of the transformer feed-forward (src/models/transformers.py:484-508, exact-erf GELU):
 * proj [rows, 2*inner] (row stride in floats) -> [rows, inner] = proj[:, :inner] * gelu(proj[:, inner:]).
 * bias [2*inner] (may be NULL) is added to proj first: the projection's bias when its GEMM ran without one.
 * The result goes to exactly one of out (fp32) and out_split (the AMAV_SPLIT_FP16X2 activation operand [rows, 3*inner]
 * of the output projection, pre-scaled by 2^split_scale_exp); the other is NULL. */
int amav_geglu(int64_t rows, int inner, const float *proj_dev, int64_t proj_row_stride, const float *bias_dev,
               float *out_dev, void *out_split_dev, int split_scale_exp, void *stream);

/* Residual adds + LayerNorm of BasicTransformerBlock (src/models/transformers.py:292-399) in one pass over [rows, dim]
 * (dim in {256, 512, 768, 1024}):  h = ((add + add_bias) + hidden);  h = (batch_row[row / rows_per_batch] + h);
 * hidden_out = h;  norm = LayerNorm(h) * weight + bias.  `add` [rows, dim], `add_bias` [dim] (the bias of the projection
 * that produced `add`, when its GEMM ran without one) and `batch_row` [batches, dim] may be NULL; hidden_out may alias
 * hidden.  The normalised rows go to exactly one of out_norm (fp32 [rows, dim]) and out_norm_split (the activation
 * operand of amav_split_operand in split_format / split_scale_exp, for the projection that follows); the other is
 * NULL. */
int amav_add_layernorm(int64_t rows, int dim, int64_t rows_per_batch, const float *add_dev, const float *add_bias_dev,
                       const float *batch_row_dev, const float *hidden_dev, float *hidden_out_dev,
                       const float *weight_dev, const float *bias_dev, float eps, float *out_norm_dev,
                       void *out_norm_split_dev, int split_format, int split_scale_exp, void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * Point refiner: the sparse / serialised operators of the reference's
 * PointTransformerV3 (src/models/point_transformer/pointtransformer_v3.py:81-145,328-499,618-759; point_encoder.py:25-40;
 * called from src/models/renderer.py:143-151) over a batch of CLOUDS (one per frame; n = all their points, a
 * cloud's points contiguous).  Replaces spconv.SubMConv3d (hash-table submanifold convolution), torch_scatter.segment_csr
 * and the padded-patch softmax attention.  Deterministic semantics (oracle/ptv3.py header): per-cloud
 * grid origin / depth / patch size, stable sorts, a voxel is seen by its neighbours through its lowest row.
 *
 * amav_cloud_voxelize   grid [n,3] = floor(resolution * p) - per-cloud minimum; cloud_depth [clouds] = bit length of
 *                       the cloud's largest grid coordinate; bounds [clouds,6] int32 scratch.     (point_encoder.py:33)
 * amav_cloud_codes      keys [4,n] int64 = cloud << 48 | code for the orders z, z-trans, hilbert, hilbert-trans
 *                       (serialization/default.py:10-27, z_order.py:86-118, hilbert.py:93-190); a stable sort of a row
 *                       orders every cloud at once.
 * amav_cloud_neighbors  nbr [n, ksize^3] int32: row gathered by tap (a,b,c) (offset (a,b,c) - ksize/2 on x,y,z), -1
 *                       where the voxel is empty; centre tap = the row itself.  sorted_keys / order: the z-order keys
 *                       ascending and the rows in that order; cloud_start [clouds+1] int32.         (spconv SubMConv3d)
 * amav_subm_pair_gemm   the convolution's products, only where a voxel exists: pairs grouped by tap (tap_start [taps+1]
 *                       int32), pair p of tap t: products[p] [cout] = feat[pair_src[p]] [cin] x weights[t] [cin][cout]
 *                       (weights [taps,cin,cout]); tile_start [taps+1] int32 = prefix sum of ceil(pairs of tap / 128),
 *                       tiles = its last entry; cin, cout multiples of 32.  fp32 MFMA.
 * amav_subm_pair_gemm_split  the same products as three fp16 partial products per fp32 product on the 16-bit MFMA pipe
 *                       (fp32 accumulation, the fp32 result to 2^-22): weights_split = the buffer of
 *                       amav_subm_prepare_weights_split (weights as two fp16 parts scaled by one power of two, in MFMA
 *                       fragment order; prepare once per layer into 256-byte aligned memory of
 *                       amav_subm_weights_split_bytes); feat [n_rows, cin] is scaled by one power of two from its largest
 *                       magnitude, measured by a pre-pass into scratch16 (16 bytes of device memory).
 * amav_subm_pair_sum    out [n,cout] = bias + sum over taps (ascending) of products[pair_of[i][tap]] (pair_of [n,taps]
 *                       int32, -1: no voxel); bias may be NULL.
 * amav_patch_attention  out [n, heads*head_dim] = softmax(q k^T * scale) v inside patches of a serialised order.
 *                       qkv [n, 3*heads*head_dim] (q | k | v, head-major inside each); order [n] int64 rows in serialised
 *                       order; patch_desc [patches,4] int32 = {first sorted position, K, own, 0}: slot j of the patch is
 *                       sorted position first + j for j < own and first + j - K otherwise (a cloud's last patch borrows
 *                       the tail of the one before, pointtransformer_v3.py:419-432); only slots < own are stored.
 *                       head_dim in {16, 32, 64}; max_patch = largest K.                (pointtransformer_v3.py:449-499)
 * amav_cluster_max      out [clusters,C] = gelu(scale * max over rows members[seg[j]..seg[j+1]) of x + shift)
 *                       (segment_csr 'max' + BatchNorm(eval) + GELU, pointtransformer_v3.py:693-719)
 * amav_bn_gelu          out = gelu(x * scale + shift), [rows, C]                                    (:785-788,738-744)
 * amav_unpool_merge     skip = gelu(x * scale + shift); sum = skip + up[cluster[row]]               (:748-755)
 * amav_rows_norm        out_sum = base + (weight_a ? LayerNorm_a(x) : x); out_norm = LayerNorm_b(out_sum), rows of
 *                       channels in {32, 64, 128, 256, 512}: `feat + cpe(...)` + norm1 and `feat + attn` + norm2 of a
 *                       Block in one pass each (:595-609); weight_a / bias_a may both be NULL.
 */
int amav_cloud_voxelize(int64_t n, int clouds, const float *points_dev, const int32_t *cloud_of_dev, float resolution,
                        int32_t *grid_dev, int32_t *cloud_depth_dev, int32_t *bounds_dev, void *stream);
int amav_cloud_codes(int64_t n, const int32_t *grid_dev, const int32_t *cloud_of_dev, const int32_t *cloud_depth_dev,
                     int64_t *keys_dev, void *stream);
int amav_cloud_neighbors(int64_t n, int ksize, const int32_t *grid_dev, const int32_t *cloud_of_dev,
                         const int32_t *cloud_depth_dev, const int32_t *cloud_start_dev, const int64_t *sorted_keys_dev,
                         const int64_t *order_dev, int32_t *nbr_dev, void *stream);
int amav_subm_pair_gemm(int64_t pairs, int tiles, int taps, int cin, int cout, const float *feat_dev,
                        const int32_t *pair_src_dev, const int32_t *tap_start_dev, const int32_t *tile_start_dev,
                        const float *weights_dev, float *products_dev, void *stream);
size_t amav_subm_weights_split_bytes(int taps, int cin, int cout);
int amav_subm_prepare_weights_split(int taps, int cin, int cout, const float *weights_dev, void *out_dev,
                                    size_t out_bytes, void *stream);
int amav_subm_pair_gemm_split(int64_t pairs, int tiles, int taps, int cin, int cout, int64_t n_rows,
                              const float *feat_dev, const int32_t *pair_src_dev, const int32_t *tap_start_dev,
                              const int32_t *tile_start_dev, const void *weights_split_dev, void *scratch16_dev,
                              float *products_dev, void *stream);
int amav_subm_pair_sum(int64_t n, int taps, int cout, const float *products_dev, const int32_t *pair_of_dev,
                       const float *bias_dev, float *out_dev, void *stream);
int amav_patch_attention(int patches, int max_patch, int heads, int head_dim, const float *qkv_dev,
                         const int64_t *order_dev, const int32_t *patch_desc_dev, float *out_dev, float scale,
                         void *stream);
int amav_cluster_max(int64_t clusters, int channels, const float *x_dev, const int64_t *members_dev,
                     const int64_t *seg_dev, const float *scale_dev, const float *shift_dev, float *out_dev, void *stream);
int amav_bn_gelu(int64_t rows, int channels, const float *x_dev, const float *scale_dev, const float *shift_dev,
                 float *out_dev, void *stream);
int amav_rows_norm(int64_t rows, int channels, const float *x_dev, const float *base_dev, const float *weight_a_dev,
                   const float *bias_a_dev, const float *weight_b_dev, const float *bias_b_dev, float eps,
                   float *out_sum_dev, float *out_norm_dev, void *stream);
int amav_unpool_merge(int64_t rows, int channels, const float *x_dev, const float *scale_dev, const float *shift_dev,
                      const float *up_dev, const int64_t *cluster_dev, float *skip_dev, float *sum_dev, void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * Library GEMM of the fp16 x 2 split projections: out[rows, n] fp32 = alpha * a[rows, k3] x
 * w[n, k3]^T with fp16 operands and fp32 accumulation -- the three partial products of an fp32-equivalent nn.Linear
 * (src/models/transformers.py:70-84, 448, 505) concatenated along K.  hipBLASLt does the arithmetic; `algo_index` names
 * one of its kernels (as found by amav_gemm_split_fp16_tune for this shape on this library build), -1 or an index the
 * library does not accept selects its own heuristic's choice; -2 runs this library's own kernel instead (n % 128 == 0,
 * k3 = 3 K with K % 32 == 0, operands laid out [h2 | h1 | h1] and [g1 | g2 | g1]: it reads the first two thirds and issues
 * the three partial products itself; correct, 70-80 % of the tuned library kernels' speed, not on the product path).  `workspace` may be NULL for kernels that need none.
 * amav_gemm_split_fp16_tune synchronises: it times every kernel of the library on the given operands (`repeats` runs
 * each; with `copies` > 1 the three buffers hold that many consecutive operand sets and run i uses set i % copies, which
 * times a kernel as it runs inside the step, not out of a hot L2) and reports the fastest one's index and time next to
 * the heuristic choice's time.
 * amav_gemm_library_version: the library build the indices belong to. */
int amav_gemm_split_fp16(int64_t rows, int n, int k3, const void *a_fp16, const void *w_fp16, float alpha, float *out,
                         int algo_index, void *workspace, size_t workspace_bytes, void *stream);
int amav_gemm_split_fp16_tune(int64_t rows, int n, int k3, const void *a_fp16, const void *w_fp16, float *out, void *workspace,
                              size_t workspace_bytes, int repeats, int copies, int32_t *best_index, float *best_ms,
                              float *heuristic_ms, void *stream);
const char *amav_gemm_library_version(void);

#ifdef __cplusplus
}
#endif
#endif /* AMAV_H */
