"""Tensor-level wrappers over the C ABI (include/amav.h).  torch is used for device memory and streams only.

Every function requires CUDA(HIP) float32 tensors and raises on anything else: there is no CPU path.
"""
import ctypes

import torch

from . import _lib
from ._lib import AmavError, Attr, BodyTables, PoseParts, RasterArgs, check

SCALE_BIAS = 3.9    # src/models/renderer.py:428
OPACITY_BIAS = 0.0  # src/models/renderer.py:429
SCALE_MAX = 0.1     # src/models/renderer.py:532
GAUSS_STRIDE = 16   # floats per packed Gaussian record (AMAV_GAUSS_STRIDE)
# channel offsets inside a packed record
REC_XYZ, REC_OPACITY, REC_ROT, REC_SCALE, REC_COLOR = 0, 3, 4, 8, 12


def tensor_version(t: torch.Tensor) -> int:
    """Version counter of a tensor for "has it been modified in place" cache keys, or -1 for inference tensors (created
    under torch.inference_mode(): they track no version -- reading `_version` raises -- and cannot be modified outside
    it, so identity + storage address identify their contents)."""
    return -1 if t.is_inference() else t._version


def set_option(name: str, value: str):
    """Process-wide arithmetic selection (include/amav.h, amav_set_option): attn = fp16 | bf16 | f32, lbs = split | f32,
    value "default" = the environment's choice.  The projections' GEMM format is the host-side AMAV_GEMM variable."""
    check(_lib.lib().amav_set_option(name.encode(), value.encode()), "amav_set_option")


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need(t: torch.Tensor, name: str, dtype=torch.float32):
    if not isinstance(t, torch.Tensor):
        raise AmavError(f"{name}: expected a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise AmavError(f"{name}: tensor is on {t.device}; this package only runs on an MI355X (HIP) device")
    if t.dtype != dtype:
        raise AmavError(f"{name}: dtype {t.dtype}, expected {dtype}")
    return t


def _contig(t, name, dtype=torch.float32):
    t = _need(t, name, dtype)
    return t if t.is_contiguous() else t.contiguous()


def _attr(t: torch.Tensor, name: str, width: int) -> Attr:
    """[F,N,width] tensor (any frame/element strides, unit stride on the last axis) -> amav_attr."""
    _need(t, name)
    if t.dim() != 3 or t.shape[2] != width:
        raise AmavError(f"{name}: expected [F,N,{width}], got {tuple(t.shape)}")
    if width > 1 and t.stride(2) != 1:
        t = t.contiguous()
    return Attr(t.data_ptr(), t.stride(0), t.stride(1), 0), t


class Event:
    """hipEvent through the C ABI (bench.py times the blend kernel with a pair of these)."""

    def __init__(self):
        h = ctypes.c_void_p()
        check(_lib.lib().amav_event_create(ctypes.byref(h)), "amav_event_create")
        self.handle = h

    def record(self):
        check(_lib.lib().amav_event_record(self.handle, _stream()), "amav_event_record")

    def elapsed_ms(self, stop) -> float:
        ms = ctypes.c_float(0)
        check(_lib.lib().amav_event_elapsed_ms(self.handle, stop.handle, ctypes.byref(ms)), "amav_event_elapsed_ms")
        return ms.value

    def __del__(self):
        try:
            _lib.lib().amav_event_destroy(self.handle)
        except Exception:
            pass


# --------------------------------------------------------------------------------------------------------- camera
def camera_from_intrinsics(K, E, height, width, znear=0.01, zfar=100.0):
    """K [F,3,3], E [F,4,4] -> (viewmatrix [F,16], projmatrix [F,16], tanfov [F,2], campos [F,3]); no host sync.

    Replaces src/models/renderer.py:486-510 (which costs >= 6 host syncs per frame in the reference).
    """
    K = _contig(K.reshape(-1, 3, 3), "K")
    E = _contig(E.reshape(-1, 4, 4), "E")
    F = K.shape[0]
    if E.shape[0] != F:
        raise AmavError(f"camera: {F} intrinsics vs {E.shape[0]} extrinsics")
    dev = K.device
    view = torch.empty(F, 16, device=dev)
    proj = torch.empty(F, 16, device=dev)
    tanfov = torch.empty(F, 2, device=dev)
    campos = torch.empty(F, 3, device=dev)
    check(_lib.lib().amav_camera_from_intrinsics(F, K.data_ptr(), E.data_ptr(), int(height), int(width), znear, zfar,
                                                 view.data_ptr(), proj.data_ptr(), tanfov.data_ptr(),
                                                 campos.data_ptr(), _stream()), "amav_camera_from_intrinsics")
    return view, proj, tanfov, campos


# ----------------------------------------------------------------------------------------------------- rasterizer
class RasterWorkspace:
    """Caller-owned scratch of the rasterizer for one problem size (reusable across calls and hipGraph-safe)."""

    def __init__(self, num_frames, num_gaussians, height, width, instance_capacity, device):
        self.key = (num_frames, num_gaussians, height, width)
        self.capacity = int(instance_capacity)
        nbytes = _lib.lib().amav_rasterize_workspace_bytes(num_frames, num_gaussians, height, width, self.capacity)
        if nbytes == 0:
            raise AmavError(f"amav_rasterize_workspace_bytes rejected {self.key} capacity={self.capacity}")
        self.buffer = torch.empty(nbytes, dtype=torch.uint8, device=device)

    def tile_counts(self, out=None):
        """Per-tile Gaussian list lengths of the last forward, int32 [F * tiles] (no host sync)."""
        F, N, H, W = self.key
        n = F * ((H + 15) // 16) * ((W + 15) // 16)
        if out is None:
            out = torch.empty(n, dtype=torch.int32, device=self.buffer.device)
        check(_lib.lib().amav_rasterize_tile_counts(self.buffer.data_ptr(), F, N, H, W, self.capacity, out.data_ptr(),
                                                    _stream()), "amav_rasterize_tile_counts")
        return out

    def status(self):
        """(total_instances, overflowed) of the last forward.  Synchronises the current stream."""
        total, _, over = self.status_full()
        return total, over

    def status_full(self):
        """(total_instances, max_instances_of_a_frame, overflowed).  Synchronises the current stream."""
        total, mx, over = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int32(0)
        check(_lib.lib().amav_rasterize_status(self.buffer.data_ptr(), ctypes.byref(total), ctypes.byref(mx),
                                               ctypes.byref(over), _stream()), "amav_rasterize_status")
        return total.value, mx.value, bool(over.value)


# A list of (Event, Event) pairs: every rasterize() call pops one and records it around its blend kernel (bench.py).
PROFILE_EVENTS = None
# diagnostic: int64 CUDA tensor [F*tiles, 6] that receives the blend kernel's per-tile clock stamps (tools/)
DEBUG_STAMPS = None


def default_instance_capacity(num_frames, num_gaussians, per_gaussian=16):
    return int(num_frames) * int(num_gaussians) * per_gaussian


def rasterize(means3d, rotations, scales, opacities, colors, viewmatrix, projmatrix, tanfov, height, width,
              bg=(1.0, 1.0, 1.0), apply_activations=False, scale_modifier=1.0, antialiasing=False, clamp_output=False,
              want_inv_depth=False, want_radii=False, workspace=None, check_overflow=True, out_rgba=None,
              profile_events=None, wire=None):
    """Batched tile rasterizer.  Gaussian attributes are [F,N,*] (frame stride 0 = shared across frames).

    Returns dict(rgba [F,H,W,4], inv_depth [F,H,W] | None, radii [F,N] | None, workspace).
    With check_overflow the call synchronises once to read the instance count and transparently retries with a
    larger workspace; without it the caller must consult workspace.status() before trusting the output.
    `wire`: (uint8 buffer, capacity in tiles) -- the rasterizer also writes the frame exchange's tile-sparse wire buffer
    (include/amav.h, amav_raster_args.wire; needs clamp_output): what frames_pack_tiles would produce from these frames
    with the tile counts as hint, without the extra pass.
    """
    a_m, means3d = _attr(means3d, "means3d", 3)
    a_r, rotations = _attr(rotations, "rotations", 4)
    a_s, scales = _attr(scales, "scales", 3)
    a_o, opacities = _attr(opacities, "opacities", 1)
    a_c, colors = _attr(colors, "colors", 3)
    F, N = means3d.shape[0], means3d.shape[1]
    for name, t in (("rotations", rotations), ("scales", scales), ("opacities", opacities), ("colors", colors)):
        if t.shape[0] != F or t.shape[1] != N:
            raise AmavError(f"{name}: shape {tuple(t.shape)} does not match means3d [F={F},N={N},3]")
    viewmatrix = _contig(viewmatrix.reshape(F, 16), "viewmatrix")
    projmatrix = _contig(projmatrix.reshape(F, 16), "projmatrix")
    tanfov = _contig(tanfov.reshape(F, 2), "tanfov")
    dev = means3d.device
    H, W = int(height), int(width)
    if out_rgba is None:
        out_rgba = torch.empty(F, H, W, 4, device=dev)
    else:
        _need(out_rgba, "out_rgba")
        if tuple(out_rgba.shape) != (F, H, W, 4) or not out_rgba.is_contiguous():
            raise AmavError(f"out_rgba must be contiguous [F,H,W,4] = {(F, H, W, 4)}")
    inv_depth = torch.empty(F, H, W, device=dev) if want_inv_depth else None
    radii = torch.empty(F, N, dtype=torch.int32, device=dev) if want_radii else None
    if workspace is None:
        workspace = RasterWorkspace(F, N, H, W, default_instance_capacity(F, N), dev)
    elif workspace.key != (F, N, H, W):
        raise AmavError(f"workspace was sized for {workspace.key}, call is {(F, N, H, W)}")

    def launch(ws):
        args = RasterArgs()
        args.num_frames, args.num_gaussians, args.height, args.width = F, N, H, W
        args.means3d, args.rotations, args.scales, args.opacities, args.colors = a_m, a_r, a_s, a_o, a_c
        args.viewmatrix, args.projmatrix, args.tanfov = viewmatrix.data_ptr(), projmatrix.data_ptr(), tanfov.data_ptr()
        args.bg = (ctypes.c_float * 3)(*[float(b) for b in bg])
        args.scale_modifier = float(scale_modifier)
        args.apply_activations = int(bool(apply_activations))
        args.scale_bias, args.scale_max, args.opacity_bias = SCALE_BIAS, SCALE_MAX, OPACITY_BIAS
        args.antialiasing = int(bool(antialiasing))
        args.clamp_output = int(bool(clamp_output))
        args.out_rgba = out_rgba.data_ptr()
        args.out_inv_depth = inv_depth.data_ptr() if inv_depth is not None else None
        args.out_radii = radii.data_ptr() if radii is not None else None
        args.workspace, args.workspace_bytes = ws.buffer.data_ptr(), ws.buffer.numel()
        args.instance_capacity = ws.capacity
        if DEBUG_STAMPS is not None:
            args.debug_stamps = DEBUG_STAMPS.data_ptr()
        if wire is not None:
            buf, cap = wire
            if buf.dtype != torch.uint8 or not buf.is_cuda or not buf.is_contiguous():
                raise AmavError("rasterize: wire must be a contiguous uint8 device buffer")
            args.wire, args.wire_bytes, args.wire_capacity_tiles = buf.data_ptr(), buf.numel(), int(cap)
        ev = profile_events
        if ev is None and PROFILE_EVENTS:
            ev = PROFILE_EVENTS.pop(0)
        if ev is not None:
            args.profile_start_event, args.profile_stop_event = ev[0].handle, ev[1].handle
        check(_lib.lib().amav_rasterize_forward(ctypes.byref(args), _stream()), "amav_rasterize_forward")

    launch(workspace)
    if check_overflow:
        total, max_frame, over = workspace.status_full()
        if over:  # every frame owns capacity / F instances: size the retry by the fullest frame
            workspace = RasterWorkspace(F, N, H, W, F * max_frame, dev)
            launch(workspace)
            total, max_frame, over = workspace.status_full()
            if over:
                raise AmavError(f"rasterizer overflowed twice (instances={total}, fullest frame {max_frame})")
    return dict(rgba=out_rgba, inv_depth=inv_depth, radii=radii, workspace=workspace)


def frames_to_rgb8(rgba, out=None):
    """fp32 RGBA [...,4] (contiguous) -> uint8 RGB [...,3], truncating like src/main2.py:351."""
    rgba = _need(rgba, "rgba")
    if not rgba.is_contiguous() or rgba.shape[-1] != 4:
        raise AmavError("frames_to_rgb8: need a contiguous [...,4] tensor")
    shape = tuple(rgba.shape[:-1]) + (3,)
    if out is None:
        out = torch.empty(shape, dtype=torch.uint8, device=rgba.device)
    elif tuple(out.shape) != shape or out.dtype != torch.uint8 or not out.is_contiguous():
        raise AmavError(f"frames_to_rgb8: out must be contiguous uint8 {shape}")
    check(_lib.lib().amav_frames_to_rgb8(rgba.numel() // 4, rgba.data_ptr(), out.data_ptr(), _stream()),
          "amav_frames_to_rgb8")
    return out


def frames_wire_bytes(F, H, W, capacity_tiles):
    n = _lib.lib().amav_frames_wire_bytes(int(F), int(H), int(W), int(capacity_tiles))
    if n == 0:
        raise AmavError(f"frames_wire_bytes: bad sizes F={F} H={H} W={W} capacity={capacity_tiles}")
    return int(n)


def frames_pack_tiles(rgba, capacity_tiles, bg=(1.0, 1.0, 1.0), wire=None, tile_hint=None):
    """fp32 RGBA [F,H,W,4] -> tile-sparse wire buffer (uint8 tensor, include/amav.h).  No host sync; use
    frames_wire_count() (which synchronises) to read how many tiles were stored.  `tile_hint`: int32 [F*tiles], zero
    where the caller knows the tile is background (RasterWorkspace.tile_counts())."""
    rgba = _need(rgba, "rgba")
    if rgba.dim() != 4 or rgba.shape[-1] != 4 or not rgba.is_contiguous() or rgba.dtype != torch.float32:
        raise AmavError("frames_pack_tiles: need a contiguous fp32 [F,H,W,4] tensor")
    F, H, W = (int(x) for x in rgba.shape[:3])
    need = frames_wire_bytes(F, H, W, capacity_tiles)
    if wire is None:
        wire = torch.empty(need, dtype=torch.uint8, device=rgba.device)
    elif wire.dtype != torch.uint8 or not wire.is_contiguous() or wire.numel() < need:
        raise AmavError(f"frames_pack_tiles: wire must be a contiguous uint8 buffer of >= {need} bytes")
    bg3 = (ctypes.c_float * 3)(*[float(c) for c in bg])
    hint_ptr = None
    if tile_hint is not None:
        tile_hint = _need(tile_hint, "tile_hint", torch.int32)
        if not tile_hint.is_contiguous() or tile_hint.numel() != F * ((H + 15) // 16) * ((W + 15) // 16):
            raise AmavError("frames_pack_tiles: tile_hint must be a contiguous int32 [F * tiles] tensor")
        hint_ptr = tile_hint.data_ptr()
    check(_lib.lib().amav_frames_pack_tiles(F, H, W, rgba.data_ptr(), bg3, hint_ptr, int(capacity_tiles),
                                            wire.data_ptr(), wire.numel(), _stream()), "amav_frames_pack_tiles")
    return wire


def frames_wire_count(wire):
    """(stored tiles, capacity) of a packed wire buffer; synchronises."""
    head = wire[:16].view(torch.int32).cpu()
    return int(head[1]), int(head[2])


def frames_unpack_tiles(wire_all, num_buffers, F, H, W, capacity_tiles, out=None, status=None, state=None):
    """`num_buffers` gathered wire buffers (a uint8 tensor [num_buffers, stride]) -> uint8 RGB [num_buffers*F,H,W,3].
    `status` (int32 [1], accumulated) becomes non-zero when a sender had to drop tiles.

    `state` (int32 [num_buffers * F * tiles], made by frames_tile_state()) selects the differential form for an `out`
    buffer that is reused from step to step: only stored tiles and tiles that must return to background are written
    (include/amav.h, amav_frames_unpack_tiles_delta); `state` belongs to `out` and is updated in place."""
    wire_all = _need(wire_all, "wire_all", torch.uint8)
    if wire_all.dtype != torch.uint8 or not wire_all.is_contiguous() or wire_all.dim() != 2 or \
            wire_all.shape[0] != num_buffers:
        raise AmavError("frames_unpack_tiles: wire_all must be a contiguous uint8 [num_buffers, stride] tensor")
    dev = wire_all.device
    if out is None:
        if state is not None:
            raise AmavError("frames_unpack_tiles: `state` describes a buffer the caller keeps: pass it as `out`")
        out = torch.empty(num_buffers * F, H, W, 3, dtype=torch.uint8, device=dev)
    elif tuple(out.shape) != (num_buffers * F, H, W, 3) or out.dtype != torch.uint8 or not out.is_contiguous():
        raise AmavError(f"frames_unpack_tiles: out must be contiguous uint8 {(num_buffers * F, H, W, 3)}")
    if status is None:
        status = torch.zeros(1, dtype=torch.int32, device=dev)
    if state is not None:
        tiles = num_buffers * F * ((H + 15) // 16) * ((W + 15) // 16)
        state = _need(state, "state", torch.int32)
        if state.numel() != tiles or not state.is_contiguous():
            raise AmavError(f"frames_unpack_tiles: state must be a contiguous int32 tensor of {tiles} entries")
        check(_lib.lib().amav_frames_unpack_tiles_delta(int(num_buffers), int(F), int(H), int(W), int(capacity_tiles),
                                                        wire_all.data_ptr(), wire_all.shape[1], out.data_ptr(),
                                                        state.data_ptr(), status.data_ptr(), _stream()),
              "amav_frames_unpack_tiles_delta")
        return out, status
    check(_lib.lib().amav_frames_unpack_tiles(int(num_buffers), int(F), int(H), int(W), int(capacity_tiles),
                                              wire_all.data_ptr(), wire_all.shape[1], out.data_ptr(), status.data_ptr(),
                                              _stream()), "amav_frames_unpack_tiles")
    return out, status


DELTA_UNPACK_MAX_TILES = 64 * 1024 // 4 - 16 * 128  # amav_frames_unpack_tiles_delta keeps (T + 16 * 128) ints in LDS


def frames_delta_unpack_supported(H, W) -> bool:
    """Whether amav_frames_unpack_tiles_delta accepts frames of this size (width a multiple of 16 and a per-frame tile
    table that fits its 64 KiB of LDS: 14 336 tiles -- the reference's 1296 x 2304 frames have 11 664, a 3840 x 2160
    frame 32 400)."""
    return W % 16 == 0 and ((H + 15) // 16) * ((W + 15) // 16) <= DELTA_UNPACK_MAX_TILES


def frames_tile_state(num_buffers, F, H, W, device):
    """Fresh per-tile state of a reusable dense output buffer for the differential unpack: every tile unknown (-1)."""
    if not frames_delta_unpack_supported(H, W):
        raise AmavError(f"frames_tile_state: {H}x{W} frames are outside the differential unpack's limits "
                        "(use frames_unpack_tiles without `state`)")
    return torch.full((num_buffers * F * ((H + 15) // 16) * ((W + 15) // 16),), -1, dtype=torch.int32, device=device)


# ------------------------------------------------------------------------------------------------------------ LBS
def body_tables_struct(tables: dict) -> BodyTables:
    """dict of device tensors prepared by body_model.BodyModel.device_tables() -> amav_body_tables."""
    t = BodyTables()
    t.num_verts, t.num_joints = tables["v_template"].shape[0], tables["parents"].shape[0]
    t.num_coeffs, t.skin_k = tables["j_dirs"].shape[1], tables["skin_idx"].shape[1]
    for k in ("v_template", "blend", "j_template", "j_dirs", "skin_w"):
        setattr(t, k, _contig(tables[k], k).data_ptr())
    for k in ("parents", "skin_idx"):
        setattr(t, k, _contig(tables[k], k, torch.int32).data_ptr())
    split = tables.get("blend_split")
    t.blend_split = None if split is None else _contig(split, "blend_split", torch.uint8).data_ptr()
    return t


def lbs_prepare_blend_split(tables: dict):
    """-> uint8 device buffer: the blend table as two scaled fp16 parts in MFMA fragment order (include/amav.h,
    amav_lbs_prepare_blend_split).  Put it into the tables dict as "blend_split": lbs_forward then runs the blend product
    on the 16-bit matrix pipe."""
    ts = body_tables_struct({k: v for k, v in tables.items() if k != "blend_split"})
    nbytes = _lib.lib().amav_lbs_blend_split_bytes(ctypes.byref(ts))
    if nbytes == 0:
        raise AmavError("amav_lbs_blend_split_bytes rejected the tables: " + _lib.lib().amav_last_error().decode())
    out = torch.empty(nbytes, dtype=torch.uint8, device=tables["blend"].device)
    check(_lib.lib().amav_lbs_prepare_blend_split(ctypes.byref(ts), out.data_ptr(), nbytes, _stream()),
          "amav_lbs_prepare_blend_split")
    return out


def lbs_forward(tables: dict, full_pose, coeffs, want_transforms=False):
    """full_pose [F, J*3], coeffs [F, n_coeff] -> vertices [F,V,3] (and A [F,J,12]).  renderer.py:261-274."""
    full_pose = _contig(full_pose, "full_pose")
    coeffs = _contig(coeffs, "coeffs")
    F = full_pose.shape[0]
    ts = body_tables_struct(tables)
    if full_pose.shape != (F, ts.num_joints * 3) or coeffs.shape != (F, ts.num_coeffs):
        raise AmavError(f"lbs: full_pose {tuple(full_pose.shape)} / coeffs {tuple(coeffs.shape)} do not match "
                        f"J={ts.num_joints}, n_coeff={ts.num_coeffs}")
    dev = full_pose.device
    nbytes = _lib.lib().amav_lbs_workspace_bytes(F, ctypes.byref(ts))
    if nbytes == 0:
        raise AmavError("amav_lbs_workspace_bytes rejected the tables: " + _lib.lib().amav_last_error().decode())
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    verts = torch.empty(F, ts.num_verts, 3, device=dev)
    A = torch.empty(F, ts.num_joints, 12, device=dev) if want_transforms else None
    check(_lib.lib().amav_lbs_forward(F, ctypes.byref(ts), full_pose.data_ptr(), coeffs.data_ptr(), verts.data_ptr(),
                                      A.data_ptr() if A is not None else None, ws.data_ptr(), nbytes, _stream()),
          "amav_lbs_forward")
    return (verts, A) if want_transforms else verts


def lbs_forward_parts(tables: dict, pose_parts, coeff_parts, pose_mean=None, want_transforms=False):
    """lbs_forward with the pose / coefficients as the SMPL-X call's keyword arguments hold them (renderer.py:261-272):
    `pose_parts` = float32 tensors [F, joints_p * 3] (rows may be strided, elements contiguous) concatenated in order,
    `coeff_parts` likewise (betas, expression), `pose_mean` [J*3] is added to the concatenated pose (smplx's
    full_pose += pose_mean).  No torch.cat / add launches: the joint-chain kernel reads the parts."""
    ts = body_tables_struct(tables)
    F = int(pose_parts[0].shape[0])
    pp = PoseParts()
    keep = []

    def rows(tt, name):
        _need(tt, name)
        if tt.dtype != torch.float32 or tt.dim() != 2 or tt.shape[0] != F or (tt.shape[1] > 1 and tt.stride(1) != 1):
            raise AmavError(f"lbs: {name} must be float32 [F={F}, n] with contiguous rows, got {tuple(tt.shape)} {tt.dtype}")
        if F > 1 and tt.stride(0) < tt.shape[1]:
            tt = tt.contiguous()  # broadcast rows
        keep.append(tt)
        return tt

    if not 1 <= len(pose_parts) <= 8 or not 1 <= len(coeff_parts) <= 4:
        raise AmavError(f"lbs: {len(pose_parts)} pose parts (1..8), {len(coeff_parts)} coefficient parts (1..4)")
    pp.num_pose_parts, pp.num_coeff_parts = len(pose_parts), len(coeff_parts)
    for q, part in enumerate(pose_parts):
        part = rows(part, f"pose part {q}")
        if part.shape[1] % 3:
            raise AmavError(f"lbs: pose part {q} has {part.shape[1]} columns (not axis-angle triples)")
        pp.pose[q], pp.pose_joints[q] = part.data_ptr(), part.shape[1] // 3
        pp.pose_stride[q] = part.stride(0) if F > 1 else part.shape[1]
    for q, part in enumerate(coeff_parts):
        part = rows(part, f"coefficient part {q}")
        pp.coeff[q], pp.coeff_count[q] = part.data_ptr(), part.shape[1]
        pp.coeff_stride[q] = part.stride(0) if F > 1 else part.shape[1]
    if sum(pp.pose_joints[q] for q in range(len(pose_parts))) != ts.num_joints or \
            sum(pp.coeff_count[q] for q in range(len(coeff_parts))) != ts.num_coeffs:
        raise AmavError(f"lbs: the parts do not add up to J={ts.num_joints} joints / {ts.num_coeffs} coefficients")
    if pose_mean is not None:
        pose_mean = _contig(pose_mean.reshape(-1), "pose_mean")
        if pose_mean.numel() != ts.num_joints * 3:
            raise AmavError(f"lbs: pose_mean has {pose_mean.numel()} entries, expected {ts.num_joints * 3}")
        pp.pose_mean = pose_mean.data_ptr()
    dev = keep[0].device
    nbytes = _lib.lib().amav_lbs_workspace_bytes(F, ctypes.byref(ts))
    if nbytes == 0:
        raise AmavError("amav_lbs_workspace_bytes rejected the tables: " + _lib.lib().amav_last_error().decode())
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    verts = torch.empty(F, ts.num_verts, 3, device=dev)
    A = torch.empty(F, ts.num_joints, 12, device=dev) if want_transforms else None
    check(_lib.lib().amav_lbs_forward_parts(F, ctypes.byref(ts), ctypes.byref(pp), verts.data_ptr(),
                                            A.data_ptr() if A is not None else None, ws.data_ptr(), nbytes, _stream()),
          "amav_lbs_forward_parts")
    return (verts, A) if want_transforms else verts


def points_gather(vertices, idx4):
    """vertices [F,V,3], idx4 [N,4] int32 -> points [F,N,3] (baked subdivision + subset, renderer.py:276-288)."""
    vertices = _contig(vertices, "vertices")
    idx4 = _contig(idx4, "idx4", torch.int32)
    F, V, _ = vertices.shape
    N = idx4.shape[0]
    out = torch.empty(F, N, 3, device=vertices.device)
    check(_lib.lib().amav_points_gather(F, V, N, vertices.data_ptr(), idx4.data_ptr(), out.data_ptr(), _stream()),
          "amav_points_gather")
    return out


# ------------------------------------------------------------------------------------------------------- triplane
def pack_head_weights(weights: dict, channels: int, device):
    """The five head Linear layers (renderer.py:51-55) -> (head_w_plane [3,C,16], head_w_point [16,4]).

    `weights` maps 'xyz_layer' | 'rotation_layer' | 'scaling_layer' | 'opacity_layer' | 'shs_layer' to
    (weight [out, 3C+3], bias [out]).  Output channel order = packed record layout (include/amav.h).
    """
    C = channels
    rows = {"xyz_layer": (0, 3), "opacity_layer": (3, 1), "rotation_layer": (4, 4), "scaling_layer": (8, 3),
            "shs_layer": (12, 3)}
    Wcat = torch.zeros(16, 3 * C + 3, device=device)
    bias = torch.zeros(16, device=device)
    for name, (o, n) in rows.items():
        w, b = weights[name]
        if tuple(w.shape) != (n, 3 * C + 3):
            raise AmavError(f"{name}.weight has shape {tuple(w.shape)}, expected {(n, 3 * C + 3)}")
        Wcat[o:o + n] = w.detach().to(device=device, dtype=torch.float32)
        bias[o:o + n] = b.detach().to(device=device, dtype=torch.float32)
    w_plane = Wcat[:, 3:].reshape(16, 3, C).permute(1, 2, 0).contiguous()  # [3, C, 16]
    w_point = torch.cat([Wcat[:, :3], bias[:, None]], dim=1).contiguous()   # [16, 4]
    return w_plane, w_point


def points_bbox(points):
    """points [F,N,3] -> [F,6] = per frame (min xyz, max xyz); a frame with a NaN gets the infinite box."""
    points = _contig(points, "points")
    if points.dim() != 3 or points.shape[-1] != 3:
        raise AmavError(f"points_bbox: expected [F,N,3], got {tuple(points.shape)}")
    F, N = int(points.shape[0]), int(points.shape[1])
    out = torch.empty(F, 6, device=points.device)
    check(_lib.lib().amav_points_bbox(F, N, points.data_ptr(), out.data_ptr(), _stream()), "amav_points_bbox")
    return out


def triplane_project(tokens, head_w_plane, resolution, region=None, out=None):
    """tokens [F, C, 3*R*R] (reference token layout, renderer.py:85-91) -> projected planes [F,3,R,R,16].
    `region` = (boxes [F,6] from points_bbox, radius): only the texels that sampling points inside the boxes can touch
    are projected (include/amav.h, amav_triplane_project_region); the others are left unwritten (of `out`, when given)."""
    _need(tokens, "tokens")
    if tokens.dim() != 3 or tokens.stride(2) != 1 or tokens.stride(1) != tokens.shape[2]:
        tokens = tokens.contiguous()
    F, C, S = tokens.shape
    R = int(resolution)
    if S != 3 * R * R:
        raise AmavError(f"tokens last dim {S} != 3*R*R = {3 * R * R}")
    head_w_plane = _contig(head_w_plane, "head_w_plane")
    if tuple(head_w_plane.shape) != (3, C, 16):
        raise AmavError(f"head_w_plane {tuple(head_w_plane.shape)} != {(3, C, 16)}")
    if out is None:
        out = torch.empty(F, 3, R, R, 16, device=tokens.device)
    elif tuple(out.shape) != (F, 3, R, R, 16) or not out.is_contiguous() or out.dtype != torch.float32 or not out.is_cuda:
        raise AmavError(f"triplane_project: out must be a contiguous float32 device tensor {(F, 3, R, R, 16)}")
    boxes, radius = None, 1.0
    if region is not None:
        boxes, radius = region
        boxes = _contig(boxes, "region boxes")
        if tuple(boxes.shape) != (F, 6) or not float(radius) > 0.0:
            raise AmavError(f"triplane_project: region boxes {tuple(boxes.shape)} != {(F, 6)} or radius {radius} <= 0")
    check(_lib.lib().amav_triplane_project_region(F, C, R, tokens.data_ptr(), tokens.stride(0), head_w_plane.data_ptr(),
                                                  out.data_ptr(), boxes.data_ptr() if boxes is not None else None,
                                                  float(radius), _stream()), "amav_triplane_project_region")
    return out


def _decode_out(out, F, N, device):
    if out is None:
        return torch.empty(F, N, GAUSS_STRIDE, device=device)
    _need(out, "out")
    if tuple(out.shape) != (F, N, GAUSS_STRIDE) or not out.is_contiguous():
        raise AmavError(f"out must be contiguous {(F, N, GAUSS_STRIDE)}, got {tuple(out.shape)}")
    return out


def triplane_sample_decode(proj, points, transl, radius, head_w_point, out=None):
    """proj [F,3,R,R,16], points [F,N,3], transl [F,3] | None -> packed Gaussians [F,N,16]."""
    proj = _contig(proj, "proj")
    points = _contig(points, "points")
    head_w_point = _contig(head_w_point, "head_w_point")
    F, _, R, _, _ = proj.shape
    N = points.shape[1]
    if points.shape[0] != F:
        raise AmavError(f"points has {points.shape[0]} frames, proj has {F}")
    if transl is not None:
        transl = _contig(transl.reshape(F, 3), "transl")
    out = _decode_out(out, F, N, proj.device)
    check(_lib.lib().amav_triplane_sample_decode(F, N, R, proj.data_ptr(), points.data_ptr(),
                                                 transl.data_ptr() if transl is not None else None, float(radius),
                                                 head_w_point.data_ptr(), out.data_ptr(), _stream()),
          "amav_triplane_sample_decode")
    return out


def triplane_sample_decode_indexed(proj, vertices, idx4, transl, radius, head_w_point, out=None):
    """triplane_sample_decode with points_gather fused in: vertices [F,V,3] + idx4 [N,4] instead of points."""
    proj = _contig(proj, "proj")
    vertices = _contig(vertices, "vertices")
    idx4 = _contig(idx4, "idx4", torch.int32)
    head_w_point = _contig(head_w_point, "head_w_point")
    F, _, R, _, _ = proj.shape
    if vertices.shape[0] != F:
        raise AmavError(f"vertices has {vertices.shape[0]} frames, proj has {F}")
    V, N = vertices.shape[1], idx4.shape[0]
    if transl is not None:
        transl = _contig(transl.reshape(F, 3), "transl")
    out = _decode_out(out, F, N, proj.device)
    check(_lib.lib().amav_triplane_sample_decode_indexed(F, N, R, V, proj.data_ptr(), vertices.data_ptr(),
                                                         idx4.data_ptr(),
                                                         transl.data_ptr() if transl is not None else None,
                                                         float(radius), head_w_point.data_ptr(), out.data_ptr(),
                                                         _stream()), "amav_triplane_sample_decode_indexed")
    return out


def triplane_sample_features(planes, points, radius):
    """planes [F,3,C,R,R] (any strides with unit stride along W and R along H), points [F,N,3] -> [F,N,3C]."""
    _need(planes, "planes")
    points = _contig(points, "points")
    F, P, C, R, R2 = planes.shape
    if P != 3 or R != R2:
        raise AmavError(f"planes must be [F,3,C,R,R], got {tuple(planes.shape)}")
    if planes.stride(4) != 1 or planes.stride(3) != R:
        planes = planes.contiguous()
    N = points.shape[1]
    out = torch.empty(F, N, 3 * C, device=planes.device)
    check(_lib.lib().amav_triplane_sample_features(F, N, C, R, planes.data_ptr(), planes.stride(0), planes.stride(1),
                                                   planes.stride(2), points.data_ptr(), float(radius),
                                                   out.data_ptr(), _stream()), "amav_triplane_sample_features")
    return out


# ---------------------------------------------------------------------------------------------- stage-1 reductions
def cell_segments(cell_of, cells):
    """cell_of int [..., N] (cell of every point) -> (order int32 [..., N], seg int32 [..., cells + 1]): point ids
    stably sorted by cell and the offset of every cell's run.  Plumbing (library sort / bincount), feeds the segment
    kernels below."""
    flat = cell_of.reshape(-1, cell_of.shape[-1]).long()
    order = torch.argsort(flat, dim=1, stable=True)
    counts = torch.zeros(flat.shape[0], cells, dtype=torch.long, device=flat.device)
    counts.scatter_add_(1, flat, torch.ones_like(flat))
    seg = torch.zeros(flat.shape[0], cells + 1, dtype=torch.long, device=flat.device)
    seg[:, 1:] = counts.cumsum(1)
    shape = tuple(cell_of.shape[:-1])
    return order.to(torch.int32).reshape(shape + (flat.shape[1],)), seg.to(torch.int32).reshape(shape + (cells + 1,))


def cell_pool_max(feat, cell_of, cells, segments=None):
    """pool_local (triplane_net.py:226-238): feat [B,N,C], cell_of int32 [B,3,N] -> [B,N,C]: for each of the three
    planes the channel-wise maximum over the points that share the point's cell, summed over the planes."""
    feat = _contig(feat, "feat")
    B, N, C = feat.shape
    cell_of = _contig(cell_of, "cell_of", torch.int32)
    if tuple(cell_of.shape) != (B, 3, N):
        raise AmavError(f"cell_of must be int32 [B,3,N] = {(B, 3, N)}, got {tuple(cell_of.shape)}")
    order, seg = segments if segments is not None else cell_segments(cell_of, cells)
    order, seg = _contig(order, "order", torch.int32), _contig(seg, "seg", torch.int32)
    cellmax = torch.empty(B, 3, cells, C, device=feat.device)
    out = torch.empty_like(feat)
    check(_lib.lib().amav_cell_max(B, N, C, int(cells), feat.data_ptr(), order.data_ptr(), seg.data_ptr(),
                                   cellmax.data_ptr(), _stream()), "amav_cell_max")
    check(_lib.lib().amav_cell_gather(B, N, C, int(cells), cellmax.data_ptr(), cell_of.data_ptr(), out.data_ptr(),
                                      _stream()), "amav_cell_gather")
    return out


def cell_splat_mean(feat, cell_of, cells, segments=None):
    """generate_plane_features (triplane_net.py:240-244): feat [B,N,C], cell_of int32 [B,N] -> [B,C,cells]: per-cell
    mean of the points' features (summed in ascending point id), zero for an empty cell."""
    feat = _contig(feat, "feat")
    B, N, C = feat.shape
    cell_of = _contig(cell_of, "cell_of", torch.int32)
    if tuple(cell_of.shape) != (B, N):
        raise AmavError(f"cell_of must be int32 [B,N] = {(B, N)}, got {tuple(cell_of.shape)}")
    order, seg = segments if segments is not None else cell_segments(cell_of, cells)
    order, seg = _contig(order, "order", torch.int32), _contig(seg, "seg", torch.int32)
    out = torch.empty(B, C, cells, device=feat.device)
    check(_lib.lib().amav_cell_mean(B, N, C, int(cells), feat.data_ptr(), order.data_ptr(), seg.data_ptr(),
                                    out.data_ptr(), _stream()), "amav_cell_mean")
    return out


def points_project(points, w2c, intrinsics, features, radius_px):
    """points_projection (graphic_utils.py:275-331): points [B,N,3], w2c [B,4,4], intrinsics [B,3,3], features
    [B,C,H,W] -> [B,N,C] (include/amav.h, amav_points_project)."""
    points, w2c = _contig(points, "points"), _contig(w2c, "w2c")
    intrinsics, features = _contig(intrinsics, "intrinsics"), _contig(features, "features")
    B, N, _ = points.shape
    _, C, H, W = features.shape
    if features.shape[0] != B or tuple(w2c.shape) != (B, 4, 4) or tuple(intrinsics.shape) != (B, 3, 3):
        raise AmavError("points_project: batch sizes / matrix shapes do not match")
    nbytes = _lib.lib().amav_points_project_workspace_bytes(B, N, H, W)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=points.device)
    out = torch.empty(B, N, C, device=points.device)
    check(_lib.lib().amav_points_project(B, N, C, H, W, points.data_ptr(), w2c.data_ptr(), intrinsics.data_ptr(),
                                         features.data_ptr(), float(radius_px), out.data_ptr(), ws.data_ptr(), nbytes,
                                         _stream()), "amav_points_project")
    return out


# ------------------------------------------------------------------------------------------------------ attention
# A list of (Event, Event) pairs: every selfattn() call pops one and records it around its kernels (bench.py)
ATTN_PROFILE_EVENTS = None


def selfattn(q, k, v, heads, scale=None, bounds=None, split_out_exp=None):
    """softmax(q k^T * scale) v for [B,S,H*D] fp32 tensors (row stride may exceed H*D), D = 64.  bounds: (|q|, |k|, |v|)
    upper bounds the caller can prove, which spare the kernel its magnitude pre-pass (include/amav.h,
    amav_selfattn_forward_bounded); None: measured.  split_out_exp: return instead the fp16 x 2 activation operand
    [B*S, 3*H*D] of the projection that follows (split_operand(out, fmt=SPLIT_FP16X2, scale_exp=split_out_exp), written
    by the kernel's own last pass: amav_selfattn_forward_split_out)."""
    for name, t in (("q", q), ("k", k), ("v", v)):
        _need(t, name)
        if t.dim() != 3 or t.stride(2) != 1 or t.stride(0) != t.shape[1] * t.stride(1):
            raise AmavError(f"{name}: need [B,S,H*D] with unit inner stride and dense batch stride")
    B, S, HD = q.shape
    D = HD // heads
    if k.shape != q.shape or v.shape != q.shape or not (q.stride(1) == k.stride(1) == v.stride(1)):
        raise AmavError("selfattn: q, k, v must share shape and row stride")
    out = torch.empty(B, S, HD, device=q.device)
    nbytes = _lib.lib().amav_selfattn_workspace_bytes(B, S, heads, D)
    if nbytes == 0:
        raise AmavError(f"amav_selfattn_workspace_bytes rejected B={B} S={S} H={heads} D={D}")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=q.device)
    ev = ATTN_PROFILE_EVENTS.pop(0) if ATTN_PROFILE_EVENTS else None
    if ev is not None:
        ev[0].record()
    qb, kb, vb = (float(x) for x in bounds) if bounds is not None else (0.0, 0.0, 0.0)
    split = None if split_out_exp is None else _split_buffer(B * S, HD, SPLIT_FP16X2, q.device)
    check(_lib.lib().amav_selfattn_forward_split_out(B, S, heads, D, q.data_ptr(), k.data_ptr(), v.data_ptr(), q.stride(1),
                                                     out.data_ptr(), HD, float(scale if scale is not None else D ** -0.5),
                                                     qb, kb, vb, split.data_ptr() if split is not None else None,
                                                     int(split_out_exp or 0), ws.data_ptr(), nbytes, _stream()),
          "amav_selfattn_forward_split_out")
    if ev is not None:
        ev[1].record()
    return out if split is None else split


_GEMM_WS = {}


def gemm_split_fp16(a, w, alpha=1.0, algo_index=-1):
    """a [rows, k3] fp16 x w [n, k3]^T fp16 -> [rows, n] fp32 (fp32 accumulate, scaled by alpha) through hipBLASLt with
    the kernel named by `algo_index` (-1: the library's own heuristic; include/amav.h, amav_gemm_split_fp16)."""
    a, w = _need(a, "a", torch.float16), _need(w, "w", torch.float16)
    if a.dim() != 2 or w.dim() != 2 or a.shape[1] != w.shape[1] or not a.is_contiguous() or not w.is_contiguous():
        raise AmavError("gemm_split_fp16: need contiguous a [rows, k3] and w [n, k3]")
    out = torch.empty(a.shape[0], w.shape[0], device=a.device)
    ws = _GEMM_WS.get(a.device)
    if ws is None:
        ws = _GEMM_WS[a.device] = torch.empty(32 << 20, dtype=torch.uint8, device=a.device)
    check(_lib.lib().amav_gemm_split_fp16(a.shape[0], w.shape[0], a.shape[1], a.data_ptr(), w.data_ptr(), float(alpha),
                                          out.data_ptr(), int(algo_index), ws.data_ptr(), ws.numel(), _stream()),
          "amav_gemm_split_fp16")
    return out


def gemm_split_fp16_tune(a, w, repeats=10):
    """Times every hipBLASLt kernel on these operands (synchronises; seconds per shape) -> (best index, best ms,
    ms of the library's heuristic choice).  a [copies, rows, k3], w [copies, n, k3] (or 2-D: one set): run i uses set
    i % copies, so with several sets the kernels are timed out of MALL / HBM as inside the transformer step."""
    a, w = _need(a, "a", torch.float16), _need(w, "w", torch.float16)
    if a.dim() == 2:
        a, w = a[None], w[None]
    copies = a.shape[0]
    if w.shape[0] != copies or not a.is_contiguous() or not w.is_contiguous():
        raise AmavError("gemm_split_fp16_tune: need contiguous a [copies, rows, k3], w [copies, n, k3]")
    out = torch.empty(copies, a.shape[1], w.shape[1], device=a.device)
    a, w = a.view(-1, a.shape[-1]), w.view(-1, w.shape[-1])
    rows, n = a.shape[0] // copies, w.shape[0] // copies
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=a.device)
    idx, best, heur = ctypes.c_int32(-1), ctypes.c_float(0), ctypes.c_float(0)
    check(_lib.lib().amav_gemm_split_fp16_tune(rows, n, a.shape[1], a.data_ptr(), w.data_ptr(), out.data_ptr(),
                                               ws.data_ptr(), ws.numel(), int(repeats), int(copies), ctypes.byref(idx),
                                               ctypes.byref(best), ctypes.byref(heur), _stream()), "amav_gemm_split_fp16_tune")
    return int(idx.value), float(best.value), float(heur.value)


def gemm_library_version() -> str:
    return _lib.lib().amav_gemm_library_version().decode()


SPLIT_BF16X3, SPLIT_FP16X2 = 0, 1  # include/amav.h


def _split_buffer(rows, k, fmt, device):
    if fmt == SPLIT_FP16X2:
        return torch.empty(rows, 3 * k, dtype=torch.float16, device=device)
    if fmt == SPLIT_BF16X3:
        return torch.empty(rows, 6 * k, dtype=torch.bfloat16, device=device)
    raise AmavError(f"unknown split format {fmt}")


def geglu(proj, bias=None, split_exp=None):
    """proj [..., 2*inner] (contiguous, fp32) -> [..., inner] = h * gelu(g) (exact erf) with h, g the two halves of
    proj (+ bias [2*inner], when the projection's GEMM ran without it).  With split_exp the result is written as the
    fp16 x 2 activation operand [rows, 3*inner] of the next projection, pre-scaled by 2^split_exp."""
    proj = _need(proj, "proj")
    if not proj.is_contiguous() or proj.shape[-1] % 8:
        raise AmavError("geglu: need a contiguous [..., 2*inner] tensor with inner a multiple of 4")
    inner = proj.shape[-1] // 2
    rows = proj.numel() // proj.shape[-1]
    if split_exp is None:
        out = torch.empty(*proj.shape[:-1], inner, device=proj.device)
    else:
        out = _split_buffer(rows, inner, SPLIT_FP16X2, proj.device)
    bias_ptr = None if bias is None else _shaped(bias, "bias", (2 * inner,)).data_ptr()
    check(_lib.lib().amav_geglu(rows, inner, proj.data_ptr(), proj.shape[-1], bias_ptr,
                                out.data_ptr() if split_exp is None else None,
                                None if split_exp is None else out.data_ptr(), int(split_exp or 0), _stream()),
          "amav_geglu")
    return out


def split_operand(x, weights=False, fmt=SPLIT_BF16X3, scale_exp=0):
    """x [rows, k] fp32 (unit inner stride; k a multiple of 8) -> the activation (default) or weight operand of an
    fp32-equivalent GEMM on the low-precision matrix pipe: [rows, 6 k] bf16 (SPLIT_BF16X3, any finite x) or [rows, 3 k]
    fp16 of x * 2^scale_exp (SPLIT_FP16X2; the caller bounds |x| 2^scale_exp by 32768).  include/amav.h,
    amav_split_operand."""
    x = _need(x, "x")
    if x.dim() != 2 or x.stride(1) != 1 or x.shape[1] % 8 or x.stride(0) % 4 or x.data_ptr() % 16:
        raise AmavError("split_operand: need a 16-byte aligned [rows, k] tensor, unit inner stride, k a multiple of 8")
    out = _split_buffer(x.shape[0], x.shape[1], fmt, x.device)
    check(_lib.lib().amav_split_operand(x.shape[0], x.shape[1], x.data_ptr(), x.stride(0), int(bool(weights)), int(fmt),
                                        int(scale_exp), out.data_ptr(), _stream()), "amav_split_operand")
    return out


def add_layernorm(hidden, add, batch_row, weight, bias, eps=1e-5, add_bias=None, split=None, split_exp=0):
    """hidden [B,S,dim] (contiguous), add [B,S,dim] or None (+ add_bias [dim]: the bias of the projection that produced
    it), batch_row [B,1,dim] or None -> (h = batch_row + ((add + add_bias) + hidden), LayerNorm(h) * weight + bias), two
    new tensors; with split = SPLIT_BF16X3 / SPLIT_FP16X2 the second is the activation operand of the next projection
    (split_operand's layout, pre-scaled by 2^split_exp for fp16) instead of fp32 [B,S,dim].  transformers.py:292-399."""
    hidden = _need(hidden, "hidden")
    if hidden.dim() != 3 or not hidden.is_contiguous():
        raise AmavError("add_layernorm: hidden must be a contiguous [B,S,dim] tensor")
    B, S, dim = hidden.shape
    ptr = lambda t, name, shape: None if t is None else _shaped(t, name, shape).data_ptr()
    h_out = torch.empty_like(hidden)
    out = torch.empty_like(hidden) if split is None else _split_buffer(B * S, dim, split, hidden.device)
    check(_lib.lib().amav_add_layernorm(B * S, dim, S, ptr(add, "add", (B, S, dim)), ptr(add_bias, "add_bias", (dim,)),
                                        ptr(batch_row, "batch_row", (B, 1, dim)), hidden.data_ptr(), h_out.data_ptr(),
                                        _shaped(weight, "weight", (dim,)).data_ptr(),
                                        _shaped(bias, "bias", (dim,)).data_ptr(), float(eps),
                                        out.data_ptr() if split is None else None,
                                        None if split is None else out.data_ptr(), int(split or 0), int(split_exp),
                                        _stream()), "amav_add_layernorm")
    return h_out, out


def _shaped(t, name, shape):
    t = _need(t, name)
    if tuple(t.shape) != tuple(shape) or not t.is_contiguous():
        raise AmavError(f"{name}: expected a contiguous tensor of shape {tuple(shape)}, got {tuple(t.shape)}")
    return t


# -------------------------------------------------------------------------------------------------- point refiner
def cloud_voxelize(points, cloud_of, clouds, resolution=100.0):
    """points [n,3] fp32, cloud_of int32 [n] (ascending) -> (grid int32 [n,3] from the cloud's own origin, cloud_depth
    int32 [clouds]).  point_encoder.py:33 / pointtransformer_v3.py:98-101 per cloud."""
    points, cloud_of = _contig(points, "points"), _contig(cloud_of, "cloud_of", torch.int32)
    n = points.shape[0]
    if points.dim() != 2 or points.shape[1] != 3 or cloud_of.shape != (n,):
        raise AmavError(f"cloud_voxelize: points {tuple(points.shape)} / cloud_of {tuple(cloud_of.shape)}")
    grid = torch.empty(n, 3, dtype=torch.int32, device=points.device)
    depth = torch.empty(clouds, dtype=torch.int32, device=points.device)
    bounds = torch.empty(clouds, 6, dtype=torch.int32, device=points.device)
    check(_lib.lib().amav_cloud_voxelize(n, int(clouds), points.data_ptr(), cloud_of.data_ptr(), float(resolution),
                                         grid.data_ptr(), depth.data_ptr(), bounds.data_ptr(), _stream()),
          "amav_cloud_voxelize")
    return grid, depth


def cloud_codes(grid, cloud_of, cloud_depth):
    """-> keys int64 [4,n] = cloud << 48 | code for the orders z, z-trans, hilbert, hilbert-trans."""
    grid, cloud_of = _contig(grid, "grid", torch.int32), _contig(cloud_of, "cloud_of", torch.int32)
    cloud_depth = _contig(cloud_depth, "cloud_depth", torch.int32)
    n = grid.shape[0]
    keys = torch.empty(4, n, dtype=torch.int64, device=grid.device)
    check(_lib.lib().amav_cloud_codes(n, grid.data_ptr(), cloud_of.data_ptr(), cloud_depth.data_ptr(), keys.data_ptr(),
                                      _stream()), "amav_cloud_codes")
    return keys


def cloud_neighbors(grid, cloud_of, cloud_depth, cloud_start, sorted_keys, order, ksize):
    """-> nbr int32 [n, ksize^3]: the rows a submanifold convolution gathers (-1: empty voxel)."""
    grid, cloud_of = _contig(grid, "grid", torch.int32), _contig(cloud_of, "cloud_of", torch.int32)
    cloud_depth, cloud_start = _contig(cloud_depth, "cloud_depth", torch.int32), _contig(cloud_start, "cloud_start", torch.int32)
    sorted_keys, order = _contig(sorted_keys, "sorted_keys", torch.int64), _contig(order, "order", torch.int64)
    n = grid.shape[0]
    if sorted_keys.shape != (n,) or order.shape != (n,) or cloud_start.shape[0] != cloud_depth.shape[0] + 1:
        raise AmavError("cloud_neighbors: sorted_keys / order must be [n], cloud_start [clouds + 1]")
    nbr = torch.empty(n, ksize ** 3, dtype=torch.int32, device=grid.device)
    check(_lib.lib().amav_cloud_neighbors(n, int(ksize), grid.data_ptr(), cloud_of.data_ptr(), cloud_depth.data_ptr(),
                                          cloud_start.data_ptr(), sorted_keys.data_ptr(), order.data_ptr(),
                                          nbr.data_ptr(), _stream()), "amav_cloud_neighbors")
    return nbr


def subm_pair_gemm(feat, pair_src, tap_start, tile_start, tiles, weights):
    """feat [n,C_in], pairs grouped by tap (pair_src int32 [P], tap_start / tile_start int32 [taps+1]), weights
    [taps,C_in,C_out] -> products [P,C_out] (include/amav.h)."""
    feat, weights = _contig(feat, "feat"), _contig(weights, "weights")
    pair_src = _contig(pair_src, "pair_src", torch.int32)
    tap_start, tile_start = _contig(tap_start, "tap_start", torch.int32), _contig(tile_start, "tile_start", torch.int32)
    taps, cin, cout = weights.shape
    if feat.shape[1] != cin or tap_start.shape != (taps + 1,) or tile_start.shape != (taps + 1,):
        raise AmavError("subm_pair_gemm: shapes do not match")
    products = torch.empty(pair_src.shape[0], cout, device=feat.device)
    check(_lib.lib().amav_subm_pair_gemm(pair_src.shape[0], int(tiles), taps, cin, cout, feat.data_ptr(),
                                         pair_src.data_ptr(), tap_start.data_ptr(), tile_start.data_ptr(),
                                         weights.data_ptr(), products.data_ptr(), _stream()), "amav_subm_pair_gemm")
    return products


def subm_prepare_weights_split(weights):
    """weights [taps,C_in,C_out] fp32 -> uint8 device buffer: two scaled fp16 parts in MFMA fragment order (include/amav.h,
    amav_subm_prepare_weights_split), the operand of subm_pair_gemm(..., weights_split=...)."""
    weights = _contig(weights, "weights")
    taps, cin, cout = weights.shape
    nbytes = _lib.lib().amav_subm_weights_split_bytes(taps, cin, cout)
    if nbytes == 0:
        raise AmavError(f"subm_prepare_weights_split: channels must be multiples of 32, got {cin} -> {cout}")
    out = torch.empty(nbytes, dtype=torch.uint8, device=weights.device)
    check(_lib.lib().amav_subm_prepare_weights_split(taps, cin, cout, weights.data_ptr(), out.data_ptr(), nbytes, _stream()),
          "amav_subm_prepare_weights_split")
    return out


def subm_pair_gemm_split(feat, pair_src, tap_start, tile_start, tiles, weights_split, taps, cout):
    """subm_pair_gemm on the 16-bit matrix pipe: weights_split from subm_prepare_weights_split ([taps,C_in,cout])."""
    feat = _contig(feat, "feat")
    pair_src = _contig(pair_src, "pair_src", torch.int32)
    tap_start, tile_start = _contig(tap_start, "tap_start", torch.int32), _contig(tile_start, "tile_start", torch.int32)
    n, cin = feat.shape
    if tap_start.shape != (taps + 1,) or tile_start.shape != (taps + 1,):
        raise AmavError("subm_pair_gemm_split: shapes do not match")
    need = _lib.lib().amav_subm_weights_split_bytes(taps, cin, cout)
    if need == 0 or _need(weights_split, "weights_split", torch.uint8).numel() != need:
        raise AmavError("subm_pair_gemm_split: weights_split was not prepared for these shapes")
    products = torch.empty(pair_src.shape[0], cout, device=feat.device)
    scratch = torch.empty(4, dtype=torch.int32, device=feat.device)
    check(_lib.lib().amav_subm_pair_gemm_split(pair_src.shape[0], int(tiles), taps, cin, cout, n, feat.data_ptr(),
                                               pair_src.data_ptr(), tap_start.data_ptr(), tile_start.data_ptr(),
                                               weights_split.data_ptr(), scratch.data_ptr(), products.data_ptr(),
                                               _stream()), "amav_subm_pair_gemm_split")
    return products


def subm_pair_sum(products, pair_of, bias=None):
    """products [P,C_out], pair_of int32 [n,taps] (-1: no voxel) -> [n,C_out] = bias + sum over taps in tap order."""
    products, pair_of = _contig(products, "products"), _contig(pair_of, "pair_of", torch.int32)
    n, taps = pair_of.shape
    cout = products.shape[1]
    out = torch.empty(n, cout, device=products.device)
    check(_lib.lib().amav_subm_pair_sum(n, taps, cout, products.data_ptr(), pair_of.data_ptr(),
                                        None if bias is None else _shaped(bias, "bias", (cout,)).data_ptr(), out.data_ptr(),
                                        _stream()), "amav_subm_pair_sum")
    return out


def patch_attention(qkv, order, patch_desc, heads, max_patch, scale=None):
    """qkv [n, 3*C] (q | k | v), order int64 [n], patch_desc int32 [patches,4] -> [n, C] (include/amav.h)."""
    qkv, order = _contig(qkv, "qkv"), _contig(order, "order", torch.int64)
    patch_desc = _contig(patch_desc, "patch_desc", torch.int32)
    n, C3 = qkv.shape
    C = C3 // 3
    D = C // heads
    if C3 != 3 * heads * D or order.shape != (n,) or patch_desc.dim() != 2 or patch_desc.shape[1] != 4:
        raise AmavError("patch_attention: shapes do not match")
    out = torch.empty(n, C, device=qkv.device)
    check(_lib.lib().amav_patch_attention(patch_desc.shape[0], int(max_patch), int(heads), D, qkv.data_ptr(),
                                          order.data_ptr(), patch_desc.data_ptr(), out.data_ptr(),
                                          float(scale if scale is not None else D ** -0.5), _stream()),
          "amav_patch_attention")
    return out


def cluster_max(x, members, seg, scale, shift):
    """x [n,C], members int64 [n] (rows grouped by cluster), seg int64 [clusters+1] -> gelu(max * scale + shift) [clusters,C]."""
    x, members, seg = _contig(x, "x"), _contig(members, "members", torch.int64), _contig(seg, "seg", torch.int64)
    C = x.shape[1]
    clusters = seg.shape[0] - 1
    out = torch.empty(clusters, C, device=x.device)
    check(_lib.lib().amav_cluster_max(clusters, C, x.data_ptr(), members.data_ptr(), seg.data_ptr(),
                                      _shaped(scale, "scale", (C,)).data_ptr(), _shaped(shift, "shift", (C,)).data_ptr(),
                                      out.data_ptr(), _stream()), "amav_cluster_max")
    return out


def bn_gelu(x, scale, shift):
    """gelu(x * scale + shift) for x [rows, C] (BatchNorm in eval mode folded to scale / shift)."""
    x = _contig(x, "x")
    rows, C = x.shape
    out = torch.empty_like(x)
    check(_lib.lib().amav_bn_gelu(rows, C, x.data_ptr(), _shaped(scale, "scale", (C,)).data_ptr(),
                                  _shaped(shift, "shift", (C,)).data_ptr(), out.data_ptr(), _stream()), "amav_bn_gelu")
    return out


def unpool_merge(x, scale, shift, up, cluster):
    """-> (skip = gelu(x * scale + shift), skip + up[cluster]) for x [n,C], up [m,C], cluster int64 [n]."""
    x, up, cluster = _contig(x, "x"), _contig(up, "up"), _contig(cluster, "cluster", torch.int64)
    n, C = x.shape
    if up.shape[1] != C or cluster.shape != (n,):
        raise AmavError("unpool_merge: shapes do not match")
    skip, total = torch.empty_like(x), torch.empty_like(x)
    check(_lib.lib().amav_unpool_merge(n, C, x.data_ptr(), _shaped(scale, "scale", (C,)).data_ptr(),
                                       _shaped(shift, "shift", (C,)).data_ptr(), up.data_ptr(), cluster.data_ptr(),
                                       skip.data_ptr(), total.data_ptr(), _stream()), "amav_unpool_merge")
    return skip, total


def rows_norm(x, base, norm_b, norm_a=None):
    """-> (s = base + (LayerNorm_a(x) if norm_a else x), LayerNorm_b(s)) for [n, C] rows, C in {32,...,512};
    norm_a / norm_b are nn.LayerNorm modules (weight, bias, eps)."""
    x, base = _contig(x, "x"), _contig(base, "base")
    n, C = x.shape
    if base.shape != x.shape:
        raise AmavError("rows_norm: x and base must have the same shape")
    out_sum, out_norm = torch.empty_like(x), torch.empty_like(x)
    ptr = lambda t: _shaped(t.detach(), "norm parameter", (C,)).data_ptr()
    check(_lib.lib().amav_rows_norm(n, C, x.data_ptr(), base.data_ptr(),
                                    None if norm_a is None else ptr(norm_a.weight), None if norm_a is None else ptr(norm_a.bias),
                                    ptr(norm_b.weight), ptr(norm_b.bias), float(norm_b.eps), out_sum.data_ptr(),
                                    out_norm.data_ptr(), _stream()), "amav_rows_norm")
    return out_sum, out_norm
