"""ctypes binding of libamav_hip.so (the C ABI declared in include/amav.h).

There is no CPU or eager fallback: if the library is missing, or an entry point fails, an exception is raised.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# AMAV_LIB: another build of the SAME library (tools/ablate_render.sh builds diagnostic variants of the blend kernel
# with -DAMAV_ABLATE=n); never a different implementation -- the symbol table is checked against include/amav.h either way
LIB_PATH = os.environ.get("AMAV_LIB") or os.path.join(_HERE, "csrc", "libamav_hip.so")

c_float_p = ctypes.c_void_p  # device pointers travel as integers (tensor.data_ptr())


class AmavError(RuntimeError):
    pass


class Attr(ctypes.Structure):
    """amav_attr: element (f, i) at ptr[f * frame_stride + i * elem_stride] (strides in floats)."""

    _fields_ = [("ptr", ctypes.c_void_p), ("frame_stride", ctypes.c_int64), ("elem_stride", ctypes.c_int32),
                ("_pad", ctypes.c_int32)]


class RasterArgs(ctypes.Structure):
    _fields_ = [
        ("num_frames", ctypes.c_int32), ("num_gaussians", ctypes.c_int32), ("height", ctypes.c_int32),
        ("width", ctypes.c_int32),
        ("means3d", Attr), ("rotations", Attr), ("scales", Attr), ("opacities", Attr), ("colors", Attr),
        ("viewmatrix", ctypes.c_void_p), ("projmatrix", ctypes.c_void_p), ("tanfov", ctypes.c_void_p),
        ("bg", ctypes.c_float * 3), ("scale_modifier", ctypes.c_float),
        ("apply_activations", ctypes.c_int32),
        ("scale_bias", ctypes.c_float), ("scale_max", ctypes.c_float), ("opacity_bias", ctypes.c_float),
        ("antialiasing", ctypes.c_int32), ("clamp_output", ctypes.c_int32),
        ("out_rgba", ctypes.c_void_p), ("out_inv_depth", ctypes.c_void_p), ("out_radii", ctypes.c_void_p),
        ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_size_t),
        ("instance_capacity", ctypes.c_int64),
        ("profile_start_event", ctypes.c_void_p), ("profile_stop_event", ctypes.c_void_p),
        ("debug_stamps", ctypes.c_void_p),
        ("wire", ctypes.c_void_p), ("wire_bytes", ctypes.c_size_t), ("wire_capacity_tiles", ctypes.c_int64),
    ]


class BodyTables(ctypes.Structure):
    _fields_ = [
        ("num_verts", ctypes.c_int32), ("num_joints", ctypes.c_int32), ("num_coeffs", ctypes.c_int32),
        ("skin_k", ctypes.c_int32),
        ("v_template", ctypes.c_void_p), ("blend", ctypes.c_void_p), ("j_template", ctypes.c_void_p),
        ("j_dirs", ctypes.c_void_p), ("parents", ctypes.c_void_p), ("skin_idx", ctypes.c_void_p),
        ("skin_w", ctypes.c_void_p), ("blend_split", ctypes.c_void_p),
    ]


class PoseParts(ctypes.Structure):
    _fields_ = [
        ("num_pose_parts", ctypes.c_int32), ("num_coeff_parts", ctypes.c_int32),
        ("pose", ctypes.c_void_p * 8), ("pose_joints", ctypes.c_int32 * 8), ("pose_stride", ctypes.c_int64 * 8),
        ("pose_mean", ctypes.c_void_p),
        ("coeff", ctypes.c_void_p * 4), ("coeff_count", ctypes.c_int32 * 4), ("coeff_stride", ctypes.c_int64 * 4),
    ]


# name -> (restype, argtypes); every symbol include/amav.h declares
SIGNATURES = {
    "amav_version": (ctypes.c_char_p, []),
    "amav_set_option": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_char_p]),
    "amav_gemm_split_fp16": (ctypes.c_int, [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_float, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t,
                                            ctypes.c_void_p]),
    "amav_gemm_split_fp16_tune": (ctypes.c_int, [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                                 ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                                 ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_float),
                                                 ctypes.POINTER(ctypes.c_float), ctypes.c_void_p]),
    "amav_gemm_library_version": (ctypes.c_char_p, []),
    "amav_last_error": (ctypes.c_char_p, []),
    "amav_device_count": (ctypes.c_int, []),
    "amav_event_create": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p)]),
    "amav_event_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "amav_event_record": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "amav_event_elapsed_ms": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]),
    "amav_camera_from_intrinsics": (ctypes.c_int, [ctypes.c_int, c_float_p, c_float_p, ctypes.c_int, ctypes.c_int,
                                                   ctypes.c_float, ctypes.c_float, c_float_p, c_float_p, c_float_p,
                                                   c_float_p, ctypes.c_void_p]),
    "amav_rasterize_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                         ctypes.c_int64]),
    "amav_rasterize_forward": (ctypes.c_int, [ctypes.POINTER(RasterArgs), ctypes.c_void_p]),
    "amav_rasterize_status": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64),
                                             ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int32),
                                             ctypes.c_void_p]),
    "amav_frames_to_rgb8": (ctypes.c_int, [ctypes.c_int64, c_float_p, ctypes.c_void_p, ctypes.c_void_p]),
    "amav_add_layernorm": (ctypes.c_int, [ctypes.c_int64, ctypes.c_int, ctypes.c_int64, c_float_p, c_float_p, c_float_p,
                                          c_float_p, c_float_p, c_float_p, c_float_p, ctypes.c_float, c_float_p,
                                          ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "amav_geglu": (ctypes.c_int, [ctypes.c_int64, ctypes.c_int, c_float_p, ctypes.c_int64, c_float_p, c_float_p,
                                  ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "amav_split_operand": (ctypes.c_int, [ctypes.c_int64, ctypes.c_int, c_float_p, ctypes.c_int64, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "amav_frames_wire_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64]),
    "amav_frames_pack_tiles": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_float_p,
                                              ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_int64,
                                              ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "amav_rasterize_tile_counts": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                  ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
    "amav_frames_unpack_tiles": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64,
                                                ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p,
                                                ctypes.c_void_p]),
    "amav_frames_unpack_tiles_delta": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                      ctypes.c_int64, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "amav_cell_max": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_float_p, ctypes.c_void_p,
                                     ctypes.c_void_p, c_float_p, ctypes.c_void_p]),
    "amav_cell_gather": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_float_p,
                                        ctypes.c_void_p, c_float_p, ctypes.c_void_p]),
    "amav_cell_mean": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_float_p,
                                      ctypes.c_void_p, ctypes.c_void_p, c_float_p, ctypes.c_void_p]),
    "amav_points_project_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "amav_points_project": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           c_float_p, c_float_p, c_float_p, c_float_p, ctypes.c_float, c_float_p,
                                           ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "amav_cloud_voxelize": (ctypes.c_int, [ctypes.c_int64, ctypes.c_int, c_float_p, ctypes.c_void_p, ctypes.c_float,
                                           ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "amav_cloud_codes": (ctypes.c_int, [ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_void_p, ctypes.c_void_p]),
    "amav_cloud_neighbors": (ctypes.c_int, [ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.c_void_p]),
    "amav_subm_pair_gemm": (ctypes.c_int, [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           c_float_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, c_float_p,
                                           c_float_p, ctypes.c_void_p]),
    "amav_subm_weights_split_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "amav_subm_prepare_weights_split": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_float_p, ctypes.c_void_p,
                                                       ctypes.c_size_t, ctypes.c_void_p]),
    "amav_subm_pair_gemm_split": (ctypes.c_int, [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                 ctypes.c_int64, c_float_p, ctypes.c_void_p, ctypes.c_void_p,
                                                 ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, c_float_p,
                                                 ctypes.c_void_p]),
    "amav_subm_pair_sum": (ctypes.c_int, [ctypes.c_int64, ctypes.c_int, ctypes.c_int, c_float_p, ctypes.c_void_p,
                                          c_float_p, c_float_p, ctypes.c_void_p]),
    "amav_patch_attention": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_float_p,
                                            ctypes.c_void_p, ctypes.c_void_p, c_float_p, ctypes.c_float,
                                            ctypes.c_void_p]),
    "amav_cluster_max": (ctypes.c_int, [ctypes.c_int64, ctypes.c_int, c_float_p, ctypes.c_void_p, ctypes.c_void_p,
                                        c_float_p, c_float_p, c_float_p, ctypes.c_void_p]),
    "amav_bn_gelu": (ctypes.c_int, [ctypes.c_int64, ctypes.c_int, c_float_p, c_float_p, c_float_p, c_float_p,
                                    ctypes.c_void_p]),
    "amav_rows_norm": (ctypes.c_int, [ctypes.c_int64, ctypes.c_int, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p,
                                      c_float_p, ctypes.c_float, c_float_p, c_float_p, ctypes.c_void_p]),
    "amav_unpool_merge": (ctypes.c_int, [ctypes.c_int64, ctypes.c_int, c_float_p, c_float_p, c_float_p, c_float_p,
                                         ctypes.c_void_p, c_float_p, c_float_p, ctypes.c_void_p]),
    "amav_lbs_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.POINTER(BodyTables)]),
    "amav_lbs_blend_split_bytes": (ctypes.c_size_t, [ctypes.POINTER(BodyTables)]),
    "amav_lbs_prepare_blend_split": (ctypes.c_int, [ctypes.POINTER(BodyTables), ctypes.c_void_p, ctypes.c_size_t,
                                                    ctypes.c_void_p]),
    "amav_lbs_forward": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(BodyTables), c_float_p, c_float_p, c_float_p,
                                        c_float_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "amav_lbs_forward_parts": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(BodyTables), ctypes.POINTER(PoseParts), c_float_p,
                                              c_float_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "amav_points_gather": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_float_p, ctypes.c_void_p,
                                          c_float_p, ctypes.c_void_p]),
    "amav_points_bbox": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, c_float_p, c_float_p, ctypes.c_void_p]),
    "amav_triplane_project_region": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_float_p, ctypes.c_int64,
                                                    c_float_p, c_float_p, c_float_p, ctypes.c_float, ctypes.c_void_p]),
    "amav_triplane_project": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_float_p, ctypes.c_int64,
                                             c_float_p, c_float_p, ctypes.c_void_p]),
    "amav_triplane_sample_decode": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_float_p, c_float_p,
                                                   c_float_p, ctypes.c_float, c_float_p, c_float_p,
                                                   ctypes.c_void_p]),
    "amav_triplane_sample_decode_indexed": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                           c_float_p, c_float_p, ctypes.c_void_p, c_float_p,
                                                           ctypes.c_float, c_float_p, c_float_p, ctypes.c_void_p]),
    "amav_triplane_sample_features": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                     c_float_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                                     c_float_p, ctypes.c_float, c_float_p, ctypes.c_void_p]),
    "amav_selfattn_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "amav_selfattn_forward": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_float_p,
                                             c_float_p, c_float_p, ctypes.c_int64, c_float_p, ctypes.c_int64,
                                             ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "amav_selfattn_forward_bounded": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_float_p,
                                                     c_float_p, c_float_p, ctypes.c_int64, c_float_p, ctypes.c_int64,
                                                     ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                                     ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "amav_selfattn_forward_split_out": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_float_p,
                                                     c_float_p, c_float_p, ctypes.c_int64, c_float_p, ctypes.c_int64,
                                                     ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                                     ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
}

_lib = None


def lib():
    """Load libamav_hip.so once.  Raises AmavError when it has not been built (run __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AmavError(
                f"{LIB_PATH} is missing: the HIP extension is the only execution path of this package "
                "(no CPU fallback). Build it with `python -c 'import __graft_entry__ as g; g.build()'`.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError here = header and library out of sync
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().amav_last_error().decode("utf-8", "replace")
        raise AmavError(f"{what or 'amav call'} failed ({rc}): {msg}")
