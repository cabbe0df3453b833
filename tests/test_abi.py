"""The C-ABI library loads and exports every symbol include/amav.h declares; host-side argument checks answer with
error codes (no kernel is launched here: there is no GPU in this container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as entry
    from audio_motion_avatar_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        entry.build()
    return _lib.lib()


def header_symbols():
    text = open(os.path.join(ROOT, "include", "amav.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(amav_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from audio_motion_avatar_amd import _lib

    names = header_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/amav.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes binding and header disagree"


def test_struct_layouts_match_the_header(lib):
    """Field order of the ctypes mirrors follows the header (sizes are what the compiler lays out for it)."""
    from audio_motion_avatar_amd import _lib

    text = open(os.path.join(ROOT, "include", "amav.h")).read()
    body = re.search(r"typedef struct amav_raster_args \{(.*?)\} amav_raster_args;", text, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if decl:
            fields += [re.sub(r"\[.*\]", "", f).strip(" *") for f in decl.split(None, 1)[1].replace("*", " ").split(",")]
    fields = [f.split()[-1] if " " in f else f for f in fields]
    assert [f for f, _ in _lib.RasterArgs._fields_] == fields
    assert ctypes.sizeof(_lib.Attr) == 24 and ctypes.sizeof(_lib.BodyTables) == 16 + 8 * 8


def test_version_and_error_reporting(lib):
    assert lib.amav_version().startswith(b"amav-hip")
    assert lib.amav_rasterize_forward(None, None) == -1                      # AMAV_ERR_INVALID_ARG
    assert b"args is NULL" in lib.amav_last_error()
    assert lib.amav_rasterize_workspace_bytes(0, 10, 16, 16, 100) == 0       # rejected sizes
    assert lib.amav_rasterize_workspace_bytes(2, 100, 64, 64, 1000) > 2 * 100 * 64
    assert lib.amav_lbs_forward(1, None, None, None, None, None, None, 0, None) == -1
    assert lib.amav_triplane_project(0, 4, 4, None, 0, None, None, None) == -1
    assert lib.amav_frames_to_rgb8(3, None, None, None) == -1                # not a multiple of 4
    assert lib.amav_points_gather(1, 1, 1, None, None, None, None) == -1


def test_split_product_entry_points_validate_before_touching_the_device(lib):
    """Argument checks and size queries of the split-product entry points run on the host (no launch on a bad call)."""
    import ctypes

    from audio_motion_avatar_amd import _lib

    fake = 4096  # a non-NULL, 16-byte aligned address that is never dereferenced on these paths
    # operand split: k must be a multiple of 8, formats 0 / 1, fp16 exponent in range
    assert lib.amav_split_operand(4, 12, fake, 12, 0, 0, 0, fake, None) == -1 and b"multiple of 8" in lib.amav_last_error()
    assert lib.amav_split_operand(4, 16, fake, 16, 0, 7, 0, fake, None) == -1 and b"format" in lib.amav_last_error()
    assert lib.amav_split_operand(4, 16, fake, 16, 0, 1, 500, fake, None) == -1
    assert lib.amav_split_operand(4, 16, None, 16, 0, 0, 0, fake, None) == -1
    # GEGLU / residual-LayerNorm: exactly one of the fp32 and the split output
    assert lib.amav_geglu(4, 16, fake, 32, None, fake, fake, 0, None) == -1 and b"exactly one" in lib.amav_last_error()
    assert lib.amav_geglu(4, 16, fake, 32, None, None, None, 0, None) == -1
    assert lib.amav_add_layernorm(4, 512, 4, None, None, None, fake, fake, fake, fake, 1e-5, fake, fake, 0, 0, None) == -1
    assert b"exactly one" in lib.amav_last_error()
    assert lib.amav_add_layernorm(4, 512, 4, None, fake, None, fake, fake, fake, fake, 1e-5, fake, None, 0, 0, None) == -1
    assert b"add_bias without add" in lib.amav_last_error()
    # attention: all three bounds or none
    assert lib.amav_selfattn_forward_bounded(1, 64, 1, 64, fake, fake, fake, 64, fake, 64, 0.125, 1.0, 0.0, 1.0, fake, 1 << 30,
                                             None) == -1
    assert b"all three bounds" in lib.amav_last_error()
    # prepared operands: sizes = header + two fp16 parts of every entry
    assert lib.amav_subm_weights_split_bytes(27, 64, 128) == 256 + 27 * 64 * 128 * 2 * 2
    assert lib.amav_subm_weights_split_bytes(27, 48, 128) == 0 and lib.amav_subm_weights_split_bytes(0, 64, 64) == 0
    assert lib.amav_subm_prepare_weights_split(27, 64, 128, fake, fake, 16, None) == -3        # AMAV_ERR_WORKSPACE
    t = _lib.BodyTables()
    t.num_verts, t.num_joints, t.num_coeffs, t.skin_k = 10475, 55, 20, 4
    for name in ("v_template", "blend", "j_template", "j_dirs", "parents", "skin_idx", "skin_w"):
        setattr(t, name, fake)
    kb = 20 + 54 * 9  # 506 blend rows, padded to 512: 32 k-steps of 16 per 32-vertex tile
    assert lib.amav_lbs_blend_split_bytes(ctypes.byref(t)) == 256 + 328 * 32 * 2 * 3 * 512 * 2 and kb == 506
    assert lib.amav_lbs_prepare_blend_split(ctypes.byref(t), fake + 8, 1 << 30, None) == -1    # not 256-byte aligned


def test_product_refuses_to_run_without_a_device():
    """No CPU fallback: a CPU tensor is an error, not a slow path."""
    import torch

    from audio_motion_avatar_amd import AmavError, ops

    with pytest.raises(AmavError, match="only runs on an MI355X"):
        ops.lbs_forward({}, torch.zeros(1, 165), torch.zeros(1, 20))
    with pytest.raises(AmavError):
        ops.triplane_project(torch.zeros(1, 4, 12), torch.zeros(3, 4, 16), 2)


def test_missing_library_fails_loudly(monkeypatch):
    from audio_motion_avatar_amd import AmavError, _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libamav_hip.so")
    with pytest.raises(AmavError, match="no CPU fallback"):
        _lib.lib()
