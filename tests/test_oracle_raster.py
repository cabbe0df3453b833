"""Pins the rasterizer oracle without reference vectors (none exist): analytic known answers,
an independent second restatement, the fp64 build, and the committed golden fixtures."""
import math
import os

import numpy as np
import pytest
import torch

from helpers import oracle_frames, random_scene
from oracle import camera, rasterizer as R

GOLD = os.path.join(os.path.dirname(__file__), "golden")
K64 = torch.tensor([[64.0, 0, 32.0], [0, 64.0, 32.0], [0, 0, 1]], dtype=torch.float64)


def one(xyz, scale, opacity, color, bg=(1, 1, 1), H=64, W=64, rot=None, dtype=np.float64, K=K64):
    n = len(xyz)
    rot = rot if rot is not None else [[1, 0, 0, 0]] * n
    view, proj, tx, ty, _ = camera.camera_setup(K, torch.eye(4, dtype=torch.float64), H, W)
    return R.rasterize_c(np.array(xyz, float), np.array(rot, float), np.array(scale, float),
                         np.array(opacity, float).reshape(-1, 1), np.array(color, float), view.numpy(), proj.numpy(), tx,
                         ty, bg, H, W, dtype=dtype)


def test_single_centred_gaussian_closed_form():
    """Gaussian projecting exactly onto pixel (32,32): colour = a*c + (1-a)*bg with a = min(0.99, opacity)."""
    # pixel centre convention: px = fx*x/z + cx - 0.5  ->  x = 0.5 * z / fx puts the mean on pixel 32
    z = 2.0
    x = 0.5 * z / 64.0
    o = one([[x, x, z]], [[0.02] * 3], [0.6], [[0.2, 0.5, 0.9]], bg=(1, 1, 1))
    a = 0.6
    assert o["radii"][0] > 0
    np.testing.assert_allclose(o["color"][:, 32, 32], [a * 0.2 + 0.4, a * 0.5 + 0.4, a * 0.9 + 0.4], atol=1e-12)
    np.testing.assert_allclose(o["alpha"][32, 32], a, atol=1e-12)
    np.testing.assert_allclose(o["inv_depth"][32, 32], a / z, atol=1e-12)
    # sigma_px = 0.02 * 64 / 2 = 0.64 px (+0.3 low-pass): one pixel away alpha = o * exp(-0.5 / (0.64^2 + 0.3))
    var = 0.64 ** 2 + 0.3
    # (rtol 1e-4: the mean sits 0.5 px off the optical axis, so the EWA Jacobian adds a (x/z)^2 ~ 6e-5 term)
    np.testing.assert_allclose(o["alpha"][32, 33], a * math.exp(-0.5 / var), rtol=1e-4)
    # radius: lambda_max = mid + sqrt(max(0.1, mid^2 - det)); isotropic => the 0.1 floor applies
    assert o["radii"][0] == math.ceil(3 * math.sqrt(var + math.sqrt(0.1)))


def test_opacity_is_capped_at_099():
    z = 2.0
    x = 0.5 * z / 64.0
    o = one([[x, x, z]], [[0.02] * 3], [1.0], [[0, 0, 0]])
    np.testing.assert_allclose(o["alpha"][32, 32], 0.99, atol=1e-12)


def test_front_to_back_order_by_depth_not_by_index():
    z = 2.0
    far = one([[0.5 * 3 / 64, 0.5 * 3 / 64, 3.0], [0.5 * z / 64, 0.5 * z / 64, z]], [[0.05] * 3] * 2, [0.5, 0.5],
              [[1, 0, 0], [0, 0, 1]], bg=(0, 0, 0))
    # the nearer (index 1, blue) is blended first: C = 0.5*blue + 0.5*0.5*red
    np.testing.assert_allclose(far["color"][:, 32, 32], [0.25, 0, 0.5], atol=1e-9)
    np.testing.assert_allclose(far["alpha"][32, 32], 0.75, atol=1e-9)


def test_alpha_below_1_over_255_is_skipped():
    z = 2.0
    x = 0.5 * z / 64.0
    o = one([[x, x, z]], [[0.02] * 3], [0.5 / 255.0], [[0, 0, 0]])
    assert o["radii"][0] > 0 and o["instances"] > 0
    assert np.count_nonzero(o["alpha"]) == 0
    np.testing.assert_array_equal(o["color"], np.ones_like(o["color"]))


def test_transmittance_stop_excludes_the_saturating_gaussian():
    """Three opaque layers: T = 0.01, 1e-4 ... the Gaussian that would push T below 1e-4 is NOT blended."""
    z = [2.0, 2.1, 2.2]
    xyz = [[0.5 * d / 64, 0.5 * d / 64, d] for d in z]
    o = one(xyz, [[0.05] * 3] * 3, [1.0, 1.0, 1.0], [[1, 0, 0], [0, 1, 0], [0, 0, 1]], bg=(0, 0, 0))
    # after two layers T = 0.01 * 0.01 = 1e-4 (not < 1e-4, so blended); the third gives 1e-6 < 1e-4 -> stop
    np.testing.assert_allclose(o["color"][:, 32, 32], [0.99, 0.99 * 0.01, 0.0], atol=1e-9)
    np.testing.assert_allclose(o["alpha"][32, 32], 1 - 1e-4, atol=1e-9)


def test_near_plane_cull_and_empty_scene():
    o = one([[0, 0, 0.2], [0, 0, -1.0]], [[0.05] * 3] * 2, [0.9, 0.9], [[0, 0, 0]] * 2)
    assert o["instances"] == 0 and not o["radii"].any()
    np.testing.assert_array_equal(o["color"], np.ones_like(o["color"]))


def test_tile_rectangle_limits_contribution():
    """A pixel outside the Gaussian's 3-sigma tile rectangle gets nothing even where alpha would exceed 1/255."""
    z = 2.0
    x = (15.5 + 0.5) * z / 64.0 - 32 * z / 64.0  # mean on pixel (15.5+..): near a tile border
    o = one([[x, 0.5 * z / 64, z]], [[0.16] * 3], [0.99], [[0, 0, 0]], H=64, W=64)
    r = int(o["radii"][0])
    px = 16.0 - 0.5 + 0.5  # fx*x/z + cx - 0.5
    first_tile_out = int((px + r + 15) // 16)
    if first_tile_out < 4:
        col = first_tile_out * 16
        assert np.count_nonzero(o["alpha"][:, col:]) == 0


@pytest.mark.parametrize("seed,N,H,W", [(1, 150, 48, 64), (2, 300, 64, 80)])
def test_c_oracle_matches_independent_torch_restatement(seed, N, H, W):
    s = random_scene(seed, N, H, W, 1)
    view, proj, tx, ty, _ = camera.camera_setup(s["K"][0].double(), s["E"][0].double(), H, W)
    c = R.rasterize_c(s["xyz"][0], s["rot"][0], s["scale"][0], s["opacity"][0], s["color"][0], view, proj, tx, ty,
                      [1, 1, 1], H, W, dtype=np.float64)
    t = R.rasterize_torch(*[s[k][0].double() for k in ("xyz", "rot", "scale", "opacity", "color")], view, proj, tx, ty,
                          [1, 1, 1], H, W)
    assert np.abs(c["color"] - t["color"].numpy()).max() < 1e-12
    assert np.abs(c["alpha"] - t["alpha"].numpy()).max() < 1e-12
    assert np.array_equal(c["radii"], t["radii"].numpy())


def test_fp32_build_tracks_fp64_build():
    s = random_scene(5, 400, 64, 64, 1)
    a, b = oracle_frames(s, np.float32)[0], oracle_frames(s, np.float64)[0]
    stable = (a["unstable"] == 0) & (b["unstable"] == 0)
    assert stable.mean() > 0.99
    assert (np.abs(a["color"] - b["color"]) * stable).max() < 1e-4
    assert np.array_equal(a["radii"], b["radii"])


@pytest.mark.parametrize("name", ["raster_64.npz", "raster_256.npz"])
def test_golden_fixture(name):
    g = np.load(os.path.join(GOLD, name))
    scene = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("in_")}
    scene["H"], scene["W"] = int(g["H"]), int(g["W"])
    for f, o in enumerate(oracle_frames(scene, np.float32)):
        stable = g["unstable"][f] == 0
        assert (np.abs(o["color"] - g["color"][f]) * stable).max() < 1e-4
        assert (np.abs(o["alpha"] - g["alpha"][f]) * stable).max() < 1e-4
        assert np.array_equal(o["radii"], g["radii"][f])
        assert o["instances"] == g["instances"][f]


def test_world2view_double_inverse_is_the_extrinsic():
    """graphic_utils.py:67-78 inverts [R^T|t] twice: numerically E."""
    s = random_scene(9, 1, 64, 64, 3)
    for f in range(3):
        E = s["E"][f].double()
        got = camera.world2view2(E[:3, :3].T.contiguous().T.T, E[:3, 3])
        assert torch.allclose(got, E, atol=1e-12)


def test_render_batch_restatement_shapes_and_activation():
    g = torch.Generator().manual_seed(3)
    N = 200
    gauss = dict(xyz=torch.randn(2, N, 3, generator=g) * 0.2 + torch.tensor([0, 0, 2.0]),
                 rot=torch.nn.functional.normalize(torch.randn(2, N, 4, generator=g), dim=-1),
                 scale=torch.randn(2, N, 3, generator=g), opacity=torch.randn(2, N, 1, generator=g),
                 color=torch.rand(2, N, 3, generator=g))
    K = torch.tensor([[48.0, 0, 24], [0, 48.0, 24], [0, 0, 1]]).expand(1, 2, 3, 3)
    E = torch.eye(4).expand(1, 2, 4, 4)
    img = R.render_batch(gauss, K, E, (48, 48))
    assert img.shape == (1, 2, 48, 48, 3) and img.min() >= 0 and img.max() <= 1
    dbg = R.render_batch(gauss, K, E, (48, 48), debug=True)   # scales 0.01, opacity 0.1 (renderer.py:535-537)
    assert not torch.equal(img, dbg)
