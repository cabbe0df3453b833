"""GPU parity of the fused triplane decode (project + sample + heads + construct_gaussians) against the oracle
(torch F.grid_sample + F.linear on CPU: the reference's own arithmetic).  Tolerance 2e-5 absolute on O(1) values:
the two sides sum the same 771 products per output in a different order."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def make_case(seed, F, N, C, R, wstd=0.02):
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g)
    tokens = rn(F, C, 3 * R * R)
    points = rn(F, N, 3) * 0.6
    points[:, :8] *= 4.0  # some points outside the radius: clamp + zero padding at the border
    transl = rn(F, 3)
    params = {}
    for name, n in (("xyz_layer", 3), ("rotation_layer", 4), ("scaling_layer", 3), ("opacity_layer", 1),
                    ("shs_layer", 3)):
        params[f"gaussian_decoder.{name}.weight"] = rn(n, 3 * C + 3) * wstd
        params[f"gaussian_decoder.{name}.bias"] = rn(n) * 0.5
    return tokens, points, transl, params


@pytest.mark.parametrize("F,N,C,R", [(1, 256, 16, 8), (3, 1000, 256, 32), (2, 777, 64, 6), (1, 64, 32, 5),
                                     (1, 50000, 512, 128)])  # last: BASELINE configs[4] (stress) decode shape
def test_fused_decode_matches_grid_sample_plus_heads(F, N, C, R):
    from audio_motion_avatar_amd import ops
    from oracle import triplane as orc

    tokens, points, transl, params = make_case(F * 31 + N, F, N, C, R)
    radius = 1.4
    planes = orc.tokens_to_planes(tokens[None], R)
    ref = orc.decode_gaussians(params, planes, points, transl, radius)
    ref64 = orc.decode_gaussians({k: v.double() for k, v in params.items()}, planes.double(), points.double(),
                                 transl.double(), radius)
    heads = {n: (params[f"gaussian_decoder.{n}.weight"], params[f"gaussian_decoder.{n}.bias"])
             for n in ("xyz_layer", "rotation_layer", "scaling_layer", "opacity_layer", "shs_layer")}
    w_plane, w_point = ops.pack_head_weights(heads, C, "cuda")
    proj = ops.triplane_project(tokens.cuda(), w_plane, R)
    rec = ops.triplane_sample_decode(proj, points.cuda(), transl.cuda(), radius, w_point).cpu()
    got = dict(xyz=rec[..., 0:3], opacity=rec[..., 3:4], rot=rec[..., 4:8], scale=rec[..., 8:11],
               color=rec[..., 12:15])
    for k, v in got.items():
        assert (v - ref[k]).abs().max() <= 2e-5, k
        # against fp64: 2e-5, or (long sums: C = 512 is 1539 terms, and normalising a short quaternion amplifies their
        # rounding) no worse than four times the fp32 reference's own distance from fp64
        own = (ref[k].double() - ref64[k]).abs().max().item()
        assert (v.double() - ref64[k]).abs().max() <= max(2e-5, 4 * own), k
    assert torch.count_nonzero(rec[..., 11]) == 0 and torch.count_nonzero(rec[..., 15]) == 0


@pytest.mark.parametrize("F,N,C,R", [(2, 5000, 64, 32), (1, 333, 16, 8)])
def test_fused_decode_does_not_depend_on_point_order(F, N, C, R):
    """A point's record depends on nothing but that point: the same points stored along a space-filling curve
    (RendererConfig.subset_order = "spatial", where the lanes of a wave share texels) or in random order give the same
    records bit for bit, and match the oracle to the usual tolerance."""
    from audio_motion_avatar_amd import ops
    from oracle import triplane as orc

    tokens, points, transl, params = make_case(99 + N, F, N, C, R)
    q = ((points.clamp(-1.4, 1.4) + 1.4) / 2.8 * 1023).long()
    code = torch.zeros(F, N, dtype=torch.long)
    for bit in range(10):
        for axis in range(3):
            code |= ((q[..., axis] >> bit) & 1) << (3 * bit + axis)
    order = torch.argsort(code, dim=1)
    sorted_pts = torch.gather(points, 1, order[..., None].expand(-1, -1, 3)).contiguous()
    heads = {n: (params[f"gaussian_decoder.{n}.weight"], params[f"gaussian_decoder.{n}.bias"])
             for n in ("xyz_layer", "rotation_layer", "scaling_layer", "opacity_layer", "shs_layer")}
    w_plane, w_point = ops.pack_head_weights(heads, C, "cuda")
    proj = ops.triplane_project(tokens.cuda(), w_plane, R)
    rec_sorted = ops.triplane_sample_decode(proj, sorted_pts.cuda(), transl.cuda(), 1.4, w_point).cpu()
    rec_random = ops.triplane_sample_decode(proj, points.cuda(), transl.cuda(), 1.4, w_point).cpu()
    assert torch.equal(rec_sorted, torch.gather(rec_random, 1, order[..., None].expand(-1, -1, 16)))
    ref = orc.decode_gaussians(params, orc.tokens_to_planes(tokens[None], R), sorted_pts, transl, 1.4)
    assert (rec_sorted[..., 0:3] - ref["xyz"]).abs().max() <= 2e-5
    assert (rec_sorted[..., 12:15] - ref["color"]).abs().max() <= 2e-5


@pytest.mark.parametrize("F,N,C,R", [(2, 300, 16, 8), (1, 500, 256, 32)])
def test_sample_features_matches_grid_sample(F, N, C, R):
    from audio_motion_avatar_amd import ops
    from oracle import triplane as orc

    tokens, points, _, _ = make_case(77, F, N, C, R)
    planes = orc.tokens_to_planes(tokens[None], R)
    ref = orc.sample_from_triplane(planes, points, 1.4)
    got = ops.triplane_sample_features(planes.cuda(), points.cuda(), 1.4).cpu()
    assert got.shape == ref.shape
    assert (got - ref).abs().max() <= 1e-5


def test_sample_features_on_the_token_layout_matches_grid_sample():
    """The refiner samples the [F, C, 3 R^2] token slab in place (a permuted view, channel stride 3 R^2), 64-channel
    tiles, a ragged last point tile, points outside the radius."""
    from audio_motion_avatar_amd import ops
    from oracle import triplane as orc

    F, N, C, R = 3, 130, 128, 16
    tokens, points, _, _ = make_case(78, F, N, C, R)
    points[0, :5] *= 4.0
    ref = orc.sample_from_triplane(orc.tokens_to_planes(tokens[None], R), points, 1.4)
    view = tokens.cuda().view(F, C, 3, R, R).permute(0, 2, 1, 3, 4)
    got = ops.triplane_sample_features(view, points.cuda(), 1.4).cpu()
    assert (got - ref).abs().max() <= 1e-5


def test_indexed_decode_equals_gather_then_decode_bitwise():
    """amav_triplane_sample_decode_indexed == amav_points_gather + amav_triplane_sample_decode, bit for bit."""
    from audio_motion_avatar_amd import ops

    F, V, N, C, R = 3, 500, 800, 32, 8
    g = torch.Generator().manual_seed(4)
    tokens, _, transl, params = make_case(9, F, N, C, R)
    verts = (torch.randn(F, V, 3, generator=g) * 0.6).cuda()
    idx = torch.randint(0, V, (N, 4), generator=g, dtype=torch.int32).cuda()
    heads = {n: (params[f"gaussian_decoder.{n}.weight"], params[f"gaussian_decoder.{n}.bias"])
             for n in ("xyz_layer", "rotation_layer", "scaling_layer", "opacity_layer", "shs_layer")}
    w_plane, w_point = ops.pack_head_weights(heads, C, "cuda")
    proj = ops.triplane_project(tokens.cuda(), w_plane, R)
    a = ops.triplane_sample_decode(proj, ops.points_gather(verts, idx), transl.cuda(), 1.4, w_point)
    b = ops.triplane_sample_decode_indexed(proj, verts, idx, transl.cuda(), 1.4, w_point)
    assert torch.equal(a, b)


@pytest.mark.parametrize("F,N,C,R,spread", [(3, 2000, 64, 64, 0.35), (2, 500, 32, 32, 0.9), (2, 300, 16, 8, 3.0), (1, 64, 8, 4, 0.2)])
def test_projection_of_the_sampled_region_gives_the_same_gaussians(F, N, C, R, spread):
    """amav_triplane_project_region projects only the texels that points inside the frame's box can sample.  The planes
    are poisoned with NaN first: had the sampling kernels formed one tap address outside the projected rectangle --
    zero-padded taps read the clamped address with weight 0 -- the records would not be finite; they are bit-identical
    to the full projection's, for direct points, for the indexed (subdivision) form with the box of the VERTICES, with
    points beyond the radius (spread 3.0: clamped coordinates), and the rectangle really is a part of the plane."""
    from audio_motion_avatar_amd import ops

    tokens, _, transl, params = make_case(F * 7 + N, F, N, C, R)
    g = torch.Generator().manual_seed(N)
    centre = torch.randn(F, 1, 3, generator=g) * 0.2
    points = (centre + torch.randn(F, N, 3, generator=g) * torch.tensor([0.3, 0.6, 0.12]) * spread).cuda()
    radius = 1.4
    heads = {n: (params[f"gaussian_decoder.{n}.weight"], params[f"gaussian_decoder.{n}.bias"])
             for n in ("xyz_layer", "rotation_layer", "scaling_layer", "opacity_layer", "shs_layer")}
    w_plane, w_point = ops.pack_head_weights(heads, C, "cuda")
    tok = tokens.cuda()
    full = ops.triplane_project(tok, w_plane, R)
    want = ops.triplane_sample_decode(full, points, transl.cuda(), radius, w_point)
    boxes = ops.points_bbox(points)
    assert torch.equal(boxes[:, :3], points.min(1).values) and torch.equal(boxes[:, 3:], points.max(1).values)
    part = torch.full((F, 3, R, R, 16), float("nan"), device="cuda")
    ops.triplane_project(tok, w_plane, R, region=(boxes, radius), out=part)
    written = ~torch.isnan(part[..., 0])
    assert torch.equal(part[written], full[written])
    if spread < 1.0 and R >= 32:
        assert 0.02 < written.float().mean() < 0.7
    got = ops.triplane_sample_decode(part, points, transl.cuda(), radius, w_point)
    assert torch.isfinite(got).all() and torch.equal(got, want)
    # the subdivision form: points are averages of vertices, the box is the vertices'
    V = max(16, N // 3)
    verts = points[:, :V].contiguous()
    idx = torch.randint(0, V, (N, 4), generator=g, dtype=torch.int32).cuda()
    part.fill_(float("nan"))
    ops.triplane_project(tok, w_plane, R, region=(ops.points_bbox(verts), radius), out=part)
    a = ops.triplane_sample_decode_indexed(part, verts, idx, transl.cuda(), radius, w_point)
    b = ops.triplane_sample_decode_indexed(full, verts, idx, transl.cuda(), radius, w_point)
    assert torch.isfinite(a).all() and torch.equal(a, b)
    # a NaN among the points: the whole plane is projected
    bad = points.clone()
    bad[0, 5, 1] = float("nan")
    nb = ops.points_bbox(bad)
    assert torch.isinf(nb[0]).all() and torch.isfinite(nb[1:]).all()
    part.fill_(float("nan"))
    ops.triplane_project(tok, w_plane, R, region=(nb, radius), out=part)
    assert not torch.isnan(part[0]).any()
    with pytest.raises(ops.AmavError):
        ops.triplane_project(tok, w_plane, R, region=(boxes[:, :5], radius))
