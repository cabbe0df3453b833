"""CPU tests of oracle/ptv3.py: against the reference-run fixtures (ref_ptv3_codes: tier 1, ref_ptv3: tier 2) and
against independent known answers (dense conv3d, Hilbert-curve adjacency)."""
import torch

from helpers import ref_fixture, seeded_params
from oracle import ptv3 as o_pt


def test_codes_match_reference_encode():
    """Bit-exact: serialization/default.py:10-27 for the four orders at depths 1..16."""
    a, meta, tier = ref_fixture("ptv3_codes")
    assert tier == 1
    for depth in meta["depths"]:
        for order in meta["orders"]:
            got = o_pt.encode(a[f"grid_{depth}"], a[f"batch_{depth}"], depth, order)
            assert torch.equal(got, a[f"code_{depth}_{order}"]), (depth, order)


def test_hilbert_is_a_space_filling_curve():
    """Known answer: at depth 3 the 512 codes are a bijection and consecutive codes are face-adjacent cells."""
    r = torch.arange(8)
    grid = torch.stack(torch.meshgrid(r, r, r, indexing="ij"), -1).reshape(-1, 3)
    code = o_pt.hilbert_code(grid, 3)
    assert sorted(code.tolist()) == list(range(512))
    path = grid[torch.argsort(code)]
    assert torch.all((path[1:] - path[:-1]).abs().sum(1) == 1)


def test_z_order_known_answer():
    grid = torch.tensor([[1, 0, 0], [0, 1, 0], [0, 0, 1], [3, 5, 6]])
    # x -> bit 3i+2, y -> 3i+1, z -> 3i;  (3,5,6) = x 011, y 101, z 110 -> bits (i=2: 0,1,1)(i=1: 1,0,1)(i=0: 1,1,0)
    assert o_pt.z_order_code(grid, 3).tolist() == [4, 2, 1, 0b011101110]


def test_subm_conv_equals_dense_conv_on_active_sites():
    """spconv's SubMConv3d == torch's dense conv3d (cross-correlation, zero padding k//2) read at the active voxels."""
    g = torch.Generator().manual_seed(3)
    for k in (3, 5):
        S, n, ci, co = 7, 60, 5, 4
        flat = torch.randperm(S ** 3, generator=g)[:n]
        grid = torch.stack([flat // (S * S), (flat // S) % S, flat % S], -1)
        feat = torch.randn(n, ci, generator=g, dtype=torch.float64)
        w = torch.randn(co, k, k, k, ci, generator=g, dtype=torch.float64)
        b = torch.randn(co, generator=g, dtype=torch.float64)
        nbr = o_pt.neighbor_table(grid, torch.zeros(n, dtype=torch.long), k)
        got = o_pt.subm_conv3d(feat, nbr, w, b)
        dense = torch.zeros(1, ci, S, S, S, dtype=torch.float64)
        dense[0, :, grid[:, 0], grid[:, 1], grid[:, 2]] = feat.t()
        ref = torch.nn.functional.conv3d(dense, w.permute(0, 4, 1, 2, 3), b, padding=k // 2)
        assert torch.allclose(got, ref[0, :, grid[:, 0], grid[:, 1], grid[:, 2]].t(), atol=1e-12)


def test_neighbor_table_shared_voxels():
    """Definition 4: neighbours see a voxel's lowest-index point; the centre tap is the point itself."""
    grid = torch.tensor([[1, 1, 1], [1, 1, 2], [1, 1, 2], [1, 1, 1]])
    nbr = o_pt.neighbor_table(grid, torch.zeros(4, dtype=torch.long), 3)
    centre, plus_z, minus_z = 13, 14, 12
    assert nbr[:, centre].tolist() == [0, 1, 2, 3]
    assert nbr[:, plus_z].tolist() == [1, -1, -1, 1]
    assert nbr[:, minus_z].tolist() == [-1, 0, 0, -1]


def test_patch_layout_matches_reference_padding():
    """pointtransformer_v3.py:392-447 worked by hand: 10 points, patch 4 -> last patch = points 8, 9 + 6, 7."""
    K, pad, unpad = o_pt.patch_layout(10, 4)
    assert K == 4 and pad.tolist() == [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 6, 7] and unpad.tolist() == list(range(10))
    K, pad, _ = o_pt.patch_layout(3, 4)
    assert K == 3 and pad.tolist() == [0, 1, 2]
    K, pad, _ = o_pt.patch_layout(8, 4)
    assert K == 4 and pad.tolist() == list(range(8))


def test_network_matches_reference_run():
    """The reference's PointTransformerV3 classes, run on CPU by the generator (tier 2), one cloud at a time."""
    a, meta, tier = ref_fixture("ptv3")
    assert tier == 2
    cfg = meta["cfg"]
    p = seeded_params(meta["params"], "point_encoder.point_transformer.")
    p = {"point_encoder.point_transformer." + k: v for k, v in p.items()}
    for ci in range(meta["clouds"]):
        grid = o_pt.frame_grid(a[f"pts_{ci}"])
        assert torch.equal(grid, a[f"grid_{ci}"].long())
        _, order, _, _ = o_pt.serialization(grid, torch.zeros(grid.shape[0], dtype=torch.long))
        assert torch.equal(order, a[f"order_{ci}"])
        out = o_pt.ptv3_cloud(p, "point_encoder.point_transformer.", grid, a[f"feat_{ci}"], cfg)
        ref = a[f"out_{ci}"]
        assert out.shape == ref.shape
        assert (out - ref).abs().max() <= 2e-5 * max(1.0, float(ref.abs().max())), float((out - ref).abs().max())


def test_encoder_is_frame_independent():
    """Definition 2: a frame's features do not depend on its batch mates (equal N, different clouds)."""
    a, meta, _ = ref_fixture("ptv3")
    p = {"pe.point_transformer." + k: v for k, v in seeded_params(meta["params"], "point_encoder.point_transformer.").items()}
    pts = torch.stack([a["pts_1"], a["pts_0"][:300]])
    feats = torch.stack([a["feat_1"], a["feat_0"][:300]])
    both = o_pt.encoder_forward(p, "pe.", pts, feats, meta["cfg"])
    assert (both[:300] - a["out_1"]).abs().max() <= 2e-5 * float(a["out_1"].abs().max())
