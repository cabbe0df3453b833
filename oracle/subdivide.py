"""Edge-midpoint mesh subdivision (pytorch3d.ops.SubdivideMeshes is absent; parity unpinned).

Reference call sites: src/models/renderer.py:227-243 (init_smplx_subdivider) and :276-288 (applied to the POSED
vertices every forward).  Algorithm restated from pytorch3d: the new vertex list is the
old vertices followed by the midpoints of the unique undirected edges in ascending (min_id, max_id) order; each
face (v0,v1,v2) with edge ids e12,e20,e01 (offset by V) splits into (v0,e01,e20), (v1,e12,e01), (v2,e20,e12),
(e12,e20,e01).  Test infrastructure only (see oracle/__init__.py).
"""
import numpy as np
import torch


def unique_edges(faces: np.ndarray, num_verts: int):
    """faces [F,3] int -> (edges [E,2] sorted ascending by (min,max), face_to_edge [F,3] = ids of e12,e20,e01)."""
    f = faces.astype(np.int64)
    e = np.concatenate([f[:, [1, 2]], f[:, [2, 0]], f[:, [0, 1]]], axis=0)
    e = np.sort(e, axis=1)
    key = e[:, 0] * num_verts + e[:, 1]
    uniq, inverse = np.unique(key, return_inverse=True)
    edges = np.stack([uniq // num_verts, uniq % num_verts], axis=1)
    face_to_edge = inverse.reshape(3, -1).T
    return edges, face_to_edge


def subdivide_faces(faces: np.ndarray, face_to_edge: np.ndarray, num_verts: int) -> np.ndarray:
    fe = face_to_edge + num_verts
    f0 = np.stack([faces[:, 0], fe[:, 2], fe[:, 1]], axis=1)
    f1 = np.stack([faces[:, 1], fe[:, 0], fe[:, 2]], axis=1)
    f2 = np.stack([faces[:, 2], fe[:, 1], fe[:, 0]], axis=1)
    return np.concatenate([f0, f1, f2, fe], axis=0)


def subdivision_levels(faces: np.ndarray, num_verts: int, steps: int):
    """Edge table of each subdivision level, as the reference's subdivider_list applies them (renderer.py:238-241)."""
    levels = []
    f = faces.astype(np.int64)
    v = num_verts
    for _ in range(steps):
        edges, f2e = unique_edges(f, v)
        levels.append(edges)
        f = subdivide_faces(f, f2e, v)
        v = v + edges.shape[0]
    return levels


def subdivide_verts(verts: torch.Tensor, edges: np.ndarray) -> torch.Tensor:
    """verts [B,V,3] -> [B,V+E,3]: midpoints = mean of the two end points."""
    e = torch.as_tensor(edges)
    mid = verts[:, e].mean(dim=2)
    return torch.cat([verts, mid], dim=1)
