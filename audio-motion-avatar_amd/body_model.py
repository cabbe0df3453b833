"""SMPL-X body model on the HIP LBS kernels (replaces smplx.SMPLX as the reference uses it).

Reference: src/models/renderer.py:206-225 (construction: neutral, num_betas=10, use_pca=False, flat_hand_mean from
cfg), :232-233 (`.v_template`, `.faces` feed the subdivider), :261-274 (`__call__` with nine keyword tensors ->
`.vertices`), :276-288 (subdivide the posed mesh, pick a vertex subset).

The SMPL-X model file is licence-gated and absent from this image, so `BodyModel.synthetic()` builds a seeded
SMPL-X-SHAPED stand-in (10 475 vertices, 55 joints with the SMPL-X kinematic tree, 10+10 shape/expression
directions, 486 pose-corrective rows, <= 4 skinning weights per vertex).  `BodyModel.from_npz()` loads the real
arrays when a path is given.  All arithmetic on vertices happens in the HIP kernels (csrc/lbs.hip); this file only
prepares immutable tables once, on the host.
"""
import os

import numpy as np
import torch

from . import ops
from ._lib import AmavError

NUM_JOINTS = 55
# SMPL-X kinematic tree: 0-21 body, 22 jaw, 23/24 eyes, 25-39 left hand, 40-54 right hand
SMPLX_PARENTS = np.array(
    [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 15, 15, 15,
     20, 25, 26, 20, 28, 29, 20, 31, 32, 20, 34, 35, 20, 37, 38,
     21, 40, 41, 21, 43, 44, 21, 46, 47, 21, 49, 50, 21, 52, 53], dtype=np.int64)

# vertices sampled after densification, by subdivide_steps (src/models/renderer.py:14-18)
SUBDIVIDE_VERTS = {0: 10000, 1: 30000, 2: 30000}


def _rest_joints() -> np.ndarray:
    """Plausible SMPL-X-like T-pose joint locations in metres (y up, z forward, pelvis at the origin)."""
    J = np.zeros((NUM_JOINTS, 3))
    body = {
        0: (0, 0, 0), 1: (0.07, -0.09, 0), 2: (-0.07, -0.09, 0), 3: (0, 0.11, -0.02), 4: (0.10, -0.48, 0.0),
        5: (-0.10, -0.48, 0.0), 6: (0, 0.25, 0.0), 7: (0.09, -0.90, -0.03), 8: (-0.09, -0.90, -0.03),
        9: (0, 0.31, 0.01), 10: (0.10, -0.95, 0.09), 11: (-0.10, -0.95, 0.09), 12: (0, 0.52, -0.02),
        13: (0.06, 0.43, 0.0), 14: (-0.06, 0.43, 0.0), 15: (0, 0.61, 0.02), 16: (0.18, 0.46, -0.02),
        17: (-0.18, 0.46, -0.02), 18: (0.44, 0.46, -0.03), 19: (-0.44, 0.46, -0.03), 20: (0.69, 0.46, -0.03),
        21: (-0.69, 0.46, -0.03), 22: (0, 0.60, 0.05), 23: (0.03, 0.67, 0.08), 24: (-0.03, 0.67, 0.08),
    }
    for k, v in body.items():
        J[k] = v
    # fingers: index, middle, pinky, ring, thumb (3 joints each), fanned in z
    fan = [0.03, 0.01, -0.035, -0.012, 0.045]
    first = [0.095, 0.10, 0.085, 0.095, 0.035]
    seg = [0.032, 0.035, 0.022, 0.03, 0.03]
    for side, wrist, base in ((1, 20, 25), (-1, 21, 40)):
        for fi in range(5):
            for k in range(3):
                x = J[wrist, 0] + side * (first[fi] + k * seg[fi])
                y = J[wrist, 1] - (0.01 if fi == 4 else 0.0) * (k + 1)
                z = J[wrist, 2] + fan[fi] * (1 + 0.3 * k)
                J[base + fi * 3 + k] = (x, y, z)
    return J


def _synthetic_arrays(seed: int = 42, num_verts: int = 10475, num_betas: int = 10, num_expr: int = 10):
    """Seeded SMPL-X-shaped arrays.  One cylinder patch of 25-vertex rings per bone (and per end effector)."""
    rng = np.random.default_rng(seed)
    J = _rest_joints()
    parents = SMPLX_PARENTS
    ring = 25
    if num_verts % ring:
        raise AmavError("synthetic body: num_verts must be a multiple of 25")
    total_rings = num_verts // ring
    children = {j: [c for c in range(NUM_JOINTS) if parents[c] == j] for j in range(NUM_JOINTS)}

    def radius_of(j):
        if j in (3, 6, 9):
            return 0.13
        if j in (1, 2):
            return 0.085
        if j in (4, 5):
            return 0.06
        if j in (7, 8, 10, 11):
            return 0.042
        if j in (12,):
            return 0.055
        if j in (13, 14, 16, 17):
            return 0.055
        if j in (18, 19):
            return 0.042
        if j in (20, 21):
            return 0.033
        if j == 15:
            return 0.095
        if j in (22, 23, 24):
            return 0.02
        return 0.009  # fingers

    # segments: (joint the patch follows, skin joints (a, b), start, end, radius)
    segs = []
    for j in range(1, NUM_JOINTS):
        segs.append((parents[j], j, J[parents[j]], J[j], radius_of(j)))
    tips = {15: (np.array([0, 0.17, 0.0]), 0.10), 10: (np.array([0, -0.02, 0.12]), 0.035),
            11: (np.array([0, -0.02, 0.12]), 0.035)}
    for j in range(NUM_JOINTS):
        if not children[j] and j not in (22, 23, 24):
            d, r = tips.get(j, (None, None))
            if d is None:  # finger tips continue along the finger
                v = J[j] - J[parents[j]]
                d, r = v / (np.linalg.norm(v) + 1e-9) * 0.022, 0.008
            segs.append((j, j, J[j], J[j] + d, r))
    weight = np.array([max(np.linalg.norm(e - s), 0.02) * (r ** 0.5) for (_, _, s, e, r) in segs])
    rings = np.maximum(2, np.floor(weight / weight.sum() * total_rings).astype(int))
    while rings.sum() > total_rings:
        rings[np.argmax(rings)] -= 1
    while rings.sum() < total_rings:
        rings[np.argmax(weight / rings)] += 1

    verts, faces, skin = [], [], []
    theta = np.arange(ring) / ring * 2 * np.pi
    for (ja, jb, s, e, r), nr in zip(segs, rings):
        axis = e - s
        L = np.linalg.norm(axis) + 1e-9
        axis = axis / L
        ref = np.array([0, 0, 1.0]) if abs(axis[2]) < 0.9 else np.array([1.0, 0, 0])
        u = np.cross(axis, ref)
        u /= np.linalg.norm(u)
        w = np.cross(axis, u)
        base = len(verts) * ring
        for k in range(nr):
            t = k / (nr - 1)
            # slightly flattened torso, tapered ends
            taper = 0.75 + 0.25 * np.sin(np.pi * t)
            sx, sz = (1.25, 0.8) if r > 0.1 else (1.0, 1.0)
            c = s + axis * (t * L)
            pts = c[None] + (np.cos(theta)[:, None] * u[None] * sx + np.sin(theta)[:, None] * w[None] * sz) * r * taper
            verts.append(pts)
            wa = np.zeros((ring, NUM_JOINTS))
            sm = t * t * (3 - 2 * t)
            if ja == jb:
                wa[:, ja] = 1.0
            else:
                wa[:, ja] = 1.0 - sm
                wa[:, jb] = sm
                gp = parents[ja]
                if gp >= 0 and t < 0.3:  # a little grandparent influence near the proximal joint
                    g = 0.25 * (0.3 - t) / 0.3
                    wa[:, ja] *= 1 - g
                    wa[:, jb] *= 1 - g
                    wa[:, gp] += g
            skin.append(wa)
        for k in range(nr - 1):
            a0 = base + k * ring
            b0 = a0 + ring
            for i in range(ring):
                i1 = (i + 1) % ring
                faces.append((a0 + i, a0 + i1, b0 + i))
                faces.append((a0 + i1, b0 + i1, b0 + i))
        for a0, flip in ((base, True), (base + (nr - 1) * ring, False)):  # fan caps
            for i in range(1, ring - 1):
                faces.append((a0, a0 + i + 1, a0 + i) if flip else (a0, a0 + i, a0 + i + 1))
    v_template = np.concatenate(verts, 0)
    lbs_weights = np.concatenate(skin, 0)
    lbs_weights /= lbs_weights.sum(1, keepdims=True)
    faces = np.asarray(faces, dtype=np.int64)
    V = v_template.shape[0]
    # joint regressor: mean of the ring vertices closest to each joint
    J_regressor = np.zeros((NUM_JOINTS, V))
    for j in range(NUM_JOINTS):
        d = np.linalg.norm(v_template - J[j][None], axis=1)
        near = np.argsort(d)[:ring]
        J_regressor[j, near] = 1.0 / ring
    shapedirs = rng.normal(0, 0.006, (V, 3, num_betas)) * np.linspace(1.0, 0.3, num_betas)[None, None]
    expr_dirs = rng.normal(0, 0.002, (V, 3, num_expr))
    head = lbs_weights[:, [15, 22, 23, 24]].sum(1) > 0.5
    expr_dirs *= head[:, None, None]
    posedirs = rng.normal(0, 0.0015, ((NUM_JOINTS - 1) * 9, V * 3))
    return dict(v_template=v_template, faces=faces, shapedirs=shapedirs, expr_dirs=expr_dirs, posedirs=posedirs,
                J_regressor=J_regressor, lbs_weights=lbs_weights, parents=parents.copy(),
                pose_mean=np.zeros(NUM_JOINTS * 3))


def _npz_arrays(path: str, num_betas: int, num_expr: int, flat_hand_mean: bool):
    """Arrays of a real SMPL-X model file (SMPLX_NEUTRAL.npz), loaded without unpickling anything."""
    if os.path.isdir(path):
        path = os.path.join(path, "SMPLX_NEUTRAL.npz")
    data = np.load(path, allow_pickle=False)
    shapedirs_all = np.asarray(data["shapedirs"], dtype=np.float64)
    V = data["v_template"].shape[0]
    posedirs = np.asarray(data["posedirs"], dtype=np.float64).reshape(V * 3, -1).T  # [(J-1)*9, V*3]
    parents = np.asarray(data["kintree_table"])[0].astype(np.int64)
    parents[0] = -1
    pose_mean = np.zeros(NUM_JOINTS * 3)
    if not flat_hand_mean:
        pose_mean[75:120] = np.asarray(data["hands_meanl"], dtype=np.float64)
        pose_mean[120:165] = np.asarray(data["hands_meanr"], dtype=np.float64)
    # smplx (body_models.py, SMPL / SMPLX.__init__): a model file with fewer than 300 + 100 components is the "10
    # shape + 10 expression" release (SMPL-X v1.0), whose expression directions sit at 10:20
    if shapedirs_all.ndim < 3:
        shapedirs_all = shapedirs_all[:, :, None]
    if shapedirs_all.shape[-1] < 400:
        num_betas, num_expr = min(num_betas, 10), min(num_expr, 10)
        expr_start = 10
    else:
        expr_start = 300
    return dict(v_template=np.asarray(data["v_template"], dtype=np.float64),
                faces=np.asarray(data["f"]).astype(np.int64),
                shapedirs=shapedirs_all[:, :, :num_betas],
                expr_dirs=shapedirs_all[:, :, expr_start:expr_start + num_expr],
                posedirs=posedirs,
                J_regressor=np.asarray(data["J_regressor"], dtype=np.float64),
                lbs_weights=np.asarray(data["weights"], dtype=np.float64),
                parents=parents, pose_mean=pose_mean)


def _unique_edges(faces: np.ndarray, num_verts: int):
    """Unique undirected edges in ascending (min, max) order + per-face edge ids (e12, e20, e01)."""
    e = np.concatenate([faces[:, [1, 2]], faces[:, [2, 0]], faces[:, [0, 1]]], axis=0)
    e = np.sort(e, axis=1)
    uniq, inverse = np.unique(e[:, 0] * num_verts + e[:, 1], return_inverse=True)
    return np.stack([uniq // num_verts, uniq % num_verts], axis=1), inverse.reshape(3, -1).T


def build_subdivision_table(faces: np.ndarray, num_verts: int, levels: int) -> np.ndarray:
    """[V', 4] base-vertex ids of every vertex after `levels` (1 or 2) edge-midpoint subdivisions.

    Row (a0, b0, a1, b1) means 1/2 (1/2 (v[a0]+v[b0]) + 1/2 (v[a1]+v[b1])): the order in which the reference's
    SubdivideMeshes chain evaluates it (renderer.py:282-283), so the gather kernel is bit-identical to it.
    """
    if levels not in (1, 2):
        raise AmavError("subdivision levels must be 1 or 2")
    f = faces.astype(np.int64)
    base = np.arange(num_verts, dtype=np.int64)
    pair = np.stack([base, base], axis=1)  # every current vertex as a pair of base vertices
    e1, f2e = _unique_edges(f, num_verts)
    pair1 = np.concatenate([pair, e1], axis=0)  # level-1 vertices as base pairs
    if levels == 1:
        return np.concatenate([pair1, pair1], axis=1).astype(np.int32)
    fe = f2e + num_verts
    f1 = np.concatenate([np.stack([f[:, 0], fe[:, 2], fe[:, 1]], 1), np.stack([f[:, 1], fe[:, 0], fe[:, 2]], 1),
                         np.stack([f[:, 2], fe[:, 1], fe[:, 0]], 1), fe], axis=0)
    v1 = pair1.shape[0]
    e2, _ = _unique_edges(f1, v1)
    keep = np.concatenate([pair1, pair1], axis=1)              # level-1 vertices carried over unchanged
    mids = np.concatenate([pair1[e2[:, 0]], pair1[e2[:, 1]]], axis=1)
    return np.concatenate([keep, mids], axis=0).astype(np.int32)


class _BodyOutput:
    """`.vertices`, and `.full_pose` (smplx's output field; nothing on the path reads it) assembled on first use."""

    def __init__(self, vertices, make_full_pose):
        self.vertices = vertices
        self._make_full_pose = make_full_pose

    @property
    def full_pose(self):
        return self._make_full_pose()


class BodyModel(torch.nn.Module):
    """Drop-in for the `smplx.SMPLX` object the reference builds (renderer.py:206-225)."""

    def __init__(self, arrays: dict, device="cuda", synthetic=False):
        super().__init__()
        self.synthetic = synthetic
        self.faces = arrays["faces"].astype(np.int64)  # numpy, as smplx exposes it (renderer.py:233)
        f32 = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float32))
        self.register_buffer("v_template", f32(arrays["v_template"]), persistent=False)
        self.register_buffer("lbs_weights", f32(arrays["lbs_weights"]), persistent=False)
        self.register_buffer("pose_mean", f32(arrays["pose_mean"]), persistent=False)
        self.parents = arrays["parents"].astype(np.int64)
        self.num_verts = int(arrays["v_template"].shape[0])
        self.num_joints = int(self.parents.shape[0])
        self.num_betas = int(arrays["shapedirs"].shape[2])
        self.num_expression_coeffs = int(arrays["expr_dirs"].shape[2])
        self._arrays = arrays  # float64 host copies (fixtures / oracle inputs are cut from these)

        V, J = self.num_verts, self.num_joints
        dirs = np.concatenate([arrays["shapedirs"], arrays["expr_dirs"]], axis=2).astype(np.float64)  # [V,3,NC]
        nc = dirs.shape[2]
        blend = np.concatenate([dirs.reshape(V * 3, nc).T, arrays["posedirs"].astype(np.float64)], axis=0)
        kb = blend.shape[0]
        # tile-major component planes [ceil(V/32), KB, 3, 32] (zero padded): a 32-vertex tile's rows are one contiguous
        # slab, which is what the skinning kernels stream (csrc/lbs.hip)
        vt = (V + 31) // 32
        planes = np.zeros((kb, 3, vt * 32), np.float64)
        planes[:, :, :V] = blend.reshape(kb, V, 3).transpose(0, 2, 1)
        blend = np.ascontiguousarray(planes.reshape(kb, 3, vt, 32).transpose(2, 0, 1, 3))
        Jreg = arrays["J_regressor"].astype(np.float64)
        j_template = Jreg @ arrays["v_template"].astype(np.float64)                  # [J,3]
        j_dirs = np.einsum("jv,vcl->jcl", Jreg, dirs).reshape(J * 3, nc)            # [J*3, NC]
        W = arrays["lbs_weights"].astype(np.float64)
        nnz = (W != 0).sum(1)
        kmax = int(nnz.max())
        skin_idx = np.zeros((V, kmax), np.int32)
        skin_w = np.zeros((V, kmax), np.float32)
        for v in range(V):
            js = np.nonzero(W[v])[0]
            skin_idx[v, :js.size] = js
            skin_w[v, :js.size] = W[v, js]
        self.register_buffer("_blend", f32(blend), persistent=False)
        self.register_buffer("_j_template", f32(j_template), persistent=False)
        self.register_buffer("_j_dirs", f32(j_dirs), persistent=False)
        self.register_buffer("_parents32", torch.as_tensor(self.parents.astype(np.int32)), persistent=False)
        self.register_buffer("_skin_idx", torch.as_tensor(skin_idx), persistent=False)
        self.register_buffer("_skin_w", torch.as_tensor(skin_w), persistent=False)
        self.to(device)

    # ---- construction -------------------------------------------------------------------------------------------
    @classmethod
    def synthetic_model(cls, seed=42, device="cuda", num_betas=10, num_expression_coeffs=10):
        return cls(_synthetic_arrays(seed, num_betas=num_betas, num_expr=num_expression_coeffs), device, True)

    @classmethod
    def from_npz(cls, path, device="cuda", num_betas=10, num_expression_coeffs=10, flat_hand_mean=True):
        return cls(_npz_arrays(path, num_betas, num_expression_coeffs, flat_hand_mean), device, False)

    @classmethod
    def create(cls, smplx_model_path=None, device="cuda", num_betas=10, num_expression_coeffs=10,
               flat_hand_mean=True, seed=42):
        """Real model when `smplx_model_path` points at SMPLX_NEUTRAL.npz (or its directory), else synthetic."""
        if smplx_model_path:
            p = smplx_model_path
            if os.path.isdir(p):
                p = os.path.join(p, "SMPLX_NEUTRAL.npz")
            if os.path.exists(p):
                return cls.from_npz(p, device, num_betas, num_expression_coeffs, flat_hand_mean)
        return cls.synthetic_model(seed, device, num_betas, num_expression_coeffs)

    # ---- tables for the C ABI ----------------------------------------------------------------------------------
    def device_tables(self) -> dict:
        tables = dict(v_template=self.v_template, blend=self._blend, j_template=self._j_template, j_dirs=self._j_dirs,
                      parents=self._parents32, skin_idx=self._skin_idx, skin_w=self._skin_w)
        if self._blend.is_cuda:
            # the blend table as two fp16 parts for the 16-bit matrix pipe (csrc/lbs.hip, skin_f16_kernel), built once
            # per device placement of the table
            key = (self._blend.data_ptr(), ops.tensor_version(self._blend))
            if getattr(self, "_blend_split", None) is None or self._blend_split[0] != key:
                self._blend_split = (key, ops.lbs_prepare_blend_split(tables))
            tables["blend_split"] = self._blend_split[1]
        return tables

    def oracle_arrays(self, dtype=torch.float32) -> dict:
        """The model as plain CPU tensors (what tests hand to oracle.lbs; no product code consumes this)."""
        a = self._arrays
        t = lambda x: torch.as_tensor(np.asarray(x, dtype=np.float32)).to(dtype)
        return dict(v_template=t(a["v_template"]), shapedirs=t(a["shapedirs"]), expr_dirs=t(a["expr_dirs"]),
                    posedirs=t(a["posedirs"]), J_regressor=t(a["J_regressor"]), lbs_weights=t(a["lbs_weights"]),
                    parents=torch.as_tensor(a["parents"]), pose_mean=t(a["pose_mean"]))

    # ---- forward ---------------------------------------------------------------------------------------------
    def full_pose(self, global_orient, body_pose, jaw_pose, leye_pose, reye_pose, left_hand_pose, right_hand_pose):
        B = global_orient.shape[0]
        fp = torch.cat([global_orient.reshape(B, 3), body_pose.reshape(B, 63), jaw_pose.reshape(B, 3),
                        leye_pose.reshape(B, 3), reye_pose.reshape(B, 3), left_hand_pose.reshape(B, 45),
                        right_hand_pose.reshape(B, 45)], dim=1)
        return fp + self.pose_mean

    def forward(self, global_orient, body_pose, betas, left_hand_pose, right_hand_pose, jaw_pose, leye_pose,
                reye_pose, expression, **unused):
        """Same keyword call as renderer.py:261-272; returns an object with `.vertices` [B,V,3] (and, computed on
        demand, smplx's `.full_pose`).  float32 arguments go to the joint-chain kernel as they are (it concatenates on
        load and adds pose_mean: smplx's torch.cat + add + torch.cat were three launches); other dtypes are assembled
        with torch first, in their own dtype as smplx does."""
        B = global_orient.shape[0]
        pose = [global_orient.reshape(B, 3), body_pose.reshape(B, 63), jaw_pose.reshape(B, 3), leye_pose.reshape(B, 3),
                reye_pose.reshape(B, 3), left_hand_pose.reshape(B, 45), right_hand_pose.reshape(B, 45)]
        coeff = [betas.reshape(B, -1), expression.reshape(B, -1)]
        if all(p.dtype == torch.float32 and (p.shape[1] == 1 or p.stride(1) == 1) for p in pose + coeff):
            verts = ops.lbs_forward_parts(self.device_tables(), pose, coeff, pose_mean=self.pose_mean)
        else:
            fp = (torch.cat(pose, dim=1) + self.pose_mean).float()
            verts = ops.lbs_forward(self.device_tables(), fp, torch.cat(coeff, dim=1).float())
        return _BodyOutput(verts, lambda: (torch.cat(pose, dim=1) + self.pose_mean).float())
