"""SMPL-X forward + linear blend skinning, restated on CPU (smplx==0.1.28 is absent; parity unpinned).

Reference call sites: src/models/renderer.py:206-225 (model construction: neutral, num_betas=10, use_pca=False,
flat_hand_mean from cfg) and renderer.py:245-274 (get_smpl_vertices -> SMPLX.forward -> lbs).  Algorithm restated
from smplx.body_models.SMPLX.forward and smplx.lbs.{lbs,batch_rodrigues,batch_rigid_transform}, SURVEY.md
Appendix A.2.  Test infrastructure only (see oracle/__init__.py).

The body model is passed as a dict of tensors (keys as in the SMPL-X npz / smplx buffers):
    v_template [V,3], shapedirs [V,3,n_betas], expr_dirs [V,3,n_expr], posedirs [(J-1)*9, V*3],
    J_regressor [J,V], lbs_weights [V,J], parents [J] (int64, parents[0] = -1), pose_mean [J*3].
All arithmetic follows the dtype of the model tensors (float32 = the reference's, float64 = truth).
"""
import torch
import torch.nn.functional as F


def batch_rodrigues(rot_vecs: torch.Tensor) -> torch.Tensor:
    """[M,3] axis-angle -> [M,3,3].  angle = ||r + 1e-8|| (eps added component-wise before the norm)."""
    batch_size = rot_vecs.shape[0]
    dtype = rot_vecs.dtype
    angle = torch.norm(rot_vecs + 1e-8, dim=1, keepdim=True)
    rot_dir = rot_vecs / angle
    cos = torch.unsqueeze(torch.cos(angle), dim=1)
    sin = torch.unsqueeze(torch.sin(angle), dim=1)
    rx, ry, rz = torch.split(rot_dir, 1, dim=1)
    zeros = torch.zeros((batch_size, 1), dtype=dtype)
    K = torch.cat([zeros, -rz, ry, rz, zeros, -rx, -ry, rx, zeros], dim=1).view((batch_size, 3, 3))
    ident = torch.eye(3, dtype=dtype).unsqueeze(dim=0)
    return ident + sin * K + (1 - cos) * torch.bmm(K, K)


def batch_rigid_transform(rot_mats: torch.Tensor, joints: torch.Tensor, parents: torch.Tensor):
    """Kinematic chain.  Returns posed joints [B,J,3] and rest-pose-removed transforms A [B,J,4,4]."""
    joints = torch.unsqueeze(joints, dim=-1)
    rel_joints = joints.clone()
    rel_joints[:, 1:] -= joints[:, parents[1:]]
    R = rot_mats.reshape(-1, 3, 3)
    t = rel_joints.reshape(-1, 3, 1)
    transforms_mat = torch.cat([F.pad(R, [0, 0, 0, 1]), F.pad(t, [0, 0, 0, 1], value=1)], dim=2).reshape(
        -1, joints.shape[1], 4, 4
    )
    chain = [transforms_mat[:, 0]]
    for i in range(1, parents.shape[0]):
        chain.append(torch.matmul(chain[int(parents[i])], transforms_mat[:, i]))
    transforms = torch.stack(chain, dim=1)
    posed_joints = transforms[:, :, :3, 3]
    joints_homogen = F.pad(joints, [0, 0, 0, 1])
    rel_transforms = transforms - F.pad(torch.matmul(transforms, joints_homogen), [3, 0, 0, 0, 0, 0, 0, 0])
    return posed_joints, rel_transforms


def lbs(coeffs, full_pose, model):
    """coeffs [B, n_betas+n_expr], full_pose [B, J*3] (pose_mean already added) -> vertices [B,V,3], joints, A."""
    B = coeffs.shape[0]
    dtype = model["v_template"].dtype
    shapedirs = torch.cat([model["shapedirs"], model["expr_dirs"]], dim=-1)
    v_shaped = model["v_template"] + torch.einsum("bl,mkl->bmk", coeffs, shapedirs)
    J = torch.einsum("bik,ji->bjk", v_shaped, model["J_regressor"])
    rot_mats = batch_rodrigues(full_pose.reshape(-1, 3)).view(B, -1, 3, 3)
    ident = torch.eye(3, dtype=dtype)
    pose_feature = (rot_mats[:, 1:, :, :] - ident).view(B, -1)
    pose_offsets = torch.matmul(pose_feature, model["posedirs"]).view(B, -1, 3)
    v_posed = pose_offsets + v_shaped
    J_transformed, A = batch_rigid_transform(rot_mats, J, model["parents"])
    num_joints = model["J_regressor"].shape[0]
    W = model["lbs_weights"].unsqueeze(dim=0).expand(B, -1, -1)
    T = torch.matmul(W, A.view(B, num_joints, 16)).view(B, -1, 4, 4)
    homogen = torch.ones(B, v_posed.shape[1], 1, dtype=dtype)
    v_homo = torch.matmul(T, torch.unsqueeze(torch.cat([v_posed, homogen], dim=2), dim=-1))
    return v_homo[:, :, :3, 0], J_transformed, A


def smplx_forward(model, global_orient, body_pose, betas, left_hand_pose, right_hand_pose, jaw_pose, leye_pose,
                  reye_pose, expression):
    """SMPLX.forward as the reference calls it (renderer.py:261-272): no transl, use_pca=False.

    Joint order of full_pose: global(1), body(21), jaw, leye, reye, left hand(15), right hand(15).
    """
    B = global_orient.shape[0]
    full_pose = torch.cat(
        [
            global_orient.reshape(B, 1, 3),
            body_pose.reshape(B, 21, 3),
            jaw_pose.reshape(B, 1, 3),
            leye_pose.reshape(B, 1, 3),
            reye_pose.reshape(B, 1, 3),
            left_hand_pose.reshape(B, 15, 3),
            right_hand_pose.reshape(B, 15, 3),
        ],
        dim=1,
    ).reshape(B, 165)
    full_pose = full_pose + model["pose_mean"]
    coeffs = torch.cat([betas, expression], dim=-1)
    verts, joints, _ = lbs(coeffs, full_pose, model)
    return verts, joints


def get_smpl_vertices(model, smpl_params, densify=None):
    """Renderer.get_smpl_vertices (renderer.py:245-290).

    `densify` = None, or (levels, idx): `levels` is the list of per-level edge tables from
    oracle.subdivide.subdivision_levels(faces, n) and `idx` the fixed vertex subset (the reference draws a new
    torch.randperm subset on every call, renderer.py:287; the deterministic semantics fix it once).
    """
    B, T = smpl_params["global_orient"].shape[:2]
    r = lambda k: smpl_params[k].reshape(B * T, -1)
    verts, _ = smplx_forward(model, r("global_orient"), r("body_pose"), r("betas"), r("left_hand_pose"),
                             r("right_hand_pose"), r("jaw_pose"), r("leye_pose"), r("reye_pose"), r("expression"))
    if densify is not None:
        from .subdivide import subdivide_verts

        levels, idx = densify
        for edges in levels:
            verts = subdivide_verts(verts, edges)
        verts = verts[:, idx, :]
    return verts
