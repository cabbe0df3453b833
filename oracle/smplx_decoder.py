"""SMPLXDecoder.forward restated on CPU (src/models/smplx_decoder.py:83-145).

Parameters come as a mapping with the reference's names (`mlp.0.weight`, `dec_body_pose.weight`, ...;
smplx_decoder.py:46-81).  Test infrastructure only (oracle/__init__.py).
"""
import torch.nn.functional as F

from .rotation import matrix_to_axis_angle, rotation_6d_to_matrix


def smplx_decoder_forward(params, tokens, prefix="smpl_decoder.", hand_joint_num=15):
    lin = lambda name, x: F.linear(x, params[prefix + name + ".weight"], params[prefix + name + ".bias"])
    B = tokens.shape[0]
    x = tokens.reshape(B, -1)  # [256 channels][80 tokens] row-major (smplx_decoder.py:86)
    x = F.relu(lin("mlp.0", x))
    x = F.relu(lin("mlp.2", x))
    feat = F.relu(lin("mlp.4", x))
    aa = lambda d6, n: matrix_to_axis_angle(rotation_6d_to_matrix(d6.reshape(B, n, 6))).reshape(B, n, 3)
    hand6 = lin("dec_hand_pose", feat)
    return {
        "betas": lin("dec_body_shape", feat),
        "transl": lin("dec_transl", feat),
        "global_orient": aa(lin("dec_body_root_pose", feat), 1).reshape(B, 3),
        "body_pose": aa(lin("dec_body_pose", feat), 21),
        "left_hand_pose": aa(hand6[:, : hand_joint_num * 6], hand_joint_num),
        "right_hand_pose": aa(hand6[:, hand_joint_num * 6:], hand_joint_num),
        "jaw_pose": aa(lin("dec_face_jaw_pose", feat), 1).reshape(B, 3),
        "leye_pose": aa(lin("dec_leye_pose", feat), 1).reshape(B, 3),
        "reye_pose": aa(lin("dec_reye_pose", feat), 1).reshape(B, 3),
        "expression": lin("dec_face_expression", feat),
    }
