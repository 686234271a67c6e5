"""Mirror of lib/utils/reba.py::REBA on the MI355X kernel (`pr_reba`).

    reba = REBA(debug)                                   base.py:86
    results = reba(poses, joint_cams, add_info)          base.py:151  -> [{'score', 'log_score'}, ...]
    reba.eval_items, reba.action_level(score), reba.log  base.py:154-160
"""
import numpy as np

from poserisk_release_amd import ops

from _scorer import to_device_poses

_J = {n: i for i, n in enumerate(('Pelvis', 'L_Hip', 'R_Hip', 'Torso', 'L_Knee', 'R_Knee', 'Spine', 'L_Ankle',
                                  'R_Ankle', 'Chest', 'L_Toe', 'R_Toe', 'Neck', 'L_Thorax', 'R_Thorax', 'Head',
                                  'L_Shoulder', 'R_Shoulder', 'L_Elbow', 'R_Elbow', 'L_Wrist', 'R_Wrist',
                                  'L_Hand', 'R_Hand'))}


class REBA:
    def __init__(self, debug=False):
        self.joint_name = tuple(_J)
        self.eval_items = ['Trunk', 'Neck', 'Leg', 'Upper_arm (L,R)', 'Lower_arm (L,R)', 'Wrist (L,R)']
        self.debugging = debug
        self.angle_log = {}
        self.log = []          # accumulates across calls, never cleared (Q19)

    def __call__(self, poses, joint_cams, add_info):
        # joint_cams is accepted and ignored, exactly as every rule of the reference does (Q6)
        packed = ops.reba(to_device_poses(poses), add_info["REBA"]).cpu().numpy()
        if self.debugging:
            P = poses.cpu().numpy() if hasattr(poses, "cpu") else np.asarray(poses)
            self.log.extend(self._angle_log(p) for p in P)
        return [{'score': np.int64(r[0]),
                 'log_score': [int(r[1]), int(r[2]), int(r[3]), f'{r[4]},{r[5]}', f'{r[6]},{r[7]}', f'{r[8]},{r[9]}']}
                for r in packed]

    @staticmethod
    def _angle_log(p):
        """The per-rule angle strings the reference keeps when debugging (reba.py:142, 152, ... 391)."""
        a = lambda j, k: p[_J[j]][k]
        f = lambda v: f'{v:.1f}'
        le = max(a('L_Elbow', 1), a('L_Elbow', 2))
        re = max(a('R_Elbow', 1), a('R_Elbow', 2))
        return {
            'trunk_bending': f(a('Torso', 0)), 'trunk_twist': f(a('Torso', 1)),
            'trunk_side_bending': f(a('Torso', 2)), 'neck_bending': f(a('Neck', 0)),
            'neck_twist': f"{f(a('Neck', 2))},{f(a('Neck', 1))}",
            'leg_bending': f"L {f(a('L_Knee', 0))} R {f(a('R_Knee', 0))}",
            'upper_arm_bending': f"L {f(a('L_Shoulder', 2))},{f(a('L_Shoulder', 1))} R {f(a('R_Shoulder', 2))},{f(a('R_Shoulder', 1))}",
            'shoulder_rise': f"L {f(a('L_Thorax', 2))} R {f(a('R_Thorax', 2))}",
            # reba.py:334 prints L angle1,angle2 and R angle3,angle4 where angle3 is the LEFT shoulder's y (Q14)
            'upper_arm_abducted_rotated': f"L {f(a('L_Shoulder', 2))},{f(a('L_Shoulder', 0))} R {f(a('L_Shoulder', 1))},{f(a('R_Shoulder', 2))}",
            'lower_arm_bending': f"L {f(le)} R {f(re)}",
            'wrist_bending': f"L {f(a('L_Wrist', 2))} R {f(a('R_Wrist', 2))}",
            'wrist_side_bending_or_twisted': f"L {f(a('L_Wrist', 1))},{f(a('L_Wrist', 0))} R {f(a('R_Wrist', 1))},{f(a('R_Wrist', 0))}",
        }

    def action_level(self, score):
        """reba.py:83-104."""
        score = round(score)
        table = ((1, 1, 1, "Negligible risk"), (2, 3, 2, "Low risk. Change may be needed."),
                 (4, 7, 3, "Medium risk. Further Investigate. Change Soon."),
                 (8, 10, 4, "High risk. Investigate and implement change"))
        for lo, hi, level, name in table:
            if lo <= score <= hi:
                return level, name
        if score >= 11:
            return 5, "Very high risk. Implement change"
        return None, None
