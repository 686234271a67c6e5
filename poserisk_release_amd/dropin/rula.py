"""Mirror of lib/utils/rula.py::RULA on the MI355X kernel (`pr_rula`).  Same surface as reba.REBA
(base.py:86,168-176)."""
import numpy as np

from poserisk_release_amd import ops

from _scorer import to_device_poses
from reba import _J


class RULA:
    def __init__(self, debug=False):
        self.joint_name = tuple(_J)
        self.eval_items = ['Upper_arm (L,R)', 'Lower_arm (L,R)', 'Wrist (L,R)', 'Wrist_twist (L,R)', 'Neck', 'Trunk',
                           'Leg']
        self.debugging = debug
        self.angle_log = {}
        self.log = []

    def __call__(self, poses, joint_cams, add_info):
        packed = ops.rula(to_device_poses(poses), add_info["RULA"]).cpu().numpy()
        if self.debugging:
            P = poses.cpu().numpy() if hasattr(poses, "cpu") else np.asarray(poses)
            self.log.extend(self._angle_log(p) for p in P)
        return [{'score': np.int64(r[0]),
                 'log_score': [f'{r[1]},{r[2]}', f'{r[3]},{r[4]}', f'{r[5]},{r[6]}', f'{r[7]},{r[8]}',
                               int(r[9]), int(r[10]), int(r[11])]}
                for r in packed]

    @staticmethod
    def _angle_log(p):
        """Per-rule angle strings (rula.py:198, 216, 287, ... 420)."""
        a = lambda j, k: p[_J[j]][k]
        f = lambda v: f'{v:.1f}'
        le = max(a('L_Elbow', 1), a('L_Elbow', 2))
        re = max(a('R_Elbow', 1), a('R_Elbow', 2))
        r2, r1 = a('R_Shoulder', 2), a('R_Shoulder', 1)
        # rula.py:182-183: inside the first gate, |angle4| < 20 overwrites the logged angle with 1 (Q15)
        r1_logged = 1 if (-70 < r2 < 110 and abs(r1) < 20) else r1
        return {
            'upper_arm_bending': f"L {f(a('L_Shoulder', 2))},{f(a('L_Shoulder', 1))} R {f(r2)},{f(r1_logged)}",
            'shoulder_rise': f"L {f(a('L_Thorax', 2))} R {f(a('R_Thorax', 2))}",
            'upper_arm_abducted': f"L {f(a('L_Shoulder', 2))} R {f(a('L_Shoulder', 1))}",
            'lower_arm_bending': f"L {f(le)} R {f(re)}",
            'bent_from_midline_or_out_to_side': f"L {f(a('L_Thorax', 0))} R {f(a('R_Thorax', 0))}",
            'wrist_bending': f"L {f(a('L_Wrist', 2))} R {f(a('R_Wrist', 2))}",
            'wrist_side_bending': f"L {f(a('L_Wrist', 1))} R {f(a('R_Wrist', 1))}",
            'wrist_twist': f"L {f(a('L_Wrist', 0))} R {f(a('R_Wrist', 0))}",
            'trunk_bending': f(a('Torso', 0)), 'trunk_side_bending': f(a('Torso', 2)),
            'trunk_twisted': f(a('Torso', 1)), 'neck_bending': f(a('Neck', 0)),
            'neck_side_bending_twisted': f"{f(a('Neck', 2))}, {f(a('Neck', 1))}",
        }

    def action_level(self, score):
        """rula.py:100-118."""
        score = round(score)
        for lo, hi, level, name in ((1, 2, 1, "Acceptable posture"),
                                    (3, 4, 2, "Further investigation, change may be needed"),
                                    (5, 6, 3, "Further investigation, change soon")):
            if lo <= score <= hi:
                return level, name
        if score >= 7:
            return 4, "Investigate and implement change"
        return None, None
