"""RULA scorer restated as vectorised numpy (first-true-wins rule chains).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Pinned by tests/golden/scores.npz
(reference ``RULA.__call__`` run on the same angles, exact integer match).

Follows ``lib/utils/rula.py:66-98`` (driver), ``:120-156`` (groups) and ``:158-422``
(rules), including SURVEY.md 7.3 Q15-Q17.
"""
import numpy as np

from .reba_ref import chain, pymax
from .risk_tables import J, RULA_A, RULA_B, RULA_C


def rula_subscores(pose, info):
    """pose f64[N,24,3] degrees, info = add_info["RULA"] -> dict of int arrays; one frame f64[24,3] -> scalars."""
    P = np.asarray(pose, dtype=np.float64)
    g = (lambda name, k: P[J[name], k]) if P.ndim == 2 else (lambda name, k: P[:, J[name], k])
    l2, l1 = g('L_Shoulder', 2), g('L_Shoulder', 1)
    r2, r1 = g('R_Shoulder', 2), g('R_Shoulder', 1)

    # --- group A ----------------------------------------------------------------
    # upper_arm_bending, rula.py:158-199
    ua_l = chain(1,
                 ((l2 > -70) & (l2 < 110),
                  chain(1, (np.abs(l1) < 20, 1), ((l1 > 20) | ((l1 > -45) & (l1 < -20)), 2),
                        ((l1 > -90) & (l1 <= -45), 3), (l1 < -90, 4))),
                 (l2 > -20,
                  chain(1, (np.abs(l1) < 20, 1), ((l1 > 20) & (l1 < 70), 2), (l1 > 70, 2),
                        ((l1 > -70) & (l1 < -20), 4), (l1 < -70, 4)))) - info["Arm_supported_leaning_L"]
    ua_r = chain(1,
                 ((r2 > -70) & (r2 < 110),
                  # rula.py:182-183: "angle4=1" assigns the ANGLE, so the score stays 0 (Q15)
                  chain(1, (np.abs(r1) < 20, 0), ((r1 < -20) | ((r1 > 20) & (r1 <= 45)), 2),
                        ((r1 > 45) & (r1 <= 90), 3), (r1 > 90, 4))),
                 (r2 < 20,
                  chain(1, (np.abs(r1) < 20, 1), ((r1 > -70) & (r1 < -20), 2), (r1 < -70, 2),
                        ((r1 > 20) & (r1 < 70), 4), (r1 > 70, 4)))) - info["Arm_supported_leaning_R"]

    # shoulder_rise, rula.py:201-217
    rise = lambda a: chain(0, (np.abs(a) < 10, 0), (np.abs(a) >= 10, 1))
    ua_l = ua_l + rise(g('L_Thorax', 2))
    ua_r = ua_r + rise(g('R_Thorax', 2))

    # upper_arm_abducted, rula.py:249-288 (right arm has no trailing else: stays 0)
    ua_l = ua_l + chain(0,
                        ((l2 > -110) & (l2 < -20), chain(0, (l2 < 45, 0), (l2 > 45, 1))),
                        (l2 > -20,
                         chain(0, (np.abs(l1) < 20, 1), ((l1 > 20) & (l1 < 70), 1), (l1 > 70, 0),
                               ((l1 > -70) & (l1 < -20), 1), (l1 < -70, 0))))
    ua_r = ua_r + chain(0,
                        ((r2 > 20) & (r2 < 110), chain(0, (r2 > 45, 0), (r2 < 45, 1))),
                        (r2 < 20,
                         chain(0, (np.abs(r1) < 20, 1), ((r1 > -70) & (r1 < -20), 1), (r1 < -70, 0),
                               ((r1 > 20) & (r1 < 70), 1), (r1 > 70, 0))))

    # lower_arm_bending, rula.py:290-309
    a = pymax(g('L_Elbow', 1), g('L_Elbow', 2))
    la_l = chain(1, ((a > -100) & (a < -60), 1), ((a < -100) | ((a > -60) & (a < 0)), 2))
    a = pymax(g('R_Elbow', 1), g('R_Elbow', 2))
    la_r = chain(1, ((a > 60) & (a < 100), 1), ((a > 100) | ((a > 0) & (a < 60)), 2))
    # bent_from_midline_or_out_to_side, rula.py:311-326 (Q16)
    a = g('L_Thorax', 0)
    la_l = la_l + chain(0, ((a < 10) | ((a > -45) & (a < -10)), 0), ((a > 10) | (a < -45), 1))
    a = g('R_Thorax', 0)
    la_r = la_r + chain(0, ((a > -10) | ((a > 10) & (a < 45)), 0), ((a < -10) | (a > 45), 1))

    # wrist_bending :328-346, wrist_side_bending :348-363, wrist_twist :365-380
    bend = lambda a: chain(1, (np.abs(a) < 1, 1), ((np.abs(a) > 1) & (np.abs(a) < 15), 2), (np.abs(a) > 15, 3))
    side = lambda a: chain(0, (np.abs(a) < 10, 0), (np.abs(a) > 10, 1))
    twist = lambda a: chain(1, (np.abs(a) < 45, 1), (np.abs(a) > 45, 2))
    wr_l = bend(g('L_Wrist', 2)) + side(g('L_Wrist', 1))
    wr_r = bend(g('R_Wrist', 2)) + side(g('R_Wrist', 1))
    wt_l, wt_r = twist(g('L_Wrist', 0)), twist(g('R_Wrist', 0))

    ua_l, ua_r = np.clip(ua_l, 1, 6), np.clip(ua_r, 1, 6)
    la_l, la_r = np.clip(la_l, 1, 3), np.clip(la_r, 1, 3)
    wr_l, wr_r = np.clip(wr_l, 1, 4), np.clip(wr_r, 1, 4)
    wt_l, wt_r = np.clip(wt_l, 1, 2), np.clip(wt_r, 1, 2)
    a_l = RULA_A[ua_l - 1, la_l - 1, wr_l - 1, wt_l - 1] + info["A_Muscle_use_L"] + info["A_Load/Force_L"]
    a_r = RULA_A[ua_r - 1, la_r - 1, wr_r - 1, wt_r - 1] + info["A_Muscle_use_R"] + info["A_Load/Force_R"]
    score_a = np.maximum(a_l, a_r)

    # --- group B, rula.py:143-156 ---------------------------------------------------
    a = g('Neck', 0)                                                        # rula.py:406-414
    neck = chain(1, ((a > -5) & (a < 10), 1), ((a > 10) & (a < 20), 2), (a > 20, 3), (a < -5, 4))
    a1, a2 = g('Neck', 2), g('Neck', 1)                                     # rula.py:416-422
    neck = neck + chain(0, ((np.abs(a1) < 10) & (np.abs(a2) < 10), 0), ((np.abs(a1) > 10) | (np.abs(a2) > 10), 1))
    a = g('Torso', 0)                                                       # rula.py:382-390
    trunk = chain(1, (np.abs(a) < 5, 1), ((a > 5) & (a < 20), 2), ((a > 20) & (a < 60), 3), (a > 60, 4))
    trunk = trunk + side(g('Torso', 1)) + side(g('Torso', 2))               # rula.py:392-404
    leg = np.zeros_like(trunk) + info["Legs_bilateral_weight_bearing"]
    neck, trunk, leg = np.clip(neck, 1, 6), np.clip(trunk, 1, 6), np.clip(leg, 1, 2)
    score_b = RULA_B[neck - 1, trunk - 1, leg - 1] + info["B_Muscle_use"] + info["B_Load/Force"]

    # --- final, rula.py:83-85 ---------------------------------------------------------
    score = RULA_C[np.clip(score_a, 1, 7) - 1, np.clip(score_b, 1, 7) - 1]
    return dict(score=score, upper_arm=np.stack([ua_l, ua_r], -1), lower_arm=np.stack([la_l, la_r], -1),
                wrist=np.stack([wr_l, wr_r], -1), wrist_twist=np.stack([wt_l, wt_r], -1),
                neck=neck, trunk=trunk, leg=leg)


def rula_packed(pose, info):
    """int32[N,12]: score, uaL, uaR, laL, laR, wrL, wrR, wtL, wtR, neck, trunk, leg."""
    s = rula_subscores(pose, info)
    cols = [s['score'], s['upper_arm'], s['lower_arm'], s['wrist'], s['wrist_twist'], s['neck'], s['trunk'], s['leg']]
    if np.ndim(s['score']) == 0:        # one frame f64[24,3] -> int32[1,12]
        return np.concatenate([np.atleast_1d(c) for c in cols]).astype(np.int32)[None]
    return np.column_stack(cols).astype(np.int32)


def rula_call(pose, info):
    """Same return shape as ``RULA.__call__`` (rula.py:87-91)."""
    p = rula_packed(pose, info)
    return [{'score': np.int64(r[0]),
             'log_score': [f'{r[1]},{r[2]}', f'{r[3]},{r[4]}', f'{r[5]},{r[6]}', f'{r[7]},{r[8]}',
                           int(r[9]), int(r[10]), int(r[11])]}
            for r in p]
