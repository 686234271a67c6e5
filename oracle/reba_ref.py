"""REBA scorer restated as vectorised numpy (first-true-wins rule chains).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Pinned by tests/golden/scores.npz
(reference ``REBA.__call__`` run on the same angles, exact integer match).

Follows ``lib/utils/reba.py:50-81`` (driver), ``:106-138`` (groups) and ``:140-392``
(rules), including the behaviours listed in SURVEY.md 7.3 (Q11-Q14, Q17, Q18):
every comparison is strict, equality falls through to the trailing default,
``joint_cam`` is never read.
"""
import numpy as np

from .risk_tables import J, REBA_A, REBA_B, REBA_C


def chain(default, *pairs):
    """if / elif ... / else over arrays: ``pairs`` are (condition, value), first true wins.  With scalar conditions (one
    frame at a time, the reference's arrangement: ``for pose in poses``) it IS an if / elif chain."""
    if all(np.ndim(c) == 0 for c, _ in pairs):
        for c, v in pairs:
            if c:
                return v
        return default
    conds = [np.asarray(c) for c, _ in pairs]
    vals = [v for _, v in pairs]
    return np.select(conds, vals, default=default)


def pymax(a, b):
    """Python's builtin ``max(a, b)`` on scalars: b only if b > a (NaN-order preserving)."""
    return np.where(b > a, b, a)


def _left_open_branch(a2):
    # reba.py:213-219 (and :232-238): "angle2>20 or angle2<70" shadows everything below it (Q13)
    return chain(1, (np.abs(a2) < 20, 1), ((a2 > 20) | (a2 < 70), 2), (a2 > 70, 2),
                 ((a2 > -70) & (a2 < -20), 4), (a2 < -70, 4))


def reba_subscores(pose, info):
    """pose f64[N,24,3] degrees, info = add_info["REBA"] -> dict of int arrays [N]; one frame f64[24,3] -> dict of scalars
    (pairs as arrays [2]), computed on scalars."""
    P = np.asarray(pose, dtype=np.float64)
    g = (lambda name, k: P[J[name], k]) if P.ndim == 2 else (lambda name, k: P[:, J[name], k])
    sitting = info["Sitting"] > 0

    # --- group A ----------------------------------------------------------------
    a = g('Torso', 0)                                                       # reba.py:140-148
    trunk = chain(1, (np.abs(a) < 5, 1), (((a > 5) & (a < 20)) | ((a > -20) & (a < -5)), 2),
                  (((a > 20) & (a < 60)) | (a < -20), 3), (a > 60, 4))
    a = g('Torso', 1)                                                       # reba.py:158-164
    trunk = trunk + chain(0, (np.abs(a) < 10, 0), (np.abs(a) > 10, 1))
    trunk = trunk + 0                                                       # reba.py:150-156 (Q11)
    a = g('Neck', 0)                                                        # reba.py:166-172 (Q12)
    neck = chain(1, ((a > -5) & (a < 20), 1), ((a < 20) | (a < -5), 2))
    a1, a2 = g('Neck', 2), g('Neck', 1)                                     # reba.py:174-181
    neck = neck + chain(0, ((np.abs(a1) < 10) & (np.abs(a2) < 10), 0), ((np.abs(a1) > 10) | (np.abs(a2) > 10), 1))

    def knee(a):                                                            # reba.py:183-201
        return chain(0, (a < 30, 0), ((a > 30) & (a < 60), 1), ((a > 60) & sitting, 2))
    leg = info["Legs_bilateral_weight_bearing/walking"] + np.maximum(knee(g('L_Knee', 0)), knee(g('R_Knee', 0)))

    trunk, neck, leg = np.clip(trunk, 1, 5), np.clip(neck, 1, 3), np.clip(leg, 1, 4)
    score_a = REBA_A[trunk - 1, neck - 1, leg - 1] + info["Load/Force Score"]

    # --- group B ----------------------------------------------------------------
    l2, l1, l0 = g('L_Shoulder', 2), g('L_Shoulder', 1), g('L_Shoulder', 0)
    r2, r1, r0 = g('R_Shoulder', 2), g('R_Shoulder', 1), g('R_Shoulder', 0)

    # upper_arm_bending, reba.py:203-243
    ua_l = chain(1,
                 ((l2 > -110) & (l2 < -20),
                  chain(1, (np.abs(l1) < 20, 1), ((l1 > 20) | ((l1 > -45) & (l1 < -20)), 2),
                        ((l1 > -90) & (l1 <= -45), 3), (l1 < -90, 4))),
                 (l2 > -20, _left_open_branch(l1))) - info["Arm_supported_leaning_L"]
    ua_r = chain(1,
                 ((r2 > 20) & (r2 < 110),
                  chain(1, (np.abs(r1) < 20, 1), ((r1 < -20) | ((r1 > 20) & (r1 <= 45)), 2),
                        ((r1 > 45) & (r1 <= 90), 3), (r1 > 90, 4))),
                 (l2 > -20, _left_open_branch(l1))) - info["Arm_supported_leaning_R"]   # LEFT angles (Q13)

    # shoulder_rise, reba.py:245-260
    rise = lambda a: chain(0, (np.abs(a) < 10, 0), (np.abs(a) >= 10, 1))
    ua_l = ua_l + rise(g('L_Thorax', 2))
    ua_r = ua_r + rise(g('R_Thorax', 2))

    # upper_arm_abducted_rotated, reba.py:292-335
    in1 = (l2 > -110) & (l2 < -20)
    in2 = ~in1 & (l2 > -20)
    s1 = chain(0,
               (in1, chain(0, ((l2 < 45) & (np.abs(l0) < 10), 0), ((l2 > 45) | (np.abs(l0) > 10), 1))),
               (in2, chain(0, (np.abs(l1) < 20, 1), ((l1 > 20) | (l1 < 70), 1), (l1 > 70, 0),
                           ((l1 > -70) & (l1 < -20), 1), (l1 < -70, 0))
                + (np.abs(l0) > 10).astype(np.int64)))
    rn1 = (r2 > 20) & (r2 < 110)
    rn2 = ~rn1 & (r2 < 20)
    s2 = chain(0,
               (rn1, chain(0, ((r2 > 45) & (np.abs(r0) < 10), 0), ((r2 < 45) | (np.abs(r0) > 10), 1))),
               (rn2, chain(0, (np.abs(r1) < 20, 1), ((r1 > -70) & (r1 < -20), 1), (r1 < -70, 0),
                           ((r1 > 20) & (r1 < 70), 1), (r1 > 70, 0))))
    s1 = s1 + (rn2 & (np.abs(r0) > 10)).astype(np.int64)                    # reba.py:331 bumps LEFT (Q14)
    ua_l = ua_l + s1
    ua_r = ua_r + s2

    # lower_arm_bending, reba.py:337-356
    a = pymax(g('L_Elbow', 1), g('L_Elbow', 2))
    la_l = chain(1, ((a > -100) & (a < -60), 1), ((a < -100) | ((a > -60) & (a < 0)), 2))
    a = pymax(g('R_Elbow', 1), g('R_Elbow', 2))
    la_r = chain(1, ((a > 60) & (a < 100), 1), ((a > 100) | ((a > 0) & (a < 60)), 2))

    # wrist_bending :358-373, wrist_side_bending_or_twisted :375-392
    bend = lambda a: chain(1, (np.abs(a) < 15, 1), (np.abs(a) > 15, 2))
    side = lambda p, q: chain(0, ((np.abs(p) < 10) & (np.abs(q) < 10), 0), ((np.abs(p) > 10) | (np.abs(q) > 10), 1))
    wr_l = bend(g('L_Wrist', 2)) + side(g('L_Wrist', 1), g('L_Wrist', 0))
    wr_r = bend(g('R_Wrist', 2)) + side(g('R_Wrist', 1), g('R_Wrist', 0))

    ua_l, ua_r = np.clip(ua_l, 1, 6), np.clip(ua_r, 1, 6)
    la_l, la_r = np.clip(la_l, 1, 2), np.clip(la_r, 1, 2)
    wr_l, wr_r = np.clip(wr_l, 1, 3), np.clip(wr_r, 1, 3)
    b_l = REBA_B[ua_l - 1, la_l - 1, wr_l - 1]
    b_r = REBA_B[ua_r - 1, la_r - 1, wr_r - 1]
    score_b = np.maximum(b_l, b_r) + info["Coupling"]

    # --- final, reba.py:65-69 ------------------------------------------------------
    sa = np.clip(score_a, 1, 12)
    sb = np.clip(score_b, 1, 12)
    score = REBA_C[sa - 1, sb - 1] + info["Activity_Score"]
    return dict(score=score, trunk=trunk, neck=neck, leg=leg, upper_arm=np.stack([ua_l, ua_r], -1),
                lower_arm=np.stack([la_l, la_r], -1), wrist=np.stack([wr_l, wr_r], -1))


def reba_packed(pose, info):
    """int32[N,10]: score, trunk, neck, leg, uaL, uaR, laL, laR, wrL, wrR (the HIP kernel's record)."""
    s = reba_subscores(pose, info)
    cols = [s['score'], s['trunk'], s['neck'], s['leg'], s['upper_arm'], s['lower_arm'], s['wrist']]
    if np.ndim(s['score']) == 0:        # one frame f64[24,3] -> int32[1,10]
        return np.concatenate([np.atleast_1d(c) for c in cols]).astype(np.int32)[None]
    return np.column_stack(cols).astype(np.int32)


def reba_call(pose, info):
    """Same return shape as ``REBA.__call__`` (reba.py:71-75): list of {'score', 'log_score'}."""
    p = reba_packed(pose, info)
    return [{'score': np.int64(r[0]),
             'log_score': [int(r[1]), int(r[2]), int(r[3]), f'{r[4]},{r[5]}', f'{r[6]},{r[7]}', f'{r[8]},{r[9]}']}
            for r in p]
