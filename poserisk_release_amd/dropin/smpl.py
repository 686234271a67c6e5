"""Mirror of lib/utils/smpl.py::SMPL: three gendered SMPL layers + the metadata base.py reads.

    smpl = SMPL()                                           base.py:80
    verts, joints = smpl.layer['neutral'](pose, betas)      coord_utils.py:15, base.py:279
    smpl.face, smpl.joint_regressor, smpl.joints_name_upper, smpl.skeleton, smpl.vertex_num

Model files are looked up like the reference does (CWD-relative `data/base_data/human_models`,
smpl.py:9; `SMPL_{NEUTRAL,FEMALE,MALE}.pkl`, smpl_layer.py:30-35), or under $POSERISK_SMPL_DIR, as
`.pkl` (no chumpy needed) or `.npz`.  `SMPL(models={'neutral': dict, ...})` takes arrays directly.
"""
import os
import os.path as osp

import numpy as np

from poserisk_release_amd.smpl_io import load_smpl_model
from poserisk_release_amd.smpl_layer import SMPLLayer

_FILES = {'neutral': 'SMPL_NEUTRAL', 'female': 'SMPL_FEMALE', 'male': 'SMPL_MALE'}


class SMPL(object):
    def __init__(self, models=None, device=None):
        self.model_path = os.environ.get('POSERISK_SMPL_DIR', osp.join('data', 'base_data', 'human_models'))
        self._device = device
        self._models = models or {}
        self.layer = {g: self.get_layer(g) for g in ('male', 'female', 'neutral')}
        neutral = self.layer['neutral']
        self.vertex_num = neutral.num_verts            # 6890 for the real model
        self.face = neutral.th_faces.numpy()
        reg = neutral.th_J_regressor.numpy().astype(np.float32)
        # smpl.py:15-28: nose, L/R eye, L/R ear as one-hot rows appended to the joint regressor
        self.face_kps_vertex = (331, 2802, 6262, 3489, 3990)
        extra = np.zeros((5, reg.shape[1]), np.float32)
        for row, v in enumerate(self.face_kps_vertex):
            if v < reg.shape[1]:
                extra[row, v] = 1.0
        self.joint_regressor = np.concatenate((reg, extra))
        self.joint_num = 24
        self.joints_name = ('Pelvis', 'L_Hip', 'R_Hip', 'Torso', 'L_Knee', 'R_Knee', 'Spine', 'L_Ankle', 'R_Ankle',
                            'Chest', 'L_Toe', 'R_Toe', 'Neck', 'L_Thorax', 'R_Thorax', 'Head', 'L_Shoulder',
                            'R_Shoulder', 'L_Elbow', 'R_Elbow', 'L_Wrist', 'R_Wrist', 'L_Hand', 'R_Hand')
        self.joints_name_upper = [n.upper() for n in self.joints_name]
        self.flip_pairs = ((1, 2), (4, 5), (7, 8), (10, 11), (13, 14), (16, 17), (18, 19), (20, 21), (22, 23),
                           (25, 26), (27, 28))
        self.skeleton = ((0, 1), (1, 4), (4, 7), (7, 10), (0, 2), (2, 5), (5, 8), (8, 11), (0, 3), (3, 6), (6, 9),
                         (9, 14), (14, 17), (17, 19), (19, 21), (21, 23), (9, 13), (13, 16), (16, 18), (18, 20),
                         (20, 22), (9, 12), (12, 15))
        self.root_joint_idx = self.joints_name.index('Pelvis')

    def get_layer(self, gender):
        if gender in self._models:
            model = self._models[gender]
        elif 'neutral' in self._models and not self._find(gender):
            model = self._models['neutral']
        else:
            path = self._find(gender)
            if path is None:
                raise FileNotFoundError(
                    f"{_FILES[gender]}.pkl/.npz not found under '{self.model_path}' (licensed SMPL download, "
                    "reference README.md:36); pass SMPL(models=...) or set POSERISK_SMPL_DIR")
            model = load_smpl_model(path)
        return SMPLLayer(model, gender=gender, device=self._device)

    def _find(self, gender):
        for ext in ('.pkl', '.npz'):
            p = osp.join(self.model_path, _FILES[gender] + ext)
            if osp.isfile(p):
                return p
        return None
