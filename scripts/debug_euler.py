import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import coord_ref, rodrigues_cv
from poserisk_release_amd import ops
g = np.load('tests/golden/euler.npz')
dev = torch.device('cuda', 0)
aa, eul, st = ops.pose_to_euler(torch.from_numpy(g['rotmat']).to(dev))
aa = aa.cpu().numpy(); eul = eul.cpu().numpy()
ref = np.stack([coord_ref.axis_angle_to_euler_angle(f) for f in aa])
d = np.abs(eul - ref); d = np.minimum(d, 360 - d)
idx = np.argsort(d.reshape(-1))[::-1][:12]
for i in idx:
    f, j, k = np.unravel_index(i, d.shape)
    R = rodrigues_cv.rotvec_to_rotmat(aa[f, j])
    print(f, j, k, 'aa', aa[f, j], 'ours', eul[f, j], 'ref', ref[f, j], 'd', d[f, j, k])
    print('   R', R.reshape(-1), 'sy', np.sqrt(R[0,0]*R[0,0]+R[1,0]*R[1,0]))
print('frac bad', np.mean(d > 1e-9), 'per-joint any', np.mean((d > 1e-9).any(-1)))
