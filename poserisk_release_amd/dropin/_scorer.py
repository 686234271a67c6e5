"""Shared host logic of the REBA / RULA mirrors: device call + the reference's result shapes."""
import numpy as np
import torch


def to_device_poses(poses):
    if isinstance(poses, torch.Tensor) and poses.device.type == "cuda":
        return poses.double()
    if not torch.cuda.is_available():
        from poserisk_release_amd._lib import PoseRiskHipError
        raise PoseRiskHipError("REBA/RULA score on an MI355X (no CPU fallback)")
    return torch.as_tensor(np.asarray(poses, dtype=np.float64)).to(torch.device("cuda", torch.cuda.current_device()))
