"""SPIN state dict <-> the canonical float32 weight blob `pr_hmr_create` takes.

Key names are SPIN's (what `torch.load(cfg.SPIN.checkpoint)['model']` holds, lib/core/base.py:83-84);
the order is the one documented in include/poserisk_hip.h.
"""
import numpy as np

RESNET_PLANES = (64, 128, 256, 512)
RESNET_BLOCKS = (3, 4, 6, 3)
_BN = ("weight", "bias", "running_mean", "running_var")


def blob_keys():
    """[(key, shape)] in blob order."""
    out = [("conv1.weight", (64, 3, 7, 7))] + [(f"bn1.{s}", (64,)) for s in _BN]
    inpl = 64
    for L, (pl, nb) in enumerate(zip(RESNET_PLANES, RESNET_BLOCKS), start=1):
        for b in range(nb):
            p = f"layer{L}.{b}"
            out.append((f"{p}.conv1.weight", (pl, inpl, 1, 1)))
            out += [(f"{p}.bn1.{s}", (pl,)) for s in _BN]
            out.append((f"{p}.conv2.weight", (pl, pl, 3, 3)))
            out += [(f"{p}.bn2.{s}", (pl,)) for s in _BN]
            out.append((f"{p}.conv3.weight", (pl * 4, pl, 1, 1)))
            out += [(f"{p}.bn3.{s}", (pl * 4,)) for s in _BN]
            if b == 0:
                out.append((f"{p}.downsample.0.weight", (pl * 4, inpl, 1, 1)))
                out += [(f"{p}.downsample.1.{s}", (pl * 4,)) for s in _BN]
            inpl = pl * 4
    out += [("fc1.weight", (1024, 2205)), ("fc1.bias", (1024,)), ("fc2.weight", (1024, 1024)),
            ("fc2.bias", (1024,)), ("decpose.weight", (144, 1024)), ("decpose.bias", (144,)),
            ("decshape.weight", (10, 1024)), ("decshape.bias", (10,)), ("deccam.weight", (3, 1024)),
            ("deccam.bias", (3,)), ("init_pose", (144,)), ("init_shape", (10,)), ("init_cam", (3,))]
    return out


def _to_numpy(v):
    if hasattr(v, "detach"):
        v = v.detach().cpu().numpy()
    return np.asarray(v, dtype=np.float32)


def missing_keys(state_dict):
    sd = {k[7:] if k.startswith("module.") else k: v for k, v in state_dict.items()}
    return [k for k, _ in blob_keys() if k not in sd]


def flatten_state_dict(state_dict):
    """state dict (numpy arrays or torch tensors; a leading 'module.' is stripped like
    funcs_utils.check_data_pararell does) -> contiguous float32 blob."""
    sd = {k[7:] if k.startswith("module.") else k: v for k, v in state_dict.items()}
    parts = []
    for key, shape in blob_keys():
        if key not in sd:
            raise KeyError(f"state dict lacks '{key}' (needed by the HIP encoder)")
        a = _to_numpy(sd[key])
        if a.size != int(np.prod(shape)):
            raise ValueError(f"'{key}' has {a.size} elements, expected shape {shape}")
        parts.append(a.reshape(-1))
    return np.ascontiguousarray(np.concatenate(parts), dtype=np.float32)
