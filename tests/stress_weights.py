"""A "trained-like" synthetic SPIN state dict for numerics stress tests (test infrastructure; uses the CPU oracle).

The benign synthetic weights (poserisk_release_amd/synth.py: He-normal filters, BatchNorm gamma = 1, variance in
[0.5, 1.5], decoder gain 0.01) have none of what a trained, BN-folded checkpoint has and what Winograd F(4x4,3x3)
in fp32 is sensitive to.  The real checkpoint is a licensed download and absent, so this builds the properties:

  * heavy-tailed filters: every weight is multiplied by exp(N(0, 0.8^2)), every input channel by exp(N(0, 1)) and
    every output channel by exp(N(0, 1.5^2)) -- conv outputs then span ~1e-4 .. 1e+4 in variance across channels;
  * BatchNorm running_mean / running_var are what a trained network holds: the statistics of the conv outputs on
    data (one calibration pass of the oracle in training mode, momentum 1), so running_var spans the decades above;
  * gamma log-uniform in [0.1, 10] (bn1, bn2) and [0.05, 2] (bn3, downsample), beta ~ N(0, 0.5 gamma): sparse,
    offset, outlier-rich post-ReLU activations;
  * decoder layers with 30x the synthetic gain, so that rotation matrices react to encoder-feature errors the way a
    trained regressor's O(1) weights would.
"""
import numpy as np
import torch

from oracle import hmr_ref
from poserisk_release_amd import synth


def trained_like_state_dict(seed=5, calib_frames=8, dec_gain=30.0):
    rng = np.random.default_rng(seed)
    sd = {k: np.array(v, copy=True) for k, v in synth.hmr_state_dict(seed=1).items()}
    for k in list(sd):
        if k.endswith("weight") and sd[k].ndim == 4:
            co, ci = sd[k].shape[:2]
            w = sd[k] * np.exp(rng.normal(0.0, 0.8, sd[k].shape))
            if ci > 3:
                w = w * np.exp(rng.normal(0.0, 1.0, (1, ci, 1, 1)))
            w = w * np.exp(rng.normal(0.0, 1.5, (co, 1, 1, 1)))
            sd[k] = w.astype(np.float32)
        elif k.endswith("running_var"):
            base = k[: -len("running_var")]
            c = sd[k].shape[0]
            lo, hi = (0.05, 2.0) if (base.endswith("bn3.") or base.endswith("downsample.1.")) else (0.1, 10.0)
            gamma = np.exp(rng.uniform(np.log(lo), np.log(hi), c))
            sd[base + "weight"] = gamma.astype(np.float32)
            sd[base + "bias"] = (rng.normal(0.0, 0.5, c) * gamma).astype(np.float32)
    for k in ("decpose.weight", "decshape.weight", "deccam.weight"):
        sd[k] = (sd[k] * dec_gain).astype(np.float32)
    # BatchNorm statistics from data, as training leaves them
    model = hmr_ref.build(sd)
    for mod in model.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.momentum = 1.0
    model.train()
    with torch.no_grad():
        model.features(torch.from_numpy(synth.crops(calib_frames, seed=77)))
    model.eval()
    out = {k: v.detach().numpy().astype(np.float32) for k, v in model.state_dict().items()
           if not k.endswith("num_batches_tracked")}
    for k in ("init_pose", "init_shape", "init_cam"):
        out[k] = sd[k]
    return out
