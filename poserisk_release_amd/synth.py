"""Seeded synthetic stand-ins for the licensed assets the hot path needs.

The SPIN checkpoint, `smpl_mean_params.npz` and the SMPL `.pkl` files are licensed downloads
(reference README.md:36-37) and are absent here, so benchmarks and parity tests run on
random-initialised weights of the same architecture and a random SMPL model of the same sizes.
Everything is generated with NumPy PCG64 from fixed seeds and is reproducible without the
reference.
"""
import numpy as np

SMPL_PARENTS = (-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21)
RESNET_PLANES = (64, 128, 256, 512)
RESNET_BLOCKS = (3, 4, 6, 3)


def hmr_state_dict(seed=1):
    """SPIN-keyed state dict of HMR (float32 numpy arrays).

    He-normal convs (fan-out), BN gamma=1 beta=0 running_mean~N(0,0.1^2) running_var~U(0.5,1.5),
    PyTorch-default uniform FCs, Xavier(gain 0.01) decoders, init_pose = 6-D identity x24,
    init_shape = 0, init_cam = (0.9, 0, 0).
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = {}

    def conv(name, cout, cin, k):
        sd[name + ".weight"] = rng.normal(0.0, np.sqrt(2.0 / (k * k * cout)), (cout, cin, k, k)).astype(np.float32)

    def bn(name, c):
        sd[name + ".weight"] = np.ones(c, np.float32)
        sd[name + ".bias"] = np.zeros(c, np.float32)
        sd[name + ".running_mean"] = rng.normal(0.0, 0.1, c).astype(np.float32)
        sd[name + ".running_var"] = rng.uniform(0.5, 1.5, c).astype(np.float32)

    def linear(name, cout, cin, xavier_gain=None):
        if xavier_gain is None:
            b = 1.0 / np.sqrt(cin)
            sd[name + ".weight"] = rng.uniform(-b, b, (cout, cin)).astype(np.float32)
        else:
            b = xavier_gain * np.sqrt(6.0 / (cin + cout))
            sd[name + ".weight"] = rng.uniform(-b, b, (cout, cin)).astype(np.float32)
        bb = 1.0 / np.sqrt(cin)
        sd[name + ".bias"] = rng.uniform(-bb, bb, cout).astype(np.float32)

    conv("conv1", 64, 3, 7)
    bn("bn1", 64)
    inpl = 64
    for L, (pl, nb) in enumerate(zip(RESNET_PLANES, RESNET_BLOCKS), start=1):
        for b in range(nb):
            p = f"layer{L}.{b}"
            conv(p + ".conv1", pl, inpl, 1); bn(p + ".bn1", pl)
            conv(p + ".conv2", pl, pl, 3); bn(p + ".bn2", pl)
            conv(p + ".conv3", pl * 4, pl, 1); bn(p + ".bn3", pl * 4)
            if b == 0:
                conv(p + ".downsample.0", pl * 4, inpl, 1); bn(p + ".downsample.1", pl * 4)
            inpl = pl * 4
    linear("fc1", 1024, 2048 + 144 + 13)
    linear("fc2", 1024, 1024)
    linear("decpose", 144, 1024, xavier_gain=0.01)
    linear("decshape", 10, 1024, xavier_gain=0.01)
    linear("deccam", 3, 1024, xavier_gain=0.01)
    sd["init_pose"] = np.tile(np.array([1, 0, 0, 1, 0, 0], np.float32), 24).reshape(1, 144)
    sd["init_shape"] = np.zeros((1, 10), np.float32)
    sd["init_cam"] = np.array([[0.9, 0.0, 0.0]], np.float32)
    return sd


def crops(batch, seed=0):
    """f32[B,3,224,224] uniform [0,1): the un-normalised range CropDataset produces (Q1)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.random((batch, 3, 224, 224), dtype=np.float32)


def smpl_model(V=6890, seed=2, nnz=4, dense_weights=False, n_betas=10, model_betas=None):
    """Dict of the SMPL model constants with SMPL's shapes (J=24, 207 pose-blend columns).

    weights: each vertex is bound to `nnz` joints (like the real model's <=4) unless dense_weights;
    J_regressor: each joint regresses from 32 random vertices, rows sum to 1.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    J = 24
    v_template = (rng.normal(0.0, 0.3, (V, 3)) * np.array([0.6, 1.0, 0.3])).astype(np.float32)
    shapedirs = rng.normal(0.0, 0.01, (V, 3, n_betas)).astype(np.float32)
    posedirs = rng.normal(0.0, 0.005, (V, 3, (J - 1) * 9)).astype(np.float32)
    J_regressor = np.zeros((J, V), np.float32)
    for j in range(J):
        idx = rng.choice(V, size=min(32, V), replace=False)
        w = rng.random(idx.shape[0]).astype(np.float32) + 0.1
        J_regressor[j, idx] = w / w.sum()
    if dense_weights:
        weights = rng.random((V, J)).astype(np.float32)
    else:
        weights = np.zeros((V, J), np.float32)
        for v in range(V):
            idx = rng.choice(J, size=nnz, replace=False)
            weights[v, idx] = rng.random(nnz).astype(np.float32) + 0.05
    weights = (weights / weights.sum(axis=1, keepdims=True)).astype(np.float32)
    mb = np.zeros(n_betas, np.float32) if model_betas is None else np.asarray(model_betas, np.float32)
    return dict(v_template=v_template, shapedirs=shapedirs, posedirs=posedirs, J_regressor=J_regressor,
                weights=weights, parents=np.array(SMPL_PARENTS, np.int32), model_betas=mb)


def poses(n, seed=3, scale=0.3):
    """f32[n,72] axis-angle ~ N(0, scale^2)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.normal(0.0, scale, (n, 72)).astype(np.float32)


def betas(n, seed=4):
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.normal(0.0, 1.0, (n, 10)).astype(np.float32)


def rotmats(n, seed=5, scale=0.8):
    """f32[n,24,3,3] rotation matrices from random rot6d (Gram-Schmidt), like the regressor's output."""
    rng = np.random.Generator(np.random.PCG64(seed))
    x = rng.normal(0.0, scale, (n * 24, 3, 2)) + np.array([[1, 0], [0, 1], [0, 0]])
    a1, a2 = x[:, :, 0], x[:, :, 1]
    b1 = a1 / np.linalg.norm(a1, axis=1, keepdims=True)
    u = a2 - (b1 * a2).sum(1, keepdims=True) * b1
    b2 = u / np.linalg.norm(u, axis=1, keepdims=True)
    b3 = np.cross(b1, b2)
    return np.stack([b1, b2, b3], axis=-1).reshape(n, 24, 3, 3).astype(np.float32)


DEFAULT_INFO = {  # main/default_information.json
    "REBA": {"Legs_bilateral_weight_bearing/walking": 0, "Sitting": 0, "Load/Force Score": 0,
             "Arm_supported_leaning_L": 0, "Arm_supported_leaning_R": 0, "Coupling": 0, "Activity_Score": 0},
    "RULA": {"Arm_supported_leaning_L": 0, "Arm_supported_leaning_R": 0, "A_Muscle_use_L": 0,
             "A_Muscle_use_R": 0, "A_Load/Force_L": 0, "A_Load/Force_R": 0,
             "Legs_bilateral_weight_bearing": 0, "B_Muscle_use": 0, "B_Load/Force": 0},
}
EXAMPLE_INFO = {  # example/additional_information.json
    "REBA": dict(DEFAULT_INFO["REBA"], **{"Legs_bilateral_weight_bearing/walking": 1, "Sitting": 1}),
    "RULA": dict(DEFAULT_INFO["RULA"]),
}
