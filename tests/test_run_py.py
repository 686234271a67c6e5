"""main/run.py against the drop-in: the reference's own import order and calls.

    import __init_path                                   main/run.py:4    (sys.path: lib, data, lib/utils, ...)
    from funcs_utils import save_checkpoint, save_plot, check_data_pararell, count_parameters        main/run.py:7
        (the reference's lib/utils/funcs_utils.py:16 does `from core.config import cfg`: with the drop-in
        installed that must resolve to the drop-in's config, before run.py's own import of it)
    from core.config import cfg, update_config           main/run.py:8
    args = parser.parse_args()                           main/run.py:10-21
    os.environ['CUDA_VISIBLE_DEVICES'] = str(args.gpu)   main/run.py:26
    from core.base import Predictor                      main/run.py:29
    predictor = Predictor(args)                          main/run.py:31   ONE argument
    predictor(args.input, args.info, args.output)        main/run.py:32

The reference's files never travel, so the sequence is written out here and run in a fresh interpreter from the
root of a stand-in checkout (empty `lib/core/base.py` that must NOT be the module imported, licensed assets
replaced by synthetic ones at the reference's locations: lib/core/config.py:45-47, lib/utils/smpl.py:9).
"""
import json
import os
import pickle
import subprocess
import sys
import types

import numpy as np
import pytest
import torch

from conftest import REPO
from poserisk_release_amd import synth


def _checkout(root, with_assets=True, V=6890):
    """A directory shaped like the PoseRisk checkout at the moment main/run.py starts."""
    (root / "main").mkdir(parents=True)
    (root / "lib" / "core").mkdir(parents=True)
    (root / "lib" / "utils").mkdir()
    (root / "lib" / "core" / "__init__.py").write_text("")
    (root / "lib" / "core" / "base.py").write_text("raise ImportError('reference lib/core/base.py imported: the drop-in must shadow it')\n")
    (root / "lib" / "core" / "config.py").write_text("raise ImportError('reference lib/core/config.py imported (needs easydict)')\n")
    # stand-in for the reference's lib/utils/funcs_utils.py (which needs cv2): its first project import and the four
    # names main/run.py:7 takes from it
    (root / "lib" / "utils" / "funcs_utils.py").write_text(
        "from core.config import cfg\n"
        "def save_checkpoint(*a, **k): raise NotImplementedError\n"
        "def save_plot(*a, **k): raise NotImplementedError\n"
        "def check_data_pararell(*a, **k): raise NotImplementedError\n"
        "def count_parameters(*a, **k): raise NotImplementedError\n")
    (root / "main" / "default_information.json").write_text(json.dumps(synth.DEFAULT_INFO))
    if not with_assets:
        return
    spin = root / "lib" / "SPIN" / "data"
    spin.mkdir(parents=True)
    sd = synth.hmr_state_dict(seed=1)
    np.savez(spin / "smpl_mean_params.npz", pose=sd["init_pose"].reshape(-1), shape=sd["init_shape"].reshape(-1),
             cam=sd["init_cam"].reshape(-1))
    # SPIN's checkpoint: {'model': state_dict of tensors, 'optimizer': ..., ...}; torchvision-style extra keys
    model = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items() if not k.startswith("init_")}
    model["bn1.num_batches_tracked"] = torch.tensor(0)
    model["fc.weight"] = torch.zeros(4, 4)                     # unused head: strict=False must let it pass
    torch.save({"model": model, "epoch": 3, "optimizer": {"state": {}, "param_groups": []}},
               spin / "model_checkpoint.pt")
    models = root / "data" / "base_data" / "human_models"
    models.mkdir(parents=True)
    m = synth.smpl_model(V=V, seed=2)
    kt = np.stack([np.asarray(m["parents"]).astype(np.uint32), np.arange(24, dtype=np.uint32)])
    dd = dict(v_template=m["v_template"], shapedirs=m["shapedirs"], posedirs=m["posedirs"], weights=m["weights"],
              J_regressor=m["J_regressor"], kintree_table=kt, f=np.arange(12, dtype=np.uint32).reshape(4, 3))
    with open(models / "SMPL_NEUTRAL.pkl", "wb") as fh:
        pickle.dump(dd, fh, protocol=2)
    for g in ("MALE", "FEMALE"):                               # smpl.py:10 loads all three genders
        os.symlink(models / "SMPL_NEUTRAL.pkl", models / f"SMPL_{g}.pkl")


def _clip(root):
    rng = np.random.default_rng(9)
    frames = rng.integers(0, 256, (9, 240, 320, 3), dtype=np.uint8)
    fr = [1, 2, 3, 4, 5, 6, 8]
    tr = {8: {'bbox': np.stack([np.array([160 + 3 * i, 120 - 2 * i, 90, 180], np.float32) for i in range(len(fr))]),
              'frames': np.array(fr)}}
    src = root / "example"
    src.mkdir()
    np.save(src / "frames.npy", frames)
    with open(src / "tracking.pkl", "wb") as f:
        pickle.dump(tr, f)
    (src / "additional_information.json").write_text(json.dumps(synth.EXAMPLE_INFO))
    return src


# What main/run.py executes, line for line, with INTEGRATION.md's one added line (dropin.install()) where
# main/__init_path.py ends.  argv[1] = this repository, the process runs from the checkout's root.
_RUN_PY = r'''
import os, sys
import os.path as osp
import argparse
import torch
# ---- main/__init_path.py:14-32 (this_dir = <checkout>/main)
this_dir = osp.join(os.getcwd(), 'main')
for rel in (('..', 'lib'), ('..', 'data'), ('..', 'lib', 'utils'), ('..', 'lib', 'multi_person_tracker'),
            ('..', 'lib', 'SPIN'), ('..', 'lib', 'smplpytorch')):
    p = osp.join(this_dir, *rel)
    if p not in sys.path:
        sys.path.insert(0, p)
# ---- the line INTEGRATION.md adds at the end of main/__init_path.py
sys.path.insert(0, sys.argv[1])
import poserisk_release_amd.dropin as dropin
dropin.install()
del sys.argv[1]
# ---- main/run.py:7
from funcs_utils import save_checkpoint, save_plot, check_data_pararell, count_parameters
# ---- main/run.py:8
from core.config import cfg, update_config
# ---- main/run.py:10-21
parser = argparse.ArgumentParser(description='Estimate RULA and REBA score')
parser.add_argument('--gpu', type=str, default='0', help='assign multi-gpus by comma concat')
parser.add_argument('--type', type=str, default='REBA,RULA', help='Score type')
parser.add_argument('--input', type=str, default='example/input.mp4', help='input video')
parser.add_argument('--info', type=str, default='example/additional_information.json')
parser.add_argument('--output', type=str, default='output', help='output directory')
parser.add_argument('--visualize', type=bool, default=True, help='do result visualization')
parser.add_argument('--debug', action='store_true', help='for debuging')
parser.add_argument('--debug_joints', type=str, default='')
parser.add_argument('--debug_frame', type=int, default=-1)
args = parser.parse_args()
# ---- main/run.py:26-32
os.environ['CUDA_VISIBLE_DEVICES'] = str(args.gpu)
print("Work on GPU: ", os.environ['CUDA_VISIBLE_DEVICES'])
from core.base import Predictor
predictor = Predictor(args)
out = predictor(args.input, args.info, args.output)
# ---- end of main/run.py; what follows checks the run
import core.base, core.config, funcs_utils
assert osp.realpath(funcs_utils.__file__).startswith(osp.realpath(osp.join(os.getcwd(), 'lib', 'utils'))), funcs_utils.__file__   # the checkout's own module
assert funcs_utils.cfg is cfg and core.config.__file__.startswith(osp.dirname(dropin.__file__)), core.config.__file__
assert core.base.__file__.startswith(osp.dirname(dropin.__file__)), core.base.__file__
assert cfg.root_dir == os.getcwd(), (cfg.root_dir, os.getcwd())
import json, pickle
import numpy as np
from poserisk_release_amd import synth
from models import hmr
from smpl import SMPL
assert predictor.precision == cfg.SPIN.precision == os.environ.get("EXPECT_PRECISION", "fp32"), (predictor.precision, cfg.SPIN.precision)
assert predictor.lanes == cfg.DATASET.hip_lanes == int(os.environ.get("EXPECT_LANES", "2")), predictor.lanes
assert predictor.spin_model._precision == {"fp32": 0, "bf16": 1}[predictor.precision]      # the encoder that really ran
model = hmr(precision=predictor.precision)
model.load_state_dict(synth.hmr_state_dict(seed=1), strict=False)
smpl = SMPL(models={"neutral": synth.smpl_model(V=6890, seed=2)})
injected = Predictor(args, spin_model=model, smpl_model=smpl, batch_size=4)
frames = np.load(osp.join(args.input, 'frames.npy'))
with open(osp.join(args.input, 'tracking.pkl'), 'rb') as f:
    tr = pickle.load(f)
want = injected.score_frames(frames, tr, json.load(open(args.info)))
for k in ("result", "joint_cam", "debug_result"):
    assert out[k].shape == want[k].shape and np.array_equal(out[k], want[k]), k
for k in ("reba", "rula"):
    assert np.array_equal(out[k][1], want[k][1]) and np.array_equal(out[k][2], want[k][2]), k
    assert np.array_equal(np.array(out[k][0], float), np.array(want[k][0], float), equal_nan=True), k
if predictor.precision == "bf16":
    # what the one config line costs: the same clip through the fp32 encoder (the agreement tests/test_hip_parity.py's
    # bf16 pipeline test measures on 256 frames: rotmat within 5e-2, most frames the same scores)
    m32 = hmr()
    m32.load_state_dict(synth.hmr_state_dict(seed=1), strict=False)
    args32 = argparse.Namespace(**dict(vars(args), dtype="fp32"))        # args.dtype wins over cfg.SPIN.precision
    p32 = Predictor(args32, spin_model=m32, smpl_model=smpl, batch_size=4)
    assert p32.precision == "fp32"
    f32 = p32.score_frames(frames, tr, json.load(open(args.info)))
    assert not np.array_equal(out["debug_result"], f32["debug_result"])          # another encoder really ran
    d = np.abs(out["debug_result"] - f32["debug_result"]).max()
    de = np.abs(out["result"] - f32["result"]); de = np.minimum(de, 360 - de)
    both = float(((np.abs(out["reba"][1] - f32["reba"][1]) <= 1) & (np.abs(out["rula"][1] - f32["rula"][1]) <= 1)).mean())
    print("BF16-VS-FP32 axis-angle max %.3e rad, euler median %.3e deg, frames with both scores within one point %.2f" % (d, float(np.median(de)), both))
    assert d < 5e-2 and both >= 0.5, (d, both)
print("RUNPY-OK", out["result"].shape[0], out["reba"][0][4], out["rula"][0][4])
'''


@pytest.mark.gpu
def test_run_py_sequence_against_the_dropin(gpu_device, tmp_path):
    root = tmp_path / "PoseRisk"
    _checkout(root)
    _clip(root)
    script = tmp_path / "run_py_replay.py"
    script.write_text(_RUN_PY)
    env = {k: v for k, v in os.environ.items() if k not in ("POSERISK_ROOT", "POSERISK_SMPL_DIR")}
    r = subprocess.run([sys.executable, str(script), REPO, "--input", "example", "--output", "output"],
                       cwd=str(root), env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "RUNPY-OK 7" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    for name in ("reba_result.txt", "rula_result.txt", "REBA_score.png", "RULA_score.png"):
        assert (root / "output" / name).is_file(), name
    assert (root / "output" / "reba_result.txt").read_text().startswith("AVG Score: ")


@pytest.mark.gpu
def test_run_py_with_one_config_line_runs_the_bf16_encoder(gpu_device, tmp_path):
    """BASELINE configs[2] through the plugin surface: main/run.py unchanged (its --cfg option is commented out, run.py:20-24),
    ONE config line -- `SPIN: {precision: bf16}` in the YAML $POSERISK_CFG names -- and `Predictor(args)` builds the bf16
    encoder (lib/core/base.py:81 -> hmr(cfg.SPIN.SMPL_MEAN_PARAMS)); `DATASET: {hip_lanes: 3}` rides along.  The replay
    checks the run bit for bit against a Predictor with an injected bf16 model and, beside it, against the fp32 encoder
    within the agreement the bf16 pipeline test measures."""
    root = tmp_path / "PoseRisk"
    _checkout(root)
    _clip(root)
    (root / "mi355x.yaml").write_text("SPIN:\n  precision: bf16\nDATASET:\n  hip_lanes: 3\n")
    script = tmp_path / "run_py_replay.py"
    script.write_text(_RUN_PY)
    env = {k: v for k, v in os.environ.items() if k not in ("POSERISK_ROOT", "POSERISK_SMPL_DIR")}
    env.update(POSERISK_CFG=str(root / "mi355x.yaml"), EXPECT_PRECISION="bf16", EXPECT_LANES="3")
    r = subprocess.run([sys.executable, str(script), REPO, "--input", "example", "--output", "output"],
                       cwd=str(root), env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "RUNPY-OK 7" in r.stdout and "BF16-VS-FP32" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    print([ln for ln in r.stdout.splitlines() if ln.startswith("BF16-VS-FP32")][0])
    assert (root / "output" / "reba_result.txt").read_text().startswith("AVG Score: ")


def _args(**kw):
    d = dict(gpu="0", type="REBA,RULA", input="example/input.mp4", info="example/additional_information.json",
             output="output", visualize=True, debug=False, debug_joints="", debug_frame=-1)
    d.update(kw)
    return types.SimpleNamespace(**d)


_MISSING = r'''
import os, sys
sys.path.insert(0, os.path.join(os.getcwd(), 'lib'))
sys.path.insert(0, sys.argv[1])
import poserisk_release_amd.dropin as dropin
dropin.install()
from core.config import cfg, update_config
import types
from core.base import Predictor, load_spin_model, default_information
from poserisk_release_amd import synth
assert cfg.DATASET.batch_size == 8 and cfg.DATASET.workers == 16 and cfg.DATASET.bbox_scale == 1.2
assert cfg["DATASET"]["min_frame_ratio"] == 0.33 and cfg.SPIN.IMG_RES == 224 and cfg.MODEL.input_shape == (224, 224)
assert cfg.SPIN.checkpoint == os.path.join(os.getcwd(), 'lib', 'SPIN', 'data', 'model_checkpoint.pt')
assert cfg.SPIN.SMPL_MEAN_PARAMS.endswith(os.path.join('lib', 'SPIN', 'data', 'smpl_mean_params.npz'))
assert default_information() == synth.DEFAULT_INFO            # main/default_information.json of the checkout
args = types.SimpleNamespace(gpu='0', type='REBA,RULA', debug=False, debug_joints='', debug_frame=-1)
try:
    Predictor(args)
    raise SystemExit("no error for the missing SMPL model")
except FileNotFoundError as e:
    assert "SMPL_" in str(e) and "human_models" in str(e), e
from smpl import SMPL
smpl = SMPL(models={"neutral": synth.smpl_model(V=50, seed=4)})
try:
    Predictor(args, smpl_model=smpl)
    raise SystemExit("no error for the missing SPIN files")
except FileNotFoundError as e:
    assert "smpl_mean_params.npz" in str(e) and "cfg.SPIN.SMPL_MEAN_PARAMS" in str(e), e
import numpy as np, torch
os.makedirs(os.path.dirname(cfg.SPIN.checkpoint))
sd = synth.hmr_state_dict(seed=1)
np.savez(cfg.SPIN.SMPL_MEAN_PARAMS, pose=sd["init_pose"].reshape(-1), shape=sd["init_shape"].reshape(-1), cam=sd["init_cam"].reshape(-1))
try:
    Predictor(args, smpl_model=smpl)
    raise SystemExit("no error for the missing checkpoint")
except FileNotFoundError as e:
    assert "model_checkpoint.pt" in str(e) and "cfg.SPIN.checkpoint" in str(e), e
torch.save({"state_dict": {}}, cfg.SPIN.checkpoint)
try:
    Predictor(args, smpl_model=smpl)
    raise SystemExit("no error for a checkpoint without 'model'")
except KeyError as e:
    assert "'model'" in str(e), e
model = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items() if not k.startswith("init_")}
del model["layer2.1.conv2.weight"]
torch.save({"model": model}, cfg.SPIN.checkpoint)
try:
    Predictor(args, smpl_model=smpl)
    raise SystemExit("no error for an incomplete checkpoint")
except KeyError as e:
    assert "layer2.1.conv2.weight" in str(e), e
model = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items() if not k.startswith("init_")}
torch.save({"model": {"module." + k: v for k, v in model.items()}}, cfg.SPIN.checkpoint)      # DataParallel prefix
p = Predictor(args, smpl_model=smpl)
assert p.batch_size == 64 and p.run_reba and p.run_rula and p.debug_joints is None
assert not p.spin_model.load_state_dict({}, strict=False)[0]          # nothing missing: the file's weights are in
open('t.yaml', 'w').write('DATASET:\n  hip_batch_size: 16\n  bbox_scale: 1.1\n')
update_config('t.yaml')
assert Predictor(args, smpl_model=smpl).batch_size == 16 and cfg.DATASET.bbox_scale == 1.1
assert cfg.SPIN.precision == 'fp32' and cfg.DATASET.hip_lanes == 2 and cfg.DATASET.hip_world_size == 0      # MI355X knobs' defaults
from core.base import encoder_precision
assert encoder_precision() == 'fp32' and encoder_precision(types.SimpleNamespace(dtype='bfloat16')) == 'bf16'
open('t.yaml', 'w').write('SPIN:\n  precision: bf16\nDATASET:\n  hip_lanes: 3\n')
update_config('t.yaml')
p = Predictor(args, smpl_model=smpl)
assert p.precision == 'bf16' and p.spin_model._precision == 1 and p.lanes == 3 and p.world_size == 1
assert Predictor(types.SimpleNamespace(dtype='fp32', lanes=1, **vars(args)), smpl_model=smpl).precision == 'fp32'   # args win
try:
    Predictor(types.SimpleNamespace(dtype='fp8', **vars(args)), smpl_model=smpl)
    raise SystemExit("unknown dtype accepted")
except ValueError as e:
    assert "fp8" in str(e)
try:
    Predictor(types.SimpleNamespace(world_size=8, **vars(args)), smpl_model=smpl)
    raise SystemExit("world size 8 accepted in a process started alone")
except RuntimeError as e:
    assert "torch.distributed.run" in str(e) and "--nproc-per-node 8" in str(e), e
open('t.yaml', 'w').write('DATASET:\n  nope: 1\n')
try:
    update_config('t.yaml')
    raise SystemExit("unknown key accepted")
except ValueError as e:
    assert "DATASET.nope" in str(e)
print("CFG-OK")
'''


def test_predictor_one_argument_construction_and_config(tmp_path):
    """CPU: `core.config` resolves to the drop-in's mirror, its paths are the reference's (relative to the checkout
    found through sys.path), and `Predictor(args)` names whichever licensed file is missing at construction."""
    root = tmp_path / "PoseRisk"
    _checkout(root, with_assets=False)
    script = tmp_path / "cfg_check.py"
    script.write_text(_MISSING)
    env = {k: v for k, v in os.environ.items() if k not in ("POSERISK_ROOT", "POSERISK_SMPL_DIR")}
    r = subprocess.run([sys.executable, str(script), REPO], cwd=str(root), env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "CFG-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.gpu
def test_validate_checkpoint_script_on_the_stand_in_checkpoint(gpu_device, tmp_path):
    """scripts/validate_checkpoint.py -- the one command a licence holder runs on the REAL model_checkpoint.pt -- against the
    stand-in checkout's files (same formats, synthetic weights): it loads SPIN's checkpoint dict, the mean parameters and
    SMPL_NEUTRAL.pkl by the reference's names, prints the conv-form table (every form inside 1e-4 of the fp32 oracle) and
    the bf16-vs-fp32 score agreement, and exits 0."""
    root = tmp_path / "PoseRisk"
    _checkout(root)
    spin = root / "lib" / "SPIN" / "data"
    r = subprocess.run([sys.executable, os.path.join(REPO, "scripts", "validate_checkpoint.py"),
                        "--checkpoint", str(spin / "model_checkpoint.pt"), "--mean-params", str(spin / "smpl_mean_params.npz"),
                        "--smpl-dir", str(root / "data" / "base_data" / "human_models"), "--frames", "8"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    rows = [ln for ln in r.stdout.splitlines() if ln.strip()[:1].isdigit() and "|" in ln]
    assert [int(ln.split("|")[0]) for ln in rows] == [0, 2, 4, 5] and all(ln.rstrip().endswith("yes") for ln in rows), r.stdout
    assert "REBA score identical on" in r.stdout and "fp32 oracle vs fp64" in r.stdout
    # a checkpoint without the encoder's tensors is named, not crashed on
    bad = tmp_path / "bad.pt"
    torch.save({"model": {"fc1.weight": torch.zeros(2, 2)}}, bad)
    r = subprocess.run([sys.executable, os.path.join(REPO, "scripts", "validate_checkpoint.py"), "--checkpoint", str(bad),
                        "--mean-params", str(spin / "smpl_mean_params.npz")], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "lacks" in r.stderr
