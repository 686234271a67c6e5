"""The Python mirror of the reference's plugin surface: names, call shapes, results."""
import json

import numpy as np
import pytest
import torch

from conftest import golden
from poserisk_release_amd import dropin, synth

dropin.install()
import coord_utils  # noqa: E402
import reba  # noqa: E402
import rula  # noqa: E402
from core import base  # noqa: E402
from models import hmr  # noqa: E402
from smpl import SMPL  # noqa: E402


def test_surface_names_and_action_levels():
    g = golden("scores.npz")
    r, u = reba.REBA(), rula.RULA()
    assert r.eval_items == ['Trunk', 'Neck', 'Leg', 'Upper_arm (L,R)', 'Lower_arm (L,R)', 'Wrist (L,R)']
    assert u.eval_items[3] == 'Wrist_twist (L,R)' and len(u.eval_items) == 7
    assert [r.action_level(s)[0] or 0 for s in range(1, 16)] == g["reba_action"].tolist()
    assert [u.action_level(s)[0] or 0 for s in range(1, 10)] == g["rula_action"].tolist()
    assert callable(hmr) and hasattr(coord_utils, "get_joint_cam") and hasattr(base, "Predictor")


def test_debug_angle_logs_match_reference():
    g = golden("scores.npz")
    rl, ul = json.loads(str(g["reba_debug_log_json"])), json.loads(str(g["rula_debug_log_json"]))
    for i in range(8):
        assert reba.REBA._angle_log(g["pose"][1 + i]) == rl[i]
        assert rula.RULA._angle_log(g["pose"][1 + i]) == ul[i]


def test_aggregate_quirks():
    a = base.aggregate(np.array([3, 7, 7, 2, 9, 4, 4, 4, 10, 1, 5]))
    assert a == (round(56 / 11, 3), 7.6, 10.0, 10, 4)
    assert np.isnan(base.aggregate(np.array([1, 2, 3]))[2])          # Q20


def test_smpl_pkl_ingest_without_chumpy(tmp_path):
    """A pickle shaped like the official SMPL file (chumpy objects, scipy sparse regressor, uint32 kintree)."""
    import pickle
    import sys
    import types

    import scipy.sparse as sp
    from poserisk_release_amd.smpl_io import load_smpl_model
    m = synth.smpl_model(V=50, seed=4)
    fake = types.ModuleType("chumpy")
    ch = types.ModuleType("chumpy.ch")

    class Ch:                      # pickles as chumpy.ch.Ch with a state dict, like the real files
        def __init__(self, x):
            self.x = x

        def __getstate__(self):
            return {"x": self.x}
    Ch.__module__, Ch.__qualname__ = "chumpy.ch", "Ch"
    ch.Ch = Ch
    fake.ch = ch
    sys.modules["chumpy"], sys.modules["chumpy.ch"] = fake, ch
    try:
        kt = np.stack([np.asarray(m["parents"]).astype(np.uint32), np.arange(24, dtype=np.uint32)])
        dd = dict(v_template=Ch(m["v_template"].astype(np.float64)), shapedirs=Ch(m["shapedirs"].astype(np.float64)),
                  posedirs=m["posedirs"].astype(np.float64), weights=Ch(m["weights"].astype(np.float64)),
                  J_regressor=sp.csc_matrix(m["J_regressor"].astype(np.float64)), kintree_table=kt,
                  f=np.arange(12, dtype=np.uint32).reshape(4, 3), bs_type="lrotmin")
        p = tmp_path / "SMPL_NEUTRAL.pkl"
        with open(p, "wb") as fh:
            pickle.dump(dd, fh, protocol=2)
    finally:
        del sys.modules["chumpy"], sys.modules["chumpy.ch"]
    got = load_smpl_model(str(p))
    for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "weights"):
        np.testing.assert_array_equal(got[k], m[k])
    assert got["parents"].tolist() == list(synth.SMPL_PARENTS) and got["f"].shape == (4, 3)
    assert got["model_betas"].shape == (10,) and not got["model_betas"].any()


@pytest.mark.gpu
def test_scorer_classes_match_reference_golden(gpu_device):
    g = golden("scores.npz")
    infos = json.loads(str(g["infos_json"]))
    pose = g["pose"][:500]
    jc = np.zeros((500, 24, 3), np.float32)
    res = reba.REBA()(pose, jc, infos["loaded"])
    want = g["reba_loaded"][:500]
    assert [int(r["score"]) for r in res] == want[:, 0].tolist()
    assert isinstance(res[0]["score"], np.int64)
    assert res[7]["log_score"] == [int(want[7, 1]), int(want[7, 2]), int(want[7, 3]), f"{want[7,4]},{want[7,5]}",
                                   f"{want[7,6]},{want[7,7]}", f"{want[7,8]},{want[7,9]}"]
    res = rula.RULA(True)(pose, jc, infos["example"])
    want = g["rula_example"][:500]
    assert [int(r["score"]) for r in res] == want[:, 0].tolist()
    assert res[3]["log_score"][4:] == want[3, 9:].tolist() and res[3]["log_score"][0] == f"{want[3,1]},{want[3,2]}"


@pytest.mark.gpu
def test_coord_utils_surface(gpu_device):
    g = golden("euler.npz")
    aa = coord_utils.rot_to_angle(g["rotmat"][3])
    np.testing.assert_allclose(aa, g["axis_angle"][3], atol=1e-6)
    assert aa.dtype == np.float32 and aa.shape == (24, 3)
    e = coord_utils.axis_angle_to_euler_angle(g["axis_angle"][3])
    assert e.dtype == np.float64
    d = np.abs(e - g["euler_deg"][3])
    assert np.minimum(d, 360 - d).max() < 1e-5
    jg = golden("joint_cam.npz")
    smpl = SMPL(models={"neutral": synth.smpl_model(V=6890, seed=2)}, device=gpu_device)
    assert smpl.joint_regressor.shape == (29, 6890) and smpl.vertex_num == 6890
    poses = jg["full_axis_angle_in"].copy()
    jc = coord_utils.get_joint_cam(poses, smpl)
    np.testing.assert_allclose(jc, jg["full_joint_cam"], atol=1e-2)
    np.testing.assert_array_equal(poses, jg["full_axis_angle_after"])      # caller sees the root overwrite (Q5)


@pytest.mark.gpu
def test_predictor_score_crops(gpu_device):
    import types
    sd = synth.hmr_state_dict(seed=1)
    model = hmr()
    model.load_state_dict(sd, strict=False)
    smpl = SMPL(models={"neutral": synth.smpl_model(V=6890, seed=2)}, device=gpu_device)
    args = types.SimpleNamespace(gpu="0", type="REBA,RULA", debug=False, debug_joints="", debug_frame=-1)
    pred = base.Predictor(args, spin_model=model, smpl_model=smpl)
    crops = synth.crops(12, seed=2)
    loader = [torch.from_numpy(crops[i:i + 8]) for i in range(0, 12, 8)]       # a ragged last batch
    out = pred.score_crops(loader, synth.EXAMPLE_INFO)
    assert out["result"].shape == (12, 24, 3) and out["result"].dtype == np.float64
    assert out["joint_cam"].shape == (12, 24, 3) and np.all(out["debug_result"][:, 0] == np.array([3.14, 0, 0], np.float32))
    final, scores, logs, (level, name) = out["reba"]
    assert len(final) == 5 and scores.shape == (12,) and logs.shape == (12, 6) and level in (1, 2, 3, 4, 5)
    from oracle import reba_ref
    np.testing.assert_array_equal(scores, reba_ref.reba_packed(out["result"], synth.EXAMPLE_INFO["REBA"])[:, 0])
    # one batch of 12 gives the same frames as 8 + 4 (frames independent)
    out2 = pred.score_crops([torch.from_numpy(crops)], synth.EXAMPLE_INFO)
    np.testing.assert_array_equal(out2["result"], out["result"])


@pytest.mark.gpu
def test_predictor_score_frames_end_to_end(gpu_device):
    """Frames + tracker dict -> crops -> pose -> scores on the GPU, against the CPU oracle chain."""
    import types
    from oracle import crop_ref, hmr_ref, pipeline_ref, smpl_ref
    sd = synth.hmr_state_dict(seed=1)
    sm = synth.smpl_model(V=6890, seed=2)
    model = hmr()
    model.load_state_dict(sd, strict=False)
    smpl = SMPL(models={"neutral": sm}, device=gpu_device)
    args = types.SimpleNamespace(gpu="0", type="REBA,RULA", debug=False, debug_joints="", debug_frame=-1)
    pred = base.Predictor(args, spin_model=model, smpl_model=smpl, batch_size=4)
    rng = np.random.default_rng(9)
    frames = rng.integers(0, 256, (9, 240, 320, 3), dtype=np.uint8)
    mk = lambda fr, w, h: {'bbox': np.stack([np.array([160 + 3 * i, 120 - 2 * i, w, h], np.float32) for i in range(len(fr))]),
                           'frames': np.array(fr)}
    tr = {4: mk([0, 1], 200, 200), 8: mk([2, 3, 4, 5, 6, 8], 90, 180)}     # id 4: 2 frames < 0.33*9
    out = pred.score_frames(frames, tr, synth.EXAMPLE_INFO)
    assert out['frames'].tolist() == [2, 3, 4, 5, 6, 8] and out['result'].shape == (6, 24, 3)
    crops = np.stack([crop_ref.crop_to_tensor(frames[f], b, 1.2) for f, b in zip(out['frames'], out['bboxes'])])
    om = smpl_ref.SMPLModel(sm["v_template"], sm["shapedirs"], sm["posedirs"], sm["J_regressor"], sm["weights"])
    want = pipeline_ref.run(hmr_ref.build(sd), om, crops, synth.EXAMPLE_INFO)
    d = np.abs(out['result'] - want['euler'])
    assert np.minimum(d, 360 - d).max() < 2e-2
    np.testing.assert_allclose(out['joint_cam'], want['joint_cam'], atol=0.10)        # millimetres = 1e-4 m


_SHARD_WORKER = r"""
import sys, types
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from poserisk_release_amd import dropin, synth
dropin.install()
from core import base
from models import hmr
from smpl import SMPL
rank = int(sys.argv[3])
dev = torch.device("cuda", 0)                      # rehearsal: both ranks share the one GPU of the box
model = hmr(); model.load_state_dict(synth.hmr_state_dict(seed=1), strict=False)
smpl = SMPL(models={"neutral": synth.smpl_model(V=6890, seed=2)}, device=dev)
args = types.SimpleNamespace(gpu="0", type="REBA,RULA", debug=False, debug_joints="", debug_frame=-1, lanes=2)
pred = base.Predictor(args, spin_model=model, smpl_model=smpl, batch_size=2)
rng = np.random.default_rng(9)
frames = rng.integers(0, 256, (9, 240, 320, 3), dtype=np.uint8)
tr = {8: {'bbox': np.stack([np.array([160 + 3 * i, 120 - 2 * i, 90, 180], np.float32) for i in range(7)]),
          'frames': np.array([1, 2, 3, 4, 5, 6, 8])}}
whole = pred.score_frames(frames, tr, synth.EXAMPLE_INFO)            # no process group yet: all 7 frames here
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + sys.argv[2], rank=rank, world_size=2)
part = pred.score_frames(frames, tr, synth.EXAMPLE_INFO)             # 3 + 4 frames, one gather
for k in ("result", "joint_cam", "debug_result"):
    assert part[k].shape == whole[k].shape and np.array_equal(part[k], whole[k]), k
for k in ("reba", "rula"):
    assert np.array_equal(np.array(part[k][0], float), np.array(whole[k][0], float), equal_nan=True), k   # top-10 % is NaN for N < 10
    assert np.array_equal(part[k][1], whole[k][1]), k
dist.destroy_process_group()
print("ok")
"""


@pytest.mark.gpu
def test_predictor_shards_frames_across_ranks(gpu_device, tmp_path):
    """SURVEY.md 8e through the plugin surface: two ranks each score a contiguous shard of the track and
    gather once; every rank ends with the same frames, bit for bit, as one process scoring them all."""
    import os, subprocess, sys
    from conftest import REPO
    script = tmp_path / "w.py"
    script.write_text(_SHARD_WORKER)
    port = str(31500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), REPO, port, str(r)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and "ok" in o, o[-3000:]


_JOIN_WORKER = r"""
import os, sys, types
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from poserisk_release_amd import dropin, synth
dropin.install()
from core import base
from models import hmr
from smpl import SMPL
assert not dist.is_initialized()
model = hmr(); model.load_state_dict(synth.hmr_state_dict(seed=1), strict=False)
smpl = SMPL(models={"neutral": synth.smpl_model(V=6890, seed=2)}, device=torch.device("cuda", 0))
# what main/run.py builds (run.py:11-19) + the world size the launcher started
args = types.SimpleNamespace(gpu="0", type="REBA,RULA", debug=False, debug_joints="", debug_frame=-1, world_size=2)
pred = base.Predictor(args, spin_model=model, smpl_model=smpl, batch_size=2)      # joins the process group itself
assert dist.is_initialized() and dist.get_world_size() == 2 and pred.world_size == 2 and pred.device == torch.device("cuda", 0)
want = np.load(sys.argv[2])
rng = np.random.default_rng(9)
frames = rng.integers(0, 256, (9, 240, 320, 3), dtype=np.uint8)
tr = {8: {'bbox': np.stack([np.array([160 + 3 * i, 120 - 2 * i, 90, 180], np.float32) for i in range(7)]),
          'frames': np.array([1, 2, 3, 4, 5, 6, 8])}}
part = pred.score_frames(frames, tr, synth.EXAMPLE_INFO)             # this rank's 3 or 4 frames, one gather
assert np.array_equal(part["result"], want["result"]) and np.array_equal(part["joint_cam"], want["joint_cam"])
assert np.array_equal(part["reba"][1], want["reba"]) and np.array_equal(part["rula"][1], want["rula"])
try:
    base.Predictor(types.SimpleNamespace(**dict(vars(args), world_size=4)), spin_model=model, smpl_model=smpl)
    raise SystemExit("a world size other than the group's was accepted")
except RuntimeError as e:
    assert "2 ranks" in str(e), e
dist.destroy_process_group()
sys.stdout.write("ok rank " + os.environ["RANK"] + "\n"); sys.stdout.flush()   # ONE write: two ranks share the launcher's pipe
"""


@pytest.mark.gpu
def test_predictor_joins_the_process_group_the_launcher_started(gpu_device, tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 2 main/run.py ...`: `Predictor(args)` finds WORLD_SIZE = 2, joins the
    process group on its own (one process per GPU; here gloo and both ranks on the box's one GPU --
    POSERISK_DIST_BACKEND / POSERISK_SHARE_GPU are the rehearsal's knobs, RCCL and cuda:LOCAL_RANK the defaults), shards the
    track, gathers once, and every rank has the single-process result bit for bit; a Predictor asking for another world size
    is refused."""
    import os, subprocess, sys
    from conftest import REPO
    frames, tr = _video()
    whole = _predictor(gpu_device).score_frames(frames, tr, synth.EXAMPLE_INFO)
    np.savez(tmp_path / "want.npz", result=whole["result"], joint_cam=whole["joint_cam"], reba=whole["reba"][1], rula=whole["rula"][1])
    script = tmp_path / "join.py"
    script.write_text(_JOIN_WORKER)
    env = dict(os.environ, POSERISK_DIST_BACKEND="gloo", POSERISK_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(32600 + os.getpid() % 1000), str(script), REPO, str(tmp_path / "want.npz")],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok rank 0" in r.stdout and "ok rank 1" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


_CALL_WORKER = r"""
import os, pickle, sys, types
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
rank = int(os.environ["RANK"])
work = sys.argv[3]
# stand-ins for the reference's front end (cv2 video decoding, multi_person_tracker): deterministic, and they leave a marker
# naming the rank that ran them
video = np.load(os.path.join(work, "video.npy"))
cv2 = types.ModuleType("cv2")
cv2.CAP_PROP_FPS, cv2.CAP_PROP_FRAME_WIDTH, cv2.CAP_PROP_FRAME_HEIGHT = 5, 3, 4
class VideoCapture:
    def __init__(self, path): self.i = 0; open(os.path.join(work, f"decoded_by_rank{rank}"), "w").close()
    def get(self, prop): return {5: 25.0, 3: float(video.shape[2]), 4: float(video.shape[1])}[prop]
    def isOpened(self): return True
    def read(self):
        if self.i >= len(video): return False, None
        self.i += 1
        return True, video[self.i - 1]
    def release(self): pass
cv2.VideoCapture = VideoCapture
cv2.resize = lambda img, size: img
cv2.imwrite = lambda path, img: np.save(path + ".npy", img)
cv2.imread = lambda path: np.load(path + ".npy")
sys.modules["cv2"] = cv2
mpt = types.ModuleType("multi_person_tracker")
with open(os.path.join(work, "clip", "tracking.pkl"), "rb") as f:
    TRACK = pickle.load(f)
class MPT:
    def __init__(self, **kw): pass
    def __call__(self, image_path):
        assert len(os.listdir(image_path)) == len(video)
        open(os.path.join(work, f"tracked_by_rank{rank}"), "w").close()
        return TRACK
mpt.MPT = MPT
sys.modules["multi_person_tracker"] = mpt
from poserisk_release_amd import dropin, synth
dropin.install()
from core import base
from models import hmr
from smpl import SMPL
model = hmr(); model.load_state_dict(synth.hmr_state_dict(seed=1), strict=False)
smpl = SMPL(models={"neutral": synth.smpl_model(V=6890, seed=2)}, device=torch.device("cuda", 0))
args = types.SimpleNamespace(gpu="0", type="REBA,RULA", debug=True, debug_joints="L_Hip,Neck", debug_frame=-1, world_size=2)
pred = base.Predictor(args, spin_model=model, smpl_model=smpl, batch_size=2)
want = np.load(sys.argv[2])
info = os.path.join(work, "info.json")
# (1) the front end found on disk, (2) the reference's own front end (stand-ins above): rank 0 runs it, everyone scores its shard
for inp, outdir in ((os.path.join(work, "clip"), "out_dir"), (os.path.join(work, "video.mp4"), "out_video")):
    out = pred(inp, info, os.path.join(work, outdir))
    assert out["frames"].tolist() == [1, 2, 3, 4, 5, 6, 8], out["frames"]
    assert np.array_equal(out["result"], want["result"]) and np.array_equal(out["joint_cam"], want["joint_cam"])
    assert np.array_equal(out["reba"][1], want["reba"]) and np.array_equal(out["rula"][1], want["rula"])
assert out["fps"] == 25.0
dist.barrier()
names = sorted(os.listdir(work))
assert "decoded_by_rank0" in names and "tracked_by_rank0" in names and "decoded_by_rank1" not in names and "tracked_by_rank1" not in names, names
assert not os.path.exists(os.path.join(work, "out_video", "tmp"))
# (3) ranks holding different tracks are told so by every rank, instead of hanging in the gather
frames = video[..., ::-1].copy()
other = {k: dict(v) for k, v in TRACK.items()}
if rank == 1:
    other[8] = dict(bbox=TRACK[8]["bbox"][:-1], frames=TRACK[8]["frames"][:-1])
try:
    pred.score_frames(frames, other, synth.EXAMPLE_INFO)
    raise SystemExit("ranks with different tracks were not refused")
except RuntimeError as e:
    assert "same target track" in str(e), e
dist.barrier()
dist.destroy_process_group()
sys.stdout.write(f"ok rank {rank}\n"); sys.stdout.flush()       # ONE write: two ranks share the launcher's pipe
"""


@pytest.mark.gpu
def test_predictor_call_under_two_ranks_runs_the_front_end_once(gpu_device, tmp_path):
    """The advisor's round-5 finding: under `torch.distributed.run` every rank ran load_front_end (rmtree of <output>/tmp,
    JPEGs, the tracker) and wrote the same report files.  Now rank 0 runs the front end and broadcasts frames + tracking,
    every rank checks that all hold the same track before sharding, and rank 0 alone writes the reports: two ranks under the
    launcher (gloo, both on the box's one GPU), `predictor(input, info, output)` on a frames.npy directory and through
    stand-ins of cv2 + multi_person_tracker, every rank bit-identical to one process; ranks with different tracks are refused
    on every rank."""
    import os, pickle, subprocess, sys
    from conftest import REPO
    frames, tr = _video()
    whole = _predictor(gpu_device).score_frames(frames, tr, synth.EXAMPLE_INFO)
    np.savez(tmp_path / "want.npz", result=whole["result"], joint_cam=whole["joint_cam"], reba=whole["reba"][1], rula=whole["rula"][1])
    (tmp_path / "clip").mkdir()
    np.save(tmp_path / "clip" / "frames.npy", frames)
    with open(tmp_path / "clip" / "tracking.pkl", "wb") as f:
        pickle.dump(tr, f)
    np.save(tmp_path / "video.npy", frames[..., ::-1].copy())          # what cv2 would decode: BGR
    (tmp_path / "video.mp4").write_bytes(b"stand-in")
    (tmp_path / "info.json").write_text(json.dumps(synth.EXAMPLE_INFO))
    script = tmp_path / "call.py"
    script.write_text(_CALL_WORKER)
    env = dict(os.environ, POSERISK_DIST_BACKEND="gloo", POSERISK_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(33600 + os.getpid() % 1000), str(script), REPO, str(tmp_path / "want.npz"),
                        str(tmp_path)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok rank 0" in r.stdout and "ok rank 1" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    for d in ("out_dir", "out_video"):
        txt = (tmp_path / d / "reba_result.txt").read_text()
        assert txt.startswith(f"AVG Score: {whole['reba'][0][0]} ") and (tmp_path / d / "rula_result.txt").is_file()
        assert (tmp_path / d / "debug" / "REBA_score_log.csv").is_file()


def _predictor(gpu_device, **kw):
    import types
    model = hmr()
    model.load_state_dict(synth.hmr_state_dict(seed=1), strict=False)
    smpl = SMPL(models={"neutral": synth.smpl_model(V=6890, seed=2)}, device=gpu_device)
    args = types.SimpleNamespace(gpu="0", type="REBA,RULA", debug=kw.pop("debug", False),
                                 debug_joints=kw.pop("debug_joints", ""), debug_frame=-1)
    return base.Predictor(args, spin_model=model, smpl_model=smpl, batch_size=4)


def _video(n=9):
    rng = np.random.default_rng(9)
    frames = rng.integers(0, 256, (n, 240, 320, 3), dtype=np.uint8)
    fr = [1, 2, 3, 4, 5, 6, 8]
    tr = {8: {'bbox': np.stack([np.array([160 + 3 * i, 120 - 2 * i, 90, 180], np.float32) for i in range(len(fr))]),
              'frames': np.array(fr)}}
    return frames, tr


@pytest.mark.gpu
def test_predictor_call_writes_the_reference_reports(gpu_device, tmp_path):
    """run.py:31 `predictor(input, info, output)` with the front end found on disk (frames.npy + tracking.pkl)."""
    import pickle
    frames, tr = _video()
    src = tmp_path / "clip"
    src.mkdir()
    np.save(src / "frames.npy", frames)
    with open(src / "tracking.pkl", "wb") as f:
        pickle.dump(tr, f)
    info = tmp_path / "info.json"
    info.write_text(json.dumps(synth.EXAMPLE_INFO))
    pred = _predictor(gpu_device, debug=True, debug_joints="L_Hip,Neck")
    out = pred(str(src), str(info), str(tmp_path / "out"))
    assert out["frames"].tolist() == [1, 2, 3, 4, 5, 6, 8] and out["fps"] == 30.0
    direct = pred.score_frames(frames, tr, synth.EXAMPLE_INFO)
    np.testing.assert_array_equal(out["result"], direct["result"])
    txt = (tmp_path / "out" / "reba_result.txt").read_text()
    final, _, _, (level, name) = out["reba"]
    assert txt.startswith(f"AVG Score: {final[0]} \n%50 Score: {final[1]} ") and txt.endswith(f"Action: {name} ")
    assert f"Action level: {level} " in txt and (tmp_path / "out" / "rula_result.txt").is_file()
    rows = (tmp_path / "out" / "debug" / "REBA_score_log.csv").read_text().strip().splitlines()
    assert len(rows) == 1 + 9 and rows[0].startswith("Frame,Final_score,Joint Score,Trunk,Neck")
    assert rows[1] == "0" and rows[2].startswith(f"1,{out['reba'][1][0]},")      # frame 0 is not in the track
    for name in ("REBA_eval_pose_log.csv", "RULA_score_log.csv", "RULA_eval_pose_log.csv", "pose_log.csv"):
        assert (tmp_path / "out" / "debug" / name).is_file()
    # no info file -> main/default_information.json
    out2 = pred(str(src), str(tmp_path / "missing.json"), str(tmp_path / "out2"))
    np.testing.assert_array_equal(out2["result"], out["result"])


@pytest.mark.gpu
def test_predictor_debug_frame_branch(gpu_device, tmp_path):
    """run.py --debug --debug_frame 4 (base.py:128-135, 273-282): that frame's mesh as OBJ (mm) and the 3-D skeleton plot,
    no score reports."""
    import pickle, types
    frames, tr = _video()
    src = tmp_path / "clip"
    src.mkdir()
    np.save(src / "frames.npy", frames)
    with open(src / "tracking.pkl", "wb") as f:
        pickle.dump(tr, f)
    model = hmr()
    model.load_state_dict(synth.hmr_state_dict(seed=1), strict=False)
    sm = synth.smpl_model(V=6890, seed=2)
    sm["f"] = np.arange(30).reshape(10, 3)
    smpl = SMPL(models={"neutral": sm}, device=gpu_device)
    args = types.SimpleNamespace(gpu="0", type="REBA,RULA", debug=True, debug_joints="", debug_frame=4)
    pred = base.Predictor(args, spin_model=model, smpl_model=smpl, batch_size=4)
    out = pred(str(src), "", str(tmp_path / "out"))
    dbg = tmp_path / "out" / "debug"
    assert (dbg / "joint_3d.png").is_file() and not (tmp_path / "out" / "reba_result.txt").exists()
    lines = (dbg / "smpl_model.obj").read_text().splitlines()
    assert sum(l.startswith("v ") for l in lines) == 6890 and sum(l.startswith("f ") for l in lines) == 10
    idx = out["frames"].tolist().index(4)
    verts, _ = smpl.layer["neutral"](torch.from_numpy(out["debug_result"][idx]).reshape(1, 72), torch.zeros(1, 10))
    v0 = np.array(lines[0].split()[1:], np.float64)
    np.testing.assert_allclose(v0, verts[0, 0].cpu().numpy().astype(np.float32) * 1000, rtol=1e-6)


@pytest.mark.gpu
def test_predictor_call_with_the_reference_front_end_modules(gpu_device, tmp_path, monkeypatch):
    """Without a prepared directory `__call__` drives cv2 + multi_person_tracker exactly as base.py:47-74 does
    (decode, resize to width 800, JPEGs for the tracker, read back as BGR); both are stand-ins here."""
    import sys, types
    frames, tr = _video()
    big = np.repeat(np.repeat(frames, 4, axis=1), 4, axis=2)          # 960 x 1280 source video
    store, calls = {}, {}

    class Cap:
        def __init__(self, path): self.i = 0; calls["video"] = path
        def get(self, prop): return {5: 25.0, 3: 1280.0, 4: 960.0}[prop]
        def isOpened(self): return True
        def read(self):
            self.i += 1
            return (self.i <= len(big)), (big[self.i - 1][..., ::-1] if self.i <= len(big) else None)
        def release(self): pass

    cv2 = types.ModuleType("cv2")
    cv2.CAP_PROP_FPS, cv2.CAP_PROP_FRAME_WIDTH, cv2.CAP_PROP_FRAME_HEIGHT = 5, 3, 4
    cv2.VideoCapture = Cap
    cv2.resize = lambda img, wh: np.ascontiguousarray(img[::img.shape[0] // wh[1], ::img.shape[1] // wh[0]][:wh[1], :wh[0]])
    cv2.imwrite = lambda path, img: store.__setitem__(path, img.copy()) or True
    cv2.imread = lambda path: store[path]
    mpt = types.ModuleType("multi_person_tracker")

    class MPT:
        def __init__(self, **kw): calls["mpt"] = kw
        def __call__(self, folder):
            calls["folder"] = folder
            s = 800 / 1280
            return {k: {'bbox': v['bbox'] * 4 * s, 'frames': v['frames']} for k, v in tr.items()}
    mpt.MPT = MPT
    monkeypatch.setitem(sys.modules, "cv2", cv2)
    monkeypatch.setitem(sys.modules, "multi_person_tracker", mpt)
    pred = _predictor(gpu_device)
    out = pred("clip.mp4", "", str(tmp_path / "o"))
    assert calls["video"] == "clip.mp4" and calls["folder"].endswith("tmp") and calls["mpt"]["detector_type"] == "yolo"
    assert out["fps"] == 25.0 and len(store) == 9 and next(iter(store.values())).shape == (600, 800, 3)
    assert out["result"].shape == (7, 24, 3) and (tmp_path / "o" / "reba_result.txt").is_file()
    # the same frames handed over directly (BGR) give the same angles
    direct = pred.score_frames(np.stack([store[k] for k in sorted(store)]), out_tr(calls, tr), synth.DEFAULT_INFO, bgr=True)
    np.testing.assert_array_equal(direct["result"], out["result"])


def out_tr(calls, tr):
    s = 800 / 1280
    return {k: {'bbox': v['bbox'] * 4 * s, 'frames': v['frames']} for k, v in tr.items()}
