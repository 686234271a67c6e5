"""bench.py's N>1 path: one rank per GPU under torch.distributed.run, the per-step all-gather of the
916-byte SMPL-parameter records on a side stream.  Rehearsed with two ranks sharing the box's one GPU over gloo,
exactly as the driver launches it otherwise; the launcher starts before any process touches the GPU."""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO


def _launch(extra, port, timeout=900, ranks=2):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(REPO, "bench.py"),
           "--gpus", str(ranks), "--steps", "2", "--warmup", "1", "--cpu-frames", "0", "--no-roofline"] + extra
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    return subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.gpu
def test_bench_two_ranks_gather_the_records(gpu_device):
    port = 29700 + os.getpid() % 200
    # no --check-gather on the command line: with more than one rank the check is on by default
    r = _launch(["--backend", "gloo", "--share-gpu", "--batch", "64", "--lanes", "3"], port)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # rank 0 prints ONE line
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1
    cfg = line["config"]
    assert cfg["global_batch"] == 128 and cfg["frames_per_gpu_per_step"] == 64
    assert cfg["exchange"] == "all-gather of 916-B per-frame SMPL params per step"
    assert cfg["dist_backend"] == "gloo" and cfg["dist_world_size"] == 2
    assert line["gather_verified"] is True
    assert line["scaling"] == "weak" and line["value"] > 0 and line["unit"] == "frames/s"
    assert abs(line["value"] - 2 * 2 * 64 / (line["ms_per_step"] * 2 * 1e-3)) / line["value"] < 1e-3
    # the record diagnoses itself: every rank's own step time, the comm stream's time per step, what the backend saw
    pr = line["per_rank_ms_per_step"]
    assert len(pr["all"]) == 2 and pr["min"] <= pr["max"] and abs(pr["max"] - line["ms_per_step"]) < 1e-3
    assert line["comm_ms_per_step"] is not None and line["comm_ms_per_step"] > 0
    d = line["dist"]
    assert d["backend"] == "gloo" and d["world_size"] == 2 and [r_["rank"] for r_ in d["ranks"]] == [0, 1]
    assert all(r_["device"] == "cuda:0" for r_ in d["ranks"]) and d["distinct_devices"] == 1      # --share-gpu rehearsal
    assert "rccl_version (torch.cuda.nccl.version)" in d
    sp = line["value_spread"]
    assert sp["regions"] == 5 and sp["min"] <= sp["median"] <= sp["max"] and sp["min"] <= line["value"] <= sp["max"]


@pytest.mark.gpu
def test_bench_config3_slices_four_ranks_share_the_gpu(gpu_device):
    """configs[3]'s per-rank workload (--batch 256 --lanes 2, fp32) under the driver's launcher with as many ranks as the
    one-GPU box admits beside the test runner -- FOUR (its process guard allows six processes on the card; the full eight
    ranks' exchange runs on CPU in tests/test_host_cpu.py::test_record_exchange_at_config3_shape_eight_gloo_ranks): 1 024
    frames per step, the gather verified over all 1 024 rows on every rank, four step times, and ranks 1-3 leave with
    status 0 while rank 0 finishes its line.  What stays unexercised is rendezvous over RCCL and the xGMI transport."""
    port = 29950 + os.getpid() % 200
    r = _launch(["--backend", "gloo", "--share-gpu", "--batch", "256", "--lanes", "2"], port, timeout=1200, ranks=4)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    cfg = line["config"]
    assert line["n_gpus"] == 4 and cfg["global_batch"] == 1024 and cfg["frames_per_gpu_per_step"] == 256
    assert cfg["batches_in_flight"] == 2 and cfg["dist_world_size"] == 4
    assert line["gather_verified"] is True and line["scaling"] == "weak"
    pr = line["per_rank_ms_per_step"]
    assert len(pr["all"]) == 4 and all(t > 0 for t in pr["all"])
    assert abs(line["value"] - 4 * 2 * 256 / (line["ms_per_step"] * 2 * 1e-3)) / line["value"] < 1e-3
    assert [r_["rank"] for r_ in line["dist"]["ranks"]] == [0, 1, 2, 3] and line["dist"]["distinct_devices"] == 1


@pytest.mark.gpu
def test_bench_refuses_a_world_size_mismatch(gpu_device):
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       cwd=REPO, capture_output=True, text=True, timeout=300,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert r.returncode != 0 and "torch.distributed.run" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_bench_single_gpu_line_carries_the_contract(gpu_device):
    """The default (N = 1) run: one JSON line with the driver's fields, the `roofline` of the dominant kernel and the
    `cpu_baseline` of a bounded oracle sample (here 16 frames so that the test stays short)."""
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "3", "--warmup", "1", "--cpu-frames", "16"],
                       cwd=REPO, capture_output=True, text=True, timeout=900,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["warmup"] == 1 and line["higher_is_better"] is True
    assert line["unit"] == "frames/s" and line["dtype"] == "f32" and line["vs_baseline"] is None
    assert line["config"]["workload"].startswith("configs[1]") and "model" not in line["config"]
    assert abs(line["value"] - 3 * 64 / (line["ms_per_step"] * 3 * 1e-3)) / line["value"] < 1e-3
    roof = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    assert roof["bound"] == "mfma" and roof["unit"] == "TFLOP/s" and roof["peak"] == 157.3
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3 and 0 < roof["frac"] < 1
    assert roof["mfma_executed_tflops"] < roof["achieved"]          # ten layers run with a quarter of the multiplies
    # the roofline pass runs one batch in flight: its conv time belongs beside the one-batch step time
    assert roof["batches_in_flight"] == 1 and roof["conv_ms_per_step"] < roof["ms_per_step_same_mode"]
    assert line["config"]["batches_in_flight"] == 3        # (three steps are too few for the overlap to show in ms_per_step)
    sp = line["value_spread"]
    assert sp["regions"] == 5 and sp["min"] <= sp["median"] <= sp["max"] and sp["min"] <= line["value"] <= sp["max"]
    lo = line["lanes_overlap"]              # events on the lanes' own streams: a batch lasts longer than a step
    assert lo["batches"] == 3 and lo["batch_ms_mean"] > 0 and 0 < lo["some_batch_running_frac"] <= 1
    cpu = line["cpu_baseline"]
    assert "frame by frame" in cpu["sample"].lower()          # the scorers in the reference's per-frame arrangement
    oc = line["other_configs"]              # configs[2] and configs[3]'s per-GPU slice, measured in the same process
    assert [c["workload"].split(":")[0] for c in oc] == ["configs[2]", "configs[3]'s per-GPU slice"]
    assert oc[0]["dtype"].startswith("bf16") and oc[1]["dtype"] == "f32" and all(c["frames_per_step"] == 256 for c in oc)
    for c in oc:
        assert c["steps"] == 3 and c["value"] > 0 and abs(c["value"] - 256 / (c["ms_per_step"] * 1e-3)) / c["value"] < 1e-3
        assert 0 < c["roofline"]["frac"] < 1 and c["roofline"]["conv_ms_per_step"] > 0
    assert oc[0]["roofline"]["peak"] == 2500.0 and oc[1]["roofline"]["peak"] == 157.3
    assert line["library"]["build"] == "gfx950 release" and line["library"]["path"].endswith("libposerisk_hip.so")
    assert cpu["kind"] == "port" and cpu["unit"] == "frames/s" and cpu["cores"] >= 1 and cpu["value"] > 0 and "16 frames" in cpu["sample"]
