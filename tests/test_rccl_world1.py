"""The RCCL branch of the path's one exchange, EXECUTED: one rank, world size 1, on the box's one MI355X.

SURVEY.md 8e: per-frame SMPL-parameter records are all-gathered once per batch (they feed the all-frames aggregation of
lib/core/base.py:263-271).  Until round 5 `pipeline.init_distributed("nccl")` + `RecordExchange.step` ->
`all_gather_into_tensor` had run against a recording stub and on gloo only.  Here they run against RCCL itself, in a fresh
child process (the process group is initialised before the child touches the GPU), with three lanes, eager and hipGraph
replay.  What a world of one cannot show is rendezvous between processes and xGMI transport -- nothing else differs."""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO, measured

CHILD = os.path.join(REPO, "tests", "rccl_world1_child.py")
ENV_DROP = ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")


def _child(steps, lanes, graph, port):
    env = {k: v for k, v in os.environ.items() if k not in ENV_DROP}
    r = subprocess.run([sys.executable, CHILD, str(steps), str(lanes), str(int(graph)), str(port)], cwd=REPO, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.parametrize("graph", [False, True])
def test_record_exchange_runs_on_rccl_with_one_rank(gpu_device, graph):
    port = 29300 + os.getpid() % 200 + (50 if graph else 0)
    rec = _child(20, 3, graph, port)
    assert rec["backend"] == "nccl" and rec["world_size"] == 1
    assert rec["gathered_equals_pack_record"] is True           # every step's collective delivered that step's records
    assert rec["records_differ_between_steps"] is True          # (the check could not pass on stale data)
    assert rec["release_after_set_every_step"] is True          # the lane waits for the comm stream's read ...
    assert rec["lanes_with_unconsumed_release"] == 3            # ... and every earlier wait was consumed by a forward
    assert rec["comm_ms_per_step"] > 0
    measured(f"rccl_world1_comm_ms_per_step{'_graph' if graph else ''}", rec["comm_ms_per_step"], unit="ms")
    print("[rccl]", json.dumps(rec))


@pytest.mark.gpu
def test_bench_force_exchange_runs_the_rccl_branch(gpu_device):
    """`bench.py --force-exchange`: the N>1 code of the benchmark (init_distributed('nccl'), RecordExchange per step, fence
    with a barrier, the second-route gather check, the `dist` object) on one rank."""
    env = {k: v for k, v in os.environ.items() if k not in ENV_DROP}
    env["MASTER_PORT"] = str(29550 + os.getpid() % 100)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--force-exchange", "--steps", "5", "--warmup", "2",
                        "--cpu-frames", "0", "--no-roofline", "--no-other-configs", "--repeats", "2"],
                       cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert line["n_gpus"] == 1 and line["config"]["dist_backend"] == "nccl" and line["config"]["dist_world_size"] == 1
    assert line["config"]["exchange"].startswith("all-gather")
    assert line["gather_verified"] is True
    assert line["comm_ms_per_step"] is not None and line["comm_ms_per_step"] > 0
    assert line["dist"]["backend"] == "nccl" and line["dist"]["distinct_devices"] == 1
    measured("bench_force_exchange_comm_ms_per_step", line["comm_ms_per_step"], unit="ms")
