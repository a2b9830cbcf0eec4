"""bench.py's N > 1 control flow (process group from the torchrun environment, sharded pool, gradient all-reduce inside
the step, barrier / ramp / timed loop, MAX / SUM reductions over ranks, ONE JSON line from rank 0): on the CPU with a
stub workload under gloo (world size 2), and -- on the GPU box -- the real camera+LiDAR+radar workload as two ranks
sharing the one device."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))



def _free_port():
    """A TCP port nobody holds right now (bound to port 0, read back, released): two suites on one machine cannot collide the
    way a pid-derived port can."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]

def _run(extra, nproc=2, timeout=600):
    port = _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc)] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                     # rank 0 only
    return json.loads(lines[0])


def test_two_rank_control_flow_on_cpu():
    d = _run(["--stub-cpu", "--steps", "5", "--warmup", "2"])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["scaling"] == "weak"
    # steps 2..6 of the 4-batch pool (100, 110, 120, 130 rows): both ranks' rows are summed
    assert d["edges_summed_over_ranks"] == 2 * (120 + 130 + 100 + 110 + 120)
    assert d["untimed_clock_ramp_steps"] == 8 and d["value"] > 0


def test_gpus_n_without_a_launcher_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (the command shape of the driver's N = 1 run): the
    process starts the two ranks itself as children and relays rank 0's single JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub-cpu", "--steps", "5", "--warmup", "2"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["edges_summed_over_ranks"] == 2 * (120 + 130 + 100 + 110 + 120)


def test_single_process_stub():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub-cpu", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1


@pytest.mark.gpu
def test_two_ranks_share_the_device_over_gloo():
    """The real workload with the flat all-reduce between backward and Adam, two ranks on device 0: eager steps, and the
    N > 1 timed region proper (forward + backward graph | eager all-reduce | optimizer graph)."""
    d = _run(["--backend", "gloo", "--all-ranks-on-device-0", "--steps", "3", "--warmup", "1", "--ramp-ms", "0", "--no-cpu-baseline", "--no-graph"],
             timeout=900)
    assert d["n_gpus"] == 2 and d["value"] > 0 and "all-reduce" in d["config"]["workload"]
    assert d["timed_region"].startswith("eager")
    g = _run(["--backend", "gloo", "--all-ranks-on-device-0", "--steps", "4", "--warmup", "1", "--ramp-ms", "0", "--no-cpu-baseline"],
             timeout=900)
    assert g["n_gpus"] == 2 and g["value"] > 0
    assert g["timed_region"].startswith("hipGraph replay") and "all-reduce" in g["timed_region"], g["timed_region"]
    # ... and with the encoders of the next batch enqueued under the step (train_step.EncodeAhead inside graph A)
    a = _run(["--backend", "gloo", "--all-ranks-on-device-0", "--steps", "4", "--warmup", "1", "--ramp-ms", "0", "--no-cpu-baseline",
              "--encode-ahead"], timeout=900)
    assert a["n_gpus"] == 2 and a["value"] > 0 and a["timed_region"].startswith("hipGraph replay")
    assert "EncodeAhead" in a["config"]["workload"]


@pytest.mark.gpu
def test_scene_inference_shards_scenes_over_two_ranks():
    """bench.py --mode infer --scene at N = 2 (two ranks on device 0 over gloo): every rank scores its own scene, the line carries
    the sum of the ranks' edges over the slowest rank's time, no collective on the data path."""
    d = _run(["--backend", "gloo", "--all-ranks-on-device-0", "--mode", "infer", "--scene", "--steps", "20", "--no-cpu-baseline"], timeout=900)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["value_uncached"] > 0
    assert "scenes sharded over 2 ranks" in d["config"]["parallelism"]


def test_perf_guard_flags_a_slower_secondary(tmp_path):
    """tools/perf_guard.py: +7 % on a secondary workload (round 2's PoseGNN loss) fails, +1 % passes."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = {"ms_per_step": 5.0, "config": {"workload": "w"}, "secondary": {"pose_gnn": {"ms_per_step": 0.92}, "x": {"ms_per_step": 3.8}}}
    (tmp_path / "base.json").write_text("log line\n" + json.dumps(base) + "\n")
    for pose, rc in ((0.988, 1), (0.93, 0)):
        new = json.loads(json.dumps(base))
        new["secondary"]["pose_gnn"]["ms_per_step"] = pose
        (tmp_path / "new.json").write_text(json.dumps(new))
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "perf_guard.py"), str(tmp_path / "new.json"),
                            "--baseline", str(tmp_path / "base.json")], capture_output=True, text=True)
        assert r.returncode == rc, r.stdout + r.stderr
        assert ("REGRESSION" in r.stdout) == bool(rc)
