"""RCCL next to the captured step on a 1-GPU box: backend "nccl" with ONE rank, and the N > 1 timed region of bench.py --
graph A (forward + backward, writes FlatAdam's gradient buffer) | `FlatGradSync.sync(force_collective=True)`: all_reduce(AVG) on
the non-default launch stream | graph B (optimizer) -- replayed over the pool.  A mean over one rank changes no bit, so the
state after such a step must equal the eager step's (bench.collective_check).  This is the sequence every rank of the driver's
2/4/8-GPU runs executes (dist.py; train_resnet_ae_ddp.py:125-172 is the reference's only distributed code)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_force_collective_is_a_no_op_without_a_process_group():
    import torch
    from batch3dmot_amd.dist import FlatGradSync
    p = torch.nn.Parameter(torch.ones(3))
    p.grad = torch.full((3,), 2.0)
    FlatGradSync([p]).sync(force=True, force_collective=True)     # no process group: nothing to call
    assert torch.equal(p.grad, torch.full((3,), 2.0))


@pytest.mark.gpu
def test_rccl_all_reduce_between_graph_replays_one_rank():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-collective", "--steps", "6", "--warmup", "2",
                        "--ramp-ms", "0", "--no-cpu-baseline", "--no-secondary"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["value"] > 0
    assert d["timed_region"].startswith("hipGraph replay") and "all-reduce" in d["timed_region"], d["timed_region"]
    chk = d["replay_vs_eager_loss"]["collective_vs_eager_state"]
    assert chk["equal"] and chk["tensors"] > 100, chk
