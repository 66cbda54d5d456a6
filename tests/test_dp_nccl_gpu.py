"""Two ranks on two GPUs over RCCL (backend 'nccl'), through the benchmark's own launcher: the weak- and
strong-scaling lines of the full forward and the training step's gradient all-reduce (DistributedDataParallel, the
exchange step of /root/reference/tools/train.py:162-172).  Skipped on a one-GPU box (the driver's scaling run covers
N > 1 there); the rendezvous / sharding / max-over-ranks logic itself is covered on the CPU by tests/test_dp_gloo.py."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*argv, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=env, capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.fixture(scope="module")
def two_gpus():
    if torch.cuda.device_count() < 2:       # (device_count does not initialise the GPU in this process)
        pytest.skip("needs two GPUs")


def test_two_rank_forward_weak_and_strong(two_gpus):
    common = ("--gpus", "2", "--steps", "3", "--warmup", "1", "--points", "4096", "--no-extras", "--no-cpu-baseline")
    weak = _bench(*common, "--batch", "4")
    assert weak["n_gpus"] == 2 and weak["scaling"] == "weak" and weak["config"]["global_batch"] == 8
    assert weak["config"]["parallelism"] == "dp2" and weak["value"] > 0
    strong = _bench(*common, "--global-batch", "4")
    assert strong["n_gpus"] == 2 and strong["scaling"] == "strong" and strong["config"]["global_batch"] == 4
    assert abs(strong["value"] - 4 * 1e3 / strong["ms_per_step"]) <= 1e-2 * strong["value"]   # whole-job frames / step time


def test_two_rank_train_step_all_reduces_gradients(two_gpus):
    line = _bench("--gpus", "2", "--train", "--steps", "2", "--warmup", "1", "--batch", "2", "--points", "4096")
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2" and line["config"]["global_batch"] == 4
    assert line["final_loss"] == line["final_loss"] and line["value"] > 0     # finite loss after DDP steps
