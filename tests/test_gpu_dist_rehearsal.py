"""The N > 1 code path of `bench.py --gpus N` on the one GPU a test box has: two ranks, both on cuda:0, gloo collectives on
the device tensors (`CR_REHEARSE_ONE_GPU=1`).  Functional only: the two-segment dense-region backward, the three all-reduce
phases of TrainStep (RoI-head FC weights | RPN head + FPN + DLA level5 | rest), the parameter broadcast and the loss-vector
all-reduce run with real inter-process exchanges, and both ranks must hold identical parameters afterwards."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_rank_rehearsal_keeps_parameters_in_sync():
    env = dict(os.environ, CR_REHEARSE_ONE_GPU="1", CR_BENCH_BF16="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29617", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    comm = d["config"]["comm"]
    assert d["n_gpus"] == 2 and comm["world_size"] == 2 and comm["backend"].startswith("gloo")
    assert comm["two_segment_backward"] is True
    assert comm["params_in_sync_after_run"] is True
    assert d["config"]["valid"] and d["config"]["skipped_steps"] == 0
    assert d["config"]["final_loss"] == d["config"]["final_loss"]            # finite
