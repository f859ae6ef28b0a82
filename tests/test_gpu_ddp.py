"""-m gpu: the data-parallel training step with two real processes.  The GPU box has ONE device, so the
two ranks share cuda:0 and the collective runs on gloo (CUDA tensors staged through the host); the step itself -- native
backward into the flat gradient buffer, one all-reduce, fused clip + AdamW + EMA -- is the code that runs under RCCL."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_training_step_matches_the_whole_batch():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "ddp_gpu_worker.py")]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    print(out.stdout[-3000:], out.stderr[-3000:])
    assert out.returncode == 0 and "DDP_GPU_OK" in out.stdout
