"""-m gpu: the data-parallel training step with two real processes.  On a box with at least two GPUs the
ranks take one device each and the collectives run on RCCL ("nccl": side-stream all-reduce / all_to_all ordered by events
against the backward walk -- the configuration of BASELINE configs[3]); on the usual one-GPU box both ranks share cuda:0 and the
collective runs on gloo (CUDA tensors staged through the host).  The step itself -- native backward into the flat gradient
buffer, the gradient exchange, fused clip + AdamW + EMA -- and every assertion are the same in both cases."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _free_parent_cache():
    """every test of this module starts GPU child processes (tests/gpu_util.py release_cached_gpu_memory)"""
    from tests.gpu_util import release_cached_gpu_memory
    release_cached_gpu_memory()
    yield

TWO_GPUS = torch.cuda.device_count() >= 2          # does not initialise the GPU
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


def test_bench_multi_rank_flow_on_one_device():
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, barrier + max-over-ranks timing, rank 0 prints
    ONE JSON line), here with two ranks sharing the box's single GPU over gloo (DFH_DIST_BACKEND): the value is meaningless,
    the flow is what is checked."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", DFH_DIST_BACKEND="nccl" if TWO_GPUS else "gloo",
               PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    print(out.stdout[-2000:], out.stderr[-2000:])
    assert out.returncode == 0
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1 and rec["value"] > 0 and rec["scaling"] == "weak"
    assert rec["roofline"]["bound"] == "mfma" and rec["cpu_baseline"] is None


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher environment: the parent starts two fresh ranks itself (bench.launch_ranks), relays
    rank 0's one JSON line and exits 0.  Two GPUs: one device per rank over RCCL; one GPU: both ranks on it over gloo."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DFH_DIST_BACKEND="nccl" if TWO_GPUS else "gloo", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-profile"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    print(out.stdout[-2000:], out.stderr[-2000:])
    assert out.returncode == 0
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["value"] > 0
    assert rec["collective_backend"] == ("nccl" if TWO_GPUS else "gloo")
    assert rec["rccl_ranks"] == (2 if TWO_GPUS else None)


def test_bench_train_mode_launches_its_own_ranks():
    """`python bench.py --mode train --gpus 2` with NO launcher environment (the training twin of the test above): two fresh ranks, the
    gradient exchange inside the backward (fp32 wire: the default), one JSON line whose multi-rank fields let a slow rank or an exposed
    exchange be read off the one record -- per-rank ms_per_step min / max and comm_exposed_ms (max) / comm_exposed_ms_min."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DFH_DIST_BACKEND="nccl" if TWO_GPUS else "gloo", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "train", "--gpus", "2", "--steps", "1", "--warmup", "1", "--outfits", "1",
           "--no-cpu-baseline", "--no-profile"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    print(out.stdout[-2000:], out.stderr[-2000:])
    assert out.returncode == 0
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 1 and rec["value"] > 0 and rec["unit"] == "items/s" and rec["grad_wire"] == "fp32"
    assert rec["comm_exposed_ms"] is not None and rec["comm_exposed_ms_min"] is not None and rec["comm_exposed_ms"] >= rec["comm_exposed_ms_min"] >= 0
    assert 0 < rec["rank_ms_per_step"]["min"] <= rec["rank_ms_per_step"]["max"] <= rec["ms_per_step"] * 1.05
    assert rec["collective_backend"] == ("nccl" if TWO_GPUS else "gloo")


def test_rccl_branch_runs_in_a_world_of_one():
    """The "nccl" (RCCL) branch of the data-parallel step on the box's one GPU: a process group of ONE rank with
    DFH_DIST_SINGLE_RANK=1 (difashion_amd.dist.active) sends every collective of the step through RCCL on its side stream --
    tests/rccl_single_rank_worker.py lists what must hold (averaging over one rank is the identity)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               DFH_DIST_SINGLE_RANK="1", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_single_rank_worker.py")], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    print(out.stdout[-3000:], out.stderr[-3000:])
    assert out.returncode == 0 and "RCCL single-rank run OK" in out.stdout
