"""Two data-parallel ranks on the one GPU of the test box (collectives over gloo): the reducer, the weight-gradient side
lane and the bucket hooks are the code the 8-GPU RCCL run uses; only the transport differs."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_two_ranks_average_gradients_and_stay_identical():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(HERE, "_ddp_gpu_worker.py"), "resnet18", "vit_tiny_test", "convnext_test"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    for arch in ("resnet18", "vit_tiny_test", "convnext_test"):
        assert f"ddp-ok {arch}" in out.stdout, out.stdout[-2000:]


def test_bench_gpus_2_launches_its_own_ranks():
    """`python bench.py --gpus 2` exactly as the driver starts N = 1 -- no launcher around it, no WORLD_SIZE: bench.py starts
    its two ranks itself (fresh children, the parent never touches the GPU) and rank 0 prints the one JSON line.  Rehearsal on
    the one GPU of the test box: ICAMD_DIST_BACKEND=gloo lets both ranks share it (on an N-GPU node the same command runs one
    rank per GPU over RCCL).  Reference entry point: `torchrun --nproc_per_node=N train.py` (README.md:21, utils.py:339-375)."""
    import json
    root = os.path.dirname(HERE)
    env = dict(os.environ, ICAMD_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8",
           "--hw", "64", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["n_ranks_seen"] == 2 and rec["config"]["global_batch"] == 16
    assert rec["scaling"] == "weak" and rec["value"] > 0 and rec["steps"] == 2
