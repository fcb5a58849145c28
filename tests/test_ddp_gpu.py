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
