import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (gfx950) GPU; run with -m gpu on the GPU box")


def pytest_collection_modifyitems(config, items):
    # a gpu-marked test on a GPU-less host is a configuration error, not a pass: skip loudly
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible (gpu tests run on the MI355X box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def start_loss_curve_oracle(config):
    """Child process computing the CPU-oracle loss curves of tests/test_zz_loss_curve_gpu.py (no GPU; half of the host's cores)."""
    import subprocess
    import tempfile
    out = os.path.join(tempfile.mkdtemp(prefix="icamd_loss_curve_"), "curves.json")
    threads = max(2, (os.cpu_count() or 4) // 2)
    try:
        threads = max(2, len(os.sched_getaffinity(0)) // 2)
    except AttributeError:
        pass
    proc = subprocess.Popen([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "_loss_curve_oracle.py"), out,
                             str(threads)], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
    config._icamd_loss_curve_job = (proc, out)
    return config._icamd_loss_curve_job


def pytest_collection_finish(session):
    # the loss-curve test's oracle needs ~5 minutes of host time: start it now, beside the other tests, if that test was selected
    import torch
    if not torch.cuda.is_available():
        return
    if any(item.nodeid.endswith("test_resnet50_loss_curve_tracks_oracle") for item in session.items):
        start_loss_curve_oracle(session.config)


def pytest_unconfigure(config):
    job = getattr(config, "_icamd_loss_curve_job", None)
    if job is not None and job[0].poll() is None:
        job[0].kill()
