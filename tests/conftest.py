import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "perf: asserts a duration (latency / wall clock), not a result; always collected LAST, so a "
                                       "slow or differently clocked box can never pre-empt a parity test under -x")


def pytest_collection_modifyitems(config, items):
    """parity first, timing last: every test marked `perf` moves behind all the others (stable order within each group)"""
    items.sort(key=lambda it: 1 if it.get_closest_marker("perf") else 0)


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def pytest_sessionstart(session):
    """`-m gpu` runs only: start the 1-rank RCCL job of tests/test_gpu_rccl.py as a FRESH child process now, before anything in
    this process has touched the GPU (a process that has initialised the GPU must not fork + exec on the GPU boxes).  The child
    runs the compress CLI under WORLD_SIZE=1 with backend nccl (= RCCL) while the first tests run; the test collects it."""
    import subprocess
    import tempfile
    expr = session.config.getoption("-m") or ""
    session.config._sgic_rccl_child = None
    if "gpu" not in expr or "not gpu" in expr:
        return
    try:
        import torch
        if torch.cuda.device_count() < 1:      # counting devices does not initialise the GPU
            return
        import numpy as np
        from PIL import Image
    except Exception:   # noqa: BLE001 -- no GPU stack: the gpu tests will say so themselves
        return
    root = tempfile.mkdtemp(prefix="sgic_rccl_")
    src = os.path.join(root, "imgs")
    os.makedirs(src)
    rng = np.random.default_rng(5)
    for i in range(6):
        Image.fromarray(rng.integers(0, 256, (256, 256, 3), dtype=np.uint8)).save(os.path.join(src, f"im{i}.png"))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    log = open(os.path.join(root, "child.log"), "w")
    proc = subprocess.Popen([sys.executable, os.path.join(ROOT, "compress.py"), "--dataset_dir", src, "--save_dir",
                             os.path.join(root, "out"), "--small", "--batch_size", "4"], env=env, stdout=log, stderr=subprocess.STDOUT,
                            cwd=ROOT)
    # the same CLI as a TWO-rank job on this one card (backend gloo, both ranks on cuda:0): shards, the gather and rank 0's index
    # assembly with more than one process, on real GPU outputs (tests/test_gpu_rccl.py compares it with the 1-rank job's files)
    port2 = _free_port()
    two = []
    for r in range(2):
        env2 = dict(os.environ, WORLD_SIZE="2", RANK=str(r), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port2),
                    SGIC_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
        lg = open(os.path.join(root, f"two_rank{r}.log"), "w")
        two.append((subprocess.Popen([sys.executable, os.path.join(ROOT, "compress.py"), "--dataset_dir", src, "--save_dir",
                                      os.path.join(root, "out2"), "--small", "--batch_size", "2"], env=env2, stdout=lg,
                                     stderr=subprocess.STDOUT, cwd=ROOT), lg.name))
    session.config._sgic_rccl_child = {"proc": proc, "root": root, "log": log.name, "src": src, "two": two}


def pytest_sessionfinish(session, exitstatus):
    ch = getattr(session.config, "_sgic_rccl_child", None)
    if ch and ch["proc"].poll() is None:
        ch["proc"].kill()          # the exact child this session started
    for pr, _ in (ch or {}).get("two", []):
        if pr.poll() is None:
            pr.kill()


@pytest.fixture(scope="session")
def rccl_child(request):
    return getattr(request.config, "_sgic_rccl_child", None)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
