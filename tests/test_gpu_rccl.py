"""The one collective of the path on the real backend, within one GPU: backend "nccl" IS RCCL on ROCm.
No multi-GPU box is available to the builder, so the 1 -> 8 curve does not exist; what CAN be shown on one card is that
librccl loads, a communicator is built, and the two collectives of compress.py -- the error-flag all-reduce and the
all-gather of the CLIP vectors (reference counterpart: /root/reference/src/compress.py:43-55,294-306) -- run ON THE DEVICE and
give the same index as the run without a process group.  The world-2 logic is covered on gloo (tests/test_distributed_cpu.py)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_cli_under_one_rank_rccl_child_process(rccl_child, tmp_path):
    """compress.py as a 1-rank RCCL job (WORLD_SIZE=1, backend nccl) in a fresh child process started by conftest before this
    process touched the GPU: every collective of the multi-GPU path runs, and the outputs are byte-identical to a plain run."""
    if rccl_child is None:
        pytest.skip("the RCCL child is started by conftest at session start of a `-m gpu` run (before this process touches the GPU)")
    rc = rccl_child["proc"].wait(timeout=900)
    log = open(rccl_child["log"]).read()
    assert rc == 0, log[-3000:]
    rec = json.loads([ln for ln in log.splitlines() if ln.startswith("{")][-1])
    assert rec["collectives"] == "nccl" and rec["images"] == 6 and rec["n_gpus"] == 1, rec
    from sgic_amd import compress
    out = tmp_path / "plain"
    assert "WORLD_SIZE" not in os.environ
    assert compress.main(["--dataset_dir", rccl_child["src"], "--save_dir", str(out), "--small", "--batch_size", "4"]) == 0
    child_out = os.path.join(rccl_child["root"], "out")
    a = open(os.path.join(child_out, "faiss", "index.faiss"), "rb").read()
    b = open(out / "faiss" / "index.faiss", "rb").read()
    assert len(a) == 45 + 6 * 4 * _dim(a) and a == b, "the gathered index differs from the single-process one"
    for i in range(6):
        assert open(os.path.join(child_out, "bitstreams", f"im{i}.c2df"), "rb").read() == open(out / "bitstreams" / f"im{i}.c2df", "rb").read()
    print(f"[rccl] 1-rank nccl job: {rec}")


def test_two_rank_job_on_one_card_matches_the_one_rank_job(rccl_child):
    """compress.py as a TWO-rank job (backend gloo, both ranks on this card; started by conftest): each rank compresses its
    contiguous shard, the CLIP vectors are gathered, rank 0 assembles the index -- and every output file is byte-identical to the
    1-rank RCCL job's (same kernels, M-invariant arithmetic: a shard of 3 images gives the streams a run of 6 gives)"""
    if rccl_child is None:
        pytest.skip("started by conftest at session start of a `-m gpu` run")
    logs = []
    for pr, lg in rccl_child["two"]:
        rc = pr.wait(timeout=900)
        logs.append(open(lg).read())
        assert rc == 0, logs[-1][-3000:]
    assert rccl_child["proc"].wait(timeout=900) == 0
    rec = json.loads([ln for ln in logs[0].splitlines() if ln.startswith("{")][-1])
    assert rec["collectives"] == "gloo" and rec["images"] == 6 and rec["n_gpus"] == 2, rec
    one, two = os.path.join(rccl_child["root"], "out"), os.path.join(rccl_child["root"], "out2")
    for sub, names in (("bitstreams", [f"im{i}.c2df" for i in range(6)]), ("clip_vecs", [f"im{i}.npy" for i in range(6)]),
                       ("faiss", ["index.faiss"])):
        for n in names:
            assert open(os.path.join(one, sub, n), "rb").read() == open(os.path.join(two, sub, n), "rb").read(), (sub, n)
    ids = [[os.path.basename(x) for x in open(os.path.join(d, "faiss", "ids.txt")).read().split()] for d in (one, two)]   # doc ids carry the save dir
    assert ids[0] == ids[1] and len(ids[0]) == 6
    print(f"[gloo x2 on one card] {rec}")


def _dim(index_bytes):
    return int(np.frombuffer(index_bytes[4:8], dtype="<i4")[0])


def test_gather_and_error_flag_on_device_world1_nccl():
    """in-process: init backend nccl at world 1, run dist.gather_vectors and compress.py's error-flag all-reduce on cuda tensors"""
    import socket
    import torch.distributed as dist
    from sgic_amd.dist import gather_vectors
    assert not dist.is_initialized()
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        assert dist.get_backend() == "nccl"
        g = torch.Generator(device="cuda").manual_seed(3)
        local = torch.randn(37, 512, device="cuda", generator=g)
        allv = gather_vectors(local, 37, 0, 1)
        assert allv.is_cuda and allv.data_ptr() != local.data_ptr() and torch.equal(allv, local)   # went through the collective
        flag = torch.tensor([0.0], device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        rate = torch.tensor([123.5], device="cuda", dtype=torch.float64)
        dist.all_reduce(rate, op=dist.ReduceOp.SUM)
        dist.barrier()
        assert float(flag.item()) == 0.0 and float(rate.item()) == 123.5
    finally:
        dist.destroy_process_group()
