"""bench.py's self-launching multi-rank path, rehearsed on CPU (gloo, no GPU work): `python bench.py --gpus N` from a
plain shell must start N ranks itself and report n_gpus = N; a launcher/flag mismatch must fail loudly."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=300)


def test_self_launch_two_ranks_reports_n_gpus_2():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "8", "--rehearse-cpu"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                      # ONE JSON line, from rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["config"]["global_batch"] == 16
    assert len(d["per_rank_seconds"]) == 2 and d["data"] == "rehearsal" and d["scaling"] == "weak"


def test_single_rank_rehearsal():
    r = _run(["--steps", "2", "--rehearse-cpu"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_gpus_flag_must_match_world_size():
    """torchrun started 2 ranks but --gpus says 1 (or the reverse): refuse instead of printing a mislabelled n_gpus"""
    r = _run(["--gpus", "1", "--rehearse-cpu"], env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"}, drop=())
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_failed_rank_fails_the_launch():
    """a rank that dies must not leave the launcher (or its peers) hanging: non-zero exit, no JSON line"""
    r = _run(["--gpus", "2", "--steps", "1", "--rehearse-cpu"], env_extra={"SGIC_BENCH_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
