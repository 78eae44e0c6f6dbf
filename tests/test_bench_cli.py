"""bench.py's --gpus contract: the flag decides how many ranks run, and a
box that cannot give that many GPUs makes the run fail loudly instead of printing n_gpus: 1."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None, timeout=600):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=e, timeout=timeout)


def test_more_ranks_than_gpus_is_refused_before_anything_runs():
    import torch
    n = torch.cuda.device_count()
    r = _run(["--gpus", str(n + 1) if n else "2"])
    assert r.returncode == 2 and r.stdout.strip() == ""
    assert "refusing" in r.stderr


def test_launcher_world_must_match_the_flag():
    r = _run(["--gpus", "1"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "WORLD_SIZE=2" in r.stderr


@pytest.mark.gpu
def test_gpus_flag_starts_that_many_ranks_and_relays_one_line():
    """The launcher path end to end on a one-GPU box: two rank processes (both on GPU 0, gloo:
    a functional check, RCCL refuses two ranks on one device), one JSON line with n_gpus = 2."""
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--frames", "4", "--settle-s", "0.01",
              "--map-points", "200000", "--no-cpu-baseline", "--no-subrecords"],
             env={"VELO_BENCH_ONE_DEVICE": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["exchange"]["ranks"] == 2
    assert out["frames_per_s"] > 0 and "FUNCTIONAL" in out["config"]["parallelism"]


@pytest.mark.gpu
def test_refused_capi_communicator_at_world_2_falls_back_on_every_rank_and_marks_the_line():
    """the first contact of the C-ABI RCCL transport with more than one rank must not be
    able to hang or blank the run.  Reachable on a one-GPU box: two ranks on GPU 0 (gloo rendezvous) ask for
    --exchange capi; every rank measures with the torch transport first, then tries the C-ABI transport in a
    fresh child process; RCCL refuses two ranks on one device, the children leave with a non-zero code (6 =
    refused, 5 = their own watchdog), the parents agree, and rank 0 prints the line it holds, marked.  The
    run ends, with ONE line, exit code 0 (no process that touched a failing collective exits 0: the children
    did not), and a machine-readable capi_transport field (bench.py watchdog)."""
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--frames", "4", "--settle-s", "0.01",
              "--map-points", "200000", "--no-cpu-baseline", "--no-subrecords", "--exchange", "capi",
              "--capi-timeout-s", "90"],
             env={"VELO_BENCH_ONE_DEVICE": "1"}, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0
    # (measured on the MI355X box: RCCL answers ncclInvalidUsage for two ranks on one device -> "unavailable",
    # in ~6 s; "timeout" is what a hang gives -- seen when the children inherited the launcher's agent-store
    # variables and waited for each other, which is how that bug was found)
    assert out["capi_transport"] in ("ok", "unavailable", "timeout")
    if out["capi_transport"] != "ok":
        assert "NOT measured" in out["exchange"]["note"]
        assert out["exchange"]["transport"].startswith("torch.distributed")
        assert "ended with exit code" in r.stderr
