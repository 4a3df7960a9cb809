"""bench.py launches itself for --gpus N > 1: the parent starts torch.distributed.run as a CHILD process
before touching any GPU, relays rank 0's JSON line and exits with the child's code.  Exercised here on the CPU box through
--dry (gloo, no GPU work): launcher, process group, the all-reduce of the packed normal equations."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _run(*extra):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--backend", "gloo", "--dry"] + list(extra),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=300)
    return p


def test_self_launch_two_ranks_gloo_dry():
    p = _run("--gpus", "2", "--samples", "1000")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, p.stdout          # ONE JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and d["dry"] is True
    assert d["samples_reduced"] == 2000.0     # the count travels inside the one all-reduce payload
    # round 4: the legs a multi-GPU run adds, through their CPU plumbing -- configs[2]'s R-factor all-gather + fold, configs[4]'s split
    # of the 256 chains, and rank 0 waiting for the other ranks' processes to be gone before it would start the library child
    assert d["legs"] == ["config4", "config3_sharded", "config5_sharded"]
    assert d["r_factor_gather_fold_rel_err"] <= 1e-12
    assert d["config5_chains_per_rank"] == [128, 128] and d["config5_chains_total"] == 256
    assert d["distinct_pids"] == 2 and d["other_ranks_exited"] is True


def test_self_launch_eight_ranks_gloo_dry():
    """The driver's largest run (`--gpus 8`) through its CPU plumbing: eight ranks, every leg of the one JSON line."""
    p = _run("--gpus", "8", "--samples", "500")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["n_ranks_seen"] == 8 and d["samples_reduced"] == 4000.0
    assert d["legs"] == ["config4", "config3_sharded", "config5_sharded"]
    assert d["r_factor_gather_fold_rel_err"] <= 1e-12
    assert d["config5_chains_per_rank"] == [32] * 8 and d["distinct_pids"] == 8 and d["other_ranks_exited"] is True


def test_single_rank_dry_needs_no_launcher():
    p = _run("--gpus", "1", "--samples", "10")
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["n_ranks_seen"] == 1


def test_world_size_mismatch_is_an_error_not_an_assert():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry", "--backend", "gloo"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=120)
    assert p.returncode == 2 and "WORLD_SIZE=3" in p.stderr


def test_single_process_argument_path_dry():
    """`bench.py --single-process --gpus N` = configs[3] through the library's in-process RCCL path (one process, no torch.distributed,
    no launcher); on the CPU box only the argument path can run."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--single-process", "--gpus", "8", "--dry", "--samples", "1000"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=120)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert d == {"metric": "config4_library", "dry": True, "n_gpus": 8, "steps": 20, "samples_per_gpu": 1000,
                 "legs": ["config4_library", "config3_library", "config5_library"]}
