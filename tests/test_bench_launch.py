"""bench.py's own launcher (CPU test): `python bench.py --gpus N` with no WORLD_SIZE in the environment must start N
ranks itself -- one process per GPU through torch.distributed.run -- before it touches the GPU; a rank count that does
not match --gpus is an error, never a silent 1-GPU run.  `--launch-check` makes the ranks rendezvous over gloo and
report instead of benchmarking, so the test needs no GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                              "TORCHELASTIC_RUN_ID")}
    return env


def test_gpus_2_starts_two_ranks():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], env=_env(), cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout                    # rank 0 prints ONE line
    r = json.loads(lines[0])
    assert r["gpus_requested"] == 2 and r["ranks_seen"] == 2
    assert len(set(r["pids"])) == 2 and os.getpid() not in r["pids"]


def test_rank_count_mismatch_is_an_error():
    env = dict(_env(), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE 1" in p.stderr
