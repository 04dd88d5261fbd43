"""CPU: bench.py's own launcher. `--gpus N` without a launcher around it must start N ranks itself (as a child
torch.distributed.run) or fail loudly — never report a single-GPU number under an N-GPU label. The rendezvous, the
relay of the JSON line and the band exchange are exercised over gloo with `--dry-run` (nothing rendered, value null)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(*argv, env=None, timeout=300):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "GSR_FORCE_DIST"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH, *argv], capture_output=True, text=True, timeout=timeout, env=e)


def test_gpus_n_starts_n_ranks_and_relays_one_json_line():
    for n in (2, 3):
        p = _run("--gpus", str(n), "--dry-run", "--width", "200", "--height", "120")
        assert p.returncode == 0, p.stderr[-2000:]
        lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1, p.stdout                       # exactly ONE line on stdout, the JSON
        rec = json.loads(lines[0])
        assert rec["n_gpus"] == n and rec["dry_run"] is True and rec["exchange_ok"] is True and rec["value"] is None
        assert rec["bands"][0][0] == 0 and rec["bands"][0][-1] == 8 and len(rec["bands"][0]) == n + 1
        assert rec["bands"][1] != rec["bands"][0]              # the re-cut moved a boundary towards the heavier rows


def test_gpus_n_with_fewer_devices_fails_loudly():
    p = _run("--gpus", "2")                                    # no HIP device in the CPU container
    assert p.returncode != 0 and p.stdout.strip() == ""
    assert "refusing to report" in p.stderr


def test_world_size_must_match_gpus():
    p = _run("--gpus", "2", env={"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and p.stdout.strip() == "" and "WORLD_SIZE=4" in p.stderr


def test_single_gpu_run_without_a_device_fails_loudly():
    p = _run("--gpus", "1", "--steps", "1")
    assert p.returncode != 0 and p.stdout.strip() == "" and "no CPU path" in p.stderr
