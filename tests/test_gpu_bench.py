"""GPU: bench.py end to end on one device — the plain single-GPU line, and the sharded code path forced onto one rank
(GSR_FORCE_DIST=1: nccl process group, scene broadcast, band exchange object, re-cut, per-rank split), which is all of the
multi-GPU path a one-GPU box can run before the driver's 8-GPU node does."""
import json
import os
import subprocess
import sys

import pytest

from helpers import ROOT

pytestmark = pytest.mark.gpu

CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline")


def _run_bench(extra_env, *args):
    env = dict(os.environ, **extra_env)
    env.setdefault("MASTER_PORT", "29541")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", "--no-cpu-baseline",
                          "--splats", "300000", *args], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = out.stdout.strip().splitlines()[-1]
    return json.loads(line)


def test_single_gpu_line_has_the_contract_keys():
    j = _run_bench({})
    for k in CONTRACT_KEYS:
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["warmup"] == 2 and j["higher_is_better"] is True
    assert j["metric"] == "forward_msplats_per_s" and j["unit"] == "Msplats/s" and j["dtype"] == "f32" and j["data"] == "synthetic"
    assert abs(j["value"] - 0.3 / (j["ms_per_step"] * 1e-3)) <= 1e-2 * j["value"]
    r = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert "workload" in j["config"] and j["config"]["frame_alg_gbs"] > 0
    assert j["config"]["spinup_frames"] == 16            # (untimed frames in front of the warm-up steps are said in the line)
    j0 = _run_bench({}, "--spinup", "0", "--no-extras")
    assert j0["config"]["spinup_frames"] == 0 and j0["steps"] == 3 and j0["warmup"] == 2


def test_sharded_path_forced_onto_one_rank():
    j = _run_bench({"GSR_FORCE_DIST": "1", "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1"})
    for k in CONTRACT_KEYS:
        assert k in j, k
    cfg = j["config"]
    assert j["n_gpus"] == 1 and cfg["rccl_ranks"] == 1 and cfg["parallelism"] == "tile-rows x1"
    assert cfg["band_exchange"]["transport"] in ("rccl", "torch") and cfg["bands"][0] == 0
    (pr,) = cfg["per_rank"]
    assert pr["rank"] == 0 and pr["tile_rows"] == [cfg["bands"][0], cfg["bands"][1]]
    # the split a poor N-GPU number would be attributed with: device time of the render and of the exchange behind it
    assert pr["render_ms"] > 0.0 and pr["exchange_ms"] >= 0.0 and pr["render_ms"] <= 1.5 * pr["ms_per_step"] + 0.2
    assert pr["num_rendered"] == cfg["num_rendered"] > 0
