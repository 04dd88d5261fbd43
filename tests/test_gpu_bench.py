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


# what VERDICT r5 asked the record to carry, in the order bench.py writes it (the first eight describe the headline frame)
RECORD_KEYS = ("workload", "splats", "num_rendered", "records_staged", "binning_plan", "ms_median", "ms_p99", "untimed_frames",
               "ms_no_sorted_lists", "ms_serial", "ms_no_tile_history", "ms_faint", "ms_pose_outside", "ms_pose_far",
               "ms_forward_backward", "path_ms_p50", "path_ms_p99", "path_ms_nohist_p50", "randview_ms", "randview_ms_nohist",
               "frame_alg_gbs", "parallelism", "rccl_ranks")


def _run_bench(extra_env, *args, splats="300000"):
    env = dict(os.environ, **extra_env)
    env.setdefault("MASTER_PORT", "29541")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", "--no-cpu-baseline",
                          *(["--splats", splats] if splats else []), *args], capture_output=True, text=True, env=env, timeout=900)
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
    cfg = j["config"]
    assert "workload" in cfg and cfg["frame_alg_gbs"] > 0
    # the driver's record keeps the first 24 keys of `config`, flat scalars only: everything measured must be among them
    assert len(cfg) <= 24, list(cfg)
    assert all(isinstance(v, (int, float, str, bool)) for v in cfg.values()), cfg
    assert list(cfg)[:8] == list(RECORD_KEYS[:8])
    # every frame rendered before the timed region is said in the line: spin-up + the counted frame + the warm-up steps
    assert cfg["untimed_frames"] == 16 + 1 + 2 == j["warmup_effective"] and j["detail"]["spinup_frames"] == 16
    assert cfg["ms_median"] > 0 and cfg["ms_p99"] >= cfg["ms_median"]
    j0 = _run_bench({}, "--spinup", "0", "--no-extras")
    assert j0["config"]["untimed_frames"] == 3 and j0["detail"]["spinup_frames"] == 0 and j0["steps"] == 3 and j0["warmup"] == 2
    assert len(j0["config"]) <= 24


def test_the_default_workload_s_record_carries_every_measured_figure():
    """The driver's command (the default workload, its extras included): every figure VERDICT r5 asked for is among the 24 keys
    the driver's record keeps of `config`, as flat scalars, the headline frame's first."""
    j = _run_bench({}, splats=None)
    cfg = j["config"]
    assert len(cfg) <= 24, list(cfg)
    assert all(isinstance(v, (int, float, str, bool)) for v in cfg.values()), cfg
    for k in RECORD_KEYS:
        assert k in cfg, k
    assert list(cfg)[:8] == list(RECORD_KEYS[:8])
    assert cfg["splats"] == 5_834_784 and cfg["path_ms_p99"] >= cfg["path_ms_p50"] > 0 and cfg["ms_pose_far"] > 0
    assert 0.0 < cfg["blend_valu_frac"] < 1.0
    assert j["warmup_effective"] == cfg["untimed_frames"] == 16 + 1 + 2


def test_sharded_path_forced_onto_one_rank():
    j = _run_bench({"GSR_FORCE_DIST": "1", "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1"})
    for k in CONTRACT_KEYS:
        assert k in j, k
    cfg, det = j["config"], j["detail"]
    assert j["n_gpus"] == 1 and cfg["rccl_ranks"] == 1 and cfg["parallelism"] == "tile-rows x1"
    assert len(cfg) <= 24 and all(isinstance(v, (int, float, str, bool)) for v in cfg.values()), cfg
    assert cfg["band_exchange"] in ("rccl", "torch") and det["band_exchange"]["transport"] == cfg["band_exchange"] and det["bands"][0] == 0
    assert cfg["bands"] == f"{det['bands'][0]}-{det['bands'][1]}" and float(cfg["per_rank_render_ms"]) > 0.0
    (pr,) = det["per_rank"]
    assert pr["rank"] == 0 and pr["tile_rows"] == [det["bands"][0], det["bands"][1]]
    # the split a poor N-GPU number would be attributed with: device time of the render and of the exchange behind it
    assert pr["render_ms"] > 0.0 and pr["exchange_ms"] >= 0.0 and pr["render_ms"] <= 1.5 * pr["ms_per_step"] + 0.2
    assert pr["num_rendered"] == cfg["num_rendered"] > 0
