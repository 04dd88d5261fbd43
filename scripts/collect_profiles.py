"""Turns gpurun_out/<round>/ (scripts/profile.sh, run on the GPU box) into the summaries committed under profiles/.
Usage: python scripts/collect_profiles.py [round, default r05]"""
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_stats_table as kst

ROUND = sys.argv[1] if len(sys.argv) > 1 else "r06"
SRC, DST = f"gpurun_out/{ROUND}", "profiles"


HEADERS = {
    "clock_ramp": ("# python scripts/clock_ramp.py on one MI355X: per-call host times of the headline frame in ms (o = blend beside the emission, r = tiles reordered).\n"
                   "# The first dozen frames after a few milliseconds of idling run on lower clocks, whatever the tile history (settled after two calls):\n"
                   "# bench.py renders --spinup (16) untimed frames before its warm-up steps (config.spinup_frames).\n"),
    "micro_event_gap": ("# scripts/micro/event_gap.hip on one MI355X: kernel A, [event], kernel B on one stream; a second stream waits for the event and runs kernel C\n"),
    "micro_valu_issue": ("# scripts/micro/valu_issue.hip on one MI355X: what a wave-instruction of the classes the blend is made of holds a SIMD for, by the\n"
                         "# number of waves resident on the SIMD. First figure: cycles of the median wave; in brackets: from the kernel's WALL time (launch and\n"
                         "# tail included) — the conservative one, which bench.py's VALU_CYCLES are taken from (4 and 8 waves per SIMD).\n"),
    "deep_tiles": ("# python scripts/deep_tiles_table.py on one MI355X: blend stage (ms, HIP events) with one wave per tile and with 4 / 8 / 16 waves that share a\n"
                   "# tile's walk (GSR_FLAG_NO_DEEP_TILES, GSR_FLAG_DEEP_TILES_ALL, GSR_FLAG_DEEP_WAVES_8 / _16), and what the library picks by itself.\n"),
    "path_stages": ("# python scripts/path_stages.py on one MI355X: stage times (ms, HIP events of a profiled frame) along bench.py's camera path, every 4th pose;\n"
                    "# frame = host time of an unprofiled frame of the same pose; ovl = the blend ran beside the emission (its time then lies inside the emission's).\n"
                    "# Poses 0-40 (fewer than 16 instances per visible Gaussian): four waves per tile (round 5, one wave: blend 0.50-0.62 ms there).\n"),
    "ply_path": ("# python scripts/ply_path.py on one MI355X: BASELINE config 2's code path at its size — a 5 834 784-splat scene FILE in the reference's format\n"
                 "# (scripts/make_ply.py from scenes.trained_like: the garden .ply is not available offline) through the loader and the forward.\n"),
}


def bench_line(path):
    try:
        lines = [ln for ln in open(path) if ln.startswith('{"metric')]
        return json.loads(lines[-1]) if lines else None
    except OSError:
        return None


def stats_file(name):
    f = glob.glob(f"{SRC}/trace_{name}/**/*kernel_stats.csv", recursive=True)
    return f[0] if f else None


def write_trace(name, cmd):
    import csv
    f = stats_file(name)
    if not f:
        print("missing trace", name)
        return {}
    line = bench_line(f"{SRC}/trace_{name}.log")
    # (the scene generator's torch kernels — 50 M scene built on the device — are not the library's)
    rows = [r for r in csv.DictReader(open(f)) if "at::native" not in r["Name"] and not r["Name"].startswith("void at::")]
    with open(f"{DST}/{ROUND}_{name}_kernel_stats.txt", "w") as o:
        o.write(f"# rocprofv3 --kernel-trace --stats -- python3 bench.py {cmd}\n# bench line of the same run:\n{json.dumps(line)}\n")
        o.write("# kernel  calls  average / min / max microseconds  share of device time\n")
        for r in rows[:28]:
            o.write(f"{kst.short(r['Name']):44s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs']) / 1e3:9.1f} "
                    f"min_us={float(r['MinNs']) / 1e3:9.1f} max_us={float(r['MaxNs']) / 1e3:9.1f}  {float(r['Percentage']):5.1f} %\n")
    return {kst.short(r["Name"]): float(r["AverageNs"]) / 1e3 for r in rows}


def pmc_values(name):
    vals = {}
    try:
        for ln in open(f"{SRC}/pmc_{name}.txt"):
            k = ln.split()[0] if ln.split() else ""
            for n, v in re.findall(r"(\w+)=([0-9.e+-]+)", ln):
                try:
                    vals.setdefault(k, {})[n] = float(v)
                except ValueError:
                    pass
    except OSError:
        print("missing pmc", name)
    return vals


def main():
    os.makedirs(DST, exist_ok=True)
    write_trace("head", "--steps 20 --warmup 5 --no-cpu-baseline --no-extras   (the headline frame)")
    write_trace("head_precomp", "--steps 20 --warmup 5 --no-cpu-baseline --no-extras --colors-precomp   (the headline frame, colours passed as colorsPrecomp)")
    write_trace("outside", "--steps 5 --warmup 2 --no-cpu-baseline --no-extras --pose 0,0,-14")
    write_trace("far", "--steps 5 --warmup 2 --no-cpu-baseline --no-extras --pose 0,0,-30")
    write_trace("bound", "--steps 5 --warmup 2 --no-cpu-baseline --no-extras --opacity-scale 0.1")
    write_trace("stress50M", "--steps 5 --warmup 2 --no-cpu-baseline --no-extras --scene stress --splats 50000000")
    write_trace("stress50M_precomp", "--steps 5 --warmup 2 --no-cpu-baseline --no-extras --scene stress --splats 50000000 --colors-precomp")
    write_trace("4k", "--steps 5 --warmup 2 --no-cpu-baseline --no-extras --width 3840 --height 2160")
    with open(f"{DST}/{ROUND}_pmc.txt", "w") as o:
        o.write("# rocprofv3 --pmc <set> --kernel-trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras [frame]\n"
                "# one run per counter set; per-kernel averages over dispatches. FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE\n"
                "# tallies a 128-B request of a wide coalesced read at 64 B (MI355X_MICROARCH.md, HBM): doubled before it is compared with bytes.\n"
                "# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, GRBM_GUI_ACTIVE is summed over the 8 XCDs.\n")
        for name, what in (("head_sq", "headline frame"), ("head_lds", "headline frame"), ("head_fetch", "headline frame"),
                           ("head_write", "headline frame"), ("precomp_fetch", "headline frame, --colors-precomp"),
                           ("precomp_write", "headline frame, --colors-precomp"), ("bound_sq", "opacities x 0.1"),
                           ("outside_sq", "pose (0,0,-14)"), ("far_sq", "pose (0,0,-30)"), ("far_onewave_sq", "pose (0,0,-30), --no-deep-tiles"),
                           ("head_cls", "headline frame"), ("head_cls2", "headline frame"), ("far_cls", "pose (0,0,-30)"), ("far_cls2", "pose (0,0,-30)"),
                           ("far_onewave_cls", "pose (0,0,-30), --no-deep-tiles"), ("far_onewave_cls2", "pose (0,0,-30), --no-deep-tiles"),
                           ("bound_cls", "opacities x 0.1"), ("bound_cls2", "opacities x 0.1"), ("outside_cls", "pose (0,0,-14)"), ("outside_cls2", "pose (0,0,-14)"),
                           ("stress_fetch", "50 M stress scene"), ("stress_write", "50 M stress scene"),
                           ("stress_precomp_fetch", "50 M stress scene, --colors-precomp"),
                           ("stress_precomp_write", "50 M stress scene, --colors-precomp")):
            o.write(f"## {name}: {what}\n")
            try:
                o.write(open(f"{SRC}/pmc_{name}.txt").read())
            except OSError:
                o.write("(missing)\n")
    fetch, write, sq = pmc_values("head_fetch"), pmc_values("head_write"), pmc_values("head_sq")

    def traffic(k):
        if k in fetch and k in write and "FETCH_SIZE" in fetch[k] and "WRITE_SIZE" in write[k]:
            return int((2 * fetch[k]["FETCH_SIZE"] + write[k]["WRITE_SIZE"]) * 1024)
        return None

    def clock(v):       # effective clock from GRBM_GUI_ACTIVE (sum over 8 XCDs) and the dispatch's duration
        return None

    out = {"_comment": f"HBM bytes per launch from rocprofv3 PMC passes (profiles/{ROUND}_pmc.txt): (2 x FETCH_SIZE + WRITE_SIZE) KiB -> bytes, "
                       "FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950; blend_insts: SQ_INSTS_VALU / SQ_INSTS_SALU per "
                       "launch of the blend kernel on the same frame. Workload: the bench.py default frame (N=5834784, 1920x1080).",
           "blocks": {}, "sort": {}}
    for stage, kern in (("duplicate", "block_emit_kernel"), ("preprocess", "preprocess_kernel"), ("blend", "blend_blocks_kernel")):
        t = traffic(kern)
        if t is not None:
            out["blocks"][stage] = t
    # the preprocess kernel's traffic with the colours precomputed, bench frame and 50 M
    for key, fn, wn in (("preprocess_colors_precomp", "precomp_fetch", "precomp_write"), ("preprocess_50M", "stress_fetch", "stress_write"),
                        ("preprocess_50M_colors_precomp", "stress_precomp_fetch", "stress_precomp_write")):
        f2, w2 = pmc_values(fn).get("preprocess_kernel", {}), pmc_values(wn).get("preprocess_kernel", {})
        if "FETCH_SIZE" in f2 and "WRITE_SIZE" in w2:
            out["blocks"][key] = int((2 * f2["FETCH_SIZE"] + w2["WRITE_SIZE"]) * 1024)
    old = json.load(open(f"{DST}/pmc_traffic_r01.json"))
    out["sort"] = old.get("sort", {})
    out["sort_note"] = "sort-plan figures are round 1's (profiles/r01_final_pmc.txt): that plan's kernels other than the blend are unchanged"
    blend_kernels = ("blend_blocks_kernel", "blend_deep_kernel", "blend_deep8_kernel", "blend_deep16_kernel", "blend_group_kernel", "blend_wave_kernel")

    def blend_counters(*sets):
        """The blend kernel's counters of several PMC passes of one frame, merged (per launch averages)."""
        merged, kern = {}, None
        for name in sets:
            pv = pmc_values(name)
            k = next((k for k in blend_kernels if k in pv), None)
            if k:
                kern = kern or k
                merged.update({n: v for n, v in pv[k].items() if n not in ("dispatches", "avg_us") or n not in merged})
        return (dict(merged, kernel=kern, clock_ghz=2.4) if merged else {})

    b = blend_counters("head_sq", "head_cls", "head_cls2")
    if "SQ_INSTS_VALU" in b:
        out["blend_insts"] = dict(b, source=f"profiles/{ROUND}_pmc.txt, sets head_sq / head_cls / head_cls2")
    for frame in ("bound", "outside", "far", "far_onewave"):
        v = blend_counters(f"{frame}_sq", f"{frame}_cls", f"{frame}_cls2")
        if v:
            out[f"blend_insts_{frame}"] = v
    json.dump(out, open(f"{DST}/pmc_traffic_{ROUND}.json", "w"), indent=1)
    import shutil
    if os.path.exists(f"{SRC}/band_projection.json"):
        shutil.copy(f"{SRC}/band_projection.json", f"{DST}/band_projection.json")
    for txt in ("band_timings", "band_timings_4k", "parity", "soak", "history_similarity", "soak_trained_like", "micro_gather_dc",
                "micro_scatter_records", "micro_xcd_placement", "micro_event_gap", "clock_ramp", "micro_valu_issue", "deep_tiles", "path_stages",
                "ply_path", "soak_garden_like"):
        if os.path.exists(f"{SRC}/{txt}.txt"):
            head = HEADERS.get(txt)
            if head:
                with open(f"{DST}/{ROUND}_{txt}.txt", "w") as o:
                    o.write(head + open(f"{SRC}/{txt}.txt").read())
            else:
                shutil.copy(f"{SRC}/{txt}.txt", f"{DST}/{ROUND}_{txt}.txt")
    thr = sorted(glob.glob(f"{SRC}/thresholds_*.txt"))
    if thr:
        with open(f"{DST}/{ROUND}_trained_like.txt", "w") as o:
            o.write("# scripts/thresholds_check.py on one MI355X: is the library's own choice (binning plan, the blend's feed, blend beside the emission)\n"
                    "# within 5 % of the best forced one? Scenes: scenes.trained_like (flat splats on surfaces, heavy-tailed scales, bimodal opacity, 500 huge\n"
                    "# background splats) and the garden-like stand-in, 1 M and 5.83 M splats, 1920 x 1080, a random first-person tour of 60 poses each (every\n"
                    "# fifth from further out, every seventh from 48 units away). Round 4's switch points (plan at R/V = 6, feed and overlap at 48) on the\n"
                    "# first version of these tours: trained_like 1 M 48 %, 5.83 M 52 % of the poses within 5 % (gpurun_out/r5c).\n")
            for f in thr:
                o.write("\n" + open(f).read())
    for name in ("bench_default", "bench_backward", "bench_backward_outside", "bench_backward_nolists", "bench_4k", "bench_stress50M",
                 "bench_stress50M_precomp", "bench_stress50M_inria_sh3", "bench_forced_dist_1rank"):
        d = bench_line(f"{SRC}/{name}.json")
        if d is None:
            print(name, "missing")
            continue
        open(f"{DST}/{ROUND}_{name}.json", "w").write(json.dumps(d, indent=1))
        print(name, d["ms_per_step"], {k: v for k, v in d["stage_ms"].items() if v}, "roofline", d["roofline"]["frac"])


if __name__ == "__main__":
    main()
