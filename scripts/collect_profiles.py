"""Turns the raw rocprofv3 / bench outputs of one GPU run (gpurun_out/) into the summaries kept under profiles/.
Usage: python scripts/collect_profiles.py <tag of the run, e.g. r01e>"""
import csv, json, re, sys
tag = sys.argv[1]
rows = list(csv.DictReader(open(f"gpurun_out/prof_{tag}/run_kernel_stats.csv")))
line = [l for l in open("gpurun_out/bench_prof.log") if l.startswith('{"metric')][-1].strip()
with open("profiles/r01_blocks_kernel_stats.txt", "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline   (block binning plan, the default for this frame)\n# bench line of the same run:\n" + line + "\n")
    f.write("# kernel stats (Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs, StdDev):\n")
    f.write(",".join('"%s"' % k for k in rows[0].keys()) + "\n")
    for r in rows[:26]:
        f.write(",".join('"%s"' % r[k] if k == "Name" else r[k] for k in r.keys()) + "\n")
with open("profiles/r01_blocks_pmc.txt", "w") as f:
    f.write("# rocprofv3 --pmc <set> --kernel-trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline   (one run per counter set; per-kernel averages over dispatches; block binning plan)\n")
    f.write("# FETCH_SIZE / WRITE_SIZE are in KiB. On gfx950 FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads (MI355X_MICROARCH.md, HBM): double it before comparing with bytes.\n")
    for i in range(3):
        f.write(open(f"gpurun_out/pmc_{tag}_sum_{i}.txt").read())
vals = {}
for i in range(2):
    for l in open(f"gpurun_out/pmc_{tag}_sum_{i}.txt"):
        k = l.split()[0]
        for n, v in re.findall(r"(\w+)=([0-9.e+]+)", l):
            vals.setdefault(k, {})[n] = float(v)
traffic = lambda k: int((2 * vals[k]["FETCH_SIZE"] + vals[k]["WRITE_SIZE"]) * 1024)
j = json.load(open("profiles/pmc_traffic_r01.json"))
j["blocks"] = {"duplicate": traffic("block_emit_kernel"), "preprocess": traffic("preprocess_kernel"), "blend": traffic("blend_blocks_kernel")}
json.dump(j, open("profiles/pmc_traffic_r01.json", "w"), indent=1)
for name, out in (("bench_default", "r01_bench_default.json"), ("bench_backward", "r01_bench_backward.json"), ("bench_4k", "r01_bench_4k.json"), ("bench_overlap", "r01_bench_overlap.json")):
    try:
        d = json.loads([l for l in open(f"gpurun_out/{name}.log") if l.startswith('{"metric')][-1])
        open(f"profiles/{out}", "w").write(json.dumps(d, indent=1))
        print(name, d["ms_per_step"], d["stage_ms"], d["roofline"]["frac"])
    except Exception as e:
        print(name, "missing", e)
