set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/s9
mkdir -p $OUT
B="--no-cpu-baseline --no-extras --steps 8 --warmup 3"
run() { python3 bench.py $B "$@" > $OUT/x.json 2>/dev/null; python3 - "$@" <<P
import json,sys
d=json.load(open("$OUT/x.json"))
c=d["config"]
print(" ".join(sys.argv[1:]), "| ms", d["ms_per_step"], "plan", c["binning_plan"], "R/N", round(c["num_rendered"]/c["splats"],2), "R/V", round(c["num_rendered"]/max(c["visible"],1),2), {k:v for k,v in d["stage_ms"].items() if v>0})
P
}
run --scene stress --splats 50000000 --plan sort
run --scene stress --splats 50000000 --plan blocks
run --scene stress --splats 20000000 --plan sort --pose 0,0,-12
run --scene stress --splats 20000000 --plan blocks --pose 0,0,-12
run --scene stress --splats 20000000 --plan sort --pose 0,0,-8
run --scene stress --splats 20000000 --plan blocks --pose 0,0,-8
run --plan sort --pose 0,0,-30
run --plan blocks --pose 0,0,-30
run --plan sort --pose 0,0,-50
run --plan blocks --pose 0,0,-50
