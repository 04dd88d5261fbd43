set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/s5
mkdir -p $OUT
B="--no-cpu-baseline --no-extras"
for tag in $TAGS; do
  if [ $tag != new ]; then export GSR_LIB_TAG=$tag; else unset GSR_LIB_TAG; fi
  python3 bench.py --steps 20 --warmup 5 $B > $OUT/head_$tag.json 2>/dev/null
  python3 bench.py --steps 10 --warmup 3 $B --scene stress --splats 50000000 > $OUT/s50_$tag.json 2>/dev/null
  python3 bench.py --steps 10 --warmup 3 $B --width 3840 --height 2160 > $OUT/k4_$tag.json 2>/dev/null
  python3 - <<P
import json
for f in ("head","s50","k4"):
    d=json.load(open("$OUT/%s_$tag.json"%f))
    print("$tag",f,d["ms_per_step"],{k:v for k,v in d["stage_ms"].items() if k in ("depth_order","sort_pass2")})
P
done
