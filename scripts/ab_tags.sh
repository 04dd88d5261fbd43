#!/bin/bash
# A/B of library variants built with `GSR_DEFINES=... python -m gsrast_amd.build --tag <tag>`: runs the same bench command
# under every tag (GSR_LIB_TAG) and prints frame time + stage times. Usage: scripts/ab_tags.sh OUT "tag1 tag2 ..." bench args...
out=$1; tags=$2; shift 2
mkdir -p "$(dirname "$out")"
: > "$out"
for t in $tags; do
  tag=$t; [ "$t" = "base" ] && tag=""
  GSR_LIB_TAG=$tag python bench.py --no-cpu-baseline --no-extras "$@" > /tmp/ab_$t.json 2> /tmp/ab_$t.err || { echo "$t FAILED" >> "$out"; tail -3 /tmp/ab_$t.err >> "$out"; continue; }
  python - "$t" /tmp/ab_$t.json >> "$out" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(f"{sys.argv[1]:8s} {d['ms_per_step']:.4f} ms  R={d['config']['num_rendered']}  " + "  ".join(f"{k}={v:.4f}" for k, v in d["stage_ms"].items()))
PY
done
cat "$out"
