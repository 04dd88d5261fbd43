#!/bin/bash
# Rebuilds the library with each set of defines and prints the per-stage times of bench.py.
# Usage: bash scripts/sweep_variants.sh "<defines A>" "<defines B>" ...
for DEFS in "$@"; do
  GSR_DEFINES="$DEFS" python -m gsrast_amd.build --force > /dev/null 2>gpurun_out/build_err.txt || { echo "BUILD FAILED: $DEFS"; tail -5 gpurun_out/build_err.txt; continue; }
  echo "== $DEFS"
  timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['stage_ms'])"
done
python -m gsrast_amd.build --force > /dev/null 2>&1
