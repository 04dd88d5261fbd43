#!/bin/bash
# Usage (on the GPU box, from the repo root): bash scripts_profile.sh <tag> [bench args...]
# Writes rocprofv3 kernel-trace stats for one bench.py run into gpurun_out/prof_<tag>/ and a
# compact per-kernel summary to gpurun_out/prof_<tag>_summary.txt
set -u
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p gpurun_out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o run -- python3 bench.py "$@" > gpurun_out/prof_${TAG}_bench.log 2>&1
STATS=$(find "$OUT" -name '*kernel_stats.csv' | head -1)
{
  echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py $*"
  echo "# bench line:"; grep '^{' gpurun_out/prof_${TAG}_bench.log | tail -1
  echo "# kernel stats (Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs, StdDev):"
  if [ -n "$STATS" ]; then head -40 "$STATS"; else echo "no stats file found"; ls -R "$OUT" | head; fi
} > gpurun_out/prof_${TAG}_summary.txt
tail -5 gpurun_out/prof_${TAG}_bench.log
head -30 gpurun_out/prof_${TAG}_summary.txt
