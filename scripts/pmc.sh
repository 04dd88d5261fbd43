#!/bin/bash
# Usage: bash scripts/pmc.sh <tag> "<counters pass 1>" ["<counters pass 2>" ...] -- <python script + args>
# One rocprofv3 run per counter set (PMC only + kernel trace, as the pool requires); per-kernel
# averages are written to gpurun_out/pmc_<tag>.txt
set -u
TAG=$1; shift
SETS=()
while [ "$1" != "--" ]; do SETS+=("$1"); shift; done
shift
export TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/pmc_${TAG}.txt
i=0
for S in "${SETS[@]}"; do
  OUT=gpurun_out/pmc_${TAG}_$i
  timeout -k 10 ${PMC_TIMEOUT:-180} rocprofv3 --pmc $S --kernel-trace --output-format csv -d "$OUT" -o run -- python3 "$@" > gpurun_out/pmc_${TAG}_$i.log 2>&1
  F=$(find "$OUT" -name '*counter_collection.csv' | head -1)
  echo "## counters: $S" >> gpurun_out/pmc_${TAG}.txt
  if [ -n "$F" ]; then
    python3 - "$F" >> gpurun_out/pmc_${TAG}.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = r["Kernel_Name"].split("(")[0][-60:]
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "dispatches", len(next(iter(d.values()))))
PY
  else
    echo "no counter file; log tail:" >> gpurun_out/pmc_${TAG}.txt; tail -5 gpurun_out/pmc_${TAG}_$i.log >> gpurun_out/pmc_${TAG}.txt
  fi
  i=$((i+1))
done
cat gpurun_out/pmc_${TAG}.txt
