#!/bin/bash
# What one launch of colors_visible_kernel costs the memory: read requests at the L2's memory side, by size.
# Run ON THE GPU BOX: bash scripts/pmc_colors.sh OUTDIR   (one rocprofv3 --pmc pass per counter set; the program after -- is python3 itself)
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=${1:-gpurun_out/pmc_colors}
mkdir -p $OUT
ARGS="--steps 3 --warmup 2 --no-cpu-baseline --no-extras --scene stress --splats 50000000"
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $set | tr ' ' '_')
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$name -o run -- python3 bench.py $ARGS > $OUT/$name.log 2>&1
  f=$(find $OUT/$name -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then python3 scripts/pmc_summary.py "$f" > $OUT/$name.txt; else echo "no counter file for $set" > $OUT/$name.txt; tail -5 $OUT/$name.log >> $OUT/$name.txt; fi
  grep -E "colors_visible|onesweep|emit_chunk|preprocess_kernel|visible_compact" $OUT/$name.txt
done
