set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/s7
mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_robustness.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py tests/test_inria_profile.py -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -5 $OUT/pytest.log
B="--no-cpu-baseline --no-extras"
for i in 1 2; do
  python3 bench.py --steps 10 --warmup 3 $B --scene stress --splats 50000000 > $OUT/s50.json 2>/dev/null
  python3 bench.py --steps 10 --warmup 3 $B --plan sort > $OUT/headsort.json 2>/dev/null
  python3 bench.py --steps 10 --warmup 3 $B --plan sort --pose 0,0,-14 > $OUT/outsort.json 2>/dev/null
  python3 - <<P
import json
for f in ("s50","headsort","outsort"):
    d=json.load(open("$OUT/%s.json"%f))
    print(f,d["ms_per_step"],d["stage_ms"])
P
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o run -- python3 bench.py --steps 5 --warmup 2 $B --scene stress --splats 50000000 > $OUT/trace.log 2>&1
f=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
python3 scripts/kernel_stats_table.py $f 40 | grep -v vectorized | grep -v elementwise > $OUT/trace.txt
rm -rf $OUT/trace
cat $OUT/trace.txt | head -14
