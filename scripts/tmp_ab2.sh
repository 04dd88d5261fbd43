set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/s4
mkdir -p $OUT
timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_robustness.py -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -3 $OUT/pytest.log
B="--no-cpu-baseline --no-extras"
for tag in ${TAGS:-new new}; do
  if [ $tag != new ]; then export GSR_LIB_TAG=$tag; else unset GSR_LIB_TAG; fi
  python3 bench.py --steps 20 --warmup 5 $B > $OUT/head_$tag.json 2>/dev/null
  python3 bench.py --steps 10 --warmup 3 $B --pose 0,0,-14 > $OUT/outside_$tag.json 2>/dev/null
  python3 bench.py --steps 10 --warmup 3 $B --width 3840 --height 2160 > $OUT/k4_$tag.json 2>/dev/null
  python3 - <<P
import json
for f in ("head","outside","k4"):
    d=json.load(open("$OUT/%s_$tag.json"%f))
    print("$tag",f,d["ms_per_step"],d["stage_ms"])
P
done
unset GSR_LIB_TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o run -- python3 bench.py --steps 10 --warmup 2 $B > $OUT/trace.log 2>&1
f=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
python3 scripts/kernel_stats_table.py $f 40 | grep -v vectorized | grep -v elementwise > $OUT/trace.txt
rm -rf $OUT/trace
cat $OUT/trace.txt | head -22
