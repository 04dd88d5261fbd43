set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/s10
mkdir -p $OUT
python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -15 $OUT/pytest.log
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python3 - <<P
import json
d=json.load(open("$OUT/bench.json"))
print(d["value"], d["ms_per_step"], d["stage_ms"])
for k in ("pose_outside","no_sorted_lists","blend_bound","blend_bound_pose_outside","forward_backward","forward_backward_no_sorted_lists"):
    e=d[k]; print(k, e["ms_per_step"], e["binning_plan"], e.get("blend_from_sorted_lists"), e["stage_ms"])
P
