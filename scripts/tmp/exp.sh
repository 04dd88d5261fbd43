python -m pytest tests/test_inria_profile.py tests/test_ply.py tests/test_gpu_robustness.py -x -q 2>&1 | tail -3
for cfg in "--scene stress --splats 50000000 --semantics inria --sh-degree 3" "--semantics inria --sh-degree 3" "--scene stress --splats 50000000 --semantics inria --sh-degree 1"; do
    python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline $cfg 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cfg=[$cfg] ms', r['ms_per_step'], 'preprocess', r['stage_ms']['preprocess'], r['kernels']['preprocess'])"
done
