python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_robustness.py tests/test_inria_profile.py -x -q 2>&1 | tail -3
for cfg in "" "--pose 0,0,-14" "--opacity-scale 0.1" "--pose 0,0,-14 --opacity-scale 0.1" "--scene stress --splats 50000000" "--width 3840 --height 2160"; do
    python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline $cfg 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cfg=[$cfg] ms', r['ms_per_step'], 'blend', r['stage_ms']['blend'], 'R_f', r['config']['records_staged'], r['config']['binning_plan'])"
done
