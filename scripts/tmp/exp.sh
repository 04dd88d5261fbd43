python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3
for cfg in "" "--pose 0,0,-14" "--width 3840 --height 2160"; do
    python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline $cfg 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cfg=[$cfg] ms', r['ms_per_step'], r['stage_ms'])"
done
