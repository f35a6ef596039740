for v in "ECB_KEEP_TABLE=1" "ECB_EC_CAP_LOG2=26" "ECB_EC_CAP_LOG2=23" "X=1"; do
 env $v timeout -k 5 120 python bench.py --workload c3 --steps 3 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$v ms_per_step=%.2f k_stream_ms=%.3f launches=%.1f ecs=%s' % (d['ms_per_step'], d['roofline']['kernel_ms_per_launch'], d['roofline']['launches_per_step'], d['config']['ecs']))
"
done
