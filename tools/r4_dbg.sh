timeout 900 python -m pytest tests/test_render_gpu.py tests/test_query_gpu.py -x -q 2>&1 | tail -3
timeout 900 python bench.py --no-cpu-baseline --no-train-leg --no-f32-anchor 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.strip().startswith('{')][0]); print(d['value'], d['ms_per_step'], d['roofline_query']['avg_launch_ms'], d['roofline_query']['frac'], d['stage_ms'])"
