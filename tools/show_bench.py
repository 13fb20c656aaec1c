"""Prints the headline fields of a bench.py JSON line: python tools/show_bench.py <file with the line last>"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print("value", d["value"], d["unit"], "ms_per_step", d["ms_per_step"])
print("stage_ms", d.get("stage_ms"))
for key in ("roofline", "roofline_query"):
    r = d.get(key) or {}
    print(key, {k: r.get(k) for k in ("achieved", "frac", "mfma_busy", "avg_launch_ms", "in_frame_ms", "traffic")})
for key in ("train_step", "train_step_sharded"):
    t = d.get(key) or {}
    print(key, {x: t.get(x) for x in ("ms_per_step", "captured_ms_per_step", "compute_ms", "step_form", "error")})
print("amortised", d.get("amortised_ms"))
c = d.get("cpu_baseline") or {}
print("cpu", c.get("value"), c.get("cores"), c.get("rays_gathering_another_pixel"))
