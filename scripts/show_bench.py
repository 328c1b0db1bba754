"""dev: pretty-print the JSON line bench.py wrote (value, ms/step, per-kernel table)."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{d['value'] / 1e6:.1f} Mvox/s  {d['ms_per_step']:.3f} ms/step  roofline {d['roofline']['kernel']} frac {d['roofline']['frac']}")
for k, v in sorted(d.get("kernels", {}).items(), key=lambda kv: -kv[1]["ms_per_step"]):
    print(f"  {k:55s} x{v['launches_per_step']:4.0f} {v['ms_per_step']:7.3f} ms  {v.get('achieved_tflops', 0):6.1f} TFLOP/s")
if d.get("cpu_baseline"):
    print("cpu_baseline", d["cpu_baseline"])
