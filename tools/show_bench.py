#!/usr/bin/env python3
"""Prints the figures of a bench.py JSON line that matter when comparing builds: python3 tools/show_bench.py <file>"""
import json
import sys

d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
w = d["workloads"]
print(f"step {d['ms_per_step']:.4f} ms  value {d['value']:.0f}  roofline frac {d['roofline'].get('frac', 0):.4f} (kernel {d['roofline'].get('kernel_ms', 0):.4f} ms)  "
      f"allocs in steps {d.get('device_allocations_in_timed_steps')}")
if "clip" in w:
    print(f"clip {w['clip']['ms_per_step'] * 1e3:.2f} us")
if "filip" in w:
    f = w["filip"]
    print(f"filip {f['ms_per_step']:.4f} ms  backward {f.get('backward_ms', 0):.4f} ms  step_mfma_frac {f.get('step_mfma_frac', 0):.4f}")
for k, v in w.items():
    if k not in ("clip", "filip"):
        print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a != "note"})
if "cpu_baseline" in d:
    print("cpu_baseline", d["cpu_baseline"])
