"""Folds the errors recorded by a GPU test run (FP_RECORD_BARS=<file>) into tests/golden/measured_bars.json: key -> the LARGEST error recorded for it.
    FP_RECORD_BARS=gpurun_out/bars.jsonl python -m pytest tests -m gpu -q ; python tools/update_bars.py [gpurun_out/bars.jsonl]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "bars.jsonl")
dst = os.path.join(ROOT, "tests", "golden", "measured_bars.json")
bars = {}
for line in open(src):
    r = json.loads(line)
    bars[r["key"]] = max(bars.get(r["key"], 0.0), r["measured"])
with open(dst, "w") as f:
    json.dump({k: float(f"{v:.4g}") for k, v in sorted(bars.items())}, f, indent=1)
print(f"{len(bars)} bars -> {dst}")
