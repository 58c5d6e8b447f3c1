"""Folds the errors recorded by a GPU test run (FP_RECORD_BARS=<file>) into tests/golden/measured_bars.json: key -> the LARGEST error recorded for it in that run.
Keys the run did not touch KEEP their recorded bar (a partial run -- `-k`, one file -- must not drop the others into their loose fallback tolerances), and every
change is printed old -> new, so a bar that loosens shows up in the diff of the commit AND in the output of this script.
    FP_RECORD_BARS=gpurun_out/bars.jsonl python -m pytest tests -m gpu -q ; python tools/update_bars.py [gpurun_out/bars.jsonl] [--only-new]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
only_new = "--only-new" in sys.argv      # add keys that have no record yet, leave every recorded bar as it is
src = args[0] if args else os.path.join(ROOT, "gpurun_out", "bars.jsonl")
dst = os.path.join(ROOT, "tests", "golden", "measured_bars.json")
run = {}
for line in open(src):
    r = json.loads(line)
    run[r["key"]] = max(run.get(r["key"], 0.0), r["measured"])
old = json.load(open(dst)) if os.path.exists(dst) else {}
new = dict(old)
for k, v in sorted(run.items()):
    v = float(f"{v:.4g}")
    if k not in old:
        print(f"  new       {k}: {v:.4g}")
        new[k] = v
    elif only_new or v == old[k]:
        continue
    else:
        print(f"  {'LOOSENED' if v > old[k] else 'tightened'} {k}: {old[k]:.4g} -> {v:.4g}")
        new[k] = v
with open(dst, "w") as f:
    json.dump({k: new[k] for k in sorted(new)}, f, indent=1)
print(f"{len(run)} key(s) in this run, {len(new) - len(old)} new, {len(new)} bars -> {dst} ({len(old) - len(set(old) & set(run))} recorded key(s) untouched by this run kept)")
