cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_split.py tests/test_gpu_vit.py -q -m gpu -x 2>&1 | tail -3
python bench.py --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/bench_now.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_now.json'))
print(d['value'], d['ms_per_step'])
print(json.dumps(d.get('parity_mode'), indent=None)[:1500])
PY
