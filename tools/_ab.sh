cd $GRAFT_REPO_ROOT
for v in 0 2 0 2; do
echo "== default variant $v"
FP_ATTN_DEFAULT_VARIANT=$v python bench.py --skip-probes --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
done
