cd $GRAFT_REPO_ROOT
for v in 0 1 2 3 0 1 2 3; do echo "== MVF_WGRAD_BIG=$v"; MVF_WGRAD_BIG=$v python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], 'wgrad', r['wgrad']['ms_per_step'])"; done
