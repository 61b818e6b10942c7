for v in 0 1; do
MVF_STENCIL_TSPLIT=$v timeout 600 python bench.py --steps 10 --warmup 3 --per-layer --no-eager-compare --no-cpu-baseline --no-other-configs 2>&1 >/dev/null | grep -E "mvf " | sed "s/^/TSPLIT=$v /"
done
for v in 0 1 0 1; do
MVF_STENCIL_TSPLIT=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-eager-compare --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('train TSPLIT=$v', d['value'], d['ms_per_step'], 'mvf', d['roofline']['mvf']['ms_per_step'])"
done
