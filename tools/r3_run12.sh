timeout 900 python -m pytest tests/test_conv_gpu.py -q -m gpu -p no:cacheprovider -k "stem_direct" 2>&1 | grep -v amdgpu.ids | tail -5
for v in 0 1 2 7 14; do
MVF_STEM_TPW=$v timeout 600 python bench.py --steps 10 --warmup 3 --per-layer --no-eager-compare --no-cpu-baseline --no-other-configs 2>&1 >/dev/null | grep -E "K147" | head -1 | sed "s/^/TPW=$v /"
done
for v in 0 1 7; do
MVF_STEM_TPW=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-eager-compare --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('train TPW=$v', d['value'], d['ms_per_step'])"
done
